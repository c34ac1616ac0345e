// gm_inst_bs20.hip -- K1v, the scalar-load bit-sliced Hamming kernel, instantiated for L = 20 only.
// Superseded by K1L (gm_lookup.hip.h); kept as the A/B baseline behind gm_set_variant (20..36, 5).
#include "gm_host.h"
#include "gm_bitslice.hip.h"

namespace {

constexpr int BS_GROUP_PAD = 4;   // n_groups is padded to a multiple of this (kernels take G | 4)

template <int L, bool NEAREST, int G, int NW>
int launch_bs_cfg(gm_index* idx, int64_t n_q, int k, int wps_cap, bool one_chunk, int32_t* ids, int32_t* dists,
                  int32_t* counts, hipStream_t st) {
    // Work split: one chunk = 64 queries, one wave takes c consecutive chunks.  All workgroups must be
    // resident at once (otherwise the last, partly filled round runs at low occupancy for a whole wave
    // lifetime), so c = ceil(chunks / resident wave slots).
    auto kern = gm::k_hamming_knn_bs<L, NEAREST, G, NW>;
    const int64_t chunks = (n_q + gm::WAVE - 1) / gm::WAVE;
    size_t lds = sizeof(uint32_t) * (size_t)k * gm::WAVE * NW;
    int per_cu = 0;     // resident workgroups per CU for this instantiation and list size
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern),
                                                          gm::WAVE * NW, lds));
    int wps = per_cu * NW / 4 > 0 ? per_cu * NW / 4 : 1;
    if (wps_cap > 0 && wps_cap < wps) wps = wps_cap;
    const int64_t slots = (int64_t)idx->n_simd * wps;
    int64_t cpw = one_chunk ? 1 : (chunks + slots - 1) / slots;
    const int64_t waves = (chunks + cpw - 1) / cpw;
    const int64_t blocks = (waves + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(gm::WAVE * NW), lds, st, idx->d_vert, (int)idx->n_db,
                       (int)idx->n_groups, idx->d_q, n_q, k, (int)cpw, ids, dists, counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L, bool NEAREST>
int launch_bs(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
              hipStream_t st) {
    const int v = idx->variant;
    if (L == 20) {   // A/B knobs (experiments): 2x = one chunk per wave, 3x = balanced chunks; x = config
        switch (v) {
            case 20: return launch_bs_cfg<20, NEAREST, 2, 1>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 21: return launch_bs_cfg<20, NEAREST, 1, 1>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 22: return launch_bs_cfg<20, NEAREST, 3, 1>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 23: return launch_bs_cfg<20, NEAREST, 2, 4>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 24: return launch_bs_cfg<20, NEAREST, 2, 8>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 25: return launch_bs_cfg<20, NEAREST, 1, 8>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 26: return launch_bs_cfg<20, NEAREST, 2, 16>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 27: return launch_bs_cfg<20, NEAREST, 4, 1>(idx, n_q, k, 0, true, ids, dists, counts, st);
            case 30: return launch_bs_cfg<20, NEAREST, 2, 1>(idx, n_q, k, 0, false, ids, dists, counts, st);
            case 33: return launch_bs_cfg<20, NEAREST, 2, 4>(idx, n_q, k, 0, false, ids, dists, counts, st);
            case 34: return launch_bs_cfg<20, NEAREST, 2, 8>(idx, n_q, k, 0, false, ids, dists, counts, st);
            case 35: return launch_bs_cfg<20, NEAREST, 1, 8>(idx, n_q, k, 0, false, ids, dists, counts, st);
            case 36: return launch_bs_cfg<20, NEAREST, 2, 16>(idx, n_q, k, 0, false, ids, dists, counts, st);
            default: break;
        }
    }
    return launch_bs_cfg<L, NEAREST, 2, 1>(idx, n_q, k, 0, true, ids, dists, counts, st);
}

}  // namespace

int gm_launch_bs20(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                   hipStream_t st) {
    return nearest ? launch_bs<20, true>(idx, n_q, k, ids, dists, counts, st)
                   : launch_bs<20, false>(idx, n_q, k, ids, dists, counts, st);
}

int gm_build_vertical(gm_index* idx, hipStream_t st) {
    const int W = gm::vert_stride(idx->L);
    idx->n_groups = round_up((idx->n_db + gm::GROUP - 1) / gm::GROUP, BS_GROUP_PAD);
    if (hipMalloc(&idx->d_vert, sizeof(uint32_t) * idx->n_groups * W) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "hipMalloc(db bit-sliced groups) failed");
    int64_t words = idx->n_groups * W;
    hipLaunchKernelGGL(gm::k_to_vertical, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, idx->d_db,
                       idx->n_db, idx->d_vert, idx->n_groups, idx->L, W);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
