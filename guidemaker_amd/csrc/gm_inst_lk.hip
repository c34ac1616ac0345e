// gm_inst_lk.hip -- instantiates the K1L look-up Hamming kernels for ONE guide length (-DGM_LEN=<L>).
#include "gm_host.h"
#include "gm_lookup.hip.h"

#ifndef GM_LEN
#error "compile with -DGM_LEN=<guide length>"
#endif

namespace {

template <int L, int KCAP, bool NEAREST>
int launch_lk_cfg(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                  hipStream_t st) {
    const int64_t blocks = (n_q + gm::WAVE - 1) / gm::WAVE;
    size_t lds = 2 * (size_t)gm::lk_tile_bytes(L) + (KCAP == 0 ? sizeof(uint32_t) * (size_t)k * gm::WAVE : 0);
    // (Capping the occupancy to make the number of workgroup rounds integral was measured and does not
    // pay: throughput rises with resident waves up to the VGPR/LDS limit -- profiles/r01/NOTES.md.)
    if (idx->variant >= 40 && idx->variant < 50) {     // A/B knob: cap resident waves per SIMD via LDS padding
        const size_t pad = (size_t)160 * 1024 / (4 * (size_t)(idx->variant - 40 > 0 ? idx->variant - 40 : 1)) / 512 * 512;
        if (pad > lds) lds = pad;
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gm::k_hamming_knn_lk<L, KCAP, NEAREST>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gm::k_hamming_knn_lk<L, KCAP, NEAREST>), dim3((unsigned)blocks), dim3(gm::WAVE), lds, st,
                       idx->d_onehot, (int)idx->n_db, (int)idx->n_lk_tiles, idx->d_q, n_q, k, ids, dists, counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

// list capacity by k: registers for k <= 20 (the reference CLI's range is 2..20, cli.py:55-56), LDS beyond
template <int L, bool NEAREST>
int launch_lk(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
              hipStream_t st) {
    if (NEAREST) return launch_lk_cfg<L, 1, true>(idx, n_q, 1, ids, dists, counts, st);
    if (idx->variant == 7) return launch_lk_cfg<L, 0, false>(idx, n_q, k, ids, dists, counts, st);   // A/B: LDS lists
    if (k <= 4) return launch_lk_cfg<L, 4, false>(idx, n_q, k, ids, dists, counts, st);
    if (k <= 10) return launch_lk_cfg<L, 10, false>(idx, n_q, k, ids, dists, counts, st);
    if (k <= 20) return launch_lk_cfg<L, 20, false>(idx, n_q, k, ids, dists, counts, st);
    return launch_lk_cfg<L, 0, false>(idx, n_q, k, ids, dists, counts, st);
}

}  // namespace

#define GM_CAT2(a, b) a##b
#define GM_CAT(a, b) GM_CAT2(a, b)

int GM_CAT(gm_launch_lk_L, GM_LEN)(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                                   int32_t* counts, hipStream_t st) {
    return nearest ? launch_lk<GM_LEN, true>(idx, n_q, k, ids, dists, counts, st)
                   : launch_lk<GM_LEN, false>(idx, n_q, k, ids, dists, counts, st);
}
