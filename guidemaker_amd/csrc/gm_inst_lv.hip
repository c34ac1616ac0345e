// gm_inst_lv.hip -- instantiates the K2L Levenshtein scan kernels for ONE guide length (-DGM_LEN=<L>).
#include "gm_host.h"
#include "gm_leven_bs.hip.h"

#ifndef GM_LEN
#error "compile with -DGM_LEN=<guide length>"
#endif

namespace {

template <int L, int KCAP>
int launch_lv_cfg(gm_index* idx, int64_t n_q, int k, int n_seg, int tiles_per_seg, const int32_t* seed_d,
                  const int32_t* seed_c, int seed_k, uint32_t* keys, hipStream_t st, const gm_lv_opt& o) {
    const int64_t chunks = (n_q + gm::WAVE - 1) / gm::WAVE;
    const size_t lds = 2 * (size_t)gm::LV_TILE * 16;
    hipLaunchKernelGGL((gm::k_leven_knn_lk<L, KCAP>), dim3((unsigned)chunks, (unsigned)n_seg), dim3(gm::WAVE), lds, st,
                       idx->d_peq, (int)idx->n_db, (int)idx->n_lv_tiles, tiles_per_seg, o.q ? o.q : idx->d_q, n_q, k, seed_d,
                       seed_c, seed_k, keys);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L, int KCAP>
int launch_lvb_cfg(gm_index* idx, int64_t n_q, int k, int n_seg, int tiles_per_seg, const int32_t* seed_d,
                   const int32_t* seed_c, int seed_k, uint32_t* keys, hipStream_t st, const gm_lv_opt& o) {
    const int64_t chunks = (n_q + gm::WAVE - 1) / gm::WAVE;
    const size_t lds = 2 * (size_t)gm::lk_tile_bytes(L);
    // band-only workgroups first (the bulk once the bounds are seeded), then the workgroups that need the full matrix
    hipLaunchKernelGGL((gm::k_leven_knn_bs<L, KCAP, false>), dim3((unsigned)chunks, (unsigned)n_seg), dim3(gm::WAVE), lds, st,
                       idx->d_onehot, (int)idx->n_db, (int)idx->n_lk_tiles, tiles_per_seg, o.q ? o.q : idx->d_q, n_q, k, seed_d,
                       seed_c, seed_k, keys, o.cap_cnt, o.cap_thr);
    hipLaunchKernelGGL((gm::k_leven_knn_bs<L, KCAP, true>), dim3((unsigned)chunks, (unsigned)n_seg), dim3(gm::WAVE), lds, st,
                       idx->d_onehot, (int)idx->n_db, (int)idx->n_lk_tiles, tiles_per_seg, o.q ? o.q : idx->d_q, n_q, k, seed_d,
                       seed_c, seed_k, keys, o.cap_cnt, o.cap_thr);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L, int KCAP>
int launch_lvp_cfg(const gm_lp_copies& P, const gm_lp_chunk* chunks, int64_t n_chunks, const uint2* q, int64_t n_q, int k,
                   const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys, int radius, const int* n_chunks_dev,
                   hipStream_t st) {
    const size_t lds = 2 * (size_t)gm::lk_tile_bytes(L);
    hipLaunchKernelGGL((gm::k_leven_part_bs<L, KCAP, false>), dim3((unsigned)n_chunks), dim3(gm::WAVE), lds, st, P, chunks, q, n_q, k,
                       seed_d, seed_c, seed_k, keys, radius, n_chunks_dev);
    // (every bound is <= radius + 1: the full-matrix launch has work only when that exceeds what the band serves)
    if (radius + 1 > 2 * gm::lvb_band(L) + 2)
        hipLaunchKernelGGL((gm::k_leven_part_bs<L, KCAP, true>), dim3((unsigned)n_chunks), dim3(gm::WAVE), lds, st, P, chunks, q, n_q, k,
                           seed_d, seed_c, seed_k, keys, radius, n_chunks_dev);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

}  // namespace

#define GM_CAT2(a, b) a##b
#define GM_CAT(a, b) GM_CAT2(a, b)

// algo 0 = K2L (Myers per pair, match masks from LDS), 1 = K2B (bit-sliced DP over 32 targets per word)
int GM_CAT(gm_launch_lv_L, GM_LEN)(gm_index* idx, int algo, int kcap, int64_t n_q, int k, int n_seg, int tiles_per_seg,
                                   const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys,
                                   hipStream_t st, const gm_lv_opt* opt) {
    const gm_lv_opt o = opt ? *opt : gm_lv_opt{};
    if (algo == 1) {
        switch (kcap) {
            case 1: return launch_lvb_cfg<GM_LEN, 1>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
            case 10: return launch_lvb_cfg<GM_LEN, 10>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
            case 20: return launch_lvb_cfg<GM_LEN, 20>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
            default: return gm_fail(GM_ERR_ARG, "leven bit-sliced kernel: unsupported list capacity");
        }
    }
    switch (kcap) {
        case 1: return launch_lv_cfg<GM_LEN, 1>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
        case 10: return launch_lv_cfg<GM_LEN, 10>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
        case 20: return launch_lv_cfg<GM_LEN, 20>(idx, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, o);
        default: return gm_fail(GM_ERR_ARG, "leven look-up kernel: unsupported list capacity");
    }
}

// K2P: the partitioned Levenshtein scan (gm_levpart.hip)
int GM_CAT(gm_launch_lvp_L, GM_LEN)(const gm_lp_copies* P, const gm_lp_chunk* chunks, int64_t n_chunks, const uint2* q,
                                    int64_t n_q, int k, int kcap, const int32_t* seed_d, const int32_t* seed_c, int seed_k,
                                    uint32_t* keys, int radius, const int* n_chunks_dev, hipStream_t st) {
    switch (kcap) {
        case 10: return launch_lvp_cfg<GM_LEN, 10>(*P, chunks, n_chunks, q, n_q, k, seed_d, seed_c, seed_k, keys, radius, n_chunks_dev, st);
        case 20: return launch_lvp_cfg<GM_LEN, 20>(*P, chunks, n_chunks, q, n_q, k, seed_d, seed_c, seed_k, keys, radius, n_chunks_dev, st);
        default: return gm_fail(GM_ERR_ARG, "partitioned leven kernel: unsupported list capacity");
    }
}
