// gm_inst_lk.hip -- instantiates the K1L look-up Hamming kernels for ONE guide length (-DGM_LEN=<L>).
#include "gm_host.h"
#include "gm_lookup.hip.h"
#include "gm_pigeon_scan.hip.h"

#ifndef GM_LEN
#error "compile with -DGM_LEN=<guide length>"
#endif

namespace {

template <int L, int KCAP, bool NEAREST, bool SEEDED>
int launch_lk_cfg(gm_index* idx, const gm_lk_args& a, hipStream_t st) {
    const int64_t chunks = (a.n_q + gm::WAVE - 1) / gm::WAVE;
    size_t lds = 2 * (size_t)gm::lk_tile_bytes(L) + (KCAP == 0 ? sizeof(uint32_t) * (size_t)a.k * gm::WAVE : 0);
    // (Capping the occupancy to make the number of workgroup rounds integral was measured and does not
    // pay: throughput rises with resident waves up to the VGPR/LDS limit -- profiles/r01/NOTES.md.)
    auto kern = gm::k_hamming_knn_lk<L, KCAP, NEAREST, SEEDED>;
    // (once per instantiation, device and size: the call costs ~5 us of host time, and the re-scan launches sit on the
    // critical path right after a read-back)
    static size_t lds_set[64] = {};
    const int dev = idx->device >= 0 && idx->device < 64 ? idx->device : 0;
    if (lds > lds_set[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)chunks, (unsigned)a.n_seg), dim3(gm::WAVE), lds, st, idx->d_onehot,
                       (int)idx->n_db, a.n_tiles > 0 ? a.n_tiles : (int)idx->n_lk_tiles, a.tiles_per_seg, a.seg_base, a.q ? a.q : idx->d_q, a.n_q, a.k,
                       a.seed_keys, a.out_keys, a.ids, a.dists, a.counts, a.n_dev, a.seed_all, a.q_rows);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L, int KCAP, bool NEAREST>
int launch_lk_seed(gm_index* idx, const gm_lk_args& a, hipStream_t st) {
    return a.seed_keys || a.seed_all ? launch_lk_cfg<L, KCAP, NEAREST, true>(idx, a, st)
                       : launch_lk_cfg<L, KCAP, NEAREST, false>(idx, a, st);
}

// list capacity by k: registers for k <= 20 (the reference CLI's range is 2..20, cli.py:55-56), LDS beyond
template <int L>
int launch_lk(gm_index* idx, const gm_lk_args& a, hipStream_t st) {
    switch (a.kcap) {
        case 1: return launch_lk_seed<L, 1, true>(idx, a, st);
        case 5: return launch_lk_seed<L, 5, false>(idx, a, st);
        case 10: return launch_lk_seed<L, 10, false>(idx, a, st);
        case 20: return launch_lk_seed<L, 20, false>(idx, a, st);
        case 0: return launch_lk_cfg<L, 0, false, false>(idx, a, st);
        default: return gm_fail(GM_ERR_ARG, "look-up kernel: unsupported list capacity");
    }
}

template <int L>
int launch_pg_cfg(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks, const int* c_begin,
                  const int* c_end, int64_t grid, int64_t n_q, int k, int thr0, uint32_t* lists, hipStream_t st) {
    (void)idx;
    size_t lds = gm::pg_lds_bytes(L);
#ifdef GM_PG_LDS_PAD      // occupancy experiments (timing builds): LDS bytes per workgroup, e.g. 6826 = 6 waves per SIMD, 10240 = 4
    if (lds < (size_t)GM_PG_LDS_PAD) lds = (size_t)GM_PG_LDS_PAD;
#endif
    if (blocks->cut_id != 0xffffffffu)
        hipLaunchKernelGGL((gm::k_hamming_scan_pg<L, true>), dim3((unsigned)grid), dim3(gm::WAVE), lds, st, *blocks, chunks,
                           c_begin, c_end, idx->d_q, n_q, k, thr0, lists);
    else
        hipLaunchKernelGGL((gm::k_hamming_scan_pg<L, false>), dim3((unsigned)grid), dim3(gm::WAVE), lds, st, *blocks, chunks,
                           c_begin, c_end, idx->d_q, n_q, k, thr0, lists);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L, int KCAP>
int launch_pg4_kc(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks, const int* c_begin,
                  const int* c_end, int64_t grid, int64_t n_q, int k, int thr0, uint32_t* keys, hipStream_t st) {
    const size_t lds = gm::pg4_lds_bytes(L);
    if (blocks->cut_id != 0xffffffffu)
        hipLaunchKernelGGL((gm::k_hamming_scan_pg4<L, true, KCAP>), dim3((unsigned)grid), dim3(gm::WAVE), lds, st, *blocks, chunks,
                           c_begin, c_end, idx->d_q, n_q, k, thr0, keys);
    else
        hipLaunchKernelGGL((gm::k_hamming_scan_pg4<L, false, KCAP>), dim3((unsigned)grid), dim3(gm::WAVE), lds, st, *blocks, chunks,
                           c_begin, c_end, idx->d_q, n_q, k, thr0, keys);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int L>
int launch_pg3_cfg(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks, const int* c_begin,
                   const int* c_end, int64_t grid, int64_t n_q, int k, int kcap, int thr0, uint32_t* keys, hipStream_t st) {
    switch (kcap) {
        case 1: return launch_pg4_kc<L, 1>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, thr0, keys, st);
        case 5: return launch_pg4_kc<L, 5>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, thr0, keys, st);
        case 10: return launch_pg4_kc<L, 10>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, thr0, keys, st);
        case 20: return launch_pg4_kc<L, 20>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, thr0, keys, st);
        default: return gm_fail(GM_ERR_ARG, "pigeonhole kernel: unsupported list capacity");
    }
}

}  // namespace

#define GM_CAT2(a, b) a##b
#define GM_CAT(a, b) GM_CAT2(a, b)

int GM_CAT(gm_launch_lk_L, GM_LEN)(gm_index* idx, const gm_lk_args* a, hipStream_t st) {
    return launch_lk<GM_LEN>(idx, *a, st);
}

int GM_CAT(gm_launch_pg_L, GM_LEN)(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks,
                                   const int* c_begin, const int* c_end, int64_t grid, int64_t n_q, int k, int thr0,
                                   uint32_t* lists, hipStream_t st) {
    return launch_pg_cfg<GM_LEN>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, thr0, lists, st);
}

int GM_CAT(gm_launch_pg3_L, GM_LEN)(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks,
                                    const int* c_begin, const int* c_end, int64_t grid, int64_t n_q, int k, int kcap,
                                    int thr0, uint32_t* keys, hipStream_t st) {
    return launch_pg3_cfg<GM_LEN>(idx, blocks, chunks, c_begin, c_end, grid, n_q, k, kcap, thr0, keys, st);
}
