// gmhip.hip -- C ABI (include/gmhip.h) over the gfx950 kernels in gm_kernels.hip.h.
// Host side of libgmhip.so: handle management, HBM workspaces, launches, HIP-event timing.
#include "gm_host.h"
#include "gm_kernels.hip.h"
#include "gm_lookup.hip.h"   // k_to_onehot only; the scan kernels are instantiated in gm_inst_lk.hip
#include "gm_leven.hip.h"    // k_to_peq, k_leven_merge; the scan kernels are instantiated in gm_inst_lv.hip

#include <atomic>
#include <charconv>
#include <cstdlib>
#include <utility>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace {
thread_local std::string g_err;
}

int gm_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define fail gm_fail

namespace {

int ensure_q(gm_index* idx, int64_t n_q) {
    if (n_q <= idx->q_cap) return GM_OK;
    if (idx->d_q) (void)hipFree(idx->d_q);
    idx->d_q = nullptr;
    idx->q_cap = 0;
    int64_t cap = round_up(n_q, 4096);
    if (hipMalloc(&idx->d_q, sizeof(uint2) * cap) != hipSuccess)
        return fail(GM_ERR_ALLOC, "hipMalloc(query planes) failed");
    idx->q_cap = cap;
    return GM_OK;
}

int ensure_stage(gm_index* idx, int64_t bytes) {
    if (bytes <= idx->stage_cap) return GM_OK;
    if (idx->d_stage) (void)hipFree(idx->d_stage);
    idx->d_stage = nullptr;
    idx->stage_cap = 0;
    int64_t cap = round_up(bytes, 1 << 20);
    if (hipMalloc(&idx->d_stage, cap) != hipSuccess) return fail(GM_ERR_ALLOC, "hipMalloc(staging) failed");
    idx->stage_cap = cap;
    return GM_OK;
}

int planes_launch(const uint64_t* d_in, uint2* d_out, int64_t n, int64_t n_pad, hipStream_t st) {
    if (n_pad <= 0) return GM_OK;
    int64_t blocks = (n_pad + 255) / 256;
    hipLaunchKernelGGL(gm::k_to_planes, dim3((unsigned)blocks), dim3(256), 0, st, d_in, d_out, n, n_pad);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

// queries per lane by list size: keeps the per-wave list (k * R * 256 B) near 10 KiB
int pick_r(int k) { return k <= 10 ? 4 : (k <= 20 ? 2 : 1); }

template <int R>
int launch_hamming_knn(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                       hipStream_t st) {
    int64_t blocks = (n_q + R * gm::WAVE - 1) / (R * gm::WAVE);
    size_t lds = sizeof(uint2) * 2 * gm::TILE + sizeof(uint32_t) * (size_t)k * R * gm::WAVE;
    int n_tiles = (int)(idx->n_db_pad / gm::TILE);
    auto kern = gm::k_hamming_knn<R, 8>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(gm::WAVE), lds, st, idx->d_db, (int)idx->n_db,
                       n_tiles, idx->d_q, n_q, k, ids, dists, counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

template <int R>
int launch_leven_knn(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                     hipStream_t st) {
    int64_t blocks = (n_q + R * gm::WAVE - 1) / (R * gm::WAVE);
    size_t lds = sizeof(uint2) * 2 * gm::TILE + sizeof(uint32_t) * (size_t)k * R * gm::WAVE;
    int n_tiles = (int)(idx->n_db_pad / gm::TILE);
    auto kern = gm::k_leven_knn<R>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(gm::WAVE), lds, st, idx->d_db, (int)idx->n_db,
                       n_tiles, idx->d_q, n_q, k, idx->L, ids, dists, counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

int launch_lk_len(gm_index* idx, const gm_lk_args& a, hipStream_t st) {
    switch (idx->L) {
#define GM_CASE(LEN) case LEN: return gm_launch_lk_L##LEN(idx, &a, st);
        GM_FOR_EACH_LEN(GM_CASE)
#undef GM_CASE
        default: return fail(GM_ERR_ARG, "look-up kernel: guidelen out of range");
    }
}

int timing_begin(gm_index* idx, hipStream_t st);
int timing_end(gm_index* idx, hipStream_t st, int64_t pairs);

int ensure_u32(uint32_t** buf, int64_t* cap, int64_t need) {
    if (need <= *cap) return GM_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    if (hipMalloc(buf, sizeof(uint32_t) * need) != hipSuccess)
        return fail(GM_ERR_ALLOC, "hipMalloc(segment lists) failed");
    *cap = need;
    return GM_OK;
}

int launch_merge(int kcap, const uint32_t* keys, int n_seg, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                 int32_t* counts, bool nearest, hipStream_t st, const int* n_dev = nullptr) {
    const int nr = nearest ? 1 : 0;
    if (n_seg >= 128 && n_q <= 8192) {      // a handful of candidates, hundreds of lists: a whole wave per candidate
        constexpr int SP = 64;
        const dim3 grid((unsigned)((n_q * SP + 255) / 256)), block(256);
        switch (kcap) {
            case 1: hipLaunchKernelGGL((gm::k_merge_lists_split<1, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            case 5: hipLaunchKernelGGL((gm::k_merge_lists_split<5, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            case 10: hipLaunchKernelGGL((gm::k_merge_lists_split<10, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            default: hipLaunchKernelGGL((gm::k_merge_lists_split<20, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
        }
        HIP_TRY(hipGetLastError());
        return GM_OK;
    }
    if (n_seg >= 16 && n_q <= 65536) {      // few candidates, many lists: 16 lanes per candidate
        constexpr int SP = 16;
        const dim3 grid((unsigned)((n_q * SP + 255) / 256)), block(256);
        switch (kcap) {
            case 1: hipLaunchKernelGGL((gm::k_merge_lists_split<1, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            case 5: hipLaunchKernelGGL((gm::k_merge_lists_split<5, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            case 10: hipLaunchKernelGGL((gm::k_merge_lists_split<10, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
            default: hipLaunchKernelGGL((gm::k_merge_lists_split<20, SP>), grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
        }
        HIP_TRY(hipGetLastError());
        return GM_OK;
    }
    const dim3 grid((unsigned)((n_q + 255) / 256)), block(256);
    switch (kcap) {
        case 1: hipLaunchKernelGGL(gm::k_merge_lists<1>, grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
        case 5: hipLaunchKernelGGL(gm::k_merge_lists<5>, grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
        case 10: hipLaunchKernelGGL(gm::k_merge_lists<10>, grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
        default: hipLaunchKernelGGL(gm::k_merge_lists<20>, grid, block, 0, st, keys, n_seg, n_q, k, ids, dists, counts, nr, n_dev); break;
    }
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

// K1L Hamming scan of the candidates in idx->d_q.  nearest: k = 1, distances only (into dists).
// timed: record the HIP events around the dominant launch (false when used as the Levenshtein seeder).

// guess_thr > 0 (K1P's re-scan): the caller knows a distance bound that almost surely holds for these rows;
// every segment is scanned in ONE seeded launch under it (no unseeded first phase).  A row whose list does
// not fill under the guess is NOT exact -- the caller checks the counts and repeats without a guess.
// n_expect >= 0 (speculative follow-ups): n_q is an upper bound the launch is sized for, about n_expect rows are real --
// the pool is cut for THAT many (S4: 249 rows under a bound of 2 359 got a tenth of the segments they fill the chip with).
int run_hamming_lookup(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                       int32_t* counts, bool timed, hipStream_t st, const uint2* qplanes = nullptr,
                       int guess_thr = 0, int64_t n_expect = -1, const int32_t* q_rows = nullptr) {
    gm_lk_args a;
    a.q = qplanes;
    a.q_rows = q_rows;
    a.n_dev = qplanes != nullptr ? idx->dev_rows : nullptr;      // (K1P's re-scan of compact rows in a speculative run)
    a.kcap = nearest ? 1 : (idx->variant == 7 || k > 20 ? 0 : (k <= 5 ? 5 : (k <= 10 ? 10 : 20)));
    a.n_q = n_q;
    a.k = nearest ? 1 : k;
    a.ids = ids;
    a.dists = dists;
    a.counts = counts;
    const int64_t n_real = n_expect >= 0 && n_expect < n_q ? (n_expect > 0 ? n_expect : 1) : n_q;
    const int64_t chunks = (n_real + gm::WAVE - 1) / gm::WAVE;      // (the chunks that do work; the grid below covers n_q)
    const int64_t n_tiles = idx->n_lk_tiles;                       // even
    // work items wanted per resident wave slot: more items = smaller tail round, but every extra segment
    // re-collects up to k keys and lengthens the merge, so fewer for long lists (measured on S1-sized
    // pools, profiles/r01/NOTES.md: best segment count 32+ / 8 / 6 / 4 for list capacity 1 / 4 / 10 / 20)
    const int per_slot = a.kcap == 1 ? 40 : (a.kcap == 5 ? 14 : (a.kcap == 10 ? 10 : 7));
    const int64_t want_items = (int64_t)idx->n_simd * 5 * per_slot;
    int64_t n_seg = (want_items + chunks - 1) / chunks;
    if (n_seg > n_tiles / 2) n_seg = n_tiles / 2;
    // the merge is one thread per candidate walking n_seg lists: with few candidates it is the serial part,
    // so long lists get fewer segments (5 000 candidates, k = 10: 512 segments merged in 1.2 ms, 64 in 0.1 ms)
    int64_t seg_cap = a.kcap == 1 ? 512 : (a.kcap == 5 ? 128 : 64);      // (lists x keys the merge walks)
    // a handful of candidates (K1P's re-scan on large pools: tens of rows): the scan itself is the serial part, so
    // cut the pool finer -- enough (chunk, segment) workgroups for one per wave slot; the 16-lanes-per-candidate merge
    // walks them in a few microseconds
    const int64_t fill = (int64_t)idx->n_simd * 8 / (chunks > 0 ? chunks : 1);
    if (fill > seg_cap) seg_cap = fill < 512 ? fill : 512;
    if (n_seg > seg_cap) n_seg = seg_cap;
    while (n_seg > 2 && (n_seg + 1) * n_q * (a.kcap > 0 ? a.kcap : 1) > ((int64_t)1 << 26)) n_seg /= 2;      // (segment lists: <= 256 MiB)
    if (guess_thr > 0 && idx->lk_seg > 0) n_seg = idx->lk_seg < n_tiles / 2 ? idx->lk_seg : n_tiles / 2;
    int rc;
    if (a.kcap == 0 || n_seg < 2) {           // single pass over the whole pool
        a.n_seg = 1;
        a.tiles_per_seg = (int)n_tiles;
        if (timed && (rc = timing_begin(idx, st))) return rc;
        if (timed) idx->last_kernel = nearest ? "k_hamming_knn_lk(nearest)" : "k_hamming_knn_lk";
        if ((rc = launch_lk_len(idx, a, st))) return rc;
        return timed ? timing_end(idx, st, n_q * idx->n_db) : GM_OK;
    }
    // two-phase segmented scan: segment 0 unseeded, the others seeded from it, then merge
    int64_t tps = (n_tiles + n_seg - 1) / n_seg;
    tps += tps & 1;
    n_seg = (n_tiles + tps - 1) / tps;
    if ((rc = ensure_u32(&idx->d_keys, &idx->keys_cap, (n_seg + 1) * n_q * a.kcap))) return rc;
    a.tiles_per_seg = (int)tps;
    a.out_keys = idx->d_keys;
    if (guess_thr > 0) {      // one seeded launch over all segments under the caller's bound
        a.n_seg = (int)n_seg;
        a.seg_base = 0;
        a.seed_all = ((uint32_t)guess_thr << 27) + 1u;
        if ((rc = launch_lk_len(idx, a, st))) return rc;
        return launch_merge(a.kcap, idx->d_keys, (int)n_seg, n_q, a.k, ids, dists, counts, nearest, st, a.n_dev);
    }
    a.n_seg = 1;
    a.seg_base = 0;
    if ((rc = launch_lk_len(idx, a, st))) return rc;               // phase A
    a.n_seg = (int)n_seg - 1;
    a.seg_base = 1;
    a.seed_keys = idx->d_keys;
    if (timed && (rc = timing_begin(idx, st))) return rc;
    if (timed) idx->last_kernel = nearest ? "k_hamming_knn_lk(nearest,seeded)" : "k_hamming_knn_lk(seeded)";
    if ((rc = launch_lk_len(idx, a, st))) return rc;               // phase B: the dominant launch
    const int64_t seg0 = tps * gm::LK_TG * gm::LK_GROUP;          // pool targets covered by segment 0
    const int64_t in_b = idx->n_db > seg0 ? idx->n_db - seg0 : 0;
    if (timed && (rc = timing_end(idx, st, n_q * in_b))) return rc;
    return launch_merge(a.kcap, idx->d_keys, (int)n_seg, n_q, a.k, ids, dists, counts, nearest, st, a.n_dev);
}

int launch_lv_by_len(gm_index* idx, int algo, int kcap, int64_t n_q, int k, int n_seg, int tiles_per_seg,
                     const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys, hipStream_t st,
                     const gm_lv_opt* opt = nullptr) {
    switch (idx->L) {
#define GM_CASE(LEN) \
    case LEN: return gm_launch_lv_L##LEN(idx, algo, kcap, n_q, k, n_seg, tiles_per_seg, seed_d, seed_c, seed_k, keys, st, opt);
        GM_FOR_EACH_LEN(GM_CASE)
#undef GM_CASE
        default: return fail(GM_ERR_ARG, "leven look-up kernel: guidelen out of range");
    }
}

// K2L pipeline: (1) Hamming k-NN with K1L -> seeds (lev <= hamming) written into the caller's output
// buffers, (2) segmented Levenshtein scan into scratch lists, (3) merge into the output buffers.
// The query planes are already in idx->d_q.  d1 != nullptr: nearest-distance form (k = 1).
bool use_leven_lookup(const gm_index* idx, int k) {
    return idx->metric == GM_METRIC_LEVEN && idx->d_peq != nullptr && idx->d_onehot != nullptr && k <= 20 &&
           (idx->variant == 0 || idx->variant == 2);      // 0 = K2B bit-sliced DP, 2 = K2L (A/B baseline)
}

int run_hamming_pigeon(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                       int32_t* counts, hipStream_t st, bool* used, bool timed);

// Levenshtein narrow-band first pass: every candidate's seeded bound (Hamming k-th distance + 1; L + 1 without k
// Hamming neighbours) is kept for the second pass, and the candidates seeded at most one above cap_thr are counted
__global__ __launch_bounds__(256) void k_lev_seed_stats(const int32_t* __restrict__ dists, const int32_t* __restrict__ counts,
                                                        int64_t n, int k, int L, int cap_thr, uint32_t* __restrict__ thr_out,
                                                        int* __restrict__ cnt) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int thr = counts[i] >= k ? dists[i * k + (k - 1)] + 1 : L + 1;
        thr_out[i] = (uint32_t)thr;
        if (thr <= cap_thr + 1) atomicAdd(&s_cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt != 0) atomicAdd(cnt, s_cnt);
}

// seeds of the second pass: the kept bound - 1 (the kernel adds the 1 back) of the rows listed in inc
__global__ __launch_bounds__(256) void k_lev_gather_seeds(const uint32_t* __restrict__ thr_all, const int32_t* __restrict__ inc,
                                                          int64_t n, int32_t* __restrict__ seeds) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) seeds[j] = (int32_t)thr_all[inc[j]] - 1;
}

// ---- K2P (gm_levpart.hip): the pigeonhole partition for Levenshtein ----
// every stride-th candidate with its Hamming seeds, compacted (the planner's sample)
__global__ __launch_bounds__(256) void k_lev_sample_rows(const uint2* __restrict__ q, const int32_t* __restrict__ dists,
                                                         const int32_t* __restrict__ counts, int64_t stride, int64_t n_s, int k,
                                                         uint2* __restrict__ q_s, int32_t* __restrict__ d_s, int32_t* __restrict__ c_s) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_s * k) return;
    const int64_t j = t / k;
    const int s = (int)(t % k);
    if (dists != nullptr) d_s[t] = dists[j * stride * k + s];
    if (s == 0) {
        q_s[j] = q[j * stride];
        if (counts != nullptr) c_s[j] = counts[j * stride];
    }
}

// rows the partition of radius r cannot vouch for (list not full, or its k-th entry beyond r): count 0 = "take the second pass"
__global__ __launch_bounds__(256) void k_lev_unvouched(int32_t* __restrict__ counts, const int32_t* __restrict__ dists, int64_t n,
                                                       int k, int r) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (counts[i] < k || dists[i * k + (k - 1)] > r)) counts[i] = 0;
}

// Partitioned first pass of run_leven_lookup: ids / dists / counts receive the partition's result (*used); rows it cannot
// vouch for come back with count 0 and go on to the caller's further passes.  It runs UNSEEDED: its bounds are clamped
// to radius + 1 anyway, so the Hamming k-NN of all candidates (the seeds of the plain scan) would buy nothing -- only the
// planner's sample and the rows that reach the plain pass get Hamming seeds.  The radius is chosen from a sample: 1 024
// candidates through the plain K2B scan (under their own Hamming seeds) give the distribution of the k-th EDIT distance,
// and r minimises the modelled cost below.
int run_leven_partition(gm_index* idx, int64_t n_q, int k, int kcap, int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st,
                        bool* used, int* r_used, int* r2_used) {
    *used = false;
    *r2_used = 0;
    const int tune = gm_lp_tune_radius(idx);
    if (tune < 0 || (kcap != 10 && kcap != 20) || idx->L < 12) return GM_OK;
    int rc;
    int r = tune, r2 = 0;
    if (r == 0) {
        if (idx->n_db < 65536 || n_q < 8192) return GM_OK;
        int64_t p_nq = -1;
        int p_k = -1, p_r = 0, p_r2 = 0;
        gm_lp_plan_get(idx, &p_nq, &p_k, &p_r, &p_r2);
        if (p_nq != n_q || p_k != k) {
            double p_match = 0.25;
            if ((rc = gm_pg_pool_stats(idx, st, &p_match))) return rc;
            const int64_t n_s = 1024, stride = n_q / n_s;
            uint2* q_s = nullptr;
            int32_t *d_s = nullptr, *c_s = nullptr, *i_s = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&q_s), sizeof(uint2) * n_s) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&d_s), sizeof(int32_t) * n_s * k) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&i_s), sizeof(int32_t) * n_s * k) != hipSuccess ||
                hipMalloc(reinterpret_cast<void**>(&c_s), sizeof(int32_t) * n_s) != hipSuccess) {
                (void)hipFree(q_s); (void)hipFree(d_s); (void)hipFree(i_s); (void)hipFree(c_s);
                return fail(GM_ERR_ALLOC, "hipMalloc(K2P sample) failed");
            }
            hipLaunchKernelGGL(k_lev_sample_rows, dim3((unsigned)((n_s * k + 255) / 256)), dim3(256), 0, st, idx->d_q,
                               (const int32_t*)nullptr, (const int32_t*)nullptr, stride, n_s, k, q_s, d_s, c_s);
            // (the sample's seeds: the exact Hamming k-NN of these 1 024 rows, lev <= hamming)
            if ((rc = run_hamming_lookup(idx, false, n_s, k, i_s, d_s, c_s, false, st, q_s))) {
                (void)hipFree(q_s); (void)hipFree(d_s); (void)hipFree(i_s); (void)hipFree(c_s);
                return rc;
            }
            const int64_t n_tiles = idx->n_lk_tiles;
            int64_t want = 64 < n_tiles ? 64 : n_tiles;
            const int tps = (int)((n_tiles + want - 1) / want);
            const int n_seg = (int)((n_tiles + tps - 1) / tps);
            rc = ensure_u32(&idx->d_keys2, &idx->keys2_cap, (int64_t)n_seg * n_s * kcap);
            gm_lv_opt o{};
            o.q = q_s;
            if (!rc) rc = launch_lv_by_len(idx, 1, kcap, n_s, k, n_seg, tps, d_s, c_s, k, idx->d_keys2, st, &o);
            if (!rc) rc = launch_merge(kcap, idx->d_keys2, n_seg, n_s, k, i_s, d_s, c_s, false, st);
            std::vector<int32_t> hd((size_t)(n_s * k)), hc((size_t)n_s);
            if (!rc && (hipMemcpyAsync(hd.data(), d_s, sizeof(int32_t) * hd.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipMemcpyAsync(hc.data(), c_s, sizeof(int32_t) * hc.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess))
                rc = fail(GM_ERR_DEVICE, "K2P sample read-back failed");
            (void)hipFree(q_s); (void)hipFree(d_s); (void)hipFree(i_s); (void)hipFree(c_s);
            if (rc) return rc;
            // cost(r) = 1.2 x [share of the pool the layout of radius r visits] + [share of the rows whose k-th neighbour lies
            // beyond r] x tail(r); tail = the plain pass over those rows (1.0), or -- enough rows -- a second partitioned level of
            // radius r + 1 over them: its visits + a fixed price + the share of THEM beyond r + 1
            double frac_r[9] = {0}, beyond_r[9] = {0};
            bool have_r[9] = {false};
            for (int rr = 1; rr <= 7; ++rr) {
                int bs[GM_LP_MAXBLK], bl[GM_LP_MAXBLK], cb[GM_LP_MAXCOPY], cs[GM_LP_MAXCOPY], fc[GM_LP_MAXBLK + 1];
                if (!gm_lp_layout(idx->L, rr, p_match, bs, bl, cb, cs, fc, &frac_r[rr])) continue;
                have_r[rr] = true;
                int64_t beyond = 0;
                for (int64_t j = 0; j < n_s; ++j)
                    if (hc[(size_t)j] < k || hd[(size_t)(j * k + k - 1)] > rr) ++beyond;
                beyond_r[rr] = (double)beyond / (double)n_s;
            }
            int best_r = 0, best_r2 = 0;
            // (the model tracks the measurements: S2, r = 7: modelled 0.78 of the plain pass, measured 7.1 of 9.2 ms; PAO1 cut at
            // L = 23, r = 5 + 6: 0.28 / 0.26 -- a partition has to promise 15 %)
            double best = 0.85;
            const int tune2 = gm_lp_tune_radius2(idx);
            for (int rr = 1; rr <= 7; ++rr) {
                if (!have_r[rr]) continue;
                double tail = 1.0;
                int r2 = 0;
                if (tune2 >= 0 && rr + 1 <= 7 && have_r[rr + 1] && beyond_r[rr] * (double)n_q >= 16384.0) {
                    const double t2 = 1.2 * frac_r[rr + 1] + 0.05 + beyond_r[rr + 1] / beyond_r[rr];
                    if (t2 < 0.8) {
                        tail = t2;
                        r2 = rr + 1;
                    }
                }
                const double cost = 1.2 * frac_r[rr] + beyond_r[rr] * tail;
                if (std::getenv("GM_DEBUG"))
                    fprintf(stderr, "[gm] K2P radius %d: visits %.3f of the pool, %.3f of the sample beyond it, tail %.3f%s -> cost %.3f\n",
                            rr, frac_r[rr], beyond_r[rr], tail, r2 ? " (second level)" : "", cost);
                if (cost < best) {
                    best = cost;
                    best_r = rr;
                    best_r2 = r2;
                }
            }
            p_r2 = best_r2;
            gm_lp_plan_set(idx, n_q, k, best_r, best_r2);
            p_r = best_r;
        }
        r = p_r;
        r2 = p_r2;
    }
    if (gm_lp_tune_radius2(idx) > 0) r2 = gm_lp_tune_radius2(idx);      // (tests, A/B)
    if (gm_lp_tune_radius2(idx) < 0 || r2 <= r) r2 = 0;
    if (r <= 0) return GM_OK;
    if ((rc = gm_lp_prepare(idx, r, st))) {
        if (rc == GM_ERR_ARG || rc == GM_ERR_ALLOC) return GM_OK;      // no layout for this length / no room: the plain pass
        return rc;
    }
    if ((rc = gm_lp_candidates(idx, idx->d_q, n_q, st))) return rc;
    const uint32_t* keys = nullptr;
    int n_lists = 0;
    if ((rc = timing_begin(idx, st))) return rc;
    idx->last_kernel = "k_leven_part_bs";
    if ((rc = gm_lp_scan(idx, idx->d_q, n_q, k, kcap, nullptr, nullptr, 1, st, &keys, &n_lists))) return rc;
    if ((rc = timing_end(idx, st, n_q * idx->n_db))) return rc;
    if ((rc = gm_lp_merge(idx, kcap, n_q, k, ids, dists, counts, st))) return rc;
    hipLaunchKernelGGL(k_lev_unvouched, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, st, counts, dists, n_q, k, r);
    HIP_TRY(hipGetLastError());
    *used = true;
    *r_used = r;
    *r2_used = r2;
    idx->lp_r1 = r;
    return GM_OK;
}

// Second K2P level: the n2 compact rows the first level could not vouch for (q2; bounds: their Hamming seeds) through the
// layout of radius r2, into the compact result arrays; rows still beyond r2 come back with count 0.
namespace {
struct LpLevelScope {      // the second level's pool copies and workspaces stand in for the first's while it runs
    gm_index* idx;
    explicit LpLevelScope(gm_index* i) : idx(i) { std::swap(idx->lp, idx->lp_deeper); }
    ~LpLevelScope() { std::swap(idx->lp, idx->lp_deeper); }
};
}  // namespace
int run_leven_partition_deeper(gm_index* idx, int r2, const uint2* q2, int64_t n2, int k, int kcap, const int32_t* seeds2,
                               int32_t* ids2, int32_t* dists2, int32_t* counts2, hipStream_t st, bool* used) {
    *used = false;
    LpLevelScope scope(idx);
    int rc;
    if ((rc = gm_lp_prepare(idx, r2, st))) {
        if (rc == GM_ERR_ARG || rc == GM_ERR_ALLOC) return GM_OK;      // no layout / no room: the plain second pass takes the rows
        return rc;
    }
    if ((rc = gm_lp_candidates(idx, q2, n2, st))) return rc;
    const uint32_t* keys = nullptr;
    int n_lists = 0;
    if ((rc = gm_lp_scan(idx, q2, n2, k, kcap, seeds2, nullptr, 1, st, &keys, &n_lists))) return rc;
    if ((rc = gm_lp_merge(idx, kcap, n2, k, ids2, dists2, counts2, st))) return rc;
    hipLaunchKernelGGL(k_lev_unvouched, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, counts2, dists2, n2, k, r2);
    HIP_TRY(hipGetLastError());
    *used = true;
    return GM_OK;
}

int run_leven_lookup(gm_index* idx, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                     int32_t* d1, hipStream_t st) {
    const bool nearest = d1 != nullptr;
    const int kcap = nearest ? 1 : (k <= 10 ? 10 : 20);    // (a capacity-5 instantiation measured slower than 10)
    int rc = GM_OK;
    // the partitioned pass (K2P) where the planner finds a radius that pays: it runs first and unseeded (its bounds are
    // clamped to the radius); only the rows it cannot vouch for need Hamming seeds, further down
    const bool try_part = !nearest && idx->variant != 2 && gm_lp_tune_radius(idx) >= 0 &&
                          (gm_lp_tune_radius(idx) > 0 || (idx->n_db >= 65536 && n_q >= 8192));
    bool part_used = false;
    int r_part = 0, r_part2 = 0;
    idx->lp_r1 = idx->lp_r2 = 0;
    idx->lp_rows2 = idx->lp_pairs2 = idx->lp_left2 = 0;
    if (try_part && (rc = run_leven_partition(idx, n_q, k, kcap, ids, dists, counts, st, &part_used, &r_part, &r_part2))) return rc;
    // seeds of the plain scan: the exact Hamming k-NN of the same candidates (lev <= hamming) -- through the
    // pigeonhole-partitioned scan where the planner finds a layout (pools >= 32 768), else the plain two-phase scan
    bool seeded = false;
    if (part_used) {
        // (nothing to seed here)
    } else if (idx->variant == 0) {
        rc = nearest ? run_hamming_pigeon(idx, true, n_q, 1, nullptr, d1, nullptr, st, &seeded, false)
                     : run_hamming_pigeon(idx, false, n_q, k, ids, dists, counts, st, &seeded, false);
        if (rc) return rc;
    }
    if (!seeded && !part_used)
        rc = nearest ? run_hamming_lookup(idx, true, n_q, 1, nullptr, d1, nullptr, false, st)
                     : run_hamming_lookup(idx, false, n_q, k, ids, dists, counts, false, st);
    if (rc) return rc;
    // enough (chunk, segment) workgroups to fill the chip several times over, segments of whole tiles
    const int algo = idx->variant == 2 ? 0 : 1;
    const int64_t n_tiles = algo == 1 ? idx->n_lk_tiles : idx->n_lv_tiles;
    const int64_t chunks = (n_q + gm::WAVE - 1) / gm::WAVE;
    // about eight rounds of (chunk, segment) workgroups at the kernel's four waves per SIMD: long work items in two
    // ragged rounds cost 10 % on S2 (swept 1 .. 256 segments: 12.6 ms at 4, 11.0 at 16-32, 12.4 at 256 where the merge
    // of the segment lists takes over)
    int64_t want = ((int64_t)idx->n_simd * 32 + chunks - 1) / chunks;
    if (want > 64) want = 64;
    if (idx->lv_seg > 0) want = idx->lv_seg;
    if (want < 1) want = 1;
    if (want > n_tiles) want = n_tiles;
    const int tiles_per_seg = (int)((n_tiles + want - 1) / want);
    const int n_seg = (int)((n_tiles + tiles_per_seg - 1) / tiles_per_seg);
    // own scratch: the Hamming seeder enqueued above may still be using idx->d_keys (the plain first pass only: after K2P the
    // scratch is sized for the rows it leaves, further down)
    if (!part_used && (rc = ensure_u32(&idx->d_keys2, &idx->keys2_cap, (int64_t)n_seg * n_q * kcap))) return rc;
    // Narrow-band first pass (K2B, k-NN, pools large enough for the partitioned Hamming seeder).  The Hamming seeds put
    // most bounds one above what the next narrower band serves (S2, L = 23: 85 % of the candidates at "lev < 9", band 4),
    // while the k-th Levenshtein neighbour itself is closer (7 for 97 % of them).  So when most seeds are that close, ALL
    // bounds are lowered to 2 (BAND - 1) + 2 ("lev < 8": band 3, 149 of 187 cells), a list that fills under it is exact
    // (every target within 7 was enumerated), and the candidates whose list stays short (S2: 3 %) are scanned again
    // under their own seeds.  The kernel takes the decision from a counter on the device: no read-back for it.
    const int band = ((3 * idx->L + 4) / 8) / 2;      // lvb_band(L), gm_leven_bs.hip.h
    const int cap_thr = 2 * (band - 1) + 2;
    bool two_pass = part_used || (!nearest && algo == 1 && seeded && band - 1 >= 2 && idx->lv_two_pass && n_q >= 4096);
    gm_lv_opt opt{};
    uint32_t* thr_all = nullptr;
    if (two_pass && !part_used) {
        // [n_q bounds][n_q compact seeds][1 counter]
        if ((rc = ensure_u32(&idx->d_lvthr, &idx->lvthr_cap, 2 * n_q + 4))) return rc;
        thr_all = idx->d_lvthr;
        int* cnt = reinterpret_cast<int*>(idx->d_lvthr + 2 * n_q);
        HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_lev_seed_stats, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0, st, dists, counts, n_q, k, idx->L,
                           cap_thr, thr_all, cnt);
        HIP_TRY(hipGetLastError());
        opt.cap_cnt = cnt;
        opt.cap_thr = cap_thr;
    }
    if (!part_used) {      // the plain first pass over all candidates
        rc = timing_begin(idx, st);
        if (rc) return rc;
        idx->last_kernel = algo == 1 ? (nearest ? "k_leven_knn_bs(nearest)" : "k_leven_knn_bs")
                                     : (nearest ? "k_leven_knn_lk(nearest)" : "k_leven_knn_lk");
        rc = nearest ? launch_lv_by_len(idx, algo, kcap, n_q, 1, n_seg, tiles_per_seg, d1, nullptr, 1, idx->d_keys2, st)
                     : launch_lv_by_len(idx, algo, kcap, n_q, k, n_seg, tiles_per_seg, dists, counts, k, idx->d_keys2, st, &opt);
        if (rc) return rc;
        rc = timing_end(idx, st, n_q * idx->n_db);
        if (rc) return rc;
        rc = nearest ? launch_merge(kcap, idx->d_keys2, n_seg, n_q, 1, nullptr, d1, nullptr, true, st)
                     : launch_merge(kcap, idx->d_keys2, n_seg, n_q, k, ids, dists, counts, false, st);
    }
    if (rc || !two_pass) return rc;
    // second pass: the candidates whose list did not fill (none when the first pass ran under the seeds themselves)
    if ((rc = gm_pg_collect_short(idx, n_q, k, counts, st))) return rc;
    int64_t n2 = 0;
    if ((rc = gm_pg_read_bad(idx, &n2, st))) return rc;
    idx->last_incomplete = n2;
    if (part_used) idx->last_evaluated = gm_lp_last_pairs(idx, 0);      // (the read-back above synchronised the stream)
    if (n2 * 4 > n_q && !part_used) idx->lv_two_pass = 0;      // the lowered bound left a quarter of the lists short on this pool: not again
    if (n2 == 0) return GM_OK;
    const uint2* q2 = nullptr;
    int32_t *ids2 = nullptr, *dists2 = nullptr, *counts2 = nullptr;
    if ((rc = gm_pg_gather(idx, n2, k, false, &q2, &ids2, &dists2, &counts2, st))) return rc;
    int32_t* seeds2 = nullptr;
    if (!part_used) {
        seeds2 = reinterpret_cast<int32_t*>(idx->d_lvthr + n_q);
        hipLaunchKernelGGL(k_lev_gather_seeds, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, thr_all, gm_pg_inc_list(idx), n2,
                           seeds2);
        HIP_TRY(hipGetLastError());
    }
    if (part_used && r_part2 > r_part && (n2 >= 4096 || gm_lp_tune_radius2(idx) > 0)) {
        // K2P's second level: these rows through the layout of radius r + 1 (unseeded: bounds clamped to that radius), then
        // only what THAT leaves takes the plain pass
        bool used2 = false;
        if ((rc = run_leven_partition_deeper(idx, r_part2, q2, n2, k, kcap, nullptr, ids2, dists2, counts2, st, &used2))) return rc;
        if (used2) {
            if ((rc = gm_pg_scatter(idx, n2, k, false, ids, dists, counts, st))) return rc;
            idx->lp_r2 = r_part2;
            idx->lp_rows2 = n2;
            if ((rc = gm_pg_collect_short(idx, n_q, k, counts, st))) return rc;
            int64_t n3 = 0;
            if ((rc = gm_pg_read_bad(idx, &n3, st))) return rc;
            idx->lp_left2 = n3;
            idx->last_incomplete = n3;
            idx->lp_pairs2 = gm_lp_last_pairs(idx, 1);
            if (std::getenv("GM_DEBUG"))
                fprintf(stderr, "[gm] K2P second level: %lld rows, radius %d, %.3g pairs, %lld rows left for the plain pass\n",
                        (long long)n2, r_part2, (double)idx->lp_pairs2, (long long)n3);
            if (n3 == 0) return GM_OK;
            n2 = n3;
            if ((rc = gm_pg_gather(idx, n2, k, false, &q2, &ids2, &dists2, &counts2, st))) return rc;
        }
    }
    int seed_k2 = 1;
    const int32_t* seed_c2 = nullptr;
    if (part_used) {
        // the rows the partition leaves: their Hamming k-NN (plain two-phase look-up scan over the compact rows) are the
        // seeds of the plain pass, read from the compact result arrays the pass then overwrites
        if ((rc = run_hamming_lookup(idx, false, n2, k, ids2, dists2, counts2, false, st, q2))) return rc;
        seeds2 = dists2;
        seed_c2 = counts2;
        seed_k2 = k;
    }
    const int64_t chunks2 = (n2 + gm::WAVE - 1) / gm::WAVE;
    int64_t want2 = ((int64_t)idx->n_simd * 8 + chunks2 - 1) / chunks2;
    if (want2 > 64) want2 = 64;
    if (want2 < 1) want2 = 1;
    {
        // whole rounds of the chip's wave slots (four waves per SIMD for this kernel): a few thousand rows x 64 segments are
        // 1.1 rounds -- the ragged second one costs as much as the first (S2: 4 443 rows, 0.47 -> 0.3 ms)
        const int64_t slots = (int64_t)idx->n_simd * 4;
        int64_t rounds = (chunks2 * want2 + slots / 2) / slots;
        if (rounds < 1) rounds = 1;
        const int64_t w = rounds * slots / chunks2;
        if (w >= 1 && w <= 64 && chunks2 * want2 > slots / 2 && chunks2 * want2 < 2 * slots) want2 = w;      // (few rows only)
    }
    if (want2 > n_tiles) want2 = n_tiles;
    const int tps2 = (int)((n_tiles + want2 - 1) / want2);
    const int n_seg2 = (int)((n_tiles + tps2 - 1) / tps2);
    if ((rc = ensure_u32(&idx->d_keys2, &idx->keys2_cap, (int64_t)n_seg2 * n2 * kcap))) return rc;
    gm_lv_opt opt2{};
    opt2.q = q2;
    if ((rc = launch_lv_by_len(idx, algo, kcap, n2, k, n_seg2, tps2, seeds2, seed_c2, seed_k2, idx->d_keys2, st, &opt2))) return rc;
    if ((rc = launch_merge(kcap, idx->d_keys2, n_seg2, n2, k, ids2, dists2, counts2, false, st))) return rc;
    return gm_pg_scatter(idx, n2, k, false, ids, dists, counts, st);
}

// K1P prefix re-scan (gm_pg_prefix_fraction): K1L over the first `frac` of the pool (id order) for the n compacted rows
// under the bound guess_thr; its segment lists complete the rows' partial K1P lists in place (ids / dists / counts: the
// ORIGINAL rows), and the rows still short become the handle's list of incomplete rows (gm_pg_merge_prefix).
int run_hamming_prefix(gm_index* idx, bool nearest, int64_t n, int k, int kcap, const uint2* q2, const int32_t* q_rows, int guess_thr,
                       double frac, int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st) {
    gm_lk_args a;
    a.q = q2;
    a.q_rows = q_rows;
    a.n_dev = idx->dev_rows;
    a.kcap = kcap;
    a.n_q = n;
    a.k = nearest ? 1 : k;
    const int64_t chunks = (n + gm::WAVE - 1) / gm::WAVE;
    int64_t n_tiles_p = (int64_t)std::ceil((double)idx->n_lk_tiles * frac);
    n_tiles_p += n_tiles_p & 1;
    if (n_tiles_p < 2) n_tiles_p = 2;
    if (n_tiles_p > idx->n_lk_tiles) n_tiles_p = idx->n_lk_tiles;
    int64_t n_seg = ((int64_t)idx->n_simd * 8 + chunks - 1) / chunks;      // about one workgroup per wave slot
    if (n_seg > 64) n_seg = 64;
    if (idx->lk_seg > 0) n_seg = idx->lk_seg;      // (experiments)
    if (n_seg > n_tiles_p / 2) n_seg = n_tiles_p / 2;
    if (n_seg < 1) n_seg = 1;
    int64_t tps = (n_tiles_p + n_seg - 1) / n_seg;
    tps += tps & 1;
    n_seg = (n_tiles_p + tps - 1) / tps;
    int rc;
    if ((rc = ensure_u32(&idx->d_keys, &idx->keys_cap, n_seg * n * kcap))) return rc;
    a.n_seg = (int)n_seg;
    a.seg_base = 0;
    a.tiles_per_seg = (int)tps;
    a.seed_all = ((uint32_t)guess_thr << 27) + 1u;
    a.out_keys = idx->d_keys;
    a.n_tiles = (int)n_tiles_p;
    if ((rc = launch_lk_len(idx, a, st))) return rc;
    return gm_pg_merge_prefix(idx, kcap, idx->d_keys, (int)n_seg, n, a.k, nearest, ids, dists, counts, st);
}

// K1P (gm_pigeon.hip): exact Hamming k-NN from B block-sorted pool copies, then a deeper K1P level (larger radius,
// layout of its own) or the plain scan for the candidates whose merged list did not fill.  Returns through *used
// whether K1P ran (the layout planner may decide that the plain scan is the better choice for this pool / k).
namespace {
// the K1P state of cascade level `level` >= 1 swapped into the handle while that level runs on the compact candidate
// set q (every gm_pg_* call works on idx->pg / idx->d_q)
struct PgLevelScope {
    gm_index* idx;
    gm_pg_level_state* lv;
    uint2* saved_q;
    int saved_level;
    void swap_state() {
        std::swap(idx->pg, lv->pg);
        std::swap(idx->plan_nq, lv->plan_nq);
        std::swap(idx->plan_k, lv->plan_k);
        std::swap(idx->plan_found, lv->plan_found);
        std::swap(idx->plan_cascade, lv->plan_cascade);
        std::swap(idx->plan_parent_radius, lv->plan_parent_radius);
        std::swap(idx->plan_cost, lv->plan_cost);
        std::swap(idx->pg_flooded_k1, lv->pg_flooded_k1);
        std::swap(idx->plan_lay, lv->plan_lay);
        std::swap(idx->plan_p_short, lv->plan_p_short);
    }
    PgLevelScope(gm_index* i, int level, const uint2* q) : idx(i), lv(&i->pg_deeper[level - 1]), saved_q(i->d_q), saved_level(i->pg_level) {
        swap_state();
        idx->d_q = const_cast<uint2*>(q);
        idx->pg_level = level;
    }
    ~PgLevelScope() {
        swap_state();
        idx->d_q = saved_q;
        idx->pg_level = saved_level;
    }
};
}  // namespace

static const std::vector<uint32_t>& idx_cum(const gm_index* idx) { return idx->plan_cum; }      // (level 0's sample: not swapped)
static int idx_cum_k(const gm_index* idx) { return idx->plan_cum_k; }

// A speculative run (SpecRun != nullptr): the counts the host used to wait for are taken from the handle's last query of
// this shape (+ a margin), the device's counts go to mailboxes, and run_hamming_pigeon checks them once at the end.
namespace {
struct SpecCheck {
    gm_pg_pending pend;
    int level, kind;      // kind 0: rows left by the merge, 1: by the prefix re-scan, 2: by the plain re-scan under the guess
    int64_t cap;
};
struct SpecRun {
    std::vector<SpecCheck> checks;
};
int64_t spec_cap(int64_t last) { return last <= 0 ? 0 : last + last / 4 + 2048; }
int64_t spec_cap_small(int64_t last) { return last <= 0 ? 0 : last + last / 4 + 64; }
}  // namespace

static int run_pigeon_level(gm_index* idx, int level, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                            int32_t* counts, hipStream_t st, bool* used, bool timed, const double* p_short_in,
                            int parent_radius, SpecRun* sp, int64_t n_q_dec, const int* n_q_dev) {
    // (n_q_dec: the row count DECISIONS are taken on -- n_q itself, or, in a speculative run on an upper bound of the row
    // count, the actual count of the run it learnt from, so that both take the same path; n_q_dev: in such a run the device
    // address of the level's actual candidate count, which its grouping and merge kernels clamp to)
    *used = false;
    struct DevCountScope {      // the device counts this level's launches clamp to; the caller's come back on every exit
        gm_index* i;
        const int *nq, *rows;
        ~DevCountScope() {
            i->dev_nq = nq;
            i->dev_rows = rows;
        }
    } dev_scope{idx, idx->dev_nq, idx->dev_rows};
    idx->dev_nq = n_q_dev;
    idx->dev_rows = nullptr;
    const int keff = nearest ? 1 : k;
    if (keff > 20 || idx->n_db < keff || n_q < 1) return GM_OK;
    // variants 100 + B force the plain layout of B blocks; 200 + B one expansion block (even lengths), 300 + B
    // the same with the expansion block one base longer, 400 + B / 500 + B two / three expansion blocks one
    // base longer than the even split, 600 + B / 700 + B two / three expansion blocks of even length; 800 + B:
    // block 0 with the two-substitution neighbourhood, the others plain; 900 + B: block 0 two-substitution, block 1
    // one-substitution (tests, A/B); the result is exact for any layout
    const int vcode = idx->variant / 100;
    const bool forced = level == 0 && vcode >= 1 && vcode <= 9 && idx->variant % 100 >= 2 && idx->variant % 100 <= GM_PG_MAXB;
    int rc;
    double p_match = 0.25;
    if (idx->n_db >= 32768 || forced || level > 0) {
        if ((rc = gm_pg_pool_stats(idx, st, &p_match))) return rc;
    }
    gm_pg_layout lay{};
    bool planned = false;      // the layout is the planner's own (not a test / experiment override)
    if (forced) {
        const int B = idx->variant % 100;
        int expand = vcode == 1 ? 0 : (vcode <= 3 ? 1 : (vcode <= 5 ? vcode - 2 : (vcode <= 7 ? vcode - 4 : vcode - 7)));
        const int len0 = (idx->L + B - 1) / B + (vcode >= 3 && vcode <= 5 ? 1 : 0);
        if (expand >= B || len0 > GM_PG_MAXLEN) return GM_OK;      // not a usable layout: plain scan
        if (expand > 0 && idx->L - expand * len0 < 2 * (B - expand)) return GM_OK;
        if (idx->L / B < 2 && expand == 0) return GM_OK;
        if (!gm_pg_layout_make(idx->L, B, expand, len0, vcode >= 8 ? 1 : 0, &lay)) return GM_OK;
    } else if (gm_pg_layout_forced(idx, &lay)) {
        // (gm_set_param "pg_layout" / "pg_layout2": experiments, tests)
        if (level > 0) {
            idx->plan_p_short.clear();
            if (p_short_in) idx->plan_p_short.assign(p_short_in, p_short_in + idx->L + 1);
        }
    } else {
        // (the pool and its composition belong to the handle; a deeper level's candidate count varies a little from query
        // to query: its plan stands while the count stays within an eighth of the planned one)
        // (a speculative run hands a deeper level its PREDICTED row count + margin: the plan of the run it learnt from stands)
        const bool stale = level == 0 ? idx->plan_nq != n_q
                                      : (idx->plan_nq < 0 || idx->plan_parent_radius != parent_radius ||      // (the parent re-planned)
                                         (!sp && std::llabs((long long)(n_q - idx->plan_nq)) * 8 > (long long)idx->plan_nq));
        if (stale || idx->plan_k != keff) {
            // how far the k-th neighbour lies, measured: a plain scan of every (n_q / 2048)-th candidate against the
            // whole pool (~0.1 ms, once per plan) -- the independent-base model knows the pool's composition, not the
            // spread between candidates or the repeats of a real genome
            std::vector<double> p_short;
            constexpr int64_t SAMPLE = 2048;
            if (level == 0) idx->plan_cum.clear();      // (a sample belongs to the candidates it was drawn from)
            if (level > 0) {
                if (p_short_in) p_short.assign(p_short_in, p_short_in + idx->L + 1);      // the parent's shares, conditional
            } else if (n_q >= 16 * SAMPLE && idx->n_db >= 32768 && idx->L >= 12 && gm_pg_sampling(idx)) {      // (gm_pg_plan's own preconditions)
                const uint2* qs = nullptr;
                int32_t *ids_s = nullptr, *dists_s = nullptr, *counts_s = nullptr;
                if ((rc = gm_pg_sample_rows(idx, SAMPLE, n_q / SAMPLE, st))) return rc;
                if ((rc = gm_pg_gather(idx, SAMPLE, keff, nearest, &qs, &ids_s, &dists_s, &counts_s, st))) return rc;
                (void)ids_s; (void)dists_s; (void)counts_s; (void)qs;
                // targets within d = 0 .. 8 of every sample row: the k-th neighbour's distance AND the hits a radius costs
                if ((rc = gm_pg_sample_hist(idx, SAMPLE, st, &idx->plan_cum))) return rc;
                idx->plan_cum_k = keff;
                std::vector<int64_t> beyond((size_t)idx->L + 2, 0);      // rows whose k-th neighbour lies beyond r
                for (int64_t j = 0; j < SAMPLE; ++j) {
                    int kth = idx->L + 1;
                    for (int d = 0; d < GM_PG_SAMPLE_BINS; ++d)
                        if (idx->plan_cum[(size_t)(j * GM_PG_SAMPLE_BINS + d)] >= (uint32_t)keff) {
                            kth = d;
                            break;
                        }
                    for (int r = 0; r < kth && r <= idx->L; ++r) ++beyond[(size_t)r];
                }
                p_short.resize((size_t)idx->L + 1);
                for (int r = 0; r <= idx->L; ++r) p_short[(size_t)r] = ((double)beyond[(size_t)r] + 0.5) / (double)(SAMPLE + 1);
            }
            if (std::getenv("GM_DEBUG") && !p_short.empty()) {
                fprintf(stderr, "[gm] level %d planner (%lld rows): share with the k-th neighbour beyond r =", level, (long long)n_q);
                for (int r = 2; r <= 9 && r <= idx->L; ++r) fprintf(stderr, " %d:%.4f", r, p_short[(size_t)r]);
                fprintf(stderr, "\n");
                if (!idx->plan_cum.empty()) {
                    const int n_s = (int)(idx->plan_cum.size() / GM_PG_SAMPLE_BINS);
                    fprintf(stderr, "[gm] level %d sample: mean targets within d (rows beyond the parent's radius %d) =", level, parent_radius);
                    for (int d = 2; d < GM_PG_SAMPLE_BINS; ++d) {
                        double sum = 0.0;
                        int rows = 0;
                        for (int j = 0; j < n_s; ++j) {
                            const uint32_t* c = idx->plan_cum.data() + (size_t)j * GM_PG_SAMPLE_BINS;
                            int kth = GM_PG_SAMPLE_BINS;
                            for (int dd = 0; dd < GM_PG_SAMPLE_BINS; ++dd)
                                if (c[dd] >= (uint32_t)keff) {
                                    kth = dd;
                                    break;
                                }
                            if (level > 0 && kth <= parent_radius) continue;
                            sum += c[d];
                            ++rows;
                        }
                        fprintf(stderr, " %d:%.1f", d, rows ? sum / rows : 0.0);
                    }
                    fprintf(stderr, "\n");
                }
            }
            const int deeper = idx->pg_cascade != 0 ? GM_PG_LEVELS - 1 - level : 0;
            double cost = 1.0;
            idx->plan_cascade = 0;
            // (the histogram sample belongs to level 0; a deeper level plans on its rows beyond the parent's radius)
            gm_pg_sample smp{};
            const bool have_smp = !idx_cum(idx).empty() && idx_cum_k(idx) == keff && !p_short.empty();
            if (have_smp) {
                smp.n = (int)(idx_cum(idx).size() / GM_PG_SAMPLE_BINS);
                smp.cum = idx_cum(idx).data();
                smp.k = keff;
                smp.min_kth = level > 0 ? parent_radius : -1;
            }
            idx->plan_found = gm_pg_plan(idx->n_db, n_q, idx->L, keff, p_match, &idx->plan_lay,
                                         p_short.empty() ? nullptr : p_short.data(), deeper, &cost, &idx->plan_cascade,
                                         have_smp ? &smp : nullptr);
            double expect = 0.0;
            if (idx->plan_found && !p_short.empty()) {
                const int r = gm_pg_layout_radius(idx->plan_lay);
                expect = p_short[(size_t)(r < idx->L ? r : idx->L)];
            }
            if (std::getenv("GM_DEBUG")) {
                fprintf(stderr, "[gm] level %d plan for %lld candidates: %s", level, (long long)n_q, idx->plan_found ? "blocks" : "none");
                for (int b = 0; idx->plan_found && b < idx->plan_lay.B; ++b) fprintf(stderr, " %d/%d", idx->plan_lay.blen[b], idx->plan_lay.trad[b]);
                fprintf(stderr, " cost %.4f, expects %.4f short%s\n", cost, expect, idx->plan_cascade ? " -> deeper level" : "");
            }
            gm_pg_expect_short(idx, expect);
            {      // targets within the planned radius per candidate of THIS level (sample rows beyond the parent's radius)
                double hits = 0.0;
                if (idx->plan_found && have_smp) {
                    const int r = gm_pg_layout_radius(idx->plan_lay), rr = r < GM_PG_SAMPLE_BINS ? r : GM_PG_SAMPLE_BINS - 1;
                    double sum = 0.0;
                    int rows = 0;
                    for (int j = 0; j < smp.n; ++j) {
                        const uint32_t* c = smp.cum + (size_t)j * GM_PG_SAMPLE_BINS;
                        int kth = GM_PG_SAMPLE_BINS;
                        for (int dd = 0; dd < GM_PG_SAMPLE_BINS; ++dd)
                            if (c[dd] >= (uint32_t)keff) {
                                kth = dd;
                                break;
                            }
                        if (kth <= smp.min_kth) continue;
                        sum += (double)c[rr];
                        ++rows;
                    }
                    hits = rows ? sum / rows : 0.0;
                    for (int e = GM_PG_SAMPLE_BINS - 1; e < r; ++e) hits *= 4.0;      // (beyond the histogram: extrapolated)
                }
                gm_pg_expect_hits(idx, hits);
            }
            idx->pg_flooded_k1 = 0;      // (what a previous candidate set did to the hit lists says nothing about this one)
            idx->plan_cost = cost;
            idx->plan_p_short = p_short;
            idx->plan_nq = n_q;
            idx->plan_k = keff;
            idx->plan_parent_radius = parent_radius;
        }
        if (!idx->plan_found) return GM_OK;
        if (idx->pg_flooded_k1 == keff + 1) return GM_OK;
        if (level > 0 && idx->pg_cascade == 1) {
            // a level of its own has a fixed price (grouping, merge, read-back: ~150 us): for a few thousand rows the
            // prefix / plain re-scan of the parent is the cheaper way (S1: 5 414 rows, 0.10 ms against 0.19)
            const double fixed = 300e-6 * 2.6e13 / ((double)idx->n_db * (double)n_q_dec) + 11700.0 / (double)idx->n_db;
            if (idx->plan_cost + fixed >= 1.0) return GM_OK;
        }
        lay = idx->plan_lay;
        planned = true;
    }
    if (level > 0 && gm_pg_layout_radius(lay) <= parent_radius) return GM_OK;      // (a deeper level must reach further)
    if ((rc = gm_pg_prepare(idx, &lay, st))) {
        if (forced && rc == GM_ERR_ARG) return GM_OK;      // forced layout not usable for this length
        return rc;
    }
    if ((rc = gm_pg_candidates(idx, n_q, keff, st))) return rc;
    const int kcap = nearest ? 1 : (k <= 5 ? 5 : (k <= 10 ? 10 : 20));
    if (timed && (rc = timing_begin(idx, st))) return rc;
    if (timed) idx->last_kernel = gm_pg_hit_path_is_registers(idx) ? "k_hamming_scan_pg4" : "k_hamming_scan_pg";
    if ((rc = gm_pg_scan(idx, n_q, keff, kcap, st))) {
        if (rc == GM_ERR_ALLOC) return GM_OK;      // no room for the slot lists: the plain scan takes over
        return rc;
    }
    if (timed && (rc = timing_end(idx, st, n_q * idx->n_db))) return rc;
    if ((rc = gm_pg_merge(idx, kcap, n_q, keff, ids, dists, counts, nearest, st))) return rc;
    int64_t n_inc = 0, pairs = 0;
    int deferred = -1;      // speculative run: index of this level's "rows left by the merge" check while it has no publish yet
    auto publish_merge_counts = [&]() -> int {      // (called with THIS level's state in the handle)
        if (deferred < 0) return GM_OK;
        gm_pg_pending p{};
        const int e = gm_pg_publish_async(idx, 1, st, &p);
        if (e) return e;
        sp->checks[(size_t)deferred].pend = p;
        deferred = -1;
        return GM_OK;
    };
    if (sp) {
        n_inc = spec_cap(idx->spec.ninc[level]);
        if (n_inc > n_q) n_inc = n_q;
        pairs = level < GM_PG_LEVELS ? idx->casc_pairs_last[level] : 0;
        SpecCheck c{};
        c.level = level;
        c.kind = 0;
        c.cap = n_inc;
        // (no publish of its own yet: the level's counters as of the merge -- rows left, pairs, the histogram of the short
        // rows -- do not change any more, so the level's NEXT publish, after the prefix / plain re-scan, carries them too;
        // only a level that ends without one publishes here, `publish_merge_counts` below)
        sp->checks.push_back(c);
        deferred = (int)sp->checks.size() - 1;
        idx->dev_rows = gm_pg_count_ptr(idx, 0);      // the rows the merge left: every re-scan launch below clamps to it
    } else {
        if ((rc = gm_pg_incomplete(idx, &n_inc, &pairs, st))) return rc;
        if (level < GM_PG_LEVELS) {
            idx->spec.ninc[level] = n_inc;
            idx->spec.nleft[level] = idx->spec.nbad[level] = -1;
            idx->casc_pairs_last[level] = pairs;
        }
    }
    if (timed) {
        idx->last_evaluated = pairs;
        idx->last_incomplete = n_inc;
    }
    const int radius = gm_pg_radius(idx);
    if (level < GM_PG_LEVELS) {
        idx->casc_nq[level] = n_q;
        idx->casc_radius[level] = radius;
        idx->casc_pairs[level] = pairs;
        idx->casc_left[level] = n_inc;
        idx->casc_levels = level + 1;
    }
    if (std::getenv("GM_DEBUG"))
        fprintf(stderr, "[gm] level %d: %lld candidates, radius %d, %.3g pairs evaluated, %lld rows left short\n", level,
                (long long)n_q, radius, (double)pairs, (long long)n_inc);
    double expect = 0.0;      // share of the rows the plan itself expects short at this radius
    if (planned && !idx->plan_p_short.empty()) expect = idx->plan_p_short[(size_t)(radius < idx->L ? radius : idx->L)];
    // A pool of repeat families floods the hit lists and K1P ends up handing most rows to the plain scan: twice the
    // work of the plain scan alone.  The answer of this query stands (exact either way); remember it for the next.
    if (!sp && planned && (double)n_inc > (expect + 1.0 / 3.0) * (double)n_q && n_q >= 4096 && gm_pg_debug_flags(idx) == 0)
        idx->pg_flooded_k1 = keff + 1;
    const int64_t n_inc_dec = sp && level < GM_PG_LEVELS ? idx->spec.ninc[level] : n_inc;
    int64_t n_real = sp && level < GM_PG_LEVELS ? idx->spec.ninc[level] : -1;      // speculative run: rows expected under the bound n_inc
    if (n_inc > 0) {
        const uint2* q2 = nullptr;
        const int32_t* q_rows = nullptr;      // (the K1L re-scans read the listed rows in place: no compact copy, no gather launch)
        int32_t *ids2 = nullptr, *dists2 = nullptr, *counts2 = nullptr;
        // Cascade: many rows left short (their k-th neighbour lies beyond this level's radius: the planner chose the
        // radius the majority needs) -> a deeper K1P level over the compact rows, planned for THEIR number and the
        // conditional distribution of their k-th neighbour; what that level leaves takes its own re-scan paths.
        const bool go_deeper = level + 1 < GM_PG_LEVELS && idx->pg_cascade != 0 && idx->pg_flooded_k1 != keff + 1 &&
            (idx->pg_cascade == 2 ? n_inc_dec >= 64
                                  : (n_inc_dec >= 4096 && !idx->plan_p_short.empty() &&
                                     (idx->plan_cascade != 0 || n_inc_dec >= 16384)));      // (the plan counted on a deeper level, or far more rows are short than it expected)
        if ((rc = gm_pg_gather(idx, n_inc, keff, nearest, &q2, &ids2, &dists2, &counts2, st, go_deeper ? nullptr : &q_rows))) return rc;
        if (go_deeper) {
            std::vector<double> ps2;
            if (!idx->plan_p_short.empty()) {
                const int L = idx->L;
                const double base = idx->plan_p_short[(size_t)(radius < L ? radius : L)];
                ps2.resize((size_t)L + 1);
                for (int rr = 0; rr <= L; ++rr) {
                    const double c = rr <= radius || base <= 0.0 ? 1.0 : idx->plan_p_short[(size_t)rr] / base;
                    ps2[(size_t)rr] = c < 1.0 ? c : 1.0;
                }
            }
            bool used2 = false;
            {
                const gm_pigeon* parent = gm_pg_state(idx);
                const int* rows_dev = idx->dev_rows;      // (this level's count: the deeper level's candidates)
                PgLevelScope scope(idx, level + 1, q2);
                gm_pg_inherit(idx, parent);
                rc = run_pigeon_level(idx, level + 1, nearest, n_inc, k, ids2, dists2, counts2, st, &used2, false,
                                      ps2.empty() ? nullptr : ps2.data(), radius, sp, n_inc_dec, rows_dev);
            }
            if (rc) return rc;
            if (used2) {
                if ((rc = gm_pg_scatter(idx, n_inc, keff, nearest, ids, dists, counts, st))) return rc;
                if ((rc = publish_merge_counts())) return rc;
                *used = true;
                return GM_OK;
            }
        }
        // A row that is short without an overflow knows all its neighbours within r; the k - c it lacks are the targets
        // at r + 1 with the smallest ids, and the first part of the pool (id order) almost surely holds them: scan only
        // that prefix, fold in the partial lists, and hand the plain re-scan just the rows that are still short (and
        // those whose lists overflowed).
        const double frac = sp && level < GM_PG_LEVELS ? idx->spec.frac[level] : gm_pg_prefix_fraction(idx, keff);
        if (!sp && level < GM_PG_LEVELS) idx->spec.frac[level] = frac;
        if (frac <= 0.6) {
            if ((rc = run_hamming_prefix(idx, nearest, n_inc, keff, kcap, q2, q_rows, radius + 2, frac, ids, dists, counts, st)))
                return rc;
            int64_t n_left = 0;
            if (sp) {
                n_left = spec_cap_small(level < GM_PG_LEVELS ? idx->spec.nleft[level] : 0);
                if (n_left > n_inc) n_left = n_inc;
                SpecCheck c{};
                c.level = level;
                c.kind = 1;
                c.cap = n_left;
                if ((rc = gm_pg_publish_async(idx, 2, st, &c.pend))) return rc;
                if (deferred >= 0) {
                    sp->checks[(size_t)deferred].pend = c.pend;
                    deferred = -1;
                }
                sp->checks.push_back(c);
                idx->dev_rows = gm_pg_count_ptr(idx, 1);      // the rows the collect left (as of this publish)
            } else {
                if ((rc = gm_pg_read_bad(idx, &n_left, st))) return rc;
                if (level < GM_PG_LEVELS) idx->spec.nleft[level] = n_left;
            }
            if (std::getenv("GM_DEBUG"))
                fprintf(stderr, "[gm] prefix re-scan: %lld rows, fraction %.3f, %lld left%s\n", (long long)n_inc, frac, (long long)n_left, sp ? " (predicted)" : "");
            if (n_left == 0) {
                *used = true;
                return GM_OK;
            }
            n_inc = n_left;
            n_real = sp && level < GM_PG_LEVELS ? idx->spec.nleft[level] : -1;
            if ((rc = gm_pg_gather(idx, n_inc, keff, nearest, &q2, &ids2, &dists2, &counts2, st, &q_rows))) return rc;
        }
        // these rows have fewer than k neighbours within r (or a flooded slot list): their k-th neighbour is
        // almost surely at r + 1, so scan under the bound d < r + 2 first and fall back to the unseeded scan
        // only if a list still does not fill
        const int guess = radius + 2;
        if ((rc = run_hamming_lookup(idx, nearest, n_inc, keff, ids2, dists2, counts2, false, st, q2, guess, n_real, q_rows))) return rc;
        if ((rc = gm_pg_scatter(idx, n_inc, keff, nearest, ids, dists, counts, st))) return rc;
        // the rows whose list did not fill under the guess (and only those) go through the unseeded scan
        int64_t n_bad = 0;
        if ((rc = gm_pg_collect_bad(idx, n_inc, keff, nearest, st))) return rc;
        if (sp) {
            n_bad = spec_cap_small(level < GM_PG_LEVELS ? idx->spec.nbad[level] : 0);
            if (n_bad > n_inc) n_bad = n_inc;
            SpecCheck c{};
            c.level = level;
            c.kind = 2;
            c.cap = n_bad;
            if ((rc = gm_pg_publish_async(idx, 3, st, &c.pend))) return rc;
            if (deferred >= 0) {
                sp->checks[(size_t)deferred].pend = c.pend;
                deferred = -1;
            }
            sp->checks.push_back(c);
            idx->dev_rows = gm_pg_count_ptr(idx, 1);
        } else {
            if ((rc = gm_pg_read_bad(idx, &n_bad, st))) return rc;
            if (level < GM_PG_LEVELS) idx->spec.nbad[level] = n_bad;
        }
        if (std::getenv("GM_DEBUG"))
            fprintf(stderr, "[gm] plain re-scan: %lld rows under the bound %d, %lld left for the unseeded scan\n", (long long)n_inc,
                    guess, (long long)n_bad);
        if (n_bad > 0) {
            if ((rc = gm_pg_gather(idx, n_bad, keff, nearest, &q2, &ids2, &dists2, &counts2, st, &q_rows))) return rc;
            if ((rc = run_hamming_lookup(idx, nearest, n_bad, keff, ids2, dists2, counts2, false, st, q2, 0,
                                         sp && level < GM_PG_LEVELS ? idx->spec.nbad[level] : -1, q_rows)))
                return rc;
            if ((rc = gm_pg_scatter(idx, n_bad, keff, nearest, ids, dists, counts, st))) return rc;
        }
    }
    if ((rc = publish_merge_counts())) return rc;
    *used = true;
    return GM_OK;
}

int run_hamming_pigeon(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                       int32_t* counts, hipStream_t st, bool* used, bool timed = true) {
    idx->casc_levels = 0;
    auto& S = idx->spec;
    const bool same_shape = S.valid && S.n_q == n_q && S.k == k && S.nearest == nearest;
    if (!same_shape) S.cooldown = 0;
    const bool speculate = same_shape && idx->pg_spec != 0 && S.cooldown == 0 && gm_pg_mailbox_usable(idx) && !std::getenv("GM_DEBUG");
    if (S.cooldown > 0) --S.cooldown;
    idx->spec_last = 0;
    if (speculate) {
        // Everything is enqueued on predicted row counts; ONE look at the mailboxes afterwards.
        SpecRun sp;
        int rc = run_pigeon_level(idx, 0, nearest, n_q, k, ids, dists, counts, st, used, timed, nullptr, -1, &sp, n_q, nullptr);
        if (rc) return rc;
        bool ok = true;
        for (const SpecCheck& c : sp.checks) {
            int64_t n_inc = 0, n_bad = 0, pairs = 0;
            if (c.pend.pg == nullptr || (rc = gm_pg_pending_wait(&c.pend, c.kind == 0, st, &n_inc, &n_bad, &pairs))) {
                ok = false;      // (no mailbox on this platform after all: the synchronous run below decides)
                break;
            }
            const int64_t actual = c.kind == 0 ? n_inc : n_bad;
            if (actual > c.cap) ok = false;
            if (c.level < GM_PG_LEVELS) {
                if (c.kind == 0) {
                    S.ninc[c.level] = actual;
                    idx->casc_left[c.level] = actual;
                    idx->casc_pairs[c.level] = idx->casc_pairs_last[c.level] = pairs;
                    if (c.level == 0 && timed) {
                        idx->last_evaluated = pairs;
                        idx->last_incomplete = actual;
                    }
                    if (c.level + 1 < GM_PG_LEVELS && c.level + 1 < idx->casc_levels) idx->casc_nq[c.level + 1] = actual;
                } else if (c.kind == 1) {
                    S.nleft[c.level] = actual;
                } else {
                    S.nbad[c.level] = actual;
                }
            }
        }
        idx->spec_last = ok ? 1 : 2;
        if (ok) return GM_OK;
        // a level left more rows than its cap covers: this query again, waiting for the counts; the next few stay that way
        S.valid = false;
        S.cooldown = 4;
        idx->casc_levels = 0;
    }
    const int rc = run_pigeon_level(idx, 0, nearest, n_q, k, ids, dists, counts, st, used, timed, nullptr, -1, nullptr, n_q, nullptr);
    if (rc == GM_OK && *used) {
        S.valid = true;
        S.n_q = n_q;
        S.k = k;
        S.nearest = nearest;
    } else {
        S.valid = false;
    }
    return rc;
}

// variant 0 = default, 7 = K1L with its k-lists in LDS, 8 = plain two-phase K1L (no K1P), 100 + ... = forced K1P layouts; 1 =
// popcount kernel K1 (anything else that is not listed runs K1 as well)
bool use_lookup(const gm_index* idx) {
    return idx->metric == GM_METRIC_HAMMING && idx->d_onehot != nullptr && (idx->variant == 0 || idx->variant == 7 || idx->variant == 8 || (idx->variant >= 102 && idx->variant < 1000 && idx->variant % 100 >= 2 && idx->variant % 100 <= GM_PG_MAXB));
}

// (one event pair per timed launch: with a reserved history the pair of the history slot serves gm_last_kernel_ms as
// well -- every event record is a marker packet in the queue and ~5 us of idle GPU around the dominant kernel)
int timing_begin(gm_index* idx, hipStream_t st) {
    idx->last_ev0 = idx->hist0.empty() ? idx->ev0 : idx->hist0[idx->hist_count % idx->hist0.size()];
    HIP_TRY(hipEventRecord(idx->last_ev0, st));
    return GM_OK;
}

int timing_end(gm_index* idx, hipStream_t st, int64_t pairs) {
    idx->last_ev1 = idx->hist1.empty() ? idx->ev1 : idx->hist1[idx->hist_count % idx->hist1.size()];
    HIP_TRY(hipEventRecord(idx->last_ev1, st));
    if (!idx->hist1.empty()) idx->hist_count++;
    idx->timed = true;
    idx->last_pairs = pairs;
    idx->last_evaluated = pairs;
    idx->last_incomplete = 0;
    return GM_OK;
}

void timing_clear(gm_index* idx) {
    idx->last_ev0 = idx->last_ev1 = nullptr;
    idx->timed = false;
    for (auto e : idx->hist0) (void)hipEventDestroy(e);
    for (auto e : idx->hist1) (void)hipEventDestroy(e);
    idx->hist0.clear();
    idx->hist1.clear();
    idx->hist_count = 0;
}

int check_index(const gm_index* idx) {
    if (!idx) return fail(GM_ERR_ARG, "null index handle");
    return GM_OK;
}

}  // namespace

extern "C" {

const char* gm_version(void) { return "gmhip 0.1.0 (gfx950)"; }

const char* gm_last_error(void) { return g_err.c_str(); }

int gm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gm_pack(const char* seqs, int64_t n, int guidelen, uint64_t* out) {
    if (!seqs || !out || n < 0 || guidelen < 1 || guidelen > GM_MAX_GUIDELEN)
        return fail(GM_ERR_ARG, "gm_pack: bad argument");
    static const struct Lut {
        uint8_t code[256];
        Lut() {
            for (int i = 0; i < 256; ++i) code[i] = 4;      // 4 = not a base
            code[(int)'A'] = 0;
            code[(int)'C'] = 1;
            code[(int)'G'] = 2;
            code[(int)'T'] = 3;
        }
    } lut;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t v = 0;
        unsigned bad = 0;
        const unsigned char* s = reinterpret_cast<const unsigned char*>(seqs) + i * guidelen;
        for (int p = 0; p < guidelen; ++p) {
            const unsigned c = lut.code[s[p]];
            bad |= c;
            v |= (uint64_t)(c & 3u) << (2 * p);
        }
        if (bad & 4u) return fail(GM_ERR_SEQ, "gm_pack: non-ACGT base in sequence " + std::to_string(i));
        out[i] = v;
    }
    return GM_OK;
}

int gm_unpack(const uint64_t* packed, int64_t n, int guidelen, char* out) {
    if (!packed || !out || n < 0 || guidelen < 1 || guidelen > GM_MAX_GUIDELEN)
        return fail(GM_ERR_ARG, "gm_unpack: bad argument");
    static const char B[4] = {'A', 'C', 'G', 'T'};
    for (int64_t i = 0; i < n; ++i)
        for (int p = 0; p < guidelen; ++p) out[i * guidelen + p] = B[(packed[i] >> (2 * p)) & 3u];
    return GM_OK;
}

// str(float) of Python for every value, separated by '\n' (host code: the reference's 'CFD Similar Guides' column is a
// list of str(float) per row, core.py:1136, and Python spends 2 s on 5.9 M of them).  Python's repr: the shortest
// decimal string that reads back as the same double -- std::to_chars' digits --, fixed notation for a decimal point
// position in (-4, 16], else exponent notation with at least two exponent digits.  Returns the bytes written, or -1
// when `cap` is too small (25 bytes per value always suffice).
static int64_t format_doubles_serial(const double* v, int64_t n, char* out, int64_t cap) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (w + (i ? 25 : 24) > cap) return -1;      // (a value takes at most 24 characters, + the separator before it)
        if (i) out[w++] = '\n';
        const double x = v[i];
        if (x != x) {
            std::memcpy(out + w, "nan", 3);
            w += 3;
            continue;
        }
        if (x == HUGE_VAL || x == -HUGE_VAL) {
            const char* t = x > 0 ? "inf" : "-inf";
            const size_t len = std::strlen(t);
            std::memcpy(out + w, t, len);
            w += (int64_t)len;
            continue;
        }
        char buf[40];
        auto res = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);      // [-]d[.ddd]e[+-]XX, shortest
        const char* p = buf;
        if (*p == '-') out[w++] = *p++;
        char digits[24];
        int nd = 0;
        while (p < res.ptr && *p != 'e') {
            if (*p != '.') digits[nd++] = *p;
            ++p;
        }
        int e10 = 0;
        if (p < res.ptr && *p == 'e') {
            ++p;
            const bool neg = *p == '-';
            if (*p == '+' || *p == '-') ++p;
            while (p < res.ptr) e10 = e10 * 10 + (*p++ - '0');
            if (neg) e10 = -e10;
        }
        const int decpt = e10 + 1;      // digits * 10^(decpt - nd): the decimal point sits after `decpt` digits
        if (nd == 1 && digits[0] == '0') {      // +-0.0
            std::memcpy(out + w, "0.0", 3);
            w += 3;
        } else if (decpt > -4 && decpt <= 16) {
            if (decpt <= 0) {
                out[w++] = '0';
                out[w++] = '.';
                for (int z = 0; z < -decpt; ++z) out[w++] = '0';
                std::memcpy(out + w, digits, (size_t)nd);
                w += nd;
            } else if (decpt >= nd) {
                std::memcpy(out + w, digits, (size_t)nd);
                w += nd;
                for (int z = 0; z < decpt - nd; ++z) out[w++] = '0';
                out[w++] = '.';
                out[w++] = '0';
            } else {
                std::memcpy(out + w, digits, (size_t)decpt);
                w += decpt;
                out[w++] = '.';
                std::memcpy(out + w, digits + decpt, (size_t)(nd - decpt));
                w += nd - decpt;
            }
        } else {
            out[w++] = digits[0];
            if (nd > 1) {
                out[w++] = '.';
                std::memcpy(out + w, digits + 1, (size_t)(nd - 1));
                w += nd - 1;
            }
            out[w++] = 'e';
            int ex = decpt - 1;
            out[w++] = ex < 0 ? '-' : '+';
            if (ex < 0) ex = -ex;
            char eb[8];
            int ne = 0;
            do {
                eb[ne++] = (char)('0' + ex % 10);
                ex /= 10;
            } while (ex);
            if (ne < 2) eb[ne++] = '0';
            while (ne) out[w++] = eb[--ne];
        }
    }
    return w;
}

int64_t gm_format_doubles(const double* v, int64_t n, char* out, int64_t cap) {
    if (!v || !out || n < 0) return -1;
    if (n == 0) return 0;
    if (cap < 24) return -1;
    // slices on the host's threads, each into its own 25-byte-per-value region of `out`, then closed up in order
    unsigned hw = std::thread::hardware_concurrency();
    int64_t n_thr = hw ? (int64_t)hw : 4;
    if (n_thr > 32) n_thr = 32;
    if (n_thr > n / 65536 + 1) n_thr = n / 65536 + 1;
    if (n_thr <= 1 || cap < 25 * n) return format_doubles_serial(v, n, out, cap);
    const int64_t per = (n + n_thr - 1) / n_thr;
    std::vector<int64_t> len((size_t)n_thr, 0);
    std::vector<std::thread> pool;
    for (int64_t t = 0; t < n_thr; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        pool.emplace_back([=, &len] { len[(size_t)t] = format_doubles_serial(v + lo, hi - lo, out + 25 * lo, 25 * (hi - lo)); });
    }
    for (auto& th : pool) th.join();
    int64_t w = 0;
    for (int64_t t = 0; t < (int64_t)pool.size(); ++t) {
        if (len[(size_t)t] < 0) return -1;
        if (t) out[w++] = '\n';
        std::memmove(out + w, out + 25 * (t * per), (size_t)len[(size_t)t]);
        w += len[(size_t)t];
    }
    return w;
}

// The two neighbour columns of the reference's output table (Annotation._format_guide_table, core.py:913-931), row by row
// from the k-NN result arrays: out_seqs = ";".join(sequences of the row's counts[r] hits), out_dists = ";".join(str(d)),
// rows separated by '\n' (host code, the host's threads).  Returns 0 and the two lengths, -1 when a buffer is too small
// (n * k * (guidelen + 1) and n * k * 12 bytes always suffice).
static void format_rows_serial(const uint64_t* db, int64_t n_db, int L, const int32_t* ids, const int32_t* dists,
                               const int32_t* counts, int64_t n, int k, char* os, char* od, int64_t* ls, int64_t* ld,
                               std::atomic<int>* bad) {
    // four bases per table look-up: byte of the packed word -> 4 characters
    static const struct Lut {
        uint32_t w[256];
        Lut() {
            static const char B[4] = {'A', 'C', 'G', 'T'};
            for (int v = 0; v < 256; ++v) {
                char c[4] = {B[v & 3], B[(v >> 2) & 3], B[(v >> 4) & 3], B[(v >> 6) & 3]};
                std::memcpy(&w[v], c, 4);
            }
        }
    } lut;
    int64_t ws = 0, wd = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (r) {
            os[ws++] = '\n';
            od[wd++] = '\n';
        }
        const int c = counts[r] < k ? counts[r] : k;
        for (int j = 0; j < c; ++j) {
            if (j) {
                os[ws++] = ';';
                od[wd++] = ';';
            }
            const int32_t id = ids[r * k + j];
            if (id < 0 || id >= n_db) {
                bad->store(1, std::memory_order_relaxed);      // (several slices may report it)
                continue;
            }
            uint64_t w = db[id];
            char seq[32];
            for (int p = 0; p < 28; p += 4, w >>= 8) std::memcpy(seq + p, &lut.w[w & 0xff], 4);
            std::memcpy(os + ws, seq, (size_t)L);
            ws += L;
            int d = dists[r * k + j];
            if (d >= 0 && d < 10) {
                od[wd++] = (char)('0' + d);
            } else if (d >= 10 && d < 100) {
                od[wd++] = (char)('0' + d / 10);
                od[wd++] = (char)('0' + d % 10);
            } else {
                if (d < 0) {
                    od[wd++] = '-';
                    d = -d;
                }
                char t[12];
                int nt = 0;
                do {
                    t[nt++] = (char)('0' + d % 10);
                    d /= 10;
                } while (d);
                while (nt) od[wd++] = t[--nt];
            }
        }
    }
    *ls = ws;
    *ld = wd;
}

int gm_format_neighbor_rows(const uint64_t* db_packed, int64_t n_db, int guidelen, const int32_t* ids, const int32_t* dists,
                            const int32_t* counts, int64_t n, int k, char* out_seqs, int64_t cap_seqs, char* out_dists,
                            int64_t cap_dists, int64_t* len_seqs, int64_t* len_dists) {
    if (!db_packed || !ids || !dists || !counts || !out_seqs || !out_dists || !len_seqs || !len_dists || n < 0 || k < 1 ||
        guidelen < 1 || guidelen > GM_MAX_GUIDELEN)
        return fail(GM_ERR_ARG, "gm_format_neighbor_rows: bad argument");
    *len_seqs = *len_dists = 0;
    if (n == 0) return GM_OK;
    const int64_t ss = (int64_t)k * (guidelen + 1), sd = (int64_t)k * 12;      // bytes per row at most (incl. the separator)
    if (cap_seqs < n * ss || cap_dists < n * sd) return fail(GM_ERR_ARG, "gm_format_neighbor_rows: buffer too small");
    std::atomic<int> bad{0};
    unsigned hw = std::thread::hardware_concurrency();
    int64_t n_thr = hw ? (int64_t)hw : 4;
    if (n_thr > 32) n_thr = 32;
    if (n_thr > n / 16384 + 1) n_thr = n / 16384 + 1;
    const int64_t per = (n + n_thr - 1) / n_thr;
    std::vector<int64_t> ls((size_t)n_thr, 0), ld((size_t)n_thr, 0);
    std::vector<std::thread> pool;
    std::atomic<int>* bad_p = &bad;
    for (int64_t t = 0; t < n_thr; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        pool.emplace_back([=, &ls, &ld] {
            format_rows_serial(db_packed, n_db, guidelen, ids + lo * k, dists + lo * k, counts + lo, hi - lo, k,
                               out_seqs + lo * ss, out_dists + lo * sd, &ls[(size_t)t], &ld[(size_t)t], bad_p);
        });
    }
    for (auto& th : pool) th.join();
    if (bad.load()) return fail(GM_ERR_ARG, "gm_format_neighbor_rows: id out of range");
    int64_t ws = 0, wd = 0;
    for (int64_t t = 0; t < (int64_t)pool.size(); ++t) {
        if (t) {
            out_seqs[ws++] = '\n';
            out_dists[wd++] = '\n';
        }
        std::memmove(out_seqs + ws, out_seqs + t * per * ss, (size_t)ls[(size_t)t]);
        std::memmove(out_dists + wd, out_dists + t * per * sd, (size_t)ld[(size_t)t]);
        ws += ls[(size_t)t];
        wd += ld[(size_t)t];
    }
    *len_seqs = ws;
    *len_dists = wd;
    return GM_OK;
}

int gm_index_create_dev(const void* d_packed, int64_t n_db, int guidelen, int metric, int device,
                        void* stream, gm_index** out) {
    GM_ENTER();
    if (!out) return fail(GM_ERR_ARG, "gm_index_create: null out");
    *out = nullptr;
    if (n_db < 0 || n_db > GM_MAX_DB) return fail(GM_ERR_ARG, "gm_index_create: n_db out of range");
    if (n_db > 0 && !d_packed) return fail(GM_ERR_ARG, "gm_index_create: null data");
    if (guidelen < 1 || guidelen > GM_MAX_GUIDELEN) return fail(GM_ERR_ARG, "gm_index_create: guidelen must be 1..27");
    if (metric != GM_METRIC_HAMMING && metric != GM_METRIC_LEVEN)
        return fail(GM_ERR_ARG, "gm_index_create: unknown metric");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GM_ERR_DEVICE, "no HIP device available (libgmhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GM_ERR_ARG, "gm_index_create: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    gm_index* idx = new gm_index();
    idx->device = device;
    idx->L = guidelen;
    idx->metric = metric;
    idx->n_db = n_db;
    idx->n_db_pad = n_db > 0 ? round_up(n_db, gm::TILE) : 0;
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        idx->n_simd = prop.multiProcessorCount * 4;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_db > 0) {
        if (hipMalloc(&idx->d_db, sizeof(uint2) * idx->n_db_pad) != hipSuccess) {
            delete idx;
            return fail(GM_ERR_ALLOC, "hipMalloc(db planes) failed");
        }
        int rc = planes_launch(static_cast<const uint64_t*>(d_packed), idx->d_db, n_db, idx->n_db_pad, st);
        if (rc != GM_OK) {
            gm_index_free(idx);
            return rc;
        }
        if (guidelen >= 10 && guidelen <= 27) {
            // one-hot table for the look-up kernel
            const int64_t lk_groups = round_up((n_db + gm::LK_GROUP - 1) / gm::LK_GROUP, 2 * gm::LK_TG);   // even tile count
            idx->n_lk_tiles = lk_groups / gm::LK_TG;
            const int64_t lk_words = lk_groups * gm::lk_group_words(guidelen);
            if (hipMalloc(&idx->d_onehot, sizeof(uint32_t) * lk_words) != hipSuccess) {
                gm_index_free(idx);
                return fail(GM_ERR_ALLOC, "hipMalloc(db one-hot table) failed");
            }
            hipLaunchKernelGGL(gm::k_to_onehot, dim3((unsigned)((lk_words + 255) / 256)), dim3(256), 0, st,
                               idx->d_db, n_db, idx->d_onehot, lk_groups, guidelen, 0, 0);
            if (hipGetLastError() != hipSuccess) {
                gm_index_free(idx);
                return fail(GM_ERR_DEVICE, "k_to_onehot launch failed");
            }
            if (metric == GM_METRIC_LEVEN) {   // K2L: per-target match masks
                const int64_t n_pad = round_up(n_db, gm::LV_TILE);
                idx->n_lv_tiles = n_pad / gm::LV_TILE;
                if (hipMalloc(&idx->d_peq, sizeof(uint4) * n_pad) != hipSuccess) {
                    gm_index_free(idx);
                    return fail(GM_ERR_ALLOC, "hipMalloc(db match masks) failed");
                }
                hipLaunchKernelGGL(gm::k_to_peq, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, idx->d_db,
                                   n_db, idx->d_peq, n_pad, guidelen);
                if (hipGetLastError() != hipSuccess) {
                    gm_index_free(idx);
                    return fail(GM_ERR_DEVICE, "k_to_peq launch failed");
                }
            }
        }
    }
    if (hipEventCreate(&idx->ev0) != hipSuccess || hipEventCreate(&idx->ev1) != hipSuccess ||
        hipEventCreate(&idx->evq0) != hipSuccess || hipEventCreate(&idx->evq1) != hipSuccess) {
        gm_index_free(idx);
        return fail(GM_ERR_DEVICE, "hipEventCreate failed");
    }
    *out = idx;
    return GM_OK;
}

int gm_index_create(const uint64_t* packed, int64_t n_db, int guidelen, int metric, int device,
                    gm_index** out) {
    GM_ENTER();
    if (!out) return fail(GM_ERR_ARG, "gm_index_create: null out");
    *out = nullptr;
    if (n_db < 0 || n_db > GM_MAX_DB) return fail(GM_ERR_ARG, "gm_index_create: n_db out of range");
    if (n_db > 0 && !packed) return fail(GM_ERR_ARG, "gm_index_create: null data");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GM_ERR_DEVICE, "no HIP device available (libgmhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GM_ERR_ARG, "gm_index_create: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    uint64_t* d_tmp = nullptr;
    if (n_db > 0) {
        if (hipMalloc(&d_tmp, sizeof(uint64_t) * n_db) != hipSuccess)
            return fail(GM_ERR_ALLOC, "hipMalloc(packed db) failed");
        hipError_t e = hipMemcpy(d_tmp, packed, sizeof(uint64_t) * n_db, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d_tmp);
            return fail(GM_ERR_DEVICE, std::string("hipMemcpy(db): ") + hipGetErrorString(e));
        }
    }
    int rc = gm_index_create_dev(d_tmp, n_db, guidelen, metric, device, nullptr, out);
    if (rc == GM_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(GM_ERR_DEVICE, "db upload failed");
    if (d_tmp) (void)hipFree(d_tmp);
    if (rc != GM_OK && *out) {
        gm_index_free(*out);
        *out = nullptr;
    }
    return rc;
}

void gm_index_free(gm_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    if (idx->d_db) (void)hipFree(idx->d_db);
    gm_lp_free(idx);
    gm_pg_free(idx);
    for (auto& lv : idx->pg_deeper) {      // the deeper cascade levels' K1P state
        idx->pg = lv.pg;
        lv.pg = nullptr;
        gm_pg_free(idx);
    }
    if (idx->d_onehot) (void)hipFree(idx->d_onehot);
    if (idx->d_peq) (void)hipFree(idx->d_peq);
    if (idx->d_keys) (void)hipFree(idx->d_keys);
    if (idx->d_keys2) (void)hipFree(idx->d_keys2);
    if (idx->d_lvthr) (void)hipFree(idx->d_lvthr);
    if (idx->d_q) (void)hipFree(idx->d_q);
    if (idx->d_stage) (void)hipFree(idx->d_stage);
    timing_clear(idx);
    if (idx->evq0) (void)hipEventDestroy(idx->evq0);
    if (idx->evq1) (void)hipEventDestroy(idx->evq1);
    if (idx->ev0) (void)hipEventDestroy(idx->ev0);
    if (idx->ev1) (void)hipEventDestroy(idx->ev1);
    delete idx;
}

int64_t gm_index_size(const gm_index* idx) { return idx ? idx->n_db : -1; }
int gm_index_guidelen(const gm_index* idx) { return idx ? idx->L : -1; }
int gm_index_metric(const gm_index* idx) { return idx ? idx->metric : -1; }

int gm_set_variant(gm_index* idx, int variant) {
    if (check_index(idx)) return GM_ERR_ARG;
    idx->variant = variant;
    idx->spec.valid = false;
    return GM_OK;
}

int gm_set_param(gm_index* idx, const char* name, int value) {
    if (check_index(idx)) return GM_ERR_ARG;
    if (!name) return fail(GM_ERR_ARG, "gm_set_param: null name");
    idx->spec.valid = false;      // (the follow-up counts learnt under the old settings say nothing about the new ones)
    if (std::strcmp(name, "pg_spec") == 0) {
        idx->pg_spec = value != 0;
        return GM_OK;
    }
    if (std::strcmp(name, "lk_seg") == 0) {
        idx->lk_seg = value;
        return GM_OK;
    }
    if (std::strcmp(name, "lv_seg") == 0) {
        idx->lv_seg = value;
        return GM_OK;
    }
    if (std::strcmp(name, "lv_two_pass") == 0) {
        idx->lv_two_pass = value;
        return GM_OK;
    }
    {
        bool known = false;
        const int rc = gm_lp_set_param(idx, name, value, &known);
        if (known) return rc;
    }
    return gm_pg_set_param(idx, name, value);
}

static int knn_query_dev_impl(gm_index* idx, const void* d_q, int64_t n_q, int k, void* d_ids, void* d_dists,
                              void* d_counts, void* stream);

int gm_knn_query_dev(gm_index* idx, const void* d_q, int64_t n_q, int k, void* d_ids, void* d_dists,
                     void* d_counts, void* stream) {
    GM_ENTER();
    if (check_index(idx)) return GM_ERR_ARG;
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipEventRecord(idx->evq0, reinterpret_cast<hipStream_t>(stream)));
    int rc = knn_query_dev_impl(idx, d_q, n_q, k, d_ids, d_dists, d_counts, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(idx->evq1, reinterpret_cast<hipStream_t>(stream)));
    return GM_OK;
}

static int knn_query_dev_impl(gm_index* idx, const void* d_q, int64_t n_q, int k, void* d_ids, void* d_dists,
                              void* d_counts, void* stream) {
    if (check_index(idx)) return GM_ERR_ARG;
    if (n_q < 0 || k < 1 || k > GM_MAX_K) return fail(GM_ERR_ARG, "gm_knn_query: need n_q >= 0, 1 <= k <= 64");
    if (n_q == 0) return GM_OK;
    if (!d_q || !d_ids || !d_dists || !d_counts) return fail(GM_ERR_ARG, "gm_knn_query: null buffer");
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* ids = static_cast<int32_t*>(d_ids);
    int32_t* dists = static_cast<int32_t*>(d_dists);
    int32_t* counts = static_cast<int32_t*>(d_counts);
    if (idx->n_db == 0) {   // empty index: no hits
        HIP_TRY(hipMemsetAsync(ids, 0xff, sizeof(int32_t) * n_q * k, st));
        HIP_TRY(hipMemsetAsync(dists, 0xff, sizeof(int32_t) * n_q * k, st));
        HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * n_q, st));
        return GM_OK;
    }
    int rc = ensure_q(idx, n_q);
    if (rc) return rc;
    rc = planes_launch(static_cast<const uint64_t*>(d_q), idx->d_q, n_q, n_q, st);
    if (rc) return rc;
    idx->casc_levels = 0;
    if (use_leven_lookup(idx, k)) return run_leven_lookup(idx, n_q, k, ids, dists, counts, nullptr, st);
    if (use_lookup(idx)) {
        if (idx->variant == 0 || idx->variant > 100) {      // (variant 8 = plain two-phase scan, the A/B baseline of K1P)
            bool used = false;
            int rc = run_hamming_pigeon(idx, false, n_q, k, ids, dists, counts, st, &used);
            if (rc || used) return rc;
        }
        return run_hamming_lookup(idx, false, n_q, k, ids, dists, counts, true, st);
    }
    rc = timing_begin(idx, st);
    if (rc) return rc;
    const int R = pick_r(k);
    if (idx->metric == GM_METRIC_HAMMING) {
        idx->last_kernel = "k_hamming_knn";
        rc = R == 4 ? launch_hamming_knn<4>(idx, n_q, k, ids, dists, counts, st)
           : R == 2 ? launch_hamming_knn<2>(idx, n_q, k, ids, dists, counts, st)
                    : launch_hamming_knn<1>(idx, n_q, k, ids, dists, counts, st);
    } else {
        idx->last_kernel = "k_leven_knn";
        rc = R == 4 ? launch_leven_knn<4>(idx, n_q, k, ids, dists, counts, st)
           : R == 2 ? launch_leven_knn<2>(idx, n_q, k, ids, dists, counts, st)
                    : launch_leven_knn<1>(idx, n_q, k, ids, dists, counts, st);
    }
    if (rc) return rc;
    return timing_end(idx, st, n_q * idx->n_db);
}

int gm_knn_query(gm_index* idx, const uint64_t* q, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                 int32_t* counts) {
    GM_ENTER();
    if (check_index(idx)) return GM_ERR_ARG;
    if (n_q < 0 || k < 1 || k > GM_MAX_K) return fail(GM_ERR_ARG, "gm_knn_query: need n_q >= 0, 1 <= k <= 64");
    if (n_q == 0) return GM_OK;
    if (!q || !ids || !dists || !counts) return fail(GM_ERR_ARG, "gm_knn_query: null buffer");
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t bq = round_up(sizeof(uint64_t) * n_q, 256);
    const int64_t bl = round_up(sizeof(int32_t) * n_q * k, 256);
    const int64_t bc = round_up(sizeof(int32_t) * n_q, 256);
    int rc = ensure_stage(idx, bq + 2 * bl + bc);
    if (rc) return rc;
    char* base = static_cast<char*>(idx->d_stage);
    HIP_TRY(hipMemcpy(base, q, sizeof(uint64_t) * n_q, hipMemcpyHostToDevice));
    rc = gm_knn_query_dev(idx, base, n_q, k, base + bq, base + bq + bl, base + bq + 2 * bl, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ids, base + bq, sizeof(int32_t) * n_q * k, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dists, base + bq + bl, sizeof(int32_t) * n_q * k, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(counts, base + bq + 2 * bl, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost));
    return GM_OK;
}

static int nearest_dist_dev_impl(gm_index* idx, const void* d_q, int64_t n_q, void* d_d1, void* stream);

int gm_nearest_dist_dev(gm_index* idx, const void* d_q, int64_t n_q, void* d_d1, void* stream) {
    GM_ENTER();
    if (check_index(idx)) return GM_ERR_ARG;
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipEventRecord(idx->evq0, reinterpret_cast<hipStream_t>(stream)));
    int rc = nearest_dist_dev_impl(idx, d_q, n_q, d_d1, stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(idx->evq1, reinterpret_cast<hipStream_t>(stream)));
    return GM_OK;
}

static int nearest_dist_dev_impl(gm_index* idx, const void* d_q, int64_t n_q, void* d_d1, void* stream) {
    if (check_index(idx)) return GM_ERR_ARG;
    if (n_q < 0) return fail(GM_ERR_ARG, "gm_nearest_dist: n_q < 0");
    if (n_q == 0) return GM_OK;
    if (!d_q || !d_d1) return fail(GM_ERR_ARG, "gm_nearest_dist: null buffer");
    HIP_TRY(hipSetDevice(idx->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int32_t* d1 = static_cast<int32_t*>(d_d1);
    if (idx->n_db == 0) {
        HIP_TRY(hipMemsetAsync(d1, 0xff, sizeof(int32_t) * n_q, st));
        return GM_OK;
    }
    int rc = ensure_q(idx, n_q);
    if (rc) return rc;
    rc = planes_launch(static_cast<const uint64_t*>(d_q), idx->d_q, n_q, n_q, st);
    if (rc) return rc;
    idx->casc_levels = 0;
    if (use_leven_lookup(idx, 1)) return run_leven_lookup(idx, n_q, 1, nullptr, nullptr, nullptr, d1, st);
    if (use_lookup(idx)) {
        if (idx->variant == 0 || idx->variant > 100) {
            bool used = false;
            int rc = run_hamming_pigeon(idx, true, n_q, 1, nullptr, d1, nullptr, st, &used);
            if (rc || used) return rc;
        }
        return run_hamming_lookup(idx, true, n_q, 1, nullptr, d1, nullptr, true, st);
    }
    int n_tiles = (int)(idx->n_db_pad / gm::TILE);
    rc = timing_begin(idx, st);
    if (rc) return rc;
    if (idx->metric == GM_METRIC_HAMMING) {
        constexpr int R = 4;
        int64_t blocks = (n_q + R * gm::WAVE - 1) / (R * gm::WAVE);
        idx->last_kernel = "k_hamming_nearest";
        hipLaunchKernelGGL((gm::k_hamming_nearest<R, 8>), dim3((unsigned)blocks), dim3(gm::WAVE), 0, st,
                           idx->d_db, n_tiles, idx->d_q, n_q, d1);
    } else {
        constexpr int R = 2;
        int64_t blocks = (n_q + R * gm::WAVE - 1) / (R * gm::WAVE);
        idx->last_kernel = "k_leven_nearest";
        hipLaunchKernelGGL((gm::k_leven_nearest<R>), dim3((unsigned)blocks), dim3(gm::WAVE), 0, st,
                           idx->d_db, (int)idx->n_db, n_tiles, idx->d_q, n_q, idx->L, d1);
    }
    HIP_TRY(hipGetLastError());
    return timing_end(idx, st, n_q * idx->n_db);
}

int gm_nearest_dist(gm_index* idx, const uint64_t* q, int64_t n_q, int32_t* d1) {
    GM_ENTER();
    if (check_index(idx)) return GM_ERR_ARG;
    if (n_q < 0) return fail(GM_ERR_ARG, "gm_nearest_dist: n_q < 0");
    if (n_q == 0) return GM_OK;
    if (!q || !d1) return fail(GM_ERR_ARG, "gm_nearest_dist: null buffer");
    HIP_TRY(hipSetDevice(idx->device));
    const int64_t bq = round_up(sizeof(uint64_t) * n_q, 256);
    int rc = ensure_stage(idx, bq + sizeof(int32_t) * n_q);
    if (rc) return rc;
    char* base = static_cast<char*>(idx->d_stage);
    HIP_TRY(hipMemcpy(base, q, sizeof(uint64_t) * n_q, hipMemcpyHostToDevice));
    rc = gm_nearest_dist_dev(idx, base, n_q, base + bq, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(d1, base + bq, sizeof(int32_t) * n_q, hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_cfd_batch_dev(const void* d_guides, const void* d_offs, int64_t n_pairs, int guidelen,
                     const void* d_mm, void* d_out, void* stream) {
    GM_ENTER();
    if (n_pairs < 0 || guidelen < 1 || guidelen > GM_MAX_GUIDELEN)
        return fail(GM_ERR_ARG, "gm_cfd_batch: bad argument");
    if (n_pairs == 0) return GM_OK;
    if (!d_guides || !d_offs || !d_mm || !d_out) return fail(GM_ERR_ARG, "gm_cfd_batch: null buffer");
    int64_t blocks = (n_pairs + 256 * gm::CFD_PER_THREAD - 1) / (256 * gm::CFD_PER_THREAD);
    hipLaunchKernelGGL(gm::k_cfd, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       static_cast<const uint64_t*>(d_guides), static_cast<const uint64_t*>(d_offs), n_pairs,
                       guidelen, static_cast<const double*>(d_mm), static_cast<double*>(d_out));
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

int gm_cfd_batch(const uint64_t* guides, const uint64_t* offs, int64_t n_pairs, int guidelen,
                 const double* mm, double* out, int device) {
    GM_ENTER();
    if (n_pairs < 0 || guidelen < 1 || guidelen > GM_MAX_GUIDELEN)
        return fail(GM_ERR_ARG, "gm_cfd_batch: bad argument");
    if (n_pairs == 0) return GM_OK;
    if (!guides || !offs || !mm || !out) return fail(GM_ERR_ARG, "gm_cfd_batch: null buffer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GM_ERR_DEVICE, "no HIP device available (libgmhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(GM_ERR_ARG, "gm_cfd_batch: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    const int64_t bs = round_up(sizeof(uint64_t) * n_pairs, 256);
    const int64_t bm = 320 * sizeof(double);
    char* base = nullptr;
    if (hipMalloc(&base, 3 * bs + bm) != hipSuccess) return fail(GM_ERR_ALLOC, "hipMalloc(cfd) failed");
    int rc = GM_OK;
    hipError_t e = hipMemcpy(base, guides, sizeof(uint64_t) * n_pairs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + bs, offs, sizeof(uint64_t) * n_pairs, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(base + 3 * bs, mm, bm, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = gm_cfd_batch_dev(base, base + bs, n_pairs, guidelen, base + 3 * bs, base + 2 * bs, nullptr);
        if (rc == GM_OK) e = hipMemcpy(out, base + 2 * bs, sizeof(double) * n_pairs, hipMemcpyDeviceToHost);
    }
    (void)hipFree(base);
    if (rc != GM_OK) return rc;
    if (e != hipSuccess) return fail(GM_ERR_DEVICE, std::string("gm_cfd_batch: ") + hipGetErrorString(e));
    return GM_OK;
}

int gm_pack_keys_dev(const void* d_ids, const void* d_dists, int64_t n_q, int k, void* d_keys, void* stream) {
    GM_ENTER();
    if (n_q < 0 || k < 1) return fail(GM_ERR_ARG, "gm_pack_keys: bad argument");
    if (n_q == 0) return GM_OK;
    if (!d_ids || !d_dists || !d_keys) return fail(GM_ERR_ARG, "gm_pack_keys: null buffer");
    const int64_t n = n_q * k;
    hipLaunchKernelGGL(gm::k_pack_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), static_cast<const int32_t*>(d_ids),
                       static_cast<const int32_t*>(d_dists), n, static_cast<uint32_t*>(d_keys));
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

int gm_unpack_keys_dev(const void* d_keys, int64_t n_q, int k, void* d_ids, void* d_dists, void* d_counts,
                       void* stream) {
    GM_ENTER();
    if (n_q < 0 || k < 1) return fail(GM_ERR_ARG, "gm_unpack_keys: bad argument");
    if (n_q == 0) return GM_OK;
    if (!d_keys || !d_ids || !d_dists || !d_counts) return fail(GM_ERR_ARG, "gm_unpack_keys: null buffer");
    hipLaunchKernelGGL(gm::k_unpack_keys, dim3((unsigned)((n_q + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), static_cast<const uint32_t*>(d_keys), n_q, k,
                       static_cast<int32_t*>(d_ids), static_cast<int32_t*>(d_dists), static_cast<int32_t*>(d_counts));
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

double gm_last_kernel_ms(gm_index* idx) {
    if (!idx || !idx->timed) return -1.0;
    if (hipSetDevice(idx->device) != hipSuccess) return -1.0;
    if (!idx->last_ev0 || !idx->last_ev1) return -1.0;
    if (hipEventSynchronize(idx->last_ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, idx->last_ev0, idx->last_ev1) != hipSuccess) return -1.0;
    return (double)ms;
}

int gm_timing_reserve(gm_index* idx, int n) {
    if (check_index(idx)) return GM_ERR_ARG;
    if (n < 0 || n > 4096) return fail(GM_ERR_ARG, "gm_timing_reserve: 0 <= n <= 4096");
    HIP_TRY(hipSetDevice(idx->device));
    timing_clear(idx);
    for (int i = 0; i < n; ++i) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
            return fail(GM_ERR_DEVICE, "hipEventCreate failed");
        idx->hist0.push_back(a);
        idx->hist1.push_back(b);
    }
    return GM_OK;
}

int gm_timing_read(gm_index* idx, double* ms_out, int max_n) {
    if (check_index(idx)) return GM_ERR_ARG;
    if (!ms_out || max_n < 0) return fail(GM_ERR_ARG, "gm_timing_read: bad argument");
    if (idx->hist0.empty()) return 0;
    if (hipSetDevice(idx->device) != hipSuccess) return fail(GM_ERR_DEVICE, "hipSetDevice failed");
    const int64_t cap = (int64_t)idx->hist0.size();
    const int64_t have = idx->hist_count < cap ? idx->hist_count : cap;
    const int64_t first = idx->hist_count - have;
    int n = 0;
    for (int64_t i = first; i < idx->hist_count && n < max_n; ++i, ++n) {
        float ms = 0.f;
        if (hipEventSynchronize(idx->hist1[i % cap]) != hipSuccess ||
            hipEventElapsedTime(&ms, idx->hist0[i % cap], idx->hist1[i % cap]) != hipSuccess)
            return fail(GM_ERR_DEVICE, "gm_timing_read: event query failed");
        ms_out[n] = (double)ms;
    }
    return n;
}

double gm_last_query_ms(gm_index* idx) {
    if (!idx || !idx->evq0) return -1.0;
    if (hipSetDevice(idx->device) != hipSuccess) return -1.0;
    if (hipEventSynchronize(idx->evq1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, idx->evq0, idx->evq1) != hipSuccess) return -1.0;
    return (double)ms;
}

const char* gm_last_kernel_name(gm_index* idx) { return idx ? idx->last_kernel : ""; }
int64_t gm_last_kernel_pairs(gm_index* idx) { return idx ? idx->last_pairs : 0; }
int64_t gm_last_kernel_evaluated(gm_index* idx) { return idx ? idx->last_evaluated : 0; }
int64_t gm_last_incomplete(gm_index* idx) { return idx ? idx->last_incomplete : 0; }
int gm_pigeon_blocks(gm_index* idx) { return idx ? gm_pg_blocks_of(idx) : 0; }
int gm_last_speculation(gm_index* idx) { return idx ? idx->spec_last : 0; }
int gm_pigeon_block_info(gm_index* idx, int b, int* positions, int64_t* pairs) {
    if (!idx || !positions || !pairs) return 0;
    return gm_pg_block_info(idx, b, positions, pairs);
}
int gm_pigeon_plan(int64_t n_db, int guidelen, int k, double p_match) {
    gm_pg_layout lay{};
    if (!gm_pg_plan(n_db, n_db, guidelen, k, p_match, &lay)) return 0;
    // B + 100 n1 + 1000 len1 (only when the radius-1 blocks are not of the even-split length) + 10000 n2 + 100000 len2
    int n1 = 0, n2 = 0, len1 = 0, len2 = 0;
    for (int b = 0; b < lay.B; ++b) {
        if (lay.trad[b] == 2) {
            ++n2;
            len2 = lay.blen[b];
        } else if (lay.trad[b] == 1) {
            ++n1;
            len1 = lay.blen[b] > len1 ? lay.blen[b] : len1;
        }
    }
    const int even = (guidelen + lay.B - 1) / lay.B;
    return lay.B + 100 * n1 + (n1 && len1 != even ? 1000 * len1 : 0) + 10000 * n2 + 100000 * len2;
}

int gm_pigeon_plan_dist(int64_t n_db, int64_t n_q, int guidelen, int k, double p_match, const double* p_short,
                        int levels, int* blen, int* trad, double* cost, int* cascade) {
    gm_pg_layout lay{};
    double c = 1.0;
    int casc = 0;
    if (levels < 1) levels = 1;
    if (levels > GM_PG_LEVELS) levels = GM_PG_LEVELS;
    const int found = gm_pg_plan(n_db, n_q, guidelen, k, p_match, &lay, p_short, levels - 1, &c, &casc);
    if (cost) *cost = c;
    if (cascade) *cascade = found ? casc : 0;
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (blen) blen[b] = found && b < lay.B ? lay.blen[b] : 0;
        if (trad) trad[b] = found && b < lay.B ? lay.trad[b] : 0;
    }
    return found ? lay.B : 0;
}

int gm_pigeon_plan_sample(int64_t n_db, int64_t n_q, int guidelen, int k, double p_match, const uint32_t* cum, int n_rows,
                          int levels, int* blen, int* trad, double* cost, int* cascade) {
    gm_pg_layout lay{};
    double c = 1.0;
    int casc = 0;
    if (levels < 1) levels = 1;
    if (levels > GM_PG_LEVELS) levels = GM_PG_LEVELS;
    gm_pg_sample smp{};
    smp.n = n_rows;
    smp.cum = cum;
    smp.k = k;
    const int found = gm_pg_plan(n_db, n_q, guidelen, k, p_match, &lay, nullptr, levels - 1, &c, &casc, cum && n_rows > 0 ? &smp : nullptr);
    if (cost) *cost = c;
    if (cascade) *cascade = found ? casc : 0;
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (blen) blen[b] = found && b < lay.B ? lay.blen[b] : 0;
        if (trad) trad[b] = found && b < lay.B ? lay.trad[b] : 0;
    }
    return found ? lay.B : 0;
}

int gm_levpart_layout(int guidelen, int r, double p_match, int* blen, int* n_copies, double* fraction) {
    int bs[GM_LP_MAXBLK] = {0}, bl[GM_LP_MAXBLK] = {0}, cb[GM_LP_MAXCOPY], cs[GM_LP_MAXCOPY], fc[GM_LP_MAXBLK + 1];
    double f = 0.0;
    if (!(p_match >= 0.25 && p_match < 0.9)) p_match = 0.25;
    const int n = gm_lp_layout(guidelen, r, p_match, bs, bl, cb, cs, fc, &f);
    if (n_copies) *n_copies = n;
    if (fraction) *fraction = n ? f : 0.0;
    for (int j = 0; j < GM_LP_MAXBLK; ++j)
        if (blen) blen[j] = n && j <= r ? bl[j] : 0;
    return n ? r + 1 : 0;
}

int gm_levpart_info(gm_index* idx, int* r1, int* r2, int64_t* rows2, int64_t* pairs2, int64_t* left2) {
    if (!idx) return 0;
    if (r1) *r1 = idx->lp_r1;
    if (r2) *r2 = idx->lp_r2;
    if (rows2) *rows2 = idx->lp_rows2;
    if (pairs2) *pairs2 = idx->lp_pairs2;
    if (left2) *left2 = idx->lp_left2;
    return (idx->lp_r1 > 0 ? 1 : 0) + (idx->lp_r2 > 0 ? 1 : 0);
}

int gm_pigeon_cascade_info(gm_index* idx, int level, int64_t* n_q, int* radius, int64_t* pairs, int64_t* left) {
    if (!idx || level < 0 || level >= idx->casc_levels || level >= GM_PG_LEVELS) return idx ? idx->casc_levels : 0;
    if (n_q) *n_q = idx->casc_nq[level];
    if (radius) *radius = idx->casc_radius[level];
    if (pairs) *pairs = idx->casc_pairs[level];
    if (left) *left = idx->casc_left[level];
    return idx->casc_levels;
}

}  // extern "C"
