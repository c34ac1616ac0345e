// gm_pigeon.hip -- host side of K1P, the pigeonhole-partitioned Hamming scan (kernels: k_hamming_scan_pg and
// k_merge_hits in gm_pigeon_scan.hip.h).  Exact k-NN without evaluating all pairs:
//
//   two sequences with at most r mismatches agree exactly on at least one of r + 1 disjoint blocks of
//   positions; with the one-substitution neighbour buckets of X of the blocks scanned as well, B blocks
//   serve r = B + X - 1 (see gm_pg_plan).  Pool side (once per index and layout): for every block a stable
//   sort of the pool by the block's value (ids ascend inside a bucket), every bucket padded to a multiple of 128
//   positions (one LDS-DMA unit of the scan; padding matches nothing), a record {planes, pool id} per position and
//   a one-hot table of the copy.  Query side: the candidates are grouped by the same values and cut into chunks of
//   <= 64 that do not straddle buckets; one workgroup scans one chunk against a run of target buckets (its own
//   bucket and / or neighbour buckets of an expansion block) and appends its hits to the candidates' hit lists; the
//   lists of a candidate are merged (duplicates dropped).  Work on S1: 5 % of all pairs (L = 20, blocks 5,5,5,5,
//   three of them expanded).  A candidate whose merged list is full is exact: every target within r was
//   enumerated.  The rest (the layout is chosen so that they are few) are re-scanned by the plain two-phase
//   kernel (gmhip.hip).
//
// Everything here is integer bookkeeping on the device: block keys, a stable rocPRIM sort per pool copy, one
// counting sort of the candidates for all blocks, a chunk table built by a count / scan / fill triple.  The only
// host round trip per query is the read-back of the incomplete-row count.
#include "gm_host.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gm_pigeon_scan.hip.h"      // kernels, geometry

namespace {

struct PgLayoutArg {              // per-block geometry + offsets, by value
    int B;
    int ebase[GM_PG_MAXB + 1];    // first entry of block b in the flattened (block, bucket) list
    const int32_t* coff[GM_PG_MAXB];
    const int2* bdesc[GM_PG_MAXB];
    int blen[GM_PG_MAXB];         // bases of block b
    int nbk[GM_PG_MAXB];          // target buckets per candidate bucket of block b (own + neighbours within its radius)
    int nsub[GM_PG_MAXB];         // workgroups those target buckets are split over
    int per[GM_PG_MAXB];          // target buckets per such workgroup
    int lbase[GM_PG_MAXB];        // hit list of (b, s) = lbase[b] + s
    int smax;                     // max nsub
    int skiplen[GM_PG_MAXB];      // positions of its own block copy b's table leaves out
    int n_entries;
};

struct PgCounters {               // device-resident
    int cstart[GM_PG_MAXB + 1];   // chunk-table range of block b: [cstart[b], cstart[b + 1]); cstart[B] = all chunks
    int n_chunks;
    int n_inc;
    unsigned long long pairs;     // pairs the last K1P launch evaluates (chunks x 64 x scanned groups x 64)
    unsigned long long bpairs[GM_PG_MAXB];      // ... per block
    int n_bad;                    // re-scanned rows that did not fill under the guessed bound
    int n_bad_pub;                // ... as of the last k_publish (which re-zeroes n_bad): what a plain copy of the counters reads
    int n_over;                   // incomplete rows whose hit lists overflowed (nothing about them is known)
    int chist[24];                // the other incomplete rows by the number c < k of neighbours they have within r
};

// The two per-query read-backs (rows left incomplete by the merge; re-scanned rows that did not fill) go through a
// mailbox in pinned, host-coherent memory: a one-wave kernel copies the counters into it and then raises a sequence
// number; the host polls that word.  A hipMemcpyAsync + hipStreamSynchronize pair costs ~25 us of idle GPU per
// read-back on this platform (copy kernel, interrupt, wake-up, next launch), the mailbox ~15.
struct PgMailbox {
    PgCounters c;
    volatile unsigned seq;
    unsigned pad[3];
};
constexpr int PG_BOXES = 4;

__global__ __launch_bounds__(64) void k_publish(PgCounters* __restrict__ src, PgMailbox* __restrict__ box, unsigned seq) {
    constexpr int W = (int)(sizeof(PgCounters) / sizeof(int));
    const int* s = reinterpret_cast<const int*>(src);
    volatile int* d = reinterpret_cast<volatile int*>(&box->c);
    for (int i = threadIdx.x; i < W; i += 64) d[i] = s[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        box->seq = seq;
        src->n_bad_pub = src->n_bad;
        src->n_bad = 0;      // (read: the next count starts from zero without a memset of its own)
    }
}

__global__ __launch_bounds__(256) void k_block_keys(const uint2* __restrict__ planes, int64_t n, int start, int len,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 e = planes[i];
    uint32_t key = 0;
    for (int j = 0; j < len; ++j) {
        const int p = start + j;
        key |= (((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1)) << (2 * j);
    }
    keys[i] = key;
    rows[i] = (uint32_t)i;
}

// off[v] = first position whose key is >= v (v = 0 .. nb); keys ascending
__global__ __launch_bounds__(256) void k_bucket_offsets(const uint32_t* __restrict__ keys, int64_t n, int nb,
                                                        int32_t* __restrict__ off) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nb) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)v) lo = mid + 1; else hi = mid;
    }
    off[v] = (int32_t)lo;
}

// base composition of the pool: counts[c] = occurrences of base code c over all positions
__global__ __launch_bounds__(256) void k_base_counts(const uint2* __restrict__ planes, int64_t n, int L,
                                                     unsigned long long* __restrict__ counts) {
    __shared__ unsigned s[4];
    if (threadIdx.x < 4) s[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t mask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    unsigned c[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 e = planes[i];
        c[0] += __popc(~e.x & ~e.y & mask);
        c[1] += __popc(e.x & ~e.y & mask);
        c[2] += __popc(~e.x & e.y & mask);
        c[3] += __popc(e.x & e.y & mask);
    }
    for (int b = 0; b < 4; ++b) atomicAdd(&s[b], c[b]);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

#include "gm_group.hip.h"      // PgKeyArg, PgSubArg, k_cand_hist / k_cand_offsets / k_cand_scatter, gm_group_launch

// sorted position i (key v = keys[i], pool row rows[i]) -> position bdesc[v].x * 64 + (i - raw_off[v]) of the padded copy
__global__ __launch_bounds__(256) void k_scatter_recs(const uint2* __restrict__ planes, const uint32_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ rows, int64_t n,
                                                      const int32_t* __restrict__ raw_off, const int2* __restrict__ bdesc,
                                                      uint4* __restrict__ recs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = keys[i], row = rows[i];
    const int64_t pos = (int64_t)bdesc[v].x * gm::LK_GROUP + (i - raw_off[v]);
    const uint2 e = planes[row];
    recs[pos] = make_uint4(e.x, e.y, row, 0u);
}

// one-hot table of a padded copy (format of gm::k_to_onehot; padding positions -- id 0xffffffff -- set no bit, so
// they match nothing); the table leaves out sequence positions [skip_start, skip_start + skip_len), L = positions kept
__global__ __launch_bounds__(256) void k_onehot_recs(const uint4* __restrict__ recs, uint32_t* __restrict__ table,
                                                     int64_t n_groups, int L, int skip_start, int skip_len) {
    const int gw = gm::lk_group_words(L);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_groups * gw) return;
    const int64_t G = t / gw;
    const int w = (int)(t % gw);
    int p = w >> 3;
    const int b = (w >> 1) & 3, h = w & 1;
    if (p >= skip_start) p += skip_len;
    uint32_t out = 0;
    for (int j = 0; j < 32; ++j) {
        const uint4 e = recs[G * gm::LK_GROUP + 32 * h + j];
        const uint32_t code = ((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1);
        out |= (uint32_t)(e.z != 0xffffffffu && code == (uint32_t)b) << j;
    }
    table[t] = out;
}

__device__ __forceinline__ void pg_entry(const PgLayoutArg& A, int e, int& b, int& v) {
    b = 0;
    while (b + 1 < A.B && e >= A.ebase[b + 1]) ++b;
    v = e - A.ebase[b];
}

// gsum[entry x smax + s] = the 64 x 64 groups one chunk of that (block, bucket) entry evaluates in run s of its target
// buckets.  A property of the pool copies and the launch geometry, not of the candidates: computed when those change
// (gm_pg_candidates keeps a signature), not per query -- the walk over a radius-2 block's 211 target buckets per thread
// made k_fill_chunks 61 us of every S4 step.
__global__ __launch_bounds__(256) void k_group_sums(PgLayoutArg A, uint32_t* __restrict__ gsum) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / A.smax, s = t % A.smax;
    if (e >= A.n_entries) return;
    int b = 0, v = 0;
    pg_entry(A, e, b, v);
    uint32_t groups = 0;
    if (s < A.nsub[b]) {
        const int n_bk = A.nbk[b], per = A.per[b];
        const int e_lo = s * per, e_hi = e_lo + per < n_bk ? e_lo + per : n_bk;
        for (int u = e_lo; u < e_hi; ++u) {
            int add;
            groups += (uint32_t)A.bdesc[b][gm::pg_target_bucket(v, u, A.blen[b], A.skiplen[b], add)].y;
        }
    }
    gsum[t] = groups;
}

// one thread per ((block, bucket) entry, run of target buckets)
__global__ __launch_bounds__(256) void k_fill_chunks(PgLayoutArg A, const int* __restrict__ cbase, const uint32_t* __restrict__ gsum,
                                                     gm_pg_chunk* __restrict__ chunks, PgCounters* __restrict__ ctr) {
    __shared__ unsigned long long s_pairs[GM_PG_MAXB];      // pairs per block, summed per workgroup first: thousands of
    if (threadIdx.x < GM_PG_MAXB) s_pairs[threadIdx.x] = 0ull;      // 64-bit atomics on one address take ~150 us
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / A.smax, s = t % A.smax;
    int b = 0, v = 0;
    bool live = e < A.n_entries;
    if (live) {
        pg_entry(A, e, b, v);
        live = s < A.nsub[b];
    }
    int c0 = 0, c1 = 0;
    if (live) {
        c0 = A.coff[b][v];
        c1 = A.coff[b][v + 1];
        live = c1 > c0;
    }
    if (live) {
        const int n_bk = A.nbk[b], per = A.per[b];
        const int n_ch = (c1 - c0 + 63) / 64;
        const int e_lo = s * per, e_hi = e_lo + per < n_bk ? e_lo + per : n_bk;
        const int lst = A.lbase[b] + s;
        int w = ctr->cstart[b] + cbase[e] + s * n_ch;      // (cbase: offset inside the block)
        for (int c = c0; c < c1; c += 64, ++w)
            chunks[w] = gm_pg_chunk{b, v, c, c1 - c < 64 ? c1 - c : 64, e_lo, e_hi, lst, 0};
        // the 64 x 64 groups the scan evaluates for this run
        atomicAdd(&s_pairs[b], (unsigned long long)n_ch * 64ull * (unsigned long long)gsum[t] * gm::LK_GROUP);
    }
    __syncthreads();
    if (threadIdx.x < GM_PG_MAXB && s_pairs[threadIdx.x] != 0ull)
        atomicAdd(&ctr->bpairs[threadIdx.x], s_pairs[threadIdx.x]);      // (summed over the blocks on the host)
}

__global__ __launch_bounds__(256) void k_count_incomplete(const int32_t* __restrict__ counts, const int32_t* __restrict__ d1,
                                                          int64_t n, int need, int* __restrict__ n_bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && (counts != nullptr ? counts[i] < need : d1[i] < 0);
    const uint64_t bal = __builtin_amdgcn_ballot_w64(bad);      // one atomic per wave that has any
    if (bal != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(n_bad, __builtin_popcountll(bal));
}

__global__ __launch_bounds__(256) void k_stride_rows(int32_t* __restrict__ inc, int64_t n, int64_t stride) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) inc[j] = (int32_t)(j * stride);
}

__global__ __launch_bounds__(256) void k_gather_rows(const uint2* __restrict__ q, const int32_t* __restrict__ inc, int64_t n,
                                                     uint2* __restrict__ q2, const int* __restrict__ n_dev) {
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;      // (speculative follow-ups: the launch was sized for an upper bound)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) q2[j] = q[(uint32_t)inc[j] & ~gm::PG_INC_OVERFLOW];
}

__global__ __launch_bounds__(256) void k_scatter_rows(const int32_t* __restrict__ inc, int64_t n, int k,
                                                      const int32_t* __restrict__ ids2, const int32_t* __restrict__ dists2,
                                                      const int32_t* __restrict__ counts2, int32_t* __restrict__ ids,
                                                      int32_t* __restrict__ dists, int32_t* __restrict__ counts,
                                                      const int* __restrict__ n_dev) {
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const int64_t j = t / k;
    const int s = (int)(t % k);
    const int64_t row = (int64_t)((uint32_t)inc[j] & ~gm::PG_INC_OVERFLOW);
    if (ids != nullptr) ids[row * k + s] = ids2[t];
    dists[row * k + s] = dists2[t];
    if (counts != nullptr && s == 0) counts[row] = counts2[j];
}


// rows that are still short after the prefix re-scan (count below `need` / no distance): their ORIGINAL row numbers,
// compacted into out (order irrelevant), counted in *n_out
__global__ __launch_bounds__(256) void k_collect_bad(const int32_t* __restrict__ inc, int64_t n, int need,
                                                     const int32_t* __restrict__ counts2, const int32_t* __restrict__ d1,
                                                     int32_t* __restrict__ out, int* __restrict__ n_out,
                                                     const int* __restrict__ n_dev) {
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool take = j < n && (counts2 != nullptr ? counts2[j] < need : d1[j] < 0);
    gm::gm_block_append(take, take ? (int32_t)((uint32_t)inc[j] & ~gm::PG_INC_OVERFLOW) : 0, out, n_out);
}


// Planner sample: hist[s][d] = pool targets at Hamming distance d <= PG_SAMPLE_DMAX of sample row s (lane = target, the
// sample rows broadcast from LDS, counters in LDS, one flush of the non-zero counters per workgroup).  2 048 rows x
// 1.2 M targets = 2.4 10^9 popcount evaluations, ~0.4 ms once per plan: it gives the planner, per sample row, the
// distance of its k-th neighbour AND the number of hits a layout of radius r would have to store for it.
constexpr int PG_SAMPLE_DMAX = 8;
constexpr int PG_SAMPLE_BINS = PG_SAMPLE_DMAX + 1;
__global__ __launch_bounds__(1024) void k_sample_hist(const uint2* __restrict__ planes, int64_t n, const uint2* __restrict__ qs,
                                                      int n_s, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t s_mem[];      // [n_s * 2] sample planes, then [n_s * BINS] counters
    uint2* s_q = reinterpret_cast<uint2*>(s_mem);
    uint32_t* s_h = s_mem + 2 * n_s;
    for (int i = threadIdx.x; i < n_s; i += blockDim.x) s_q[i] = qs[i];
    for (int i = threadIdx.x; i < n_s * PG_SAMPLE_BINS; i += blockDim.x) s_h[i] = 0u;
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const uint2 e = planes[t];
        for (int sIdx = 0; sIdx < n_s; ++sIdx) {
            const uint2 c = s_q[sIdx];      // (wave-uniform address: a broadcast read)
            const int d = __popc((c.x ^ e.x) | (c.y ^ e.y));
            if (d <= PG_SAMPLE_DMAX) atomicAdd(&s_h[sIdx * PG_SAMPLE_BINS + d], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_s * PG_SAMPLE_BINS; i += blockDim.x)
        if (s_h[i] != 0u) atomicAdd(&hist[i], s_h[i]);
}

// id cut-off: per bucket of a pool copy, the number of its 64-target groups that hold a target with id < cut_id (ids
// ascend inside a bucket: binary search for the first id >= cut_id)
__global__ __launch_bounds__(256) void k_bucket_cut(const uint4* __restrict__ recs, const int2* __restrict__ bdesc, int nb,
                                                    uint32_t cut_id, int* __restrict__ bcut) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nb) return;
    const int2 d = bdesc[v];
    const int64_t base = (int64_t)d.x * gm::LK_GROUP;
    int lo = 0, hi = d.y * gm::LK_GROUP;      // positions of the bucket (padding = id 0xffffffff sorts last)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (recs[base + mid].z < cut_id) lo = mid + 1; else hi = mid;
    }
    bcut[v] = (lo + gm::LK_GROUP - 1) / gm::LK_GROUP;      // lo = targets below the cut-off
}


// id cut-off, per candidate: under the independent-position model with the pool's base frequencies, a candidate with
// bases b_p mismatches a random target at position p with probability 1 - freq[b_p]; lam_lt / lam_r = expected targets
// closer than r / at exactly r.  The candidate takes the cut-off (hits at r only from ids below f n_db) when that still
// leaves it f lam_r >= (k - lam_lt)+ + 2 sqrt(k) + 1 of them.
struct PgFreqArg {
    float miss[4];      // 1 - frequency of base b in the pool
};
__global__ __launch_bounds__(256) void k_cand_cutflag(const uint2* __restrict__ planes, int64_t n, int L, PgFreqArg F, float n_db,
                                                      int r, int k, float f, uint8_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 e = planes[i];
    float dp[9];      // P(j mismatches so far), j = 0 .. 8 (r <= 8: three blocks of radius 2)
#pragma unroll
    for (int j = 0; j < 9; ++j) dp[j] = j == 0 ? 1.0f : 0.0f;
    for (int p = 0; p < L; ++p) {
        const int b = (int)(((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1));
        const float m = F.miss[b];
#pragma unroll
        for (int j = 8; j >= 1; --j) dp[j] = dp[j] * (1.0f - m) + dp[j - 1] * m;
        dp[0] *= 1.0f - m;
    }
    float lt = 0.0f, at = 0.0f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        if (j < r) lt += dp[j];
        if (j == r) at = dp[j];
    }
    const float lam_lt = n_db * lt + 1.0f, lam_r = n_db * at;
    const float need = (float)k - lam_lt > 0.0f ? (float)k - lam_lt : 0.0f;
    flag[i] = f * lam_r >= need + 2.0f * sqrtf((float)k) + 1.0f ? 1 : 0;
}


// rows i of [0, n) with counts[i] < need, compacted (order irrelevant) into out, counted in *n_out
__global__ __launch_bounds__(256) void k_collect_short(const int32_t* __restrict__ counts, int64_t n, int need,
                                                       int32_t* __restrict__ out, int* __restrict__ n_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    gm::gm_block_append(i < n && counts[i] < need, (int32_t)i, out, n_out);
}

template <class T>
int ensure(T** p, int64_t* cap, int64_t need, const char* what) {
    if (need <= *cap && *p != nullptr) return GM_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    if (need < 1) need = 1;
    if (hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (size_t)need) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, std::string("hipMalloc(") + what + ") failed");
    *cap = need;
    return GM_OK;
}

inline dim3 grid_for(int64_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

// the pool side of one layout (everything gm_pg_prepare builds; it depends on the block lengths only), kept aside while
// another layout is in use: get_control_seqs queries the same index with 10 n, 100 n, 1000 n ... candidates and
// get_neighbors with yet another count -- each may plan other block lengths, and rebuilding the copies (radix sorts +
// host round trips) per call would cost more than the queries
struct PgPoolSide {
    int B = 0;
    int blen[GM_PG_MAXB]{}, bstart[GM_PG_MAXB]{}, ebase[GM_PG_MAXB + 1]{}, skiplen[GM_PG_MAXB]{}, Lp[GM_PG_MAXB]{}, max_bucket[GM_PG_MAXB]{};
    uint32_t* table[GM_PG_MAXB]{};
    uint4* recs[GM_PG_MAXB]{};
    int2* bdesc[GM_PG_MAXB]{};
    int32_t* coff[GM_PG_MAXB]{};
    int* bcut[GM_PG_MAXB]{};
    uint32_t bcut_for = 0xffffffffu;
    int gap_mode = 0;
    uint64_t stamp = 0;      // last use (the least recently used side is dropped)
};
constexpr int PG_POOL_SIDES = 3;      // layouts whose pool copies are kept per K1P level (the one in use included)

struct gm_pigeon {
    int B = 0;
    std::vector<PgPoolSide> stash;     // pool sides of other layouts
    uint64_t use_stamp = 0;
    double p_match = 0.0;              // P(two pool sequences carry the same base at a position); 0 = not measured
    int max_bucket[GM_PG_MAXB]{};      // largest bucket of every pool copy
    int trad[GM_PG_MAXB]{};            // search radius of every block (gm_pg_layout): the layout serves
                                       // r = sum(trad[b] + 1) - 1
    int nbk[GM_PG_MAXB]{}, nsub[GM_PG_MAXB]{}, per[GM_PG_MAXB]{}, lbase[GM_PG_MAXB]{};      // see PgLayoutArg
    int smax = 1;
    int skiplen[GM_PG_MAXB]{};         // positions of its own block copy b's table leaves out
    int Lp[GM_PG_MAXB]{};              // positions the scan evaluates under block b = L - skiplen[b] (>= 10)
    int64_t bgrid[GM_PG_MAXB]{};       // upper bound of block b's chunk count
    int64_t last_bpairs[GM_PG_MAXB]{}; // pairs block b's part of the last launch evaluated
    int tune_sub = 0;                  // workgroups per expansion chunk (gm_set_param "pg_sub"); 0 = choose: 2 on the first
                                       // cascade level, 8 (two target buckets per workgroup, up to six workgroups per chunk)
                                       // on the deeper ones: their few, uneven chunks want finer work items (PAO1 2.99 ->
                                       // 2.85 ms/step; first levels of 100 k .. 300 k candidates lose 3 .. 5 % with 8)
    bool deep = false;                 // the state of a cascade level >= 1 (gm_pg_inherit)
    int tune_per = 0;                  // target buckets per workgroup, 0 = 16 / pg_sub (gm_set_param "pg_per")
    int tune_hit = 1;                  // hit path (gm_set_param "pg_hit"): 0 = hit buffer in HBM (k_hamming_scan_pg), 2 = register lists
                                       // (k_hamming_scan_pg4), 1 = chosen per level (pg_use_reglists)
    bool hit_reg = false;              // the hit path of the launch being prepared / merged
    int tune_layout = 0;               // experiments: forced layout, decimal digit pairs (len, radius) per block
    int tune_gap = 0;                  // table gaps: 0 = choose, 1 = common gap (one launch), 2 = per-block gaps
    int tune_run = 0;                  // chunk-table entries per XCD run (gm_set_param "pg_run"); 0 = choose: 128, or 32 with
                                       // register lists (one workgroup per chunk: a run of 32 covers the chunks of ~10 buckets;
                                       // PAO1 / S3 level 0 -2 .. -4 %, S1 equal: profiles/r04/NOTES.md)
    int tune_msplit = 0;               // lanes per candidate in the merge, 0 = choose (gm_set_param "pg_msplit")
    int tune_sample = 1;               // 0 = the layout planner uses the Poisson model instead of a scanned sample (gm_set_param "pg_sample")
    int tune_prefix = 1;               // 0 = the incomplete rows go straight to the plain re-scan (gm_set_param "pg_prefix")
    int tune_cut = 0;                  // id cut-off: 0 = choose, 1 .. 99 = that percentage of the ids, 100 = off ("pg_cut")
    double expect_hits = 0.0;          // targets within the layout's radius per candidate, from the planner's sample (0: unknown)
    bool cut_dropped = false;          // the cut-off left more than 1 % of the candidates short (beyond expect_short) on this pool: not used again
    double expect_short = 0.0;         // share of the candidates the planner's sample expects short WITHOUT the cut-off (0: no sample)
    double base_freq[4] = {0.25, 0.25, 0.25, 0.25};      // base composition of the pool (gm_pg_pool_stats)
    uint8_t* cflag = nullptr;          // per candidate: takes the id cut-off
    int64_t cflag_cap = 0;
    uint32_t* shist = nullptr;         // planner sample: distance histogram per sample row
    int64_t shist_cap = 0;
    double cut_f = 1.0;                // fraction of the ids whose distance-r targets are taken (1.0 = no cut-off)
    uint32_t cut_id = 0xffffffffu;     // ... as an id
    int* bcut[GM_PG_MAXB]{};           // per copy: groups per bucket that hold an id below the cut-off
    uint32_t bcut_for = 0xffffffffu;   // the cut-off id the bcut arrays were built for
    int bstart[GM_PG_MAXB]{}, blen[GM_PG_MAXB]{};
    int ebase[GM_PG_MAXB + 1]{};
    // pool side
    uint32_t* table[GM_PG_MAXB]{};
    uint4* recs[GM_PG_MAXB]{};
    int2* bdesc[GM_PG_MAXB]{};
    int32_t* raw_off = nullptr;        // build scratch: bucket offsets of the unpadded sorted order
    int64_t raw_off_cap = 0;
    // query side (grow-only)
    uint32_t* cperm[GM_PG_MAXB]{};
    int64_t cperm_cap[GM_PG_MAXB]{};
    uint32_t* ctmp[GM_PG_MAXB]{};      // two-level grouping (gm_group.hip.h): first-pass entries per block; partition counters
    int64_t ctmp_cap[GM_PG_MAXB]{};
    int* part_words = nullptr;
    int64_t part_words_cap = 0;
    int32_t* inc_slots = nullptr;      // the merge's incomplete rows per workgroup (gm::pg_block_stash) and their counters
    int64_t inc_slots_cap = 0;
    int* inc_meta = nullptr;
    int64_t inc_meta_cap = 0;
    uint32_t* gsum = nullptr;          // k_group_sums: groups per chunk of every (entry, run of target buckets) ...
    int64_t gsum_cap = 0;
    std::vector<int64_t> gsum_sig;     // ... for this geometry and these pool copies
    int tune_slots = 0;                // gm_set_param "pg_slots": the merge's list of incomplete rows: 0 = choose, 1 = atomics, 2 = workgroup slots
    int tune_group = 0;                // gm_set_param "pg_group": 0 = choose (two levels from 262 144 candidates), 1 = one level, 2 = two levels
    int32_t* coff[GM_PG_MAXB]{};
    uint32_t *k0 = nullptr, *k1 = nullptr, *r0 = nullptr, *r1 = nullptr;
    int64_t k0_cap = 0, k1_cap = 0, r0_cap = 0, r1_cap = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_cap = 0;
    bool hist_dirty = true;       // the histogram buffer has not been zeroed yet (the offsets kernel re-zeroes it after use)
    int *nch = nullptr, *cbase = nullptr, *hist = nullptr, *cursor = nullptr;
    int64_t nch_cap = 0, cbase_cap = 0, hist_cap = 0, cursor_cap = 0;
    gm_pg_chunk* chunks = nullptr;
    int64_t chunks_cap = 0;
    PgCounters* ctr = nullptr;
    PgMailbox* box = nullptr;         // pinned, host-coherent mailboxes of the per-query read-backs (k_publish): [PG_BOXES];
                                      // slot 0 = the synchronous read-backs, 1 .. = the speculative run's (one per kind)
    PgMailbox* box_dev = nullptr;     // ... their device address
    unsigned box_seq = 0;
    PgCounters last_c{};              // host copy of the counters as of the last completed after-merge read-back
    bool box_off = false;             // the mailbox word once failed to arrive while the stream ran: plain copies from then on
    int64_t grid = 0;
    uint32_t* lists = nullptr;    // hit lists (gm_pg_blocks::loff / lcap)
    int64_t lists_cap = 0;
    uint32_t* hdr = nullptr;      // one header word per (candidate, list) (gm_pg_blocks::hdr)
    int64_t hdr_cap = 0;
    uint32_t* keys3 = nullptr;    // LDS-list scan: raw lists [n_lists][n_q][kcap]
    int64_t keys3_cap = 0;
    int64_t cap_nq = -1;          // candidate count / layout the list capacities were sized for
    int cap_sub_for = -1;
    uint64_t cap_layout_for = 0;
    double mem_limit = 0.0;
    gm_pg_blocks blocks{};
    int32_t* inc = nullptr;
    int64_t inc_cap = 0;
    int32_t* inc_b = nullptr;     // second list of incomplete rows (the prefix re-scan's leftovers; swapped with inc)
    int64_t inc_b_cap = 0;
    uint2* q2 = nullptr;
    int64_t q2_cap = 0;
    int32_t *ids2 = nullptr, *dists2 = nullptr, *counts2 = nullptr;
    int64_t ids2_cap = 0, dists2_cap = 0, counts2_cap = 0;
};

namespace {

// the device counters + the host mailbox of the read-backs (also used by the Levenshtein passes, which may run without a
// K1P layout ever having been prepared on the handle)
int pg_ensure_counters(gm_pigeon* pg) {
    if (!pg->ctr) {
        if (hipMalloc(reinterpret_cast<void**>(&pg->ctr), sizeof(PgCounters)) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P counters) failed");
        if (hipMemset(pg->ctr, 0, sizeof(PgCounters)) != hipSuccess) return gm_fail(GM_ERR_DEVICE, "hipMemset(K1P counters) failed");
    }
    if (!pg->box) {
        if (hipHostMalloc(reinterpret_cast<void**>(&pg->box), sizeof(PgMailbox) * PG_BOXES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "hipHostMalloc(K1P mailbox) failed");
        std::memset(pg->box, 0, sizeof(PgMailbox) * PG_BOXES);
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&pg->box_dev), pg->box, 0) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "hipHostGetDevicePointer(K1P mailbox) failed");
    }
    return GM_OK;
}

int sort_pairs(gm_pigeon* pg, int64_t n, int bits, hipStream_t st) {
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, pg->k0, pg->k1, pg->r0, pg->r1, (size_t)n, 0u, (unsigned)bits, st));
    if (bytes > pg->sort_tmp_cap || pg->sort_tmp == nullptr) {
        if (pg->sort_tmp) (void)hipFree(pg->sort_tmp);
        pg->sort_tmp = nullptr;
        pg->sort_tmp_cap = 0;
        if (hipMalloc(&pg->sort_tmp, bytes ? bytes : 1) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "hipMalloc(radix sort workspace) failed");
        pg->sort_tmp_cap = bytes;
    }
    HIP_TRY(rocprim::radix_sort_pairs(pg->sort_tmp, bytes, pg->k0, pg->k1, pg->r0, pg->r1, (size_t)n, 0u, (unsigned)bits, st));
    return GM_OK;
}

int ensure_sort_buffers(gm_pigeon* pg, int64_t n) {
    int rc;
    if ((rc = ensure(&pg->k0, &pg->k0_cap, n, "sort keys"))) return rc;
    if ((rc = ensure(&pg->k1, &pg->k1_cap, n, "sort keys"))) return rc;
    if ((rc = ensure(&pg->r0, &pg->r0_cap, n, "sort rows"))) return rc;
    return ensure(&pg->r1, &pg->r1_cap, n, "sort rows");
}

void free_pool_side(gm_pigeon* pg) {
    pg->gsum_sig.clear();      // (the group sums belong to these copies)
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (pg->table[b]) (void)hipFree(pg->table[b]);
        if (pg->recs[b]) (void)hipFree(pg->recs[b]);
        if (pg->bdesc[b]) (void)hipFree(pg->bdesc[b]);
        if (pg->coff[b]) (void)hipFree(pg->coff[b]);
        if (pg->bcut[b]) (void)hipFree(pg->bcut[b]);
        pg->bcut[b] = nullptr;
        pg->table[b] = nullptr;
        pg->recs[b] = nullptr;
        pg->bdesc[b] = nullptr;
        pg->coff[b] = nullptr;
    }
    pg->B = 0;
    pg->cap_nq = -1;
    pg->bcut_for = 0xffffffffu;
}

void free_side(PgPoolSide& ps) {
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (ps.table[b]) (void)hipFree(ps.table[b]);
        if (ps.recs[b]) (void)hipFree(ps.recs[b]);
        if (ps.bdesc[b]) (void)hipFree(ps.bdesc[b]);
        if (ps.coff[b]) (void)hipFree(ps.coff[b]);
        if (ps.bcut[b]) (void)hipFree(ps.bcut[b]);
    }
    ps = PgPoolSide();
}

// the layout in use -> the stash (its device arrays change owner; pg is left without a pool side)
void stash_pool_side(gm_pigeon* pg) {
    if (pg->B < 2) {
        free_pool_side(pg);
        return;
    }
    PgPoolSide ps;
    ps.B = pg->B;
    ps.bcut_for = pg->bcut_for;
    ps.gap_mode = pg->tune_gap;
    ps.stamp = pg->use_stamp;
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        ps.blen[b] = pg->blen[b];
        ps.bstart[b] = pg->bstart[b];
        ps.skiplen[b] = pg->skiplen[b];
        ps.Lp[b] = pg->Lp[b];
        ps.max_bucket[b] = pg->max_bucket[b];
        ps.table[b] = pg->table[b];
        ps.recs[b] = pg->recs[b];
        ps.bdesc[b] = pg->bdesc[b];
        ps.coff[b] = pg->coff[b];
        ps.bcut[b] = pg->bcut[b];
        pg->table[b] = nullptr;
        pg->recs[b] = nullptr;
        pg->bdesc[b] = nullptr;
        pg->coff[b] = nullptr;
        pg->bcut[b] = nullptr;
    }
    for (int b = 0; b <= GM_PG_MAXB; ++b) ps.ebase[b] = pg->ebase[b];
    pg->B = 0;
    pg->cap_nq = -1;
    pg->bcut_for = 0xffffffffu;
    while ((int)pg->stash.size() >= PG_POOL_SIDES - 1) {      // drop the least recently used
        size_t old = 0;
        for (size_t i = 1; i < pg->stash.size(); ++i)
            if (pg->stash[i].stamp < pg->stash[old].stamp) old = i;
        free_side(pg->stash[old]);
        pg->gsum_sig.clear();      // (its addresses may come back with other contents)
        pg->stash.erase(pg->stash.begin() + (long)old);
    }
    pg->stash.push_back(ps);
}

// a stashed pool side with these block lengths (and table gaps) -> pg; false: none
bool unstash_pool_side(gm_pigeon* pg, const gm_pg_layout* lay) {
    for (size_t i = 0; i < pg->stash.size(); ++i) {
        const PgPoolSide& ps = pg->stash[i];
        bool same = ps.B == lay->B && ps.gap_mode == pg->tune_gap;
        for (int b = 0; b < lay->B && same; ++b) same = ps.blen[b] == lay->blen[b];
        if (!same) continue;
        pg->B = ps.B;
        pg->bcut_for = ps.bcut_for;
        for (int b = 0; b < GM_PG_MAXB; ++b) {
            pg->blen[b] = ps.blen[b];
            pg->bstart[b] = ps.bstart[b];
            pg->skiplen[b] = ps.skiplen[b];
            pg->Lp[b] = ps.Lp[b];
            pg->max_bucket[b] = ps.max_bucket[b];
            pg->table[b] = ps.table[b];
            pg->recs[b] = ps.recs[b];
            pg->bdesc[b] = ps.bdesc[b];
            pg->coff[b] = ps.coff[b];
            pg->bcut[b] = ps.bcut[b];
        }
        for (int b = 0; b <= GM_PG_MAXB; ++b) pg->ebase[b] = ps.ebase[b];
        pg->cap_nq = -1;
        pg->stash.erase(pg->stash.begin() + (long)i);
        return true;
    }
    return false;
}

double log_choose(int n, int r) { return std::lgamma(n + 1.0) - std::lgamma(r + 1.0) - std::lgamma(n - r + 1.0); }

}  // namespace

// Block lengths of a layout.  X = 0: the even split of L over B blocks (longer ones first).  X > 0: the X
// expansion blocks (blocks 0 .. X-1) have lenx bases each, the other B - X share the rest as evenly as
// possible and are not longer than lenx.  false: not a usable layout.
static bool pg_block_lengths(int L, int B, int X, int lenx, int max_len, int (&blen)[GM_PG_MAXB]) {
    if (B < 2 || B > GM_PG_MAXB || X < 0 || X >= B) return false;
    if (X == 0) {
        const int base = L / B, extra = L % B;
        for (int b = 0; b < B; ++b) blen[b] = base + (b < extra ? 1 : 0);
    } else {
        const int others = B - X, rest = L - X * lenx;
        if (rest < 2 * others) return false;
        const int base = rest / others, extra = rest % others;
        for (int b = 0; b < X; ++b) blen[b] = lenx;
        for (int b = X; b < B; ++b) blen[b] = base + (b - X < extra ? 1 : 0);
        if (blen[X] > lenx) return false;
    }
    for (int b = 0; b < B; ++b)
        if (blen[b] < 2 || blen[b] > max_len) return false;
    return true;
}

// P(at most r mismatches over n positions) when a position matches with probability pm
static double tail_within(int n, int r, double pm) {
    double cum = 0.0;
    for (int i = 0; i <= r && i <= n; ++i)
        cum += std::exp(log_choose(n, i) + i * std::log(1.0 - pm) + (n - i) * std::log(pm));
    return cum;
}

// Legacy layout forms (forced variants, tests): B blocks, the first X of them with radius 1 and lenx bases (see
// pg_block_lengths), the others radius 0; n2 > 0: block 0 gets radius 2 instead.
bool gm_pg_layout_make(int L, int B, int X, int lenx, int n2, gm_pg_layout* out) {
    int blen[GM_PG_MAXB];
    if (!pg_block_lengths(L, B, X, lenx, GM_PG_MAXLEN, blen)) return false;
    out->B = B;
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        out->blen[b] = b < B ? blen[b] : 0;
        out->trad[b] = b < X ? 1 : 0;
    }
    if (n2 > 0) out->trad[0] = 2;
    return true;
}

// experiments: layout from gm_set_param "pg_layout" (e.g. 826161 = blocks of 8, 6, 6 bases with radii 2, 1, 1)
bool gm_pg_layout_forced(const gm_index* idx, gm_pg_layout* out) {
    if (!idx->pg || idx->pg->tune_layout <= 0) return false;
    int digits[16], n = 0;
    for (int v = idx->pg->tune_layout; v > 0 && n < 16; v /= 10) digits[n++] = v % 10;
    if (n % 2 != 0 || n / 2 > GM_PG_MAXB) return false;
    out->B = n / 2;
    for (int b = 0; b < out->B; ++b) {
        out->blen[b] = digits[n - 1 - 2 * b];
        out->trad[b] = digits[n - 2 - 2 * b];
    }
    return true;
}

int gm_pg_layout_radius(const gm_pg_layout& lay) {
    int r = -1;
    for (int b = 0; b < lay.B; ++b) r += lay.trad[b] + 1;
    return r;
}

// Layout choice under an independent-base model with the pool's own composition (p_match = sum_b f_b^2).
// A layout = blocks with a search radius t_b each (gm_pg_types.h): n2 blocks with the two-substitution
// neighbourhood (len2 bases each), n1 with the one-substitution neighbourhood, n0 plain; it serves
// r = 3 n2 + 2 n1 + n0 - 1.  A pair within r that is not found under any block would have more than t_b mismatches
// in every block, i.e. at least r + 1 in total.
// cost = group evaluations per pair + P(Poisson(n P(d <= r)) < k) (rows re-scanned):
//   sum_b buckets_b x share_b x (1 + 128 / targets per bucket) x (1 + 32 / candidates per bucket)
// (share = p_match^len; a bucket's targets are padded to whole 64-target groups and every bucket visit costs about
// 1.5 group evaluations on top -- descriptor load, a DMA unit waited for cold; measured on S3 / S4, profiles/r02 --;
// a bucket's candidates are padded to whole 64-lane chunks; under a skewed composition a
// neighbour bucket holds rho = ((1 - p) / 3) / p times the targets of the own bucket per substitution).
// p_short (optional, [L + 1]): measured share of the candidates whose k-th neighbour lies beyond r (gm_pg_sample_rows +
// a plain scan of the sample, gmhip.hip) -- replaces the Poisson term, which knows the pool's composition but not the
// spread between candidates (on a GC-rich real genome an AT-rich candidate has a fraction of the average's neighbours)
// nor repeats.
// Cascade (deeper_levels > 0, p_short given): the rows a layout leaves short need not take the plain re-scan (cost 1.0 =
// the whole pool per row).  When they are many they form a candidate set of their own, n_q x p_short[r] rows whose k-th
// neighbour lies beyond r with the conditional shares p_short[r'] / p_short[r], and a deeper K1P level with a larger
// radius (and, the set being small, fewer and larger buckets) serves them at its own planned cost + the fixed price of
// one more level (grouping, merge, read-back: ~300 us, expressed in pool scans per row).  On a real genome the k-th
// neighbour's distance spreads over 3 .. 7 (P. aeruginosa PAO1, k = 10: 5 for half the guides, beyond 5 for 12.6 %),
// and the first level then runs the radius the MAJORITY needs instead of the radius nearly all need.
// Returns 1 and fills *out when a layout beats the plain scan clearly (cost < 0.45), else 0; *cost_out = its cost,
// *cascade_out = 1 when the cost assumes that a deeper level takes the short rows.
constexpr double PG_PAIR_RATE = 2.6e13;          // evaluated pairs per second of the look-up scans (S1 .. S4, profiles/r02)
constexpr double PG_LEVEL_OVERHEAD_US = 300.0;   // fixed price of one more K1P level (small grouping kernels, two read-backs, launch gaps)
constexpr double PG_CASCADE_MIN_ROWS = 4096.0;   // fewer short rows than this: the plain / prefix re-scan
// Hits.  Every target within the radius is a hit: queued, its record gathered, its distance recomputed, a key stored.  On
// P. aeruginosa PAO1 that is 57 % of the first level's scan (time-only ablation, profiles/r03/NOTES.md: 1.89 ms with, 0.82
// without hits for 72 hits per candidate at r = 5), i.e. a hit costs about as much as 300 .. 450 pair evaluations, and
// the hits per candidate grow 4 .. 5 x per unit of radius.  With a sample (gm_pg_sample: targets within d of every
// sampled candidate) the planner prices them: + hits(r) x PG_HIT_PAIRS / n_db pool scans per candidate.
constexpr double PG_HIT_PAIRS = 300.0;
constexpr double PG_PLAIN_TAIL = 1.5;
// Every level also pays per CANDIDATE, whatever its layout: the counting sort of the candidates by block value, the chunk
// table and the merge of the hit lists -- S1: 0.17 of 0.93 ms for 599 093 candidates, PAO1: 0.53 of 3.8 ms for 1.17 M:
// about 0.45 ns per candidate = 11 700 pair evaluations.  It does not move the choice between layouts of one level, so
// it is charged where it matters: in the price of a FURTHER cascade level (a first level that leaves 60 % of the rows
// short buys a second one for 60 % of them).
constexpr double PG_ROW_PAIRS = 11700.0;
int gm_pg_plan(int64_t n_db, int64_t n_q, int L, int k, double p_match, gm_pg_layout* out, const double* p_short,
               int deeper_levels, double* cost_out, int* cascade_out, const gm_pg_sample* smp) {
    if (cost_out) *cost_out = 1.0;
    if (cascade_out) *cascade_out = 0;
    if (L < 12 || L > GM_MAX_GUIDELEN || n_db < 32768 || k < 1 || k > 20) return 0;
    if (!(p_match >= 0.25 && p_match < 0.9)) p_match = 0.25;
    if (n_q < 1) n_q = n_db;
    // from the sample: the share of the (selected) rows whose k-th neighbour lies beyond r, and their mean hits within r
    double ps_smp[GM_MAX_GUIDELEN + 2], hits_smp[GM_MAX_GUIDELEN + 2];
    const double* hits = nullptr;
    if (smp != nullptr && smp->n > 0 && smp->cum != nullptr) {
        int64_t rows = 0;
        double beyond[GM_PG_SAMPLE_BINS + 1] = {0}, hsum[GM_PG_SAMPLE_BINS] = {0};
        for (int i = 0; i < smp->n; ++i) {
            const uint32_t* c = smp->cum + (size_t)i * GM_PG_SAMPLE_BINS;
            // (beyond the histogram: such a row counts as short at EVERY radius the histogram does not reach -- beyond[BINS] is
            // the share carried, as an upper bound, for r >= BINS; with kth = BINS it stayed 0 and a radius >= 9 was priced as
            // leaving no row short: ADVICE r03)
            int kth = GM_PG_SAMPLE_BINS + 1;
            for (int d = 0; d < GM_PG_SAMPLE_BINS; ++d)
                if (c[d] >= (uint32_t)smp->k) {
                    kth = d;
                    break;
                }
            if (kth <= smp->min_kth) continue;
            ++rows;
            for (int r = 0; r < kth && r <= GM_PG_SAMPLE_BINS; ++r) beyond[r] += 1.0;
            for (int d = 0; d < GM_PG_SAMPLE_BINS; ++d) hsum[d] += (double)(c[d] < 4096u ? c[d] : 4096u);
        }
        if (rows > 0) {
            for (int r = 0; r <= L + 1 && r <= GM_MAX_GUIDELEN + 1; ++r) {
                const int rr = r < GM_PG_SAMPLE_BINS ? r : GM_PG_SAMPLE_BINS;      // (beyond d = 8: the share seen there)
                ps_smp[r] = (beyond[rr] + 0.5) / ((double)rows + 1.0);
                hits_smp[r] = r < GM_PG_SAMPLE_BINS ? hsum[r] / (double)rows
                                                    : hits_smp[r - 1] * 4.0;      // (extrapolated)
            }
            p_short = ps_smp;
            hits = hits_smp;
        }
    }
    const double rho = ((1.0 - p_match) / 3.0) / p_match;
    double best = 0.45;      // K1P has to be clearly cheaper than scanning everything
    int found = 0;
    // price of a row left short at radius r (memoised: it depends on r only)
    double tail_of[GM_MAX_GUIDELEN + 2];
    int tail_casc[GM_MAX_GUIDELEN + 2];
    for (int r = 0; r <= GM_MAX_GUIDELEN + 1; ++r) {
        tail_of[r] = -1.0;
        tail_casc[r] = 0;
    }
    auto tail = [&](int r, double p_inc) {
        if (r > L) r = L;
        if (tail_of[r] >= 0.0) return tail_of[r];
        // the plain re-scan: the whole pool per row under a guessed bound, once more without it for the rows that still do not
        // fill, gather / merge / scatter around both -- measured ~1.5 pool scans per row (PAO1: 10 766 rows, 0.6 ms)
        const double n_sub = (double)n_q * p_inc;
        double t = (p_short != nullptr && n_sub >= 2.0 * PG_CASCADE_MIN_ROWS) ? PG_PLAIN_TAIL : 1.0;      // (a few thousand rows: the prefix re-scan, cheaper)
        if (deeper_levels > 0 && p_short != nullptr && p_inc > 0.0 && n_sub >= PG_CASCADE_MIN_ROWS) {
            double ps2[GM_MAX_GUIDELEN + 2];
            for (int rr = 0; rr <= L; ++rr) {
                const double c = rr <= r ? 1.0 : p_short[rr] / p_inc;
                ps2[rr] = c < 1.0 ? c : 1.0;
            }
            gm_pg_layout sub{};
            double c2 = 1.0;
            gm_pg_sample sub_smp{};
            if (hits != nullptr) {      // the deeper level sees the rows whose k-th neighbour lies beyond r
                sub_smp = *smp;
                sub_smp.min_kth = r;
            }
            if (gm_pg_plan(n_db, (int64_t)n_sub, L, k, p_match, &sub, ps2, deeper_levels - 1, &c2, nullptr,
                           hits != nullptr ? &sub_smp : nullptr)) {
                const double fixed = PG_LEVEL_OVERHEAD_US * 1e-6 * PG_PAIR_RATE / ((double)n_db * n_sub) + PG_ROW_PAIRS / (double)n_db;
                if (c2 + fixed < t) {
                    t = c2 + fixed;
                    tail_casc[r] = 1;
                }
            }
        }
        tail_of[r] = t;
        return t;
    };
    for (int n2 = 0; n2 <= 2; ++n2)      // (two radius-2 blocks pay off for long guides on pools of millions: L = 27, 4.2 M: 43 -> 32 ms)
        for (int n1 = 0; n1 <= 5; ++n1)      // (four radius-1 blocks: 262 k targets, L = 20, k = 10: 0.64 -> 0.46 ms per query)
            for (int n0 = 0; n2 + n1 + n0 <= GM_PG_MAXB; ++n0) {
                const int B = n2 + n1 + n0;
                if (B < 2) continue;
                const int r = 3 * n2 + 2 * n1 + n0 - 1;
                const double mean = (double)n_db * tail_within(L, r, p_match);
                double p_inc = 0.0, term = std::exp(-mean);      // P(Poisson(mean) < k)
                for (int j = 0; j < k; ++j) {
                    p_inc += term;
                    term *= mean / (j + 1);
                }
                if (mean > 700.0) p_inc = 0.0;
                if (p_short != nullptr) p_inc = p_short[r < L ? r : L];
                if (p_inc > 1.0) p_inc = 1.0;
                double p_tail = p_inc * tail(r, p_inc);
                if (hits != nullptr) p_tail += hits[r < L ? r : L] * PG_HIT_PAIRS / (double)n_db;
                if (p_tail >= best) continue;
                for (int len2 = n2 ? 4 : 0; len2 <= (n2 ? GM_PG_MAXLEN : 0); ++len2)
                    for (int len1 = (n1 && n0) ? 2 : 0; len1 <= ((n1 && n0) ? GM_PG_MAXLEN : 0); ++len1)
                    for (int e1 = 0; e1 < ((n1 && n0) ? n1 : 1); ++e1) {      // e1 of the radius-1 blocks one base longer
                        gm_pg_layout lay{};
                        lay.B = B;
                        int rest = L - n2 * len2, nb = 0;
                        for (int b = 0; b < n2; ++b, ++nb) {
                            lay.blen[nb] = len2;
                            lay.trad[nb] = 2;
                        }
                        bool ok = true;
                        if (n1 && n0) {      // radius-1 blocks of len1 (+ 1 for the first e1), the plain blocks share the rest evenly
                            rest -= n1 * len1 + e1;
                            for (int b = 0; b < n1; ++b, ++nb) {
                                lay.blen[nb] = len1 + (b < e1 ? 1 : 0);
                                lay.trad[nb] = 1;
                            }
                            for (int b = 0; b < n0; ++b, ++nb) {
                                lay.blen[nb] = rest / n0 + (b < rest % n0 ? 1 : 0);
                                lay.trad[nb] = 0;
                            }
                            ok = rest >= 2 * n0 && lay.blen[n2 + n1] <= len1;
                        } else {             // one class of blocks besides the radius-2 block: even split
                            const int m = n1 + n0;
                            if (m == 0) {
                                ok = rest == 0;
                            } else {
                                for (int b = 0; b < m; ++b, ++nb) {
                                    lay.blen[nb] = rest / m + (b < rest % m ? 1 : 0);
                                    lay.trad[nb] = n1 ? 1 : 0;
                                }
                                ok = rest >= 2 * m;
                            }
                        }
                        if (!ok) continue;
                        double work = 0.0;
                        for (int b = 0; b < B && ok; ++b) {
                            const int len = lay.blen[b];
                            if (len < 2 || len > GM_PG_MAXLEN) ok = false;
                            const double share = std::pow(p_match, (double)len);
                            const double tpb = share * (double)n_db, cpb = share * (double)n_q;
                            if (tpb < 8.0 || cpb < 8.0) ok = false;      // buckets far smaller than a group: never worth it
                            double passes = 1.0;
                            if (lay.trad[b] >= 1) passes += 3.0 * len * rho;
                            if (lay.trad[b] >= 2) passes += 9.0 * (len * (len - 1) / 2) * rho * rho;
                            work += share * passes * (1.0 + 128.0 / tpb) * (1.0 + 32.0 / cpb);
                        }
                        if (!ok) continue;
                        int lists = 0;
                        for (int b = 0; b < B; ++b) lists += (gm_pg_buckets_of(lay.blen[b], lay.trad[b]) + 7) / 8;
                        if (lists > GM_PG_MAXLISTS) continue;
                        const double cost = work + p_tail;
                        if (cost < best) {
                            best = cost;
                            *out = lay;
                            found = 1;
                            if (cascade_out) *cascade_out = tail_casc[r < L ? r : L];
                        }
                    }
            }
    if (cost_out) *cost_out = best;
    return found;
}

// measures the pool's base composition once per index (one small kernel + a 32-byte read-back)
int gm_pg_pool_stats(gm_index* idx, hipStream_t st, double* p_match) {
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    if (pg->p_match == 0.0) {
        unsigned long long* d_counts = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d_counts), 4 * sizeof(unsigned long long)) != hipSuccess)
            return gm_fail(GM_ERR_ALLOC, "hipMalloc(base counts) failed");
        unsigned long long h[4] = {0, 0, 0, 0};
        hipError_t e = hipMemsetAsync(d_counts, 0, sizeof(h), st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_base_counts, dim3(1024), dim3(256), 0, st, idx->d_db, idx->n_db, idx->L, d_counts);
            e = hipMemcpyAsync(h, d_counts, sizeof(h), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_counts);
        if (e != hipSuccess) return gm_fail(GM_ERR_DEVICE, std::string("pool composition: ") + hipGetErrorString(e));
        const double total = (double)(h[0] + h[1] + h[2] + h[3]);
        double pm = 0.0;
        for (int b = 0; b < 4; ++b) {
            pm += (h[b] / total) * (h[b] / total);
            pg->base_freq[b] = h[b] / total;
        }
        pg->p_match = pm;
    }
    *p_match = pg->p_match;
    return GM_OK;
}

// a deeper cascade level: the pool is the same (composition), the knobs follow level 0; its layout and copies are its own
void gm_pg_inherit(gm_index* idx, const gm_pigeon* parent) {
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    if (!parent) return;
    pg->p_match = parent->p_match;
    for (int b = 0; b < 4; ++b) pg->base_freq[b] = parent->base_freq[b];
    pg->tune_sub = parent->tune_sub;
    pg->deep = true;
    pg->tune_per = parent->tune_per;
    pg->tune_hit = parent->tune_hit;
    pg->tune_run = parent->tune_run;
    pg->tune_group = parent->tune_group;
    pg->tune_slots = parent->tune_slots;
    pg->tune_msplit = parent->tune_msplit;
    pg->tune_prefix = parent->tune_prefix;
    pg->tune_layout = idx->pg_layout_deeper;
}
const gm_pigeon* gm_pg_state(const gm_index* idx) { return idx->pg; }

int gm_pg_blocks_of(const gm_index* idx) { return idx->pg ? idx->pg->B : 0; }
int gm_pg_radius(const gm_index* idx) {
    if (!idx->pg) return 0;
    int r = -1;
    for (int b = 0; b < idx->pg->B; ++b) r += idx->pg->trad[b] + 1;
    return r;
}

int gm_pg_set_param(gm_index* idx, const char* name, int value) {
    if (std::strcmp(name, "pg_cascade") == 0) {
        if (value < 0 || value > 2) return gm_fail(GM_ERR_ARG, "pg_cascade: 0 (off), 1 (planner), 2 (whenever rows are left short)");
        idx->pg_cascade = value;
        idx->plan_nq = -1;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_layout2") == 0) {      // forced layout of the deeper cascade levels (tests)
        idx->pg_layout_deeper = value;
        for (auto& lv : idx->pg_deeper) lv.plan_nq = -1;
        return GM_OK;
    }
    if (!idx->pg) idx->pg = new gm_pigeon();
    if (std::strcmp(name, "pg_sub") == 0) {
        if (value < 0 || value > 19) return gm_fail(GM_ERR_ARG, "pg_sub: 0 (choose), 1..19");
        idx->pg->tune_sub = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_per") == 0) {
        if (value < 0 || value > 4096) return gm_fail(GM_ERR_ARG, "pg_per: 0 (from pg_sub), 1..4096");
        idx->pg->tune_per = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_hit") == 0) {
        if (value < 0 || value > 2) return gm_fail(GM_ERR_ARG, "pg_hit: 0 (hit buffer), 1 (chosen per level), 2 (register lists)");
        idx->pg->tune_hit = value;
        idx->pg->cap_nq = -1;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_layout") == 0) {
        idx->pg->tune_layout = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_gap") == 0) {      // (takes effect when the pool copies are next built)
        idx->pg->tune_gap = value;
        idx->pg->B = 0;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_debug") == 0)      // (timing experiments are separate builds: make CXXFLAGS+=-DGM_PG_TIMING=...)
        return gm_fail(GM_ERR_ARG, "pg_debug: not in this build (timing experiments are compile-time: -DGM_PG_TIMING)");
    if (std::strcmp(name, "pg_cut") == 0) {
        if (value < 0 || value > 100) return gm_fail(GM_ERR_ARG, "pg_cut: 0 (choose), 1..99 (percent of the ids), 100 (off)");
        idx->pg->tune_cut = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_prefix") == 0) {
        idx->pg->tune_prefix = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_sample") == 0) {      // (takes effect when the layout is next planned)
        idx->pg->tune_sample = value;
        idx->plan_nq = -1;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_msplit") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return gm_fail(GM_ERR_ARG, "pg_msplit: 0, 1, 2 or 4");
        idx->pg->tune_msplit = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_run") == 0) {
        if (value < 0 || value > 65536) return gm_fail(GM_ERR_ARG, "pg_run: 0 (choose), 1..65536");
        idx->pg->tune_run = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_slots") == 0) {      // the register-list merge's list of incomplete rows: 0 = choose, 1 = atomics, 2 = slots
        if (value < 0 || value > 2) return gm_fail(GM_ERR_ARG, "pg_slots: 0, 1 or 2");
        idx->pg->tune_slots = value;
        return GM_OK;
    }
    if (std::strcmp(name, "pg_group") == 0) {      // candidate grouping: 0 = choose, 1 = one counting sort, 2 = two levels
        if (value < 0 || value > 2) return gm_fail(GM_ERR_ARG, "pg_group: 0, 1 or 2");
        idx->pg->tune_group = value;
        return GM_OK;
    }
    return gm_fail(GM_ERR_ARG, std::string("unknown parameter ") + name);
}

void gm_pg_free(gm_index* idx) {
    gm_pigeon* pg = idx->pg;
    if (!pg) return;
    free_pool_side(pg);
    for (auto& ps : pg->stash) free_side(ps);
    pg->stash.clear();
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (pg->cperm[b]) (void)hipFree(pg->cperm[b]);
        if (pg->ctmp[b]) (void)hipFree(pg->ctmp[b]);
    }
    void* bufs[] = {pg->part_words, pg->gsum, pg->inc_slots, pg->inc_meta, pg->k0, pg->k1, pg->r0, pg->r1, pg->sort_tmp, pg->nch, pg->cbase, pg->hist, pg->cursor, pg->chunks, pg->ctr,
                    pg->lists, pg->hdr, pg->keys3, pg->raw_off, pg->inc, pg->inc_b, pg->cflag, pg->shist, pg->q2, pg->ids2, pg->dists2, pg->counts2};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (pg->box) (void)hipHostFree(pg->box);
    delete pg;
    idx->pg = nullptr;
}

// Block-sorted, bucket-padded pool copies for the layout (they depend on the block lengths only; the radii just
// select which buckets a query visits).
int gm_pg_prepare(gm_index* idx, const gm_pg_layout* lay, hipStream_t st) {
    const int L = idx->L, B = lay->B;
    int sum = 0;
    if (B < 2 || B > GM_PG_MAXB) return gm_fail(GM_ERR_ARG, "gm_pg_prepare: bad layout");
    for (int b = 0; b < B; ++b) {
        if (lay->blen[b] < 2 || lay->blen[b] > GM_PG_MAXLEN || lay->trad[b] < 0 || lay->trad[b] > 2)
            return gm_fail(GM_ERR_ARG, "gm_pg_prepare: bad layout");
        sum += lay->blen[b];
    }
    if (sum != L) return gm_fail(GM_ERR_ARG, "gm_pg_prepare: bad layout");
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    bool same = pg->B == B;
    for (int b = 0; b < B && same; ++b) same = pg->blen[b] == lay->blen[b];
    for (int b = 0; b < B; ++b) pg->trad[b] = lay->trad[b];
    ++pg->use_stamp;
    if (same) return GM_OK;
    stash_pool_side(pg);      // (the copies of the layout in use are kept for the next query that plans it again)
    if (unstash_pool_side(pg, lay)) {
        for (int b = 0; b < B; ++b) pg->trad[b] = lay->trad[b];
        return GM_OK;
    }
    for (int b = 0; b < B; ++b) pg->trad[b] = lay->trad[b];
    const int* blen = lay->blen;
    const int64_t n = idx->n_db;
    int pos = 0, base = blen[0];
    pg->ebase[0] = 0;
    for (int b = 0; b < B; ++b) {
        pg->bstart[b] = pos;
        pg->blen[b] = blen[b];
        base = blen[b] < base ? blen[b] : base;      // shortest block
        pos += pg->blen[b];
        pg->ebase[b + 1] = pg->ebase[b] + (1 << (2 * pg->blen[b]));
    }
    int rc;
    if ((rc = ensure_sort_buffers(pg, n))) return rc;
    if ((rc = pg_ensure_counters(pg))) return rc;
    // every pair of a bucket agrees on the copy's own block (a neighbour bucket differs there on known positions):
    // each copy's table leaves its own block out and the scan of that copy's chunks runs the kernel instantiated for
    // the remaining positions (at least 10: the look-up kernels start there)
    // Option A: every table leaves out as many positions as the SHORTEST block has: one launch for all blocks.
    // Option B: block b leaves out min(its length, G) positions, G = the length of the block with the most bucket
    // visits: fewer positions (and a cheaper adder tree: 16 match words cost 35 ops per half, 15 cost 27) for most
    // pairs, at the price of one launch per run of blocks with equal positions.  B is taken when the modelled VALU
    // work drops by more than a tenth.
    {
        auto tree_ops = [](int n) {      // v_bitop3 per half group: carry-save count of n one-bit words + 5-bit compare
            int ops = 0;
            while (n > 0) {
                int words = n, carries = 0;
                while (words > 1) {
                    if (words == 2) {
                        ops += 2;
                        carries += 1;
                        words = 1;
                    } else {
                        const int t = words / 3;
                        ops += 2 * t;
                        carries += t;
                        words = t + words % 3;
                    }
                }
                n = carries;
            }
            return ops + 5;
        };
        double visits[GM_PG_MAXB], tot = 0.0;
        int G = blen[0];
        double best_v = -1.0;
        for (int b = 0; b < B; ++b) {
            visits[b] = (double)gm_pg_buckets_of(blen[b], lay->trad[b]) * std::pow(0.25, (double)blen[b]);
            tot += visits[b];
            if (visits[b] > best_v) {
                best_v = visits[b];
                G = blen[b];
            }
        }
        double cost_a = 0.0, cost_b = 0.0;
        for (int b = 0; b < B; ++b) {
            int sa = base, sb = blen[b] < G ? blen[b] : G;
            if (L - sa < 10) sa = L - 10;
            if (L - sb < 10) sb = L - 10;
            cost_a += visits[b] / tot * tree_ops(L - sa);
            cost_b += visits[b] / tot * tree_ops(L - sb);
        }
        // Measured (profiles/r02/NOTES.md): on S1 / S3 / S4 option B loses -- (6,5,5,4): 0.65 -> 0.73 ms although the
        // model says -20 % VALU work; two launches drain and refill the chip and the small one runs badly -- so the
        // common gap is the default and B stays a knob (gm_set_param "pg_gap" = 2).
        (void)cost_a;
        (void)cost_b;
        const bool per_block = idx->pg->tune_gap == 2;
        for (int b = 0; b < B; ++b) {
            pg->skiplen[b] = per_block ? (blen[b] < G ? blen[b] : G) : base;
            if (L - pg->skiplen[b] < 10) pg->skiplen[b] = L - 10;
            pg->Lp[b] = L - pg->skiplen[b];
        }
    }
    constexpr int GPA = GM_PG_ALIGN / gm::LK_GROUP;      // groups per alignment unit
    for (int b = 0; b < B && rc == GM_OK; ++b) {
        const int nb = 1 << (2 * pg->blen[b]);
        if ((rc = ensure(&pg->raw_off, &pg->raw_off_cap, nb + 1, "bucket offsets"))) break;
        if (hipMalloc(reinterpret_cast<void**>(&pg->bdesc[b]), sizeof(int2) * (size_t)(nb + 1)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&pg->coff[b]), sizeof(int32_t) * (size_t)(nb + 1)) != hipSuccess) {
            rc = gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P bucket tables) failed");
            break;
        }
        hipLaunchKernelGGL(k_block_keys, grid_for(n), dim3(256), 0, st, idx->d_db, n, pg->bstart[b], pg->blen[b], pg->k0,
                           pg->r0);
        if ((rc = sort_pairs(pg, n, 2 * pg->blen[b], st))) break;
        hipLaunchKernelGGL(k_bucket_offsets, grid_for(nb + 1), dim3(256), 0, st, pg->k1, n, nb, pg->raw_off);
        // bucket geometry on the host: every bucket starts on a multiple of GM_PG_ALIGN positions
        std::vector<int32_t> off((size_t)nb + 1);
        std::vector<int2> bdesc((size_t)nb + 1);
        if (hipMemcpyAsync(off.data(), pg->raw_off, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            rc = gm_fail(GM_ERR_DEVICE, "K1P bucket offsets read-back failed");
            break;
        }
        int64_t g = 0;
        int mx = 0;
        for (int v = 0; v < nb; ++v) {
            const int cnt = off[v + 1] - off[v];
            mx = cnt > mx ? cnt : mx;
            bdesc[v] = make_int2((int)g, (cnt + gm::LK_GROUP - 1) / gm::LK_GROUP);
            g += (int64_t)((cnt + GM_PG_ALIGN - 1) / GM_PG_ALIGN) * GPA;
        }
        bdesc[nb] = make_int2((int)g, 0);
        pg->max_bucket[b] = mx;
        const int64_t n_groups = g > 0 ? g : GPA;
        if (n_groups * gm::LK_GROUP >= ((int64_t)1 << 31)) {
            rc = gm_fail(GM_ERR_ARG, "K1P: padded pool copy too large");
            break;
        }
        const int64_t words = n_groups * gm::lk_group_words(pg->Lp[b]);
        if (hipMalloc(reinterpret_cast<void**>(&pg->recs[b]), sizeof(uint4) * (size_t)(n_groups * gm::LK_GROUP)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&pg->table[b]), sizeof(uint32_t) * (size_t)words) != hipSuccess) {
            rc = gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P pool copy) failed");
            break;
        }
        hipError_t e = hipMemcpyAsync(pg->bdesc[b], bdesc.data(), sizeof(int2) * bdesc.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(pg->recs[b], 0xff, sizeof(uint4) * (size_t)(n_groups * gm::LK_GROUP), st);
        if (e != hipSuccess) {
            rc = gm_fail(GM_ERR_DEVICE, std::string("K1P pool copy: ") + hipGetErrorString(e));
            break;
        }
        hipLaunchKernelGGL(k_scatter_recs, grid_for(n), dim3(256), 0, st, idx->d_db, pg->k1, pg->r1, n, pg->raw_off, pg->bdesc[b],
                           pg->recs[b]);
        hipLaunchKernelGGL(k_onehot_recs, grid_for(words), dim3(256), 0, st, pg->recs[b], pg->table[b], n_groups, pg->Lp[b],
                           pg->bstart[b], pg->skiplen[b]);
        if (hipGetLastError() != hipSuccess) rc = gm_fail(GM_ERR_DEVICE, "K1P pool-copy kernels failed to launch");
        // (the bdesc host vector is pageable: the copies above were staged by the runtime before returning)
        if (hipStreamSynchronize(st) != hipSuccess && rc == GM_OK) rc = gm_fail(GM_ERR_DEVICE, "K1P pool-copy build failed");
    }
    if (rc != GM_OK) {
        free_pool_side(pg);
        return rc;
    }
    pg->B = B;
    return GM_OK;
}

// per block: target buckets of a candidate bucket, the workgroups they are split over (about 16 / pg_sub = 8 buckets
// per workgroup, at most 6 workgroups per chunk -- every workgroup means one more hit list per candidate), and the
// hit-list numbering
// The hit path of a level.  Register lists (k_hamming_scan_pg4): exact bound, no hit buffer, a merge that reads 40 dense
// bytes per list -- at 6 instead of 8 waves per SIMD (10 more VGPRs) and with a serial sorted insert per delivered key.
// Measured (profiles/r04/NOTES.md): faster where hits are dense and chunks plentiful (PAO1 level 0: scan 1.82 -> 1.46 ms,
// merge 0.40 -> 0.10; S3 level 0: 0.58 -> 0.46), a little faster on S1 and -- through the merge -- on S4's 20-key lists,
// slower on the deeper levels' few short-handed chunks (PAO1 levels 1 + 2: 0.92 -> 1.4 ms).
static bool pg_use_reglists(const gm_pigeon* pg, int64_t n_q, int k) {
    if (pg->tune_hit != 1) return pg->tune_hit == 2;
    return k <= 20 && n_q >= 400000;
}

static void pg_geometry(gm_pigeon* pg) {
    // (register lists: every chunk's target buckets in ONE workgroup -- a list sees all the block's hits of its candidates,
    // its bound tightens three times sooner, and the merge has a third of the lists to fold)
    const int sub = pg->tune_sub > 0 ? pg->tune_sub : (pg->deep ? 8 : 2);
    const int per_target = pg->tune_per > 0 ? pg->tune_per : (pg->hit_reg ? 4096 : (16 + sub - 1) / sub);
    int cap_sub = 6;
    for (;;) {
        int lists = 0;
        pg->smax = 1;
        for (int b = 0; b < pg->B; ++b) {
            const int n_bk = gm_pg_buckets_of(pg->blen[b], pg->trad[b]);
            int n_sub = (n_bk + per_target - 1) / per_target;
            if (n_sub > cap_sub) n_sub = cap_sub;
            const int per = (n_bk + n_sub - 1) / n_sub;
            n_sub = (n_bk + per - 1) / per;      // every workgroup of a chunk has at least one target bucket
            pg->nbk[b] = n_bk;
            pg->per[b] = per;
            pg->nsub[b] = n_sub;
            pg->lbase[b] = lists;
            lists += n_sub;
            if (n_sub > pg->smax) pg->smax = n_sub;
        }
        pg->blocks.n_lists = lists;
        if (lists <= GM_PG_MAXLISTS || cap_sub == 1) break;
        --cap_sub;
    }
}

// group the candidates in idx->d_q by every block's value and build the chunk table
int gm_pg_candidates(gm_index* idx, int64_t n_q, int k, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    if (!pg || pg->B < 2) return gm_fail(GM_ERR_ARG, "gm_pg_candidates: no layout prepared");
    pg->hit_reg = pg_use_reglists(pg, n_q, k);
    if (n_q >= (int64_t)1 << 31) return gm_fail(GM_ERR_ARG, "gm_pg_candidates: too many candidates");
    int rc;
    const int n_entries = pg->ebase[pg->B];      // (block, bucket) entries = histogram bins
    {
        const int64_t cap_before = pg->hist_cap;
        const int* ptr_before = pg->hist;
        if ((rc = ensure(&pg->hist, &pg->hist_cap, n_entries, "bucket histogram"))) return rc;
        if (pg->hist_cap != cap_before || pg->hist != ptr_before) pg->hist_dirty = true;
    }
    if ((rc = ensure(&pg->cursor, &pg->cursor_cap, n_entries, "bucket cursors"))) return rc;
    PgLayoutArg A{};
    A.B = pg->B;
    PgKeyArg K{};
    K.B = pg->B;
    pg_geometry(pg);
    int64_t grid = 0;
    for (int b = 0; b <= pg->B; ++b) A.ebase[b] = K.ebase[b] = pg->ebase[b];
    // two-level grouping for large candidate sets (gm_group.hip.h): partitions = the high key bits, at least 64 per block
    // where the block has that many values; the low <= 7 bits are sorted inside a partition
    bool two_level = (pg->tune_group == 2 || (pg->tune_group == 0 && n_q >= 262144)) && n_q < ((int64_t)1 << (32 - PG_PART_LO_MAX));
    if (two_level) {
        int n_parts = 0;
        for (int b = 0; b < pg->B; ++b) {
            const int bits = 2 * pg->blen[b];
            // (low bits: at most 5 -- more partitions make the first pass's runs shorter and the second pass's workgroups
            // smaller; S4 / PAO1, first + second pass: 146 / 82 us with 7 bits, 112 / 81 with 5, 142 / 96 with 4)
            int lo = bits - 6;
            lo = lo < 0 ? 0 : (lo > PG_PART_LO ? PG_PART_LO : lo);
            K.lo_bits[b] = lo;
            K.part_base[b] = n_parts;
            n_parts += 1 << (bits - lo);
        }
        K.part_base[pg->B] = n_parts;
        if (n_parts > PG_PART_MAX) two_level = false;      // (three blocks of 8 bases: the one-level sort)
    }
    if (two_level) {
        const int n_parts = K.part_base[pg->B];
        for (int b = 0; b < pg->B; ++b) {
            if ((rc = ensure(&pg->ctmp[b], &pg->ctmp_cap[b], n_q, "candidate order (first pass)"))) return rc;
            K.ctmp[b] = pg->ctmp[b];
        }
        if (pg->part_words_cap < 2 * (int64_t)n_parts) {      // (zero on entry, left zero by the offsets kernel)
            if ((rc = ensure(&pg->part_words, &pg->part_words_cap, 2 * (int64_t)n_parts, "partition counters"))) return rc;
            HIP_TRY(hipMemsetAsync(pg->part_words, 0, sizeof(int) * (size_t)pg->part_words_cap, st));
        }
    }
    for (int b = 0; b < pg->B; ++b) {
        if ((rc = ensure(&pg->cperm[b], &pg->cperm_cap[b], n_q, "candidate order"))) return rc;
        K.bstart[b] = pg->bstart[b];
        K.blen[b] = pg->blen[b];
        K.coff[b] = pg->coff[b];
        K.cperm[b] = pg->cperm[b];
        A.coff[b] = pg->coff[b];
        A.bdesc[b] = pg->bdesc[b];
        A.blen[b] = pg->blen[b];
        A.nbk[b] = pg->nbk[b];
        A.nsub[b] = pg->nsub[b];
        A.per[b] = pg->per[b];
        A.lbase[b] = pg->lbase[b];
        A.skiplen[b] = pg->skiplen[b];
        pg->bgrid[b] = ((n_q + 63) / 64 + (1 << (2 * pg->blen[b]))) * pg->nsub[b];
        grid += pg->bgrid[b];
    }
    // counting sort of the candidates by every block's value: histogram, scan (+ chunk-table geometry), scatter.  The
    // histogram / scatter kernels privatise the (block, bucket) counters in LDS: one launch covers a run of blocks whose
    // buckets fit 64 KiB (S1: all four blocks, 6 400 buckets; S4's 7 / 7 / 6 bases: 16 384 buckets per launch); a block of
    // 8 bases (65 536 buckets) goes through device-wide atomics.
    A.smax = pg->smax;
    A.n_entries = n_entries;
    if ((rc = ensure(&pg->cbase, &pg->cbase_cap, A.n_entries, "chunk offsets"))) return rc;
    if ((rc = ensure(&pg->chunks, &pg->chunks_cap, grid, "chunk table"))) return rc;
    if (pg->nch_cap < GM_PG_MAXB + 1) {      // (+ the offsets kernel's ticket: zero on entry, left zero)
        if ((rc = ensure(&pg->nch, &pg->nch_cap, GM_PG_MAXB + 1, "chunks per block"))) return rc;
        HIP_TRY(hipMemsetAsync(pg->nch, 0, sizeof(int) * (GM_PG_MAXB + 1), st));
    }
    if (pg->hist_dirty) {      // (the offsets kernel leaves the histogram zeroed: only a fresh buffer needs this)
        HIP_TRY(hipMemsetAsync(pg->hist, 0, sizeof(int) * (size_t)pg->hist_cap, st));
        pg->hist_dirty = false;
    }
    PgSubArg nsub{};
    for (int b = 0; b < pg->B; ++b) nsub.n[b] = pg->nsub[b];
    if ((rc = gm_group_launch(idx->d_q, n_q, K, nsub, pg->hist, pg->cursor, pg->cbase, pg->nch, reinterpret_cast<int*>(pg->ctr),
                              (int)(sizeof(PgCounters) / sizeof(int)), pg->ctr->cstart, &pg->ctr->n_chunks, idx->device, st, idx->dev_nq,
                              two_level ? pg->part_words : nullptr)))
        return rc;
    {      // the group sums of this geometry over these pool copies: once, not per query
        std::vector<int64_t> sig{(int64_t)A.B, (int64_t)A.smax, (int64_t)A.n_entries};
        for (int b = 0; b < pg->B; ++b) {
            const int64_t w[] = {(int64_t)reinterpret_cast<uintptr_t>(A.bdesc[b]), pg->bstart[b], A.blen[b], A.nbk[b], A.nsub[b], A.per[b], A.skiplen[b]};
            sig.insert(sig.end(), w, w + 7);
        }
        if (sig != pg->gsum_sig || pg->gsum == nullptr) {
            if ((rc = ensure(&pg->gsum, &pg->gsum_cap, (int64_t)A.n_entries * A.smax, "group sums"))) return rc;
            hipLaunchKernelGGL(k_group_sums, grid_for((int64_t)A.n_entries * A.smax), dim3(256), 0, st, A, pg->gsum);
            HIP_TRY(hipGetLastError());
            pg->gsum_sig = sig;
        }
    }
    hipLaunchKernelGGL(k_fill_chunks, grid_for((int64_t)A.n_entries * A.smax), dim3(256), 0, st, A, pg->cbase, pg->gsum, pg->chunks, pg->ctr);
    HIP_TRY(hipGetLastError());
    pg->grid = grid;
    return GM_OK;
}

// Key capacity of every hit list: mean + 4 sigma (+ 2) of the hits expected from the
// list's target buckets under the independent-base model with the pool's own composition, taking every bucket as
// large as the LARGEST bucket of its copy; a multiple of 4 (the merge reads 16 bytes at a time), at least 4.  A
// list that still overflows (repeat families) only makes its row incomplete: the row is re-scanned by the plain
// kernel.  All lists together stay within a quarter of the free device memory (at least 2 GiB).
static void pg_list_capacity(const gm_index* idx, gm_pigeon* pg, int64_t n_q) {
    const int L = idx->L, r = gm_pg_radius(idx);
    const double pm = pg->p_match >= 0.25 ? pg->p_match : 0.25;
    pg_geometry(pg);
    gm_pg_blocks& P = pg->blocks;
    double mean[GM_PG_MAXLISTS];
    for (int b = 0; b < pg->B; ++b) {
        const int len = pg->blen[b];
        // expected hits from an own / one-substitution / two-substitution bucket (the pair already has 0 / 1 / 2
        // mismatches on the block)
        double h[3];
        // (id cut-off: of the targets at distance exactly r only the fraction cut_f is taken)
        for (int j = 0; j < 3; ++j)
            h[j] = (double)pg->max_bucket[b] * (tail_within(L - len, r - j - 1, pm) +
                                                pg->cut_f * (tail_within(L - len, r - j, pm) - tail_within(L - len, r - j - 1, pm)));
        for (int s = 0; s < pg->nsub[b]; ++s) {
            const int e_lo = s * pg->per[b], e_hi = e_lo + pg->per[b] < pg->nbk[b] ? e_lo + pg->per[b] : pg->nbk[b];
            double m = 0.0;
            for (int e = e_lo; e < e_hi; ++e) m += e == 0 ? h[0] + 1.0 : (e <= 3 * len ? h[1] : h[2]);
            mean[pg->lbase[b] + s] = m;
        }
    }
    if (pg->mem_limit == 0.0) {      // (cached: hipMemGetInfo is not free)
        size_t free_b = 0, total_b = 0;
        pg->mem_limit = 2.0 * 1024 * 1024 * 1024;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b / 4.0 > pg->mem_limit)
            pg->mem_limit = (double)free_b / 4.0;
    }
    double scale = 1.0;
    P.hdr_words = (P.n_lists + 3) / 4 * 4;      // (the lists' header words live in their own dense array)
    for (;;) {
        int64_t words = 0;
        for (int l = 0; l < P.n_lists; ++l) {
            double c = std::ceil(scale * (mean[l] + 4.0 * std::sqrt(mean[l]) + 2.0));
            if (c > 4096.0) c = 4096.0;
            int cap = ((int)c + 3) / 4 * 4;
            if (cap < 4) cap = 4;
            P.lcap[l] = cap;
            words += cap;
        }
        // row = [keys 0 .. 3 of every list][keys 4 .. of list 0][of list 1] ...: loff[l] = where list l's keys 4 .. start
        {
            int64_t at = 4 * (int64_t)P.n_lists;
            for (int l = 0; l < P.n_lists; ++l) {
                P.loff[l] = (int)at;
                at += P.lcap[l] - 4;
            }
        }
        P.row_words = (int)words;
        if ((double)words * (double)n_q * 4.0 <= pg->mem_limit || scale < 0.05) break;
        scale *= 0.75;
    }
}

int64_t gm_pg_list_words(const gm_pigeon* pg) { return pg->blocks.row_words; }

int gm_pg_scan(gm_index* idx, int64_t n_q, int k, int kcap, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    uint64_t lkey = 0;      // the radii of the layout in use (the block lengths are fixed while the copies live)
    for (int b = 0; b < pg->B; ++b) lkey = lkey * 4 + (uint64_t)pg->trad[b] + 1;
    // Id cut-off.  lam_r targets sit at distance exactly r of a candidate on average, lam_lt closer; the k best need
    // about k - lam_lt of the former, and ties go to the smaller id: take distance-r hits only from ids below f n_db with
    // f lam_r = (k - lam_lt)+ + 4 sqrt(k) + 3 (a candidate for which that is not enough ends incomplete and is re-scanned).
    // Not for k = 1 (the bound follows the best distance there) and not when it would keep most of them anyway.
    {
        const double pm = pg->p_match >= 0.25 ? pg->p_match : 0.25;
        const int r = gm_pg_radius(idx);
        const double lam_lt = (double)idx->n_db * tail_within(idx->L, r - 1, pm) + 1.0;
        const double lam_r = (double)idx->n_db * (tail_within(idx->L, r, pm) - tail_within(idx->L, r - 1, pm));
        double f = 1.0;
        if (k > 1 && lam_r > 0.0) {
            const double need = (double)k - lam_lt > 0.0 ? (double)k - lam_lt : 0.0;
            f = (need + 4.0 * std::sqrt((double)k) + 3.0) / lam_r;
        }
        // The estimate is the pool AVERAGE.  With a skewed base composition the candidates differ by orders of magnitude
        // in how many neighbours they have (S3, gc 0.38: f = 0.44 for everybody left 3.5 % of the candidates short instead
        // of 0.13 %), so every candidate decides for itself from its own bases (k_cand_cutflag): the sparse ones keep
        // all their hits at r.  A handle on which the cut-off still leaves too many rows short drops it.
        if (pg->cut_dropped) f = 1.0;
        if (pg->tune_cut >= 100) f = 1.0;
        else if (pg->tune_cut > 0 && k > 1) f = pg->tune_cut / 100.0;
        if (f > 0.75 || f <= 0.0) f = 1.0;
        const uint32_t cut_id = f < 1.0 ? (uint32_t)std::ceil(f * (double)idx->n_db) : 0xffffffffu;
        if (f != pg->cut_f) pg->cap_nq = -1;      // the list capacities depend on it
        pg->cut_f = f;
        pg->cut_id = cut_id;
        if (cut_id != 0xffffffffu && pg->bcut_for != cut_id) {
            for (int b = 0; b < pg->B; ++b) {
                const int nb = 1 << (2 * pg->blen[b]);
                if (!pg->bcut[b] && hipMalloc(reinterpret_cast<void**>(&pg->bcut[b]), sizeof(int) * (size_t)nb) != hipSuccess)
                    return gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P cut-off table) failed");
                hipLaunchKernelGGL(k_bucket_cut, grid_for(nb), dim3(256), 0, st, pg->recs[b], pg->bdesc[b], nb, cut_id, pg->bcut[b]);
            }
            HIP_TRY(hipGetLastError());
            pg->bcut_for = cut_id;
        }
        if (cut_id != 0xffffffffu) {
            if ((rc = ensure(&pg->cflag, &pg->cflag_cap, n_q, "cut-off flags"))) return rc;
            PgFreqArg F{};
            for (int b = 0; b < 4; ++b) F.miss[b] = (float)(1.0 - pg->base_freq[b]);
            hipLaunchKernelGGL(k_cand_cutflag, grid_for(n_q), dim3(256), 0, st, idx->d_q, n_q, idx->L, F, (float)idx->n_db, r, k,
                               (float)f, pg->cflag);
            HIP_TRY(hipGetLastError());
        }
    }
    const int sub_key = pg->tune_sub * 8192 + pg->tune_per;
    if (pg->hit_reg != pg_use_reglists(pg, n_q, k)) return gm_fail(GM_ERR_ARG, "gm_pg_scan: candidates were grouped for another k");
    const int kcap3 = kcap;      // (the merge reads the raw lists with the same stride)
    if (pg->hit_reg) {
        pg_geometry(pg);
        if ((rc = ensure(&pg->keys3, &pg->keys3_cap, (int64_t)pg->blocks.n_lists * n_q * kcap3, "K1P raw lists"))) return rc;
#ifdef GM_PG_STATS      // instrumented build (tools only): eight 64-bit event counters per launch, printed to stderr
        if ((rc = ensure(&pg->hdr, &pg->hdr_cap, 64, "K1P stats"))) return rc;
        pg->blocks.hdr = pg->hdr;
        HIP_TRY(hipMemsetAsync(pg->hdr, 0, 64 * sizeof(uint32_t), st));
#endif
    } else {
        if (pg->cap_nq != n_q || pg->cap_layout_for != lkey || pg->cap_sub_for != sub_key) {
            pg_list_capacity(idx, pg, n_q);
            pg->cap_nq = n_q;
            pg->cap_layout_for = lkey;
            pg->cap_sub_for = sub_key;
        }
        if ((rc = ensure(&pg->lists, &pg->lists_cap, gm_pg_list_words(pg) * n_q, "K1P hit lists"))) return rc;
        if ((rc = ensure(&pg->hdr, &pg->hdr_cap, (int64_t)pg->blocks.hdr_words * n_q, "K1P list headers"))) return rc;
        pg->blocks.hdr = pg->hdr;
    }
    gm_pg_blocks& blocks = pg->blocks;
    for (int b = 0; b < pg->B; ++b) {
        blocks.blen[b] = pg->blen[b];
        blocks.trad[b] = pg->trad[b];
        blocks.table[b] = pg->table[b];
        blocks.recs[b] = pg->recs[b];
        blocks.bdesc[b] = pg->bdesc[b];
        blocks.cperm[b] = pg->cperm[b];
        blocks.bcut[b] = pg->bcut[b];
        blocks.skip[b] = pg->bstart[b];
        blocks.skiplen[b] = pg->skiplen[b];
    }
    blocks.n_blocks = pg->B;
    blocks.cut_id = pg->cut_id;
    blocks.cflag = pg->cflag;
    const int thr0 = gm_pg_radius(idx) + 1;      // accept d < thr0, i.e. d <= r
    blocks.run = pg->tune_run > 0 ? pg->tune_run : (pg->hit_reg ? 32 : 128);
    const int64_t unit = 8 * (int64_t)blocks.run;      // the kernel's entry mapping permutes whole units
    // one launch per run of consecutive blocks whose copies keep the same number of positions
    for (int b0 = 0; b0 < pg->B;) {
        int b1 = b0 + 1;
        int64_t g = pg->bgrid[b0];
        while (b1 < pg->B && pg->Lp[b1] == pg->Lp[b0]) g += pg->bgrid[b1++];
        const int64_t grid = (g + unit - 1) / unit * unit;
        const int* c_begin = &pg->ctr->cstart[b0];
        const int* c_end = &pg->ctr->cstart[b1];
        int rc2 = GM_OK;
        switch (pg->Lp[b0]) {            // the kernel instantiated for the positions actually evaluated
#define GM_CASE(LEN)                                                                                                     \
    case LEN:                                                                                                            \
        rc2 = pg->hit_reg ? gm_launch_pg3_L##LEN(idx, &blocks, pg->chunks, c_begin, c_end, grid, n_q, k, kcap3, thr0, pg->keys3, st) \
                           : gm_launch_pg_L##LEN(idx, &blocks, pg->chunks, c_begin, c_end, grid, n_q, k, thr0, pg->lists, st);       \
        break;
            GM_FOR_EACH_LEN(GM_CASE)
#undef GM_CASE
            default: rc2 = gm_fail(GM_ERR_ARG, "pigeonhole kernel: guidelen out of range");
        }
        if (rc2) return rc2;
        b0 = b1;
    }
#ifdef GM_PG_STATS
    if (pg->hit_reg) {
        unsigned long long h[8];
        HIP_TRY(hipMemcpyAsync(h, pg->hdr, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        fprintf(stderr, "[gm stats] n_q %lld: workgroups %llu, rounds %llu (%llu at the end), hits %llu, delivered %llu, sub-rounds %llu, inserts %llu\n",
                (long long)n_q, h[5], h[0], h[4], h[1], h[2], h[3], h[6]);
    }
#endif
    return GM_OK;
}

// hit lists -> top k; rows that are not exact are listed in pg->inc (count in pg->ctr->n_inc)
int gm_pg_merge(gm_index* idx, int kcap, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                bool nearest, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->inc, &pg->inc_cap, n_q, "incomplete rows"))) return rc;
    const int nr = nearest ? 1 : 0;
    int split = kcap >= 20 ? 2 : 4;      // (S4, k = 20: 15.3 / 15.0 / 15.8 ms per query with 1 / 2 / 4; S1, k = 10: 1.23 / 1.15 / 1.13)
    if (pg->tune_msplit > 0) split = pg->tune_msplit;
    const dim3 grid = pg->hit_reg ? grid_for(n_q) : grid_for(n_q * split), block(256);
    int* n_inc = &pg->ctr->n_inc;
    // Where many rows are expected short (about one per workgroup or more: S1 0.9 %, a cascade's first level 12 .. 30 %) the
    // register-list merge lists them WITHOUT device-wide atomics: a slot per workgroup, packed by k_compact_inc (S3's first
    // level 63 -> 21 + 12 us, PAO1's 89 -> 45 + 12, S1's 45 -> 31 + 9).  Where hardly any are (S4: 249 of 3.1 M) and on the
    // hit-buffer merge (its time is its key reads: 49 -> 41 + 10) the atomics stay.
    const bool slots = pg->hit_reg && (pg->tune_slots == 2 || (pg->tune_slots == 0 && n_q >= 65536 && pg->expect_short * 256.0 >= 0.5));
    const int64_t n_wg = grid.x;
    if (slots) {
        if ((rc = ensure(&pg->inc_slots, &pg->inc_slots_cap, n_wg * gm::PG_STASH_ROWS, "incomplete rows (workgroup slots)"))) return rc;
        if ((rc = ensure(&pg->inc_meta, &pg->inc_meta_cap, n_wg * (gm::PG_STASH_META + 1), "incomplete rows (workgroup counters)"))) return rc;
    }
#define GM_MERGE(KC, SP)                                                                                                          \
    do {                                                                                                                          \
        if (pg->hit_reg)                                                                                                          \
            hipLaunchKernelGGL((gm::k_merge_pg3<KC, 1>), grid, block, 0, st, pg->keys3, pg->blocks.n_lists, n_q, k, ids, dists,    \
                               counts, nr, pg->inc, n_inc, &pg->ctr->n_over, pg->ctr->chist, slots ? pg->inc_slots : nullptr,     \
                               slots ? pg->inc_meta : nullptr, pg->cut_id != 0xffffffffu ? pg->cflag : nullptr, idx->dev_nq);     \
        else                                                                                                                      \
            hipLaunchKernelGGL((gm::k_merge_hits<KC, SP>), grid, block, 0, st, pg->blocks, pg->lists, n_q, k, ids, dists, counts,  \
                               nr, pg->inc, n_inc, &pg->ctr->n_over, pg->ctr->chist,                                              \
                               pg->cut_id != 0xffffffffu ? pg->cflag : nullptr, idx->dev_nq);                                     \
    } while (0)
#define GM_MERGE_KC(KC) do { if (split == 1) GM_MERGE(KC, 1); else if (split == 2) GM_MERGE(KC, 2); else GM_MERGE(KC, 4); } while (0)
    switch (kcap) {
        case 1: GM_MERGE_KC(1); break;
        case 5: GM_MERGE_KC(5); break;
        case 10: GM_MERGE_KC(10); break;
        default: GM_MERGE_KC(20); break;
    }
#undef GM_MERGE_KC
#undef GM_MERGE
    HIP_TRY(hipGetLastError());
    if (slots) {
        hipLaunchKernelGGL(gm::k_compact_inc, dim3((unsigned)((n_wg + gm::PG_COMPACT_WGS - 1) / gm::PG_COMPACT_WGS)), dim3(256), 0, st,
                           pg->inc_slots, pg->inc_meta, (int)n_wg, pg->inc, n_inc, &pg->ctr->n_over, pg->ctr->chist);
        HIP_TRY(hipGetLastError());
    }
    return GM_OK;
}

// the counters as of this point of the stream, in pg->box->c: publish + poll (falls back to a stream synchronisation
// when the word does not arrive within a second -- a failed launch upstream, or a platform that does not make the
// store visible while the stream runs)
static int pg_read_counters(gm_pigeon* pg, hipStream_t st) {
    static const bool polled = [] {      // GM_PG_MAILBOX=0: plain copy + stream synchronisation
        const char* e = std::getenv("GM_PG_MAILBOX");
        return !(e && e[0] == '0');
    }();
    if (!polled || pg->box_off) {
        HIP_TRY(hipMemcpyAsync(&pg->box->c, pg->ctr, sizeof(PgCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemsetAsync(&pg->ctr->n_bad, 0, sizeof(int), st));
        HIP_TRY(hipStreamSynchronize(st));
        return GM_OK;
    }
    const unsigned seq = ++pg->box_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, pg->ctr, pg->box_dev, seq);
    HIP_TRY(hipGetLastError());
    // Poll the sequence word.  Whether the mailbox works is decided from the STREAM's progress, not from a wall-clock
    // limit: a stream that legitimately needs seconds to reach k_publish (a large Levenshtein pass, the caller's own
    // work queued in front) must not switch the mailbox off.  Only when the stream has drained and the word still has
    // not arrived does this platform not make the store visible: plain copies from then on.
    for (unsigned spins = 0; pg->box->seq != seq; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0x3fff) == 0x3fff) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return gm_fail(GM_ERR_DEVICE, std::string("K1P read-back: ") + hipGetErrorString(q));
            // the stream is idle: k_publish has run
            std::atomic_thread_fence(std::memory_order_acquire);
            if (pg->box->seq == seq) break;
            pg->box_off = true;
            HIP_TRY(hipMemcpy(&pg->box->c, pg->ctr, sizeof(PgCounters), hipMemcpyDeviceToHost));
            pg->box->c.n_bad = pg->box->c.n_bad_pub;      // (k_publish has re-zeroed n_bad; it kept the value it published)
            pg->box->seq = seq;
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return GM_OK;
}

// number of rows the merge listed as incomplete + the pairs the scan evaluated (one read-back; waits for the stream
// to reach this point)
int gm_pg_incomplete(gm_index* idx, int64_t* n_inc, int64_t* pairs_evaluated, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = pg_read_counters(pg, st))) return rc;
    const PgCounters h = pg->box->c;
    pg->last_c = h;
    *n_inc = h.n_inc;
    if (pg->cut_id != 0xffffffffu && pg->tune_cut == 0 && pg->cap_nq > 0 &&
        (double)h.n_inc > (0.01 + pg->expect_short) * (double)pg->cap_nq)
        pg->cut_dropped = true;
    *pairs_evaluated = 0;
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        pg->last_bpairs[b] = (int64_t)h.bpairs[b];
        *pairs_evaluated += (int64_t)h.bpairs[b];
    }
    return GM_OK;
}

// ---- speculative follow-ups (gmhip.hip run_hamming_pigeon): no host round trip inside a steady-state query ----
// The rows a level leaves short are counted on the device; the host used to wait for that count (mailbox) to size
// the follow-up launches.  A query of the same shape as the handle's last one instead runs every follow-up on a
// PREDICTED upper bound `cap` of the number of rows (the last actual count + an eighth + 1024): grids and buffers are
// sized for cap, and every row-indexed kernel behind it clamps n to the device's count (gm_index::dev_rows / dev_nq: a
// workgroup beyond the count exits at once; measured: padding the list with duplicate rows instead cost more than the
// read-backs it saved -- +9 % pairs on PAO1's second level).
// The counters still go to a mailbox (one per read-back kind, k_publish -- a one-wave kernel of its own: letting the LAST
// workgroup of the count-producing kernel publish instead was measured and costs more than the ~5 us launch it saves,
// profiles/r04/NOTES.md); the host looks at them ONCE, after the whole
// query is enqueued: a level that left more rows than its cap makes the query run again the synchronous way.
// device address of a row count the follow-up kernels clamp their launch to: 0 = rows the merge left short (n_inc),
// 1 = rows the last collect left (n_bad as of the last publish, which re-zeroes the live counter)
const int* gm_pg_count_ptr(const gm_index* idx, int which) {
    const gm_pigeon* pg = idx->pg;
    return which == 0 ? &pg->ctr->n_inc : &pg->ctr->n_bad_pub;
}

bool gm_pg_hit_path_is_registers(const gm_index* idx) { return idx->pg && idx->pg->hit_reg; }

bool gm_pg_mailbox_usable(const gm_index* idx) {
    static const bool polled = [] {
        const char* e = std::getenv("GM_PG_MAILBOX");
        return !(e && e[0] == '0');
    }();
    return polled && idx->pg && !idx->pg->box_off && idx->pg->box != nullptr;
}

// the counters as of this point of the stream go to mailbox `slot` (1 .. PG_BOXES - 1); no wait
int gm_pg_publish_async(gm_index* idx, int slot, hipStream_t st, gm_pg_pending* out) {
    gm_pigeon* pg = idx->pg;
    if (slot < 1 || slot >= PG_BOXES) return gm_fail(GM_ERR_ARG, "gm_pg_publish_async: bad slot");
    const unsigned seq = ++pg->box_seq;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, pg->ctr, pg->box_dev + slot, seq);
    HIP_TRY(hipGetLastError());
    out->pg = pg;
    out->slot = slot;
    out->seq = seq;
    return GM_OK;
}

// waits for a publish; n_inc / n_bad / pairs as of its point of the stream.  after_merge: the counters become the
// handle's "last read-back" (prefix fraction, per-block pairs, cut-off bookkeeping), as gm_pg_incomplete does
int gm_pg_pending_wait(const gm_pg_pending* p, bool after_merge, hipStream_t st, int64_t* n_inc, int64_t* n_bad, int64_t* pairs) {
    gm_pigeon* pg = p->pg;
    PgMailbox* box = pg->box + p->slot;
    for (unsigned spins = 0; box->seq != p->seq; ++spins) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 0x3fff) == 0x3fff) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) return gm_fail(GM_ERR_DEVICE, std::string("K1P read-back: ") + hipGetErrorString(q));
            std::atomic_thread_fence(std::memory_order_acquire);
            if (box->seq == p->seq) break;
            pg->box_off = true;      // the stream has drained and the word has not arrived: no mailbox on this platform
            return gm_fail(GM_ERR_DEVICE, "K1P mailbox word did not arrive");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const PgCounters h = box->c;
    *n_inc = h.n_inc;
    *n_bad = h.n_bad;
    *pairs = 0;
    for (int b = 0; b < GM_PG_MAXB; ++b) *pairs += (int64_t)h.bpairs[b];
    if (after_merge) {
        pg->last_c = h;
        for (int b = 0; b < GM_PG_MAXB; ++b) pg->last_bpairs[b] = (int64_t)h.bpairs[b];
        if (pg->cut_id != 0xffffffffu && pg->tune_cut == 0 && pg->cap_nq > 0 &&
            (double)h.n_inc > (0.01 + pg->expect_short) * (double)pg->cap_nq)
            pg->cut_dropped = true;
    }
    return GM_OK;
}

// per block of the layout in use: positions its scan evaluates per pair, pairs its part of the last launch evaluated
int gm_pg_block_info(const gm_index* idx, int b, int* positions, int64_t* pairs) {
    if (!idx->pg || b < 0 || b >= idx->pg->B) return 0;
    *positions = idx->pg->Lp[b];
    *pairs = idx->pg->last_bpairs[b];
    return 1;
}

int gm_pg_count_incomplete(gm_index* idx, int64_t n, int need, const int32_t* counts, const int32_t* d1,
                           int64_t* n_bad, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    // (n_bad is zero here: the candidate tables' offsets kernel clears the counters, every read-back re-zeroes it)
    hipLaunchKernelGGL(k_count_incomplete, grid_for(n), dim3(256), 0, st, counts, d1, n, need, &pg->ctr->n_bad);
    HIP_TRY(hipGetLastError());
    int rc;
    if ((rc = pg_read_counters(pg, st))) return rc;
    *n_bad = pg->box->c.n_bad;
    return GM_OK;
}

int gm_pg_gather(gm_index* idx, int64_t n_inc, int k, bool nearest, const uint2** q2, int32_t** ids2, int32_t** dists2,
                 int32_t** counts2, hipStream_t st, const int32_t** rows_only) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if (rows_only == nullptr && (rc = ensure(&pg->q2, &pg->q2_cap, n_inc, "compact query"))) return rc;
    if ((rc = ensure(&pg->ids2, &pg->ids2_cap, n_inc * k, "compact ids"))) return rc;
    if ((rc = ensure(&pg->dists2, &pg->dists2_cap, n_inc * k, "compact distances"))) return rc;
    if ((rc = ensure(&pg->counts2, &pg->counts2_cap, n_inc, "compact counts"))) return rc;
    if (rows_only != nullptr) {
        *rows_only = pg->inc;
        *q2 = idx->d_q;
    } else {
        hipLaunchKernelGGL(k_gather_rows, grid_for(n_inc), dim3(256), 0, st, idx->d_q, pg->inc, n_inc, pg->q2, idx->dev_rows);
        HIP_TRY(hipGetLastError());
        *q2 = pg->q2;
    }
    *ids2 = nearest ? nullptr : pg->ids2;
    *dists2 = pg->dists2;
    *counts2 = nearest ? nullptr : pg->counts2;
    return GM_OK;
}

// Prefix re-scan.  A row the merge left incomplete WITHOUT an overflow knows every target within r (its c < k hits);
// what it lacks are the k - c targets at distance r + 1 with the smallest ids.  The plain pool is in id order, so a
// scan of the first fraction f of it finds exactly those as soon as the prefix holds k - c such targets (92 at r + 1
// in all of S1's pool per row on average; fewer around a row that is short, hence the c-dependent estimate below).
// f is chosen from the rows at hand -- the merge counts them by c -- so that the prefix pass plus the plain re-scan of
// what it is expected to leave (overflowed rows included) is cheapest; 1.0 = the plain re-scan of everything is.
double gm_pg_prefix_fraction(const gm_index* idx, int k) {
    const gm_pigeon* pg = idx->pg;
    if (!pg || pg->B < 2 || !pg->tune_prefix || !pg->box) return 1.0;
    // (under the id cut-off a short row that took it does NOT know all its neighbours at distance r: the merge files it
    // with the overflowed rows -- n_over -- and only the others, which kept all their hits, count here)
    const PgCounters& h = pg->last_c;      // as read back by gm_pg_incomplete (or by the last validated speculative run)
    const double pm = pg->p_match >= 0.25 ? pg->p_match : 0.25;
    const int r = gm_pg_radius(idx);
    const double within = (double)idx->n_db * tail_within(idx->L, r, pm);
    const double next = (double)idx->n_db * (tail_within(idx->L, r + 1, pm) - tail_within(idx->L, r, pm));
    if (within <= 0.0 || next <= 0.0) return 1.0;
    const double ratio = next / within;      // neighbours at r + 1 per neighbour within r
    int64_t n_short = 0;
    for (int c = 0; c < k && c < 24; ++c) n_short += h.chist[c];
    const int64_t n_inc = n_short + h.n_over;
    if (n_short < 64) return 1.0;
    // cost model (us; measured on S1 / S3 / S4, scaled by the pool): a plain re-scan of n rows ~ 60 + B n, the prefix
    // pass ~ 45 + f B n, and whatever it leaves (overflowed rows, rows whose prefix was too thin) pays a plain re-scan
    const double B = 0.023 * ((double)idx->n_db / 600000.0) * ((double)idx->L / 20.0);
    const double plain = 60.0 + B * (double)n_inc;
    double best = 1.0, best_cost = plain;
    for (double f = 0.05; f <= 0.6001; f += 0.05) {
        // a row with c neighbours within r sits in a neighbourhood about c / (mean within r) as dense as the average:
        // it expects f c ratio targets at r + 1 in the prefix and needs k - c of them
        double left = (double)h.n_over;
        for (int c = 0; c < k && c < 24; ++c) {
            if (h.chist[c] == 0) continue;
            const double mean = f * (c > 0 ? (double)c : 0.5) * ratio;
            double p_short = 0.0, term = std::exp(-mean);      // P(Poisson(mean) < k - c)
            for (int j = 0; j < k - c; ++j) {
                p_short += term;
                term *= mean / (j + 1);
            }
            left += (double)h.chist[c] * (p_short < 1.0 ? p_short : 1.0);
        }
        const double cost = 45.0 + f * B * (double)n_inc + (left >= 0.5 ? 60.0 + B * left : 0.0);
        if (cost < best_cost) {
            best_cost = cost;
            best = f;
        }
    }
    return best;
}

// last step of the prefix re-scan (k_merge_prefix): the segment lists `keys` of the n_inc compact rows are folded into
// the ORIGINAL rows of ids / dists / counts, and the rows still short become the new pg->inc (the old list is the
// spare); their number arrives in ctr->n_bad with the caller's next read-back
int gm_pg_merge_prefix(gm_index* idx, int kcap, const uint32_t* keys, int n_lists, int64_t n_inc, int k, bool nearest,
                       int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->inc_b, &pg->inc_b_cap, pg->inc_cap > n_inc ? pg->inc_cap : n_inc, "incomplete rows (second list)")))
        return rc;
    constexpr int SP = 16;
    const dim3 grid = grid_for(n_inc * SP), block(256);
    const int nr = nearest ? 1 : 0;
    const uint32_t min_raw = ((uint32_t)(gm_pg_radius(idx) + 1) << gm::ID_BITS) + 1u;      // keys at distance r + 1 (and beyond)
    int32_t* ids_o = nearest ? nullptr : ids;
    int32_t* counts_o = nearest ? nullptr : counts;
    int* n_bad = &pg->ctr->n_bad;
    switch (kcap) {
        case 1: hipLaunchKernelGGL((gm::k_merge_prefix<1, SP>), grid, block, 0, st, keys, n_lists, pg->inc, n_inc, k, min_raw, ids_o, dists, counts_o, nr, pg->inc_b, n_bad, idx->dev_rows); break;
        case 5: hipLaunchKernelGGL((gm::k_merge_prefix<5, SP>), grid, block, 0, st, keys, n_lists, pg->inc, n_inc, k, min_raw, ids_o, dists, counts_o, nr, pg->inc_b, n_bad, idx->dev_rows); break;
        case 10: hipLaunchKernelGGL((gm::k_merge_prefix<10, SP>), grid, block, 0, st, keys, n_lists, pg->inc, n_inc, k, min_raw, ids_o, dists, counts_o, nr, pg->inc_b, n_bad, idx->dev_rows); break;
        default: hipLaunchKernelGGL((gm::k_merge_prefix<20, SP>), grid, block, 0, st, keys, n_lists, pg->inc, n_inc, k, min_raw, ids_o, dists, counts_o, nr, pg->inc_b, n_bad, idx->dev_rows); break;
    }
    HIP_TRY(hipGetLastError());
    std::swap(pg->inc, pg->inc_b);
    std::swap(pg->inc_cap, pg->inc_b_cap);
    return GM_OK;
}

// the rows the prefix re-scan left short -> pg->inc (the old list becomes the spare); their number arrives in
// ctr->n_bad with the caller's next read-back (gm_pg_count_incomplete is NOT needed after this)
int gm_pg_collect_bad(gm_index* idx, int64_t n_inc, int need, bool nearest, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->inc_b, &pg->inc_b_cap, pg->inc_cap > n_inc ? pg->inc_cap : n_inc, "incomplete rows (second list)")))
        return rc;
    hipLaunchKernelGGL(k_collect_bad, grid_for(n_inc), dim3(256), 0, st, pg->inc, n_inc, need, nearest ? nullptr : pg->counts2,
                       nearest ? pg->dists2 : nullptr, pg->inc_b, &pg->ctr->n_bad, idx->dev_rows);
    HIP_TRY(hipGetLastError());
    std::swap(pg->inc, pg->inc_b);
    std::swap(pg->inc_cap, pg->inc_b_cap);
    return GM_OK;
}

// ctr->n_bad as of this point of the stream (read-back; re-zeroes it)
int gm_pg_read_bad(gm_index* idx, int64_t* n_bad, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = pg_read_counters(pg, st))) return rc;
    *n_bad = pg->box->c.n_bad;
    return GM_OK;
}

int gm_pg_collect_short(gm_index* idx, int64_t n, int need, const int32_t* counts, hipStream_t st) {
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = pg_ensure_counters(pg))) return rc;
    if ((rc = ensure(&pg->inc, &pg->inc_cap, n, "incomplete rows"))) return rc;
    hipLaunchKernelGGL(k_collect_short, grid_for(n), dim3(256), 0, st, counts, n, need, pg->inc, &pg->ctr->n_bad);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

// every stride-th candidate row -> pg->inc (the planner's sample; gm_pg_gather then builds the compact query)
int gm_pg_sample_rows(gm_index* idx, int64_t n, int64_t stride, hipStream_t st) {
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->inc, &pg->inc_cap, n, "incomplete rows"))) return rc;
    hipLaunchKernelGGL(k_stride_rows, grid_for(n), dim3(256), 0, st, pg->inc, n, stride);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
// distance histogram of the n sample rows gathered in pg->q2 (gm_pg_sample_rows + gm_pg_gather) against the whole pool:
// cum[s * GM_PG_SAMPLE_BINS + d] = targets within d of row s (d = 0 .. 8); synchronises the stream
int gm_pg_sample_hist(gm_index* idx, int64_t n, hipStream_t st, std::vector<uint32_t>* cum) {
    static_assert(PG_SAMPLE_BINS == GM_PG_SAMPLE_BINS, "sample bins");
    gm_pigeon* pg = idx->pg;
    if (!pg || !pg->q2 || n < 1 || n > 4096) return gm_fail(GM_ERR_ARG, "gm_pg_sample_hist: no sample gathered");
    int rc;
    if ((rc = ensure(&pg->shist, &pg->shist_cap, n * PG_SAMPLE_BINS, "sample histogram"))) return rc;
    HIP_TRY(hipMemsetAsync(pg->shist, 0, sizeof(uint32_t) * (size_t)(n * PG_SAMPLE_BINS), st));
    const size_t lds = sizeof(uint32_t) * (size_t)(n * (2 + PG_SAMPLE_BINS));
    // (function attributes belong to a device: gm_multi_* drives several devices from one process)
    static bool attr_set[64] = {};
    const int dev = idx->device >= 0 && idx->device < 64 ? idx->device : 0;
    if (!attr_set[dev]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sample_hist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_sample_hist, dim3((unsigned)(idx->n_simd / 4)), dim3(1024), lds, st, idx->d_db, idx->n_db, pg->q2, (int)n,
                       pg->shist);
    HIP_TRY(hipGetLastError());
    cum->resize((size_t)(n * PG_SAMPLE_BINS));
    HIP_TRY(hipMemcpyAsync(cum->data(), pg->shist, sizeof(uint32_t) * cum->size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t sI = 0; sI < n; ++sI)
        for (int d = 1; d < PG_SAMPLE_BINS; ++d) (*cum)[(size_t)(sI * PG_SAMPLE_BINS + d)] += (*cum)[(size_t)(sI * PG_SAMPLE_BINS + d - 1)];
    return GM_OK;
}
bool gm_pg_sampling(const gm_index* idx) { return !idx->pg || idx->pg->tune_sample != 0; }
void gm_pg_expect_short(gm_index* idx, double share) {
    if (idx->pg) idx->pg->expect_short = share;
}
void gm_pg_expect_hits(gm_index* idx, double hits) {
    if (idx->pg) idx->pg->expect_hits = hits;
}

const int32_t* gm_pg_inc_list(const gm_index* idx) { return idx->pg ? idx->pg->inc : nullptr; }
int gm_pg_debug_flags(const gm_index* idx) {
    (void)idx;
#ifdef GM_PG_TIMING
    return GM_PG_TIMING;      // a timing build's answers are wrong on purpose: the handle must not learn from them
#else
    return 0;
#endif
}

int gm_pg_scatter(gm_index* idx, int64_t n_inc, int k, bool nearest, int32_t* ids, int32_t* dists, int32_t* counts,
                  hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    hipLaunchKernelGGL(k_scatter_rows, grid_for(n_inc * k), dim3(256), 0, st, pg->inc, n_inc, k,
                       nearest ? nullptr : pg->ids2, pg->dists2, nearest ? nullptr : pg->counts2, nearest ? nullptr : ids,
                       dists, nearest ? nullptr : counts, idx->dev_rows);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
