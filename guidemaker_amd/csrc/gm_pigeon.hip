// gm_pigeon.hip -- host side of K1P, the pigeonhole-partitioned Hamming scan (kernels: k_hamming_scan_pg and
// k_merge_hits in gm_lookup.hip.h).  Exact k-NN without evaluating all pairs:
//
//   two sequences with at most r mismatches agree exactly on at least one of r + 1 disjoint blocks of
//   positions; with the one-substitution neighbour buckets of X of the blocks scanned as well, B blocks
//   serve r = B + X - 1 (see gm_pg_plan).  Pool side (once per index and layout): for every block a stable
//   sort of the pool by the block's value (ids ascend inside a bucket -> the kernels' strict test keeps the
//   canonical (distance, id) order), the permutation, the bucket offsets and a one-hot table of the sorted
//   copy.  Query side: the candidates are grouped by the same values and cut into chunks of <= 64 that do not
//   straddle buckets; one workgroup scans one chunk against one bucket (its own, or a neighbour bucket of an
//   expansion block) and appends its hits to the candidate's slot list; the lists of a candidate are merged
//   (duplicates dropped).  Work on S1: 5.6 % of all pairs (L = 20, blocks 5,5,5,5, three of them expanded).
//   A candidate whose merged list is full is exact: every target within r was enumerated.  The rest (the
//   layout is chosen so that they are few) are re-scanned by the plain two-phase kernel (gmhip.hip).
//
// Everything here is integer bookkeeping on the device: block keys, a stable rocPRIM sort per pool copy, one
// counting sort of the candidates for all blocks, a chunk table built by a count / scan / fill triple.  No
// host round trip except the read-backs of the incomplete-row counts.
#include "gm_host.h"

#include <cmath>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gm_lookup.hip.h"      // k_to_onehot, tile geometry

namespace {

struct PgLayoutArg {              // per-block geometry + offsets, by value
    int B;
    int ebase[GM_PG_MAXB + 1];    // first entry of block b in the flattened (block, bucket) list
    const int32_t* coff[GM_PG_MAXB];
    const int32_t* toff[GM_PG_MAXB];
    // expansion: after the ebase[B] (block, bucket) entries come, for each of the X expansion blocks (blocks
    // 0 .. X-1, all of the same length), nbx * xne entries (bucket, neighbour)
    int X;                        // expansion blocks, 0 = none
    int nbx;                      // buckets of an expansion block
    int xne;                      // neighbours per bucket = 3 x block length
    int xskip;                    // leading block positions the copy's table leaves out
    int n_entries;
};

struct PgEntry {
    int b, vc, vt, lst, add;      // block, candidates' bucket, target bucket, slot list, known hidden mismatches
};

struct PgCounters {               // device-resident
    int n_chunks;
    int n_inc;
    unsigned long long pairs;     // pairs the last K1P launch evaluates (chunks x 64 x scanned targets)
    int n_bad;                    // re-scanned rows that did not fill under the guessed bound
    int pad;
};

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

// ---- candidate grouping: one counting sort for ALL blocks (order inside a bucket is irrelevant) ----
struct PgKeyArg {
    int B;
    int bstart[GM_PG_MAXB], blen[GM_PG_MAXB];
    int ebase[GM_PG_MAXB + 1];
    int32_t* coff[GM_PG_MAXB];
    uint32_t* cperm[GM_PG_MAXB];
};

__device__ __forceinline__ int pg_key(const uint2 e, int start, int len) {
    int key = 0;
    for (int j = 0; j < len; ++j) {
        const int p = start + j;
        key |= (int)(((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1)) << (2 * j);
    }
    return key;
}

// hist[entry] += candidates with that (block, value); LDS-privatised per workgroup when the table fits
__global__ __launch_bounds__(1024) void k_cand_hist(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                    int* __restrict__ hist, int use_lds) {
    extern __shared__ int s_hist[];
    const int n_entries = K.ebase[K.B];
    if (use_lds) {
        for (int e = threadIdx.x; e < n_entries; e += blockDim.x) s_hist[e] = 0;
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint2 e = planes[i];
        for (int b = 0; b < K.B; ++b) {
            const int ent = K.ebase[b] + pg_key(e, K.bstart[b], K.blen[b]);
            if (use_lds) atomicAdd(&s_hist[ent], 1); else atomicAdd(&hist[ent], 1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < n_entries; e += blockDim.x)
            if (s_hist[e] != 0) atomicAdd(&hist[e], s_hist[e]);
    }
}

// per block (blockIdx.x = b): exclusive scan of its buckets -> coff[b][0..nb], cursor[entry] = bucket start
__global__ __launch_bounds__(1024) void k_cand_offsets(PgKeyArg K, const int* __restrict__ hist, int* __restrict__ cursor) {
    __shared__ int s[1024];
    const int b = blockIdx.x;
    const int e0 = K.ebase[b], nb = K.ebase[b + 1] - e0;
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    int sum = 0;
    for (int v = lo; v < hi; ++v) sum += hist[e0 + v];
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int t = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    int run = threadIdx.x ? s[threadIdx.x - 1] : 0;
    for (int v = lo; v < hi; ++v) {
        K.coff[b][v] = run;
        cursor[e0 + v] = run;
        run += hist[e0 + v];
    }
    if (threadIdx.x == 1023) K.coff[b][nb] = s[1023];
}

// cperm[b][position] = candidate row; positions handed out per bucket: rank inside the workgroup from an
// LDS counter, one global atomic per (workgroup, non-empty bucket) for the workgroup's base
__global__ __launch_bounds__(1024) void k_cand_scatter(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                       int* __restrict__ cursor, int use_lds) {
    extern __shared__ int s_cnt[];      // [n_entries] counts, then reused as bases
    const int n_entries = K.ebase[K.B];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int ent[GM_PG_MAXB], rank[GM_PG_MAXB];
    if (use_lds) {
        for (int e = threadIdx.x; e < n_entries; e += blockDim.x) s_cnt[e] = 0;
        __syncthreads();
    }
    if (i < n) {
        const uint2 e = planes[i];
#pragma unroll
        for (int b = 0; b < GM_PG_MAXB; ++b) {
            if (b < K.B) {
                ent[b] = K.ebase[b] + pg_key(e, K.bstart[b], K.blen[b]);
                rank[b] = use_lds ? atomicAdd(&s_cnt[ent[b]], 1) : atomicAdd(&cursor[ent[b]], 1);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < n_entries; e += blockDim.x) {
            const int c = s_cnt[e];
            s_cnt[e] = c != 0 ? atomicAdd(&cursor[e], c) : 0;
        }
        __syncthreads();
    }
    if (i < n) {
#pragma unroll
        for (int b = 0; b < GM_PG_MAXB; ++b) {
            if (b < K.B) {
                const int pos = use_lds ? s_cnt[ent[b]] + rank[b] : rank[b];
                K.cperm[b][pos] = (uint32_t)i;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gather_planes(const uint2* __restrict__ planes, const uint32_t* __restrict__ perm,
                                                       int64_t n, uint2* __restrict__ out, int32_t* __restrict__ perm_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = planes[perm[i]];
    perm_out[i] = (int32_t)perm[i];
}

__device__ __forceinline__ void pg_entry(const PgLayoutArg& A, int e, int& b, int& v) {
    b = 0;
    while (b + 1 < A.B && e >= A.ebase[b + 1]) ++b;
    v = e - A.ebase[b];
}

// entry -> (block, candidates' bucket, target bucket, list, add).  Neighbour e of bucket v replaces the base at
// block position e / 3 by one of the three others; if that position is one the table leaves out, the scan
// cannot see the mismatch and the chunk carries add = 1.
__device__ __forceinline__ PgEntry pg_entry_ex(const PgLayoutArg& A, int e) {
    PgEntry r;
    if (e < A.ebase[A.B]) {
        pg_entry(A, e, r.b, r.vc);
        r.vt = r.vc;
        r.lst = r.b;
        r.add = 0;
        return r;
    }
    const int rem = e - A.ebase[A.B], per = A.nbx * A.xne;
    const int xb = rem / per, x = rem % per;
    const int v = x / A.xne, ne = x % A.xne;
    const int pidx = ne / 3, alt = ne % 3;
    const int code = (v >> (2 * pidx)) & 3;
    r.b = xb;
    r.vc = v;
    r.vt = (v & ~(3 << (2 * pidx))) | (((code + 1 + alt) & 3) << (2 * pidx));
    r.lst = A.B + xb * A.xne + ne;
    r.add = pidx < A.xskip ? 1 : 0;
    return r;
}

// chunks per entry (none when either side of the entry is empty)
__global__ __launch_bounds__(256) void k_chunk_count(PgLayoutArg A, int* __restrict__ nch) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n_entries) return;
    const PgEntry en = pg_entry_ex(A, e);
    const int cnt = A.coff[en.b][en.vc + 1] - A.coff[en.b][en.vc];
    const int tgt = A.toff[en.b][en.vt + 1] - A.toff[en.b][en.vt];
    nch[e] = tgt > 0 ? (cnt + 63) / 64 : 0;
}

// n_chunks = last exclusive offset + last count
__global__ void k_chunk_total(const int* __restrict__ nch, const int* __restrict__ cbase, int n, int* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = n > 0 ? cbase[n - 1] + nch[n - 1] : 0;
}

__global__ __launch_bounds__(256) void k_fill_chunks(PgLayoutArg A, const int* __restrict__ cbase, int n_tiles,
                                                     gm_pg_chunk* __restrict__ chunks, PgCounters* __restrict__ ctr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n_entries) return;
    const PgEntry en = pg_entry_ex(A, e);
    const int c0 = A.coff[en.b][en.vc], c1 = A.coff[en.b][en.vc + 1];
    const int lo = A.toff[en.b][en.vt], hi = A.toff[en.b][en.vt + 1];
    if (c1 <= c0 || hi <= lo) return;
    int w = cbase[e];
    for (int c = c0; c < c1; c += 64, ++w)
        chunks[w] = gm_pg_chunk{en.b, en.vt, c, c1 - c < 64 ? c1 - c : 64, en.lst, en.add};
    // the 64-target groups the kernel will evaluate for this bucket (same rounding as k_hamming_scan_pg)
    (void)n_tiles;
    const int g_lo = lo / gm::LK_GROUP, g_hi = (hi + gm::LK_GROUP - 1) / gm::LK_GROUP;
    const unsigned long long pairs =
        (unsigned long long)((c1 - c0 + 63) / 64) * 64ull * (unsigned long long)(g_hi - g_lo) * gm::LK_GROUP;
    atomicAdd(&ctr->pairs, pairs);
}

__global__ __launch_bounds__(256) void k_collect_incomplete(const int32_t* __restrict__ counts, const int32_t* __restrict__ d1,
                                                            int64_t n_q, int need, int32_t* __restrict__ inc,
                                                            PgCounters* __restrict__ ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_q) return;
    const bool incomplete = counts != nullptr ? counts[i] < need : d1[i] < 0;
    if (incomplete) inc[atomicAdd(&ctr->n_inc, 1)] = (int32_t)i;
}

__global__ __launch_bounds__(256) void k_count_incomplete(const int32_t* __restrict__ counts, const int32_t* __restrict__ d1,
                                                          int64_t n, int need, int* __restrict__ n_bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (counts != nullptr ? counts[i] < need : d1[i] < 0) atomicAdd(n_bad, 1);
}

__global__ __launch_bounds__(256) void k_gather_rows(const uint2* __restrict__ q, const int32_t* __restrict__ inc, int64_t n,
                                                     uint2* __restrict__ q2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) q2[j] = q[inc[j]];
}

__global__ __launch_bounds__(256) void k_scatter_rows(const int32_t* __restrict__ inc, int64_t n, int k,
                                                      const int32_t* __restrict__ ids2, const int32_t* __restrict__ dists2,
                                                      const int32_t* __restrict__ counts2, int32_t* __restrict__ ids,
                                                      int32_t* __restrict__ dists, int32_t* __restrict__ counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * k) return;
    const int64_t j = t / k;
    const int s = (int)(t % k);
    const int64_t row = inc[j];
    if (ids != nullptr) ids[row * k + s] = ids2[t];
    dists[row * k + s] = dists2[t];
    if (counts != nullptr && s == 0) counts[row] = counts2[j];
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

struct gm_pigeon {
    int B = 0;
    double p_match = 0.0;              // P(two pool sequences carry the same base at a position); 0 = not measured
    int max_bucket[GM_PG_MAXB]{};      // largest bucket of every pool copy
    int expand = 0;                    // X: the neighbour buckets of blocks 0 .. X-1 are scanned too, B blocks then
                                       // serve radius r = B + X - 1
    int cap_x = 0;                     // slots per neighbour list
    int skip_len = 0;                  // positions of its own block each copy's table leaves out
    int Lp = 0;                        // positions the scan evaluates = L - skip_len (>= 10)
    int bstart[GM_PG_MAXB]{}, blen[GM_PG_MAXB]{};
    int ebase[GM_PG_MAXB + 1]{};
    // pool side
    uint32_t* table[GM_PG_MAXB]{};
    int32_t* perm[GM_PG_MAXB]{};
    int32_t* toff[GM_PG_MAXB]{};
    // query side (grow-only)
    uint32_t* cperm[GM_PG_MAXB]{};
    int64_t cperm_cap[GM_PG_MAXB]{};
    int32_t* coff[GM_PG_MAXB]{};
    uint32_t *k0 = nullptr, *k1 = nullptr, *r0 = nullptr, *r1 = nullptr;
    int64_t k0_cap = 0, k1_cap = 0, r0_cap = 0, r1_cap = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_cap = 0;
    int *nch = nullptr, *cbase = nullptr, *hist = nullptr, *cursor = nullptr;
    int64_t nch_cap = 0, cbase_cap = 0, hist_cap = 0, cursor_cap = 0;
    gm_pg_chunk* chunks = nullptr;
    int64_t chunks_cap = 0;
    PgCounters* ctr = nullptr;
    int64_t grid = 0;
    uint32_t* keys = nullptr;     // slot lists [B][cap][n_q]
    int64_t keys_cap = 0;
    int32_t* n_hits = nullptr;    // [B][n_q]
    int64_t n_hits_cap = 0;
    int cap = 0;
    int64_t cap_nq = -1;          // candidate count / expansion the capacities were sized for
    int cap_x_for = -1;
    bool cap_k1 = false;
    gm_pg_blocks blocks{};
    int32_t* inc = nullptr;
    int64_t inc_cap = 0;
    uint2* q2 = nullptr;
    int64_t q2_cap = 0;
    int32_t *ids2 = nullptr, *dists2 = nullptr, *counts2 = nullptr;
    int64_t ids2_cap = 0, dists2_cap = 0, counts2_cap = 0;
};

namespace {

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
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        if (pg->table[b]) (void)hipFree(pg->table[b]);
        if (pg->perm[b]) (void)hipFree(pg->perm[b]);
        if (pg->toff[b]) (void)hipFree(pg->toff[b]);
        if (pg->coff[b]) (void)hipFree(pg->coff[b]);
        pg->table[b] = nullptr;
        pg->perm[b] = nullptr;
        pg->toff[b] = nullptr;
        pg->coff[b] = nullptr;
    }
    pg->B = 0;
    pg->cap = 0;
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

// Layout choice under an independent-base model with the pool's own composition (p_match = sum_b f_b^2).
// Plain mode (X = 0): B = r + 1 blocks, every pair within r shares a block.  Expansion (X > 0): B blocks serve
// r = B + X - 1 -- a pair within r that shares no block has m_b >= 1 mismatches in every block with
// sum m_b <= r, so at most X - 1 blocks carry two or more and at least one of the X expansion blocks carries
// exactly one: the pair sits in a one-substitution neighbour bucket of that block (3 x len per bucket), and
// those buckets get one extra pass each.
// cost = fraction of the pairs evaluated + P(Poisson(n P(d <= r)) < k) (rows re-scanned); 0 when the plain
// scan is at least as good.  max_len = longest block the caller can use (buckets must hold enough candidates).
// The block lengths are chosen too: block 0 (the expansion block) may be longer than the others.
int gm_pg_plan(int64_t n_db, int64_t n_q, int L, int k, double p_match, int max_len, int* expand, int* len0_out) {
    if (expand) *expand = 0;
    if (len0_out) *len0_out = 0;
    if (max_len > GM_PG_MAXLEN) max_len = GM_PG_MAXLEN;
    if (L < 12 || L > GM_MAX_GUIDELEN || n_db < 32768 || k < 1 || k > 20) return 0;
    if (!(p_match >= 0.25 && p_match < 0.9)) p_match = 0.25;
    if (n_q < 1) n_q = n_db;
    double best = 0.45;      // K1P has to be clearly cheaper than scanning everything
    int best_B = 0;
    for (int X = 0; X <= 3; ++X) {
        for (int B = X + 1 > 2 ? X + 1 : 2; B <= GM_PG_MAXB; ++B) {
            const int r = B + X - 1;
            const int even = (L + B - 1) / B;
            // longer expansion blocks (small buckets, cheap neighbour passes) against shorter others
            for (int lenx = even; lenx <= (X ? max_len : even); ++lenx) {
                int blen[GM_PG_MAXB];
                if (!pg_block_lengths(L, B, X, lenx, max_len, blen)) continue;
                // fraction of the pairs evaluated, with the padding of a bucket's ends to 64 targets / candidates
                double work = 0.0;
                for (int b = 0; b < B; ++b) {
                    const double share = std::pow(p_match, (double)blen[b]);
                    const double pad = (1.0 + 32.0 / (share * (double)n_db)) * (1.0 + 32.0 / (share * (double)n_q));
                    double passes = 1.0;
                    if (b < X) passes += 3.0 * blen[b] * ((1.0 - p_match) / 3.0) / p_match;
                    work += share * pad * passes;
                }
                const double mean = (double)n_db * tail_within(L, r, p_match);
                double p_inc = 0.0, term = std::exp(-mean);      // P(Poisson(mean) < k)
                for (int j = 0; j < k; ++j) {
                    p_inc += term;
                    term *= mean / (j + 1);
                }
                if (mean > 700.0) p_inc = 0.0;
                const double cost = work + (p_inc < 1.0 ? p_inc : 1.0);
                if (cost < best) {
                    best = cost;
                    best_B = B;
                    if (expand) *expand = X;
                    if (len0_out) *len0_out = X ? lenx : 0;
                }
            }
        }
    }
    return best_B;
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
        for (int b = 0; b < 4; ++b) pm += (h[b] / total) * (h[b] / total);
        pg->p_match = pm;
    }
    *p_match = pg->p_match;
    return GM_OK;
}

int gm_pg_blocks_of(const gm_index* idx) { return idx->pg ? idx->pg->B : 0; }
int gm_pg_radius(const gm_index* idx) { return idx->pg ? idx->pg->B + idx->pg->expand - 1 : 0; }

void gm_pg_free(gm_index* idx) {
    gm_pigeon* pg = idx->pg;
    if (!pg) return;
    free_pool_side(pg);
    for (int b = 0; b < GM_PG_MAXB; ++b)
        if (pg->cperm[b]) (void)hipFree(pg->cperm[b]);
    void* bufs[] = {pg->k0, pg->k1, pg->r0, pg->r1, pg->sort_tmp, pg->nch, pg->cbase, pg->hist, pg->cursor, pg->chunks, pg->ctr, pg->keys,
                    pg->n_hits, pg->inc, pg->q2, pg->ids2, pg->dists2, pg->counts2};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    delete pg;
    idx->pg = nullptr;
}

int gm_pg_prepare(gm_index* idx, int B, int expand, int len0, hipStream_t st) {
    const int L = idx->L;
    int blen[GM_PG_MAXB];
    if (!pg_block_lengths(L, B, expand, len0, GM_PG_MAXLEN, blen)) return gm_fail(GM_ERR_ARG, "gm_pg_prepare: bad layout");
    if (!idx->pg) idx->pg = new gm_pigeon();
    gm_pigeon* pg = idx->pg;
    pg->expand = expand;
    bool same = pg->B == B;
    for (int b = 0; b < B && same; ++b) same = pg->blen[b] == blen[b];
    if (same) return GM_OK;            // the pool copies depend on the block lengths only
    free_pool_side(pg);
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
    if (!pg->ctr && hipMalloc(reinterpret_cast<void**>(&pg->ctr), sizeof(PgCounters)) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P counters) failed");
    // every pair of a bucket agrees on the copy's own block: leave the first `base` (= shortest block length)
    // positions of the block out of the table and run the scan for L - base positions
    pg->skip_len = base;
    if (L - pg->skip_len < 10) pg->skip_len = L - 10;
    pg->Lp = L - pg->skip_len;
    const int64_t n_groups = idx->n_lk_tiles * gm::LK_TG;
    const int64_t words = n_groups * gm::lk_group_words(pg->Lp);
    uint2* sorted_planes = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&sorted_planes), sizeof(uint2) * (size_t)n) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "hipMalloc(sorted planes) failed");
    for (int b = 0; b < B && rc == GM_OK; ++b) {
        const int nb = 1 << (2 * pg->blen[b]);
        if (hipMalloc(reinterpret_cast<void**>(&pg->table[b]), sizeof(uint32_t) * (size_t)words) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&pg->perm[b]), sizeof(int32_t) * (size_t)n) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&pg->toff[b]), sizeof(int32_t) * (size_t)(nb + 1)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&pg->coff[b]), sizeof(int32_t) * (size_t)(nb + 1)) != hipSuccess) {
            rc = gm_fail(GM_ERR_ALLOC, "hipMalloc(K1P pool copy) failed");
            break;
        }
        hipLaunchKernelGGL(k_block_keys, grid_for(n), dim3(256), 0, st, idx->d_db, n, pg->bstart[b], pg->blen[b], pg->k0,
                           pg->r0);
        if ((rc = sort_pairs(pg, n, 2 * pg->blen[b], st))) break;
        hipLaunchKernelGGL(k_bucket_offsets, grid_for(nb + 1), dim3(256), 0, st, pg->k1, n, nb, pg->toff[b]);
        hipLaunchKernelGGL(k_gather_planes, grid_for(n), dim3(256), 0, st, idx->d_db, pg->r1, n, sorted_planes, pg->perm[b]);
        hipLaunchKernelGGL(gm::k_to_onehot, grid_for(words), dim3(256), 0, st, sorted_planes, n, pg->table[b], n_groups,
                           pg->Lp, pg->bstart[b], pg->skip_len);
        if (hipGetLastError() != hipSuccess) rc = gm_fail(GM_ERR_DEVICE, "K1P pool-copy kernels failed to launch");
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == GM_OK) rc = gm_fail(GM_ERR_DEVICE, "K1P pool-copy build failed");
    (void)hipFree(sorted_planes);
    for (int b = 0; b < B && rc == GM_OK; ++b) {      // largest bucket per copy: sizes the slot lists
        const int nb = 1 << (2 * pg->blen[b]);
        std::vector<int32_t> off((size_t)nb + 1);
        if (hipMemcpy(off.data(), pg->toff[b], sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost) != hipSuccess) {
            rc = gm_fail(GM_ERR_DEVICE, "K1P bucket offsets read-back failed");
            break;
        }
        int mx = 0;
        for (int v = 0; v < nb; ++v) mx = off[v + 1] - off[v] > mx ? off[v + 1] - off[v] : mx;
        pg->max_bucket[b] = mx;
    }
    if (rc != GM_OK) {
        free_pool_side(pg);
        return rc;
    }
    pg->B = B;
    return GM_OK;
}

// group the candidates in idx->d_q by every block's value and build the chunk table
int gm_pg_candidates(gm_index* idx, int64_t n_q, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    if (!pg || pg->B < 2) return gm_fail(GM_ERR_ARG, "gm_pg_candidates: no layout prepared");
    if (n_q >= (int64_t)1 << 31) return gm_fail(GM_ERR_ARG, "gm_pg_candidates: too many candidates");
    int rc;
    const int n_entries = pg->ebase[pg->B];      // (block, bucket) entries = histogram bins
    if ((rc = ensure(&pg->hist, &pg->hist_cap, n_entries, "bucket histogram"))) return rc;
    if ((rc = ensure(&pg->cursor, &pg->cursor_cap, n_entries, "bucket cursors"))) return rc;
    PgLayoutArg A{};
    A.B = pg->B;
    PgKeyArg K{};
    K.B = pg->B;
    int64_t grid = 0;
    for (int b = 0; b <= pg->B; ++b) A.ebase[b] = K.ebase[b] = pg->ebase[b];
    for (int b = 0; b < pg->B; ++b) {
        if ((rc = ensure(&pg->cperm[b], &pg->cperm_cap[b], n_q, "candidate order"))) return rc;
        K.bstart[b] = pg->bstart[b];
        K.blen[b] = pg->blen[b];
        K.coff[b] = pg->coff[b];
        K.cperm[b] = pg->cperm[b];
        A.coff[b] = pg->coff[b];
        A.toff[b] = pg->toff[b];
        grid += (n_q + 63) / 64 + (1 << (2 * pg->blen[b]));
    }
    // counting sort of the candidates by every block's value: histogram, per-block scan, scatter
    const int use_lds = n_entries <= 8192 ? 1 : 0;
    const size_t lds = use_lds ? sizeof(int) * (size_t)n_entries : 0;
    HIP_TRY(hipMemsetAsync(pg->hist, 0, sizeof(int) * (size_t)n_entries, st));
    hipLaunchKernelGGL(k_cand_hist, grid_for(n_q, 1024), dim3(1024), lds, st, idx->d_q, n_q, K, pg->hist, use_lds);
    hipLaunchKernelGGL(k_cand_offsets, dim3(pg->B), dim3(1024), 0, st, K, pg->hist, pg->cursor);
    hipLaunchKernelGGL(k_cand_scatter, grid_for(n_q, 1024), dim3(1024), lds, st, idx->d_q, n_q, K, pg->cursor, use_lds);
    HIP_TRY(hipGetLastError());
    // chunk table: the own-bucket entries of every block, plus the neighbour buckets of the expansion blocks
    A.X = pg->expand;
    A.xne = pg->expand ? 3 * pg->blen[0] : 1;
    A.nbx = 1 << (2 * pg->blen[0]);
    A.xskip = pg->skip_len;
    A.n_entries = n_entries + pg->expand * A.nbx * A.xne;
    grid += (int64_t)pg->expand * A.xne * ((n_q + 63) / 64 + A.nbx);
    if ((rc = ensure(&pg->nch, &pg->nch_cap, A.n_entries, "chunk counts"))) return rc;
    if ((rc = ensure(&pg->cbase, &pg->cbase_cap, A.n_entries, "chunk offsets"))) return rc;
    if ((rc = ensure(&pg->chunks, &pg->chunks_cap, grid, "chunk table"))) return rc;
    HIP_TRY(hipMemsetAsync(pg->ctr, 0, sizeof(PgCounters), st));
    hipLaunchKernelGGL(k_chunk_count, grid_for(A.n_entries), dim3(256), 0, st, A, pg->nch);
    {      // exclusive scan of the per-entry chunk counts (rocPRIM; up to ~50k entries)
        size_t bytes = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, pg->nch, pg->cbase, 0, (size_t)A.n_entries, rocprim::plus<int>(), st));
        if (bytes > pg->sort_tmp_cap || pg->sort_tmp == nullptr) {
            if (pg->sort_tmp) (void)hipFree(pg->sort_tmp);
            pg->sort_tmp = nullptr;
            pg->sort_tmp_cap = 0;
            if (hipMalloc(&pg->sort_tmp, bytes ? bytes : 1) != hipSuccess)
                return gm_fail(GM_ERR_ALLOC, "hipMalloc(scan workspace) failed");
            pg->sort_tmp_cap = bytes;
        }
        HIP_TRY(rocprim::exclusive_scan(pg->sort_tmp, bytes, pg->nch, pg->cbase, 0, (size_t)A.n_entries,
                                        rocprim::plus<int>(), st));
        hipLaunchKernelGGL(k_chunk_total, dim3(1), dim3(64), 0, st, pg->nch, pg->cbase, A.n_entries, &pg->ctr->n_chunks);
    }
    hipLaunchKernelGGL(k_fill_chunks, grid_for(A.n_entries), dim3(256), 0, st, A, pg->cbase, (int)idx->n_lk_tiles,
                       pg->chunks, pg->ctr);
    HIP_TRY(hipGetLastError());
    pg->grid = grid;
    return GM_OK;
}

// slots per (candidate, list): 1.25 x (mean + 6 sigma) of the hits expected in the LARGEST bucket of any copy
// under the independent-base model with the pool's own composition (own buckets: >= 16; neighbour buckets of
// the expansion pass, where the pair already has one mismatch: >= 8).  A list that still overflows (repeat
// families) only makes its row incomplete: the row is re-scanned by the plain kernel.
static void pg_slot_capacity(const gm_index* idx, gm_pigeon* pg, int64_t n_q, int k) {
    const int L = idx->L, r = pg->B + pg->expand - 1;
    const double pm = pg->p_match >= 0.25 ? pg->p_match : 0.25;
    double worst = 0.0;
    for (int b = 0; b < pg->B; ++b) {
        const double h = (double)pg->max_bucket[b] * tail_within(L - pg->blen[b], r, pm) + 1.0;
        if (h > worst) worst = h;
    }
    double cap = std::ceil(1.25 * (worst + 6.0 * std::sqrt(worst)) + 8.0);
    if (cap < 16.0) cap = 16.0;
    if (cap > 1024.0) cap = 1024.0;
    double cap_x = 0.0;
    const int n_x = pg->expand * 3 * pg->blen[0];
    if (pg->expand) {
        int mx = 0;
        for (int b = 0; b < pg->expand; ++b) mx = pg->max_bucket[b] > mx ? pg->max_bucket[b] : mx;
        const double h = (double)mx * tail_within(L - pg->blen[0], r - 1, pm);
        cap_x = std::ceil(1.25 * (h + 6.0 * std::sqrt(h)) + 8.0);
        if (cap_x > 512.0) cap_x = 512.0;
    }
    if (k == 1) {      // nearest distance: the bound follows the best hit, a list only grows by improvements
        cap = cap > 12.0 ? 12.0 : cap;
        cap_x = cap_x > 6.0 ? 6.0 : cap_x;
    }
    // keep the slot lists within a quarter of the free device memory (at least 2 GiB)
    size_t free_b = 0, total_b = 0;
    double limit = 2.0 * 1024 * 1024 * 1024;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b / 4.0 > limit) limit = (double)free_b / 4.0;
    while ((cap > 16.0 || cap_x > 8.0) && (cap * pg->B + cap_x * n_x) * (double)n_q * 4.0 > limit) {
        cap = cap > 16.0 ? std::floor(cap * 0.75) : cap;
        cap_x = cap_x > 8.0 ? std::floor(cap_x * 0.75) : cap_x;
    }
    pg->cap = (int)cap;
    pg->cap_x = (int)cap_x;
}

int gm_pg_scan(gm_index* idx, int64_t n_q, int k, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if (pg->cap == 0 || pg->cap_nq != n_q || pg->cap_x_for != pg->expand || pg->cap_k1 != (k == 1)) {
        pg_slot_capacity(idx, pg, n_q, k);      // (cached: hipMemGetInfo is not free)
        pg->cap_nq = n_q;
        pg->cap_x_for = pg->expand;
        pg->cap_k1 = k == 1;
    }
    const int n_x = pg->expand * 3 * pg->blen[0];
    if ((rc = ensure(&pg->keys, &pg->keys_cap, ((int64_t)pg->B * pg->cap + (int64_t)n_x * pg->cap_x) * n_q,
                     "K1P slot lists")))
        return rc;
    if ((rc = ensure(&pg->n_hits, &pg->n_hits_cap, (int64_t)(pg->B + n_x) * n_q, "K1P hit counts"))) return rc;
    // a (list, candidate) pair whose target bucket is empty gets no workgroup: its count must read 0
    HIP_TRY(hipMemsetAsync(pg->n_hits, 0, sizeof(int32_t) * (size_t)(pg->B + n_x) * (size_t)n_q, st));
    gm_pg_blocks& blocks = pg->blocks;
    for (int b = 0; b < pg->B; ++b) {
        blocks.table[b] = pg->table[b];
        blocks.perm[b] = pg->perm[b];
        blocks.toff[b] = pg->toff[b];
        blocks.cperm[b] = pg->cperm[b];
        blocks.skip[b] = pg->bstart[b];
    }
    blocks.skip_len = pg->skip_len;
    blocks.n_blocks = pg->B;
    blocks.xne = pg->expand ? 3 * pg->blen[0] : 1;
    blocks.n_expand = n_x;
    const int thr0 = pg->B + pg->expand;      // accept d < thr0, i.e. d <= r = B + X - 1
    switch (pg->Lp) {            // the kernel instantiated for the positions actually evaluated
#define GM_CASE(LEN) \
    case LEN: return gm_launch_pg_L##LEN(idx, &blocks, pg->chunks, &pg->ctr->n_chunks, pg->grid, n_q, k, thr0, pg->cap, pg->cap_x, pg->keys, pg->n_hits, st);
        GM_FOR_EACH_LEN(GM_CASE)
#undef GM_CASE
        default: return gm_fail(GM_ERR_ARG, "pigeonhole kernel: guidelen out of range");
    }
}

int gm_pg_merge(gm_index* idx, int kcap, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                bool nearest, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    const dim3 grid = grid_for(n_q * gm::PG_MERGE_SPLIT), block(256);
    const int nr = nearest ? 1 : 0;
    switch (kcap) {
        case 1: hipLaunchKernelGGL(gm::k_merge_hits<1>, grid, block, 0, st, pg->blocks, pg->cap, pg->cap_x, pg->keys, pg->n_hits, n_q, k, ids, dists, counts, nr); break;
        case 5: hipLaunchKernelGGL(gm::k_merge_hits<5>, grid, block, 0, st, pg->blocks, pg->cap, pg->cap_x, pg->keys, pg->n_hits, n_q, k, ids, dists, counts, nr); break;
        case 10: hipLaunchKernelGGL(gm::k_merge_hits<10>, grid, block, 0, st, pg->blocks, pg->cap, pg->cap_x, pg->keys, pg->n_hits, n_q, k, ids, dists, counts, nr); break;
        default: hipLaunchKernelGGL(gm::k_merge_hits<20>, grid, block, 0, st, pg->blocks, pg->cap, pg->cap_x, pg->keys, pg->n_hits, n_q, k, ids, dists, counts, nr); break;
    }
    HIP_TRY(hipGetLastError());
    return GM_OK;
}

int gm_pg_incomplete(gm_index* idx, int64_t n_q, int need, const int32_t* counts, const int32_t* d1, int64_t* n_inc,
                     int64_t* pairs_evaluated, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->inc, &pg->inc_cap, n_q, "incomplete rows"))) return rc;
    hipLaunchKernelGGL(k_collect_incomplete, grid_for(n_q), dim3(256), 0, st, counts, d1, n_q, need, pg->inc, pg->ctr);
    HIP_TRY(hipGetLastError());
    PgCounters h{};
    HIP_TRY(hipMemcpyAsync(&h, pg->ctr, sizeof(PgCounters), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n_inc = h.n_inc;
    *pairs_evaluated = (int64_t)h.pairs;
    return GM_OK;
}

int gm_pg_count_incomplete(gm_index* idx, int64_t n, int need, const int32_t* counts, const int32_t* d1,
                           int64_t* n_bad, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    HIP_TRY(hipMemsetAsync(&pg->ctr->n_bad, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_count_incomplete, grid_for(n), dim3(256), 0, st, counts, d1, n, need, &pg->ctr->n_bad);
    HIP_TRY(hipGetLastError());
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, &pg->ctr->n_bad, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n_bad = h;
    return GM_OK;
}

int gm_pg_gather(gm_index* idx, int64_t n_inc, int k, bool nearest, const uint2** q2, int32_t** ids2, int32_t** dists2,
                 int32_t** counts2, hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    int rc;
    if ((rc = ensure(&pg->q2, &pg->q2_cap, n_inc, "compact query"))) return rc;
    if ((rc = ensure(&pg->ids2, &pg->ids2_cap, n_inc * k, "compact ids"))) return rc;
    if ((rc = ensure(&pg->dists2, &pg->dists2_cap, n_inc * k, "compact distances"))) return rc;
    if ((rc = ensure(&pg->counts2, &pg->counts2_cap, n_inc, "compact counts"))) return rc;
    hipLaunchKernelGGL(k_gather_rows, grid_for(n_inc), dim3(256), 0, st, idx->d_q, pg->inc, n_inc, pg->q2);
    HIP_TRY(hipGetLastError());
    *q2 = pg->q2;
    *ids2 = nearest ? nullptr : pg->ids2;
    *dists2 = pg->dists2;
    *counts2 = nearest ? nullptr : pg->counts2;
    return GM_OK;
}

int gm_pg_scatter(gm_index* idx, int64_t n_inc, int k, bool nearest, int32_t* ids, int32_t* dists, int32_t* counts,
                  hipStream_t st) {
    gm_pigeon* pg = idx->pg;
    hipLaunchKernelGGL(k_scatter_rows, grid_for(n_inc * k), dim3(256), 0, st, pg->inc, n_inc, k,
                       nearest ? nullptr : pg->ids2, pg->dists2, nearest ? nullptr : pg->counts2, nearest ? nullptr : ids,
                       dists, nearest ? nullptr : counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
