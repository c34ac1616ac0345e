// gm_levpart.hip -- host side of K2P, the pigeonhole partition for the Levenshtein scan (kernels: k_leven_part_bs and
// k_merge_lists_unique in gm_leven_bs.hip.h; driver: gmhip.hip run_leven_partition).
//
//   Two sequences of the same length L within edit distance r: cut the candidate into r + 1 disjoint blocks; at most r
//   edits touch at most r blocks, so one block is copied to the target unchanged.  Where it lands is displaced by
//   (insertions - deletions before it), and an alignment of equal-length sequences of cost <= r spends at most
//   floor(r / 2) of either: |displacement| <= floor(r / 2) -- and less for the LEFTMOST untouched block when it lies near
//   either end (gm_lp_layout).  So every target within r of a candidate carries, for some block j and displacement d, the
//   candidate's bases of block j at the block's positions + d.
//
// Pool side (once per index and radius): for every (block, displacement) pair that stays inside the sequence a "copy" of
// the pool, stable-sorted by the bases at the displaced positions, buckets starting on whole K2B tiles (256 targets), with
// the position -> pool id map and the one-hot table of the copy (all L positions: the DP needs them all).  Query side: the
// candidates grouped by every block's value on the device (gm_group.hip.h), chunk table (<= 64 candidates of one bucket
// per workgroup) laid out there too.  One workgroup compares its chunk with bucket v of every copy of its block.  Work on
// P. aeruginosa PAO1 cut at L = 23 (1.16 M targets): r = 5: 20 copies, ~13 % of the pairs, then r = 6 (30 copies, 38 %) for
// the quarter of the rows whose k-th neighbour lies beyond 5.  S2 (136 k targets, k-th neighbour at 7 of 23): r = 7, 37
// copies, 62 % of the pairs in the three-diagonal band -- 0.78 of the plain scan's time.
#include "gm_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "gm_lookup.hip.h"      // lk_group_words, LK_GROUP, LK_TG

namespace {

template <class T>
int lp_ensure(T** p, int64_t* cap, int64_t need, const char* what) {
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

inline dim3 lp_grid(int64_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

#include "gm_group.hip.h"      // the counting sort of the candidates by every block's value (shared with K1P)

// per-query counters of the grouping (device; mirrored to pinned host memory after the chunk table is filled)
struct LpCounters {
    int cstart[GM_LP_MAXBLK + 1];      // chunk range of every block
    int n_chunks;
    int pad;
    unsigned long long pairs;          // candidate x target pairs the scan will evaluate (64 x 64 groups)
};

// one thread per (block, bucket) entry: its chunks of <= 64 candidates into the block's range of the chunk table
__global__ __launch_bounds__(256) void k_lp_fill_chunks(PgKeyArg K, const int* __restrict__ cbase, const uint32_t* __restrict__ gsum,
                                                        gm_lp_chunk* __restrict__ chunks, LpCounters* __restrict__ ctr) {
    __shared__ unsigned long long s_pairs;
    if (threadIdx.x == 0) s_pairs = 0ull;
    __syncthreads();
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < K.ebase[K.B]) {
        int j = 0;
        while (j + 1 < K.B && e >= K.ebase[j + 1]) ++j;
        const int v = e - K.ebase[j];
        const int c0 = K.coff[j][v], c1 = K.coff[j][v + 1];
        if (c1 > c0) {
            int w = ctr->cstart[j] + cbase[e];
            for (int c = c0; c < c1; c += 64, ++w) chunks[w] = gm_lp_chunk{j, v, c, c1 - c < 64 ? c1 - c : 64};
            atomicAdd(&s_pairs, (unsigned long long)((c1 - c0 + 63) / 64) * gsum[e] * (unsigned long long)(gm::LK_GROUP * 64));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_pairs != 0ull) atomicAdd(&ctr->pairs, s_pairs);
}

// key[i] = bases of sequence i at positions [start, start + len), rows[i] = i
__global__ __launch_bounds__(256) void k_lp_keys(const uint2* __restrict__ planes, int64_t n, int start, int len,
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

// off[v] = first sorted position whose key is >= v (v = 0 .. nb)
__global__ __launch_bounds__(256) void k_lp_offsets(const uint32_t* __restrict__ keys, int64_t n, int nb, int32_t* __restrict__ off) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > nb) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < (uint32_t)v) lo = mid + 1; else hi = mid;
    }
    off[v] = (int32_t)lo;
}

// sorted position i (key v, pool row rows[i]) -> position pad_off[v] + (i - raw_off[v]) of the padded copy
__global__ __launch_bounds__(256) void k_lp_scatter_ord(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ rows, int64_t n,
                                                        const int32_t* __restrict__ raw_off, const int64_t* __restrict__ pad_off,
                                                        uint32_t* __restrict__ ord) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = keys[i];
    ord[pad_off[v] + (i - raw_off[v])] = rows[i];
}

// one-hot table of a copy (format of gm::k_to_onehot, all L positions); padding positions (ord = 0xffffffff) set no bit
__global__ __launch_bounds__(256) void k_lp_onehot(const uint2* __restrict__ planes, const uint32_t* __restrict__ ord,
                                                   uint32_t* __restrict__ table, int64_t n_groups, int L) {
    const int gw = gm::lk_group_words(L);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_groups * gw) return;
    const int64_t G = t / gw;
    const int w = (int)(t % gw);
    const int p = w >> 3, b = (w >> 1) & 3, h = w & 1;
    uint32_t out = 0;
    for (int j = 0; j < 32; ++j) {
        const uint32_t id = ord[G * gm::LK_GROUP + 32 * h + j];
        if (id == 0xffffffffu) continue;
        const uint2 e = planes[id];
        const uint32_t code = ((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1);
        out |= (uint32_t)(code == (uint32_t)b) << j;
    }
    table[t] = out;
}

}  // namespace

struct gm_levpart {
    // layout
    int r = -1;                                   // radius the copies were built for (-1: none)
    int n_blocks = 0;
    int bstart[GM_LP_MAXBLK]{}, blen[GM_LP_MAXBLK]{};
    int n_copies = 0;
    int copy_block[GM_LP_MAXCOPY]{}, copy_shift[GM_LP_MAXCOPY]{};
    int first_copy[GM_LP_MAXBLK + 1]{};
    // pool side
    uint32_t* table[GM_LP_MAXCOPY]{};
    uint32_t* ord[GM_LP_MAXCOPY]{};
    int2* bdesc[GM_LP_MAXCOPY]{};              // bucket v of copy c -> (first tile, 64-target groups that hold targets)
    std::vector<int2> bdesc_host[GM_LP_MAXCOPY];      // (the same on the host: the per-entry group sums are formed there)
    // query side (grow-only)
    uint32_t* cperm[GM_LP_MAXBLK]{};
    int64_t cperm_cap[GM_LP_MAXBLK]{};
    uint32_t *k0 = nullptr, *k1 = nullptr, *r0 = nullptr, *r1 = nullptr;
    int64_t k0_cap = 0, k1_cap = 0, r0_cap = 0, r1_cap = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_cap = 0;
    int32_t* off = nullptr;
    int64_t off_cap = 0;
    int64_t* pad_off = nullptr;
    int64_t pad_off_cap = 0;
    gm_lp_chunk* chunks = nullptr;
    int64_t chunks_cap = 0;
    int64_t n_chunks = 0;                         // workgroups of the scan launch (an upper bound; the table's length is ctr->n_chunks)
    // candidate grouping on the device (gm_group.hip.h)
    int ebase[GM_LP_MAXBLK + 1]{};                // (block, bucket) entries: block j's start
    int32_t* coff_all = nullptr;                  // per block 4^len + 1 candidate offsets
    int *hist = nullptr, *cursor = nullptr, *cbase = nullptr, *btot = nullptr;
    uint32_t* gsum = nullptr;                     // per entry: 64-target groups of the bucket summed over the block's copies
    void* ctr = nullptr;                          // LpCounters (device)
    void* ctr_host = nullptr;                     // ... and its pinned mirror (valid after the next synchronisation)
    uint32_t* keys = nullptr;
    int64_t keys_cap = 0;
    // knobs / plan cache (gmhip.hip)
    int tune_radius = 0;                          // gm_set_param "lp_radius": 0 = choose, 1 .. 7 = that radius, -1 = never
    int64_t plan_nq = -1;
    int plan_k = -1, plan_r = 0, plan_r2 = 0;
    int tune_radius2 = 0;                         // gm_set_param "lp_radius2": second level 0 = choose, 1 .. 7 = that radius, -1 = never
};

namespace {

int lp_sort(gm_levpart* lp, int64_t n, int bits, hipStream_t st) {
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, lp->k0, lp->k1, lp->r0, lp->r1, (size_t)n, 0u, (unsigned)bits, st));
    if (bytes > lp->sort_tmp_cap || lp->sort_tmp == nullptr) {
        if (lp->sort_tmp) (void)hipFree(lp->sort_tmp);
        lp->sort_tmp = nullptr;
        lp->sort_tmp_cap = 0;
        if (hipMalloc(&lp->sort_tmp, bytes ? bytes : 1) != hipSuccess) return gm_fail(GM_ERR_ALLOC, "hipMalloc(radix sort workspace) failed");
        lp->sort_tmp_cap = bytes;
    }
    HIP_TRY(rocprim::radix_sort_pairs(lp->sort_tmp, bytes, lp->k0, lp->k1, lp->r0, lp->r1, (size_t)n, 0u, (unsigned)bits, st));
    return GM_OK;
}

int lp_sort_buffers(gm_levpart* lp, int64_t n) {
    int rc;
    if ((rc = lp_ensure(&lp->k0, &lp->k0_cap, n, "sort keys"))) return rc;
    if ((rc = lp_ensure(&lp->k1, &lp->k1_cap, n, "sort keys"))) return rc;
    if ((rc = lp_ensure(&lp->r0, &lp->r0_cap, n, "sort rows"))) return rc;
    return lp_ensure(&lp->r1, &lp->r1_cap, n, "sort rows");
}

void lp_free_pool(gm_levpart* lp) {
    for (int c = 0; c < GM_LP_MAXCOPY; ++c) {
        if (lp->table[c]) (void)hipFree(lp->table[c]);
        if (lp->ord[c]) (void)hipFree(lp->ord[c]);
        if (lp->bdesc[c]) (void)hipFree(lp->bdesc[c]);
        lp->table[c] = nullptr;
        lp->ord[c] = nullptr;
        lp->bdesc[c] = nullptr;
        lp->bdesc_host[c].clear();
    }
    lp->r = -1;
    lp->n_copies = 0;
}

}  // namespace

// The layout for radius r on guides of L bases: r + 1 blocks.  Displacements: take the LEFTMOST block j no edit touches
// (an insertion between two blocks counts for the block on its left).  Every block before it holds an edit, so at least
// j edits precede it and at most r - j follow; its displacement is the net insertion count before it (<= the edits
// before) and, the lengths being equal, minus the net count after it (<= the edits after): |d| <= min(floor(r / 2), r - j),
// and block 0 is not displaced at all.  Looking a pair up under its leftmost untouched block only therefore needs
// 1, 2 floor(r/2) + 1, ..., 3, 1 copies per block instead of 2 floor(r/2) + 1 for each (r = 5: 20 instead of 30).  Block
// lengths (2 .. 8 bases, sum L): the split that minimises the modelled share of the pool a candidate's workgroups visit,
// sum over copies of p_match ^ length (dynamic programme over block index x start position; the blocks with many copies
// get the long keys).  Returns the number of copies (0: not a usable layout) and that share in *fraction.
int gm_lp_block_shift(int r, int j) {
    if (j == 0) return 0;
    const int a = r / 2, b = r - j;
    return a < b ? a : b;
}

int gm_lp_layout(int L, int r, double p_match, int* bstart, int* blen, int* copy_block, int* copy_shift, int* first_copy,
                 double* fraction) {
    const int B = r + 1;
    if (r < 1 || B > GM_LP_MAXBLK || L < 2 * B || L > 8 * B || L > 32) return 0;
    const double pm = p_match >= 0.25 && p_match < 1.0 ? p_match : 0.25;
    auto copies_at = [&](int j, int pos, int len) {
        const int sh = gm_lp_block_shift(r, j);
        int c = 0;
        for (int d = -sh; d <= sh; ++d)
            if (pos + d >= 0 && pos + d + len <= L) ++c;
        return c;
    };
    // best[j][pos] = least cost of blocks j .. B - 1 laid from position pos to L
    constexpr double INF = 1e300;
    double best[GM_LP_MAXBLK + 1][33];
    int pick[GM_LP_MAXBLK][33];
    for (int pos = 0; pos <= L; ++pos) best[B][pos] = pos == L ? 0.0 : INF;
    for (int j = B - 1; j >= 0; --j)
        for (int pos = 0; pos <= L; ++pos) {
            best[j][pos] = INF;
            pick[j][pos] = 0;
            for (int len = 2; len <= 8 && pos + len <= L; ++len) {
                if (best[j + 1][pos + len] >= INF) continue;
                const double c = (double)copies_at(j, pos, len) * std::pow(pm, (double)len) + best[j + 1][pos + len];
                if (c < best[j][pos] - 1e-15) {      // (ties: the shorter block first)
                    best[j][pos] = c;
                    pick[j][pos] = len;
                }
            }
        }
    if (best[0][0] >= INF) return 0;
    int pos = 0, n_copies = 0;
    for (int j = 0; j < B; ++j) {
        const int len = pick[j][pos], sh = gm_lp_block_shift(r, j);
        bstart[j] = pos;
        blen[j] = len;
        first_copy[j] = n_copies;
        for (int d = -sh; d <= sh; ++d) {
            if (pos + d < 0 || pos + d + len > L) continue;
            if (n_copies >= GM_LP_MAXCOPY) return 0;
            copy_block[n_copies] = j;
            copy_shift[n_copies] = d;
            ++n_copies;
        }
        pos += len;
    }
    first_copy[B] = n_copies;
    if (fraction) *fraction = best[0][0];
    return n_copies;
}

int gm_lp_set_param(gm_index* idx, const char* name, int value, bool* known) {
    *known = false;
    const bool first = std::strcmp(name, "lp_radius") == 0;
    if (!first && std::strcmp(name, "lp_radius2") != 0) return GM_OK;
    *known = true;
    if (value < -1 || value > 7) return gm_fail(GM_ERR_ARG, "lp_radius / lp_radius2: -1 (never), 0 (choose), 1..7");
    if (!idx->lp) idx->lp = new gm_levpart();
    (first ? idx->lp->tune_radius : idx->lp->tune_radius2) = value;
    idx->lp->plan_nq = -1;
    return GM_OK;
}

static void lp_free_one(gm_levpart* lp) {
    if (!lp) return;
    lp_free_pool(lp);
    for (int j = 0; j < GM_LP_MAXBLK; ++j)
        if (lp->cperm[j]) (void)hipFree(lp->cperm[j]);
    void* bufs[] = {lp->k0, lp->k1, lp->r0, lp->r1, lp->sort_tmp, lp->off, lp->pad_off, lp->chunks, lp->keys,
                    lp->coff_all, lp->hist, lp->cursor, lp->cbase, lp->btot, lp->gsum, lp->ctr};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (lp->ctr_host) (void)hipHostFree(lp->ctr_host);
    delete lp;
}
void gm_lp_free(gm_index* idx) {
    lp_free_one(idx->lp);
    lp_free_one(idx->lp_deeper);
    idx->lp = nullptr;
    idx->lp_deeper = nullptr;
}

gm_levpart* gm_lp_state(gm_index* idx) {
    if (!idx->lp) idx->lp = new gm_levpart();
    return idx->lp;
}
int gm_lp_tune_radius(gm_index* idx) { return gm_lp_state(idx)->tune_radius; }
int gm_lp_tune_radius2(gm_index* idx) { return gm_lp_state(idx)->tune_radius2; }
void gm_lp_plan_get(gm_index* idx, int64_t* nq, int* k, int* r, int* r2) {
    gm_levpart* lp = gm_lp_state(idx);
    *nq = lp->plan_nq;
    *k = lp->plan_k;
    *r = lp->plan_r;
    *r2 = lp->plan_r2;
}
void gm_lp_plan_set(gm_index* idx, int64_t nq, int k, int r, int r2) {
    gm_levpart* lp = gm_lp_state(idx);
    lp->plan_nq = nq;
    lp->plan_k = k;
    lp->plan_r = r;
    lp->plan_r2 = r2;
}
// pairs the last K2P grouping laid out (valid once the stream has been synchronised after gm_lp_candidates)
int64_t gm_lp_last_pairs(gm_index* idx, int level) {
    const gm_levpart* lp = level == 0 ? idx->lp : idx->lp_deeper;
    return lp && lp->ctr_host ? (int64_t) static_cast<const LpCounters*>(lp->ctr_host)->pairs : 0;
}

// pool copies for radius r (kept until another radius is asked for)
int gm_lp_prepare(gm_index* idx, int r, hipStream_t st) {
    gm_levpart* lp = gm_lp_state(idx);
    if (lp->r == r) return GM_OK;
    lp_free_pool(lp);
    const int L = idx->L;
    double frac = 0.0;
    double p_match = 0.25;      // (the pool's own base composition: the same layout the planner priced)
    {
        const int rc0 = gm_pg_pool_stats(idx, st, &p_match);
        if (rc0) return rc0;
    }
    const int n_copies = gm_lp_layout(L, r, p_match, lp->bstart, lp->blen, lp->copy_block, lp->copy_shift, lp->first_copy, &frac);
    if (n_copies < 1) return gm_fail(GM_ERR_ARG, "K2P: no layout for this radius / guide length");
    lp->n_blocks = r + 1;
    {
        // Launch order: the chunk table is laid out block by block, and a chunk's work is (copies of its block) x (bucket
        // size): blocks with the longest chunks first, so that the launch does not end on a few workgroups that each still
        // have a seven-copy chunk of large buckets to go (S2: chunks of 63 .. 238 groups, four rounds of the wave slots).
        const int B = lp->n_blocks;
        int order[GM_LP_MAXBLK];
        double work[GM_LP_MAXBLK];
        for (int j = 0; j < B; ++j) {
            order[j] = j;
            work[j] = (double)(lp->first_copy[j + 1] - lp->first_copy[j]) * std::pow(p_match >= 0.25 ? p_match : 0.25, (double)lp->blen[j]);
        }
        std::stable_sort(order, order + B, [&](int a, int b) { return work[a] > work[b]; });
        int bs[GM_LP_MAXBLK], bl[GM_LP_MAXBLK], cb[GM_LP_MAXCOPY], cs[GM_LP_MAXCOPY], fc[GM_LP_MAXBLK + 1];
        int at = 0;
        for (int slot = 0; slot < B; ++slot) {
            const int j = order[slot];
            bs[slot] = lp->bstart[j];
            bl[slot] = lp->blen[j];
            fc[slot] = at;
            for (int c = lp->first_copy[j]; c < lp->first_copy[j + 1]; ++c, ++at) {
                cb[at] = slot;
                cs[at] = lp->copy_shift[c];
            }
        }
        fc[B] = at;
        for (int j = 0; j < B; ++j) {
            lp->bstart[j] = bs[j];
            lp->blen[j] = bl[j];
        }
        for (int j = 0; j <= B; ++j) lp->first_copy[j] = fc[j];
        for (int c = 0; c < n_copies; ++c) {
            lp->copy_block[c] = cb[c];
            lp->copy_shift[c] = cs[c];
        }
    }
    const int64_t n = idx->n_db;
    int rc;
    if ((rc = lp_sort_buffers(lp, n))) return rc;
    constexpr int TT = GM_LP_TILE_TARGETS;
    for (int c = 0; c < n_copies && rc == GM_OK; ++c) {
        const int j = lp->copy_block[c], len = lp->blen[j], start = lp->bstart[j] + lp->copy_shift[c];
        const int nb = 1 << (2 * len);
        if ((rc = lp_ensure(&lp->off, &lp->off_cap, nb + 1, "bucket offsets"))) break;
        if ((rc = lp_ensure(&lp->pad_off, &lp->pad_off_cap, nb + 1, "padded bucket offsets"))) break;
        hipLaunchKernelGGL(k_lp_keys, lp_grid(n), dim3(256), 0, st, idx->d_db, n, start, len, lp->k0, lp->r0);
        if ((rc = lp_sort(lp, n, 2 * len, st))) break;
        hipLaunchKernelGGL(k_lp_offsets, lp_grid(nb + 1), dim3(256), 0, st, lp->k1, n, nb, lp->off);
        std::vector<int32_t> off((size_t)nb + 1);
        if (hipMemcpyAsync(off.data(), lp->off, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            rc = gm_fail(GM_ERR_DEVICE, "K2P bucket offsets read-back failed");
            break;
        }
        std::vector<int2> bdesc((size_t)nb);
        std::vector<int64_t> pad((size_t)nb + 1);
        int64_t tiles = 0;
        for (int v = 0; v < nb; ++v) {
            const int cnt = off[(size_t)v + 1] - off[(size_t)v];
            const int t = (cnt + TT - 1) / TT;
            bdesc[(size_t)v] = make_int2((int)tiles, (cnt + gm::LK_GROUP - 1) / gm::LK_GROUP);
            pad[(size_t)v] = tiles * TT;
            tiles += t;
        }
        pad[(size_t)nb] = tiles * TT;
        if (tiles < 1) tiles = 1;
        if (tiles * TT >= ((int64_t)1 << 31)) {
            rc = gm_fail(GM_ERR_ARG, "K2P: padded pool copy too large");
            break;
        }
        const int64_t n_pad = tiles * TT, n_groups = n_pad / gm::LK_GROUP;
        const int64_t words = n_groups * gm::lk_group_words(L);
        if (hipMalloc(reinterpret_cast<void**>(&lp->ord[c]), sizeof(uint32_t) * (size_t)n_pad) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->table[c]), sizeof(uint32_t) * (size_t)words) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->bdesc[c]), sizeof(int2) * (size_t)nb) != hipSuccess) {
            rc = gm_fail(GM_ERR_ALLOC, "hipMalloc(K2P pool copy) failed");
            break;
        }
        lp->bdesc_host[c] = bdesc;
        hipError_t e = hipMemcpyAsync(lp->bdesc[c], bdesc.data(), sizeof(int2) * bdesc.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(lp->pad_off, pad.data(), sizeof(int64_t) * pad.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(lp->ord[c], 0xff, sizeof(uint32_t) * (size_t)n_pad, st);
        if (e != hipSuccess) {
            rc = gm_fail(GM_ERR_DEVICE, std::string("K2P pool copy: ") + hipGetErrorString(e));
            break;
        }
        hipLaunchKernelGGL(k_lp_scatter_ord, lp_grid(n), dim3(256), 0, st, lp->k1, lp->r1, n, lp->off, lp->pad_off, lp->ord[c]);
        hipLaunchKernelGGL(k_lp_onehot, lp_grid(words), dim3(256), 0, st, idx->d_db, lp->ord[c], lp->table[c], n_groups, L);
        if (hipGetLastError() != hipSuccess) rc = gm_fail(GM_ERR_DEVICE, "K2P pool-copy kernels failed to launch");
        // (the host vectors are pageable: staged by the runtime before the copies return; the kernels read device memory)
        if (hipStreamSynchronize(st) != hipSuccess && rc == GM_OK) rc = gm_fail(GM_ERR_DEVICE, "K2P pool-copy build failed");
    }
    if (rc != GM_OK) {
        lp_free_pool(lp);
        return rc;
    }
    lp->r = r;
    lp->n_copies = n_copies;
    // the candidate grouping's tables for this layout: (block, bucket) entries, offsets, counters, and per entry the
    // 64-target groups its bucket holds over all copies of the block (what a chunk of that bucket will evaluate)
    {
        void* old_bufs[] = {lp->coff_all, lp->hist, lp->cursor, lp->cbase, lp->btot, lp->gsum, lp->ctr};
        for (void* b : old_bufs)
            if (b) (void)hipFree(b);
        lp->coff_all = nullptr;
        lp->hist = lp->cursor = lp->cbase = lp->btot = nullptr;
        lp->gsum = nullptr;
        lp->ctr = nullptr;
        lp->ebase[0] = 0;
        int64_t coff_words = 0;
        for (int j = 0; j < lp->n_blocks; ++j) {
            lp->ebase[j + 1] = lp->ebase[j] + (1 << (2 * lp->blen[j]));
            coff_words += (1 << (2 * lp->blen[j])) + 1;
        }
        const int n_ent = lp->ebase[lp->n_blocks];
        std::vector<uint32_t> gsum((size_t)n_ent, 0u);
        for (int c = 0; c < n_copies; ++c) {
            const int j = lp->copy_block[c];
            for (size_t v = 0; v < lp->bdesc_host[c].size(); ++v) gsum[(size_t)lp->ebase[j] + v] += (uint32_t)lp->bdesc_host[c][v].y;
        }
        if (hipMalloc(reinterpret_cast<void**>(&lp->coff_all), sizeof(int32_t) * (size_t)coff_words) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->hist), sizeof(int) * (size_t)n_ent) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->cursor), sizeof(int) * (size_t)n_ent) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->cbase), sizeof(int) * (size_t)n_ent) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->btot), sizeof(int) * (GM_PG_MAXB + 1)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&lp->gsum), sizeof(uint32_t) * (size_t)n_ent) != hipSuccess ||
            hipMalloc(&lp->ctr, sizeof(LpCounters)) != hipSuccess ||
            (lp->ctr_host == nullptr && hipHostMalloc(&lp->ctr_host, sizeof(LpCounters)) != hipSuccess)) {
            lp_free_pool(lp);
            return gm_fail(GM_ERR_ALLOC, "hipMalloc(K2P grouping tables) failed");
        }
        std::memset(lp->ctr_host, 0, sizeof(LpCounters));
        if (hipMemsetAsync(lp->hist, 0, sizeof(int) * (size_t)n_ent, st) != hipSuccess ||
            hipMemsetAsync(lp->btot, 0, sizeof(int) * (GM_PG_MAXB + 1), st) != hipSuccess ||      // (the offsets kernel's ticket)
            hipMemcpyAsync(lp->gsum, gsum.data(), sizeof(uint32_t) * gsum.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            lp_free_pool(lp);
            return gm_fail(GM_ERR_DEVICE, "K2P grouping tables upload failed");
        }
    }
    return GM_OK;
}

// group the candidates (planes q) by every block's value and lay out the chunk table -- on the device, no host round trip:
// the scan is launched over an upper bound of the chunks and reads the table's length from ctr->n_chunks
int gm_lp_candidates(gm_index* idx, const uint2* q, int64_t n_q, hipStream_t st) {
    gm_levpart* lp = gm_lp_state(idx);
    if (lp->r < 1) return gm_fail(GM_ERR_ARG, "gm_lp_candidates: no copies prepared");
    if (n_q >= (int64_t)1 << 31) return gm_fail(GM_ERR_ARG, "gm_lp_candidates: too many candidates");
    int rc;
    PgKeyArg K{};
    K.B = lp->n_blocks;
    int64_t grid = 0, coff_at = 0;
    for (int j = 0; j <= lp->n_blocks; ++j) K.ebase[j] = lp->ebase[j];
    for (int j = 0; j < lp->n_blocks; ++j) {
        if ((rc = lp_ensure(&lp->cperm[j], &lp->cperm_cap[j], n_q, "candidate order"))) return rc;
        K.bstart[j] = lp->bstart[j];
        K.blen[j] = lp->blen[j];
        K.coff[j] = lp->coff_all + coff_at;
        K.cperm[j] = lp->cperm[j];
        coff_at += (1 << (2 * lp->blen[j])) + 1;
        grid += (n_q + 63) / 64 + (1 << (2 * lp->blen[j]));
    }
    if ((rc = lp_ensure(&lp->chunks, &lp->chunks_cap, grid, "K2P chunk table"))) return rc;
    PgSubArg nsub{};
    for (int j = 0; j < lp->n_blocks; ++j) nsub.n[j] = 1;
    LpCounters* ctr = static_cast<LpCounters*>(lp->ctr);
    if ((rc = gm_group_launch(q, n_q, K, nsub, lp->hist, lp->cursor, lp->cbase, lp->btot, reinterpret_cast<int*>(ctr),
                              (int)(sizeof(LpCounters) / sizeof(int)), ctr->cstart, &ctr->n_chunks, idx->device, st)))
        return rc;
    hipLaunchKernelGGL(k_lp_fill_chunks, lp_grid(lp->ebase[lp->n_blocks]), dim3(256), 0, st, K, lp->cbase, lp->gsum, lp->chunks, ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(lp->ctr_host, lp->ctr, sizeof(LpCounters), hipMemcpyDeviceToHost, st));
    lp->n_chunks = grid;
    return GM_OK;
}

// the K2P launch over the chunk table: raw lists per block in lp->keys ([n_blocks][n_q][kcap])
int gm_lp_scan(gm_index* idx, const uint2* q, int64_t n_q, int k, int kcap, const int32_t* seed_d, const int32_t* seed_c, int seed_k,
               hipStream_t st, const uint32_t** keys_out, int* n_lists) {
    gm_levpart* lp = gm_lp_state(idx);
    int rc;
    if ((rc = lp_ensure(&lp->keys, &lp->keys_cap, (int64_t)lp->n_blocks * n_q * kcap, "K2P block lists"))) return rc;
    // (a candidate is in exactly one chunk per block: every list is written; the fill is a safeguard)
    HIP_TRY(hipMemsetAsync(lp->keys, 0xff, sizeof(uint32_t) * (size_t)((int64_t)lp->n_blocks * n_q * kcap), st));
    gm_lp_copies P{};
    for (int c = 0; c < lp->n_copies; ++c) {
        P.table[c] = lp->table[c];
        P.ord[c] = lp->ord[c];
        P.bdesc[c] = lp->bdesc[c];
    }
    for (int j = 0; j < lp->n_blocks; ++j) P.cperm[j] = lp->cperm[j];
    for (int j = 0; j <= lp->n_blocks; ++j) P.first_copy[j] = lp->first_copy[j];
    P.n_blocks = lp->n_blocks;
    *keys_out = lp->keys;
    *n_lists = lp->n_blocks;
    if (lp->n_chunks == 0) return GM_OK;
    const int* n_chunks_dev = &static_cast<const LpCounters*>(lp->ctr)->n_chunks;
    switch (idx->L) {
#define GM_CASE(LEN) \
    case LEN: return gm_launch_lvp_L##LEN(&P, lp->chunks, lp->n_chunks, q, n_q, k, kcap, seed_d, seed_c, seed_k, lp->keys, lp->r, n_chunks_dev, st);
        GM_FOR_EACH_LEN(GM_CASE)
#undef GM_CASE
        default: return gm_fail(GM_ERR_ARG, "K2P: guidelen out of range");
    }
}

// the raw per-block lists of the last gm_lp_scan folded into the top k of every candidate (a target found under several
// blocks / displacements is taken once)
int gm_lp_merge(gm_index* idx, int kcap, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st) {
    gm_levpart* lp = gm_lp_state(idx);
    if (lp->r < 1 || !lp->keys) return gm_fail(GM_ERR_ARG, "gm_lp_merge: no scan to merge");
    constexpr int SP = 4;
    const dim3 grid((unsigned)((n_q * SP + 255) / 256)), block(256);
    if (kcap == 10)
        hipLaunchKernelGGL((gm::k_merge_lists_unique<10, SP>), grid, block, 0, st, lp->keys, lp->n_blocks, n_q, k, ids, dists, counts);
    else
        hipLaunchKernelGGL((gm::k_merge_lists_unique<20, SP>), grid, block, 0, st, lp->keys, lp->n_blocks, n_q, k, ids, dists, counts);
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
