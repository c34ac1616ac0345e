// gm_lookup.hip.h -- bit-sliced Hamming k-NN with per-lane LDS table look-ups (K1L) for gfx950.
//
// Measured on MI355X (tools/valu_pattern.hip): a VALU op with an SGPR source issues at half rate, so
// in K1v (gm_bitslice.hip.h) the 2L ops that form the mismatch words from scalar-loaded pool words
// cost as much as the whole adder tree.  K1L removes them: the pool is stored ONE-HOT and bit-sliced,
//
//   table[G][p][b][h]  (uint32)   G = group of 64 pool ids, p = base position, b = base code A0 C1 G2 T3,
//                                 h = half (ids 64G+32h .. +31); bit j set <=> that target has base b at p
//
// a wave stages tiles of TG groups into its private LDS buffer (coalesced 16-byte loads, double
// buffered), and every lane (= one candidate) fetches, for position p, the 8 bytes at
// [p][its own base at p][0..1] with ONE ds_read_b64 -- the MATCH flags of 64 targets, no VALU op.
// Lanes read only 4 distinct addresses per instruction (8 banks): conflict-free, 256 B/clk.
// Then per half: carry-save adder tree over the L match words -> bit-sliced match count, ripple
// comparator `matches >= L + 1 - thr` (<=> distance < thr) -> hit word; same slow path / k-lists as K1v.
// Per 64 pairs per lane: 20 ds_read_b64 + 2 x (36 + 5) v_bitop3 + 1 v_cmp at L=20 (1.3 VALU ops/pair).
#pragma once
#ifndef GM_LK_W10
#define GM_LK_W10 6     /* waves per SIMD asked for with a 5/10-key register list at L <= 20 */
#endif
#include "gm_bitslice.hip.h"
#include <type_traits>

namespace gm {

constexpr int LK_GROUP = 64;   // pool ids per table group
constexpr int LK_TG = 4;       // groups per LDS tile
__host__ __device__ constexpr int lk_group_words(int L) { return 8 * L; }           // [L][4][2]
__host__ __device__ constexpr int lk_tile_bytes(int L) { return LK_TG * 32 * L; }

constexpr int TT_A_OR_NOTB = 0xF3;   // a | ~b                      (c ignored)
constexpr int TT_GE = 0xB2;          // (a & ~b) | (~(a ^ b) & c)   a = count bit, b = bound bit, c = ge so far
static_assert(((0xF0 | ~0xCC) & 0xFF) == TT_A_OR_NOTB, "a | ~b");
static_assert((((0xF0 & ~0xCC) | (~(0xF0 ^ 0xCC) & 0xAA)) & 0xFF) == TT_GE, "ge stage");

// planes -> one-hot bit-sliced table of L positions.  One thread per output word; ids past n_db replicate
// the last.  skip_len > 0: the table leaves out the sequence positions [skip_start, skip_start + skip_len)
// (K1P: positions on which every pair of a bucket agrees anyway), L = positions kept.
static __global__ void k_to_onehot(const uint2* __restrict__ planes, int64_t n_db, uint32_t* __restrict__ table,
                            int64_t n_groups, int L, int skip_start = 0, int skip_len = 0) {
    const int gw = lk_group_words(L);
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_groups * gw) return;
    int64_t G = t / gw;
    int w = (int)(t % gw);
    int p = w >> 3, b = (w >> 1) & 3, h = w & 1;
    if (p >= skip_start) p += skip_len;
    uint32_t out = 0;
    for (int j = 0; j < 32; ++j) {
        int64_t id = G * LK_GROUP + 32 * h + j;
        if (id >= n_db) id = n_db - 1;
        uint2 e = planes[id];
        uint32_t code = ((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1);
        out |= (uint32_t)(code == (uint32_t)b) << j;
    }
    table[t] = out;
}

// hit word of one half group: bit j set <=> matches(j) >= bound  (bound bits broadcast in Tb)
template <int L>
__device__ __forceinline__ uint32_t lk_eval_half(const uint32_t (&m)[L], const uint32_t (&Tb)[5],
                                                 uint32_t (&c)[5]) {
    bitslice_count<L>(m, c);
    uint32_t ge = GM_BITOP3(c[0], Tb[0], Tb[0], TT_A_OR_NOTB);
#pragma unroll
    for (int i = 1; i < 5; ++i) ge = GM_BITOP3(c[i], Tb[i], ge, TT_GE);
    return ge;
}

// ---------------------------------------------------------------------------------------------
// k-list policies.  Keys are (dist << 27 | id); smaller key = better neighbour.
//   RegList<KCAP>: the list lives in KCAP VGPRs, always sorted; an insert is a min/max ripple of
//     2*KCAP ops with no LDS round trips and no divergent loop.  Runtime k <= KCAP: the first KCAP-k
//     slots hold the sentinel 0 (keys are stored +1, so 0 beats every real key and never moves).
//   LdsList: sorted list in LDS (any k <= 64), see list_insert().
// Both expose: threshold(L) (distance the next neighbour must beat; L+1 while the list is not full),
// insert(key), and store().
// ---------------------------------------------------------------------------------------------
template <int KCAP>
struct RegList {
    static constexpr uint32_t EMPTY = 0xffffffffu;
    uint32_t v[KCAP];
    __device__ __forceinline__ void init(int k, uint32_t*, int) {
#pragma unroll
        for (int s = 0; s < KCAP; ++s) v[s] = s < KCAP - k ? 0u : EMPTY;
    }
    __device__ __forceinline__ int threshold(int L) const {
        return v[KCAP - 1] == EMPTY ? L + 1 : (int)((v[KCAP - 1] - 1u) >> ID_BITS);
    }
    __device__ __forceinline__ void insert(uint32_t key) {
        uint32_t x = key + 1u;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            uint32_t lo = v[s] < x ? v[s] : x;
            x = v[s] < x ? x : v[s];
            v[s] = lo;
        }
    }
    // x = key + 1 as the list stores it; EMPTY (no key) falls through
    __device__ __forceinline__ void insert_raw(uint32_t x) {
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            uint32_t lo = v[s] < x ? v[s] : x;
            x = v[s] < x ? x : v[s];
            v[s] = lo;
        }
    }
    // insert unless the key is already in the list (merging lists that may share entries)
    __device__ __forceinline__ void insert_unique(uint32_t key) {
        bool dup = false;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) dup = dup || v[s] == key + 1u;
        if (!dup) insert(key);
    }
    __device__ __forceinline__ void store(int k, int64_t qi, int32_t* __restrict__ ids,
                                          int32_t* __restrict__ dists, int32_t* __restrict__ counts,
                                          bool nearest) const {
        if (nearest) {
            dists[qi] = v[KCAP - 1] == EMPTY ? -1 : (int32_t)((v[KCAP - 1] - 1u) >> ID_BITS);
            return;
        }
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            const int j = s - (KCAP - k);
            if (j >= 0) {
                const bool ok = v[s] != EMPTY;
                const uint32_t key = v[s] - 1u;
                ids[qi * k + j] = ok ? (int32_t)(key & ID_MASK) : -1;
                dists[qi * k + j] = ok ? (int32_t)(key >> ID_BITS) : -1;
                cnt += ok ? 1 : 0;
            }
        }
        counts[qi] = cnt;
    }
};

struct LdsList {
    uint32_t* lp;
    int k, cnt;
    __device__ __forceinline__ void init(int k_, uint32_t* lists, int lane) {
        lp = lists + lane;
        k = k_;
        cnt = 0;
    }
    __device__ __forceinline__ int threshold(int L) const {
        return cnt == k ? (int)(lp[(k - 1) * WAVE] >> ID_BITS) : L + 1;
    }
    __device__ __forceinline__ void insert(uint32_t key) { list_insert(lp, WAVE, k, cnt, key); }
    __device__ __forceinline__ void store(int, int64_t qi, int32_t* __restrict__ ids, int32_t* __restrict__ dists,
                                          int32_t* __restrict__ counts, bool nearest) const {
        if (nearest) {
            dists[qi] = cnt > 0 ? (int32_t)(lp[0] >> ID_BITS) : -1;
            return;
        }
        counts[qi] = cnt;
        for (int s = 0; s < k; ++s) {
            const bool ok = s < cnt;
            const uint32_t key = ok ? lp[s * WAVE] : 0u;
            ids[qi * k + s] = ok ? (int32_t)(key & ID_MASK) : -1;
            dists[qi * k + s] = ok ? (int32_t)(key >> ID_BITS) : -1;
        }
    }
};

__device__ __forceinline__ void lk_set_bound(uint32_t (&Tb)[5], int L, int thr) {
    const uint32_t bound = (uint32_t)(L + 1 - thr);
#pragma unroll
    for (int i = 0; i < 5; ++i) Tb[i] = 0u - ((bound >> i) & 1u);
}

// slow path of one half group: walk this lane's hit bits in ascending id order
// (id_lo: K1P scans a re-ordered copy of the pool and keeps only positions id_lo <= position < n_db = the
// bucket's own range, where positions ascend with the pool ids; it translates positions to ids at the end)
template <int L, class List>
__device__ __forceinline__ void lk_take_hits(uint32_t hits, const uint32_t (&c)[5], int id0, int n_db,
                                             List& list, int& thr, int thr_seed, uint32_t (&Tb)[5],
                                             int id_lo = 0) {
    while (hits != 0u) {
        int j = __builtin_ctz(hits);
        hits &= hits - 1u;
        int matches = (int)(((c[0] >> j) & 1u) | (((c[1] >> j) & 1u) << 1) | (((c[2] >> j) & 1u) << 2) |
                            (((c[3] >> j) & 1u) << 3) | (((c[4] >> j) & 1u) << 4));
        int d = L - matches;
        int id = id0 + j;
        if (d < thr && id < n_db && id >= id_lo) {
            list.insert(((uint32_t)d << ID_BITS) | (uint32_t)id);
            int nt = list.threshold(L);
            nt = nt < thr_seed ? nt : thr_seed;      // a seeded bound never loosens while the list fills
            if (nt != thr) {
                thr = nt;
                lk_set_bound(Tb, L, thr);
            }
        }
    }
}

// stage one tile with the LDS-DMA path (global_load_lds_dwordx4: 16 B per lane straight into LDS, no
// VGPR staging): round r copies units [64r, 64r+64) of 16 bytes; completion is tracked by vmcnt
template <int L>
__device__ __forceinline__ void lk_stage(const uint32_t* __restrict__ table, int64_t tile, uint32_t* dst,
                                         int lane) {
    constexpr int UNITS = lk_tile_bytes(L) / 16;
    constexpr int ROUNDS = (UNITS + WAVE - 1) / WAVE;
    const uint32_t* src = table + tile * (lk_tile_bytes(L) / 4);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = r * WAVE + lane;
        if (u < UNITS) __builtin_amdgcn_global_load_lds(src + u * 4, dst + r * WAVE * 4, 16, 0, 0);
    }
}
__device__ __forceinline__ void lk_stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// the 2L match words of one 64-target group, as fetched by this lane
template <int L>
struct LkGroupRegs {
    uint32_t m0[L], m1[L];
};
// ptr[p] = this lane's LDS address of (position p, its base at p) in group 0 of buffer 0; WOFF = constant
// word offset of the group to read (folds into the ds_read_b64 immediate offset field)
template <int L, int WOFF>
__device__ __forceinline__ void lk_read_group(LkGroupRegs<L>& r, const uint32_t* const (&ptr)[L]) {
#pragma unroll
    for (int p = 0; p < L; ++p) {
        uint2 w = *reinterpret_cast<const uint2*>(ptr[p] + WOFF);    // ds_read_b64, per-lane address
        r.m0[p] = w.x;
        r.m1[p] = w.y;
    }
}
template <int L, class List>
__device__ __forceinline__ void lk_process_group(const LkGroupRegs<L>& r, int id0, int n_db, List& list, int& thr,
                                                 int thr_seed, uint32_t (&Tb)[5], int id_lo = 0) {
    uint32_t c0[5], c1[5];
    uint32_t h0 = lk_eval_half<L>(r.m0, Tb, c0);
    uint32_t h1 = lk_eval_half<L>(r.m1, Tb, c1);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64((h0 | h1) != 0u) != 0ull, 0)) {
        lk_take_hits<L, List>(h0, c0, id0, n_db, list, thr, thr_seed, Tb, id_lo);
        lk_take_hits<L, List>(h1, c1, id0 + 32, n_db, list, thr, thr_seed, Tb, id_lo);
    }
}

// K1L.  grid = (ceil(n_q / 64), segments) single-wave workgroups.  Dynamic LDS: 2 tiles (+ k * 64 * 4 B of
// lists when KCAP == 0).  KCAP > 0: register-resident list of capacity KCAP >= k; KCAP == 0: LDS list
// (k <= 64, single segment only).  NEAREST (k == 1): write only the nearest distance.
//
// Segmented two-phase scan (out_keys != nullptr): the pool is cut into segments of `tiles_per_seg` tiles;
// workgroup (x, y) scans segment seg_base + y for candidate chunk x and writes its raw k-list to
// out_keys[segment][candidate][KCAP].  Phase A (SEEDED = false) scans segment 0 from scratch; phase B
// (SEEDED = true) scans the other segments with the bound seeded from segment 0's list (k-th key of a
// subset of the pool bounds the k-th key of the whole pool; later ids lose ties, so the strict test
// d < seed is exact).  k_merge_lists folds the segment lists.  This gives chunks x segments equal work
// items instead of `chunks` long ones: no half-empty last round, and small candidate sets fill the chip.
//
// register budget: for L <= 20 with a list of <= 10 keys the allocator lands 2 VGPRs above the 80 that allow
// 6 waves/SIMD; asking for 6 costs no spills there (checked with -Rpass-analysis=kernel-resource-usage)
__host__ __device__ constexpr int lk_min_waves(int L, int KCAP) { return (L <= 20 && KCAP == 1) ? 8 : ((L <= 20 && KCAP > 0 && KCAP <= 10) ? GM_LK_W10 : 1); }

template <int L, int KCAP, bool NEAREST, bool SEEDED>
__global__ __launch_bounds__(WAVE, lk_min_waves(L, KCAP)) void k_hamming_knn_lk(
    const uint32_t* __restrict__ table, int n_db, int n_tiles, int tiles_per_seg, int seg_base,
    const uint2* __restrict__ q, int64_t n_q, int k, const uint32_t* __restrict__ seed_keys,
    uint32_t* __restrict__ out_keys, int32_t* __restrict__ ids, int32_t* __restrict__ dists,
    int32_t* __restrict__ counts, const int* __restrict__ n_dev, uint32_t seed_all, const int32_t* __restrict__ q_rows) {
    extern __shared__ __align__(16) uint32_t smem[];
    // (speculative follow-ups of K1P: the grid was sized for an upper bound of the rows; the count is on the device)
    const int64_t n_out = n_q;      // row stride of the segment lists: the host's number
    if (n_dev != nullptr) {
        if (*n_dev < n_q) n_q = *n_dev;
        if ((int64_t)blockIdx.x * WAVE >= n_q) return;
    }
    constexpr int TILE_W = lk_tile_bytes(L) / 4;         // words per tile
    constexpr int GW = lk_group_words(L);
    uint32_t* tiles = smem;                              // [2][TILE_W]
    const int lane = threadIdx.x;

    const int64_t qi = (int64_t)blockIdx.x * WAVE + lane;
    const bool valid = qi < n_q;
    // (q_rows != nullptr: candidate qi is row q_rows[qi] of q -- K1P's re-scans read the rows the merge listed, flag bit
    // masked, instead of a compact copy that a kernel of its own would have to gather first)
    const uint2 qv = valid ? q[q_rows != nullptr ? (int64_t)((uint32_t)q_rows[qi] & 0x7fffffffu) : qi] : make_uint2(0u, 0u);

    // LDS address of this lane's (position, base) pair inside group 0 of buffer 0
    const uint32_t* ptr[L];
#pragma unroll
    for (int p = 0; p < L; ++p) {
        uint32_t code = ((qv.x >> p) & 1u) | (((qv.y >> p) & 1u) << 1);
        ptr[p] = tiles + p * 8 + (int)code * 2;
    }
    using List = typename std::conditional<(KCAP > 0), RegList<(KCAP > 0 ? KCAP : 1)>, LdsList>::type;
    List list;
    list.init(k, smem + 2 * TILE_W, lane);
    int thr_seed = L + 1;              // L + 1 accepts every distance until the list is full
    if constexpr (SEEDED && KCAP > 0) {
        if (valid) {      // (seed_keys == nullptr: the same bound `seed_all` for every candidate -- K1P's guessed re-scans)
            const uint32_t top = seed_keys != nullptr ? seed_keys[qi * KCAP + (KCAP - 1)] : seed_all;
            if (top != 0xffffffffu) thr_seed = (int)((top - 1u) >> ID_BITS);
        }
    }
    int thr = valid ? thr_seed : 0;
    uint32_t Tb[5];
    lk_set_bound(Tb, L, thr);

    const int seg = seg_base + (int)blockIdx.y;
    const int t0 = seg * tiles_per_seg;                  // tiles_per_seg and n_tiles are even
    const int t1 = t0 + tiles_per_seg < n_tiles ? t0 + tiles_per_seg : n_tiles;

    // single-wave workgroup: LDS operations of one wave execute in issue order, no barrier needed;
    // tile t+1 is DMA'd into the other buffer while tile t is evaluated.  The tile loop is unrolled by
    // two so that the buffer offset is a compile-time constant.
    if (t0 < t1) lk_stage<L>(table, t0, tiles, lane);
    for (int t = t0; t < t1; t += 2) {
        lk_stage_wait();                                               // tile t has landed in buffer 0
        lk_stage<L>(table, t + 1, tiles + TILE_W, lane);
#pragma unroll
        for (int gl = 0; gl < LK_TG; ++gl) {
            LkGroupRegs<L> r;
            if (gl == 0) lk_read_group<L, 0 * GW>(r, ptr);
            if (gl == 1) lk_read_group<L, 1 * GW>(r, ptr);
            if (gl == 2) lk_read_group<L, 2 * GW>(r, ptr);
            if (gl == 3) lk_read_group<L, 3 * GW>(r, ptr);
            lk_process_group<L, List>(r, (t * LK_TG + gl) * LK_GROUP, n_db, list, thr, thr_seed, Tb);
        }
        lk_stage_wait();                                               // tile t+1 has landed in buffer 1
        if (t + 2 < t1) lk_stage<L>(table, t + 2, tiles, lane);
#pragma unroll
        for (int gl = 0; gl < LK_TG; ++gl) {
            LkGroupRegs<L> r;
            if (gl == 0) lk_read_group<L, TILE_W + 0 * GW>(r, ptr);
            if (gl == 1) lk_read_group<L, TILE_W + 1 * GW>(r, ptr);
            if (gl == 2) lk_read_group<L, TILE_W + 2 * GW>(r, ptr);
            if (gl == 3) lk_read_group<L, TILE_W + 3 * GW>(r, ptr);
            lk_process_group<L, List>(r, ((t + 1) * LK_TG + gl) * LK_GROUP, n_db, list, thr, thr_seed, Tb);
        }
    }

    if (valid) {
        if constexpr (KCAP > 0) {
            if (out_keys != nullptr) {
                uint32_t* out = out_keys + ((int64_t)seg * n_out + qi) * KCAP;
#pragma unroll
                for (int s = 0; s < KCAP; ++s) out[s] = list.v[s];
                return;
            }
        }
        list.store(k, qi, ids, dists, counts, NEAREST);
    }
}

// fold per-segment raw RegList<KCAP> lists [n_seg][n_q][KCAP] into the final top-k.  One thread per candidate.
template <int KCAP>
__global__ void k_merge_lists(const uint32_t* __restrict__ keys, int n_seg, int64_t n_q, int k,
                              int32_t* __restrict__ ids, int32_t* __restrict__ dists,
                              int32_t* __restrict__ counts, int nearest, const int* __restrict__ n_dev = nullptr) {
    const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q || (n_dev != nullptr && qi >= *n_dev)) return;      // (n_q stays the row stride of the lists)
    RegList<KCAP> list;
    list.init(k, nullptr, 0);
    for (int s = 0; s < n_seg; ++s) {
        const uint32_t* in = keys + ((int64_t)s * n_q + qi) * KCAP;
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            const uint32_t v = in[i];
            if (v != 0u && v != RegList<KCAP>::EMPTY) list.insert(v - 1u);
        }
    }
    list.store(k, qi, ids, dists, counts, nearest != 0);
}

// ---------------------------------------------------------------------------------------------
// Folding raw k-lists that share no entry (segment lists of disjoint pool segments) without sorted inserts: a raw list
// is ascending between a run of 0 (unused leading slots) and a run of EMPTY, i.e. bitonic once 0 reads as EMPTY; a
// bitonic merger (log2 N stages of N / 2 min / max pairs in registers) sorts it, and two ascending lists a, b fold as
// c[i] = min(a[i], b[N - 1 - i]) -- the N smallest of both, bitonic again -- + one more merger.  N = KCAP rounded up to
// a power of two; 16 keys: 16 + 64 VALU ops per fold against ~30 per sorted insert x 10 keys.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int fold_width(int kcap) { return kcap <= 1 ? 1 : (kcap <= 2 ? 2 : (kcap <= 4 ? 4 : (kcap <= 8 ? 8 : (kcap <= 16 ? 16 : 32)))); }

template <int N>
__device__ __forceinline__ void fold_sort(uint32_t (&v)[N]) {      // bitonic (any rotation) -> ascending
#pragma unroll
    for (int j = N / 2; j >= 1; j >>= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if ((i & j) == 0) {
                const uint32_t lo = v[i] < v[i | j] ? v[i] : v[i | j];
                const uint32_t hi = v[i] < v[i | j] ? v[i | j] : v[i];
                v[i] = lo;
                v[i | j] = hi;
            }
        }
    }
}
// raw list -> v (0, and every raw value below `min_raw`, reads as EMPTY), ascending
template <int KCAP, int N>
__device__ __forceinline__ void fold_load(uint32_t (&v)[N], const uint32_t* __restrict__ in, uint32_t min_raw) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) {
        const uint32_t x = in[i];
        v[i] = x < min_raw ? 0xffffffffu : x;
    }
    fold_sort<N>(v);
}
// acc, t ascending -> acc = the N smallest of both, ascending
template <int N>
__device__ __forceinline__ void fold_two(uint32_t (&acc)[N], const uint32_t (&t)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = acc[i] < t[N - 1 - i] ? acc[i] : t[N - 1 - i];
    fold_sort<N>(acc);
}
// the lists s = sub, sub + SPLIT, ... of row `row` (stride n_stride rows per list) folded by lane `sub` of SPLIT adjacent
// lanes, then across the lanes (shuffle tree): lane sub == 0 ends with the N smallest raw values of all lists, ascending
template <int KCAP, int SPLIT, int N>
__device__ __forceinline__ void fold_lists(uint32_t (&acc)[N], const uint32_t* __restrict__ keys, int n_lists, int64_t n_stride,
                                           int64_t row, int sub, uint32_t min_raw) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0xffffffffu;
    for (int s = sub; s < n_lists; s += SPLIT) {
        uint32_t t[N];
        fold_load<KCAP, N>(t, keys + ((int64_t)s * n_stride + row) * KCAP, min_raw);
        fold_two<N>(acc, t);
    }
#pragma unroll
    for (int off = SPLIT / 2; off >= 1; off >>= 1) {
        uint32_t t[N];
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = __shfl_down(acc[i], off, SPLIT);
        fold_two<N>(acc, t);
    }
}

// k_merge_lists for FEW candidates and many segments (the re-scan of K1P's incomplete rows, controls): with one
// thread per candidate the kernel is a handful of waves walking n_seg lists serially.  Here SPLIT adjacent lanes
// share a candidate, lane s folds the segments s, s + SPLIT, ..., and a shuffle tree combines the partial lists
// (fold_lists: bitonic folds, the segments share no entry).
template <int KCAP, int SPLIT>
__global__ void k_merge_lists_split(const uint32_t* __restrict__ keys, int n_seg, int64_t n_q, int k,
                                    int32_t* __restrict__ ids, int32_t* __restrict__ dists,
                                    int32_t* __restrict__ counts, int nearest, const int* __restrict__ n_dev = nullptr) {
    constexpr int N = fold_width(KCAP);
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(t % SPLIT);
    const int64_t qi_raw = t / SPLIT;
    const bool valid = qi_raw < n_q && (n_dev == nullptr || qi_raw < *n_dev);      // (n_q stays the row stride of the lists)
    if (!valid) return;      // (the SPLIT lanes of a row leave together: the shuffles below stay inside a row)
    const int64_t qi = qi_raw;
    uint32_t acc[N];
    fold_lists<KCAP, SPLIT, N>(acc, keys, n_seg, n_q, qi, sub, 1u);
    if (sub != 0) return;
    if (nearest != 0) {
        dists[qi] = acc[0] == 0xffffffffu ? -1 : (int32_t)((acc[0] - 1u) >> ID_BITS);
        return;
    }
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < KCAP; ++j) {
        if (j < k) {
            const bool ok = acc[j] != 0xffffffffu;
            const uint32_t key = acc[j] - 1u;
            ids[qi * k + j] = ok ? (int32_t)(key & ID_MASK) : -1;
            dists[qi * k + j] = ok ? (int32_t)(key >> ID_BITS) : -1;
            cnt += ok ? 1 : 0;
        }
    }
    counts[qi] = cnt;
}

// fold the per-block raw lists [n_lists][n_q][KCAP] of K2P into the top k: a target found under several blocks /
// displacements is taken once.  SPLIT lanes per candidate.
template <int KCAP, int SPLIT>
__global__ void k_merge_lists_unique(const uint32_t* __restrict__ keys, int n_lists, int64_t n_q, int k,
                                     int32_t* __restrict__ ids, int32_t* __restrict__ dists, int32_t* __restrict__ counts) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(t % SPLIT);
    const int64_t qi_raw = t / SPLIT;
    const bool valid = qi_raw < n_q;
    const int64_t qi = valid ? qi_raw : n_q - 1;
    RegList<KCAP> list;
    list.init(k, nullptr, 0);
    for (int s = sub; s < n_lists; s += SPLIT) {
        const uint32_t* in = keys + ((int64_t)s * n_q + qi) * KCAP;
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            const uint32_t v = in[i];
            if (v != 0u && v < list.v[KCAP - 1]) list.insert_unique(v - 1u);
        }
    }
#pragma unroll
    for (int off = SPLIT / 2; off >= 1; off >>= 1) {
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            const uint32_t v = __shfl_down(list.v[i], off, SPLIT);
            if (sub < off && v != 0u && v < list.v[KCAP - 1]) list.insert_unique(v - 1u);
        }
    }
    if (sub == 0 && valid) list.store(k, qi, ids, dists, counts, false);
}

}  // namespace gm
