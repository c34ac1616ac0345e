// gm_leven_bs.hip.h -- bit-sliced Levenshtein k-NN scan for gfx950 (K2B).
//
// K2L (gm_leven.hip.h) runs Myers' bit-vector algorithm per (candidate, target) pair: bits = rows of the
// DP column, ~12 VALU ops per text position per pair.  K2B turns the problem by 90 degrees, like K1L did
// for Hamming: bits = 32 pool TARGETS, lane = one candidate, and the unit-cost DP is evaluated cell by
// cell with Myers/Hyyro's delta logic on 32-target words:
//
//   cell (i, j): i = candidate position (row), j = target position (column), deltas in {-1, 0, +1}
//   encoded as (P, M) bit pairs;  Eq = "target j-th base == candidate i-th base" for 32 targets = ONE
//   ds_read_b32 from the one-hot table K1L already uses (table[G][j][base_i][half]), per-lane address.
//       D0  = Eq | Mv | Mh                    (diagonal stays)          v_bitop3
//       Ph' = Mv | ~(D0 | Pv);  Mh' = Pv & D0  (horizontal delta out)   v_bitop3, v_and
//       Pv' = Mh | ~(D0 | Ph);  Mv' = Ph & D0  (vertical delta out)     v_bitop3, v_and
//   5 full-rate ops per cell for 32 pairs: 5 L^2 / 32 ops per pair (83 at L=23) against ~290 for K2L.
//   Vertical deltas live in 2L VGPRs, the horizontal delta runs down the column in 2 VGPRs.
//   distance = L + #P - #M over the last column's vertical deltas; s = #M - #P = L - distance >= 0 is
//   formed bit-sliced (two carry-save trees + a 5-bit ripple subtract) and then goes through exactly
//   the K1L threshold comparator / hit walk / register k-lists (s plays the role of "matches").
// Same tiles (LDS-DMA, double buffered), Hamming-seeded bounds, chunks x segments grid and merge as K2L.
#pragma once
#ifndef GM_LVB_WAVES
#define GM_LVB_WAVES 3     /* waves per SIMD asked of the allocator for the full-matrix kernel (~150 VGPRs) */
#endif
#ifndef GM_LVB_BAND_WAVES
#define GM_LVB_BAND_WAVES 4     /* ... and for the band-only kernel (~110 VGPRs with both halves of a group in flight) */
#endif
#include "gm_leven.hip.h"
#include "gm_pg_types.h"

#include <utility>

namespace gm {

constexpr int TT_OR3 = (0xF0 | 0xCC | 0xAA) & 0xFF;                               // a | b | c
constexpr int TT_SUB_DIFF = (0xF0 ^ 0xCC ^ 0xAA) & 0xFF;                          // a ^ b ^ borrow
constexpr int TT_SUB_BORROW = ((~0xF0 & 0xCC) | (~(0xF0 ^ 0xCC) & 0xAA)) & 0xFF;  // (~a & b) | (~(a ^ b) & borrow)

// The DP columns are written with inline-asm LDS reads on explicit per-lane byte addresses.  Left to
// itself hipcc re-derives every address as base + loop offset (a half-rate SGPR-operand add per cell),
// hoists the loads of several columns at once and ends at 256 VGPRs or ~1 KB of scratch per lane.  With
// asm reads the address registers are bumped in place (also asm, so no copies), one column's L words
// are in flight at a time, and the wait is ours.
template <int BOFF>
__device__ __forceinline__ uint32_t lvb_read(uint32_t adr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(adr), "n"(BOFF));
    return v;
}
// wait for every outstanding LDS read; the "+v" operands pin the consumers of those words behind the wait
__device__ __forceinline__ void lvb_wait4(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// one DP column for 32 targets: BOFF = constant byte offset (column within the unrolled block + half)
template <int L, int BOFF>
__device__ __forceinline__ void lvb_column(const uint32_t (&adr)[L], uint32_t (&Pv)[L], uint32_t (&Mv)[L]) {
    constexpr int LP = (L + 3) / 4 * 4;      // rows padded to whole groups of four (dummies are never read)
    uint32_t eq[LP];
#pragma unroll
    for (int i = 0; i < LP; ++i) eq[i] = i < L ? lvb_read<BOFF>(adr[i]) : 0u;
#pragma unroll
    for (int g = 0; g < LP / 4; ++g) lvb_wait4(eq[4 * g], eq[4 * g + 1], eq[4 * g + 2], eq[4 * g + 3]);
    uint32_t Ph = 0xffffffffu, Mh = 0u;      // D(0, j) = j
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint32_t D0 = GM_BITOP3(eq[i], Mv[i], Mh, TT_OR3);
        const uint32_t Pho = GM_BITOP3(Mv[i], D0, Pv[i], TT_A_OR_NOR_BC);
        const uint32_t Mho = Pv[i] & D0;
        Pv[i] = GM_BITOP3(Mh, D0, Ph, TT_A_OR_NOR_BC);
        Mv[i] = Ph & D0;
        Ph = Pho;
        Mh = Mho;
    }
}

// DP of one half group (32 targets) against this lane's candidate.  adr[i] = LDS byte address of
// (column 0, base of candidate position i, half 0) of the group; HB = byte offset of the half (0 / 4).
// Columns are unrolled four at a time (offsets as instruction immediates; the addresses move once per
// four columns).  On return adr[] has advanced by one group (L columns x 32 bytes) and c[0..4] holds the
// bits of s = L - distance.
template <int L, int HB>
__device__ __forceinline__ void lvb_eval_half(uint32_t (&adr)[L], uint32_t (&c)[5]) {
    uint32_t Pv[L], Mv[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        Pv[i] = 0xffffffffu;      // D(i, 0) = i
        Mv[i] = 0u;
    }
#pragma unroll 1
    for (int j = 0; j + 4 <= L; j += 4) {
        lvb_column<L, HB + 0>(adr, Pv, Mv);
        lvb_column<L, HB + 32>(adr, Pv, Mv);
        lvb_column<L, HB + 64>(adr, Pv, Mv);
        lvb_column<L, HB + 96>(adr, Pv, Mv);
#pragma unroll
        for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, 0x80, %0" : "+v"(adr[i]));
    }
    if constexpr (L % 4 >= 1) lvb_column<L, HB + 0>(adr, Pv, Mv);
    if constexpr (L % 4 >= 2) lvb_column<L, HB + 32>(adr, Pv, Mv);
    if constexpr (L % 4 >= 3) lvb_column<L, HB + 64>(adr, Pv, Mv);
    if constexpr (L % 4 != 0) {
#pragma unroll
        for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(32 * (L % 4)));
    }
    // s = #M - #P over the last column (>= 0 because the distance never exceeds L)
    uint32_t cm[5], cp[5];
    bitslice_count<L>(Mv, cm);
    bitslice_count<L>(Pv, cp);
    uint32_t borrow = 0u;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        c[b] = GM_BITOP3(cm[b], cp[b], borrow, TT_SUB_DIFF);
        borrow = GM_BITOP3(cm[b], cp[b], borrow, TT_SUB_BORROW);
    }
    // finish the count HERE: without this the compiler sinks the trees below the next half's columns
    // and keeps this half's 2L delta registers alive through them
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]));
}

// ---------------------------------------------------------------------------------------------------
// Banded evaluation.  A pair can only be a hit if its distance is below the lane's bound thr, i.e. <= t = thr - 1.
// Candidate and target have the SAME length, so an alignment uses as many insertions as deletions: a path that
// reaches diagonal |i - j| = g has spent g indels getting there and needs g more to come back to (L, L), so a path of
// cost <= t never leaves the band |i - j| <= floor(t / 2) (Ukkonen's cut-off for equal lengths -- half of the
// obvious band t).  While every lane of the wave has thr <= 2 BAND + 2 only the band's cells are evaluated.  Outside
// cells are replaced by upper bounds that cost nothing: a column starts at the top of its band with the horizontal
// delta +1 (exact in row 0, an upper bound below it), and a row enters the band from below with the vertical delta
// it was initialised with (+1).  All values computed this way are >= the true ones and exact along every path
// inside the band, so the result is the true distance when that is <= 2 BAND + 1 and something >= the true distance
// (never a false hit) otherwise.
// The last column is no longer complete, so the distance is read off the DIAGONAL instead: it grows by
// 0 or 1 per step and D0(i, i) says which, i.e. s = L - distance = popcount over i of D0(i, i): one
// carry-save tree.  Columns are fully unrolled (row ranges and read offsets are compile-time); the delta
// registers of rows above the band die and those below it are still constants, so the banded code also needs
// fewer VGPRs.  L = 23, BAND = 4: 187 of 529 cells (the round-1 band |i - j| <= 9 had 347).  Two band widths are
// instantiated: BAND for the seeded bounds and BAND - 1 for waves whose lists have tightened further.  (While a
// lane's bound is above 2 BAND + 2 -- unseeded candidates, k-th neighbours far away -- the wave takes the full
// evaluation above.)
// ---------------------------------------------------------------------------------------------------
// band half-width per guide length: the k-th Hamming distance that seeds the bounds sits near L / 3 for pools of
// 10^5 .. 10^6 random targets (thr around 9 / 10 / 11 for L = 20 / 23 / 27); 2 BAND + 2 covers it: BAND = 4, 4, 5
__host__ __device__ constexpr int lvb_band(int L) { return ((3 * L + 4) / 8) / 2; }
__host__ __device__ constexpr int lvb_band_cells(int L, int band) { return L * (2 * band + 1) - band * (band + 1); }

template <int L, int HB, int BAND, int J>
__device__ __forceinline__ void lvb_column_banded(const uint32_t (&adr)[L], uint32_t (&Pv)[L], uint32_t (&Mv)[L],
                                                  uint32_t (&dd)[L]) {
    constexpr int LO = J - BAND > 0 ? J - BAND : 0;
    constexpr int HI = J + BAND < L - 1 ? J + BAND : L - 1;
    constexpr int N = HI - LO + 1, NP = (N + 3) / 4 * 4;
    uint32_t eq[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) eq[r] = r < N ? lvb_read<HB + 32 * J>(adr[LO + r]) : 0u;
#pragma unroll
    for (int g = 0; g < NP / 4; ++g) lvb_wait4(eq[4 * g], eq[4 * g + 1], eq[4 * g + 2], eq[4 * g + 3]);
    uint32_t Ph = 0xffffffffu, Mh = 0u;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const int i = LO + r;
        const uint32_t D0 = GM_BITOP3(eq[r], Mv[i], Mh, TT_OR3);
        const uint32_t Pho = GM_BITOP3(Mv[i], D0, Pv[i], TT_A_OR_NOR_BC);
        const uint32_t Mho = Pv[i] & D0;
        Pv[i] = GM_BITOP3(Mh, D0, Ph, TT_A_OR_NOR_BC);
        Mv[i] = Ph & D0;
        Ph = Pho;
        Mh = Mho;
        if (i == J) dd[J] = D0;
    }
    // the deltas leaving the column close its dependency chain: pinning them keeps the compiler from sinking
    // this column's work below the next column's reads
    asm volatile("" : "+v"(Ph), "+v"(Mh));
}

template <int L, int HB, int BAND, int... J>
__device__ __forceinline__ void lvb_columns_banded(const uint32_t (&adr)[L], uint32_t (&Pv)[L], uint32_t (&Mv)[L],
                                                   uint32_t (&dd)[L], std::integer_sequence<int, J...>) {
    (lvb_column_banded<L, HB, BAND, J>(adr, Pv, Mv, dd), ...);
}

// banded DP of one half group; adr[] is NOT advanced (every read carries its column's offset)
template <int L, int HB, int BAND>
__device__ __forceinline__ void lvb_eval_half_banded(const uint32_t (&adr)[L], uint32_t (&c)[5]) {
    uint32_t Pv[L], Mv[L], dd[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        Pv[i] = 0xffffffffu;
        Mv[i] = 0u;
    }
    lvb_columns_banded<L, HB, BAND>(adr, Pv, Mv, dd, std::make_integer_sequence<int, L>{});
    bitslice_count<L>(dd, c);
    asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]));
}

// ---------------------------------------------------------------------------------------------------
// Both halves of a group at once (band-only kernel).  The table keeps the two halves of a (column, base) entry side
// by side, so ONE ds_read_b64 per row and column fetches the Eq words of 64 targets: half the LDS instructions of
// the per-half evaluation, and two independent delta chains for the VALU to interleave.  The diagonal words are not
// kept for a tree at the end (2 L registers for two halves) but fed into a streaming carry-save counter as they
// appear: at most two pending words per weight, a full adder whenever a third arrives.  The counter's fill levels
// are template parameters (the column index is one), so every word has a fixed register.
// ---------------------------------------------------------------------------------------------------
// Level lv of the counter receives its inputs one by one; input number T (1-based, a compile-time constant): the
// first is parked in a[lv], an even one in b[lv], an odd one >= 3 closes a full adder (sum back to a[lv], carry =
// input number (T - 1) / 2 of level lv + 1).  After T inputs a level holds T == 0 ? 0 : (T odd ? 1 : 2) words.
struct LvbDiagCount {
    uint32_t a[5], b[5];
    template <int LV, int T>
    __device__ __forceinline__ void add(uint32_t x) {
        static_assert(LV < 5, "5-bit counter");
        if constexpr (T == 1) {
            a[LV] = x;
        } else if constexpr (T % 2 == 0) {
            b[LV] = x;
        } else {
            const uint32_t s = GM_BITOP3(a[LV], b[LV], x, TT_XOR3);
            const uint32_t c = GM_BITOP3(a[LV], b[LV], x, TT_MAJ);
            a[LV] = s;
            if constexpr (LV < 4) add<LV + 1, (T - 1) / 2>(c);
        }
    }
    // bits of the total once level LV has received T inputs (and the levels above theirs)
    template <int LV, int T>
    __device__ __forceinline__ void finish(uint32_t (&c)[5]) {
        constexpr int TN = T >= 1 ? (T - 1) / 2 : 0;      // inputs level LV + 1 has received so far
        if constexpr (T == 0) {
            c[LV] = 0u;
            if constexpr (LV < 4) finish<LV + 1, 0>(c);
        } else if constexpr (T % 2 == 1) {
            c[LV] = a[LV];
            if constexpr (LV < 4) finish<LV + 1, TN>(c);
        } else {      // two words pending: half adder, its carry is one more input of the next level
            c[LV] = a[LV] ^ b[LV];
            if constexpr (LV < 4) {
                add<LV + 1, TN + 1>(a[LV] & b[LV]);
                finish<LV + 1, TN + 1>(c);
            }
        }
    }
};

template <int BOFF>
__device__ __forceinline__ uint64_t lvb_read2(uint32_t adr) {
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(adr), "n"(BOFF));
    return v;
}
__device__ __forceinline__ void lvb_wait2(uint64_t& a, uint64_t& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b));
}

template <int L, int BAND, int J>
__device__ __forceinline__ void lvb_column_banded2(const uint32_t (&adr)[L], uint32_t (&Pv0)[L], uint32_t (&Mv0)[L],
                                                   uint32_t (&Pv1)[L], uint32_t (&Mv1)[L], LvbDiagCount& d0,
                                                   LvbDiagCount& d1) {
    constexpr int LO = J - BAND > 0 ? J - BAND : 0;
    constexpr int HI = J + BAND < L - 1 ? J + BAND : L - 1;
    constexpr int N = HI - LO + 1, NP = (N + 1) / 2 * 2;
    uint64_t eq[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) eq[r] = r < N ? lvb_read2<32 * J>(adr[LO + r]) : 0ull;
#pragma unroll
    for (int g = 0; g < NP / 2; ++g) lvb_wait2(eq[2 * g], eq[2 * g + 1]);
    uint32_t Ph0 = 0xffffffffu, Mh0 = 0u, Ph1 = 0xffffffffu, Mh1 = 0u;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const int i = LO + r;
        const uint32_t e0 = (uint32_t)eq[r], e1 = (uint32_t)(eq[r] >> 32);
        const uint32_t D00 = GM_BITOP3(e0, Mv0[i], Mh0, TT_OR3);
        const uint32_t D01 = GM_BITOP3(e1, Mv1[i], Mh1, TT_OR3);
        const uint32_t Pho0 = GM_BITOP3(Mv0[i], D00, Pv0[i], TT_A_OR_NOR_BC);
        const uint32_t Pho1 = GM_BITOP3(Mv1[i], D01, Pv1[i], TT_A_OR_NOR_BC);
        const uint32_t Mho0 = Pv0[i] & D00;
        const uint32_t Mho1 = Pv1[i] & D01;
        Pv0[i] = GM_BITOP3(Mh0, D00, Ph0, TT_A_OR_NOR_BC);
        Pv1[i] = GM_BITOP3(Mh1, D01, Ph1, TT_A_OR_NOR_BC);
        Mv0[i] = Ph0 & D00;
        Mv1[i] = Ph1 & D01;
        Ph0 = Pho0;
        Mh0 = Mho0;
        Ph1 = Pho1;
        Mh1 = Mho1;
        if (i == J) {
            d0.template add<0, J + 1>(D00);
            d1.template add<0, J + 1>(D01);
        }
    }
    asm volatile("" : "+v"(Ph0), "+v"(Mh0), "+v"(Ph1), "+v"(Mh1));
}

template <int L, int BAND, int... J>
__device__ __forceinline__ void lvb_columns_banded2(const uint32_t (&adr)[L], uint32_t (&Pv0)[L], uint32_t (&Mv0)[L],
                                                    uint32_t (&Pv1)[L], uint32_t (&Mv1)[L], LvbDiagCount& d0,
                                                    LvbDiagCount& d1, std::integer_sequence<int, J...>) {
    (lvb_column_banded2<L, BAND, J>(adr, Pv0, Mv0, Pv1, Mv1, d0, d1), ...);
}

// banded DP of a whole group (both halves); adr[] is NOT advanced
template <int L, int BAND>
__device__ __forceinline__ void lvb_eval_group_banded(const uint32_t (&adr)[L], uint32_t (&c0)[5], uint32_t (&c1)[5]) {
    uint32_t Pv0[L], Mv0[L], Pv1[L], Mv1[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        Pv0[i] = Pv1[i] = 0xffffffffu;
        Mv0[i] = Mv1[i] = 0u;
    }
    LvbDiagCount d0, d1;
    lvb_columns_banded2<L, BAND>(adr, Pv0, Mv0, Pv1, Mv1, d0, d1, std::make_integer_sequence<int, L>{});
    d0.template finish<0, L>(c0);
    d1.template finish<0, L>(c1);
    asm volatile("" : "+v"(c0[0]), "+v"(c0[1]), "+v"(c0[2]), "+v"(c0[3]), "+v"(c0[4]));
    asm volatile("" : "+v"(c1[0]), "+v"(c1[1]), "+v"(c1[2]), "+v"(c1[3]), "+v"(c1[4]));
}

// K2B scan.  grid = (ceil(n_q / 64), n_seg); tiles = the K1L one-hot table tiles (LK_TG groups of 64).
// Dynamic LDS: 2 tiles.  seed_*: Hamming k-NN of the same candidates (lev <= hamming).
// keys: [n_seg][n_q][KCAP] raw RegList contents, folded by k_merge_lists.
// Two instantiations share the grid: FULL = false serves the workgroups whose 64 candidates all start with a bound
// inside the band (bounds only tighten, so such a wave never needs the full matrix); it contains the banded code only,
// fits ~110 VGPRs and runs at 4 waves/SIMD.  FULL = true serves the other workgroups (unseeded rows, far k-th
// neighbours) with every path (~155 VGPRs, 3 waves/SIMD).  Each workgroup decides from its seeds alone which of the
// two launches it belongs to and leaves the other at once.
template <int L, int KCAP, bool FULL>
__global__ __launch_bounds__(WAVE, (FULL ? GM_LVB_WAVES : GM_LVB_BAND_WAVES)) void k_leven_knn_bs(const uint32_t* __restrict__ table, int n_db, int n_tiles,
                                                       int tiles_per_seg, const uint2* __restrict__ q,
                                                       int64_t n_q, int k, const int32_t* __restrict__ seed_dists,
                                                       const int32_t* __restrict__ seed_counts, int seed_k,
                                                       uint32_t* __restrict__ keys, const int* __restrict__ cap_cnt,
                                                       int cap_thr) {
    extern __shared__ __align__(16) uint32_t smem[];
    constexpr int TILE_W = lk_tile_bytes(L) / 4;
    constexpr int GW = lk_group_words(L);
    const int lane = threadIdx.x;
    const int seg = blockIdx.y;
    const int64_t qi = (int64_t)blockIdx.x * WAVE + lane;
    const bool valid = qi < n_q;
    const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);

    RegList<KCAP> list;
    list.init(k, nullptr, lane);
    int thr_seed = L + 1;
    if (valid && seed_dists != nullptr && (seed_counts == nullptr || seed_counts[qi] >= seed_k))
        thr_seed = seed_dists[qi * seed_k + (seed_k - 1)] + 1;      // lev <= hamming
    if (thr_seed > L + 1) thr_seed = L + 1;
    // narrow-band first pass (gmhip.hip run_leven_lookup): when most candidates are seeded at or just above cap_thr, every
    // bound is lowered to it -- the whole launch then runs the narrower band --, and a candidate whose list does not
    // fill under it is scanned again under its own seed
    if (cap_cnt != nullptr && thr_seed > cap_thr && (int64_t)(*cap_cnt) * 10 >= n_q * 9) thr_seed = cap_thr;
    int thr = valid ? thr_seed : 0;
    {
        const bool in_band = __builtin_amdgcn_ballot_w64(thr > 2 * lvb_band(L) + 2) == 0ull;
        if (in_band == FULL) return;      // the other launch serves this workgroup
    }
    uint32_t Tb[5];
    lk_set_bound(Tb, L, thr);

    const int t0 = seg * tiles_per_seg;
    const int t1 = t0 + tiles_per_seg < n_tiles ? t0 + tiles_per_seg : n_tiles;
    if (t0 < t1) lk_stage<L>(table, t0, smem, lane);
    for (int t = t0; t < t1; ++t) {
        const uint32_t* cur = smem + ((t - t0) & 1) * TILE_W;
        lk_stage_wait();                                               // tile t has landed
        if (t + 1 < t1) lk_stage<L>(table, t + 1, smem + ((t - t0 + 1) & 1) * TILE_W, lane);
        // LDS byte address of (group 0, column 0, base of candidate position i, half 0)
        // (re-derived from the packed candidate every tile: the opaque asm keeps the compiler from
        // caching L base offsets in registers across the whole scan)
        uint32_t adr[L];
        const uint32_t cur_b = (uint32_t)(size_t)cur;
        uint32_t qx = qv.x, qy = qv.y;
        asm volatile("" : "+v"(qx), "+v"(qy));
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const uint32_t code = ((qx >> i) & 1u) | (((qy >> i) & 1u) << 1);
            adr[i] = cur_b + code * 8u;
        }
        for (int gl = 0; gl < LK_TG; ++gl) {
            uint32_t c0[5], c1[5];
            constexpr int BAND = lvb_band(L), NARROW = BAND - 1;
            const bool banded = !FULL || __builtin_amdgcn_ballot_w64(thr > 2 * BAND + 2) == 0ull;
            const bool narrow = NARROW >= 2 && __builtin_amdgcn_ballot_w64(thr > 2 * NARROW + 2) == 0ull;
            if (narrow) {
                if constexpr (!FULL) {
                    lvb_eval_group_banded<L, (NARROW >= 2 ? NARROW : 2)>(adr, c0, c1);
                } else {
                    lvb_eval_half_banded<L, 0, (NARROW >= 2 ? NARROW : 2)>(adr, c0);
                    lvb_eval_half_banded<L, 4, (NARROW >= 2 ? NARROW : 2)>(adr, c1);
                }
#pragma unroll
                for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
            } else if (banded) {
                if constexpr (!FULL) {
                    lvb_eval_group_banded<L, BAND>(adr, c0, c1);
                } else {
                    lvb_eval_half_banded<L, 0, BAND>(adr, c0);
                    lvb_eval_half_banded<L, 4, BAND>(adr, c1);
                }
#pragma unroll
                for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
            } else if constexpr (FULL) {
                lvb_eval_half<L, 0>(adr, c0);                              // adr: -> next group
#pragma unroll
                for (int i = 0; i < L; ++i) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
                lvb_eval_half<L, 4>(adr, c1);
            } else {
                __builtin_unreachable();      // (a band-only workgroup started inside the band and bounds only tighten)
            }
            uint32_t h0 = GM_BITOP3(c0[0], Tb[0], Tb[0], TT_A_OR_NOTB), h1 = GM_BITOP3(c1[0], Tb[0], Tb[0], TT_A_OR_NOTB);
#pragma unroll
            for (int b = 1; b < 5; ++b) {
                h0 = GM_BITOP3(c0[b], Tb[b], h0, TT_GE);
                h1 = GM_BITOP3(c1[b], Tb[b], h1, TT_GE);
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64((h0 | h1) != 0u) != 0ull, 0)) {
                const int id0 = (t * LK_TG + gl) * LK_GROUP;
                lk_take_hits<L, RegList<KCAP>>(h0, c0, id0, n_db, list, thr, thr_seed, Tb);
                lk_take_hits<L, RegList<KCAP>>(h1, c1, id0 + 32, n_db, list, thr, thr_seed, Tb);
            }
        }
    }
    if (valid) {
        uint32_t* out = keys + ((int64_t)seg * n_q + qi) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) out[s] = list.v[s];
    }
}

// ---------------------------------------------------------------------------------------------------
// K2P scan (gm_pg_types.h gm_lp_copies): K2B's evaluation over the target buckets a pigeonhole partition leaves.
// Workgroup = one chunk of <= 64 candidates that carry the same bases v on block j (lane = candidate, through the
// block's candidate order) against bucket v of every copy of that block (one copy per displacement of the block).
// Targets arrive in copy order, not id order, and the same target can turn up under several displacements and blocks:
// ids come from the copy's position -> id map, the register list takes a key only once (insert_unique) and the bound
// while the list is full is "distance <= that of the k-th entry" (ties compete by id) instead of K2B's strict one.
// keys: [n_blocks][n_q][KCAP] raw lists, folded by k_merge_lists_unique.
// ---------------------------------------------------------------------------------------------------
template <int L, int KCAP>
__device__ __forceinline__ void lvp_take_hits(uint32_t hits, const uint32_t (&c)[5], const uint32_t* __restrict__ ord,
                                              int64_t pos0, RegList<KCAP>& list, int& thr, int thr_seed, uint32_t (&Tb)[5]) {
    while (hits != 0u) {
        const int j = __builtin_ctz(hits);
        hits &= hits - 1u;
        const int s = (int)(((c[0] >> j) & 1u) | (((c[1] >> j) & 1u) << 1) | (((c[2] >> j) & 1u) << 2) |
                            (((c[3] >> j) & 1u) << 3) | (((c[4] >> j) & 1u) << 4));
        const int d = L - s;
        if (d >= thr) continue;
        const uint32_t id = ord[pos0 + j];
        if (id == 0xffffffffu) continue;      // padding of the bucket
        list.insert_unique(((uint32_t)d << ID_BITS) | id);
        // (while the list is full a new target must be no farther than the k-th entry: ties compete by id)
        int nt = list.v[KCAP - 1] == RegList<KCAP>::EMPTY ? L + 1 : (int)((list.v[KCAP - 1] - 1u) >> ID_BITS) + 1;
        nt = nt < thr_seed ? nt : thr_seed;
        if (nt != thr) {
            thr = nt;
            lk_set_bound(Tb, L, thr);
        }
    }
}

template <int L, int KCAP, bool FULL>
__global__ __launch_bounds__(WAVE, (FULL ? GM_LVB_WAVES : GM_LVB_BAND_WAVES)) void k_leven_part_bs(
    gm_lp_copies P, const gm_lp_chunk* __restrict__ chunks, const uint2* __restrict__ q, int64_t n_q, int k,
    const int32_t* __restrict__ seed_dists, const int32_t* __restrict__ seed_counts, int seed_k, uint32_t* __restrict__ keys,
    int radius, const int* __restrict__ n_chunks_dev) {
    extern __shared__ __align__(16) uint32_t smem[];
    constexpr int TILE_W = lk_tile_bytes(L) / 4;
    constexpr int GW = lk_group_words(L);
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= *n_chunks_dev) return;      // (the launch covers an upper bound of the chunk table)
    const gm_lp_chunk ch = chunks[blockIdx.x];
    const bool valid = lane < ch.cn;
    const int64_t qi = valid ? (int64_t)P.cperm[ch.j][ch.c0 + lane] : 0;
    const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);

    RegList<KCAP> list;
    list.init(k, nullptr, lane);
    int thr_seed = L + 1;
    if (valid && seed_dists != nullptr && (seed_counts == nullptr || seed_counts[qi] >= seed_k))
        thr_seed = seed_dists[qi * seed_k + (seed_k - 1)] + 1;      // lev <= hamming; ties at the seed distance compete by id
    if (thr_seed > L + 1) thr_seed = L + 1;
    // the partition of radius r enumerates the targets within r and no others completely: a row is final only when k of them
    // lie within r, so nothing beyond r is worth a list entry -- and the bound r + 1 keeps the DP in its narrowest band
    if (thr_seed > radius + 1) thr_seed = radius + 1;
    int thr = valid ? thr_seed : 0;
    {
        const bool in_band = __builtin_amdgcn_ballot_w64(thr > 2 * lvb_band(L) + 2) == 0ull;
        if (in_band == FULL) return;      // the other launch serves this workgroup
    }
    uint32_t Tb[5];
    lk_set_bound(Tb, L, thr);

    for (int cp = P.first_copy[ch.j]; cp < P.first_copy[ch.j + 1]; ++cp) {
        const int2 bd = P.bdesc[cp][ch.v];      // (wave-uniform: scalar loads)
        if (bd.y <= 0) continue;
        const uint32_t* __restrict__ table = P.table[cp];
        const uint32_t* __restrict__ ord = P.ord[cp];
        const int t0 = bd.x, t1 = bd.x + (bd.y + LK_TG - 1) / LK_TG;      // (.y = 64-target groups that hold targets)
        lk_stage<L>(table, t0, smem, lane);
        for (int t = t0; t < t1; ++t) {
            const uint32_t* cur = smem + ((t - t0) & 1) * TILE_W;
            lk_stage_wait();
            if (t + 1 < t1) lk_stage<L>(table, t + 1, smem + ((t - t0 + 1) & 1) * TILE_W, lane);
            uint32_t adr[L];
            const uint32_t cur_b = (uint32_t)(size_t)cur;
            uint32_t qx = qv.x, qy = qv.y;
            asm volatile("" : "+v"(qx), "+v"(qy));
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const uint32_t code = ((qx >> i) & 1u) | (((qy >> i) & 1u) << 1);
                adr[i] = cur_b + code * 8u;
            }
            const int g_here = bd.y - (t - t0) * LK_TG < LK_TG ? bd.y - (t - t0) * LK_TG : LK_TG;      // (the last tile: no padding groups)
            for (int gl = 0; gl < g_here; ++gl) {
                uint32_t c0[5], c1[5];
                constexpr int BAND = lvb_band(L), NARROW = BAND - 1;
                const bool banded = !FULL || __builtin_amdgcn_ballot_w64(thr > 2 * BAND + 2) == 0ull;
                const bool narrow = NARROW >= 2 && __builtin_amdgcn_ballot_w64(thr > 2 * NARROW + 2) == 0ull;
                // (radius <= 5: every bound is <= 6 and two diagonals either side hold every alignment that matters)
                const bool tiny = !FULL && NARROW > 2 && __builtin_amdgcn_ballot_w64(thr > 2 * 2 + 2) == 0ull;
                if (tiny) {
                    if constexpr (!FULL && NARROW > 2) lvb_eval_group_banded<L, 2>(adr, c0, c1);
#pragma unroll
                    for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
                } else if (narrow) {
                    if constexpr (!FULL) {
                        lvb_eval_group_banded<L, (NARROW >= 2 ? NARROW : 2)>(adr, c0, c1);
                    } else {
                        lvb_eval_half_banded<L, 0, (NARROW >= 2 ? NARROW : 2)>(adr, c0);
                        lvb_eval_half_banded<L, 4, (NARROW >= 2 ? NARROW : 2)>(adr, c1);
                    }
#pragma unroll
                    for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
                } else if (banded) {
                    if constexpr (!FULL) {
                        lvb_eval_group_banded<L, BAND>(adr, c0, c1);
                    } else {
                        lvb_eval_half_banded<L, 0, BAND>(adr, c0);
                        lvb_eval_half_banded<L, 4, BAND>(adr, c1);
                    }
#pragma unroll
                    for (int i = 0; i < L; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
                } else if constexpr (FULL) {
                    lvb_eval_half<L, 0>(adr, c0);
#pragma unroll
                    for (int i = 0; i < L; ++i) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(adr[i]) : "n"(GW * 4));
                    lvb_eval_half<L, 4>(adr, c1);
                } else {
                    __builtin_unreachable();
                }
                uint32_t h0 = GM_BITOP3(c0[0], Tb[0], Tb[0], TT_A_OR_NOTB), h1 = GM_BITOP3(c1[0], Tb[0], Tb[0], TT_A_OR_NOTB);
#pragma unroll
                for (int b = 1; b < 5; ++b) {
                    h0 = GM_BITOP3(c0[b], Tb[b], h0, TT_GE);
                    h1 = GM_BITOP3(c1[b], Tb[b], h1, TT_GE);
                }
#ifdef GM_LVP_NOHITS      /* timing experiment only (results WRONG): the evaluation without the hit path */
                asm volatile("" ::"v"(h0), "v"(h1));
                if (false) {
#else
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((h0 | h1) != 0u) != 0ull, 0)) {
#endif
                    const int64_t pos0 = ((int64_t)t * LK_TG + gl) * LK_GROUP;
                    lvp_take_hits<L, KCAP>(h0, c0, ord, pos0, list, thr, thr_seed, Tb);
                    lvp_take_hits<L, KCAP>(h1, c1, ord, pos0 + 32, list, thr, thr_seed, Tb);
                }
            }
        }
    }
    if (valid) {
        uint32_t* out = keys + ((int64_t)ch.j * n_q + qi) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) out[s] = list.v[s];
    }
}

}  // namespace gm
