// gm_bitslice.hip.h -- bit-sliced ("vertical") Hamming k-NN scan for gfx950.
//
// Why: on MI355X v_bcnt_u32_b32 / v_min3_u32 / v_cmp issue at HALF rate (4 cycles per wave64) while
// v_xor_b32 and the 3-input v_bitop3_b32 issue at full rate (2 cycles) -- measured by
// tools/valu_rate.hip.  The horizontal kernel (one popcount per pair) is therefore bound at
// ~9-10 cycles per pair.  Here 32 db entries are evaluated per 32-bit word with full-rate ops only:
//
//   db "vertical" layout: group g = db ids [32g, 32g+32); word V[g][2p] holds the low code bit of
//   base p of those 32 entries (bit j = entry 32g+j), V[g][2p+1] the high bit.  These words are
//   wave-uniform and arrive through SCALAR loads (s_load_dwordx8/x16 -> SGPR operands, no LDS).
//   Per lane = one query; QL[p] / QH[p] = the query's code bits at base p broadcast to 0 / ~0.
//     M_p   = (V[2p] ^ QL[p]) | (V[2p+1] ^ QH[p])          mismatch flags of 32 entries, 2 ops
//     count = sum_p M_p as a bit-sliced 5-bit number via a 3:2 carry-save adder tree
//             (full adder = 2 x v_bitop3: xor3 0x96, majority 0xE8), 15 FA + 3 HA at L=20
//     hit   = count < thr, ripple comparator against the lane's threshold bits, 1 bitop3 per bit
//   => 40 + 36 + 5 full-rate ops per 32 pairs at L=20 (2.5 ops, ~5 cycles per pair), one v_cmp +
//   wave-uniform branch per group.  Lanes with hit bits run the slow path: for each set bit read
//   the distance back from the bit-sliced counter and insert (dist << 27 | id) into the lane's
//   sorted k-list in LDS.  Ids arrive ascending, so strict d < thr gives the (distance, id) order.
#pragma once
#include "gm_kernels.hip.h"

namespace gm {

constexpr int GROUP = 32;
__host__ __device__ constexpr int vert_stride(int L) { return (2 * L + 3) / 4 * 4; }   // words per group

#define GM_BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
// truth tables: index = (a << 2 | b << 1 | c)
constexpr int TT_XOR3 = 0x96;      // a ^ b ^ c
constexpr int TT_MAJ = 0xE8;       // majority(a, b, c)
constexpr int TT_A_OR_BXC = 0xF6;  // a | (b ^ c)
constexpr int TT_LT = 0x8E;        // less-than stage: (~a & b) | (~(a ^ b) & c), a = count bit, b = thr bit, c = lt so far
static_assert(((0xF0 | (0xCC ^ 0xAA)) & 0xFF) == TT_A_OR_BXC, "a | (b ^ c)");
static_assert((((~0xF0 & 0xCC) | (~(0xF0 ^ 0xCC) & 0xAA)) & 0xFF) == TT_LT, "lt stage");
static_assert(((0xF0 ^ 0xCC ^ 0xAA) & 0xFF) == TT_XOR3, "xor3");
static_assert((((0xF0 & 0xCC) | (0xAA & (0xF0 ^ 0xCC))) & 0xFF) == TT_MAJ, "maj");

// ---------------------------------------------------------------------------------------------
// horizontal planes -> vertical groups.  One thread per output word.
// ---------------------------------------------------------------------------------------------
static __global__ void k_to_vertical(const uint2* __restrict__ planes, int64_t n_db, uint32_t* __restrict__ V,
                              int64_t n_groups, int L, int W) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_groups * W) return;
    int64_t g = t / W;
    int w = (int)(t % W);
    uint32_t out = 0;
    if (w < 2 * L) {
        int p = w >> 1, hi = w & 1;
        for (int j = 0; j < GROUP; ++j) {
            int64_t id = g * GROUP + j;
            if (id >= n_db) id = n_db - 1;          // padding replicates the last real entry
            uint2 e = planes[id];
            uint32_t bit = ((hi ? e.y : e.x) >> p) & 1u;
            out |= bit << j;
        }
    }
    V[t] = out;
}

// ---------------------------------------------------------------------------------------------
// carry-save column reduction: N same-weight bit-sliced words -> 1 sum word + floor(N/2) carries
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void col_reduce(const uint32_t (&x)[N], uint32_t& sum, uint32_t* carry) {
    if constexpr (N == 1) {
        sum = x[0];
    } else if constexpr (N == 2) {
        sum = x[0] ^ x[1];
        carry[0] = x[0] & x[1];
    } else {
        constexpr int T = N / 3, REM = N % 3;
        uint32_t nx[T + REM];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            nx[t] = GM_BITOP3(x[3 * t], x[3 * t + 1], x[3 * t + 2], TT_XOR3);
            carry[t] = GM_BITOP3(x[3 * t], x[3 * t + 1], x[3 * t + 2], TT_MAJ);
        }
#pragma unroll
        for (int r = 0; r < REM; ++r) nx[T + r] = x[3 * T + r];
        col_reduce<T + REM>(nx, sum, carry + T);
    }
}

// bit-sliced popcount of L flag words: c[0..4] = bits of the per-entry count (L <= 31)
template <int L>
__device__ __forceinline__ void bitslice_count(const uint32_t (&m)[L], uint32_t (&c)[5]) {
    constexpr int N1 = L / 2, N2 = N1 / 2, N3 = N2 / 2, N4 = N3 / 2;
    static_assert(N4 <= 1 && L <= 31, "5-bit counter");
    uint32_t k1[N1 > 0 ? N1 : 1], k2[N2 > 0 ? N2 : 1], k3[N3 > 0 ? N3 : 1], k4[N4 > 0 ? N4 : 1];
    c[1] = c[2] = c[3] = c[4] = 0u;
    col_reduce<L>(m, c[0], k1);
    if constexpr (N1 > 0) col_reduce<N1>(k1, c[1], k2);
    if constexpr (N2 > 0) col_reduce<N2>(k2, c[2], k3);
    if constexpr (N3 > 0) col_reduce<N3>(k3, c[3], k4);
    if constexpr (N4 > 0) c[4] = k4[0];
}

// ---------------------------------------------------------------------------------------------
// K1v: bit-sliced Hamming k-NN.  grid = ceil(n_q / 64) single-wave workgroups, lane = query.
// Dynamic LDS: k * 64 * 4 B (the lists).  k == 1 doubles as the nearest-distance kernel
// (NEAREST: write only the distance).
// ---------------------------------------------------------------------------------------------
// evaluate one group of 32 db entries for this lane's query: c = bit-sliced distance, returns the
// hit word (bit j set <=> distance of entry j < thr)
template <int L>
__device__ __forceinline__ uint32_t eval_group(const uint32_t* __restrict__ vg, const uint32_t (&QL)[L],
                                               const uint32_t (&QH)[L], const uint32_t (&T)[5],
                                               uint32_t (&c)[5]) {
    uint32_t m[L];
#pragma unroll
    for (int p = 0; p < L; ++p) {
        uint32_t x = vg[2 * p] ^ QL[p];
        m[p] = GM_BITOP3(x, vg[2 * p + 1], QH[p], TT_A_OR_BXC);
    }
    bitslice_count<L>(m, c);
    uint32_t lt = ~c[0] & T[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) lt = GM_BITOP3(c[i], T[i], lt, TT_LT);
    return lt;
}

// the same on a group whose 2L words were fetched earlier (software-pipelined scalar loads)
template <int L>
struct GroupWords {
    uint32_t w[2 * L];
};
template <int L>
__device__ __forceinline__ void fetch_group(GroupWords<L>& a, const uint32_t* __restrict__ vg) {
#pragma unroll
    for (int i = 0; i < 2 * L; ++i) a.w[i] = vg[i];
}
template <int L>
__device__ __forceinline__ uint32_t eval_words(const GroupWords<L>& a, const uint32_t (&QL)[L],
                                               const uint32_t (&QH)[L], const uint32_t (&T)[5],
                                               uint32_t (&c)[5]) {
    uint32_t m[L];
#pragma unroll
    for (int p = 0; p < L; ++p) {
        uint32_t x = a.w[2 * p] ^ QL[p];
        m[p] = GM_BITOP3(x, a.w[2 * p + 1], QH[p], TT_A_OR_BXC);
    }
    bitslice_count<L>(m, c);
    uint32_t lt = ~c[0] & T[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) lt = GM_BITOP3(c[i], T[i], lt, TT_LT);
    return lt;
}

// slow path for one group: walk the lane's hit bits in ascending id order
__device__ __forceinline__ void take_hits(uint32_t lt, const uint32_t (&c)[5], int id0, int n_db, uint32_t* lp,
                                          int k, int& cnt, int& thr, uint32_t (&T)[5]) {
    while (lt != 0u) {
        int j = __builtin_ctz(lt);
        lt &= lt - 1u;
        int d = (int)(((c[0] >> j) & 1u) | (((c[1] >> j) & 1u) << 1) | (((c[2] >> j) & 1u) << 2) |
                      (((c[3] >> j) & 1u) << 3) | (((c[4] >> j) & 1u) << 4));
        int id = id0 + j;
        if (d < thr && id < n_db) {
            list_insert(lp, WAVE, k, cnt, ((uint32_t)d << ID_BITS) | (uint32_t)id);
            if (cnt == k) {
                thr = (int)(lp[(k - 1) * WAVE] >> ID_BITS);
#pragma unroll
                for (int i = 0; i < 5; ++i) T[i] = 0u - (((uint32_t)thr >> i) & 1u);
            }
        }
    }
}

// G = groups evaluated per loop iteration (one hit test per G groups); n_groups % 4 == 0.
// NW = waves per workgroup.  Every wave streams the SAME db words through the scalar cache; waves of
// one workgroup are kept within SYNC groups of each other by a barrier so that one wave's scalar-cache
// miss serves the others (a lone wave's scalar loads are latency-bound: ~2k cycles per miss).
constexpr int BS_SYNC = 32;
template <int L, bool NEAREST, int G, int NW>
__global__ __launch_bounds__(WAVE * NW) void k_hamming_knn_bs(const uint32_t* __restrict__ V, int n_db,
                                                              int n_groups, const uint2* __restrict__ q,
                                                              int64_t n_q, int k, int chunks_per_wave,
                                                              int32_t* __restrict__ ids,
                                                              int32_t* __restrict__ dists,
                                                              int32_t* __restrict__ counts) {
    extern __shared__ __align__(16) uint32_t list[];    // [NW][k][64]
    constexpr int W = vert_stride(L);
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = threadIdx.x / WAVE;
    uint32_t* lp = list + wave * k * WAVE + lane;
    const int64_t wave_global = (int64_t)blockIdx.x * NW + wave;

    // a wave owns `chunks_per_wave` consecutive 64-query chunks: the host sizes the grid so that all
    // waves are resident at once and every SIMD gets the same number of chunks (no half-empty tail)
    for (int ch = 0; ch < chunks_per_wave; ++ch) {
        // (no early exit: every wave of the workgroup must reach every barrier)
        const int64_t qi = (wave_global * chunks_per_wave + ch) * WAVE + lane;
        const bool valid = qi < n_q;
        const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);

        uint32_t QL[L], QH[L];
#pragma unroll
        for (int p = 0; p < L; ++p) {
            QL[p] = 0u - ((qv.x >> p) & 1u);
            QH[p] = 0u - ((qv.y >> p) & 1u);
        }
        int thr = valid ? 31 : 0;   // counts are <= 27 < 31: "31" accepts everything until the list is full
        int cnt = 0;
        uint32_t T[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) T[i] = 0u - (((uint32_t)thr >> i) & 1u);

        if constexpr (G == 3) {
            // ping-pong software pipeline: while group g is evaluated the scalar loads of g+1 are in
            // flight, and vice versa (n_groups is even)
            GroupWords<L> A, B;
            fetch_group<L>(A, V);
            for (int g = 0; g < n_groups; g += 2) {
                if constexpr (NW > 1) { if ((g & (BS_SYNC - 1)) == 0) __syncthreads(); }
                uint32_t cA[5], cB[5];
                fetch_group<L>(B, V + (size_t)(g + 1) * W);
                uint32_t ltA = eval_words<L>(A, QL, QH, T, cA);
                const int gn = g + 2 < n_groups ? g + 2 : g;         // last iteration refetches (unused)
                fetch_group<L>(A, V + (size_t)gn * W);
                uint32_t ltB = eval_words<L>(B, QL, QH, T, cB);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64((ltA | ltB) != 0u) != 0ull, 0)) {
                    take_hits(ltA, cA, g * GROUP, n_db, lp, k, cnt, thr, T);
                    take_hits(ltB, cB, (g + 1) * GROUP, n_db, lp, k, cnt, thr, T);
                }
            }
        } else
        for (int g = 0; g < n_groups; g += (G == 4 ? 2 : G)) {
            if constexpr (NW > 1) { if ((g & (BS_SYNC - 1)) == 0) __syncthreads(); }
            // G == 4 is a timing-only ablation: every iteration re-reads groups 0..1 (scalar-cache hits)
            const uint32_t* __restrict__ vg = V + (G == 4 ? (size_t)(g & 2) * W : (size_t)g * W);
            constexpr int GG = (G == 4 ? 2 : G);
            uint32_t c[GG][5], lt[GG];
            uint32_t any = 0u;
#pragma unroll
            for (int u = 0; u < GG; ++u) {
                lt[u] = eval_group<L>(vg + u * W, QL, QH, T, c[u]);
                any |= lt[u];
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(any != 0u) != 0ull, 0)) {
#pragma unroll
                for (int u = 0; u < GG; ++u) {
                    // thresholds may have tightened since lt[u] was formed: take_hits re-tests d < thr
                    take_hits(lt[u], c[u], (g + u) * GROUP, n_db, lp, k, cnt, thr, T);
                }
            }
        }

        if (valid) {
            if constexpr (NEAREST) {
                dists[qi] = cnt > 0 ? (int32_t)(lp[0] >> ID_BITS) : -1;
            } else {
                counts[qi] = cnt;
                for (int t = 0; t < k; ++t) {
                    bool ok = t < cnt;
                    uint32_t key = ok ? lp[t * WAVE] : 0u;
                    ids[qi * k + t] = ok ? (int32_t)(key & ID_MASK) : -1;
                    dists[qi * k + t] = ok ? (int32_t)(key >> ID_BITS) : -1;
                }
            }
        }
    }
}

}  // namespace gm
