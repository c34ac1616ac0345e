// gm_bitslice.hip.h -- the bit-sliced counting primitives shared by the look-up kernels (gm_lookup.hip.h, gm_leven_bs.hip.h,
// gm_pigeon_scan.hip.h) for gfx950.
//
// Why bit-sliced: on MI355X v_bcnt_u32_b32 / v_min3_u32 / v_cmp issue at HALF rate (4 cycles per wave64) while
// v_xor_b32 and the 3-input v_bitop3_b32 issue at full rate (2 cycles) -- measured by tools/valu_rate.hip.  A kernel
// that spends one popcount per pair is bound at ~9-10 cycles per pair; here 32 pool entries are evaluated per 32-bit word
// with full-rate ops only: L one-bit flag words (match / mismatch at position p of 32 entries) are summed into a
// bit-sliced 5-bit number by a 3:2 carry-save adder tree (full adder = 2 x v_bitop3: xor3 0x96, majority 0xE8; 15 FA +
// 3 HA at L = 20) and compared with the lane's threshold bits by a ripple comparator, one v_bitop3 per bit.
// (Round 1's scalar-load kernel K1v, which lived here as well, was superseded by K1L in round 1 and removed in round 4.)
#pragma once
#include "gm_kernels.hip.h"

namespace gm {

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

}  // namespace gm
