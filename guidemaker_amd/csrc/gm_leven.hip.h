// gm_leven.hip.h -- Levenshtein k-NN scan for gfx950 (K2L): Myers/Hyyro bit-parallel global edit
// distance with the match masks looked up from LDS.
//
//   pattern = pool target (its four match masks Peq[A,C,G,T], bit i = target has that base at i, are
//             precomputed once per index: 16 B per target, staged through LDS tiles by LDS-DMA)
//   text    = candidate (lane = one candidate); for text position j the lane needs
//             Eq_j = Peq_target[candidate base at j]  -> ONE ds_read_b32 with a per-lane address
//             (4 distinct addresses per instruction, conflict-free), no VALU select.
//   column step, all operands in VGPRs (10 VALU ops, 3-input ops as v_bitop3):
//       Xv = Eq | Mv;  t = (Eq & Pv) + Pv;  Xh = (t ^ Pv) | Eq;  Ph = Mv | ~(Xh | Pv);  Mh = Pv & Xh
//       Ph = (Ph << 1) | 1;  Pv = (Mh << 1) | ~(Xv | Ph);  Mv = Ph & Xv
//   distance = L + popcount(Pv & mask) - popcount(Mv & mask)   (sum of the last column's deltas)
//
// Work split: grid = (candidate chunks of 64) x (pool segments).  Each (chunk, segment) wave keeps
// its own sorted k-list in registers and writes it to a scratch array; k_leven_merge folds the
// segment lists into the final top-k.  Thresholds are SEEDED: lev(a,b) <= hamming(a,b), so the k-th
// Hamming neighbour distance (from the fast K1L scan of the same pool) + 1 is a valid initial bound --
// every segment starts tight, no warm-up phase and no cross-segment communication.
#pragma once
#include "gm_lookup.hip.h"

namespace gm {

constexpr int LV_TILE = 128;                 // pool targets per LDS tile (2 KiB; two tiles double-buffer)
constexpr int LV_UNROLL = 4;                 // targets per pointer bump

// truth tables, index = (a << 2 | b << 1 | c)
constexpr int TT_AXB_OR_C = ((0xF0 ^ 0xCC) | 0xAA) & 0xFF;          // (a ^ b) | c
constexpr int TT_A_OR_NOR_BC = (0xF0 | ~(0xCC | 0xAA)) & 0xFF;      // a | ~(b | c)
constexpr int TT_NOR_AB = (~(0xF0 | 0xCC)) & 0xFF;                  // ~(a | b)      (c ignored)

// planes -> Peq table (uint4 per target: masks for A, C, G, T); ids past n_db replicate the last
static __global__ void k_to_peq(const uint2* __restrict__ planes, int64_t n_db, uint4* __restrict__ peq,
                                int64_t n_pad, int L) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const uint2 e = planes[i < n_db ? i : n_db - 1];
    const uint32_t mask = (1u << L) - 1u;
    peq[i] = make_uint4(~e.x & ~e.y & mask, e.x & ~e.y & mask, ~e.x & e.y & mask, e.x & e.y & mask);
}

// stage one Peq tile (4 KiB = 4 rounds x 64 lanes x 16 B) straight into LDS
__device__ __forceinline__ void lv_stage(const uint4* __restrict__ peq, int64_t tile, uint32_t* dst, int lane) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(peq + tile * LV_TILE);
#pragma unroll
    for (int r = 0; r < LV_TILE * 16 / (WAVE * 16); ++r)
        __builtin_amdgcn_global_load_lds(src + (r * WAVE + lane) * 4, dst + r * WAVE * 4, 16, 0, 0);
}

// x << 1 as x + x: on gfx950 v_lshlrev_b32 issues at half rate, v_add_u32 at full rate
// (tools/valu_rate.hip); asm because the compiler canonicalises x + x back into a shift
__device__ __forceinline__ uint32_t shl1(uint32_t x) {
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

// WOFF = constant word offset of the target inside the LDS window addressed by ptr[]
template <int L, int WOFF>
__device__ __forceinline__ int lv_distance(const uint32_t* const (&ptr)[L], uint32_t mask) {
    uint32_t eq[L];
#pragma unroll
    for (int j = 0; j < L; ++j) eq[j] = ptr[j][WOFF];          // ds_read_b32, per-lane address
    uint32_t Pv = 0xffffffffu, Mv = 0u;
#pragma unroll
    for (int j = 0; j < L; ++j) {
        const uint32_t Eq = eq[j];
        const uint32_t Xv = Eq | Mv;
        const uint32_t t = (Eq & Pv) + Pv;
        const uint32_t Xh = GM_BITOP3(t, Pv, Eq, TT_AXB_OR_C);
        uint32_t Ph = GM_BITOP3(Mv, Xh, Pv, TT_A_OR_NOR_BC);
        const uint32_t Mh = Pv & Xh;
        Ph = shl1(Ph) | 1u;
        const uint32_t n = GM_BITOP3(Xv, Ph, Ph, TT_NOR_AB);
        Pv = shl1(Mh) | n;
        Mv = Ph & Xv;
    }
    return L + __builtin_popcount(Pv & mask) - __builtin_popcount(Mv & mask);
}

// K2L scan.  grid = (ceil(n_q / 64), n_seg).  Dynamic LDS: 2 tiles (8 KiB).
//   seed_dists/seed_counts/seed_k: Hamming k-NN of the same candidates (or nullptr): initial bound
//   keys: [n_seg][n_q][KCAP] raw RegList contents
template <int L, int KCAP>
__global__ __launch_bounds__(WAVE) void k_leven_knn_lk(const uint4* __restrict__ peq, int n_db, int n_tiles,
                                                       int tiles_per_seg, const uint2* __restrict__ q,
                                                       int64_t n_q, int k, const int32_t* __restrict__ seed_dists,
                                                       const int32_t* __restrict__ seed_counts, int seed_k,
                                                       uint32_t* __restrict__ keys) {
    extern __shared__ __align__(16) uint32_t smem[];
    constexpr int TILE_W = LV_TILE * 4;
    const int lane = threadIdx.x;
    const int seg = blockIdx.y;
    const int64_t qi = (int64_t)blockIdx.x * WAVE + lane;
    const bool valid = qi < n_q;
    const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);
    const uint32_t mask = (1u << L) - 1u;

    RegList<KCAP> list;
    list.init(k, nullptr, lane);
    int thr_seed = L + 1;
    if (valid && seed_dists != nullptr && (seed_counts == nullptr || seed_counts[qi] >= seed_k))
        thr_seed = seed_dists[qi * seed_k + (seed_k - 1)] + 1;      // lev <= hamming
    if (thr_seed > L + 1) thr_seed = L + 1;
    int thr = valid ? thr_seed : 0;

    const int t0 = seg * tiles_per_seg;
    const int t1 = t0 + tiles_per_seg < n_tiles ? t0 + tiles_per_seg : n_tiles;
    if (t0 < t1) {
        lv_stage(peq, t0, smem, lane);
        for (int t = t0; t < t1; ++t) {
            uint32_t* cur = smem + ((t - t0) & 1) * TILE_W;
            lk_stage_wait();                                              // tile t has landed
            if (t + 1 < t1) lv_stage(peq, t + 1, smem + ((t - t0 + 1) & 1) * TILE_W, lane);
            // per-lane LDS address of (target 0 of this tile, this lane's base at text position j)
            const uint32_t* ptr[L];
#pragma unroll
            for (int j = 0; j < L; ++j) {
                const uint32_t code = ((qv.x >> j) & 1u) | (((qv.y >> j) & 1u) << 1);
                ptr[j] = cur + code;
            }
            const int id_base = t * LV_TILE;
            for (int e = 0; e < LV_TILE; e += LV_UNROLL) {
                int d[LV_UNROLL];
                d[0] = lv_distance<L, 0>(ptr, mask);
                d[1] = lv_distance<L, 4>(ptr, mask);
                d[2] = lv_distance<L, 8>(ptr, mask);
                d[3] = lv_distance<L, 12>(ptr, mask);
                bool hit = false;
#pragma unroll
                for (int u = 0; u < LV_UNROLL; ++u) hit = hit || (d[u] < thr);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(hit) != 0ull, 0)) {
#pragma unroll
                    for (int u = 0; u < LV_UNROLL; ++u) {
                        const int id = id_base + e + u;
                        if (d[u] < thr && id < n_db) {
                            list.insert(((uint32_t)d[u] << ID_BITS) | (uint32_t)id);
                            const int lt = list.threshold(L);
                            thr = lt < thr_seed ? lt : thr_seed;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < L; ++j) ptr[j] += 4 * LV_UNROLL;
            }
        }
    }
    if (valid) {
        uint32_t* out = keys + ((int64_t)seg * n_q + qi) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) out[s] = list.v[s];
    }
}

}  // namespace gm
