// gm_kernels.hip.h -- gfx950 device code for the guide-RNA neighbour search (no MFMA: integer/bitwise).
//
// Data layout in HBM
//   db / query "planes": one uint2 per sequence, .x = low-bit plane, .y = high-bit plane of the
//   2-bit base codes (A0 C1 G2 T3), base p at bit p, bits >= L zero.  Mismatch mask of two
//   sequences = (a.x ^ b.x) | (a.y ^ b.y); Hamming distance = popcount of it: 4 VALU ops
//   (v_xor, v_xor, v_or, v_bcnt with the threshold offset folded into v_bcnt's addend).
//
// Scan kernels (one wave64 per workgroup, no inter-wave sync)
//   every lane owns R queries in VGPRs; the db streams through a double-buffered LDS tile that the
//   wave fills with coalesced 16-byte loads; the inner loop reads db entries with wave-uniform
//   (broadcast) ds_read_b128 and evaluates J entries x R queries per batch.  Per query a running
//   threshold thr (distance of its current k-th neighbour) is folded into the popcount:
//       D = popcount(mask) + (0x80000000 - thr)      =>  D < 0x80000000  <=>  d < thr
//   and a v_min3_u32 chain keeps the batch minimum, so the common case costs 4.5 VALU ops / pair and
//   one wave-uniform branch per batch.  Only when some lane has d < thr does the wave take the slow
//   path, which re-evaluates the batch for the hit queries and inserts (dist << 27 | id) keys into
//   that query's sorted k-list in LDS (slot-major, lane-minor: conflict-free).  Since db ids arrive
//   ascending, the strict test d < thr implements the canonical (distance, id) order exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gm {

constexpr int WAVE = 64;
constexpr int TILE = 512;              // db entries per LDS tile (4 KiB); two tiles double-buffer
constexpr uint32_t SIGN = 0x80000000u;
constexpr int THR_INF = 64;            // > any distance (L <= 27)
constexpr int ID_BITS = 27;
constexpr uint32_t ID_MASK = (1u << ID_BITS) - 1u;

__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// popcount(mask) + addend in ONE VALU op.  Written as asm because hipcc otherwise hoists a common
// addend out of min(popc(a)+C, popc(b)+C) and spends an extra v_add per pair.
__device__ __forceinline__ uint32_t bcnt_add(uint32_t mask, uint32_t addend) {
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(mask), "v"(addend));
    return d;
}

// ---------------------------------------------------------------------------------------------
// packed 2-bit (uint64) -> bit planes (uint2); entries [n, n_pad) replicate entry n-1
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t even_bits(uint64_t x) {
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    return (uint32_t)x;
}

static __global__ void k_to_planes(const uint64_t* __restrict__ in, uint2* __restrict__ out, int64_t n,
                            int64_t n_pad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    uint64_t v = in[i < n ? i : n - 1];
    out[i] = make_uint2(even_bits(v), even_bits(v >> 1));
}

// ---------------------------------------------------------------------------------------------
// sorted k-list in LDS: slot s of query (r, lane) lives at list[(s * R + r) * 64 + lane]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void list_insert(uint32_t* lp, int stride, int k, int& cnt, uint32_t key) {
    int p = cnt < k ? cnt : k - 1;
    while (p > 0) {
        uint32_t prev = lp[(p - 1) * stride];
        if (prev <= key) break;
        lp[p * stride] = prev;
        --p;
    }
    lp[p * stride] = key;
    if (cnt < k) ++cnt;
}

template <int R>
struct QState {
    uint32_t lo[R], hi[R];   // query planes
    uint32_t C[R];           // SIGN - thr
    int thr[R];              // insert iff d < thr
    int cnt[R];              // valid list entries
};

template <int R>
__device__ __forceinline__ void load_queries(QState<R>& s, const uint2* __restrict__ q, int64_t n_q,
                                             int64_t qbase, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        bool valid = qi < n_q;
        uint2 v = valid ? q[qi] : make_uint2(0u, 0u);
        s.lo[r] = v.x;
        s.hi[r] = v.y;
        s.thr[r] = valid ? THR_INF : 0;
        s.C[r] = SIGN - (uint32_t)s.thr[r];
        s.cnt[r] = 0;
    }
}

template <int R>
__device__ __forceinline__ void store_lists(const QState<R>& s, const uint32_t* list, int k, int64_t n_q,
                                            int64_t qbase, int lane, int32_t* __restrict__ ids,
                                            int32_t* __restrict__ dists, int32_t* __restrict__ counts) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        if (qi < n_q) {
            const uint32_t* lp = list + r * WAVE + lane;
            counts[qi] = s.cnt[r];
            for (int t = 0; t < k; ++t) {
                bool ok = t < s.cnt[r];
                uint32_t key = ok ? lp[t * R * WAVE] : 0u;
                ids[qi * k + t] = ok ? (int32_t)(key & ID_MASK) : -1;
                dists[qi * k + t] = ok ? (int32_t)(key >> ID_BITS) : -1;
            }
        }
    }
}

// stage one db tile: global (coalesced 16 B / lane) -> registers -> LDS
static_assert(TILE == 512, "TileRegs holds 4 x 128 entries");
struct TileRegs {
    uint4 v0, v1, v2, v3;
};
__device__ __forceinline__ void tile_fetch(TileRegs& t, const uint2* __restrict__ db, int64_t tile_idx,
                                           int lane) {
    const uint4* src = reinterpret_cast<const uint4*>(db + tile_idx * TILE) + lane;
    t.v0 = src[0 * WAVE];
    t.v1 = src[1 * WAVE];
    t.v2 = src[2 * WAVE];
    t.v3 = src[3 * WAVE];
}
__device__ __forceinline__ void tile_commit(const TileRegs& t, uint2* dst, int lane) {
    uint4* d4 = reinterpret_cast<uint4*>(dst) + lane;
    d4[0 * WAVE] = t.v0;
    d4[1 * WAVE] = t.v1;
    d4[2 * WAVE] = t.v2;
    d4[3 * WAVE] = t.v3;
}

// ---------------------------------------------------------------------------------------------
// K1: Hamming k-NN scan.  grid = ceil(n_q / (64 R)) single-wave workgroups; db is padded to a
// multiple of TILE.  Dynamic LDS: 2 tiles (8 KiB) + k * R * 64 * 4 B of lists.
// ---------------------------------------------------------------------------------------------
template <int R, int J>
__global__ __launch_bounds__(WAVE) void k_hamming_knn(const uint2* __restrict__ db, int n_db, int n_tiles,
                                                      const uint2* __restrict__ q, int64_t n_q, int k,
                                                      int32_t* __restrict__ ids,
                                                      int32_t* __restrict__ dists,
                                                      int32_t* __restrict__ counts) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint2* tiles = reinterpret_cast<uint2*>(smem);
    uint32_t* list = smem + 2 * TILE * 2;
    const int lane = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * (R * WAVE);
    constexpr int stride = R * WAVE;

    QState<R> s;
    load_queries<R>(s, q, n_q, qbase, lane);
    uint32_t m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) m[r] = 0xffffffffu;

    TileRegs pre;
    tile_fetch(pre, db, 0, lane);
    tile_commit(pre, tiles, lane);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const uint2* cur = tiles + (t & 1) * TILE;
        tile_fetch(pre, db, t + 1 < n_tiles ? t + 1 : t, lane);   // clamped: last iteration refetches

        for (int j = 0; j < TILE; j += J) {
            const uint4* e4 = reinterpret_cast<const uint4*>(cur + j);
#pragma unroll
            for (int i = 0; i < J / 2; ++i) {
                uint4 e = e4[i];    // two db entries, wave-uniform address -> broadcast ds_read_b128
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t d0 = bcnt_add((s.lo[r] ^ e.x) | (s.hi[r] ^ e.y), s.C[r]);
                    uint32_t d1 = bcnt_add((s.lo[r] ^ e.z) | (s.hi[r] ^ e.w), s.C[r]);
                    m[r] = umin(m[r], umin(d0, d1));
                }
            }
            uint32_t mall = m[0];
#pragma unroll
            for (int r = 1; r < R; ++r) mall = umin(mall, m[r]);
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mall < SIGN) != 0ull, 0)) {
                // slow path: some lane has a candidate closer than its current k-th neighbour
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (m[r] < SIGN) {
                        uint32_t* lp = list + r * WAVE + lane;
                        for (int jj = 0; jj < J; ++jj) {
                            uint2 e = cur[j + jj];
                            int d = __builtin_popcount((s.lo[r] ^ e.x) | (s.hi[r] ^ e.y));
                            int jg = t * TILE + j + jj;
                            if (d < s.thr[r] && jg < n_db) {
                                list_insert(lp, stride, k, s.cnt[r],
                                            ((uint32_t)d << ID_BITS) | (uint32_t)jg);
                                if (s.cnt[r] == k) s.thr[r] = (int)(lp[(k - 1) * stride] >> ID_BITS);
                            }
                        }
                        s.C[r] = SIGN - (uint32_t)s.thr[r];
                    }
                    m[r] = 0xffffffffu;
                }
            }
        }

        tile_commit(pre, tiles + ((t + 1) & 1) * TILE, lane);
        __syncthreads();
    }
    store_lists<R>(s, list, k, n_q, qbase, lane, ids, dists, counts);
}

// ---------------------------------------------------------------------------------------------
// K1n: Hamming nearest distance only (controls, core.py:603-606): running minimum, no lists.
// Padding entries replicate a real entry, so they never change the minimum.
// ---------------------------------------------------------------------------------------------
template <int R, int J>
__global__ __launch_bounds__(WAVE) void k_hamming_nearest(const uint2* __restrict__ db, int n_tiles,
                                                          const uint2* __restrict__ q, int64_t n_q,
                                                          int32_t* __restrict__ d1) {
    __shared__ __align__(16) uint2 tiles[2 * TILE];
    const int lane = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * (R * WAVE);
    uint32_t lo[R], hi[R], m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        uint2 v = qi < n_q ? q[qi] : make_uint2(0u, 0u);
        lo[r] = v.x;
        hi[r] = v.y;
        m[r] = 0xffffffffu;
    }
    TileRegs pre;
    tile_fetch(pre, db, 0, lane);
    tile_commit(pre, tiles, lane);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const uint2* cur = tiles + (t & 1) * TILE;
        tile_fetch(pre, db, t + 1 < n_tiles ? t + 1 : t, lane);   // clamped: last iteration refetches
        for (int j = 0; j < TILE; j += J) {
            const uint4* e4 = reinterpret_cast<const uint4*>(cur + j);
#pragma unroll
            for (int i = 0; i < J / 2; ++i) {
                uint4 e = e4[i];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t d0 = __builtin_popcount((lo[r] ^ e.x) | (hi[r] ^ e.y));
                    uint32_t d1v = __builtin_popcount((lo[r] ^ e.z) | (hi[r] ^ e.w));
                    m[r] = umin(m[r], umin(d0, d1v));
                }
            }
        }
        tile_commit(pre, tiles + ((t + 1) & 1) * TILE, lane);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        if (qi < n_q) d1[qi] = (int32_t)m[r];
    }
}

// ---------------------------------------------------------------------------------------------
// Levenshtein: Myers / Hyyro bit-parallel global edit distance, pattern = query (per lane, the four
// match masks Peq[c] in VGPRs), text = db entry (wave-uniform).  L <= 27 fits one 32-bit word.
//   per text position: Eq = Peq[c_j]; Xv = Eq|Mv; Xh = (((Eq&Pv)+Pv)^Pv)|Eq; Ph = Mv|~(Xh|Pv);
//   Mh = Pv&Xh; Ph = (Ph<<1)|1; Mh <<= 1; Pv = Mh|~(Xv|Ph); Mv = Ph&Xv
//   distance = L + popcount(Pv & mask) - popcount(Mv & mask)   (sum of the last column's deltas)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int myers_global(uint32_t pa, uint32_t pc, uint32_t pg, uint32_t pt, uint32_t tlo,
                                            uint32_t thi, int L, uint32_t mask) {
    uint32_t Pv = 0xffffffffu, Mv = 0u;
    for (int j = 0; j < L; ++j) {
        // tlo/thi are wave-uniform: the selects below are uniform choices among the lane's masks
        bool b0 = (tlo >> j) & 1u, b1 = (thi >> j) & 1u;
        uint32_t Eq = b1 ? (b0 ? pt : pg) : (b0 ? pc : pa);
        uint32_t Xv = Eq | Mv;
        uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
        uint32_t Ph = Mv | ~(Xh | Pv);
        uint32_t Mh = Pv & Xh;
        Ph = (Ph << 1) | 1u;
        Mh = Mh << 1;
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
    }
    return L + __builtin_popcount(Pv & mask) - __builtin_popcount(Mv & mask);
}

template <int R>
__global__ __launch_bounds__(WAVE) void k_leven_knn(const uint2* __restrict__ db, int n_db, int n_tiles,
                                                    const uint2* __restrict__ q, int64_t n_q, int k, int L,
                                                    int32_t* __restrict__ ids,
                                                    int32_t* __restrict__ dists,
                                                    int32_t* __restrict__ counts) {
    extern __shared__ __align__(16) uint32_t smem[];
    uint2* tiles = reinterpret_cast<uint2*>(smem);
    uint32_t* list = smem + 2 * TILE * 2;
    const int lane = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * (R * WAVE);
    constexpr int stride = R * WAVE;
    const uint32_t mask = (L >= 32) ? 0xffffffffu : ((1u << L) - 1u);

    QState<R> s;
    load_queries<R>(s, q, n_q, qbase, lane);
    uint32_t pa[R], pc[R], pg[R], pt[R];   // match masks of the query (pattern) per base
#pragma unroll
    for (int r = 0; r < R; ++r) {
        pa[r] = ~s.lo[r] & ~s.hi[r] & mask;
        pc[r] = s.lo[r] & ~s.hi[r] & mask;
        pg[r] = ~s.lo[r] & s.hi[r] & mask;
        pt[r] = s.lo[r] & s.hi[r] & mask;
    }

    TileRegs pre;
    tile_fetch(pre, db, 0, lane);
    tile_commit(pre, tiles, lane);
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const uint2* cur = tiles + (t & 1) * TILE;
        tile_fetch(pre, db, t + 1 < n_tiles ? t + 1 : t, lane);   // clamped: last iteration refetches
        int jmax = n_db - t * TILE;
        if (jmax > TILE) jmax = TILE;
        for (int j = 0; j < jmax; ++j) {
            uint2 e = cur[j];
            uint32_t tlo = __builtin_amdgcn_readfirstlane(e.x);
            uint32_t thi = __builtin_amdgcn_readfirstlane(e.y);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int d = myers_global(pa[r], pc[r], pg[r], pt[r], tlo, thi, L, mask);
                bool hit = d < s.thr[r];
                if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                    if (hit) {
                        uint32_t* lp = list + r * WAVE + lane;
                        list_insert(lp, stride, k, s.cnt[r], ((uint32_t)d << ID_BITS) | (uint32_t)(t * TILE + j));
                        if (s.cnt[r] == k) s.thr[r] = (int)(lp[(k - 1) * stride] >> ID_BITS);
                    }
                }
            }
        }
        tile_commit(pre, tiles + ((t + 1) & 1) * TILE, lane);
        __syncthreads();
    }
    store_lists<R>(s, list, k, n_q, qbase, lane, ids, dists, counts);
}

template <int R>
__global__ __launch_bounds__(WAVE) void k_leven_nearest(const uint2* __restrict__ db, int n_db, int n_tiles,
                                                        const uint2* __restrict__ q, int64_t n_q, int L,
                                                        int32_t* __restrict__ d1) {
    __shared__ __align__(16) uint2 tiles[2 * TILE];
    const int lane = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * (R * WAVE);
    const uint32_t mask = (L >= 32) ? 0xffffffffu : ((1u << L) - 1u);
    uint32_t pa[R], pc[R], pg[R], pt[R];
    int m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        uint2 v = qi < n_q ? q[qi] : make_uint2(0u, 0u);
        pa[r] = ~v.x & ~v.y & mask;
        pc[r] = v.x & ~v.y & mask;
        pg[r] = ~v.x & v.y & mask;
        pt[r] = v.x & v.y & mask;
        m[r] = THR_INF;
    }
    TileRegs pre;
    tile_fetch(pre, db, 0, lane);
    tile_commit(pre, tiles, lane);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const uint2* cur = tiles + (t & 1) * TILE;
        tile_fetch(pre, db, t + 1 < n_tiles ? t + 1 : t, lane);   // clamped: last iteration refetches
        int jmax = n_db - t * TILE;
        if (jmax > TILE) jmax = TILE;
        for (int j = 0; j < jmax; ++j) {
            uint2 e = cur[j];
            uint32_t tlo = __builtin_amdgcn_readfirstlane(e.x);
            uint32_t thi = __builtin_amdgcn_readfirstlane(e.y);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int d = myers_global(pa[r], pc[r], pg[r], pt[r], tlo, thi, L, mask);
                m[r] = d < m[r] ? d : m[r];
            }
        }
        tile_commit(pre, tiles + ((t + 1) & 1) * TILE, lane);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int64_t qi = qbase + (int64_t)r * WAVE + lane;
        if (qi < n_q) d1[qi] = m[r];
    }
}

// ---------------------------------------------------------------------------------------------
// K3: CFD (cfd_score_calculator.py:60-85): float64 product in ascending position; the table holds
// 1.0 on the Watson-Crick (= matching) entries, and x * 1.0 is exact, so the loop is branch-free.
// ---------------------------------------------------------------------------------------------
constexpr int CFD_PER_THREAD = 4;      // pairs per thread: all eight operand loads are in flight before the products

// one 32-bit half of the packed pair: positions pos0 .. pos0 + 15 (2 bits each), mismatch flags m at the even bits
__device__ __forceinline__ double cfd_half(double score, uint32_t a, uint32_t b, uint32_t m, int tbase,
                                           const double* __restrict__ smm) {
    while (m != 0u) {
        const int bit = __builtin_ctz(m);      // = 2 x (position - pos0)
        m &= m - 1u;
        const int r = (int)((a >> bit) & 3u);
        const int o = (int)((b >> bit) & 3u);
        score *= smm[tbase + bit * 8 + r * 4 - o];      // ((20 + p - L) * 16 + r * 4 + 3 - o), p * 16 = bit * 8
    }
    return score;
}

__device__ __forceinline__ double cfd_product(uint64_t a, uint64_t b, int L, const double* __restrict__ smm) {
    const int p0 = L > 20 ? L - 20 : 0;
    // only mismatching positions contribute (the reference multiplies only where wt[i] != off[i],
    // cfd_score_calculator.py:79-83); walk them in ascending position = the reference's product order.  The walk runs
    // on the two 32-bit halves of the words (positions 0-15, then 16-31): 64-bit ctz / shifts cost twice the instructions.
    const uint64_t x = a ^ b;
    uint64_t m = (x | (x >> 1)) & 0x5555555555555555ull;
    m &= ~0ull << (2 * p0);
    if (L < 32) m &= (1ull << (2 * L)) - 1ull;
    const int tbase = (20 - L) * 16 + 3;
    double score = cfd_half(1.0, (uint32_t)a, (uint32_t)b, (uint32_t)m, tbase, smm);
    return cfd_half(score, (uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(m >> 32), tbase + 16 * 16, smm);
}

// The score table sits in LDS GM_CFD_COPIES times, every copy shifted by 16 banks against the one before, and a lane
// reads the copy (lane & (COPIES - 1)): the 64 look-ups of a wave are random 8-byte entries of a 2.5 KB table -- ten
// entries per bank pair -- and with one copy a ds_read_b64 serialises ~5 ways.
#ifndef GM_CFD_COPIES
#define GM_CFD_COPIES 4
#endif
constexpr int CFD_COPY_STRIDE = 320 + 8;      // doubles: 2 624 bytes = 10 x 256 + 64: each copy starts 16 banks further on

static __global__ __launch_bounds__(256) void k_cfd(const uint64_t* __restrict__ guides, const uint64_t* __restrict__ offs,
                                                    int64_t n, int L, const double* __restrict__ mm,
                                                    double* __restrict__ out) {
    __shared__ double smm_all[GM_CFD_COPIES * CFD_COPY_STRIDE];
    for (int i = threadIdx.x; i < 320; i += blockDim.x) {
        const double v = mm[i];
#pragma unroll
        for (int c = 0; c < GM_CFD_COPIES; ++c) smm_all[c * CFD_COPY_STRIDE + i] = v;
    }
    __syncthreads();
    const double* smm = smm_all + (threadIdx.x & (GM_CFD_COPIES - 1)) * CFD_COPY_STRIDE;
    const int64_t base = (int64_t)blockIdx.x * (256 * CFD_PER_THREAD) + threadIdx.x;
    uint64_t a[CFD_PER_THREAD], b[CFD_PER_THREAD];
#pragma unroll
    for (int j = 0; j < CFD_PER_THREAD; ++j) {
        const int64_t idx = base + j * 256;
        a[j] = idx < n ? __builtin_nontemporal_load(guides + idx) : 0ull;
        b[j] = idx < n ? __builtin_nontemporal_load(offs + idx) : 0ull;
    }
#pragma unroll
    for (int j = 0; j < CFD_PER_THREAD; ++j) {
        const int64_t idx = base + j * 256;
        if (idx < n) __builtin_nontemporal_store(cfd_product(a[j], b[j], L, smm), out + idx);
    }
}

// ---------------------------------------------------------------------------------------------
// neighbour lists <-> one 32-bit key per neighbour ((dist << 27) | id, 0xffffffff = no neighbour):
// the form that crosses xGMI in the multi-GPU all-gather (one tensor, 4 B per neighbour instead of
// ids + dists + counts)
// ---------------------------------------------------------------------------------------------
static __global__ void k_pack_keys(const int32_t* __restrict__ ids, const int32_t* __restrict__ dists, int64_t n,
                                   uint32_t* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t id = ids[i], d = dists[i];
    keys[i] = id < 0 ? 0xffffffffu : (((uint32_t)d << ID_BITS) | (uint32_t)id);
}

static __global__ void k_unpack_keys(const uint32_t* __restrict__ keys, int64_t n_q, int k,
                                     int32_t* __restrict__ ids, int32_t* __restrict__ dists,
                                     int32_t* __restrict__ counts) {
    int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= n_q) return;
    int cnt = 0;
    for (int s = 0; s < k; ++s) {
        const uint32_t key = keys[qi * k + s];
        const bool ok = key != 0xffffffffu;
        ids[qi * k + s] = ok ? (int32_t)(key & ID_MASK) : -1;
        dists[qi * k + s] = ok ? (int32_t)(key >> ID_BITS) : -1;
        cnt += ok ? 1 : 0;
    }
    counts[qi] = cnt;
}

}  // namespace gm
