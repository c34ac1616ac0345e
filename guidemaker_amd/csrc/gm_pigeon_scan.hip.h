// gm_pigeon_scan.hip.h -- K1P: pigeonhole-partitioned exact Hamming k-NN scan for gfx950 (kernels; the host side
// is gm_pigeon.hip).
//
// The guide is cut into B blocks of positions.  Two sequences with at most r mismatches agree exactly on at least
// one of r + 1 disjoint blocks, so for block b it is enough to compare a candidate with the pool targets that carry
// the same bases on block b (its bucket); with the one- (two-) substitution neighbour buckets of a block scanned as
// well, the block covers every pair with up to one (two) mismatches on it, and B blocks with such radii t_b serve
// r = sum(t_b + 1) - 1 (gm_pg_types.h gm_pg_layout; gm_pigeon.hip gm_pg_plan).  The pool is stored B times, stable-sorted by
// the block's value (ids ascend inside a bucket), every bucket padded to whole LDS-DMA units, with one one-hot
// table per copy; candidates are grouped by the same value into chunks of <= 64 that never straddle a bucket.
//
// Workgroup (one wave) = one chunk of candidates (lane = candidate) against a RUN of target buckets of one copy
// (the chunk's own bucket and / or some of its neighbour buckets): K1L's per-lane LDS look-ups, carry-save adder
// trees and bit-sliced comparator over the buckets' 64-target groups, streamed through a double-buffered LDS window
// by global_load_lds DMA that runs across bucket boundaries.  Inside a bucket hits are frequent (the pairs already
// share a block: about one per 64 x 64 group on an E. coli sized pool), so the hit path is built to cost the wave
// as little as possible:
//   * ENQUEUE (per group with a hit, ~12 instructions, no loop, no divergence beyond one predicated LDS write):
//     every lane with a non-zero 64-bit hit word pushes (group, lane, hit word) into a ring in LDS at a position
//     from the wave's ballot prefix;
//   * FLUSH (when 64 entries are queued, and at the end): lane e takes entry e -- all 64 lanes busy -- , walks its
//     hit bits (almost always one), gathers the 16-byte record of the target {planes, pool id} from the sorted
//     copy, recomputes the distance with one xor / or / popcount, takes a slot in the candidate's hit list from a
//     counter in LDS and stores (distance << 27 | pool id) to the list in HBM.  A list has one writer (this
//     workgroup); a candidate's row of the hit buffer holds the counts of all its lists first, then their keys, so the
//     merge reads one header and only the key runs that are not empty.
// A pair that several blocks find (it has few mismatches on more than one of them) is REPORTED once, by the first of
// those blocks: the flush counts, on the mismatch word it has anyway, the mismatches on every earlier block, and leaves
// the entry out of the list if one of them is within that block's radius (S1: 25 -> 17.7 list entries per candidate;
// the merge sees no duplicates).  It still counts for the bounds below -- it is a real target at that distance.
// Bound tightening needs no sorted list: per candidate the flush counts the hits closer than r and closer than
// r - 1; once k of them are known nothing at distance r (r - 1) can enter the top k.  k = 1 follows the best distance.
// Id cut-off (CUT = true; gm_pigeon.hip pg_cut_fraction): when the pool holds many more targets at distance exactly r than a
// candidate needs (k = 20 of 80 on S4), only those with id < cut_id can be among the k best -- ties go to the smaller id --
// as long as enough of them exist.  Buckets are in id order, so the groups of a bucket behind its last id < cut_id are
// evaluated under the bound r instead of r + 1 (their distance-r targets never become hits), and the flush drops
// what the boundary group lets through.  A list that fills is exact as before (everything below its k-th key was
// enumerated); one that does not makes the row incomplete, and such rows take the plain re-scan.
// k_merge_hits folds the lists of a candidate (duplicates -- pairs that share several blocks -- taken once) into
// the top k; a candidate whose merged list is full is exact (every target within r was enumerated); the others,
// and those with an overflowed list, are collected for the plain scan (gmhip.hip).
#pragma once
#include "gm_lookup.hip.h"
#include "gm_pg_types.h"

namespace gm {

constexpr uint32_t PG_INC_OVERFLOW = 0x80000000u;      // flag on an entry of the incomplete-row list: a hit list overflowed
constexpr int PG_QCAP = 128;          // ring of queued hit words per wave (a flush takes 64, a group adds <= 64)

// dynamic LDS of k_hamming_scan_pg<L>: two DMA buffers of two groups each, the ring, per-candidate state
__host__ __device__ constexpr int pg_pair_words(int L) { return 2 * lk_group_words(L); }
__host__ __device__ constexpr size_t pg_lds_bytes(int L) {
    return sizeof(uint32_t) * (size_t)(2 * pg_pair_words(L) + PG_QCAP * 3 + WAVE * 2 + WAVE * 4);
}

// target bucket e of a candidate bucket v of a block of `len` bases: 0 = v itself; 1 + 3 p + a = v with the base at
// block position p replaced by the a-th of the three other bases; beyond 3 len: the two-substitution neighbours,
// 1 + 3 len + 9 pair + 3 a1 + a2 with pair = rank of (p1 < p2).  add = number of substituted positions the copy's
// table leaves out (the scan cannot see those mismatches; the comparator bound accounts for them).
__host__ __device__ __forceinline__ int pg_subst(int v, int pidx, int alt) {
    const int code = (v >> (2 * pidx)) & 3;
    return (v & ~(3 << (2 * pidx))) | (((code + 1 + alt) & 3) << (2 * pidx));
}
__host__ __device__ __forceinline__ int pg_target_bucket(int v, int e, int len, int skip_len, int& add) {
    add = 0;
    if (e == 0) return v;
    if (e <= 3 * len) {
        const int ne = e - 1, pidx = ne / 3;
        add = pidx < skip_len ? 1 : 0;
        return pg_subst(v, pidx, ne % 3);
    }
    const int e2 = e - 1 - 3 * len;
    int pi = e2 / 9;
    const int a1 = (e2 % 9) / 3, a2 = e2 % 3;
    int p1 = 0;
    while (pi >= len - 1 - p1) {      // unrank the pair (p1 < p2)
        pi -= len - 1 - p1;
        ++p1;
    }
    const int p2 = p1 + 1 + pi;
    add = (p1 < skip_len ? 1 : 0) + (p2 < skip_len ? 1 : 0);
    return pg_subst(pg_subst(v, p1, a1), p2, a2);
}

// walks the 2-group DMA units ("pairs") of the non-empty target buckets of a chunk, in order
struct PgIter {
    int e, g0, gc, pi, add, cutg;
};
// uncut (CUT only) = how many of the unit's groups still hold an id below the cut-off (they come first)
template <bool CUT>
__device__ __forceinline__ bool pg_next(const gm_pg_blocks& P, const gm_pg_chunk& ch, PgIter& it, int& g, int& ng,
                                        int& add, int& uncut) {
    if (2 * (it.pi + 1) < it.gc) {
        ++it.pi;
    } else {
        do {
            if (++it.e >= ch.e_hi) return false;
            const int vt = pg_target_bucket(ch.v, it.e, P.blen[ch.b], P.skiplen[ch.b], it.add);
            const int2 d = P.bdesc[ch.b][vt];      // one scalar 8-byte load
            it.g0 = d.x;
            it.gc = d.y;
            if constexpr (CUT) it.cutg = P.bcut[ch.b][vt];
        } while (it.gc <= 0);
        it.pi = 0;
    }
    g = it.g0 + 2 * it.pi;
    ng = it.gc - 2 * it.pi < 2 ? it.gc - 2 * it.pi : 2;
    add = it.add;
    if constexpr (CUT) {
        const int left = it.cutg - 2 * it.pi;
        uncut = left < 0 ? 0 : (left > 2 ? 2 : left);
    } else {
        uncut = 2;
    }
    return true;
}

// one DMA unit = two consecutive groups of the table (64 L bytes), straight into LDS
template <int L>
__device__ __forceinline__ void pg_stage(const uint32_t* __restrict__ table, int g, uint32_t* dst, int lane) {
    constexpr int UNITS = pg_pair_words(L) / 4;       // 16-byte units
    constexpr int ROUNDS = (UNITS + WAVE - 1) / WAVE;
    // wave-uniform base + a 32-bit unsigned lane offset: the load takes the base from SGPRs (saddr form) and no per-lane
    // 64-bit pointer has to live across the scan loop
    const uint64_t p64 = reinterpret_cast<uint64_t>(table + (int64_t)g * lk_group_words(L));
    const uint32_t p_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p64 >> 32));      // (the builtin returns int:
    const uint32_t p_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p64);                // no sign extension)
    const char* src = reinterpret_cast<const char*>(((uint64_t)p_hi << 32) | (uint64_t)p_lo);
    // (formed by an opaque instruction at every call: left to itself the compiler hoists the zero-extended 64-bit
    // offset out of the scan loop as a register pair, and at 64 VGPRs that pair is spilled and reloaded per stage)
    uint32_t lane_off;
    asm volatile("v_lshlrev_b32 %0, 4, %1" : "=v"(lane_off) : "v"(lane));
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = r * WAVE + lane;
        if (u < UNITS)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint32_t*>(src + r * WAVE * 16 + lane_off),
                                             dst + r * WAVE * 4, 16, 0, 0);
    }
}

struct PgQueue {
    uint32_t* tag;      // [PG_QCAP]  group << 6 | candidate lane
    uint2* mask;        // [PG_QCAP]  hit bits of the group's 64 targets
    uint2* qpl;         // [WAVE]     candidate planes
    int* qrow;          // [WAVE]     candidate rows
    int* cnt;           // [WAVE]     hits appended so far
    uint32_t* c12;      // [WAVE]     distinct targets known with d < thr0 - 1 (low half) / d < thr0 - 2 (high half);
                        //            k == 1: the best distance so far (whole word)
    uint32_t* c34;      // [WAVE]     ... with d < thr0 - 3 / d < thr0 - 4
    int head, tail;     // wave-uniform ring positions (monotonic)
};

// FLUSH: n (<= 64, wave-uniform) queued hit words are turned into list entries, one per lane.
__device__ __forceinline__ void pg_flush(PgQueue& Q, int n, int lane, const uint4* __restrict__ recs,
                                         uint32_t* __restrict__ lists, int off_a, int off_b, int lcap, int row_words, int k, int thr0,
                                         uint32_t cut_id = 0xffffffffu, int n_before = 0,
                                         const uint32_t* bmask = nullptr, const int* brad = nullptr) {
    uint32_t tg = 0;
    uint2 m = make_uint2(0u, 0u);
    if (lane < n) {
        const int s = (Q.head + lane) & (PG_QCAP - 1);
        tg = Q.tag[s];
        m = Q.mask[s];
    }
    Q.head += n;
    const int cl = (int)(tg & 63u);
    const int64_t pos0 = (int64_t)(tg >> 6) * LK_GROUP;
    const uint2 cq = Q.qpl[cl];
    const uint32_t rw = (uint32_t)Q.qrow[cl];      // row number; top bit = the candidate takes the id cut-off
    const int64_t row = (int64_t)(rw & 0x7fffffffu);
    const uint32_t cut_c = (rw >> 31) != 0u ? cut_id : 0xffffffffu;
    uint64_t hm = ((uint64_t)m.y << 32) | (uint64_t)m.x;
    uint32_t inc12 = 0u, inc34 = 0u;      // this entry's owned hits below the four bound levels (16-bit fields)
    int best = 64;
    while (hm != 0ull) {
        const int j = __builtin_ctzll(hm);
        hm &= hm - 1ull;
        const uint4 t = recs[pos0 + j];
        const uint32_t mism = (cq.x ^ t.x) | (cq.y ^ t.y);      // one bit per mismatching position
        const int d = __popc(mism);
        if (d == thr0 - 1 && t.z >= cut_c) continue;      // id cut-off: at distance r only ids below it count
        bool owned = true;      // no earlier block finds this pair
        for (int b = 0; b < n_before; ++b) owned = owned && __popc(mism & bmask[b]) > brad[b];
        best = d < best ? d : best;
        // (every hit counts towards the bounds, reported by this list or not: it is a real target at that distance)
        inc12 += (d < thr0 - 1 ? 1u : 0u) + (d < thr0 - 2 ? 0x10000u : 0u);
        inc34 += (d < thr0 - 3 ? 1u : 0u) + (d < thr0 - 4 ? 0x10000u : 0u);
        if (!owned) continue;
        const int slot = atomicAdd(&Q.cnt[cl], 1);
        // (the first four keys of every list of a row sit together at the row's start, the rest list by list)
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 1)      // timing build only (results WRONG): no list stores
        (void)lists;
#else
        if (slot < lcap) lists[row * row_words + (slot < 4 ? off_a : off_b) + slot] = ((uint32_t)d << ID_BITS) | t.z;
#endif
    }
    if (k == 1) {
        if (best < 64) atomicMin(&Q.c12[cl], (uint32_t)best);
    } else if ((inc12 | inc34) != 0u) {
        // one packed LDS add per entry and counter pair; a field that has reached 32 (k <= 20) takes no more, and an
        // entry adds at most 255 per field: 31 + 64 x 255 stays inside 16 bits
        const uint32_t o12 = Q.c12[cl], o34 = Q.c34[cl];
        uint32_t a0 = inc12 & 0xffffu, a1 = inc12 >> 16, a2 = inc34 & 0xffffu, a3 = inc34 >> 16;
        a0 = (o12 & 0xffffu) >= 32u ? 0u : (a0 > 255u ? 255u : a0);
        a1 = (o12 >> 16) >= 32u ? 0u : (a1 > 255u ? 255u : a1);
        a2 = (o34 & 0xffffu) >= 32u ? 0u : (a2 > 255u ? 255u : a2);
        a3 = (o34 >> 16) >= 32u ? 0u : (a3 > 255u ? 255u : a3);
        if ((a0 | a1) != 0u) atomicAdd(&Q.c12[cl], a0 | (a1 << 16));
        if ((a2 | a3) != 0u) atomicAdd(&Q.c34[cl], a2 | (a3 << 16));
    }
    __syncthreads();      // single-wave workgroup: orders the LDS atomics before the reads of the counters below
}

// bound of this lane's candidate from the counters the flushes keep (the counts are nested: d < thr0 - 4 implies
// d < thr0 - 3 ...): with k distinct targets known closer than t nothing at distance >= t can enter the top k
__device__ __forceinline__ int pg_bound(const PgQueue& Q, int lane, int k, int thr0) {
    if (k == 1) return (int)Q.c12[lane] + 1;      // ties at the best distance stay in: the smallest id among them wins in the merge
    const uint32_t a = Q.c12[lane], b = Q.c34[lane];
    const uint32_t kk = (uint32_t)k;
    return thr0 - ((a & 0xffffu) >= kk ? 1 : 0) - ((a >> 16) >= kk ? 1 : 0) - ((b & 0xffffu) >= kk ? 1 : 0) - ((b >> 16) >= kk ? 1 : 0);
}

// evaluate one 64-target group from LDS (word offset WOFF) and queue the lanes that have hits
template <int L, int WOFF, class QT>
__device__ __forceinline__ void pg_group(const uint32_t* const (&ptr)[L], const uint32_t (&Tb)[5], int g, int lane,
                                         QT& Q) {
    LkGroupRegs<L> r;
    lk_read_group<L, WOFF>(r, ptr);
    uint32_t c0[5], c1[5];
    const uint32_t h0 = lk_eval_half<L>(r.m0, Tb, c0);
    const uint32_t h1 = lk_eval_half<L>(r.m1, Tb, c1);
    const bool has = (h0 | h1) != 0u;
    const uint64_t bal = __builtin_amdgcn_ballot_w64(has);
    if (bal != 0ull) {
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (has) {
            const int s = (Q.tail + (int)before) & (PG_QCAP - 1);
            Q.tag[s] = ((uint32_t)g << 6) | (uint32_t)lane;
            Q.mask[s] = make_uint2(h0, h1);
        }
        Q.tail += __builtin_popcountll(bal);
    }
}

// lists: the hit buffer (see gm_pg_blocks::loff / lcap)
template <int L, bool CUT>
__global__ __launch_bounds__(WAVE, (L <= 20 ? 8 : 1)) void k_hamming_scan_pg(
    gm_pg_blocks P, const gm_pg_chunk* __restrict__ chunks, const int* __restrict__ c_begin,
    const int* __restrict__ c_end, const uint2* __restrict__ q, int64_t n_q, int k, int thr0,
    uint32_t* __restrict__ lists) {
    extern __shared__ __align__(16) uint32_t smem[];
    constexpr int GW = lk_group_words(L);
    constexpr int PW = pg_pair_words(L);
    // XCD-aware order: workgroups are dealt to the 8 XCDs round-robin; give every XCD runs of P.run consecutive
    // chunk-table entries (the chunks of one candidate bucket share their target buckets: same L2)
    // (one launch per run of blocks with the same number of evaluated positions: entries [*c_begin, *c_end))
    const int first = *c_begin, n = *c_end - first;
    const int wg = (int)blockIdx.x, xcd = wg & 7, slot = wg >> 3;
    const int entry = (slot / P.run) * (8 * P.run) + xcd * P.run + (slot % P.run);
    if (entry >= n) return;
    const gm_pg_chunk ch = chunks[first + entry];
    const uint32_t* __restrict__ table = P.table[ch.b];
    const uint4* __restrict__ recs = P.recs[ch.b];
    uint32_t* tiles = smem;                                     // [2][PW]
    PgQueue Q;
    Q.tag = smem + 2 * PW;
    Q.mask = reinterpret_cast<uint2*>(Q.tag + PG_QCAP);
    Q.qpl = Q.mask + PG_QCAP;
    Q.qrow = reinterpret_cast<int*>(Q.qpl + WAVE);
    Q.cnt = Q.qrow + WAVE;
    Q.c12 = reinterpret_cast<uint32_t*>(Q.cnt + WAVE);
    Q.c34 = Q.c12 + WAVE;
    Q.head = Q.tail = 0;
    const int lane = threadIdx.x;
    const bool valid = lane < ch.cn;
    const int64_t qi = valid ? (int64_t)P.cperm[ch.b][ch.c0 + lane] : 0;
    const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);
    // id cut-off per candidate: only rows whose neighbourhood is dense enough take it (cflag, k_cand_cutflag)
    const bool lane_cut = CUT && valid && P.cflag[qi] != 0;
    Q.qpl[lane] = qv;
    Q.qrow[lane] = (int)((uint32_t)qi | (lane_cut ? 0x80000000u : 0u));
    Q.cnt[lane] = 0;
    Q.c12[lane] = k == 1 ? (uint32_t)thr0 : 0u;
    Q.c34[lane] = 0u;

    // L = positions evaluated: the copy's table leaves out P.skiplen[b] positions of its own block (all pairs of
    // the bucket agree there, or differ on exactly one of them: `add`), so table position p is sequence position p
    // (+ skip_len behind the gap)
    const int skip_start = P.skip[ch.b], skip_len = P.skiplen[ch.b];
    const uint32_t* ptr[L];
#pragma unroll
    for (int p = 0; p < L; ++p) {
        const int sp = p >= skip_start ? p + skip_len : p;
        const uint32_t code = ((qv.x >> sp) & 1u) | (((qv.y >> sp) & 1u) << 1);
        ptr[p] = tiles + p * 8 + (int)code * 2;
    }
    const int lcap = P.lcap[ch.lst], row_words = P.row_words;
    // keys of candidate row: slots 0 .. 3 at lists + row * row_words + 4 lst, slot s >= 4 at ... + loff[lst] + (s - 4)
    uint32_t* __restrict__ mylists = lists;
    const int off_a = 4 * ch.lst, off_b = P.loff[ch.lst] - 4;
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 2)      // timing build only (results WRONG): no hit is accepted
    int thr = 0;
#else
    int thr = valid ? thr0 : 0;              // accept d < thr
#endif
    uint32_t Tb[5];
    // the blocks before this one (position masks, radii): a pair one of them finds is theirs to report
    uint32_t bmask[GM_PG_MAXB];
    int brad[GM_PG_MAXB];
#pragma unroll
    for (int b = 0; b < GM_PG_MAXB; ++b) {
        bmask[b] = b < ch.b ? ((1u << P.blen[b]) - 1u) << P.skip[b] : 0u;
        brad[b] = b < ch.b ? P.trad[b] : 64;
    }

    PgIter it{ch.e_lo - 1, 0, 0, 0, 0, 0};
    int gA = 0, ngA = 0, addA = 0, ucA = 2, gB = 0, ngB = 0, addB = 0, ucB = 2;
    bool okA = pg_next<CUT>(P, ch, it, gA, ngA, addA, ucA), okB = false;
    if (okA) pg_stage<L>(table, gA, tiles, lane);
    const uint32_t cut_id = CUT ? P.cut_id : 0xffffffffu;
    // comparator bound of the next group: the lane's bound, minus the mismatches the copy's table cannot see (add), and
    // behind a bucket's cut-off point at most r (thr0 - 1): distance-r targets there have ids that cannot matter
    int cur = -1;      // (add << 1 | cut) the comparator words were formed for; -1 = stale
#define GM_PG_BOUND(ADD, CUTF)                                                              \
    {                                                                                     \
        const int want = ((ADD) << 1) | ((CUTF) ? 1 : 0);                                 \
        if (want != cur) {                                                                \
            cur = want;                                                                   \
            const int t_ = (CUT && (CUTF) && lane_cut && thr > thr0 - 1) ? thr0 - 1 : thr; \
            lk_set_bound(Tb, L, t_ - (ADD));                                              \
        }                                                                                 \
    }
    // flush when a full wave of entries is queued; afterwards the bounds may have tightened
#define GM_PG_AFTER_GROUP()                                                       \
    if (Q.tail - Q.head >= WAVE) {                                                \
        pg_flush(Q, WAVE, lane, recs, mylists, off_a, off_b, lcap, row_words, k, thr0, cut_id, ch.b, bmask, brad); \
        const int nt = pg_bound(Q, lane, k, thr0);                                \
        if (valid && nt < thr) thr = nt;                                          \
        cur = -1;                                                                 \
    }
    while (okA) {
        okB = pg_next<CUT>(P, ch, it, gB, ngB, addB, ucB);
        lk_stage_wait();                                        // unit A has landed in buffer 0
        if (okB) pg_stage<L>(table, gB, tiles + PW, lane);
        GM_PG_BOUND(addA, ucA < 1)
        pg_group<L, 0>(ptr, Tb, gA, lane, Q);
        GM_PG_AFTER_GROUP()
        if (ngA > 1) {
            GM_PG_BOUND(addA, ucA < 2)
            pg_group<L, GW>(ptr, Tb, gA + 1, lane, Q);
            GM_PG_AFTER_GROUP()
        }
        if (!okB) break;
        okA = pg_next<CUT>(P, ch, it, gA, ngA, addA, ucA);
        lk_stage_wait();                                        // unit B has landed in buffer 1
        if (okA) pg_stage<L>(table, gA, tiles, lane);
        GM_PG_BOUND(addB, ucB < 1)
        pg_group<L, PW>(ptr, Tb, gB, lane, Q);
        GM_PG_AFTER_GROUP()
        if (ngB > 1) {
            GM_PG_BOUND(addB, ucB < 2)
            pg_group<L, PW + GW>(ptr, Tb, gB + 1, lane, Q);
            GM_PG_AFTER_GROUP()
        }
    }
#undef GM_PG_AFTER_GROUP
#undef GM_PG_BOUND
    while (Q.tail - Q.head > 0) {
        const int n_take = Q.tail - Q.head < WAVE ? Q.tail - Q.head : WAVE;
        pg_flush(Q, n_take, lane, recs, mylists, off_a, off_b, lcap, row_words, k, thr0, cut_id, ch.b, bmask, brad);
    }
    // every (list, candidate) pair has exactly one writer (the row number comes back from LDS: a 64-bit index kept in
    // registers across the scan loop is two VGPRs the loop does not have)
    if (lane < ch.cn) P.hdr[(int64_t)((uint32_t)Q.qrow[lane] & 0x7fffffffu) * P.hdr_words + ch.lst] = (uint32_t)Q.cnt[lane];
}

// ---------------------------------------------------------------------------------------------------------------
// K1P with the hit lists in REGISTERS (k_hamming_scan_pg4; gm_set_param "pg_hit" = 1).
//
// What the counters of the kernel above say on a real genome (profiles/r03: PAO1, ~72 targets within r per candidate):
// the hit path is 57 % of the launch, and it is latency, not issue: every hit is a scattered 4-byte store into a
// multi-GB hit buffer, every store stays in the wave's in-order vmcnt queue in front of the next table unit, the bound a
// workgroup works under is four counted levels, and a flush walks the hit bits of its fullest entry round by round.
// Here (measured steps: profiles/r04/NOTES.md):
//   * ENQUEUE is per HIT: a lane with hits pushes (position in the copy, its lane) for its lowest hit bit -- ballot
//     prefix as before -- and the few lanes with more bits go round again.  A ring entry is 5 bytes, and a flush
//     round needs no expansion: lane t takes entry t.
//   * every lane keeps ITS OWN candidate's k best keys in a sorted register list (RegList, as the plain scan K1L): the
//     comparator bound is the list's k-th distance (+ 1; ties go to the smaller id, decided on the key) -- exact, in
//     registers, no counters;
//   * FLUSH ROUND (64 queued hits, one per lane): read the entry, fetch the candidate's planes from the owner lane
//     (ds_bpermute), gather the target's record, distance by xor / or / popcount, drop what cannot beat the owner's
//     published k-th key, and DELIVER the key to the owner through a four-deep mailbox per candidate in LDS (slot from
//     a returning LDS add; a fifth key for one candidate waits a turn).  Every owner then drains its mailbox into its
//     register list: <= 4 sorted inserts of 2 KCAP min / max each, all lanes busy (an empty slot inserts the "no key"
//     sentinel, which falls through).  Two LDS round trips + one gather per round, whatever k; no atomics on list
//     slots, no capacity, no overflow: a repeat family costs inserts, not a re-scan of the row;
//   * no global store inside the scan loop: at the end every lane writes its list as one run into
//     keys[list][row][KCAP] (the raw list format of gm_lookup.hip.h), which k_merge_pg3 folds; the hit buffer, its
//     headers and the capacity model of the kernel above are not needed.  LDS per wave: 2 table units + 1.9 KB.
// A pair that several blocks find is offered to every one of their lists (each workgroup's bound then counts every hit
// it sees) and taken once by the merge (insert_unique on equal keys).
constexpr uint32_t PG4_NONE = 0xffffffffu;      // "no key" (= RegList::EMPTY: inserting it changes nothing)
constexpr int PG4_MAIL = 4;                     // mailbox slots per candidate and round
__host__ __device__ constexpr size_t pg4_lds_bytes(int L) {
    // tiles, ring (position u32 + lane u8), mailbox, mailbox counts, published k-th keys
    return sizeof(uint32_t) * (size_t)(2 * pg_pair_words(L) + PG_QCAP + PG_QCAP / 4 + PG4_MAIL * WAVE + WAVE + WAVE + 2 * WAVE);
}

struct Pg4Queue {
    uint32_t* pos;      // [PG_QCAP]  position of the hit target in the block-sorted copy
    uint8_t* ln;        // [PG_QCAP]  candidate lane
    uint32_t* mail;     // [PG4_MAIL][WAVE]  keys + 1 delivered to candidate c this round
    int* mc;            // [WAVE]     keys delivered to (or tried for) candidate c this round
    uint32_t* kth;      // [WAVE]     candidate c's k-th key + 1 as of its last drain (PG4_NONE: list not full)
    uint2* qpl;         // [WAVE]     candidate planes; bit 31 of .x = the candidate takes the id cut-off (in LDS, not
                        //            fetched from the owner lane by ds_bpermute: two VGPRs the loop does not have at 16 positions)
    int head, tail;     // wave-uniform ring positions (monotonic)
};

#ifdef GM_PG_STATS
#define GM_PG_STAT(i, v) do { if (lane == 0) st_[i] += (unsigned long long)(v); } while (0)
#else
#define GM_PG_STAT(i, v) do { } while (0)
#endif

// one round of the flush: the first <= 64 queued hits are resolved, one per lane, and delivered to their candidates'
// owner lanes
// A round in two halves, so that the record gather of one round is in flight while the next groups are evaluated
// (pg4_pipelined): ISSUE takes <= 64 entries off the ring and starts their gathers; FINISH -- at the next trigger, or at the
// end -- turns the records into keys and delivers them.  The bound a lane evaluates under lags one round (conservative: the key
// test at delivery is what decides).
struct Pg4Pending {
    int n;            // entries taken (wave-uniform); 0 = nothing pending
    int cl;           // the entry's candidate lane
    uint32_t tx, ty, tz;      // the target's record: planes, pool id
};
__device__ __forceinline__ void pg4_round_issue(Pg4Queue& Q, int lane, const uint4* __restrict__ recs, Pg4Pending& R) {
    const int avail = Q.tail - Q.head;
    R.n = avail < WAVE ? avail : WAVE;
    const int s = (Q.head + lane) & (PG_QCAP - 1);
    R.cl = 0;
    R.tx = R.ty = R.tz = 0u;
    if (lane < R.n) {
        const uint32_t pos = Q.pos[s];
        R.cl = (int)Q.ln[s];
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 4)      // timing build only (results WRONG): no record gather
        R.tx = pos;
        R.ty = pos * 31u;
        R.tz = pos * 977u & ID_MASK;
#else
        const uint4 t = recs[pos];
        R.tx = t.x;
        R.ty = t.y;
        R.tz = t.z;
#endif
    }
    Q.head += R.n;
}
template <int KCAP>
__device__ __forceinline__ void pg4_round_finish(Pg4Queue& Q, int lane, Pg4Pending& R, int thr0, uint32_t cut_id, RegList<KCAP>& list
#ifdef GM_PG_STATS
                                                 , unsigned long long (&st_)[8]
#endif
                                                 ) {
    const int n = R.n, cl = R.cl;
    R.n = 0;
    uint32_t x = PG4_NONE;
    if (lane < n) {
        const uint2 cq = Q.qpl[cl];
        const uint32_t mism = ((cq.x ^ R.tx) | (cq.y ^ R.ty)) & 0x7fffffffu;      // one bit per mismatching position
        const int d = __popc(mism);
        const uint32_t cut_c = (cq.x >> 31) != 0u ? cut_id : 0xffffffffu;      // id cut-off: at distance r only ids below it count
        const uint32_t key1 = (((uint32_t)d << ID_BITS) | R.tz) + 1u;
        if (!(d == thr0 - 1 && R.tz >= cut_c) && key1 < Q.kth[cl]) x = key1;
    }
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 1)      // timing build only (results WRONG): nothing is delivered
    x = PG4_NONE;
#endif
    GM_PG_STAT(0, 1);
    GM_PG_STAT(1, n);
#ifdef GM_PG_STATS
    { const int nd = __builtin_popcountll(__builtin_amdgcn_ballot_w64(x != PG4_NONE)); GM_PG_STAT(2, nd); }
#endif
    do {
        GM_PG_STAT(3, 1);
        if (x != PG4_NONE) {
            const int slot = atomicAdd(&Q.mc[cl], 1);
            if (slot < PG4_MAIL) {
                Q.mail[slot * WAVE + cl] = x;
                x = PG4_NONE;
            }
        }
        __syncthreads();      // single-wave workgroup: the deliveries are in LDS before the owners read
        const int got = Q.mc[lane];
        uint32_t kx[PG4_MAIL];
#pragma unroll
        for (int i = 0; i < PG4_MAIL; ++i) kx[i] = Q.mail[i * WAVE + lane];
        Q.mc[lane] = 0;
#pragma unroll
        for (int i = 0; i < PG4_MAIL; ++i) {
            const uint32_t v = i < got ? kx[i] : PG4_NONE;
            if (__builtin_amdgcn_ballot_w64(i < got) != 0ull) {
                GM_PG_STAT(6, 1);
                list.insert_raw(v);
            }
        }
        Q.kth[lane] = list.v[KCAP - 1];
        __syncthreads();
    } while (__builtin_amdgcn_ballot_w64(x != PG4_NONE) != 0ull);
}

// the hit words of one 64-target group from LDS (word offset WOFF)
template <int L, int WOFF>
__device__ __forceinline__ uint64_t pg4_eval(const uint32_t* const (&ptr)[L], const uint32_t (&Tb)[5]) {
    LkGroupRegs<L> r;
    lk_read_group<L, WOFF>(r, ptr);
    uint32_t c0[5], c1[5];
    const uint32_t h0 = lk_eval_half<L>(r.m0, Tb, c0);
    const uint32_t h1 = lk_eval_half<L>(r.m1, Tb, c1);
    return ((uint64_t)h1 << 32) | (uint64_t)h0;
}

// The descriptors of a chunk's target buckets, prefetched: lane j holds bucket e_base + j's {first group, groups | add << 28}
// (and, under the id cut-off, its groups below the cut) from ONE vector gather with all loads in flight; the scan's
// iterator then takes them with v_readlane -- no dependent scalar load per bucket in front of the next table unit's DMA
// (k_hamming_scan_pg's pg_next pays one per bucket: ~20 per workgroup with every target bucket of a chunk in one
// workgroup).  More than 64 buckets (a radius-2 block): the next 64 are fetched when the iterator gets there.
template <bool CUT>
struct Pg4Desc {
    int g0v, gcv, cutv;
    int e_base;
    __device__ __forceinline__ void load(const gm_pg_blocks& P, const gm_pg_chunk& ch, int base, int lane) {
        e_base = base;
        const int e = base + lane;
        g0v = 0;
        gcv = 0;
        cutv = 0;
        if (e < ch.e_hi) {
            int add;
            const int vt = pg_target_bucket(ch.v, e, P.blen[ch.b], P.skiplen[ch.b], add);
            const int2 d = P.bdesc[ch.b][vt];
            g0v = d.x;
            gcv = d.y > 0 ? (d.y | (add << 28)) : 0;
            if constexpr (CUT) cutv = P.bcut[ch.b][vt];
        }
    }
};
template <bool CUT>
__device__ __forceinline__ bool pg4_next(const gm_pg_blocks& P, const gm_pg_chunk& ch, Pg4Desc<CUT>& D, PgIter& it, int& g, int& ng,
                                         int& add, int& uncut, int lane) {
    if (2 * (it.pi + 1) < it.gc) {
        ++it.pi;
    } else {
        do {
            if (++it.e >= ch.e_hi) return false;
            if (it.e - D.e_base >= WAVE) D.load(P, ch, D.e_base + WAVE, lane);
            const int rel = it.e - D.e_base;
            const int gcw = __builtin_amdgcn_readlane(D.gcv, rel);
            it.g0 = __builtin_amdgcn_readlane(D.g0v, rel);
            it.gc = gcw & 0x0fffffff;
            it.add = (int)((unsigned)gcw >> 28);
            if constexpr (CUT) it.cutg = __builtin_amdgcn_readlane(D.cutv, rel);
        } while (it.gc <= 0);
        it.pi = 0;
    }
    g = it.g0 + 2 * it.pi;
    ng = it.gc - 2 * it.pi < 2 ? it.gc - 2 * it.pi : 2;
    add = it.add;
    if constexpr (CUT) {
        const int left = it.cutg - 2 * it.pi;
        uncut = left < 0 ? 0 : (left > 2 ? 2 : left);
    } else {
        uncut = 2;
    }
    return true;
}

// keys: [n_lists][n_q][KCAP] raw lists (see above)
#ifndef GM_PG4_W10
#define GM_PG4_W10 6      /* waves per SIMD asked for with a 5 / 10-key register list */
#endif
#ifndef GM_PG4_W20
#define GM_PG4_W20 4      /* ... with a 20-key list */
#endif
// the gather of a round stays in flight across the next groups where five more VGPRs fit under the occupancy the kernel has
#ifndef GM_PG4_PIPE
#define GM_PG4_PIPE 1
#endif
__host__ __device__ constexpr bool pg4_pipelined(int L, int KCAP) { return GM_PG4_PIPE != 0 && L <= 14 && KCAP == 10; }
__host__ __device__ constexpr int pg4_min_waves(int L, int KCAP) { return L > 20 ? 1 : (KCAP <= 1 ? 8 : (KCAP <= 10 ? GM_PG4_W10 : GM_PG4_W20)); }
template <int L, bool CUT, int KCAP>
__global__ __launch_bounds__(WAVE, pg4_min_waves(L, KCAP)) void k_hamming_scan_pg4(
    gm_pg_blocks P, const gm_pg_chunk* __restrict__ chunks, const int* __restrict__ c_begin,
    const int* __restrict__ c_end, const uint2* __restrict__ q, int64_t n_q, int k, int thr0,
    uint32_t* __restrict__ keys) {
    extern __shared__ __align__(16) uint32_t smem[];
    constexpr int GW = lk_group_words(L);
    constexpr int PW = pg_pair_words(L);
    const int first = *c_begin, n = *c_end - first;
    const int wg = (int)blockIdx.x, xcd = wg & 7, slot = wg >> 3;
    const int entry = (slot / P.run) * (8 * P.run) + xcd * P.run + (slot % P.run);
    if (entry >= n) return;
    const gm_pg_chunk ch = chunks[first + entry];
    const uint32_t* __restrict__ table = P.table[ch.b];
    const uint4* __restrict__ recs = P.recs[ch.b];
    uint32_t* tiles = smem;                                     // [2][PW]
    Pg4Queue Q;
    Q.pos = smem + 2 * PW;
    Q.ln = reinterpret_cast<uint8_t*>(Q.pos + PG_QCAP);
    Q.mail = Q.pos + PG_QCAP + PG_QCAP / 4;
    Q.mc = reinterpret_cast<int*>(Q.mail + PG4_MAIL * WAVE);
    Q.kth = reinterpret_cast<uint32_t*>(Q.mc + WAVE);
    Q.qpl = reinterpret_cast<uint2*>(Q.kth + WAVE);
    Q.head = Q.tail = 0;
    const int lane = threadIdx.x;
    const bool valid = lane < ch.cn;
    const int64_t qi = valid ? (int64_t)P.cperm[ch.b][ch.c0 + lane] : 0;
    const uint2 qv = valid ? q[qi] : make_uint2(0u, 0u);
    const bool lane_cut = CUT && valid && P.cflag[qi] != 0;
    RegList<KCAP> list;
    list.init(k, nullptr, 0);
    Q.mc[lane] = 0;
    Q.kth[lane] = list.v[KCAP - 1];
    Q.qpl[lane] = make_uint2(qv.x | (lane_cut ? 0x80000000u : 0u), qv.y);

    const int skip_start = P.skip[ch.b], skip_len = P.skiplen[ch.b];
    const uint32_t* ptr[L];
#pragma unroll
    for (int p = 0; p < L; ++p) {
        const int sp = p >= skip_start ? p + skip_len : p;
        const uint32_t code = ((qv.x >> sp) & 1u) | (((qv.y >> sp) & 1u) << 1);
        ptr[p] = tiles + p * 8 + (int)code * 2;
    }
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 2)      // timing build only (results WRONG): no hit is accepted
    int thr = 0;
#else
    int thr = valid ? thr0 : 0;              // accept d < thr
#endif
    uint32_t Tb[5];
    Pg4Pending R{0, 0, 0u, 0u, 0u};
    PgIter it{ch.e_lo - 1, 0, 0, 0, 0, 0};
    Pg4Desc<CUT> D;
    D.load(P, ch, ch.e_lo, lane);
    int gA = 0, ngA = 0, addA = 0, ucA = 2, gB = 0, ngB = 0, addB = 0, ucB = 2;
    bool okA = pg4_next<CUT>(P, ch, D, it, gA, ngA, addA, ucA, lane), okB = false;
    if (okA) pg_stage<L>(table, gA, tiles, lane);
    const uint32_t cut_id = CUT ? P.cut_id : 0xffffffffu;
    int cur = -1;      // (add << 1 | cut) the comparator words were formed for; -1 = stale
#ifdef GM_PG_STATS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GM_PG_STARG , st_
#else
#define GM_PG_STARG
#endif
#define GM_PG_BOUND(ADD, CUTF)                                                              \
    {                                                                                     \
        const int want = ((ADD) << 1) | ((CUTF) ? 1 : 0);                                 \
        if (want != cur) {                                                                \
            cur = want;                                                                   \
            const int t_ = (CUT && (CUTF) && lane_cut && thr > thr0 - 1) ? thr0 - 1 : thr; \
            lk_set_bound(Tb, L, t_ - (ADD));                                              \
        }                                                                                 \
    }
    // push the group's hits, lowest bit of every lane first; a round whenever 64 hits are queued (afterwards the bounds may
    // have tightened: the rest of this group's hit bits were formed under the old one and are dropped by the key test)
#define GM_PG_GROUP(WOFF, G)                                                                                   \
    {                                                                                                          \
        uint64_t hm = pg4_eval<L, WOFF>(ptr, Tb);                                                              \
        uint64_t bal = __builtin_amdgcn_ballot_w64(hm != 0ull);                                                \
        while (bal != 0ull) {                                                                                  \
            if (hm != 0ull) {                                                                                  \
                const uint32_t before =                                                                        \
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u)); \
                const int s_ = (Q.tail + (int)before) & (PG_QCAP - 1);                                         \
                Q.pos[s_] = ((uint32_t)(G) << 6) + (uint32_t)__builtin_ctzll(hm);                              \
                Q.ln[s_] = (uint8_t)lane;                                                                      \
                hm &= hm - 1ull;                                                                               \
            }                                                                                                  \
            Q.tail += __builtin_popcountll(bal);                                                               \
            if (Q.tail - Q.head >= WAVE) {                                                                     \
                if (pg4_pipelined(L, KCAP)) {                                                                  \
                    if (R.n > 0) pg4_round_finish<KCAP>(Q, lane, R, thr0, cut_id, list GM_PG_STARG);           \
                    pg4_round_issue(Q, lane, recs, R);                                                         \
                } else {                                                                                       \
                    pg4_round_issue(Q, lane, recs, R);                                                         \
                    pg4_round_finish<KCAP>(Q, lane, R, thr0, cut_id, list GM_PG_STARG);                        \
                }                                                                                              \
                const int nt = list.v[KCAP - 1] == PG4_NONE ? thr : (int)((list.v[KCAP - 1] - 1u) >> ID_BITS) + 1; \
                if (nt < thr) thr = nt;                                                                        \
                cur = -1;                                                                                      \
            }                                                                                                  \
            bal = __builtin_amdgcn_ballot_w64(hm != 0ull);                                                     \
        }                                                                                                      \
    }
    while (okA) {
        okB = pg4_next<CUT>(P, ch, D, it, gB, ngB, addB, ucB, lane);
        lk_stage_wait();                                        // unit A has landed in buffer 0
        if (okB) pg_stage<L>(table, gB, tiles + PW, lane);
        GM_PG_BOUND(addA, ucA < 1)
        GM_PG_GROUP(0, gA)
        if (ngA > 1) {
            GM_PG_BOUND(addA, ucA < 2)
            GM_PG_GROUP(GW, gA + 1)
        }
        if (!okB) break;
        okA = pg4_next<CUT>(P, ch, D, it, gA, ngA, addA, ucA, lane);
        lk_stage_wait();                                        // unit B has landed in buffer 1
        if (okA) pg_stage<L>(table, gA, tiles, lane);
        GM_PG_BOUND(addB, ucB < 1)
        GM_PG_GROUP(PW, gB)
        if (ngB > 1) {
            GM_PG_BOUND(addB, ucB < 2)
            GM_PG_GROUP(PW + GW, gB + 1)
        }
    }
#undef GM_PG_GROUP
#undef GM_PG_BOUND
    if (R.n > 0) pg4_round_finish<KCAP>(Q, lane, R, thr0, cut_id, list GM_PG_STARG);
    while (Q.tail - Q.head > 0) {
        GM_PG_STAT(4, 1);
        pg4_round_issue(Q, lane, recs, R);
        pg4_round_finish<KCAP>(Q, lane, R, thr0, cut_id, list GM_PG_STARG);
    }
#ifdef GM_PG_STATS
    if (lane == 0) {
        st_[5] = 1;
        for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(P.hdr) + i, st_[i]);
    }
#endif
    // every (list, candidate) pair has exactly one writer (the row number is fetched again: a 64-bit index kept in
    // registers across the scan loop is two VGPRs the loop does not have)
    if (lane < ch.cn) {
        const int64_t row = (int64_t)P.cperm[ch.b][ch.c0 + lane];
        uint32_t* __restrict__ out = keys + ((int64_t)ch.lst * n_q + row) * KCAP;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) out[s] = list.v[s];
    }
}

// Appends `value` of the threads with take == true to out[] (order irrelevant): the workgroup's threads are ranked by
// wave ballots + an LDS scan of the wave totals, ONE atomicAdd per workgroup on *counter hands out the range.  Every
// thread of the workgroup calls it (it contains barriers); blockDim.x = a multiple of 64, at most 1024.
__device__ __forceinline__ void gm_block_append(bool take, int32_t value, int32_t* __restrict__ out, int* __restrict__ counter) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const uint64_t bal = __builtin_amdgcn_ballot_w64(take);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) s_wave[wave] = __builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n_waves = (int)((blockDim.x + 63u) >> 6);
        int tot = 0;
        for (int w = 0; w < n_waves; ++w) {
            const int c = s_wave[w];
            s_wave[w] = tot;
            tot += c;
        }
        s_base = tot != 0 ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    if (take) out[s_base + s_wave[wave] + before] = value;
}

// The K1P merges list their incomplete rows WITHOUT device-wide atomics: a workgroup ranks its entries (ballots + an LDS scan,
// as gm_block_append), writes them to its own slot of `stash` (PG_STASH_ROWS entries per workgroup) and its count and
// counters to `meta` (the counts, dense, then PG_STASH_META words per workgroup: chist[24], overflowed rows); k_compact_inc packs the slots.
// (One returning atomic on the list cursor + up to 25 on the counters per workgroup were most of a memory-bound merge:
// S3's first level 63 -> 18 us, PAO1's 89 -> 45, S1's 45 -> 30 without them.)
constexpr int PG_STASH_ROWS = 256;      // = the merge kernels' workgroup size
constexpr int PG_STASH_META = 32;
__device__ __forceinline__ void pg_block_stash(bool take, int32_t value, const int* s_hist24, const int* s_over, int32_t* __restrict__ stash,
                                               int* __restrict__ meta) {
    __shared__ int s_wave[16];
    const uint64_t bal = __builtin_amdgcn_ballot_w64(take);
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    if (lane == 0) s_wave[wave] = __builtin_popcountll(bal);
    __syncthreads();      // (also orders the caller's LDS counters)
    const int n_waves = (int)((blockDim.x + 63u) >> 6);
    int base = 0, tot = 0;
    for (int w = 0; w < n_waves; ++w) {
        const int c = s_wave[w];
        base += w < wave ? c : 0;
        tot += c;
    }
    // meta: [gridDim.x] counts (dense: k_compact_inc sums runs of them), then PG_STASH_META counter words per workgroup
    int* __restrict__ m = meta + gridDim.x + (int64_t)blockIdx.x * PG_STASH_META;
    if (take) stash[(int64_t)blockIdx.x * PG_STASH_ROWS + base + before] = value;
    if (threadIdx.x == 0) meta[blockIdx.x] = tot;
    if (threadIdx.x >= 1 && threadIdx.x <= 24) m[threadIdx.x] = s_hist24[threadIdx.x - 1];
    if (threadIdx.x == 25) m[25] = *s_over;
}
// the slots of PG_COMPACT_WGS consecutive merge workgroups per workgroup -> inc[0 .. n_inc), *inc_count, *over_count, chist[]
constexpr int PG_COMPACT_WGS = 16;      // (64: 10 -> 28 us)
static __global__ __launch_bounds__(256) void k_compact_inc(const int32_t* __restrict__ stash, const int* __restrict__ meta, int n_wg,
                                                     int32_t* __restrict__ inc, int* __restrict__ inc_count, int* __restrict__ over_count,
                                                     int* __restrict__ chist) {
    __shared__ int s_red[4], s_off[PG_COMPACT_WGS + 1], s_cnt[PG_COMPACT_WGS];
    const int tid = threadIdx.x, w0 = blockIdx.x * PG_COMPACT_WGS;
    int before = 0;      // entries of the workgroups in front of this one's
    // (w0 is a multiple of 16: whole int4s; four of them in flight per thread -- one 4-byte load per loop iteration made this
    // walk over a few thousand counts most of the kernel's 9 .. 13 us)
    const int4* __restrict__ m4 = reinterpret_cast<const int4*>(meta);
    for (int w = tid; w < w0 / 4; w += 4 * 256) {
        int4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = w + j * 256 < w0 / 4 ? m4[w + j * 256] : make_int4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) before += v[j].x + v[j].y + v[j].z + v[j].w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = before;
    if (tid < PG_COMPACT_WGS) s_cnt[tid] = w0 + tid < n_wg ? meta[w0 + tid] : 0;      // (all of them in flight at once)
    __syncthreads();
    if (tid == 0) {
        int run = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        for (int j = 0; j < PG_COMPACT_WGS; ++j) {
            s_off[j] = run;
            run += s_cnt[j];
        }
        s_off[PG_COMPACT_WGS] = run;
        if (w0 + PG_COMPACT_WGS >= n_wg) *inc_count = run;      // (the last workgroup: the total)
    }
    if (tid >= 64 + 1 && tid <= 64 + 25) {      // counters of this workgroup's slots (a wave of its own: not behind thread 0's scan)
        const int c = tid - 64;
        int sum = 0;
        for (int j0 = 0; j0 < PG_COMPACT_WGS; j0 += 16) {
            int v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = w0 + j0 + j < n_wg ? meta[n_wg + (int64_t)(w0 + j0 + j) * PG_STASH_META + c] : 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += v[j];
        }
        if (sum != 0) atomicAdd(c <= 24 ? &chist[c - 1] : over_count, sum);
    }
    __syncthreads();
    // all the slots' entries as one index space (a slot-by-slot loop was a load -> store round trip per slot)
    const int first = s_off[0], total = s_off[PG_COMPACT_WGS] - first;
    for (int t = tid; t < total; t += 256) {
        int j = 0;
#pragma unroll
        for (int step = PG_COMPACT_WGS / 2; step >= 1; step >>= 1)
            if (s_off[j + step] - first <= t) j += step;
        inc[first + t] = stash[(int64_t)(w0 + j) * PG_STASH_ROWS + (t - (s_off[j] - first))];
    }
}

// K1P merge: SPLIT adjacent lanes per candidate (4 for lists of <= 10 keys, where the kernel waits for memory; fewer
// for 20 keys, where the sorted-insert ripples of a divergent wave are the cost), each folding every SPLIT-th hit list into its
// own register list (duplicates taken once), then lane 0 of the group takes the other lists over by shuffles and
// writes the row.  A row that is not exact -- fewer than k entries, or a list that overflowed -- is written as
// incomplete (count 0 / distance -1) and its index appended to `inc` (inc_count) for the plain scan.
template <int KCAP, int SPLIT>
__global__ void k_merge_hits(gm_pg_blocks P, const uint32_t* __restrict__ lists, int64_t n_q, int k,
                             int32_t* __restrict__ ids, int32_t* __restrict__ dists, int32_t* __restrict__ counts,
                             int nearest, int32_t* __restrict__ inc, int* __restrict__ inc_count,
                             int* __restrict__ over_count, int* __restrict__ chist, const uint8_t* __restrict__ cflag, const int* __restrict__ n_dev) {
    if (n_dev != nullptr && *n_dev < n_q) n_q = *n_dev;      // (speculative follow-ups: launched for an upper bound)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(t % SPLIT);
    const int64_t qi_raw = t / SPLIT;
    const bool valid = qi_raw < n_q;
    const int64_t qi = valid ? qi_raw : 0;      // (whole groups stay in the shuffles)
    RegList<KCAP> list;
    list.init(k, nullptr, 0);
    int overflow = 0;
    const uint32_t* __restrict__ row = lists + qi * P.row_words;      // [keys 0..3 of every list][keys 4.. of list 0][of list 1] ...
    const uint32_t* __restrict__ hrow = P.hdr + qi * P.hdr_words;      // keys appended to every list of the row
    for (int l = sub; l < P.n_lists; l += SPLIT) {
        const int lcap = P.lcap[l];                    // multiple of 4
        int c = (int)hrow[l];
        if (c > lcap) {
            overflow = 1;
            c = lcap;
        }
        // (the first four keys of all lists share the row's first lines: a row of short lists is a few lines, not one per list)
        const uint4* __restrict__ src_a = reinterpret_cast<const uint4*>(row + 4 * l);
        const uint4* __restrict__ src_b = reinterpret_cast<const uint4*>(row + P.loff[l]);
        for (int base = 0; base < c; base += 4) {      // an empty list costs no key read
            const uint4 w = base == 0 ? src_a[0] : src_b[(base >> 2) - 1];
            const uint32_t kk[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (base + j < c && kk[j] + 1u < list.v[KCAP - 1]) list.insert(kk[j]);
        }
    }
    // shuffle tree over the SPLIT lanes of a candidate: lane s takes the list of lane s + off over
#pragma unroll
    for (int off = SPLIT / 2; off >= 1; off >>= 1) {
        overflow |= __shfl_down(overflow, off, SPLIT);
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            const uint32_t v = __shfl_down(list.v[i], off, SPLIT);
            // (a partner's entries ascend: nothing that cannot beat the current k-th needs the 3 KCAP-op insert)
            if (sub < off && v != 0u && v < list.v[KCAP - 1]) list.insert(v - 1u);
        }
    }
    // The incomplete rows are appended to `inc` with ONE device-wide atomic per workgroup (gm_block_append) and counted
    // through LDS: with a thread per row on the same address, a cascade's first level -- which leaves a third of the rows
    // short on purpose -- spent 1.8 ms of S3's 2.4 ms merge in 270 000 serialised atomics (profiles/r03/NOTES.md).
    __shared__ int s_hist[24], s_over;
    if (threadIdx.x < 24) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 24) s_over = 0;
    __syncthreads();
    const bool owner = sub == 0 && valid;
    const bool full = list.v[KCAP - 1] != RegList<KCAP>::EMPTY;
    // (a short row that took the id cut-off does not know all its neighbours at r either)
    if (owner && !overflow && !full && cflag != nullptr && cflag[qi] != 0) overflow = 1;
    int32_t entry = 0;
    bool take = false;
    if (owner && overflow) {      // hits were lost: nothing about this row is known
        list.init(k, nullptr, 0);
        entry = (int32_t)((uint32_t)qi | PG_INC_OVERFLOW);
        take = true;
        atomicAdd(&s_over, 1);
    } else if (owner && !full) {      // every target within r is in the list (there are fewer than k): kept for the re-scan
        entry = (int32_t)qi;
        take = true;
        int c = 0;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) c += (list.v[s] != 0u && list.v[s] != RegList<KCAP>::EMPTY) ? 1 : 0;
        atomicAdd(&s_hist[c < 23 ? c : 23], 1);      // (the host sizes the prefix re-scan from this)
    }
    gm_block_append(take, entry, inc, inc_count);      // (contains the barrier that orders the LDS counters)
    if (threadIdx.x < 24 && s_hist[threadIdx.x] != 0) atomicAdd(&chist[threadIdx.x], s_hist[threadIdx.x]);
    if (threadIdx.x == 24 && s_over != 0) atomicAdd(over_count, s_over);
    if (!owner) return;
    list.store(k, qi, ids, dists, counts, nearest != 0);
}

// K1P merge for the register-list scan (k_hamming_scan_pg4): one thread per row folds the row's n_lists raw lists
// keys[list][row][KCAP] -- each the image of a SORTED register list: keys + 1 ascending, unused leading slots 0, empty
// slots 0xffffffff -- by two-way merges in registers: ten steps, each takes the smaller head (both when they are equal: a
// pair that several blocks found sits in several lists with the same key) and shifts the list(s) it came from by
// predicated moves.  ~21 KCAP VALU ops per list and row; the first version -- SPLIT lanes per row, insert_unique per key,
// a shuffle tree -- spent 2 400 lane-ops per row and 237 us on PAO1's first level, memory would allow ~70.
// A row whose merged list is not full (or that took the id cut-off and is short) is listed in `inc`.  There is no
// overflow: the lists are exact top-k lists of what their workgroups saw.
template <int KCAP, int SPLIT>
__global__ void k_merge_pg3(const uint32_t* __restrict__ keys, int n_lists, int64_t n_q, int k,
                            int32_t* __restrict__ ids, int32_t* __restrict__ dists, int32_t* __restrict__ counts,
                            int nearest, int32_t* __restrict__ inc, int* __restrict__ inc_count,
                            int* __restrict__ over_count, int* __restrict__ chist, int32_t* __restrict__ stash, int* __restrict__ meta,
                            const uint8_t* __restrict__ cflag, const int* __restrict__ n_dev) {
    static_assert(SPLIT == 1, "one thread per row");
    const int64_t n_stride = n_q;      // row stride of the raw lists: the host's number
    if (n_dev != nullptr && *n_dev < n_q) n_q = *n_dev;      // (speculative follow-ups: launched for an upper bound)
    const int64_t qi_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = qi_raw < n_q;
    const int64_t qi = valid ? qi_raw : 0;
    constexpr uint32_t NONE = RegList<KCAP>::EMPTY;
    const int lead = KCAP - k;      // unused leading slots of a raw list
    uint32_t a[KCAP];               // the merged keys so far, ascending, NONE-padded
#pragma unroll
    for (int i = 0; i < KCAP; ++i) a[i] = NONE;
    // a list's rows of this workgroup are one contiguous run of blockDim.x * KCAP words: fetched with coalesced dword loads
    // into LDS (a thread reading its own 4 KCAP bytes directly spreads every load instruction of a wave over 20 lines: S1's
    // merge 62 -> 102 us when the lanes per row went from 4 to 1) and read back row by row
    __shared__ uint32_t s_in[256 * KCAP];
    const int64_t row0 = (int64_t)blockIdx.x * blockDim.x;
    const int64_t rows_here = n_stride - row0 < (int64_t)blockDim.x ? n_stride - row0 : (int64_t)blockDim.x;
    const int n_words = (int)rows_here * KCAP;
    // (the NEXT list's words are fetched into registers while this one is merged -- KCAP words per thread, 256 threads --
    // and moved to LDS behind the merge: the kernel is memory-bound and ran list by list, load -> wait -> merge)
    auto fetch = [&](int l, uint32_t (&dst)[KCAP]) {
        const uint32_t* __restrict__ in = keys + ((int64_t)l * n_stride + row0) * KCAP;
#pragma unroll
        for (int i = 0; i < KCAP; ++i) {
            const int w = (int)threadIdx.x + i * 256;
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 8)      // timing build only (results WRONG): no list loads
            dst[i] = (uint32_t)(w * 2654435761u) | 1u;
            (void)in;
#else
            dst[i] = w < n_words ? in[w] : NONE;
#endif
        }
    };
    // one list: its words (in registers) -> LDS -> this thread's row -> merged into a[]
    auto fold_in = [&](const uint32_t (&src)[KCAP], bool first) {
        if (!first) __syncthreads();      // the previous list has been read
#pragma unroll
        for (int i = 0; i < KCAP; ++i) s_in[(int)threadIdx.x + i * 256] = src[i];
        __syncthreads();
        uint32_t b[KCAP];
#pragma unroll
        for (int i = 0; i < KCAP; ++i) b[i] = i < k && valid ? s_in[(int)threadIdx.x * KCAP + lead + i] : NONE;
        if (first) {
#pragma unroll
            for (int i = 0; i < KCAP; ++i) a[i] = b[i];
            return;
        }
        uint32_t out[KCAP];
#pragma unroll
        for (int s = 0; s < KCAP; ++s) {
            const uint32_t m = a[0] < b[0] ? a[0] : b[0];
            // (lane masks in SGPR pairs + explicit v_cndmask: left to itself the compiler turns the predicated shifts into
            // "elements shifted so far" counters and a ten-way select per read -- 2 000 instead of 23 instructions a step)
            const uint64_t ta = __builtin_amdgcn_ballot_w64(a[0] == m), tb = __builtin_amdgcn_ballot_w64(b[0] == m);
            out[s] = m;
#pragma unroll
            for (int i = 0; i + 1 < KCAP; ++i) {
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(a[i + 1]), "s"(ta));
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(b[i]) : "v"(b[i]), "v"(b[i + 1]), "s"(tb));
            }
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[KCAP - 1]) : "v"(a[KCAP - 1]), "v"(NONE), "s"(ta));
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(b[KCAP - 1]) : "v"(b[KCAP - 1]), "v"(NONE), "s"(tb));
        }
#pragma unroll
        for (int i = 0; i < KCAP; ++i) a[i] = out[i];
    };
    if constexpr (KCAP <= 5) {
        // short lists (k <= 5, the reference's default): a list is 5 KB per workgroup and its fetch is all latency -- the
        // first four lists are requested together (S3's first level: 68 us for 93 MB)
        constexpr int AHEAD = 4;
        uint32_t pre[AHEAD][KCAP];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
            if (j < n_lists) fetch(j, pre[j]);
#pragma unroll
        for (int j = 0; j < AHEAD; ++j)
            if (j < n_lists) fold_in(pre[j], j == 0);
        for (int l = AHEAD; l < n_lists; ++l) {
            uint32_t nxt[KCAP];
            fetch(l, nxt);
            fold_in(nxt, false);
        }
    } else {
        uint32_t cur[KCAP], nxt[KCAP];
        fetch(0, cur);
        for (int l = 0; l < n_lists; ++l) {
            if (l + 1 < n_lists) fetch(l + 1, nxt);      // (in flight while list l is merged)
            fold_in(cur, l == 0);
#pragma unroll
            for (int i = 0; i < KCAP; ++i) cur[i] = nxt[i];
        }
    }
    // -> the register-list image the rest of the pipeline stores from: unused leading slots 0, then the keys
    RegList<KCAP> list;
#pragma unroll
    for (int i = 0; i < KCAP; ++i) list.v[i] = i < lead ? 0u : NONE;
#pragma unroll
    for (int i = 0; i < KCAP; ++i)
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (j == i + lead) list.v[j] = a[i];      // (lead is wave-uniform: predicated moves, no dynamic register index)
    __shared__ int s_hist[24], s_over;
    if (threadIdx.x < 24) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 24) s_over = 0;
    __syncthreads();
    const bool owner = valid;
    const bool full = list.v[KCAP - 1] != NONE;
    // (a short row that took the id cut-off does not know all its neighbours at r: nothing about it is known)
    const bool unknown = owner && !full && cflag != nullptr && cflag[qi] != 0;
    int32_t entry = 0;
    bool take = false;
    if (unknown) {
        list.init(k, nullptr, 0);
        entry = (int32_t)((uint32_t)qi | PG_INC_OVERFLOW);
        take = true;
        atomicAdd(&s_over, 1);
    } else if (owner && !full) {      // every target within r is in the list (there are fewer than k): kept for the re-scan
        entry = (int32_t)qi;
        take = true;
        int c = 0;
#pragma unroll
        for (int s = 0; s < KCAP; ++s) c += (list.v[s] != 0u && list.v[s] != NONE) ? 1 : 0;
        atomicAdd(&s_hist[c < 23 ? c : 23], 1);      // (the host sizes the prefix re-scan from this)
    }
    // (both contain the barrier that orders the LDS counters)
    if (meta != nullptr) {      // many rows are expected short: workgroup slots, packed by k_compact_inc -- no device-wide atomics
        pg_block_stash(take, entry, s_hist, &s_over, stash, meta);
    } else {                    // hardly any (S4: 1 in 12 000): one atomic in the few workgroups that have one
        gm_block_append(take, entry, inc, inc_count);
        if (threadIdx.x < 24 && s_hist[threadIdx.x] != 0) atomicAdd(&chist[threadIdx.x], s_hist[threadIdx.x]);
        if (threadIdx.x == 24 && s_over != 0) atomicAdd(over_count, s_over);
    }
    // the rows' results leave through LDS as well: ids / dists of the workgroup's rows are one contiguous run each
    if (nearest != 0) {
        if (owner) list.store(k, qi, ids, dists, counts, true);
        return;
    }
    const int64_t rows_valid = n_q - row0 < (int64_t)blockDim.x ? n_q - row0 : (int64_t)blockDim.x;      // (may be <= 0)
    const int n_out = rows_valid > 0 ? (int)rows_valid * k : 0;
    int cnt = 0;
    for (int pass = 0; pass < 2; ++pass) {      // ids, then distances
        __syncthreads();
#pragma unroll
        for (int sI = 0; sI < KCAP; ++sI) {
            const int j = sI - lead;
            if (j >= 0) {
                const bool ok = list.v[sI] != NONE;
                const uint32_t key = list.v[sI] - 1u;
                s_in[(int)threadIdx.x * k + j] = ok ? (pass == 0 ? (key & ID_MASK) : (key >> ID_BITS)) : 0xffffffffu;
                if (pass == 0) cnt += ok ? 1 : 0;
            }
        }
        __syncthreads();
        int32_t* __restrict__ dst = (pass == 0 ? ids : dists) + row0 * k;
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 16)      // timing build only (results WRONG): no result stores
        if (n_out < 0) dst[0] = 0;
#else
        for (int w = (int)threadIdx.x; w < n_out; w += (int)blockDim.x) dst[w] = (int32_t)s_in[w];
#endif
    }
    if (owner) counts[qi] = cnt;
}

// K1P prefix re-scan, last step: the n_lists K1L segment lists of the scanned pool prefix (bound d < r + 2) of compact
// row j are folded and joined with what the row has.  A row the merge left short WITHOUT an overflow holds ALL its c < k
// neighbours within r in its result row already, so the fold takes only the prefix's targets at distance r + 1
// (`min_raw`: everything closer is one of the c), the k - c smallest of them go behind the c entries of the ORIGINAL row
// inc[j] and the row is complete; a row that does not reach k, or whose K1P lists had overflowed (flag on its inc
// entry), is reset to count 0 / distance -1 and appended to `bad` (*n_bad) for the plain re-scan.  SPLIT lanes per row.
// (One kernel for what used to be k_partial_keys + k_merge_prefix + k_scatter_rows + k_collect_bad.)
template <int KCAP, int SPLIT>
__global__ __launch_bounds__(256) void k_merge_prefix(const uint32_t* __restrict__ keys, int n_lists, const int32_t* __restrict__ inc,
                                                      int64_t n, int k, uint32_t min_raw, int32_t* __restrict__ ids,
                                                      int32_t* __restrict__ dists, int32_t* __restrict__ counts, int nearest,
                                                      int32_t* __restrict__ bad, int* __restrict__ n_bad,
                                                      const int* __restrict__ n_dev) {
    constexpr int N = fold_width(KCAP);
    const int64_t n_stride = n;      // row stride of the raw lists: the host's number
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = (int)(t % SPLIT);
    const int64_t j_raw = t / SPLIT;
    const bool valid = j_raw < n;
    const int64_t j = valid ? j_raw : 0;
    uint32_t acc[N];
    fold_lists<KCAP, SPLIT, N>(acc, keys, n_lists, n_stride, j, sub, min_raw);
    bool take = false;
    int32_t row32 = 0;
    if (sub == 0 && valid) {
        const uint32_t e = (uint32_t)inc[j];
        row32 = (int32_t)(e & ~PG_INC_OVERFLOW);
        const int64_t row = (int64_t)row32;
        const bool over = (e & PG_INC_OVERFLOW) != 0u;
        if (nearest != 0) {      // (no hit within r at all: the nearest target of the prefix at r + 1, if there is one)
            take = over || acc[0] == 0xffffffffu;
            dists[row] = take ? -1 : (int32_t)((acc[0] - 1u) >> ID_BITS);
        } else {
            const int c = over ? 0 : counts[row];
            int found = 0;
#pragma unroll
            for (int i = 0; i < KCAP; ++i) found += acc[i] != 0xffffffffu ? 1 : 0;
            take = over || c + found < k;
            if (take) {
                for (int s = 0; s < k; ++s) {
                    ids[row * k + s] = -1;
                    dists[row * k + s] = -1;
                }
                counts[row] = 0;
            } else {
#pragma unroll
                for (int i = 0; i < KCAP; ++i) {
                    const int dst = c + i;
                    if (dst < k) {
                        const uint32_t key = acc[i] - 1u;
                        ids[row * k + dst] = (int32_t)(key & ID_MASK);
                        dists[row * k + dst] = (int32_t)(key >> ID_BITS);
                    }
                }
                counts[row] = k;
            }
        }
    }
    gm_block_append(take, row32, bad, n_bad);
}

}  // namespace gm
