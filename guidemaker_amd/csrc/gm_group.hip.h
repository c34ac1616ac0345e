// gm_group.hip.h -- the candidate grouping shared by K1P (gm_pigeon.hip) and K2P (gm_levpart.hip): ONE counting sort of the
// candidates by the value of every block of a layout (histogram, per-block scan that also lays out the chunk table, scatter),
// all on the device, no host round trip.  Included inside the including file's anonymous namespace.
#pragma once

// ---- candidate grouping: one counting sort for ALL blocks (order inside a bucket is irrelevant) ----
struct PgKeyArg {
    int B;
    int bstart[GM_PG_MAXB], blen[GM_PG_MAXB];
    int ebase[GM_PG_MAXB + 1];
    int32_t* coff[GM_PG_MAXB];
    uint32_t* cperm[GM_PG_MAXB];
    // two-level grouping (large candidate sets, gm_group_launch): the second buffer per block, the low key bits the second
    // pass sorts by, and the blocks' first partitions in the flat partition numbering (part_base[B] = all partitions)
    uint32_t* ctmp[GM_PG_MAXB];
    int lo_bits[GM_PG_MAXB];
    int part_base[GM_PG_MAXB + 1];
};
constexpr int PG_PART_LO_MAX = 7;         // entry of the first pass = row << 7 | low key bits: rows < 2^25
constexpr int PG_PART_LO = 5;             // low key bits the second pass sorts by (<= PG_PART_LO_MAX)
constexpr int PG_PART_MAX = 4096;         // partitions of all blocks together (8 blocks x 512)

__device__ __forceinline__ int pg_key(const uint2 e, int start, int len) {
    int key = 0;
    for (int j = 0; j < len; ++j) {
        const int p = start + j;
        key |= (int)(((e.x >> p) & 1u) | (((e.y >> p) & 1u) << 1)) << (2 * j);
    }
    return key;
}

// hist[entry] += candidates with that (block, value); LDS-privatised per workgroup when the table fits.  A workgroup
// takes T tiles of 1024 candidates: what reaches the device-wide histogram is one atomic per (workgroup, non-empty
// bucket), and with a few thousand buckets a single tile hardly shares any (1024 candidates touch ~900 of the 4096
// buckets of a 6-base block) -- T = 4 cuts those atomics by a third and the launch from 19 to 13 us on S1 (the scatter: 53 to 34).
template <int T>
__global__ __launch_bounds__(1024) void k_cand_hist(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                    int* __restrict__ hist, int use_lds, int b_lo, int b_hi,
                                                    const int* __restrict__ n_dev) {
    extern __shared__ int s_hist[];      // the (block, bucket) entries of blocks [b_lo, b_hi)
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;      // (speculative follow-ups: the launch was sized for an upper bound)
    const int e_lo = K.ebase[b_lo], n_ent = K.ebase[b_hi] - e_lo;
    if (use_lds) {
        for (int e = threadIdx.x; e < n_ent; e += blockDim.x) s_hist[e] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * blockDim.x + threadIdx.x;
        if (i < n) {
            const uint2 e = planes[i];
            for (int b = b_lo; b < b_hi; ++b) {
                const int ent = K.ebase[b] + pg_key(e, K.bstart[b], K.blen[b]);
                if (use_lds) atomicAdd(&s_hist[ent - e_lo], 1); else atomicAdd(&hist[ent], 1);
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < n_ent; e += blockDim.x)
            if (s_hist[e] != 0) atomicAdd(&hist[e_lo + e], s_hist[e]);
    }
}

// One workgroup per block: exclusive scan of its buckets' candidate counts -> coff[b][0..nb] and the scatter cursors,
// and in the same pass the chunk table's geometry -- chunks per (block, bucket) entry = ceil(count / 64) x the
// workgroups an entry's target buckets are split over, their exclusive offsets INSIDE the block (cbase) and the block's
// total (btot) --; it leaves the histogram zeroed for the next query.  (Count and chunk sums travel through the scan as
// one 64-bit word.)  The workgroup that finishes last turns the totals into the blocks' chunk ranges.
struct PgSubArg {
    int n[GM_PG_MAXB];      // workgroups an entry's target buckets are split over, per block
};
__device__ __forceinline__ unsigned long long pg_offsets_word(unsigned c, unsigned ns) {
    return (unsigned long long)c | ((unsigned long long)((c + 63u) / 64u * ns) << 32);
}
__global__ __launch_bounds__(1024) void k_cand_offsets(PgKeyArg K, PgSubArg nsub, int* __restrict__ hist,
                                                       int* __restrict__ cursor, int* __restrict__ cbase,
                                                       int* __restrict__ btot, int* __restrict__ ctr_words, int ctr_n_words,
                                                       int* __restrict__ cstart, int* __restrict__ n_chunks,
                                                       int* __restrict__ part_words, int n_part_words) {
    __shared__ unsigned long long s[16];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int e0 = K.ebase[b], nb = K.ebase[b + 1] - e0;
    const unsigned ns = (unsigned)nsub.n[b];
    // One workgroup per block: a handful of waves living on memory latency.  Wave w owns the entries [w x span, (w + 1) x
    // span); iteration i of a wave covers 64 CONSECUTIVE entries (one coalesced load / store per array -- a thread
    // walking its own 16 consecutive entries made every store instruction 64 separate 4-byte writes: 36 us for three
    // 16 384-bucket blocks), all loads of a wave in flight at once (<= 16 iterations: blocks of <= 7 bases).
    const int iters = (nb + 1023) / 1024;
    const int lo = wave * iters * 64;
    constexpr int PER_REG = 16;
    const bool in_regs = iters <= PER_REG;
    int cnt[PER_REG];
    unsigned long long sum = 0;
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < PER_REG; ++i) {
            const int v = lo + i * 64 + lane;
            cnt[i] = i < iters && v < nb ? hist[e0 + v] : 0;
        }
#pragma unroll
        for (int i = 0; i < PER_REG; ++i) sum += pg_offsets_word((unsigned)cnt[i], ns);
    } else {
        for (int i = 0; i < iters; ++i) {
            const int v = lo + i * 64 + lane;
            sum += pg_offsets_word(v < nb ? (unsigned)hist[e0 + v] : 0u, ns);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);      // the wave's total, in every lane
    if (lane == 0) s[wave] = sum;
    __syncthreads();
    unsigned long long carry = 0, total = 0;      // entries before this wave's range / of the whole block
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned long long t = s[w];
        carry += w < wave ? t : 0ull;
        total += t;
    }
    auto emit = [&](int i, int c) {
        const int v = lo + i * 64 + lane;
        const unsigned long long x = pg_offsets_word((unsigned)c, ns);
        unsigned long long inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        const unsigned long long before = carry + inc - x;
        if (v < nb) {
            K.coff[b][v] = (int)(unsigned)before;
            cursor[e0 + v] = (int)(unsigned)before;
            cbase[e0 + v] = (int)(before >> 32);
            hist[e0 + v] = 0;
        }
        carry += __shfl(inc, 63, 64);
    };
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < PER_REG; ++i)
            if (i < iters) emit(i, cnt[i]);
    } else {
        for (int i = 0; i < iters; ++i) {
            const int v = lo + i * 64 + lane;
            emit(i, v < nb ? hist[e0 + v] : 0);
        }
    }
    // The workgroup that finishes LAST turns the blocks' totals into their chunk ranges (ctr->cstart) and clears the
    // caller's per-query counter words (a kernel of its own before: ~5 us per grouping for one wave of work); the ticket
    // btot[GM_PG_MAXB] is zero on entry and left zero.
    // (the totals cross workgroups as device-scope stores / loads, complete before the ticket is taken: a release fence
    // at device scope would write the XCD's L2 back)
    if (tid == 0) {
        K.coff[b][nb] = (int)(unsigned)total;
        __hip_atomic_store(&btot[b], (int)(total >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(&btot[GM_PG_MAXB], 1) == (int)gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (s_last == 0) return;
    for (int i = tid; i < ctr_n_words; i += 1024) ctr_words[i] = 0;
    for (int i = tid; i < n_part_words; i += 1024) part_words[i] = 0;      // (two-level grouping: partition histogram + cursors)
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int bb = 0; bb < K.B; ++bb) {
            cstart[bb] = run;
            run += __hip_atomic_load(&btot[bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cstart[K.B] = run;
        *n_chunks = run;
        btot[GM_PG_MAXB] = 0;
    }
}

// cperm[b][position] = candidate row; positions handed out per bucket: rank inside the workgroup from an
// LDS counter, one global atomic per (workgroup, non-empty bucket) for the workgroup's base; T tiles of 1024
// candidates per workgroup (see k_cand_hist); blocks [b_lo, b_hi) per launch (their entries fit the LDS table)
template <int T>
__global__ __launch_bounds__(1024) void k_cand_scatter(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                       int* __restrict__ cursor, int use_lds, int b_lo, int b_hi,
                                                       const int* __restrict__ n_dev) {
    extern __shared__ int s_cnt[];      // [entries of the blocks] counts, then reused as bases
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int e_lo = K.ebase[b_lo], n_ent = K.ebase[b_hi] - e_lo;
    int ent[T][GM_PG_MAXB], rank[T][GM_PG_MAXB];
    if (use_lds) {
        for (int e = threadIdx.x; e < n_ent; e += blockDim.x) s_cnt[e] = 0;
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * blockDim.x + threadIdx.x;
        if (i < n) {
            const uint2 e = planes[i];
#pragma unroll
            for (int b = 0; b < GM_PG_MAXB; ++b) {
                if (b >= b_lo && b < b_hi) {
                    ent[t][b] = K.ebase[b] + pg_key(e, K.bstart[b], K.blen[b]) - e_lo;
                    rank[t][b] = use_lds ? atomicAdd(&s_cnt[ent[t][b]], 1) : atomicAdd(&cursor[e_lo + ent[t][b]], 1);
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        // the workgroup's base in every bucket it touched: the returning atomics of a thread's (up to 16) entries are
        // independent -- issue them back to back instead of one memory round trip per loop iteration
        for (int p0 = 0; p0 < n_ent; p0 += 16 * 1024) {      // (passes of 16 entries per thread: up to 36 864 entries)
            int cs[16], bs[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = p0 + threadIdx.x + j * 1024;
                cs[j] = e < n_ent ? s_cnt[e] : 0;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) bs[j] = cs[j] != 0 ? atomicAdd(&cursor[e_lo + p0 + threadIdx.x + j * 1024], cs[j]) : 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = p0 + threadIdx.x + j * 1024;
                if (e < n_ent) s_cnt[e] = bs[j];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * blockDim.x + threadIdx.x;
        if (i < n) {
#pragma unroll
            for (int b = 0; b < GM_PG_MAXB; ++b) {
                if (b >= b_lo && b < b_hi) {
                    const int pos = use_lds ? s_cnt[ent[t][b]] + rank[t][b] : rank[t][b];
                    K.cperm[b][pos] = (uint32_t)i;
                }
            }
        }
    }
}


// ---- two-level grouping for large candidate sets ----
// The one-level sort above pays one device-wide returning atomic per (workgroup, non-empty bucket): with 16 384-bucket
// blocks a workgroup's 4 096 candidates hardly share a bucket, so that is nearly one atomic and one scattered 4-byte store
// per candidate and block (PAO1, 1.17 M candidates x 3 blocks: histogram 51 us + scatter 94 us).  Two levels instead:
//   k_part_hist     candidates per PARTITION (block, high key bits; <= 512 per block): LDS histogram, one atomic per
//                   (workgroup, partition);
//   k_part_scatter  entry = row << 7 | low key bits, appended to its partition's run of ctmp[b] (partition starts = scan
//                   of the partition histogram, done by every workgroup for itself; a workgroup's entries of one
//                   partition are consecutive: runs of ~32 .. 64 words);
//   k_part_sort     one workgroup per partition: LDS counting sort of its entries by the low bits (<= 128 values) ->
//                   hist[] (for k_cand_offsets), coff[b][], cperm[b][] -- reads and writes stay inside the partition's
//                   few tens of KiB.
// k_cand_offsets then runs as before (it re-derives coff from hist, lays out the chunks and clears hist / the partition
// counters).  Order inside a bucket is irrelevant, as above.
__device__ __forceinline__ int pg_part_scan_excl(int v, int* s_wave) {      // exclusive scan over a 1024-thread workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) before += w < wave ? s_wave[w] : 0;
    __syncthreads();
    return before + inc - v;
}

template <int T>
__global__ __launch_bounds__(1024) void k_part_hist(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                    int* __restrict__ phist, const int* __restrict__ n_dev) {
    __shared__ int s_hist[PG_PART_MAX];
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int n_parts = K.part_base[K.B];
    for (int e = threadIdx.x; e < n_parts; e += 1024) s_hist[e] = 0;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * 1024 + threadIdx.x;
        if (i < n) {
            const uint2 e = planes[i];
            for (int b = 0; b < K.B; ++b)
                atomicAdd(&s_hist[K.part_base[b] + (pg_key(e, K.bstart[b], K.blen[b]) >> K.lo_bits[b])], 1);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_parts; e += 1024)
        if (s_hist[e] != 0) atomicAdd(&phist[e], s_hist[e]);
}

template <int T>
__global__ __launch_bounds__(1024) void k_part_scatter(const uint2* __restrict__ planes, int64_t n, PgKeyArg K,
                                                       const int* __restrict__ phist, int* __restrict__ pcursor,
                                                       const int* __restrict__ n_dev) {
    __shared__ int s_cnt[PG_PART_MAX];       // this workgroup's entries per partition, then their first position in ctmp[b]
    __shared__ int s_start[PG_PART_MAX];     // start of the partition (all blocks in one numbering)
    __shared__ int s_wave[16];
    __shared__ int s_bsum[GM_PG_MAXB + 1];
    if (n_dev != nullptr && *n_dev < n) n = *n_dev;
    const int n_parts = K.part_base[K.B];
    // partition starts: exclusive scan of the partition histogram (4 partitions per thread); a block's partitions start
    // at s_bsum[b] in that numbering
    {
        int v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (int)threadIdx.x * 4 + j;
            v[j] = e < n_parts ? phist[e] : 0;
            sum += v[j];
        }
        int run = pg_part_scan_excl(sum, s_wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (int)threadIdx.x * 4 + j;
            if (e < n_parts) s_start[e] = run;
            run += v[j];
        }
        for (int e = threadIdx.x; e < n_parts; e += 1024) s_cnt[e] = 0;
        __syncthreads();
        if (threadIdx.x < (unsigned)K.B) s_bsum[threadIdx.x] = s_start[K.part_base[threadIdx.x]];
        __syncthreads();
    }
    int key[T][GM_PG_MAXB], rank[T][GM_PG_MAXB];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * 1024 + threadIdx.x;
        if (i < n) {
            const uint2 e = planes[i];
#pragma unroll
            for (int b = 0; b < GM_PG_MAXB; ++b) {
                if (b < K.B) {
                    key[t][b] = pg_key(e, K.bstart[b], K.blen[b]);
                    rank[t][b] = atomicAdd(&s_cnt[K.part_base[b] + (key[t][b] >> K.lo_bits[b])], 1);
                }
            }
        }
    }
    __syncthreads();
    // the workgroup's run inside every partition it touched (independent returning atomics, issued back to back)
    {
        int cs[4], bs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (int)threadIdx.x + j * 1024;
            cs[j] = e < n_parts ? s_cnt[e] : 0;
        }
#pragma unroll
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 128)      // timing build only (results WRONG): no device-wide atomics in the first pass
        for (int j = 0; j < 4; ++j) bs[j] = cs[j];
#else
        for (int j = 0; j < 4; ++j) bs[j] = cs[j] != 0 ? atomicAdd(&pcursor[(int)threadIdx.x + j * 1024], cs[j]) : 0;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (int)threadIdx.x + j * 1024;
            if (e < n_parts) {
                int b = 0;
                while (b + 1 < K.B && e >= K.part_base[b + 1]) ++b;
                s_cnt[e] = s_start[e] - s_bsum[b] + bs[j];
            }
        }
    }
    __syncthreads();
    // (staging a block's entries through LDS so that a run leaves as consecutive words was measured and is slower: the
    // cost is the partial-line writes of the short runs themselves -- 37 -> 48 us on PAO1 --, profiles/r04/NOTES.md)
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * T + t) * 1024 + threadIdx.x;
        if (i < n) {
#pragma unroll
            for (int b = 0; b < GM_PG_MAXB; ++b) {
                if (b < K.B) {
                    const int lo = K.lo_bits[b];
#if defined(GM_PG_TIMING) && (GM_PG_TIMING & 64)      // timing build only (results WRONG): no entry stores in the first pass
                    if (s_cnt[K.part_base[b] + (key[t][b] >> lo)] + rank[t][b] == -12345) K.ctmp[b][0] = 0u;
#else
                    K.ctmp[b][s_cnt[K.part_base[b] + (key[t][b] >> lo)] + rank[t][b]] =
                        ((uint32_t)i << PG_PART_LO_MAX) | (uint32_t)(key[t][b] & ((1 << lo) - 1));
#endif
                }
            }
        }
    }
}

// one workgroup per partition.  A partition's entries are walked four per thread with all four loads in flight (a real
// genome's partitions differ by an order of magnitude -- PAO1's GC-rich 3.5-mers --, and one load per loop iteration made
// the largest partition's workgroup the whole kernel: 122 us); the low-bit counters are kept per wave pair to spread the
// LDS atomics of a skewed partition.
constexpr int PG_SORT_THREADS = 1024;
constexpr int PG_SORT_COPIES = 8;      // counter copies (two waves each)
__global__ __launch_bounds__(PG_SORT_THREADS) void k_part_sort(PgKeyArg K, const int* __restrict__ phist, int* __restrict__ hist) {
    constexpr int NLO = 1 << PG_PART_LO_MAX;
    __shared__ int s_h[PG_SORT_COPIES][NLO];      // counts per copy, then the copy's write cursor
    __shared__ int s_tot[NLO];
    __shared__ int s_red[PG_SORT_THREADS / 64];
    const int p = blockIdx.x, tid = threadIdx.x, copy = (tid >> 7) & (PG_SORT_COPIES - 1);
    int b = 0;
    while (b + 1 < K.B && p >= K.part_base[b + 1]) ++b;
    const int p0 = K.part_base[b], lo = K.lo_bits[b], n_lo = 1 << lo;
    // entries of the block's partitions before this one
    int before = 0;
    for (int e = p0 + tid; e < p; e += PG_SORT_THREADS) before += phist[e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = before;
    for (int e = tid; e < PG_SORT_COPIES * NLO; e += PG_SORT_THREADS) (&s_h[0][0])[e] = 0;
    __syncthreads();
    int start = 0;
#pragma unroll
    for (int w = 0; w < PG_SORT_THREADS / 64; ++w) start += s_red[w];
    const int count = phist[p];
    const uint32_t* __restrict__ in = K.ctmp[b] + start;
    const uint32_t mask = (uint32_t)n_lo - 1u;
    for (int i0 = 0; i0 < count; i0 += 4 * PG_SORT_THREADS) {
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * PG_SORT_THREADS + tid;
            e[j] = i < count ? in[i] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * PG_SORT_THREADS + tid < count) atomicAdd(&s_h[copy][e[j] & mask], 1);
    }
    __syncthreads();
    // per low-bit value: total over the copies, and every copy's exclusive share (-> its cursor)
    if (tid < n_lo) {
        int run = 0;
#pragma unroll
        for (int c = 0; c < PG_SORT_COPIES; ++c) {
            const int v = s_h[c][tid];
            s_h[c][tid] = run;
            run += v;
        }
        s_tot[tid] = run;
    }
    __syncthreads();
    if (tid < 64) {      // exclusive scan of the <= 128 totals: two per lane
        const int c0 = 2 * tid < n_lo ? s_tot[2 * tid] : 0, c1 = 2 * tid + 1 < n_lo ? s_tot[2 * tid + 1] : 0;
        int inc = c0 + c1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d, 64);
            if (tid >= d) inc += t;
        }
        const int ex = inc - c0 - c1;
        if (2 * tid < n_lo) {
            const int bucket = ((p - p0) << lo) + 2 * tid;
            hist[K.ebase[b] + bucket] = c0;
            K.coff[b][bucket] = start + ex;
            s_tot[2 * tid] = start + ex;
        }
        if (2 * tid + 1 < n_lo) {
            const int bucket = ((p - p0) << lo) + 2 * tid + 1;
            hist[K.ebase[b] + bucket] = c1;
            K.coff[b][bucket] = start + ex + c0;
            s_tot[2 * tid + 1] = start + ex + c0;
        }
    }
    __syncthreads();
    for (int e = tid; e < PG_SORT_COPIES * n_lo; e += PG_SORT_THREADS) {
        const int c = e / n_lo, v = e - c * n_lo;
        s_h[c][v] += s_tot[v];
    }
    __syncthreads();
    uint32_t* __restrict__ out = K.cperm[b];
    for (int i0 = 0; i0 < count; i0 += 4 * PG_SORT_THREADS) {
        uint32_t e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + j * PG_SORT_THREADS + tid;
            e[j] = i < count ? in[i] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * PG_SORT_THREADS + tid < count) out[atomicAdd(&s_h[copy][e[j] & mask], 1)] = e[j] >> PG_PART_LO_MAX;
    }
}

// The launches of one grouping: histogram -> offsets (+ chunk geometry, chunks of <= 64 candidates x nsub.n[b] per bucket, and
// the chunk ranges of the blocks; also clears the caller's counter words) -> scatter.  K.coff[b] (4^len + 1 ints), K.cperm[b]
// (n_q), hist / cursor / cbase (K.ebase[B] ints; hist zeroed on entry and left zeroed), btot (GM_PG_MAXB + 1 ints, the
// last one -- the offsets kernel's ticket -- zeroed on entry and left zeroed).
inline int gm_group_launch(const uint2* q, int64_t n_q, const PgKeyArg& K, const PgSubArg& nsub, int* hist, int* cursor, int* cbase,
                           int* btot, int* ctr_words, int ctr_n_words, int* cstart, int* n_chunks, int device, hipStream_t st, const int* n_dev = nullptr,
                           int* part_words = nullptr) {
    // tiles of 1024 candidates per workgroup: as many as leave ~128 workgroups
    const int tiles = n_q >= (int64_t)1024 * 128 * 4 ? 4 : (n_q >= (int64_t)1024 * 128 * 2 ? 2 : 1);
    const dim3 cgrid((unsigned)((n_q + 1024 * tiles - 1) / (1024 * tiles)));
    if (part_words != nullptr) {      // two-level grouping (the caller set K.ctmp / lo_bits / part_base; part_words: 2 x partitions, zero)
        const int n_parts = K.part_base[K.B];
        int* phist = part_words;
        int* pcursor = part_words + n_parts;
        if (tiles == 4) {
            hipLaunchKernelGGL(k_part_hist<4>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, n_dev);
            hipLaunchKernelGGL(k_part_scatter<4>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, pcursor, n_dev);
        } else if (tiles == 2) {
            hipLaunchKernelGGL(k_part_hist<2>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, n_dev);
            hipLaunchKernelGGL(k_part_scatter<2>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, pcursor, n_dev);
        } else {
            hipLaunchKernelGGL(k_part_hist<1>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, n_dev);
            hipLaunchKernelGGL(k_part_scatter<1>, cgrid, dim3(1024), 0, st, q, n_q, K, phist, pcursor, n_dev);
        }
        hipLaunchKernelGGL(k_part_sort, dim3((unsigned)n_parts), dim3(PG_SORT_THREADS), 0, st, K, phist, hist);
        hipLaunchKernelGGL(k_cand_offsets, dim3(K.B), dim3(1024), 0, st, K, nsub, hist, cursor, cbase, btot, ctr_words, ctr_n_words, cstart,
                           n_chunks, part_words, 2 * n_parts);
        HIP_TRY(hipGetLastError());
        return GM_OK;
    }
    // (up to 36 864 counters = 144 KiB of the CU's 160 KiB: the 7 / 7 / 6-base blocks of a large pool in ONE launch each
    // for the histogram and the scatter instead of three; above the default 64 KiB the kernels need the attribute, per device)
    constexpr int LDS_ENTRIES = 36864;
    {
        static bool attr_set[64] = {};
        const int dev = device >= 0 && device < 64 ? device : 0;
        if (!attr_set[dev]) {
            const void* fns[] = {reinterpret_cast<const void*>(k_cand_hist<1>), reinterpret_cast<const void*>(k_cand_hist<2>),
                                 reinterpret_cast<const void*>(k_cand_hist<4>), reinterpret_cast<const void*>(k_cand_scatter<1>),
                                 reinterpret_cast<const void*>(k_cand_scatter<2>), reinterpret_cast<const void*>(k_cand_scatter<4>)};
            for (const void* f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ENTRIES * (int)sizeof(int)));
            attr_set[dev] = true;
        }
    }
    // The histogram / scatter kernels privatise the (block, bucket) counters in LDS: one launch covers a run of blocks whose
    // buckets fit; a block of 8 bases (65 536 buckets) goes through device-wide atomics.
    struct Run { int lo, hi, lds; };
    Run runs[GM_PG_MAXB];
    int n_runs = 0;
    for (int b = 0; b < K.B;) {
        const int nb = K.ebase[b + 1] - K.ebase[b];
        if (nb > LDS_ENTRIES) {
            runs[n_runs++] = Run{b, b + 1, 0};
            ++b;
            continue;
        }
        int hi = b + 1;
        while (hi < K.B && K.ebase[hi + 1] - K.ebase[b] <= LDS_ENTRIES) ++hi;
        runs[n_runs++] = Run{b, hi, 1};
        b = hi;
    }
#define GM_LAUNCH_T(KERN, ...)                                                                          \
    do {                                                                                                \
        if (tiles == 4) hipLaunchKernelGGL(KERN<4>, cgrid, dim3(1024), lds, st, __VA_ARGS__);            \
        else if (tiles == 2) hipLaunchKernelGGL(KERN<2>, cgrid, dim3(1024), lds, st, __VA_ARGS__);       \
        else hipLaunchKernelGGL(KERN<1>, cgrid, dim3(1024), lds, st, __VA_ARGS__);                       \
    } while (0)
    for (int i = 0; i < n_runs; ++i) {
        const size_t lds = runs[i].lds ? sizeof(int) * (size_t)(K.ebase[runs[i].hi] - K.ebase[runs[i].lo]) : 0;
        GM_LAUNCH_T(k_cand_hist, q, n_q, K, hist, runs[i].lds, runs[i].lo, runs[i].hi, n_dev);
    }
    hipLaunchKernelGGL(k_cand_offsets, dim3(K.B), dim3(1024), 0, st, K, nsub, hist, cursor, cbase, btot, ctr_words, ctr_n_words, cstart,
                       n_chunks, static_cast<int*>(nullptr), 0);
    for (int i = 0; i < n_runs; ++i) {
        const size_t lds = runs[i].lds ? sizeof(int) * (size_t)(K.ebase[runs[i].hi] - K.ebase[runs[i].lo]) : 0;
        GM_LAUNCH_T(k_cand_scatter, q, n_q, K, cursor, runs[i].lds, runs[i].lo, runs[i].hi, n_dev);
    }
#undef GM_LAUNCH_T
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
