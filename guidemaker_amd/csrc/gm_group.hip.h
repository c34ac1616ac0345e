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
};

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
                                                       int* __restrict__ cstart, int* __restrict__ n_chunks) {
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


// The launches of one grouping: histogram -> offsets (+ chunk geometry, chunks of <= 64 candidates x nsub.n[b] per bucket, and
// the chunk ranges of the blocks; also clears the caller's counter words) -> scatter.  K.coff[b] (4^len + 1 ints), K.cperm[b]
// (n_q), hist / cursor / cbase (K.ebase[B] ints; hist zeroed on entry and left zeroed), btot (GM_PG_MAXB + 1 ints, the
// last one -- the offsets kernel's ticket -- zeroed on entry and left zeroed).
inline int gm_group_launch(const uint2* q, int64_t n_q, const PgKeyArg& K, const PgSubArg& nsub, int* hist, int* cursor, int* cbase,
                           int* btot, int* ctr_words, int ctr_n_words, int* cstart, int* n_chunks, int device, hipStream_t st, const int* n_dev = nullptr) {
    // tiles of 1024 candidates per workgroup: as many as leave ~128 workgroups
    const int tiles = n_q >= (int64_t)1024 * 128 * 4 ? 4 : (n_q >= (int64_t)1024 * 128 * 2 ? 2 : 1);
    const dim3 cgrid((unsigned)((n_q + 1024 * tiles - 1) / (1024 * tiles)));
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
                       n_chunks);
    for (int i = 0; i < n_runs; ++i) {
        const size_t lds = runs[i].lds ? sizeof(int) * (size_t)(K.ebase[runs[i].hi] - K.ebase[runs[i].lo]) : 0;
        GM_LAUNCH_T(k_cand_scatter, q, n_q, K, cursor, runs[i].lds, runs[i].lo, runs[i].hi, n_dev);
    }
#undef GM_LAUNCH_T
    HIP_TRY(hipGetLastError());
    return GM_OK;
}
