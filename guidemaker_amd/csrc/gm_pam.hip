// gm_pam.hip -- PAM scan + target extraction on the GPU.
// Replaces the regex loops of PamTarget.find_targets (reference guidemaker/core.py:142-246): for every
// (overlapping) occurrence of the IUPAC PAM on either strand whose adjacent guidelen-mer lies inside
// the record and is pure ACGT, emit the 2-bit packed target (reverse-complemented for the minus
// strand), its start coordinate and the match coordinate.  Row order = forward matches ascending, then
// reverse matches ascending, exactly as the reference concatenates them (core.py:252-284).
//
// Three kernels, byte/flag work, HBM-streaming:
//   k_pam_flags : one thread per genome position: 2 flag bits (fwd / rev hit) + per-block hit counts
//   k_pam_offsets: exclusive scan of the per-block counts (one workgroup)
//   k_pam_emit  : stable compaction: rank inside the block by wave ballots, write the records
#include "gm_host.h"

#include <cstring>

namespace {

constexpr int PB = 256;          // threads per block = positions per block
constexpr int MAX_PAM = 16;

struct PamParams {
    int plen;
    int L;
    int five_prime;
    unsigned char allow_fwd[MAX_PAM];   // bit c set: base code c allowed at this PAM position
    unsigned char allow_rev[MAX_PAM];   // same for the reverse-complemented PAM
};

__device__ __forceinline__ int base_code(unsigned char ch) {
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

// does the guidelen-mer starting at ts lie inside [0, n) and consist of ACGT only?
__device__ __forceinline__ bool window_ok(const unsigned char* __restrict__ seq, int64_t n, int64_t ts, int L) {
    if (ts < 0 || ts + L > n) return false;
    bool ok = true;
    for (int j = 0; j < L; ++j) ok = ok && (base_code(seq[ts + j]) < 4);
    return ok;
}

__device__ __forceinline__ unsigned hit_bits(const unsigned char* __restrict__ seq, int64_t n, int64_t i,
                                             const PamParams& P) {
    if (i + P.plen > n) return 0u;
    bool f = true, r = true;
    for (int p = 0; p < P.plen; ++p) {
        const int c = base_code(seq[i + p]);
        f = f && c < 4 && ((P.allow_fwd[p] >> c) & 1);
        r = r && c < 4 && ((P.allow_rev[p] >> c) & 1);
    }
    unsigned bits = 0u;
    // target lies after the match for (5prime, forward) and (3prime, reverse), else before it
    if (f) {
        const int64_t ts = P.five_prime ? i + P.plen : i - P.L;
        if (window_ok(seq, n, ts, P.L)) bits |= 1u;
    }
    if (r) {
        const int64_t ts = P.five_prime ? i - P.L : i + P.plen;
        if (window_ok(seq, n, ts, P.L)) bits |= 2u;
    }
    return bits;
}

__global__ __launch_bounds__(PB) void k_pam_flags(const unsigned char* __restrict__ seq, int64_t n, PamParams P,
                                                  unsigned char* __restrict__ flags,
                                                  unsigned* __restrict__ cnt_fwd, unsigned* __restrict__ cnt_rev) {
    __shared__ unsigned s_f[PB / 64], s_r[PB / 64];
    const int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x;
    const unsigned bits = i < n ? hit_bits(seq, n, i, P) : 0u;
    if (i < n) flags[i] = (unsigned char)bits;
    const unsigned long long bf = __ballot(bits & 1u), br = __ballot(bits & 2u);
    if ((threadIdx.x & 63) == 0) {
        s_f[threadIdx.x >> 6] = (unsigned)__popcll(bf);
        s_r[threadIdx.x >> 6] = (unsigned)__popcll(br);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int w = 0; w < PB / 64; ++w) {
            a += s_f[w];
            b += s_r[w];
        }
        cnt_fwd[blockIdx.x] = a;
        cnt_rev[blockIdx.x] = b;
    }
}

// exclusive scan of both count arrays by ONE workgroup of 1024 threads; totals to tot[0], tot[1]
__global__ __launch_bounds__(1024) void k_pam_offsets(unsigned* __restrict__ cnt_fwd, unsigned* __restrict__ cnt_rev,
                                                      int64_t nblocks, long long* __restrict__ off_fwd,
                                                      long long* __restrict__ off_rev, long long* __restrict__ tot) {
    __shared__ long long s[1024];
    for (int which = 0; which < 2; ++which) {
        const unsigned* cnt = which ? cnt_rev : cnt_fwd;
        long long* off = which ? off_rev : off_fwd;
        const int64_t per = (nblocks + 1023) / 1024;
        const int64_t lo = (int64_t)threadIdx.x * per;
        const int64_t hi = lo + per < nblocks ? lo + per : nblocks;
        long long sum = 0;
        for (int64_t b = lo; b < hi; ++b) sum += cnt[b];
        s[threadIdx.x] = sum;
        __syncthreads();
        // Hillis-Steele inclusive scan over the 1024 partial sums
        for (int d = 1; d < 1024; d <<= 1) {
            long long v = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += v;
            __syncthreads();
        }
        long long run = threadIdx.x ? s[threadIdx.x - 1] : 0;
        for (int64_t b = lo; b < hi; ++b) {
            off[b] = run;
            run += cnt[b];
        }
        if (threadIdx.x == 1023) tot[which] = s[1023];
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned long long pack_window(const unsigned char* __restrict__ seq, int64_t ts, int L,
                                                          bool forward) {
    unsigned long long v = 0;
    for (int j = 0; j < L; ++j) {
        const int c = forward ? base_code(seq[ts + j]) : 3 - base_code(seq[ts + L - 1 - j]);
        v |= (unsigned long long)c << (2 * j);
    }
    return v;
}

__global__ __launch_bounds__(PB) void k_pam_emit(const unsigned char* __restrict__ seq, int64_t n, PamParams P,
                                                 const unsigned char* __restrict__ flags,
                                                 const long long* __restrict__ off_fwd,
                                                 const long long* __restrict__ off_rev, long long n_fwd,
                                                 unsigned long long* __restrict__ packed,
                                                 long long* __restrict__ tstart, long long* __restrict__ mstart) {
    __shared__ unsigned s_f[PB / 64], s_r[PB / 64];
    const int64_t i = (int64_t)blockIdx.x * PB + threadIdx.x;
    const unsigned bits = i < n ? flags[i] : 0u;
    const unsigned long long bf = __ballot(bits & 1u), br = __ballot(bits & 2u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_f[wave] = (unsigned)__popcll(bf);
        s_r[wave] = (unsigned)__popcll(br);
    }
    __syncthreads();
    unsigned pf = 0, pr = 0;
    for (int w = 0; w < wave; ++w) {
        pf += s_f[w];
        pr += s_r[w];
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (bits & 1u) {
        const long long o = off_fwd[blockIdx.x] + pf + __popcll(bf & below);
        const int64_t ts = P.five_prime ? i + P.plen : i - P.L;
        packed[o] = pack_window(seq, ts, P.L, true);
        tstart[o] = ts;
        mstart[o] = i;
    }
    if (bits & 2u) {
        const long long o = n_fwd + off_rev[blockIdx.x] + pr + __popcll(br & below);
        const int64_t ts = P.five_prime ? i - P.L : i + P.plen;
        packed[o] = pack_window(seq, ts, P.L, false);
        tstart[o] = ts;
        mstart[o] = i;
    }
}

unsigned char iupac_mask(char ch) {
    switch (ch) {   // bit 0 A, 1 C, 2 G, 3 T
        case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8;
        case 'M': return 1 | 2; case 'R': return 1 | 4; case 'W': return 1 | 8; case 'S': return 2 | 4;
        case 'Y': return 2 | 8; case 'K': return 4 | 8; case 'V': return 1 | 2 | 4; case 'H': return 1 | 2 | 8;
        case 'D': return 1 | 4 | 8; case 'B': return 2 | 4 | 8; case 'X': case 'N': return 15;
        default: return 0;
    }
}

unsigned char complement_mask(unsigned char m) {   // A<->T, C<->G
    return (unsigned char)(((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3));
}

}  // namespace

struct gm_scan {
    int device = 0;
    int64_t n = 0, nblocks = 0;
    PamParams P{};
    unsigned char* d_seq = nullptr;
    unsigned char* d_flags = nullptr;
    unsigned *d_cf = nullptr, *d_cr = nullptr;
    long long *d_of = nullptr, *d_or = nullptr, *d_tot = nullptr;
    long long n_fwd = 0, n_rev = 0;
};

extern "C" {

void gm_pam_free(gm_scan* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->d_seq) (void)hipFree(s->d_seq);
    if (s->d_flags) (void)hipFree(s->d_flags);
    if (s->d_cf) (void)hipFree(s->d_cf);
    if (s->d_cr) (void)hipFree(s->d_cr);
    if (s->d_of) (void)hipFree(s->d_of);
    if (s->d_or) (void)hipFree(s->d_or);
    if (s->d_tot) (void)hipFree(s->d_tot);
    delete s;
}

int gm_pam_count(const char* seq, int64_t n, const char* pam, int pam_len, int five_prime, int guidelen,
                 int device, gm_scan** out, int64_t* n_fwd, int64_t* n_rev) {
    GM_ENTER();
    if (!out || !n_fwd || !n_rev) return gm_fail(GM_ERR_ARG, "gm_pam_count: null output");
    *out = nullptr;
    *n_fwd = *n_rev = 0;
    if (n < 0 || (n > 0 && !seq) || !pam || pam_len < 1 || pam_len > MAX_PAM || guidelen < 1 ||
        guidelen > GM_MAX_GUIDELEN)
        return gm_fail(GM_ERR_ARG, "gm_pam_count: bad argument");
    PamParams P{};
    P.plen = pam_len;
    P.L = guidelen;
    P.five_prime = five_prime ? 1 : 0;
    for (int p = 0; p < pam_len; ++p) {
        const unsigned char m = iupac_mask(pam[p]);
        if (!m) return gm_fail(GM_ERR_ARG, "gm_pam_count: PAM is not an IUPAC DNA string");
        P.allow_fwd[p] = m;
        P.allow_rev[pam_len - 1 - p] = complement_mask(m);     // reverse complement of the PAM
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gm_fail(GM_ERR_DEVICE, "no HIP device available (libgmhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return gm_fail(GM_ERR_ARG, "gm_pam_count: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    gm_scan* s = new gm_scan();
    s->device = device;
    s->n = n;
    s->P = P;
    *out = s;
    if (n == 0) return GM_OK;
    s->nblocks = (n + PB - 1) / PB;
    bool ok = hipMalloc(&s->d_seq, n) == hipSuccess && hipMalloc(&s->d_flags, n) == hipSuccess &&
              hipMalloc(&s->d_cf, sizeof(unsigned) * s->nblocks) == hipSuccess &&
              hipMalloc(&s->d_cr, sizeof(unsigned) * s->nblocks) == hipSuccess &&
              hipMalloc(&s->d_of, sizeof(long long) * s->nblocks) == hipSuccess &&
              hipMalloc(&s->d_or, sizeof(long long) * s->nblocks) == hipSuccess &&
              hipMalloc(&s->d_tot, sizeof(long long) * 2) == hipSuccess;
    if (!ok) {
        gm_pam_free(s);
        *out = nullptr;
        return gm_fail(GM_ERR_ALLOC, "gm_pam_count: hipMalloc failed");
    }
    HIP_TRY(hipMemcpy(s->d_seq, seq, n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pam_flags, dim3((unsigned)s->nblocks), dim3(PB), 0, 0, s->d_seq, n, P, s->d_flags, s->d_cf,
                       s->d_cr);
    hipLaunchKernelGGL(k_pam_offsets, dim3(1), dim3(1024), 0, 0, s->d_cf, s->d_cr, s->nblocks, s->d_of, s->d_or,
                       s->d_tot);
    HIP_TRY(hipGetLastError());
    long long tot[2] = {0, 0};
    HIP_TRY(hipMemcpy(tot, s->d_tot, sizeof(tot), hipMemcpyDeviceToHost));
    s->n_fwd = tot[0];
    s->n_rev = tot[1];
    *n_fwd = tot[0];
    *n_rev = tot[1];
    return GM_OK;
}

int gm_pam_emit(gm_scan* s, uint64_t* packed, int64_t* tstart, int64_t* mstart) {
    GM_ENTER();
    if (!s) return gm_fail(GM_ERR_ARG, "gm_pam_emit: null scan handle");
    const long long total = s->n_fwd + s->n_rev;
    if (total == 0) return GM_OK;
    if (!packed || !tstart || !mstart) return gm_fail(GM_ERR_ARG, "gm_pam_emit: null buffer");
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long* d_p = nullptr;
    long long *d_t = nullptr, *d_m = nullptr;
    if (hipMalloc(&d_p, 8 * total) != hipSuccess || hipMalloc(&d_t, 8 * total) != hipSuccess ||
        hipMalloc(&d_m, 8 * total) != hipSuccess) {
        if (d_p) (void)hipFree(d_p);
        if (d_t) (void)hipFree(d_t);
        return gm_fail(GM_ERR_ALLOC, "gm_pam_emit: hipMalloc failed");
    }
    hipLaunchKernelGGL(k_pam_emit, dim3((unsigned)s->nblocks), dim3(PB), 0, 0, s->d_seq, s->n, s->P, s->d_flags,
                       s->d_of, s->d_or, s->n_fwd, d_p, d_t, d_m);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(packed, d_p, 8 * total, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tstart, d_t, 8 * total, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mstart, d_m, 8 * total, hipMemcpyDeviceToHost);
    (void)hipFree(d_p);
    (void)hipFree(d_t);
    (void)hipFree(d_m);
    if (e != hipSuccess) return gm_fail(GM_ERR_DEVICE, std::string("gm_pam_emit: ") + hipGetErrorString(e));
    return GM_OK;
}

}  // extern "C"
