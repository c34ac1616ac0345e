// gm_filters.hip -- the two target filters between the PAM scan and the index, on packed words
// (SURVEY.md 8(f) row N2).
//   gm_site_flags      replaces TargetProcessor.check_restriction_enzymes (reference
//                      guidemaker/core.py:354-377): flag targets that contain any of the given IUPAC
//                      sites; the caller passes each enzyme and its reverse complement, as the
//                      reference does before it builds its regex alternation.
//   gm_seed_duplicated replaces TargetProcessor.find_unique_near_pam (core.py:388-416): flag targets
//                      whose seed (a substring of `count` bases from `start`) already occurred at a
//                      smaller row index -- pandas' Series.duplicated(keep='first').
// Byte/flag work, HBM-streaming: one thread per target for the sites; for the seeds a stable radix sort
// of (seed, row) pairs (rocPRIM), one compare with the left neighbour, one scatter.
#include "gm_host.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int MAX_SITES = 64;
constexpr int MAX_SITE_LEN = GM_MAX_GUIDELEN;

struct SiteParams {
    int n_sites;
    int L;
    int len[MAX_SITES];
    int off[MAX_SITES];                           // start of the site's masks in `mask`
    unsigned char mask[MAX_SITES * MAX_SITE_LEN]; // bit c set: base code c allowed at that site position
};

__global__ __launch_bounds__(256) void k_site_flags(const unsigned long long* __restrict__ packed, int64_t n,
                                                    const SiteParams* __restrict__ P,
                                                    unsigned char* __restrict__ flags) {
    __shared__ SiteParams S;
    for (unsigned i = threadIdx.x; i < sizeof(SiteParams) / 4; i += blockDim.x)
        reinterpret_cast<unsigned*>(&S)[i] = reinterpret_cast<const unsigned*>(P)[i];
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned long long v = packed[t];
    bool any = false;
    for (int s = 0; s < S.n_sites && !any; ++s) {
        const int m = S.len[s];
        const unsigned char* mk = S.mask + S.off[s];
        for (int o = 0; o + m <= S.L && !any; ++o) {
            bool ok = true;
            for (int j = 0; j < m; ++j) ok = ok && ((mk[j] >> ((v >> (2 * (o + j))) & 3ull)) & 1);
            any = ok;      // (a zero-length site matches at once, like the empty regex alternative)
        }
    }
    flags[t] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_seed_keys(const unsigned long long* __restrict__ packed, int64_t n, int start,
                                                   int count, unsigned long long* __restrict__ keys,
                                                   unsigned* __restrict__ rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned long long m = count >= 32 ? ~0ull : ((1ull << (2 * count)) - 1ull);
    keys[t] = (packed[t] >> (2 * start)) & m;
    rows[t] = (unsigned)t;
}

// sorted by (seed, row): a row is a duplicate iff its left neighbour carries the same seed
__global__ __launch_bounds__(256) void k_seed_flags(const unsigned long long* __restrict__ keys,
                                                    const unsigned* __restrict__ rows, int64_t n,
                                                    unsigned char* __restrict__ flags) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    flags[rows[t]] = (t > 0 && keys[t] == keys[t - 1]) ? 1 : 0;
}

unsigned char site_mask(char ch) {
    switch (ch) {   // bit 0 A, 1 C, 2 G, 3 T
        case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8;
        case 'M': return 1 | 2; case 'R': return 1 | 4; case 'W': return 1 | 8; case 'S': return 2 | 4;
        case 'Y': return 2 | 8; case 'K': return 4 | 8; case 'V': return 1 | 2 | 4; case 'H': return 1 | 2 | 8;
        case 'D': return 1 | 4 | 8; case 'B': return 2 | 4 | 8; case 'X': case 'N': return 15;
        default: return 0;
    }
}

struct DevBuf {      // frees on scope exit
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

int pick_device(int device, const char* who) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return gm_fail(GM_ERR_DEVICE, "no HIP device available (libgmhip has no CPU fallback)");
    if (device < 0 || device >= ndev) return gm_fail(GM_ERR_ARG, who);
    HIP_TRY(hipSetDevice(device));
    return GM_OK;
}

}  // namespace

extern "C" {

int gm_site_flags(const uint64_t* packed, int64_t n, int guidelen, const char* const* sites, int n_sites,
                  uint8_t* flags, int device) {
    GM_ENTER();
    if (n < 0 || (n > 0 && (!packed || !flags)) || guidelen < 1 || guidelen > GM_MAX_GUIDELEN || n_sites < 0 ||
        (n_sites > 0 && !sites))
        return gm_fail(GM_ERR_ARG, "gm_site_flags: bad argument");
    if (n_sites > MAX_SITES) return gm_fail(GM_ERR_ARG, "gm_site_flags: more than 64 sites");
    SiteParams P{};
    P.L = guidelen;
    int used = 0;
    for (int s = 0; s < n_sites; ++s) {
        if (!sites[s]) return gm_fail(GM_ERR_ARG, "gm_site_flags: null site");
        const size_t m = std::strlen(sites[s]);
        if (m > (size_t)guidelen) continue;      // longer than the target: can never occur in it
        for (size_t j = 0; j < m; ++j) {
            const unsigned char k = site_mask(sites[s][j]);
            if (!k) return gm_fail(GM_ERR_SEQ, "gm_site_flags: site is not an IUPAC DNA string");
            P.mask[used + j] = k;
        }
        P.len[P.n_sites] = (int)m;
        P.off[P.n_sites] = used;
        used += (int)m;
        ++P.n_sites;
    }
    if (n == 0) return GM_OK;
    if (P.n_sites == 0) {
        std::memset(flags, 0, (size_t)n);
        return GM_OK;
    }
    int rc = pick_device(device, "gm_site_flags: bad device ordinal");
    if (rc) return rc;
    DevBuf d_p, d_f, d_s;
    if (d_p.alloc(sizeof(uint64_t) * n) != hipSuccess || d_f.alloc((size_t)n) != hipSuccess ||
        d_s.alloc(sizeof(SiteParams)) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "gm_site_flags: hipMalloc failed");
    HIP_TRY(hipMemcpy(d_p.p, packed, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_s.p, &P, sizeof(SiteParams), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_site_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0,
                       (const unsigned long long*)d_p.p, n, (const SiteParams*)d_s.p, (unsigned char*)d_f.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(flags, d_f.p, (size_t)n, hipMemcpyDeviceToHost));
    return GM_OK;
}

int gm_seed_duplicated(const uint64_t* packed, int64_t n, int guidelen, int start, int count, uint8_t* flags,
                       int device) {
    GM_ENTER();
    if (n < 0 || (n > 0 && (!packed || !flags)) || guidelen < 1 || guidelen > GM_MAX_GUIDELEN || start < 0 ||
        count < 0 || start + count > guidelen)
        return gm_fail(GM_ERR_ARG, "gm_seed_duplicated: bad argument");
    if (n >= (int64_t)1 << 32) return gm_fail(GM_ERR_ARG, "gm_seed_duplicated: more than 2^32 - 1 rows");
    if (n == 0) return GM_OK;
    int rc = pick_device(device, "gm_seed_duplicated: bad device ordinal");
    if (rc) return rc;
    DevBuf d_p, d_k0, d_k1, d_r0, d_r1, d_f, d_tmp;
    if (d_p.alloc(sizeof(uint64_t) * n) != hipSuccess || d_k0.alloc(sizeof(uint64_t) * n) != hipSuccess ||
        d_k1.alloc(sizeof(uint64_t) * n) != hipSuccess || d_r0.alloc(sizeof(unsigned) * n) != hipSuccess ||
        d_r1.alloc(sizeof(unsigned) * n) != hipSuccess || d_f.alloc((size_t)n) != hipSuccess)
        return gm_fail(GM_ERR_ALLOC, "gm_seed_duplicated: hipMalloc failed");
    HIP_TRY(hipMemcpy(d_p.p, packed, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_seed_keys, grid, block, 0, 0, (const unsigned long long*)d_p.p, n, start, count,
                       (unsigned long long*)d_k0.p, (unsigned*)d_r0.p);
    HIP_TRY(hipGetLastError());
    // stable LSD radix sort on the seed bits only: equal seeds keep ascending row order
    const unsigned end_bit = count == 0 ? 1u : (unsigned)(2 * count);
    size_t tmp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (unsigned long long*)d_k0.p, (unsigned long long*)d_k1.p,
                                      (unsigned*)d_r0.p, (unsigned*)d_r1.p, (size_t)n, 0u, end_bit, (hipStream_t)0));
    if (d_tmp.alloc(tmp_bytes) != hipSuccess) return gm_fail(GM_ERR_ALLOC, "gm_seed_duplicated: hipMalloc failed");
    HIP_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, (unsigned long long*)d_k0.p, (unsigned long long*)d_k1.p,
                                      (unsigned*)d_r0.p, (unsigned*)d_r1.p, (size_t)n, 0u, end_bit, (hipStream_t)0));
    hipLaunchKernelGGL(k_seed_flags, grid, block, 0, 0, (const unsigned long long*)d_k1.p, (const unsigned*)d_r1.p, n,
                       (unsigned char*)d_f.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(flags, d_f.p, (size_t)n, hipMemcpyDeviceToHost));
    return GM_OK;
}

}  // extern "C"
