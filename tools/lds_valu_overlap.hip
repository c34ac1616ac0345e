// lds_valu_overlap.hip -- do LDS reads of some waves overlap with VALU work of other waves on a CU?
// Workgroup of 8 waves per CU x 2 workgroups: role(wave) decides whether a wave runs a ds_read_b64 loop
// (per-lane addresses, 4 distinct) or a v_bitop3 loop.  Modes: 0 all waves LDS, 1 all waves VALU,
// 2 half/half split by wave parity (mixes roles on every SIMD), 3 half/half split by wave >= 4,
// 4 every wave alternates LDS batch / VALU batch (the scan kernels' structure).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_valu_overlap.hip -o tools/lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 1024;

__device__ __forceinline__ unsigned lds_batch(unsigned addr) {   // 20 ds_read_b64
    unsigned long long a, b, c, d, e, f, g, h, i, j;
    unsigned acc = 0;
    asm volatile("ds_read_b64 %0, %10\n ds_read_b64 %1, %10 offset:32\n ds_read_b64 %2, %10 offset:64\n"
                 "ds_read_b64 %3, %10 offset:96\n ds_read_b64 %4, %10 offset:128\n ds_read_b64 %5, %10 offset:160\n"
                 "ds_read_b64 %6, %10 offset:192\n ds_read_b64 %7, %10 offset:224\n ds_read_b64 %8, %10 offset:256\n"
                 "ds_read_b64 %9, %10 offset:288\n s_waitcnt lgkmcnt(0)\n"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h), "=v"(i), "=v"(j) : "v"(addr));
    acc ^= (unsigned)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ i ^ j);
    asm volatile("ds_read_b64 %0, %10 offset:320\n ds_read_b64 %1, %10 offset:352\n ds_read_b64 %2, %10 offset:384\n"
                 "ds_read_b64 %3, %10 offset:416\n ds_read_b64 %4, %10 offset:448\n ds_read_b64 %5, %10 offset:480\n"
                 "ds_read_b64 %6, %10 offset:512\n ds_read_b64 %7, %10 offset:544\n ds_read_b64 %8, %10 offset:576\n"
                 "ds_read_b64 %9, %10 offset:608\n s_waitcnt lgkmcnt(0)\n"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h), "=v"(i), "=v"(j) : "v"(addr));
    return acc ^ (unsigned)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ i ^ j);
}

__device__ __forceinline__ unsigned valu_batch(unsigned x, unsigned y, unsigned z) {   // 80 v_bitop3
    unsigned a0 = x, a1 = y, a2 = z, a3 = x ^ y, a4 = y ^ z, a5 = x + z, a6 = x + 1, a7 = y + 2;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(a0) : "v"(a1), "v"(a2));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xe8" : "+v"(a1) : "v"(a2), "v"(a3));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(a2) : "v"(a3), "v"(a4));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xe8" : "+v"(a3) : "v"(a4), "v"(a5));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(a4) : "v"(a5), "v"(a6));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xe8" : "+v"(a5) : "v"(a6), "v"(a7));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(a6) : "v"(a7), "v"(a0));
        asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xe8" : "+v"(a7) : "v"(a0), "v"(a1));
    }
    return a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// mixed fetch: NL ds_read_b64 + NG global_load_dwordx2 gathers (4 distinct 8-byte addresses inside one
// 32-byte span of an L1-resident table) -- does the vector-memory return path add to the LDS one?
template <int NL, int NG>
__device__ __forceinline__ unsigned mixed_batch(unsigned addr, const unsigned long long* gp) {
    unsigned long long v[NL + NG];
#pragma unroll
    for (int i = 0; i < NG; ++i)
        asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v[NL + i]) : "v"(gp), "n"(i * 32));
#pragma unroll
    for (int i = 0; i < NL; ++i)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 32));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long x = 0;
#pragma unroll
    for (int i = 0; i < NL + NG; ++i) {
        asm volatile("" : "+v"(v[i]));
        x ^= v[i];
    }
    return (unsigned)(x ^ (x >> 32));
}

// 20 reads issued without waiting; the caller computes meanwhile and then calls lds_wait()
struct Batch { unsigned long long v[20]; };
__device__ __forceinline__ void lds_issue(Batch& b, unsigned addr) {
    asm volatile("ds_read_b64 %0, %10\n ds_read_b64 %1, %10 offset:32\n ds_read_b64 %2, %10 offset:64\n"
                 "ds_read_b64 %3, %10 offset:96\n ds_read_b64 %4, %10 offset:128\n ds_read_b64 %5, %10 offset:160\n"
                 "ds_read_b64 %6, %10 offset:192\n ds_read_b64 %7, %10 offset:224\n ds_read_b64 %8, %10 offset:256\n"
                 "ds_read_b64 %9, %10 offset:288\n"
                 : "=v"(b.v[0]), "=v"(b.v[1]), "=v"(b.v[2]), "=v"(b.v[3]), "=v"(b.v[4]), "=v"(b.v[5]), "=v"(b.v[6]),
                   "=v"(b.v[7]), "=v"(b.v[8]), "=v"(b.v[9]) : "v"(addr));
    asm volatile("ds_read_b64 %0, %10 offset:320\n ds_read_b64 %1, %10 offset:352\n ds_read_b64 %2, %10 offset:384\n"
                 "ds_read_b64 %3, %10 offset:416\n ds_read_b64 %4, %10 offset:448\n ds_read_b64 %5, %10 offset:480\n"
                 "ds_read_b64 %6, %10 offset:512\n ds_read_b64 %7, %10 offset:544\n ds_read_b64 %8, %10 offset:576\n"
                 "ds_read_b64 %9, %10 offset:608\n"
                 : "=v"(b.v[10]), "=v"(b.v[11]), "=v"(b.v[12]), "=v"(b.v[13]), "=v"(b.v[14]), "=v"(b.v[15]),
                   "=v"(b.v[16]), "=v"(b.v[17]), "=v"(b.v[18]), "=v"(b.v[19]) : "v"(addr));
}
__device__ __forceinline__ unsigned batch_fold(Batch& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long x = 0;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        asm volatile("" : "+v"(b.v[i]));
        x ^= b.v[i];
    }
    return (unsigned)(x ^ (x >> 32));
}

template <int NL, int NG, bool VALU>
__global__ __launch_bounds__(512) void k_mixed(unsigned* out, const unsigned long long* tbl, unsigned seed) {
    __shared__ __align__(16) unsigned buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i * seed;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned sel = (lane * 2654435761u >> 13) & 3;
    unsigned addr = (unsigned)(size_t)buf + sel * 8 + wave * 1024;
    const unsigned long long* gp = tbl + sel + wave * 128 + (blockIdx.x & 7) * 1024;
    unsigned acc = seed + threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
        acc ^= mixed_batch<NL, NG>(addr, gp);
        if (VALU) acc = valu_batch(acc, acc >> 3, seed);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int NL, int NG, bool VALU>
float run_mixed(unsigned* d_out, const unsigned long long* tbl, int cus, int wg_per_cu = 2, int threads = 512) {
    dim3 grid(cus * wg_per_cu), block(threads);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mixed<NL, NG, VALU>), grid, block, 0, 0, d_out, tbl, 3u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mixed<NL, NG, VALU>), grid, block, 0, 0, d_out, tbl, 5u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_overlap(unsigned* out, unsigned seed) {
    __shared__ __align__(16) unsigned buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i * seed;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(size_t)buf + ((lane * 2654435761u >> 13) & 3) * 8 + wave * 1024;
    unsigned acc = seed + threadIdx.x;
    bool lds_role = MODE == 0 ? true : MODE == 1 ? false : MODE == 2 ? (wave & 1) : MODE == 3 ? (wave >= 4) : false;
    lds_role = __builtin_amdgcn_readfirstlane(lds_role);
    if (MODE == 5) {     // software pipeline: reads of batch i+1 in flight while batch i is "computed"
        Batch A, B;
        lds_issue(A, addr);
        for (int it = 0; it < ITERS; it += 2) {
            unsigned fa = batch_fold(A);
            lds_issue(B, addr);
            acc = valu_batch(acc ^ fa, acc >> 3, seed);
            unsigned fb = batch_fold(B);
            lds_issue(A, addr);
            acc = valu_batch(acc ^ fb, acc >> 3, seed);
        }
        acc ^= batch_fold(A);
    } else
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 4) {
            acc ^= lds_batch(addr);
            acc = valu_batch(acc, acc >> 3, seed);
        } else if (lds_role) {
            acc ^= lds_batch(addr);
        } else {
            acc = valu_batch(acc, acc >> 3, seed);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
float run(unsigned* d_out, int cus, int wg_per_cu = 2, int threads = 512) {
    dim3 grid(cus * wg_per_cu), block(threads);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_overlap<MODE>, grid, block, 0, 0, d_out, 3u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_overlap<MODE>, grid, block, 0, 0, d_out, 5u + rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * cus * 4 * 512));
    float t0 = run<0>(d_out, cus), t1 = run<1>(d_out, cus), t2 = run<2>(d_out, cus), t3 = run<3>(d_out, cus),
          t4 = run<4>(d_out, cus);
    printf("16 waves/CU, %d iterations of (20 ds_read_b64 | 80 v_bitop3) per wave\n", ITERS);
    printf("all 16 waves LDS            : %.3f ms\n", t0);
    printf("all 16 waves VALU           : %.3f ms\n", t1);
    printf("8 LDS + 8 VALU (by parity)  : %.3f ms   (half of each alone: %.3f / %.3f)\n", t2, t0 / 2, t1 / 2);
    printf("8 LDS + 8 VALU (wave >= 4)  : %.3f ms\n", t3);
    printf("every wave LDS then VALU    : %.3f ms   (sum %.3f, max %.3f)\n", t4, t0 + t1, t0 > t1 ? t0 : t1);
    float t5 = run<5>(d_out, cus);
    printf("every wave, reads of batch i+1 in flight during VALU of batch i (16 waves/CU): %.3f ms\n", t5);
    // same work per wave at lower occupancy: 8 waves/CU (2/SIMD) and 12 waves/CU
    printf(" 8 waves/CU: alternate %.3f ms, pipelined %.3f ms  (x2 = per-16-wave equivalent: %.3f / %.3f)\n",
           run<4>(d_out, cus, 1, 512), run<5>(d_out, cus, 1, 512), 2 * run<4>(d_out, cus, 1, 512), 2 * run<5>(d_out, cus, 1, 512));
    printf("12 waves/CU: alternate %.3f ms, pipelined %.3f ms  (x4/3: %.3f / %.3f)\n",
           run<4>(d_out, cus, 3, 256), run<5>(d_out, cus, 3, 256), run<4>(d_out, cus, 3, 256) * 4 / 3, run<5>(d_out, cus, 3, 256) * 4 / 3);
    printf("24 waves/CU: alternate %.3f ms  (x2/3: %.3f)\n", run<4>(d_out, cus, 3, 512), run<4>(d_out, cus, 3, 512) * 2 / 3);
    // mixed LDS + global-gather fetch (16 and 24 waves/CU)
    unsigned long long* tbl;
    CHECK(hipMalloc(&tbl, 8 * 1024 * sizeof(unsigned long long) + 4096));
    CHECK(hipMemset(tbl, 1, 8 * 1024 * sizeof(unsigned long long) + 4096));
    printf("mixed fetch, 16 waves/CU, fetch only : 20 LDS %.3f | 16 LDS + 4 global %.3f | 12+8 %.3f | 0+20 %.3f ms\n",
           run_mixed<20, 0, false>(d_out, tbl, cus), run_mixed<16, 4, false>(d_out, tbl, cus),
           run_mixed<12, 8, false>(d_out, tbl, cus), run_mixed<0, 20, false>(d_out, tbl, cus));
    printf("mixed fetch, 16 waves/CU, + 80 VALU  : 20 LDS %.3f | 18+2 %.3f | 16+4 %.3f | 12+8 %.3f ms\n",
           run_mixed<20, 0, true>(d_out, tbl, cus), run_mixed<18, 2, true>(d_out, tbl, cus),
           run_mixed<16, 4, true>(d_out, tbl, cus), run_mixed<12, 8, true>(d_out, tbl, cus));
    printf("mixed fetch, 24 waves/CU, + 80 VALU  : 20 LDS %.3f | 18+2 %.3f | 16+4 %.3f | 12+8 %.3f ms\n",
           run_mixed<20, 0, true>(d_out, tbl, cus, 3, 512), run_mixed<18, 2, true>(d_out, tbl, cus, 3, 512),
           run_mixed<16, 4, true>(d_out, tbl, cus, 3, 512), run_mixed<12, 8, true>(d_out, tbl, cus, 3, 512));
    return 0;
}
