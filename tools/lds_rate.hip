// lds_rate.hip -- LDS read throughput on gfx950 for the access patterns of the look-up kernel:
// ds_read_b32/b64/b128 with (a) consecutive per-lane addresses, (b) one address for all lanes,
// (c) 4 distinct addresses (lane & 3), (d) 4 distinct addresses chosen pseudo-randomly per lane.
// Build: hipcc -O3 --offload-arch=gfx950 tools/lds_rate.hip -o tools/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 2048;

template <int WIDTH, int PAT>
__global__ void k_lds(unsigned* out, unsigned seed) {
    __shared__ __align__(16) unsigned buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = i * seed;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned h = (lane * 2654435761u) >> 13;
    int sel = PAT == 0 ? lane : PAT == 1 ? 0 : PAT == 2 ? (lane & 3) : (int)(h & 3);
    unsigned addr = (unsigned)(size_t)buf + sel * WIDTH + (threadIdx.x / 64) * 2048;   // bytes; LDS addr = low 32 bits
    unsigned acc = 0;
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (WIDTH == 4) {
            unsigned a, b, c, d, e, f, g, hh;
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n"
                         "ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n"
                         "ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(hh) : "v"(addr));
            acc += a ^ b ^ c ^ d ^ e ^ f ^ g ^ hh;
        } else if constexpr (WIDTH == 8) {
            unsigned long long a, b, c, d, e, f, g, hh;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:256\n ds_read_b64 %2, %8 offset:512\n"
                         "ds_read_b64 %3, %8 offset:768\n ds_read_b64 %4, %8 offset:1024\n ds_read_b64 %5, %8 offset:1280\n"
                         "ds_read_b64 %6, %8 offset:1536\n ds_read_b64 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(hh) : "v"(addr));
            acc += (unsigned)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ hh);
        } else {
            uint4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:512\n ds_read_b128 %2, %4 offset:1024\n"
                         "ds_read_b128 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr));
            acc += a.x ^ b.y ^ c.z ^ d.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WIDTH, int PAT>
void run(const char* name, unsigned* d_out, int cus) {
    printf("%-46s", name);
    constexpr int per_iter = WIDTH == 16 ? 4 : 8;
    for (int wpc : {4, 8, 16}) {       // waves per CU (as 256-thread workgroups)
        dim3 grid(cus * wpc / 4), block(256);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_lds<WIDTH, PAT>), grid, block, 0, 0, d_out, 3u);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_lds<WIDTH, PAT>), grid, block, 0, 0, d_out, 5u + rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double instr_per_cu = (double)wpc * ITERS * per_iter;
        double cyc = best * 1e-3 * 2.4e9 / instr_per_cu;
        printf("  %2d w/CU: %5.2f cyc/instr (%4.0f B/clk)", wpc, cyc, 64.0 * WIDTH / cyc);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * cus * 16 * 64));
    run<4, 0>("ds_read_b32 consecutive", d_out, cus);
    run<4, 1>("ds_read_b32 one address", d_out, cus);
    run<4, 2>("ds_read_b32 4 addresses (lane&3)", d_out, cus);
    run<4, 3>("ds_read_b32 4 addresses (hashed)", d_out, cus);
    run<8, 0>("ds_read_b64 consecutive", d_out, cus);
    run<8, 1>("ds_read_b64 one address", d_out, cus);
    run<8, 2>("ds_read_b64 4 addresses (lane&3)", d_out, cus);
    run<8, 3>("ds_read_b64 4 addresses (hashed)", d_out, cus);
    run<16, 0>("ds_read_b128 consecutive", d_out, cus);
    run<16, 1>("ds_read_b128 one address", d_out, cus);
    run<16, 2>("ds_read_b128 4 addresses (lane&3)", d_out, cus);
    run<16, 3>("ds_read_b128 4 addresses (hashed)", d_out, cus);
    return 0;
}
