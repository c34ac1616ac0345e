// valu_rate.hip -- measures wave64 issue throughput of the integer VALU ops the scan kernels use,
// on gfx950, as lane-ops/clk/CU at several occupancies.  Build: hipcc -O3 --offload-arch=gfx950
// tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

#define OP8(INSTR)                                        \
    asm volatile(INSTR : "+v"(a0) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a1) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a2) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a3) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a4) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a5) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a6) : "v"(b), "v"(c));      \
    asm volatile(INSTR : "+v"(a7) : "v"(b), "v"(c));

template <int OP>
__global__ void k_rate(unsigned* out, unsigned seed) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,
             a7 = a0 + 7;
    unsigned b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x5bd1e995u;
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (OP == 0) { OP8("v_xor_b32 %0, %1, %0") OP8("v_xor_b32 %0, %2, %0") }
        if constexpr (OP == 1) { OP8("v_bitop3_b32 %0, %1, %0, %2 bitop3:0xde") OP8("v_bitop3_b32 %0, %2, %0, %1 bitop3:0xde") }
        if constexpr (OP == 2) { OP8("v_bcnt_u32_b32 %0, %0, %1") OP8("v_bcnt_u32_b32 %0, %0, %2") }
        if constexpr (OP == 3) { OP8("v_min3_u32 %0, %0, %1, %2") OP8("v_min3_u32 %0, %0, %2, %1") }
        if constexpr (OP == 4) { OP8("v_min_u32 %0, %1, %0") OP8("v_min_u32 %0, %2, %0") }
        if constexpr (OP == 5) { OP8("v_add_u32 %0, %1, %0") OP8("v_add_u32 %0, %2, %0") }
        if constexpr (OP == 6) { OP8("v_and_or_b32 %0, %0, %1, %2") OP8("v_and_or_b32 %0, %0, %2, %1") }
        if constexpr (OP == 7) { OP8("v_or_b32 %0, %1, %0") OP8("v_or_b32 %0, %2, %0") }
        if constexpr (OP == 8) { OP8("v_fma_f32 %0, %0, %1, %2") OP8("v_fma_f32 %0, %0, %2, %1") }
        if constexpr (OP == 9) { OP8("v_lshl_or_b32 %0, %0, 1, %1") OP8("v_lshl_or_b32 %0, %0, 1, %2") }
        if constexpr (OP == 10) { OP8("v_add3_u32 %0, %0, %1, %2") OP8("v_add3_u32 %0, %0, %2, %1") }
        if constexpr (OP == 11) { OP8("v_xor_b32_e64 %0, %1, %0") OP8("v_xor_b32_e64 %0, %2, %0") }
        if constexpr (OP == 12) { OP8("v_cmp_lt_u32 vcc, %0, %1") OP8("v_cmp_lt_u32 vcc, %0, %2") }
        if constexpr (OP == 13) { OP8("v_pk_add_u16 %0, %0, %1") OP8("v_pk_add_u16 %0, %0, %2") }
        if constexpr (OP == 14) { OP8("v_mad_u32_u24 %0, %0, %1, %2") OP8("v_mad_u32_u24 %0, %0, %2, %1") }
        if constexpr (OP == 15) { OP8("v_sad_u8 %0, %0, %1, %2") OP8("v_sad_u8 %0, %0, %2, %1") }
        if constexpr (OP == 16) { OP8("v_lshlrev_b32 %0, 1, %0") OP8("v_lshlrev_b32 %0, 1, %0") }
        if constexpr (OP == 17) { OP8("v_lshl_add_u32 %0, %0, 1, %1") OP8("v_lshl_add_u32 %0, %0, 1, %2") }
        if constexpr (OP == 18) { OP8("v_add_u32 %0, %0, %0") OP8("v_add_u32 %0, %0, %0") }
        if constexpr (OP == 19) { OP8("v_lshrrev_b32 %0, 1, %0") OP8("v_lshrrev_b32 %0, 1, %0") }
        if constexpr (OP == 20) { OP8("v_and_b32 %0, %1, %0") OP8("v_and_b32 %0, %2, %0") }
        if constexpr (OP == 21) { OP8("v_or_b32 %0, 1, %0") OP8("v_or_b32 %0, 1, %0") }
        if constexpr (OP == 22) { OP8("v_bfe_u32 %0, %0, 1, 5") OP8("v_bfe_u32 %0, %0, 1, 5") }
        if constexpr (OP == 23) { OP8("v_sub_u32 %0, %1, %0") OP8("v_sub_u32 %0, %2, %0") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP>
void run(const char* name, unsigned* d_out, int cus) {
    printf("%-16s", name);
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        int waves_per_cu = 4 * waves_per_simd;
        dim3 grid(cus * waves_per_cu), block(64);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<OP>, grid, block, 0, 0, d_out, 1u);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rate<OP>, grid, block, 0, 0, d_out, 2u + rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double lane_ops = (double)grid.x * 64 * ITERS * UNROLL;
        double per_s = lane_ops / (best * 1e-3);
        printf("  w/simd=%d: %6.2f Tlane-op/s (%5.1f lanes/clk/CU @2.4GHz)", waves_per_simd, per_s / 1e12,
               per_s / cus / 2.4e9);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * cus * 32 * 64));
    run<0>("v_xor_b32", d_out, cus);
    run<11>("v_xor_b32_e64", d_out, cus);
    run<7>("v_or_b32", d_out, cus);
    run<1>("v_bitop3_b32", d_out, cus);
    run<2>("v_bcnt_u32_b32", d_out, cus);
    run<3>("v_min3_u32", d_out, cus);
    run<4>("v_min_u32", d_out, cus);
    run<5>("v_add_u32", d_out, cus);
    run<10>("v_add3_u32", d_out, cus);
    run<6>("v_and_or_b32", d_out, cus);
    run<9>("v_lshl_or_b32", d_out, cus);
    run<12>("v_cmp_lt_u32", d_out, cus);
    run<13>("v_pk_add_u16", d_out, cus);
    run<14>("v_mad_u32_u24", d_out, cus);
    run<15>("v_sad_u8", d_out, cus);
    run<8>("v_fma_f32", d_out, cus);
    run<16>("v_lshlrev_b32", d_out, cus);
    run<19>("v_lshrrev_b32", d_out, cus);
    run<17>("v_lshl_add_u32", d_out, cus);
    run<18>("v_add_u32 x,x,x", d_out, cus);
    run<20>("v_and_b32", d_out, cus);
    run<21>("v_or_b32 inline", d_out, cus);
    run<22>("v_bfe_u32", d_out, cus);
    run<23>("v_sub_u32", d_out, cus);
    return 0;
}
