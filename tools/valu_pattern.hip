// valu_pattern.hip -- issue-rate probes for the instruction patterns of the bit-sliced kernel
// (explicit registers): VGPR bank placement of 3-source v_bitop3, SGPR operands, full-adder pairs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_pattern.hip -o tools/valu_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 4096;

#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","s40","s41","s42","s43","s44","s45","s46","s47"

template <int P>
__global__ void k_pat(unsigned* out, unsigned seed) {
    asm volatile("v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n"
                 "v_mov_b32 v104, %0\n v_mov_b32 v105, %0\n v_mov_b32 v106, %0\n v_mov_b32 v107, %0\n"
                 "v_mov_b32 v108, %0\n v_mov_b32 v109, %0\n v_mov_b32 v110, %0\n v_mov_b32 v111, %0\n"
                 "v_mov_b32 v112, %0\n v_mov_b32 v113, %0\n v_mov_b32 v114, %0\n v_mov_b32 v115, %0\n"
                 "v_mov_b32 v116, %0\n v_mov_b32 v117, %0\n v_mov_b32 v118, %0\n v_mov_b32 v119, %0\n"
                 "v_mov_b32 v120, %0\n v_mov_b32 v121, %0\n v_mov_b32 v122, %0\n v_mov_b32 v123, %0\n"
                 "v_mov_b32 v124, %0\n v_mov_b32 v125, %0\n v_mov_b32 v126, %0\n v_mov_b32 v127, %0\n"
                 "v_mov_b32 v128, %0\n v_mov_b32 v129, %0\n v_mov_b32 v130, %0\n v_mov_b32 v131, %0\n"
                 "s_mov_b32 s40, %1\n s_mov_b32 s41, %1\n s_mov_b32 s42, %1\n s_mov_b32 s43, %1\n"
                 "s_mov_b32 s44, %1\n s_mov_b32 s45, %1\n s_mov_b32 s46, %1\n s_mov_b32 s47, %1\n"
                 :: "v"(threadIdx.x * seed), "s"(seed) : CLOB);
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (P == 0) {   // bitop3, 3 VGPR sources in 3 different banks (idx mod 4), 16 independent dests
            asm volatile(
                "v_bitop3_b32 v100, v101, v102, v103 bitop3:0x96\n v_bitop3_b32 v104, v105, v106, v107 bitop3:0x96\n"
                "v_bitop3_b32 v108, v109, v110, v111 bitop3:0x96\n v_bitop3_b32 v112, v113, v114, v115 bitop3:0x96\n"
                "v_bitop3_b32 v116, v117, v118, v119 bitop3:0x96\n v_bitop3_b32 v120, v121, v122, v123 bitop3:0x96\n"
                "v_bitop3_b32 v124, v125, v126, v127 bitop3:0x96\n v_bitop3_b32 v128, v129, v130, v131 bitop3:0x96\n"
                "v_bitop3_b32 v100, v101, v102, v103 bitop3:0xe8\n v_bitop3_b32 v104, v105, v106, v107 bitop3:0xe8\n"
                "v_bitop3_b32 v108, v109, v110, v111 bitop3:0xe8\n v_bitop3_b32 v112, v113, v114, v115 bitop3:0xe8\n"
                "v_bitop3_b32 v116, v117, v118, v119 bitop3:0xe8\n v_bitop3_b32 v120, v121, v122, v123 bitop3:0xe8\n"
                "v_bitop3_b32 v124, v125, v126, v127 bitop3:0xe8\n v_bitop3_b32 v128, v129, v130, v131 bitop3:0xe8\n"
                ::: CLOB);
        }
        if constexpr (P == 1) {   // bitop3, all 3 sources in the SAME bank (101,105,109 = 1 mod 4)
            asm volatile(
                "v_bitop3_b32 v100, v101, v105, v109 bitop3:0x96\n v_bitop3_b32 v104, v113, v117, v121 bitop3:0x96\n"
                "v_bitop3_b32 v108, v102, v106, v110 bitop3:0x96\n v_bitop3_b32 v112, v114, v118, v122 bitop3:0x96\n"
                "v_bitop3_b32 v116, v103, v107, v111 bitop3:0x96\n v_bitop3_b32 v120, v115, v119, v123 bitop3:0x96\n"
                "v_bitop3_b32 v124, v125, v129, v101 bitop3:0x96\n v_bitop3_b32 v128, v126, v130, v102 bitop3:0x96\n"
                "v_bitop3_b32 v100, v101, v105, v109 bitop3:0xe8\n v_bitop3_b32 v104, v113, v117, v121 bitop3:0xe8\n"
                "v_bitop3_b32 v108, v102, v106, v110 bitop3:0xe8\n v_bitop3_b32 v112, v114, v118, v122 bitop3:0xe8\n"
                "v_bitop3_b32 v116, v103, v107, v111 bitop3:0xe8\n v_bitop3_b32 v120, v115, v119, v123 bitop3:0xe8\n"
                "v_bitop3_b32 v124, v125, v129, v101 bitop3:0xe8\n v_bitop3_b32 v128, v126, v130, v102 bitop3:0xe8\n"
                ::: CLOB);
        }
        if constexpr (P == 2) {   // bitop3 with one SGPR source
            asm volatile(
                "v_bitop3_b32 v100, v101, s40, v103 bitop3:0xf6\n v_bitop3_b32 v104, v105, s41, v107 bitop3:0xf6\n"
                "v_bitop3_b32 v108, v109, s42, v111 bitop3:0xf6\n v_bitop3_b32 v112, v113, s43, v115 bitop3:0xf6\n"
                "v_bitop3_b32 v116, v117, s44, v119 bitop3:0xf6\n v_bitop3_b32 v120, v121, s45, v123 bitop3:0xf6\n"
                "v_bitop3_b32 v124, v125, s46, v127 bitop3:0xf6\n v_bitop3_b32 v128, v129, s47, v131 bitop3:0xf6\n"
                "v_bitop3_b32 v100, v101, s40, v103 bitop3:0xf6\n v_bitop3_b32 v104, v105, s41, v107 bitop3:0xf6\n"
                "v_bitop3_b32 v108, v109, s42, v111 bitop3:0xf6\n v_bitop3_b32 v112, v113, s43, v115 bitop3:0xf6\n"
                "v_bitop3_b32 v116, v117, s44, v119 bitop3:0xf6\n v_bitop3_b32 v120, v121, s45, v123 bitop3:0xf6\n"
                "v_bitop3_b32 v124, v125, s46, v127 bitop3:0xf6\n v_bitop3_b32 v128, v129, s47, v131 bitop3:0xf6\n"
                ::: CLOB);
        }
        if constexpr (P == 3) {   // v_xor with SGPR src0 (VOP2)
            asm volatile(
                "v_xor_b32 v100, s40, v101\n v_xor_b32 v104, s41, v105\n v_xor_b32 v108, s42, v109\n v_xor_b32 v112, s43, v113\n"
                "v_xor_b32 v116, s44, v117\n v_xor_b32 v120, s45, v121\n v_xor_b32 v124, s46, v125\n v_xor_b32 v128, s47, v129\n"
                "v_xor_b32 v102, s40, v103\n v_xor_b32 v106, s41, v107\n v_xor_b32 v110, s42, v111\n v_xor_b32 v114, s43, v115\n"
                "v_xor_b32 v118, s44, v119\n v_xor_b32 v122, s45, v123\n v_xor_b32 v126, s46, v127\n v_xor_b32 v130, s47, v131\n"
                ::: CLOB);
        }
        if constexpr (P == 4) {   // 2-source VOP2 xor, VGPR only
            asm volatile(
                "v_xor_b32 v100, v102, v101\n v_xor_b32 v104, v106, v105\n v_xor_b32 v108, v110, v109\n v_xor_b32 v112, v114, v113\n"
                "v_xor_b32 v116, v118, v117\n v_xor_b32 v120, v122, v121\n v_xor_b32 v124, v126, v125\n v_xor_b32 v128, v130, v129\n"
                "v_xor_b32 v100, v103, v101\n v_xor_b32 v104, v107, v105\n v_xor_b32 v108, v111, v109\n v_xor_b32 v112, v115, v113\n"
                "v_xor_b32 v116, v119, v117\n v_xor_b32 v120, v123, v121\n v_xor_b32 v124, v127, v125\n v_xor_b32 v128, v131, v129\n"
                ::: CLOB);
        }
        if constexpr (P == 5) {   // bitop3 with only TWO distinct VGPR sources (src repeated)
            asm volatile(
                "v_bitop3_b32 v100, v101, v102, v102 bitop3:0x96\n v_bitop3_b32 v104, v105, v106, v106 bitop3:0x96\n"
                "v_bitop3_b32 v108, v109, v110, v110 bitop3:0x96\n v_bitop3_b32 v112, v113, v114, v114 bitop3:0x96\n"
                "v_bitop3_b32 v116, v117, v118, v118 bitop3:0x96\n v_bitop3_b32 v120, v121, v122, v122 bitop3:0x96\n"
                "v_bitop3_b32 v124, v125, v126, v126 bitop3:0x96\n v_bitop3_b32 v128, v129, v130, v130 bitop3:0x96\n"
                "v_bitop3_b32 v100, v101, v102, v102 bitop3:0xe8\n v_bitop3_b32 v104, v105, v106, v106 bitop3:0xe8\n"
                "v_bitop3_b32 v108, v109, v110, v110 bitop3:0xe8\n v_bitop3_b32 v112, v113, v114, v114 bitop3:0xe8\n"
                "v_bitop3_b32 v116, v117, v118, v118 bitop3:0xe8\n v_bitop3_b32 v120, v121, v122, v122 bitop3:0xe8\n"
                "v_bitop3_b32 v124, v125, v126, v126 bitop3:0xe8\n v_bitop3_b32 v128, v129, v130, v130 bitop3:0xe8\n"
                ::: CLOB);
        }
        if constexpr (P == 6) {   // v_and_b32 / v_or_b32 VOP2 mix
            asm volatile(
                "v_and_b32 v100, v102, v101\n v_or_b32 v104, v106, v105\n v_and_b32 v108, v110, v109\n v_or_b32 v112, v114, v113\n"
                "v_and_b32 v116, v118, v117\n v_or_b32 v120, v122, v121\n v_and_b32 v124, v126, v125\n v_or_b32 v128, v130, v129\n"
                "v_and_b32 v100, v103, v101\n v_or_b32 v104, v107, v105\n v_and_b32 v108, v111, v109\n v_or_b32 v112, v115, v113\n"
                "v_and_b32 v116, v119, v117\n v_or_b32 v120, v123, v121\n v_and_b32 v124, v127, v125\n v_or_b32 v128, v131, v129\n"
                ::: CLOB);
        }
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v100, v104\n v_xor_b32 %0, %0, v108\n v_xor_b32 %0, %0, v112\n v_xor_b32 %0, %0, v116\n"
                 "v_xor_b32 %0, %0, v120\n v_xor_b32 %0, %0, v124\n v_xor_b32 %0, %0, v128\n" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int P>
void run(const char* name, unsigned* d_out, int cus) {
    printf("%-44s", name);
    for (int wps : {1, 2, 4, 6, 8}) {
        dim3 grid(cus * 4 * wps), block(64);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_pat<P>, grid, block, 0, 0, d_out, 3u);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_pat<P>, grid, block, 0, 0, d_out, 5u + rep);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double instr_per_simd = (double)wps * ITERS * 16;
        double cyc = best * 1e-3 * 2.4e9 / instr_per_simd;   // cycles (at nominal 2.4 GHz) per wave-instruction per SIMD
        printf("  w%d: %4.2f", wps, cyc);
    }
    printf("   [cycles/instr/SIMD @2.4GHz]\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, sizeof(unsigned) * cus * 32 * 64));
    run<4>("v_xor_b32 v,v,v", d_out, cus);
    run<6>("v_and/v_or v,v,v", d_out, cus);
    run<3>("v_xor_b32 v,s,v", d_out, cus);
    run<0>("v_bitop3 v,v,v,v (3 banks)", d_out, cus);
    run<1>("v_bitop3 v,v,v,v (same bank)", d_out, cus);
    run<5>("v_bitop3 v,v,v,v (2 distinct srcs)", d_out, cus);
    run<2>("v_bitop3 v,v,s,v", d_out, cus);
    return 0;
}
