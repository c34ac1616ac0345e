// gm_host.h -- host-side declarations shared by the translation units of libgmhip.so.
// The templated scan kernels are instantiated one guide length per translation unit
// (gm_inst_lk.hip, gm_inst_lv.hip compiled with -DGM_LEN=<L>) so the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "gmhip.h"

struct gm_index {
    int device = 0;
    int L = 0;
    int metric = 0;
    int64_t n_db = 0;
    int64_t n_db_pad = 0;
    uint2* d_db = nullptr;        // bit planes, padded to a multiple of gm::TILE
    uint32_t* d_vert = nullptr;   // bit-sliced groups of 32 entries (Hamming, 10 <= L <= 27)
    uint32_t* d_onehot = nullptr; // one-hot bit-sliced table, groups of 64 entries (K1L)
    int64_t n_lk_tiles = 0;
    uint4* d_peq = nullptr;       // Levenshtein: match masks {A,C,G,T} per target (K2L), padded to LV_TILE
    int64_t n_lv_tiles = 0;
    uint32_t* d_keys = nullptr;   // K1L scratch: per-segment k-lists [n_seg][n_q][KCAP]
    int64_t keys_cap = 0;
    uint32_t* d_keys2 = nullptr;  // K2L scratch (its Hamming seeder uses d_keys at the same time)
    int64_t keys2_cap = 0;
    int64_t n_groups = 0;
    // workspaces (grow-only), so repeated queries do not allocate
    uint2* d_q = nullptr;
    int64_t q_cap = 0;
    void* d_stage = nullptr;      // host-pointer API: packed queries in / results out
    int64_t stage_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evq0 = nullptr, evq1 = nullptr;   // around the whole query (all kernels of one gm_*_dev call)
    bool timed = false;
    std::vector<hipEvent_t> hist0, hist1;   // optional per-launch history (gm_timing_reserve)
    int64_t hist_count = 0;
    const char* last_kernel = "";
    int64_t last_pairs = 0;
    int variant = 0;
    int n_simd = 1024;            // CUs x 4
};

int gm_fail(int code, const std::string& msg);   // records the thread-local error message, returns code

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gm_fail(GM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// K1L launchers, one per guide length (gm_inst_lk.hip)
#define GM_FOR_EACH_LEN(X) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) \
    X(24) X(25) X(26) X(27)
// arguments of one K1L launch (see k_hamming_knn_lk): kcap = list capacity (1 = nearest distance, 0 = LDS
// lists); single pass: n_seg = 1, tiles_per_seg = all tiles, out_keys = seed_keys = nullptr
struct gm_lk_args {
    int kcap = 10;
    int64_t n_q = 0;
    int k = 0;
    int n_seg = 1, tiles_per_seg = 0, seg_base = 0;
    const uint32_t* seed_keys = nullptr;
    uint32_t* out_keys = nullptr;
    int32_t *ids = nullptr, *dists = nullptr, *counts = nullptr;
};
#define GM_DECL_LK(LEN) int gm_launch_lk_L##LEN(gm_index* idx, const gm_lk_args* a, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_LK)
#undef GM_DECL_LK

// K2L launchers, one per guide length (gm_inst_lv.hip)
#define GM_DECL_LV(LEN)                                                                                        \
    int gm_launch_lv_L##LEN(gm_index* idx, int algo, int kcap, int64_t n_q, int k, int n_seg, int tiles_per_seg, \
                            const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_LV)
#undef GM_DECL_LV

// K1v (scalar-load bit-sliced kernel), kept for L = 20 as an A/B baseline (gm_inst_bs20.hip)
int gm_launch_bs20(gm_index* idx, bool nearest, int64_t n_q, int k, int32_t* ids, int32_t* dists,
                   int32_t* counts, hipStream_t st);
int gm_build_vertical(gm_index* idx, hipStream_t st);
