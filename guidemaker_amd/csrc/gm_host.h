// gm_host.h -- host-side declarations shared by the translation units of libgmhip.so.
// The templated scan kernels are instantiated one guide length per translation unit
// (gm_inst_lk.hip, gm_inst_lv.hip compiled with -DGM_LEN=<L>) so the library builds in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "gmhip.h"
#include "gm_pg_types.h"

struct gm_pigeon;   // K1P state (gm_pigeon.hip)
struct gm_levpart;  // K2P state (gm_levpart.hip)

// K1P cascade (gmhip.hip run_hamming_pigeon): the rows a K1P pass leaves short -- their k-th neighbour lies beyond the
// pass's radius -- are searched again, as a compact candidate set, by a further K1P pass with a larger radius and a
// layout of its own (fewer, larger buckets: the set is a fraction of the candidates), and only what the last pass
// leaves goes to the plain scan.  Every level owns its pool copies, workspaces and plan.
constexpr int GM_PG_LEVELS = 3;
// The planner's sample (gm_pg_sample_hist): per sampled candidate the number of pool targets within d = 0 .. 8.
constexpr int GM_PG_SAMPLE_BINS = 9;
struct gm_pg_sample {
    int n = 0;                         // rows
    const uint32_t* cum = nullptr;     // [n][GM_PG_SAMPLE_BINS] targets within d (cumulative)
    int k = 0;                         // neighbours asked for: a row's k-th neighbour lies at the first d with cum >= k
    int min_kth = -1;                  // only the rows whose k-th neighbour lies BEYOND this distance (a deeper cascade level)
};
struct gm_pg_level_state {
    gm_pigeon* pg = nullptr;
    int64_t plan_nq = -1;
    int plan_k = -1, plan_found = 0, plan_cascade = 0;
    int plan_parent_radius = -2;      // the parent level's radius the plan was made for
    int pg_flooded_k1 = 0;
    double plan_cost = 1.0;
    gm_pg_layout plan_lay{};
    std::vector<double> plan_p_short;
};

struct gm_index {
    int device = 0;
    int L = 0;
    int metric = 0;
    int64_t n_db = 0;
    int64_t n_db_pad = 0;
    uint2* d_db = nullptr;        // bit planes, padded to a multiple of gm::TILE
    uint32_t* d_onehot = nullptr; // one-hot bit-sliced table, groups of 64 entries (K1L)
    int64_t n_lk_tiles = 0;
    uint4* d_peq = nullptr;       // Levenshtein: match masks {A,C,G,T} per target (K2L), padded to LV_TILE
    int64_t n_lv_tiles = 0;
    uint32_t* d_keys = nullptr;   // K1L scratch: per-segment k-lists [n_seg][n_q][KCAP]
    int64_t keys_cap = 0;
    uint32_t* d_keys2 = nullptr;  // K2L scratch (its Hamming seeder uses d_keys at the same time)
    int64_t keys2_cap = 0;
    uint32_t* d_lvthr = nullptr;  // Levenshtein: per candidate the seeded bound (+ the compact seeds of the second pass, + a counter)
    int64_t lvthr_cap = 0;
    int64_t n_groups = 0;
    // workspaces (grow-only), so repeated queries do not allocate
    uint2* d_q = nullptr;
    int64_t q_cap = 0;
    void* d_stage = nullptr;      // host-pointer API: packed queries in / results out
    int64_t stage_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t last_ev0 = nullptr, last_ev1 = nullptr;      // the pair the last timed launch was bracketed with
    hipEvent_t evq0 = nullptr, evq1 = nullptr;   // around the whole query (all kernels of one gm_*_dev call)
    bool timed = false;
    std::vector<hipEvent_t> hist0, hist1;   // optional per-launch history (gm_timing_reserve)
    int64_t hist_count = 0;
    const char* last_kernel = "";
    int64_t last_pairs = 0;
    int64_t last_evaluated = 0;   // K1P: pairs the launch really evaluated (last_pairs = pairs it resolved)
    int64_t last_incomplete = 0;  // K1P: candidates handed to the plain scan afterwards
    int variant = 0;
    int lk_seg = 0;               // experiments (gm_set_param "lk_seg"): pool segments of K1P's re-scan, 0 = choose
    int lv_two_pass = 1;          // Levenshtein: narrow-band first pass + second pass for the short lists (gm_set_param "lv_two_pass")
    int lv_seg = 0;               // experiments (gm_set_param "lv_seg"): pool segments of the Levenshtein scan, 0 = choose
    int n_simd = 1024;            // CUs x 4
    gm_levpart* lp = nullptr;     // K2P: the Levenshtein partition's pool copies + workspaces (built on first use)
    gm_levpart* lp_deeper = nullptr;      // ... and those of its second level (radius + 1, for the rows the first leaves short)
    int64_t lp_rows2 = 0, lp_pairs2 = 0, lp_left2 = 0;      // last query: rows / pairs / rows left of the second level
    int lp_r1 = 0, lp_r2 = 0;                               // last query: radii of the two levels (0 = not run)
    gm_pigeon* pg = nullptr;      // K1P: block-sorted pool copies + query workspaces (built on first use) of the
                                  // cascade level that is running (level 0 between queries)
    // the layout planner's last answer (the enumeration costs ~30 us of host time per query otherwise)
    int64_t plan_nq = -1;
    int plan_k = -1, plan_found = 0;
    int plan_cascade = 0;         // the plan sends the rows it leaves short to a deeper K1P level
    int plan_parent_radius = -2;  // (deeper levels) the parent's radius the plan was made for: another radius, another plan
    double plan_cost = 1.0;       // the planner's cost of the plan (pool scans per candidate, re-scans included)
    std::vector<double> plan_p_short;      // measured share of the candidates whose k-th neighbour lies beyond r (empty: no sample)
    std::vector<uint32_t> plan_cum;        // level 0's sample: targets within d = 0 .. 8 of every sampled candidate (gm_pg_sample)
    int plan_cum_k = 0;
    gm_pg_level_state pg_deeper[GM_PG_LEVELS - 1];      // the state of levels 1 .. (swapped in while such a level runs)
    int pg_level = 0;             // cascade level whose state is in pg / plan_* right now
    int pg_cascade = 1;           // gm_set_param "pg_cascade": 0 = never, 1 = when the planner says so, 2 = whenever rows are left
    int pg_layout_deeper = 0;     // gm_set_param "pg_layout2": forced layout of the deeper levels (tests; digits as "pg_layout")
    // per level of the last query: candidates, radius, pairs evaluated, rows left short (gm_pigeon_cascade_info)
    int64_t casc_nq[GM_PG_LEVELS]{}, casc_pairs[GM_PG_LEVELS]{}, casc_left[GM_PG_LEVELS]{};
    int64_t casc_pairs_last[GM_PG_LEVELS]{};      // pairs per level of the last query that waited for its counts
    int casc_radius[GM_PG_LEVELS]{};
    int casc_levels = 0;
    // k (+ 1) for which the planner's layout left more than a third of the candidates incomplete on this pool (repeat
    // families flood the hit lists): later queries with that k go straight to the plain scan
    int pg_flooded_k1 = 0;
    gm_pg_layout plan_lay{};
    // Speculative follow-ups (gmhip.hip run_hamming_pigeon): what the last query of this shape left per cascade level --
    // rows after the merge, after the prefix re-scan, after the plain re-scan under the guessed bound (-1: that step did
    // not run) and the prefix fraction it used.  A query of the same shape sizes its follow-up launches from these
    // instead of waiting for the device's counts, and checks them once at the end.
    struct {
        bool valid = false;
        int64_t n_q = -1;
        int k = -1;
        bool nearest = false;
        int64_t ninc[GM_PG_LEVELS]{}, nleft[GM_PG_LEVELS]{}, nbad[GM_PG_LEVELS]{};
        double frac[GM_PG_LEVELS]{};
        int cooldown = 0;      // queries that stay synchronous after a misprediction
    } spec;
    int pg_spec = 1;              // gm_set_param "pg_spec": 0 = always wait for the counts (A/B, tests)
    int spec_last = 0;            // the last K1P query: 0 = waited for its counts, 1 = speculative, 2 = speculative, then run again
    // speculative run: device counts the kernels launched for an upper bound clamp to (nullptr: the host's n is exact) --
    // dev_nq: the candidates of the K1P level being run (grouping, merge); dev_rows: the rows of the re-scan being run
    const int* dev_nq = nullptr;
    const int* dev_rows = nullptr;
};

int gm_fail(int code, const std::string& msg);   // records the thread-local error message, returns code

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gm_fail(GM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// hipGetLastError() is per host thread and STICKY across libraries: an error left behind by somebody else's HIP call
// (RCCL probing peer devices, torch) would otherwise be reported by our next launch check.  Every public entry point
// starts by clearing it.
#define GM_ENTER() (void)hipGetLastError()

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
    uint32_t seed_all = 0;        // != 0 (and seed_keys == nullptr): every candidate's bound is this key (k-th key + 1 format)
    uint32_t* out_keys = nullptr;
    int32_t *ids = nullptr, *dists = nullptr, *counts = nullptr;
    const uint2* q = nullptr;     // candidate planes (nullptr: idx->d_q)
    const int32_t* q_rows = nullptr;      // != nullptr: candidate i is row (q_rows[i] & 0x7fffffff) of q (K1P's list of incomplete rows)
    int n_tiles = 0;              // scan only the first n_tiles tiles of the pool (0: all) -- K1P's prefix re-scan
    const int* n_dev = nullptr;   // device count n_q is an upper bound of (speculative follow-ups); nullptr: exact
};
#define GM_DECL_LK(LEN) int gm_launch_lk_L##LEN(gm_index* idx, const gm_lk_args* a, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_LK)
#undef GM_DECL_LK

// K1P launchers, one per number of evaluated positions (gm_inst_lk.hip): the launch covers the chunk-table entries
// [*c_begin, *c_end); grid = upper bound of their number, a multiple of 8 x pg_run
#define GM_DECL_PG(LEN)                                                                                      \
    int gm_launch_pg_L##LEN(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks,             \
                            const int* c_begin, const int* c_end, int64_t grid, int64_t n_q, int k, int thr0,  \
                            uint32_t* lists, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_PG)
#undef GM_DECL_PG
// ... and of the LDS-list scan (k_hamming_scan_pg3): keys = raw lists [n_lists][n_q][kcap]
#define GM_DECL_PG3(LEN)                                                                                      \
    int gm_launch_pg3_L##LEN(gm_index* idx, const gm_pg_blocks* blocks, const gm_pg_chunk* chunks,             \
                             const int* c_begin, const int* c_end, int64_t grid, int64_t n_q, int k, int kcap,  \
                             int thr0, uint32_t* keys, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_PG3)
#undef GM_DECL_PG3

// K1P host side (gm_pigeon.hip).  gm_pg_plan: number of blocks B (0 = the plain scan is the better choice);
// gm_pg_prepare: block-sorted pool copies for B; gm_pg_candidates: group the candidates in idx->d_q;
// gm_pg_scan: the K1P launch; gm_pg_merge: hit lists -> top k + the list of rows that are not exact;
// gm_pg_incomplete: how many such rows (synchronises the stream); gm_pg_gather / gm_pg_scatter: move those rows
// to / from a compact query.
int gm_pg_plan(int64_t n_db, int64_t n_q, int L, int k, double p_match, gm_pg_layout* out, const double* p_short = nullptr,
               int deeper_levels = 0, double* cost_out = nullptr, int* cascade_out = nullptr, const gm_pg_sample* smp = nullptr);
int gm_pg_sample_hist(gm_index* idx, int64_t n, hipStream_t st, std::vector<uint32_t>* cum);
void gm_pg_inherit(gm_index* idx, const gm_pigeon* parent);      // a deeper level takes over the pool's composition + the knobs
const gm_pigeon* gm_pg_state(const gm_index* idx);
int gm_pg_sample_rows(gm_index* idx, int64_t n, int64_t stride, hipStream_t st);
bool gm_pg_sampling(const gm_index* idx);
void gm_pg_expect_short(gm_index* idx, double share);      // (of the candidates, at the planned layout's radius)
void gm_pg_expect_hits(gm_index* idx, double hits);        // (targets within that radius per candidate, from the sample; 0 = unknown)
bool gm_pg_layout_make(int L, int B, int X, int lenx, int n2, gm_pg_layout* out);
int gm_pg_layout_radius(const gm_pg_layout& lay);
bool gm_pg_layout_forced(const gm_index* idx, gm_pg_layout* out);
int gm_pg_pool_stats(gm_index* idx, hipStream_t st, double* p_match);
int gm_pg_prepare(gm_index* idx, const gm_pg_layout* lay, hipStream_t st);
int gm_pg_candidates(gm_index* idx, int64_t n_q, int k, hipStream_t st);
int gm_pg_scan(gm_index* idx, int64_t n_q, int k, int kcap, hipStream_t st);
int gm_pg_merge(gm_index* idx, int kcap, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts,
                bool nearest, hipStream_t st);
int gm_pg_incomplete(gm_index* idx, int64_t* n_inc, int64_t* pairs_evaluated, hipStream_t st);
// (rows_only != nullptr: no compact copy of the candidates -- *q2 = the candidate planes as they are, *rows_only = the
// device list of the rows, for a consumer that indexes through it (the K1L re-scans))
int gm_pg_gather(gm_index* idx, int64_t n_inc, int k, bool nearest, const uint2** q2, int32_t** ids2,
                 int32_t** dists2, int32_t** counts2, hipStream_t st, const int32_t** rows_only = nullptr);
int gm_pg_scatter(gm_index* idx, int64_t n_inc, int k, bool nearest, int32_t* ids, int32_t* dists, int32_t* counts,
                  hipStream_t st);
// prefix re-scan of the incomplete rows (gmhip.hip run_hamming_pigeon): fraction of the pool (in id order) that
// almost surely holds the r + 1 neighbours such a row still needs (1.0: not worth it); gm_pg_merge_prefix folds the
// prefix's segment lists into the ORIGINAL rows behind the neighbours they already hold (rows that stay short, and rows
// whose lists had overflowed, come out with count 0 and become the new pg->inc, counted in ctr->n_bad)
double gm_pg_prefix_fraction(const gm_index* idx, int k);
int gm_pg_merge_prefix(gm_index* idx, int kcap, const uint32_t* keys, int n_lists, int64_t n_inc, int k, bool nearest,
                       int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st);
int gm_pg_collect_bad(gm_index* idx, int64_t n_inc, int need, bool nearest, hipStream_t st);
int gm_pg_read_bad(gm_index* idx, int64_t* n_bad, hipStream_t st);
// rows i of [0, n) with counts[i] < need -> pg->inc (compact; their number arrives with gm_pg_read_bad); the list itself
int gm_pg_collect_short(gm_index* idx, int64_t n, int need, const int32_t* counts, hipStream_t st);
const int32_t* gm_pg_inc_list(const gm_index* idx);
int gm_pg_debug_flags(const gm_index* idx);      // non-zero in a -DGM_PG_TIMING build (its answers are wrong on purpose)
int gm_pg_blocks_of(const gm_index* idx);
int gm_pg_block_info(const gm_index* idx, int b, int* positions, int64_t* pairs);
int gm_pg_radius(const gm_index* idx);      // r of the layout in use: every pair within r is enumerated
int gm_pg_count_incomplete(gm_index* idx, int64_t n, int need, const int32_t* counts, const int32_t* d1,
                           int64_t* n_bad, hipStream_t st);      // rows that are not full; synchronises
int gm_pg_set_param(gm_index* idx, const char* name, int value);
void gm_pg_free(gm_index* idx);
// speculative follow-ups (see gm_pigeon.hip): publish the counters to a mailbox without waiting; wait for one later;
// device address of a row count (which = 0: rows the merge left short, 1: rows the last collect left)
struct gm_pg_pending {
    gm_pigeon* pg = nullptr;
    int slot = 0;
    unsigned seq = 0;
};
bool gm_pg_mailbox_usable(const gm_index* idx);
int gm_pg_publish_async(gm_index* idx, int slot, hipStream_t st, gm_pg_pending* out);
int gm_pg_pending_wait(const gm_pg_pending* p, bool after_merge, hipStream_t st, int64_t* n_inc, int64_t* n_bad, int64_t* pairs);
const int* gm_pg_count_ptr(const gm_index* idx, int which);
bool gm_pg_hit_path_is_registers(const gm_index* idx);      // the launch being prepared runs k_hamming_scan_pg4

// K2L launchers, one per guide length (gm_inst_lv.hip)
// options of a Levenshtein scan launch: candidate planes other than idx->d_q; narrow-band first pass (K2B): bounds
// above cap_thr are lowered to it when *cap_cnt (candidates whose seed is at most cap_thr) is >= 90 % of n_q
struct gm_lv_opt {
    const uint2* q = nullptr;
    const int* cap_cnt = nullptr;
    int cap_thr = 0;
};
#define GM_DECL_LV(LEN)                                                                                        \
    int gm_launch_lv_L##LEN(gm_index* idx, int algo, int kcap, int64_t n_q, int k, int n_seg, int tiles_per_seg, \
                            const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys, hipStream_t st, \
                            const gm_lv_opt* opt);
GM_FOR_EACH_LEN(GM_DECL_LV)
#undef GM_DECL_LV

// K2P launchers, one per guide length (gm_inst_lv.hip), and its host side (gm_levpart.hip)
#define GM_DECL_LVP(LEN)                                                                                              \
    int gm_launch_lvp_L##LEN(const gm_lp_copies* P, const gm_lp_chunk* chunks, int64_t n_chunks, const uint2* q, int64_t n_q, \
                             int k, int kcap, const int32_t* seed_d, const int32_t* seed_c, int seed_k, uint32_t* keys,      \
                             int radius, const int* n_chunks_dev, hipStream_t st);
GM_FOR_EACH_LEN(GM_DECL_LVP)
#undef GM_DECL_LVP
struct gm_levpart;      // K2P state (gm_levpart.hip)
void gm_lp_free(gm_index* idx);
int gm_lp_set_param(gm_index* idx, const char* name, int value, bool* known);
int gm_lp_layout(int L, int r, double p_match, int* bstart, int* blen, int* copy_block, int* copy_shift, int* first_copy,
                 double* fraction);
int gm_lp_tune_radius(gm_index* idx);
void gm_lp_plan_get(gm_index* idx, int64_t* nq, int* k, int* r, int* r2);
void gm_lp_plan_set(gm_index* idx, int64_t nq, int k, int r, int r2);
int gm_lp_tune_radius2(gm_index* idx);
int64_t gm_lp_last_pairs(gm_index* idx, int level);      // (valid after the stream has been synchronised)
int gm_lp_prepare(gm_index* idx, int r, hipStream_t st);
int gm_lp_candidates(gm_index* idx, const uint2* q, int64_t n_q, hipStream_t st);
int gm_lp_merge(gm_index* idx, int kcap, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts, hipStream_t st);
int gm_lp_scan(gm_index* idx, const uint2* q, int64_t n_q, int k, int kcap, const int32_t* seed_d, const int32_t* seed_c, int seed_k,
               hipStream_t st, const uint32_t** keys_out, int* n_lists);
