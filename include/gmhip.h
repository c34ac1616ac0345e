/*
 * gmhip.h -- C ABI of libgmhip.so: MI355X (gfx950) guide-RNA neighbour search + CFD scoring.
 *
 * This is the drop-in boundary for GuideMaker's one data-parallel hot path.  Each entry point
 * replaces one interface of the reference (citations into USDA-ARS-GBRU/GuideMaker v0.4.2):
 *
 *   gm_pack                 guidemaker/core.py:379-386  TargetProcessor._one_hot_encode (and its
 *                           inline copy core.py:594-601): sequence -> the encoding the index eats
 *   gm_index_create         guidemaker/core.py:451-456 / 461-466  nmslib.init(space='bit_hamming' |
 *                           'leven', DistType.INT, OBJECT_AS_STRING, 'hnsw') + addDataPointBatch +
 *                           createIndex
 *   gm_knn_query            guidemaker/core.py:501-503  index.setQueryTimeParams + knnQueryBatch(k)
 *   gm_nearest_dist         guidemaker/core.py:603-606  knnQueryBatch(binseq, k=2) then i[1][0]
 *   gm_cfd_batch            guidemaker/cfd_score_calculator.py:60-85  calc_cfd, batched
 *   gm_index_free           (index object lifetime, core.py:321 self.nmslib_index)
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer; nothing is retained except inside a handle
 *   - every function returns 0 on success, <0 on error; gm_last_error() returns the thread-local
 *     message of the last failure
 *   - sequences cross the boundary 2-bit packed: uint64 per target, base p in bits [2p, 2p+1],
 *     A=0 C=1 G=2 T=3, all higher bits zero, 1 <= guidelen <= 27 (reference CLI range 10..27,
 *     cli.py:43-44)
 *   - distances are in TRUE units (mismatching bases / edit operations).  The reference sees
 *     2 x mismatches for Hamming because of the one-hot encoding (core.py:510-514); the host side
 *     applies that factor where it mirrors the reference's comparison.
 *   - neighbour lists are exact and ordered by (distance ascending, db index ascending); rows with
 *     fewer than k db points are padded with -1 and counts[] holds the valid length
 *   - *_dev variants take DEVICE pointers (HBM-resident inputs/outputs) and a hipStream_t passed as
 *     void*; they enqueue on that stream and return without waiting for the RESULTS.  Large Hamming pools go through the
 *     pigeonhole-partitioned scan, whose follow-up launches (the rows a level leaves short) are sized from row counts the
 *     device produces.  A query of the same shape (n_q, k) as the handle's previous one takes those counts from that
 *     query (upper bounds; every kernel clamps to the device's own count), enqueues EVERYTHING without a host round trip
 *     between the kernels, and looks at the counts once, after the last launch, through words a small kernel writes to
 *     pinned host memory -- so the call returns when the counts have arrived (the scans are done by then; the tail is
 *     still running), not when it was enqueued; a count above its bound makes the query run again the waiting way.  The
 *     first query of a shape waits for each count as it is needed (once per level that leaves rows, once or twice more for
 *     what the re-scans leave) and builds the block-sorted pool copies.  gm_last_speculation() tells which it was.
 *   - one host thread per handle; handles are independent; one handle = one GPU (multi-GPU is one
 *     process per GPU)
 *   - there is NO CPU fallback: without a gfx950 device every compute entry point fails with
 *     GM_ERR_DEVICE
 */
#ifndef GMHIP_H
#define GMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_METRIC_HAMMING 0
#define GM_METRIC_LEVEN 1

#define GM_OK 0
#define GM_ERR_ARG (-1)      /* bad argument */
#define GM_ERR_DEVICE (-2)   /* no usable GPU / HIP runtime error */
#define GM_ERR_ALLOC (-3)    /* out of memory */
#define GM_ERR_SEQ (-4)      /* non-ACGT base in input */

#define GM_MAX_GUIDELEN 27
#define GM_MAX_K 64
#define GM_MAX_DB (1 << 27)  /* neighbour keys are (dist << 27 | id) */

typedef struct gm_index gm_index;

const char* gm_version(void);
const char* gm_last_error(void);
int gm_device_count(void);

/* ASCII -> packed.  seqs: n rows of guidelen chars, no terminators.  GM_ERR_SEQ on a non-ACGT base. */
int gm_pack(const char* seqs, int64_t n, int guidelen, uint64_t* out);
int gm_unpack(const uint64_t* packed, int64_t n, int guidelen, char* out);
/* Python's str(float) of v[0..n), separated by '\n', into out (host code, no GPU): the 'CFD Similar Guides' column of
 * core.cfd_score (reference core.py:1136) is a list of str(float) per row.  Returns the bytes written, -1 when cap is
 * too small (25 n always suffices). */
int64_t gm_format_doubles(const double* v, int64_t n, char* out, int64_t cap);
/* The 'Similar guides' / 'Similar guide distances' columns of the reference's output table (Annotation._format_guide_table,
 * core.py:913-931) for n result rows (host code, no GPU): out_seqs = ";".join(the counts[r] neighbour sequences of row r),
 * out_dists = ";".join(str(distance)), rows separated by '\n'; db_packed = the index's targets (ids index it).  The
 * buffers need n * k * (guidelen + 1) and n * k * 12 bytes. */
int gm_format_neighbor_rows(const uint64_t* db_packed, int64_t n_db, int guidelen, const int32_t* ids, const int32_t* dists,
                            const int32_t* counts, int64_t n, int k, char* out_seqs, int64_t cap_seqs, char* out_dists,
                            int64_t cap_dists, int64_t* len_seqs, int64_t* len_dists);

/* Build the (exact, brute-force) index on GPU `device`: copies the packed db to HBM as bit planes. */
int gm_index_create(const uint64_t* packed, int64_t n_db, int guidelen, int metric, int device,
                    gm_index** out);
int gm_index_create_dev(const void* d_packed, int64_t n_db, int guidelen, int metric, int device,
                        void* stream, gm_index** out);
void gm_index_free(gm_index* idx);
int64_t gm_index_size(const gm_index* idx);
int gm_index_guidelen(const gm_index* idx);
int gm_index_metric(const gm_index* idx);

/* k nearest db points of every query.  ids/dists: [n_q * k] int32, counts: [n_q] int32. */
int gm_knn_query(gm_index* idx, const uint64_t* q, int64_t n_q, int k,
                 int32_t* ids, int32_t* dists, int32_t* counts);
int gm_knn_query_dev(gm_index* idx, const void* d_q, int64_t n_q, int k,
                     void* d_ids, void* d_dists, void* d_counts, void* stream);

/* distance to the nearest db point of every query.  d1: [n_q] int32 (-1 if the index is empty). */
int gm_nearest_dist(gm_index* idx, const uint64_t* q, int64_t n_q, int32_t* d1);
int gm_nearest_dist_dev(gm_index* idx, const void* d_q, int64_t n_q, void* d_d1, void* stream);

/* CFD score of n_pairs (guide, off-target) pairs.  mm: [20*4*4] doubles,
 * mm[(pos-1)*16 + r*4 + d], r = guide base (ACGU), d = complement of the off-target base (ACGT),
 * 1.0 on the Watson-Crick entries.  out: [n_pairs] doubles. */
int gm_cfd_batch(const uint64_t* guides, const uint64_t* offs, int64_t n_pairs, int guidelen,
                 const double* mm, double* out, int device);
int gm_cfd_batch_dev(const void* d_guides, const void* d_offs, int64_t n_pairs, int guidelen,
                     const void* d_mm, void* d_out, void* stream);

/* Neighbour lists as ONE 32-bit key per neighbour, (dist << 27) | id, 0xffffffff = none: the form that
 * crosses xGMI in the multi-GPU all-gather (4 B per neighbour instead of ids + dists + counts).
 * keys: [n_q * k] uint32.  Device pointers; enqueued on `stream`. */
int gm_pack_keys_dev(const void* d_ids, const void* d_dists, int64_t n_q, int k, void* d_keys, void* stream);
int gm_unpack_keys_dev(const void* d_keys, int64_t n_q, int k, void* d_ids, void* d_dists, void* d_counts,
                       void* stream);

/* Multi-GPU, ONE process, the devices of one node (replaces nmslib's num_threads fan-out of
 * knnQueryBatch, guidemaker/core.py:501-503, by a fan-out over GPUs).  gm_multi_create replicates the packed pool on
 * every listed device (ordinals must be distinct) and creates one RCCL communicator over them (ncclCommInitAll;
 * librccl.so is loaded on first use).  (Test mode, GMHIP_TEST_VIRTUAL_DEVICES=1 in the environment: an ordinal may
 * occur several times -- n logical shards on one GPU, device-to-device copies in place of the collective -- so that a
 * one-GPU box executes the fan-out, the shard offsets and the padding of the n > 1 path.)  gm_multi_knn_query cuts the candidates into n contiguous, equally long shards,
 * device g scans shard g (gm_knn_query_dev on its own stream, driven by its own host thread), ONE ncclAllGather
 * exchanges the neighbour lists as 32-bit keys, and the table gathered on the first device is returned: the result
 * is identical to gm_knn_query on one device.  Host pointers; blocking.  (One process per GPU over an existing
 * communicator -- torch.distributed -- is the other supported form: INTEGRATION.md.) */
typedef struct gm_multi gm_multi;
int gm_multi_create(const uint64_t* packed, int64_t n_db, int guidelen, int metric, const int* devices, int n_devices,
                    gm_multi** out);
int gm_multi_knn_query(gm_multi* m, const uint64_t* q, int64_t n_q, int k, int32_t* ids, int32_t* dists, int32_t* counts);
/* The nearest-target distance of every candidate, sharded the same way ("for get_control_seqs the random
 * query batch is split the same way"; replaces the k = 2 knnQueryBatch of guidemaker/core.py:603, the path's largest
 * query set -- up to 10^4 n random guides): device g scans shard g (gm_nearest_dist_dev), ONE all-gather of the int32
 * distances.  d1: [n_q], identical to gm_nearest_dist on one device. */
int gm_multi_nearest_dist(gm_multi* m, const uint64_t* q, int64_t n_q, int32_t* d1);
int gm_multi_devices(const gm_multi* m);
int gm_multi_has_comm(const gm_multi* m);          /* 1: an RCCL communicator exists and the all-gather goes through it */
gm_index* gm_multi_index(gm_multi* m, int i);      /* the i-th device's handle (measurement hooks, gm_set_variant) */
void gm_multi_free(gm_multi* m);

/* PAM scan + target extraction (replaces the regex loops of PamTarget.find_targets,
 * guidemaker/core.py:142-246).  seq: n upper-case ASCII bases of ONE record; pam: IUPAC string;
 * five_prime: 1 = 5'-[pam][target]-3', 0 = 5'-[target][pam]-3'.  gm_pam_count uploads the record, finds
 * every (overlapping) PAM occurrence on both strands whose adjacent guidelen-mer lies inside the record
 * and is pure ACGT, and returns the number of forward / reverse hits; gm_pam_emit writes, forward hits
 * (ascending) first, then reverse hits (ascending): the packed target (reverse-complemented for the
 * minus strand), the target's start coordinate and the PAM match's start coordinate, all on the
 * forward strand.  Arrays must hold n_fwd + n_rev elements. */
typedef struct gm_scan gm_scan;
int gm_pam_count(const char* seq, int64_t n, const char* pam, int pam_len, int five_prime, int guidelen,
                 int device, gm_scan** out, int64_t* n_fwd, int64_t* n_rev);
int gm_pam_emit(gm_scan* scan, uint64_t* packed, int64_t* tstart, int64_t* mstart);
void gm_pam_free(gm_scan* scan);

/* Target filters on packed words.  Host pointers; blocking.
 * gm_site_flags replaces TargetProcessor.check_restriction_enzymes (guidemaker/core.py:354-377):
 * flags[i] = 1 iff target i contains an occurrence of any of the n_sites (<= 64) IUPAC strings in `sites`
 * (NUL-terminated, upper case; the caller lists each enzyme AND its reverse complement, as the reference
 * does before building its regex alternation).  A site longer than guidelen never matches; an empty site
 * matches every target (the empty regex alternative).
 * gm_seed_duplicated replaces TargetProcessor.find_unique_near_pam (core.py:388-416): the seed of a
 * target is its `count` bases from position `start`; flags[i] = 1 iff an earlier row (smaller i) has the
 * same seed -- pandas' duplicated(keep="first"). */
int gm_site_flags(const uint64_t* packed, int64_t n, int guidelen, const char* const* sites, int n_sites,
                  uint8_t* flags, int device);
int gm_seed_duplicated(const uint64_t* packed, int64_t n, int guidelen, int start, int count, uint8_t* flags,
                       int device);

/* Measurement hooks: HIP-event duration (ms) of the dominant scan kernel of the last query on this
 * handle (blocks until that kernel has finished), its name, and the pairs it evaluated. */
double gm_last_kernel_ms(gm_index* idx);
/* HIP-event duration (ms) of the whole last gm_knn_query_dev / gm_nearest_dist_dev call on this handle:
 * every kernel it enqueued (packing, all scan phases, merge), device time only. */
double gm_last_query_ms(gm_index* idx);
const char* gm_last_kernel_name(gm_index* idx);
int64_t gm_last_kernel_pairs(gm_index* idx);
/* The pigeonhole-partitioned Hamming scan (K1P) resolves n_q * n_db pairs but evaluates only those that
 * share a block: the pairs its launch really computed, and the number of candidates it handed to the plain
 * scan afterwards (both equal the plain figures / 0 for the other kernels). */
int64_t gm_last_kernel_evaluated(gm_index* idx);
int64_t gm_last_incomplete(gm_index* idx);
/* Number of blocks B of the K1P layout currently built for this handle (0: none built yet). */
/* How the last K1P query sized the follow-up launches of the rows its levels left short: 0 = it waited for the device's
 * counts, 1 = from the previous query's counts, checked once at the end (gm_set_param "pg_spec"), 2 = so, but a level left
 * more rows than provided for and the query was run again the waiting way. */
int gm_last_speculation(gm_index* idx);
int gm_pigeon_blocks(gm_index* idx);
/* Block b of that layout: the positions per pair its scan evaluates (guide length minus the block itself, which its
 * copy's table leaves out) and the pairs its part of the last K1P launch evaluated.  Returns 1, or 0 when there is no
 * such block. */
int gm_pigeon_block_info(gm_index* idx, int b, int* positions, int64_t* pairs);
/* The layout planner on its own (host arithmetic only, no device needed): the layout K1P would use for a pool of n_db
 * targets of this length, k neighbours and base-match probability p_match (sum of squared base frequencies;
 * 0.25 = uniform), assuming as many candidates as targets, encoded as
 *   B + 100 n1 + 1000 len1 + 10000 n2 + 100000 len2
 * B = blocks; n1 = blocks whose one-substitution neighbour buckets are scanned too; len1 = their length, given only
 * when it differs from the even split ceil(guidelen / B); n2 = blocks with the two-substitution neighbourhood, len2 =
 * their length.  The layout serves radius 3 n2 + 2 n1 + (B - n1 - n2) - 1.  0 when the plain scan is the better choice. */
int gm_pigeon_plan(int64_t n_db, int guidelen, int k, double p_match);
/* The planner with a measured distribution (host arithmetic only): p_short[r], r = 0 .. guidelen, = share of the n_q
 * candidates whose k-th neighbour lies beyond r (what the library samples before it plans a query of >= 32 768
 * candidates); levels = K1P levels allowed (1 = a single pass, up to 3: the rows a pass leaves short go to a further
 * pass with a larger radius and a layout planned for their number, "cascade").  Fills blen[8] / trad[8] (block lengths
 * and search radii of the FIRST level's layout), *cost (modelled pool scans per candidate, everything included) and
 * *cascade (1: the cost assumes a deeper level takes the short rows).  Returns the number of blocks, 0 = plain scan. */
int gm_pigeon_plan_dist(int64_t n_db, int64_t n_q, int guidelen, int k, double p_match, const double* p_short,
                        int levels, int* blen, int* trad, double* cost, int* cascade);
/* The same with the planner's own kind of sample (what the library measures before it plans >= 32 768 candidates):
 * cum[i * 9 + d] = pool targets within Hamming distance d = 0 .. 8 of sampled candidate i (i < n_rows).  From it the
 * planner takes the share of candidates whose k-th neighbour lies beyond a radius AND the hits a radius costs. */
int gm_pigeon_plan_sample(int64_t n_db, int64_t n_q, int guidelen, int k, double p_match, const uint32_t* cum, int n_rows,
                          int levels, int* blen, int* trad, double* cost, int* cascade);
/* The layout K2P (the displaced-block partition of the Levenshtein scan) uses for radius r on guides of `guidelen` bases
 * (host arithmetic only): r + 1 blocks; block j is looked up under the displacements |d| <= min(floor(r / 2), r - j)
 * (block 0: none) -- the bound for the LEFTMOST block no edit touches -- as far as the displaced block stays inside the
 * sequence; the block lengths (2 .. 8 bases) minimise the share of the pool visited.  Fills blen[8] (block lengths),
 * *n_copies (block x displacement pairs = sorted pool copies) and *fraction (modelled share of the pool a candidate's
 * workgroups visit, sum over copies of p_match ^ block length).  Returns the number of blocks, 0 when there is no
 * layout (r < 1, r > 7, guides shorter than 2 (r + 1) or longer than 8 (r + 1) bases). */
int gm_levpart_layout(int guidelen, int r, double p_match, int* blen, int* n_copies, double* fraction);
/* K2P levels of the last Levenshtein query on this handle: *r1 = radius of the partitioned first pass (0: the plain K2B
 * scan ran), *r2 = radius of the second partitioned level (0: none), *rows2 / *pairs2 / *left2 = the rows that level took,
 * the pairs it evaluated and the rows it left for the plain second pass.  Returns the number of partitioned levels. */
int gm_levpart_info(gm_index* idx, int* r1, int* r2, int64_t* rows2, int64_t* pairs2, int64_t* left2);
/* The K1P cascade of the last query on this handle: number of levels that ran (0: the plain scan), and for level l
 * (0 = the first pass) its candidates, the radius its layout serves, the pairs it evaluated and the rows it left
 * short (handed to the next level or to the re-scan).  Returns the number of levels, or 0. */
int gm_pigeon_cascade_info(gm_index* idx, int level, int64_t* n_q, int* radius, int64_t* pairs, int64_t* left);

/* Per-launch timing history: gm_timing_reserve(idx, n) arms n HIP-event pairs (and resets the
 * history); every later scan-kernel launch on this handle records into the next pair (wrapping);
 * gm_timing_read waits for the recorded launches and writes up to max_n durations (ms, oldest
 * first) into ms_out, returning how many were written. */
int gm_timing_reserve(gm_index* idx, int n);
int gm_timing_read(gm_index* idx, double* ms_out, int max_n);

/* A/B knob for experiments, benchmarks and tests; results are identical for every value.  0 = default.
 *   Hamming:     8 = plain two-phase look-up scan K1L (no pigeonhole partition); 7 = K1L with its k-lists in LDS (the
 *                path of k > 20); 1 = popcount kernel K1 (the path of guides shorter than 10 bases); 100 + B = K1P with
 *                B blocks, no expansion; 200 + B / 300 + B one expansion block (even length / one base longer);
 *                400 + B / 500 + B two / three longer expansion blocks; 600 + B / 700 + B two / three expansion blocks of
 *                even length; 800 + B block 0 with the two-substitution neighbourhood; 900 + B the same plus block 1 with
 *                the one-substitution one.  (Round 3's 9, 20..36, 40..49 and 50..99 -- a single-pass K1L, the scalar-load
 *                kernel K1v, capped occupancy, forced segment counts -- had no test or tool left and are gone.)
 *   Levenshtein: 2 = per-pair Myers kernel K2L; 1 = plane-based kernel (also used for L < 10 and k > 20). */
int gm_set_variant(gm_index* idx, int variant);
/* Named tuning knobs (experiments; results are identical for every value).
 *   "pg_sub" (0 = choose: 2 on the first cascade level, 8 on the deeper ones; 1..19): workgroups the target buckets of one
 *   expansion chunk of the pigeonhole-partitioned scan are split over on the hit-buffer path (more = shorter work items and
 *   more hit lists per candidate);
 *   "pg_run" (0 = choose: 128, 32 on the register-list path): consecutive chunk-table entries of that scan executed on the same XCD;
 *   "pg_slots" (0 = choose: slots where about one row per workgroup or more is expected short; 1 = one device-wide atomic per
 *   workgroup; 2 = slots): how the register-list merge lists the rows it leaves incomplete (a slot per workgroup, packed by a
 *   second kernel, instead of atomics on one cursor);
 *   "pg_group" (0 = choose: two levels from 262 144 candidates; 1 = one counting sort; 2 = two levels): how that scan's
 *   candidates are grouped by block value (gm_group.hip.h: partitions of the high key bits, then an LDS sort per partition);
 *   "pg_layout": force a layout of that scan, decimal digit pairs (bases, radius) per block, e.g. 826161 = blocks of
 *   8, 6, 6 bases with the two-, one-, one-substitution neighbourhoods (0 = let the planner choose);
 *   "pg_gap": 2 = every copy's table leaves out its own block up to the length of the most-visited block (one launch
 *   per run of blocks with equal positions; measured slower), otherwise as many positions as the shortest block has
 *   (one launch, the default);
 *   "lv_seg": pool segments of the Levenshtein scan (0 = choose);
 *   "lv_two_pass": 0 = the Levenshtein scan runs once under the Hamming seeds; 1 (default) = where most seeds sit just
 *   above the next narrower band, first under that band's bound, then again for the candidates whose list stayed short;
 *   "pg_cut": K1P's id cut-off for hits at the layout's radius (0 = choose, 1..99 = that percentage of the ids, 100 = off);
 *   "pg_prefix": 0 = K1P's incomplete rows go straight to the plain re-scan (A/B, tests); 1 (default) = first a scan of
 *   the pool's first part, which holds the few neighbours at r + 1 such a row still needs (gm_pigeon.hip);
 *   "pg_sample": 1 (default) = K1P's layout planner measures how far the k-th neighbour lies on a scanned sample of the
 *   candidates (every (n_q / 2048)-th row against the whole pool, once per plan, for n_q >= 32768); 0 = it uses the
 *   independent-base model of the pool's composition alone (A/B, tests).  The result is exact either way;
 *   "pg_msplit": lanes per candidate in K1P's merge (0 = choose, 1, 2, 4);
 *   "pg_hit": K1P's hit path: 0 = every hit is stored to a hit buffer in HBM and the merge selects (k_hamming_scan_pg);
 *   2 = every lane keeps its candidate's k best keys in registers, hits reach it through a mailbox in LDS and the scan
 *   writes k keys per list (k_hamming_scan_pg4); 1 (default) = chosen per cascade level (register lists for k <= 20 on
 *   >= 400 000 candidates);
 *   "pg_spec": 1 (default) = a query of the same shape (n_q, k) as the handle's previous one sizes the follow-up launches of
 *   the rows each K1P level leaves short from that query's counts (+ a quarter + 2048 as an upper bound; the kernels clamp
 *   to the device's counts) instead of waiting for them, and checks the counts once after everything is enqueued (a
 *   level that left more runs the query again the waiting way); 0 = always wait (A/B, tests);
 *   "pg_per": target buckets per K1P workgroup (0 = choose: 16 / pg_sub, all of them with register lists);
 *   ("pg_debug", a round-3 knob that made the scan skip work for timing and its results WRONG, is gone from the library:
 *   it answers GM_ERR_ARG; such timings are separate builds, -DGM_PG_TIMING in guidemaker_amd/csrc/Makefile's CXXFLAGS);
 *   "pg_cascade": K1P levels for the rows a pass leaves short: 0 = never (they take the plain / prefix re-scan), 1 =
 *   when the planner's sample says a deeper level pays (default), 2 = whenever >= 64 rows are left (tests);
 *   "pg_layout2": forced layout of the deeper levels, digits as "pg_layout" (tests);
 *   "lp_radius": K2P, the displaced-block partition of the Levenshtein scan: -1 = never, 0 = choose from a sample on
 *   pools of >= 65 536 targets (default), 1 .. 7 = that radius (tests);
 *   "lp_radius2": K2P's second level (the rows the first leaves short, through the layout of a larger radius): -1 = never,
 *   0 = where the planner's sample says it pays (default), 1 .. 7 = that radius (tests). */
int gm_set_param(gm_index* idx, const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GMHIP_H */
