// gm_pg_types.h -- plain structs shared by the K1P kernels (gm_pigeon_scan.hip.h) and their host side (gm_pigeon.hip).
#pragma once
#include <stdint.h>

constexpr int GM_PG_MAXB = 8;        // blocks per layout
constexpr int GM_PG_MAXLEN = 8;      // bases per block (4^8 buckets)
constexpr int GM_PG_MAXLISTS = 64;   // hit lists per candidate
constexpr int GM_PG_ALIGN = 128;     // every bucket of a block-sorted pool copy starts on a multiple of 128 positions
                                     // (= 2 groups of 64 targets = one LDS-DMA unit of the scan)

// One block-sorted pool copy per block b: the pool stable-sorted by the block's value, every bucket padded to a
// multiple of GM_PG_ALIGN positions (padding = targets that match nothing).  Passed to the kernels by value.
struct gm_pg_blocks {
    const uint32_t* table[GM_PG_MAXB];   // one-hot table of copy b (K1L group format; padding positions all-zero)
    const uint4* recs[GM_PG_MAXB];       // per position {plane.x, plane.y, pool id, 0}; id = 0xffffffff on padding
    const int2* bdesc[GM_PG_MAXB];       // bucket v: .x = first 64-target group (even), .y = groups that hold targets
    const uint32_t* cperm[GM_PG_MAXB];   // candidates sorted by block b: sorted position -> candidate row
    int skip[GM_PG_MAXB];                // copy b's table leaves out sequence positions [skip[b], skip[b] + skiplen[b]):
    int skiplen[GM_PG_MAXB];             // its own block (all of it unless fewer than 10 positions would remain)
    const uint8_t* cflag;                // id cut-off: 1 = candidate row takes it (dense enough neighbourhood), 0 = it does not
    const int* bcut[GM_PG_MAXB];         // id cut-off (k_hamming_scan_pg<L, true>): groups of bucket v that hold an id < cut_id
    uint32_t cut_id;                     // hits at distance exactly r are taken only from targets with id < cut_id
                                         // (0xffffffff: no cut-off)
    int n_blocks;                        // B
    int blen[GM_PG_MAXB];                // bases of block b (the kernel decodes neighbour buckets with it)
    int trad[GM_PG_MAXB];                // search radius of block b (a pair is REPORTED by the first block that finds it)
    int n_lists;
    int run;                             // consecutive chunk-table entries executed on the same XCD
    int row_words;                       // words of one candidate's row of the hit buffer: [keys 0..3 of every list][keys 4.. of list 0] ...
    int lcap[GM_PG_MAXLISTS];            // key capacity of list l (a multiple of 4, at least 4)
    int loff[GM_PG_MAXLISTS];            // keys 0..3 of list l of candidate q: word q * row_words + 4 l; keys 4..: q * row_words + loff[l]
    uint32_t* hdr;                       // [n_q][hdr_words]: the keys appended to list l of candidate q (more than lcap[l]: the
                                         // list overflowed), written once by the list's workgroup -- a dense array of its own:
                                         // the merge reads the headers of consecutive candidates from consecutive lines
    int hdr_words;                       // n_lists rounded up to a multiple of 4
};

// A layout: B blocks of blen[b] bases (sum = guide length) with a search radius trad[b] per block: 0 = only the
// candidates' own bucket is scanned, 1 = also its 3 len one-substitution neighbour buckets, 2 = also its
// 9 C(len, 2) two-substitution neighbour buckets.  A pair that is NOT found under any block has more than trad[b]
// mismatches in every block, so every pair within r = sum(trad[b] + 1) - 1 is found.
struct gm_pg_layout {
    int B;
    int blen[GM_PG_MAXB];
    int trad[GM_PG_MAXB];
};
__host__ __device__ constexpr int gm_pg_buckets_of(int len, int t) {      // target buckets per candidate bucket
    return 1 + (t >= 1 ? 3 * len : 0) + (t >= 2 ? 9 * (len * (len - 1) / 2) : 0);
}

// one workgroup's work: cn (<= 64) candidates starting at sorted position c0 of block b's candidate order, all
// from bucket v, against the target buckets e_lo .. e_hi - 1 of copy b (numbering: pg_target_bucket); hits go to
// list lst.
struct gm_pg_chunk {
    int b, v, c0, cn, e_lo, e_hi, lst, pad;
};

// ---- K2P: the pigeonhole partition for Levenshtein (gm_levpart.hip; kernel k_leven_part_bs in gm_leven_bs.hip.h) ----
// Two sequences of the same length within edit distance r: cut the candidate into r + 1 blocks, at least one of them is
// untouched by every edit and appears in the target unchanged, displaced by at most floor(r / 2) positions (an alignment
// of equal-length sequences spends as many insertions as deletions).  Per block j and displacement d the pool is kept
// sorted by the bases it carries at the block's positions + d ("copy"); a chunk of candidates with the block value v is
// compared -- by the bit-sliced banded DP of K2B -- with bucket v of every copy of its block only.
constexpr int GM_LP_MAXBLK = 8;       // blocks (radius r <= 7)
constexpr int GM_LP_MAXCOPY = 56;     // (block, displacement) pairs: 8 x 7
constexpr int GM_LP_TILE_TARGETS = 256;      // buckets of a copy are padded to whole K2B tiles (4 groups of 64)
struct gm_lp_copies {                 // passed to the kernel by value
    const uint32_t* table[GM_LP_MAXCOPY];     // one-hot table of the copy (K1L group format, all L positions; padding all-zero)
    const uint32_t* ord[GM_LP_MAXCOPY];       // position in the copy -> pool id (0xffffffff = padding)
    const int2* bdesc[GM_LP_MAXCOPY];         // bucket v -> (first tile, 64-target groups that hold targets)
    const uint32_t* cperm[GM_LP_MAXBLK];      // candidates sorted by block j's value: sorted position -> candidate row
    int first_copy[GM_LP_MAXBLK + 1];         // the copies of block j: [first_copy[j], first_copy[j + 1])
    int n_blocks;
};
struct gm_lp_chunk {                  // one workgroup: cn (<= 64) candidates from sorted position c0 of block j, all with value v
    int j, v, c0, cn;
};
