// gm_pg_types.h -- plain structs shared by the K1P kernels (gm_pigeon_scan.hip.h) and their host side (gm_pigeon.hip).
#pragma once
#include <stdint.h>

constexpr int GM_PG_MAXB = 8;        // blocks per layout
constexpr int GM_PG_MAXLEN = 6;      // bases per block (4^6 buckets)
constexpr int GM_PG_MAXLISTS = 64;   // hit lists per candidate
constexpr int GM_PG_ALIGN = 128;     // every bucket of a block-sorted pool copy starts on a multiple of 128 positions
                                     // (= 2 groups of 64 targets = one LDS-DMA unit of the scan)

// One block-sorted pool copy per block b: the pool stable-sorted by the block's value, every bucket padded to a
// multiple of GM_PG_ALIGN positions (padding = targets that match nothing).  Passed to the kernels by value.
struct gm_pg_blocks {
    const uint32_t* table[GM_PG_MAXB];   // one-hot table of copy b (K1L group format; padding positions all-zero)
    const uint4* recs[GM_PG_MAXB];       // per position {plane.x, plane.y, pool id, 0}; id = 0xffffffff on padding
    const int32_t* goff[GM_PG_MAXB];     // bucket v starts at 64-target group goff[v] (even)
    const int32_t* gcnt[GM_PG_MAXB];     // ... and holds targets in its first gcnt[v] groups
    const uint32_t* cperm[GM_PG_MAXB];   // candidates sorted by block b: sorted position -> candidate row
    int skip[GM_PG_MAXB];                // copy b's table leaves out sequence positions [skip[b], skip[b] + skip_len)
    int skip_len;
    int n_blocks;                        // B
    int n_expand;                        // X: blocks 0 .. X-1 also scan the one-substitution neighbour buckets
    int xlen;                            // bases of an expansion block (3 * xlen neighbour buckets per bucket)
    int n_sub;                           // S: the 1 + 3 * xlen target buckets of an expansion block's chunk are split
                                         // over S workgroups; list of (b < X, s) = b * S + s, of b >= X: X * S + b - X
    int n_lists;
    int run;                             // consecutive chunk-table entries executed on the same XCD
    int debug;                           // timing experiments only (gm_set_param "pg_debug"): bit 0 = skip the list stores
    int row_words;                       // words of all lists of one candidate: its lists are contiguous in the hit buffer
    int lcap[GM_PG_MAXLISTS];            // words of list l (word 0 = hit count, then the keys); a multiple of 4
    int loff[GM_PG_MAXLISTS];            // sum of lcap over the lists before l: list l of candidate q starts at word
                                         // q * row_words + loff[l] of the hit buffer
};

// one workgroup's work: cn (<= 64) candidates starting at sorted position c0 of block b's candidate order, all
// from bucket v, against the target buckets e_lo .. e_hi - 1 of copy b (0 = bucket v itself, 1 + 3 * p + a = the
// neighbour bucket whose base at block position p is replaced by the a-th other base); hits go to list lst.
struct gm_pg_chunk {
    int b, v, c0, cn, e_lo, e_hi, lst, pad;
};
