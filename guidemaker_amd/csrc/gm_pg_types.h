// gm_pg_types.h -- plain structs shared by the K1P kernel (gm_lookup.hip.h) and its host side (gm_pigeon.hip).
#pragma once
#include <stdint.h>

constexpr int GM_PG_MAXB = 8;     // blocks per layout
constexpr int GM_PG_MAXLEN = 6;   // bases per block (4^6 buckets)

// per-block device pointers, passed to the kernel by value
struct gm_pg_blocks {
    const uint32_t* table[GM_PG_MAXB];   // one-hot table of the pool copy sorted by block b
    const int32_t* perm[GM_PG_MAXB];     // position in that copy -> pool id
    const int32_t* toff[GM_PG_MAXB];     // bucket v covers positions [toff[v], toff[v+1])
    const uint32_t* cperm[GM_PG_MAXB];   // candidates sorted by block b: sorted position -> candidate row
    int skip[GM_PG_MAXB];                // copy b's table leaves out sequence positions [skip[b], skip[b] + skip_len)
    int skip_len;
    int n_blocks;                        // B: slot lists 0 .. B-1 have `cap` slots, lists B .. have `cap_x`
    int xne;                             // neighbour lists per expansion block (3 x its length); list B + x * xne + e
                                         // belongs to expansion block x and is served by copy x
    int n_expand;                        // neighbour lists in total (expansion blocks x xne, or 0)
};

// one workgroup's work: cn (<= 64) candidates starting at sorted position c0 of block b's candidate order,
// against target bucket v of copy b.  lst = slot list the hits go to (b for the candidates' own bucket,
// B + e for the e-th one-substitution neighbour bucket of the expansion block); add = mismatches the pair is
// known to have on positions the copy's table leaves out (1 for a neighbour bucket that differs there).
struct gm_pg_chunk {
    int b, v, c0, cn, lst, add;
};
