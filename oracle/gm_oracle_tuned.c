/*
 * gm_oracle_tuned.c -- FAST exact k-NN scans on the CPU (Hamming; Levenshtein at the end of the file).  TEST INFRASTRUCTURE ONLY (see gm_oracle.c's header: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so).
 *
 * Why it exists: gm_oracle.c restates nmslib's brute_force literally (three 32-bit one-hot words per pair, built for
 * x86-64-v2) and does 7 10^7 pairs/s per core -- a whole BASELINE-sized pool (S1: 3.6 10^11 pairs, PAO1: 1.4 10^12) is
 * minutes to hours of CPU there.  This file computes THE SAME answer -- exact scan, results by (distance, db index),
 * distances in nmslib's raw bit_hamming units (2 x mismatching bases), -1 padding -- from two bit planes per sequence
 * (mismatches = popcount((a.lo ^ b.lo) | (a.hi ^ b.hi))) with 16 targets per AVX-512 VPOPCNTDQ instruction (or 8 per
 * AVX2 nibble-table popcount, or scalar popcnt), chosen at run time.  It is what the GPU suite checks WHOLE pools against
 * (tests/test_gpu_fullsize.py) and what bench.py reports as cpu_baseline kind "tuned".
 *
 * Pinning: tests/test_oracle.py::test_tuned_oracle_equals_the_port compares it with orc_knn (the pinned restatement) on
 * random, skewed, duplicate-rich and golden (Carsonella) pools, every k and guide length; so it inherits the port's pins
 * (reference known answers: /root/reference/tests/test_core.py:126,266,346,347; brute force as truth:
 * /root/reference/tests/GridOptimization.ipynb:153-163; one-hot encoding: /root/reference/guidemaker/core.py:379-386).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QB 8       /* queries per pass over the pool (their planes and bounds stay in registers) */
#define KMAX 64

static int plane_pack(const char* seqs, int64_t n, int L, uint32_t* lo, uint32_t* hi) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t a = 0, b = 0;
        for (int p = 0; p < L; ++p) {
            int c;
            switch (seqs[i * L + p]) {
                case 'A': c = 0; break;
                case 'C': c = 1; break;
                case 'G': c = 2; break;
                case 'T': c = 3; break;
                default: return -1;
            }
            a |= (uint32_t)(c & 1) << p;
            b |= (uint32_t)(c >> 1) << p;
        }
        lo[i] = a;
        hi[i] = b;
    }
    return 0;
}

typedef struct {
    int32_t d[KMAX], id[KMAX];
    int cnt;
} topk_t;

/* sorted insert; ids arrive ascending, so an equal distance never displaces an entry (canonical (distance, id) order) */
static inline void tk_insert(topk_t* t, int k, int32_t nd, int32_t nid) {
    int c = t->cnt;
    if (c == k) {
        if (nd >= t->d[k - 1]) return;
        c = k - 1;
    }
    int p = c;
    while (p > 0 && t->d[p - 1] > nd) {
        t->d[p] = t->d[p - 1];
        t->id[p] = t->id[p - 1];
        --p;
    }
    t->d[p] = nd;
    t->id[p] = nid;
    t->cnt = c + 1;
}
/* a pair enters when its distance is below this */
static inline uint32_t tk_bound(const topk_t* t, int k) { return k > 0 && t->cnt == k ? (uint32_t)t->d[k - 1] : 0x7fffffffu; }

/* ---- scalar ---- */
static void scan_scalar(const uint32_t* lo, const uint32_t* hi, int64_t n_db, const uint32_t* qlo, const uint32_t* qhi, int nq,
                        int k, topk_t* tk) {
    for (int qi = 0; qi < nq; ++qi) {
        uint32_t bound = tk_bound(&tk[qi], k);
        const uint32_t a = qlo[qi], b = qhi[qi];
        for (int64_t j = 0; j < n_db; ++j) {
            const uint32_t d = (uint32_t)__builtin_popcount((lo[j] ^ a) | (hi[j] ^ b));
            if (d < bound) {
                tk_insert(&tk[qi], k, (int32_t)d, (int32_t)j);
                bound = tk_bound(&tk[qi], k);
            }
        }
    }
}

/* ---- AVX2: popcount of 8 x u32 by the nibble table ---- */
__attribute__((target("avx2"))) static inline __m256i popcnt8_avx2(__m256i v) {
    const __m256i lut = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4, 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4);
    const __m256i m4 = _mm256_set1_epi8(0x0f);
    __m256i c = _mm256_add_epi8(_mm256_shuffle_epi8(lut, _mm256_and_si256(v, m4)),
                                _mm256_shuffle_epi8(lut, _mm256_and_si256(_mm256_srli_epi16(v, 4), m4)));
    /* bytes -> dwords */
    c = _mm256_add_epi8(c, _mm256_srli_epi32(c, 8));
    c = _mm256_add_epi8(c, _mm256_srli_epi32(c, 16));
    return _mm256_and_si256(c, _mm256_set1_epi32(0xff));
}
__attribute__((target("avx2"))) static void scan_avx2(const uint32_t* lo, const uint32_t* hi, int64_t n_db, const uint32_t* qlo,
                                                      const uint32_t* qhi, int nq, int k, topk_t* tk) {
    const int64_t n8 = n_db & ~(int64_t)7;
    uint32_t bound[QB];
    for (int qi = 0; qi < nq; ++qi) bound[qi] = tk_bound(&tk[qi], k);
    for (int64_t j = 0; j < n8; j += 8) {
        const __m256i tl = _mm256_loadu_si256((const __m256i*)(lo + j)), th = _mm256_loadu_si256((const __m256i*)(hi + j));
        for (int qi = 0; qi < nq; ++qi) {
            const __m256i v = _mm256_or_si256(_mm256_xor_si256(tl, _mm256_set1_epi32((int)qlo[qi])),
                                              _mm256_xor_si256(th, _mm256_set1_epi32((int)qhi[qi])));
            const __m256i c = popcnt8_avx2(v);
            int m = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32((int)bound[qi]), c)));
            if (m) {
                uint32_t cc[8];
                _mm256_storeu_si256((__m256i*)cc, c);
                while (m) {
                    const int t = __builtin_ctz(m);
                    m &= m - 1;
                    if (cc[t] < bound[qi]) {
                        tk_insert(&tk[qi], k, (int32_t)cc[t], (int32_t)(j + t));
                        bound[qi] = tk_bound(&tk[qi], k);
                    }
                }
            }
        }
    }
    for (int qi = 0; qi < nq; ++qi)
        for (int64_t j = n8; j < n_db; ++j) {
            const uint32_t d = (uint32_t)__builtin_popcount((lo[j] ^ qlo[qi]) | (hi[j] ^ qhi[qi]));
            if (d < tk_bound(&tk[qi], k)) tk_insert(&tk[qi], k, (int32_t)d, (int32_t)j);
        }
}

/* ---- AVX-512 VPOPCNTDQ: 16 targets per instruction ---- */
__attribute__((target("avx512f,avx512vpopcntdq"))) static void scan_avx512(const uint32_t* lo, const uint32_t* hi, int64_t n_db,
                                                                          const uint32_t* qlo, const uint32_t* qhi, int nq, int k,
                                                                          topk_t* tk) {
    const int64_t n16 = n_db & ~(int64_t)15;
    uint32_t bound[QB];
    __m512i ql[QB], qh[QB], bd[QB];
    for (int qi = 0; qi < nq; ++qi) {
        bound[qi] = tk_bound(&tk[qi], k);
        ql[qi] = _mm512_set1_epi32((int)qlo[qi]);
        qh[qi] = _mm512_set1_epi32((int)qhi[qi]);
        bd[qi] = _mm512_set1_epi32((int)bound[qi]);
    }
    for (int64_t j = 0; j < n16; j += 16) {
        const __m512i tl = _mm512_loadu_si512((const void*)(lo + j)), th = _mm512_loadu_si512((const void*)(hi + j));
        for (int qi = 0; qi < nq; ++qi) {
            const __m512i c = _mm512_popcnt_epi32(_mm512_or_si512(_mm512_xor_si512(tl, ql[qi]), _mm512_xor_si512(th, qh[qi])));
            __mmask16 m = _mm512_cmplt_epu32_mask(c, bd[qi]);
            if (m) {
                uint32_t cc[16];
                _mm512_storeu_si512((void*)cc, c);
                unsigned mm = m;
                while (mm) {
                    const int t = __builtin_ctz(mm);
                    mm &= mm - 1;
                    if (cc[t] < bound[qi]) {
                        tk_insert(&tk[qi], k, (int32_t)cc[t], (int32_t)(j + t));
                        bound[qi] = tk_bound(&tk[qi], k);
                    }
                }
                bd[qi] = _mm512_set1_epi32((int)bound[qi]);
            }
        }
    }
    for (int qi = 0; qi < nq; ++qi)
        for (int64_t j = n16; j < n_db; ++j) {
            const uint32_t d = (uint32_t)__builtin_popcount((lo[j] ^ qlo[qi]) | (hi[j] ^ qhi[qi]));
            if (d < tk_bound(&tk[qi], k)) tk_insert(&tk[qi], k, (int32_t)d, (int32_t)j);
        }
}

/* 0 = scalar, 1 = AVX2, 2 = AVX-512 VPOPCNTDQ; force < 0: the best the CPU has */
static int pick_isa(int force) {
    int best = 0;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx2")) best = 1;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vpopcntdq")) best = 2;
    return force >= 0 && force < best ? force : best;
}
int orc_tuned_isa(void) { return pick_isa(-1); }

/*
 * Exact Hamming k-NN; same contract as orc_knn(metric = 0): db, q = rows of L chars (ACGT), ids / dists [n_q * k] padded
 * with -1, dists in RAW nmslib units (2 x mismatches), counts [n_q]; rows ordered by (distance, db index).
 * force_isa: -1 = best available (tests pin every path against the port).  Returns 0, -1 on bad input.
 */
int orc_knn_hamming_tuned(const char* db, int64_t n_db, const char* q, int64_t n_q, int L, int k, int32_t* ids, int32_t* dists,
                          int32_t* counts, int threads, int force_isa) {
    if (L < 1 || L > 32 || k < 1 || k > KMAX || n_db < 0 || n_q < 0 || n_db >= ((int64_t)1 << 31)) return -1;
    uint32_t* lo = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_db + 16));
    uint32_t* hi = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_db + 16));
    uint32_t* qlo = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_q + 1));
    uint32_t* qhi = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_q + 1));
    int rc = (!lo || !hi || !qlo || !qhi) ? -1 : 0;
    if (!rc && (plane_pack(db, n_db, L, lo, hi) || plane_pack(q, n_q, L, qlo, qhi))) rc = -1;
    if (rc) {
        free(lo); free(hi); free(qlo); free(qhi);
        return -1;
    }
    const int isa = pick_isa(force_isa);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const int64_t n_blocks = (n_q + QB - 1) / QB;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t q0 = b * QB;
        const int nq = (int)(n_q - q0 < QB ? n_q - q0 : QB);
        topk_t tk[QB];
        for (int i = 0; i < nq; ++i) tk[i].cnt = 0;
        if (isa == 2) scan_avx512(lo, hi, n_db, qlo + q0, qhi + q0, nq, k, tk);
        else if (isa == 1) scan_avx2(lo, hi, n_db, qlo + q0, qhi + q0, nq, k, tk);
        else scan_scalar(lo, hi, n_db, qlo + q0, qhi + q0, nq, k, tk);
        for (int i = 0; i < nq; ++i) {
            int32_t* od = dists + (q0 + i) * k;
            int32_t* oi = ids + (q0 + i) * k;
            for (int j = 0; j < k; ++j) {
                od[j] = j < tk[i].cnt ? 2 * tk[i].d[j] : -1;
                oi[j] = j < tk[i].cnt ? tk[i].id[j] : -1;
            }
            counts[q0 + i] = tk[i].cnt;
        }
    }
    free(lo); free(hi); free(qlo); free(qhi);
    return 0;
}


/* =====================================================================================================================
 * Levenshtein.  gm_oracle.c's orc_leven is nmslib's `leven` restated literally (two-row O(L^2) dynamic programme per pair,
 * /root/reference/guidemaker/core.py:461 is the call site): ~2 10^6 pairs/s per core, hours for a genome-sized pool.  Here
 * the SAME unit-cost edit distance between two L-character strings comes from the bit-vector recurrence of Myers (J. ACM
 * 46(3), 1999) in Hyyro's form for the global distance (the horizontal delta shifted into row 0 is +1): the candidate is
 * the pattern (Peq[c] = positions where it has base c), the pool target streams through it one character per step, and
 * 16 (AVX-512) / 8 (AVX2) / 1 targets advance per instruction, a target's character chosen per lane from its two bit
 * planes.  ~20 vector operations per step and vector.  Results, order and padding exactly as orc_knn(metric = 1).
 * Pinning: tests/test_oracle.py::test_tuned_leven_equals_the_port -- every path against orc_knn on pools with planted
 * substitutions AND indels, every guide length, so it inherits the port's pins (reference known answer
 * /root/reference/tests/test_core.py:346 = Levenshtein [0, 1, 2]).
 * ===================================================================================================================== */
static inline int lev_scalar(const uint32_t peq[4], uint32_t tlo, uint32_t thi, int L) {
    uint32_t pv = 0xffffffffu, mv = 0;
    int score = L;
    const uint32_t top = 1u << (L - 1);
    for (int j = 0; j < L; ++j) {
        const uint32_t eq = peq[((tlo >> j) & 1u) | (((thi >> j) & 1u) << 1)];
        const uint32_t xv = eq | mv;
        const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
        uint32_t ph = mv | ~(xh | pv);
        uint32_t mh = pv & xh;
        score += (ph & top) ? 1 : 0;
        score -= (mh & top) ? 1 : 0;
        ph = (ph << 1) | 1u;
        mh <<= 1;
        pv = mh | ~(xv | ph);
        mv = ph & xv;
    }
    return score;
}

static void lev_scan_scalar(const uint32_t* lo, const uint32_t* hi, int64_t j0, int64_t j1, const uint32_t peq[4], int L, int k,
                            topk_t* tk) {
    for (int64_t j = j0; j < j1; ++j) {
        const uint32_t d = (uint32_t)lev_scalar(peq, lo[j], hi[j], L);
        if (d < tk_bound(tk, k)) tk_insert(tk, k, (int32_t)d, (int32_t)j);
    }
}

__attribute__((target("avx2"))) static void lev_scan_avx2(const uint32_t* lo, const uint32_t* hi, int64_t n_db, const uint32_t peq[4],
                                                          int L, int k, topk_t* tk) {
    const int64_t n8 = n_db & ~(int64_t)7;
    const __m256i pA = _mm256_set1_epi32((int)peq[0]), pC = _mm256_set1_epi32((int)peq[1]), pG = _mm256_set1_epi32((int)peq[2]),
                  pT = _mm256_set1_epi32((int)peq[3]);
    const __m256i ones = _mm256_set1_epi32(-1), one = _mm256_set1_epi32(1), zero = _mm256_setzero_si256();
    const __m256i top = _mm256_set1_epi32((int)(1u << (L - 1)));
    uint32_t bound = tk_bound(tk, k);
    for (int64_t j = 0; j < n8; j += 8) {
        const __m256i tl = _mm256_loadu_si256((const __m256i*)(lo + j)), th = _mm256_loadu_si256((const __m256i*)(hi + j));
        __m256i pv = ones, mv = zero, score = _mm256_set1_epi32(L);
        for (int p = 0; p < L; ++p) {
            const __m256i bit = _mm256_set1_epi32((int)(1u << p));
            const __m256i ml = _mm256_cmpeq_epi32(_mm256_and_si256(tl, bit), bit), mh_ = _mm256_cmpeq_epi32(_mm256_and_si256(th, bit), bit);
            const __m256i eq = _mm256_blendv_epi8(_mm256_blendv_epi8(pA, pC, ml), _mm256_blendv_epi8(pG, pT, ml), mh_);
            const __m256i xv = _mm256_or_si256(eq, mv);
            const __m256i xh = _mm256_or_si256(_mm256_xor_si256(_mm256_add_epi32(_mm256_and_si256(eq, pv), pv), pv), eq);
            __m256i ph = _mm256_or_si256(mv, _mm256_andnot_si256(_mm256_or_si256(xh, pv), ones));
            __m256i mh = _mm256_and_si256(pv, xh);
            score = _mm256_sub_epi32(score, _mm256_cmpeq_epi32(_mm256_and_si256(ph, top), top));      /* - (-1) = + 1 */
            score = _mm256_add_epi32(score, _mm256_cmpeq_epi32(_mm256_and_si256(mh, top), top));
            ph = _mm256_or_si256(_mm256_slli_epi32(ph, 1), one);
            mh = _mm256_slli_epi32(mh, 1);
            pv = _mm256_or_si256(mh, _mm256_andnot_si256(_mm256_or_si256(xv, ph), ones));
            mv = _mm256_and_si256(ph, xv);
        }
        int m = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_set1_epi32((int)bound), score)));
        if (m) {
            uint32_t cc[8];
            _mm256_storeu_si256((__m256i*)cc, score);
            while (m) {
                const int t = __builtin_ctz(m);
                m &= m - 1;
                if (cc[t] < bound) {
                    tk_insert(tk, k, (int32_t)cc[t], (int32_t)(j + t));
                    bound = tk_bound(tk, k);
                }
            }
        }
    }
    lev_scan_scalar(lo, hi, n8, n_db, peq, L, k, tk);
}

__attribute__((target("avx512f"))) static void lev_scan_avx512(const uint32_t* lo, const uint32_t* hi, int64_t n_db, const uint32_t peq[4],
                                                               int L, int k, topk_t* tk) {
    const int64_t n16 = n_db & ~(int64_t)15;
    const __m512i pA = _mm512_set1_epi32((int)peq[0]), pC = _mm512_set1_epi32((int)peq[1]), pG = _mm512_set1_epi32((int)peq[2]),
                  pT = _mm512_set1_epi32((int)peq[3]);
    const __m512i ones = _mm512_set1_epi32(-1), one = _mm512_set1_epi32(1);
    const __m512i top = _mm512_set1_epi32((int)(1u << (L - 1)));
    uint32_t bound = tk_bound(tk, k);
    for (int64_t j = 0; j < n16; j += 16) {
        const __m512i tl = _mm512_loadu_si512((const void*)(lo + j)), th = _mm512_loadu_si512((const void*)(hi + j));
        __m512i pv = ones, mv = _mm512_setzero_si512(), score = _mm512_set1_epi32(L);
        for (int p = 0; p < L; ++p) {
            const __m512i bit = _mm512_set1_epi32((int)(1u << p));
            const __mmask16 kl = _mm512_test_epi32_mask(tl, bit), kh = _mm512_test_epi32_mask(th, bit);
            const __m512i eq = _mm512_mask_blend_epi32(kh, _mm512_mask_blend_epi32(kl, pA, pC), _mm512_mask_blend_epi32(kl, pG, pT));
            const __m512i xv = _mm512_or_si512(eq, mv);
            const __m512i xh = _mm512_or_si512(_mm512_xor_si512(_mm512_add_epi32(_mm512_and_si512(eq, pv), pv), pv), eq);
            __m512i ph = _mm512_or_si512(mv, _mm512_andnot_si512(_mm512_or_si512(xh, pv), ones));
            __m512i mh = _mm512_and_si512(pv, xh);
            score = _mm512_mask_add_epi32(score, _mm512_test_epi32_mask(ph, top), score, one);
            score = _mm512_mask_sub_epi32(score, _mm512_test_epi32_mask(mh, top), score, one);
            ph = _mm512_or_si512(_mm512_slli_epi32(ph, 1), one);
            mh = _mm512_slli_epi32(mh, 1);
            pv = _mm512_or_si512(mh, _mm512_andnot_si512(_mm512_or_si512(xv, ph), ones));
            mv = _mm512_and_si512(ph, xv);
        }
        unsigned mm = _mm512_cmplt_epu32_mask(score, _mm512_set1_epi32((int)bound));
        if (mm) {
            uint32_t cc[16];
            _mm512_storeu_si512((void*)cc, score);
            while (mm) {
                const int t = __builtin_ctz(mm);
                mm &= mm - 1;
                if (cc[t] < bound) {
                    tk_insert(tk, k, (int32_t)cc[t], (int32_t)(j + t));
                    bound = tk_bound(tk, k);
                }
            }
        }
    }
    lev_scan_scalar(lo, hi, n16, n_db, peq, L, k, tk);
}

/* 0 = scalar, 1 = AVX2, 2 = AVX-512F (no VPOPCNTDQ needed here) */
static int pick_isa_lev(int force) {
    int best = 0;
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx2")) best = 1;
    if (__builtin_cpu_supports("avx512f")) best = 2;
    return force >= 0 && force < best ? force : best;
}

/*
 * Exact Levenshtein k-NN; same contract as orc_knn(metric = 1): db, q = rows of L chars (ACGT), ids / dists [n_q * k] padded
 * with -1, dists = unit-cost edit distances, counts [n_q]; rows ordered by (distance, db index).  Returns 0, -1 on bad input.
 */
int orc_knn_leven_tuned(const char* db, int64_t n_db, const char* q, int64_t n_q, int L, int k, int32_t* ids, int32_t* dists,
                        int32_t* counts, int threads, int force_isa) {
    if (L < 1 || L > 32 || k < 1 || k > KMAX || n_db < 0 || n_q < 0 || n_db >= ((int64_t)1 << 31)) return -1;
    uint32_t* lo = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_db + 16));
    uint32_t* hi = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_db + 16));
    uint32_t* qlo = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_q + 1));
    uint32_t* qhi = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n_q + 1));
    int rc = (!lo || !hi || !qlo || !qhi) ? -1 : 0;
    if (!rc && (plane_pack(db, n_db, L, lo, hi) || plane_pack(q, n_q, L, qlo, qhi))) rc = -1;
    if (rc) {
        free(lo); free(hi); free(qlo); free(qhi);
        return -1;
    }
    const int isa = pick_isa_lev(force_isa);
    const uint32_t lmask = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < n_q; ++i) {
        /* pattern = the candidate: positions where it has A / C / G / T */
        const uint32_t a = qlo[i], b = qhi[i];
        const uint32_t peq[4] = {~a & ~b & lmask, a & ~b & lmask, ~a & b & lmask, a & b & lmask};
        topk_t tk;
        tk.cnt = 0;
        if (isa == 2) lev_scan_avx512(lo, hi, n_db, peq, L, k, &tk);
        else if (isa == 1) lev_scan_avx2(lo, hi, n_db, peq, L, k, &tk);
        else lev_scan_scalar(lo, hi, 0, n_db, peq, L, k, &tk);
        int32_t* od = dists + i * k;
        int32_t* oi = ids + i * k;
        for (int j = 0; j < k; ++j) {
            od[j] = j < tk.cnt ? tk.d[j] : -1;
            oi[j] = j < tk.cnt ? tk.id[j] : -1;
        }
        counts[i] = tk.cnt;
    }
    free(lo); free(hi); free(qlo); free(qhi);
    return 0;
}
int orc_tuned_isa_leven(void) { return pick_isa_lev(-1); }
