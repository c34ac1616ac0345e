"""CPU ORACLE for the guide-RNA neighbour-search path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
guidemaker_amd/ never does.  It restates, on the CPU, what the reference computes on the hot
path (citations into /root/reference/):

  guidemaker/core.py:379-386   _one_hot_encode
  guidemaker/core.py:418-467   create_index      (nmslib bit_hamming / leven index)
  guidemaker/core.py:471-523   get_neighbors     (knnQueryBatch + threshold on the 2nd hit)
  guidemaker/core.py:586-617   get_control_seqs  (k=2 query, nearest distance, sort, top n)
  guidemaker/cfd_score_calculator.py:60-85  calc_cfd
  guidemaker/core.py:1129-1148 cfd_score

The arithmetic of nmslib==2.1.1 (absent third-party dependency, requirements.txt:31) is
restated from its published algorithm (see gm_oracle.c).  The search is EXACT (nmslib
'brute_force', the reference's own ground truth, tests/GridOptimization.ipynb:153-163), not HNSW;
ties are ordered (distance, db index); the db order is first occurrence in the targets list
instead of the hash order of `list(set(...))` (core.py:446).

Parity pinning: tests/test_oracle.py checks this module against every known answer the
reference's tests hold for the path (tests/test_core.py:126,266,346,347) and against the golden
vectors generated from the reference (tests/golden/).
"""
import ctypes
import os
import statistics
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so (gcc, OpenMP) and the HNSW baseline (g++, OpenMP)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so", "libhnsw_baseline.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i64, i32p = ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)
        L.orc_knn.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64, ctypes.c_int, ctypes.c_int,
                              ctypes.c_int, i32p, i32p, i32p, ctypes.c_int]
        L.orc_knn.restype = ctypes.c_int
        L.orc_dist_matrix.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64, ctypes.c_int,
                                      ctypes.c_int, i32p]
        L.orc_dist_matrix.restype = ctypes.c_int
        L.orc_cfd.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                              ctypes.POINTER(ctypes.c_double)]
        L.orc_cfd.restype = ctypes.c_double
        L.orc_cfd_batch.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i64, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.orc_cfd_batch.restype = None
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_knn_hamming_tuned.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64, ctypes.c_int, ctypes.c_int,
                                            i32p, i32p, i32p, ctypes.c_int, ctypes.c_int]
        L.orc_knn_hamming_tuned.restype = ctypes.c_int
        L.orc_tuned_isa.restype = ctypes.c_int
        L.orc_knn_leven_tuned.argtypes = [ctypes.c_char_p, i64, ctypes.c_char_p, i64, ctypes.c_int, ctypes.c_int,
                                          i32p, i32p, i32p, ctypes.c_int, ctypes.c_int]
        L.orc_knn_leven_tuned.restype = ctypes.c_int
        L.orc_tuned_isa_leven.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _rows(seqs, L):
    """list[str] or (n, L) uint8 array -> contiguous bytes of n*L chars."""
    if isinstance(seqs, np.ndarray):
        assert seqs.dtype == np.uint8 and seqs.ndim == 2 and seqs.shape[1] == L
        return np.ascontiguousarray(seqs).tobytes(), seqs.shape[0]
    seqs = list(seqs)
    for s in seqs:
        assert len(s) == L, "all sequences must have length %d" % L
    return "".join(seqs).encode(), len(seqs)


# ------------------------------------------------------------------ distances, pure Python
_CHARMAP = {'A': '1 0 0 0', 'C': '0 1 0 0', 'G': '0 0 1 0', 'T': '0 0 0 1'}


def one_hot_encode(seq_list):
    """core.py:379-386."""
    return [" ".join(_CHARMAP[c] for c in s) for s in seq_list]


def py_bit_hamming(onehot_a, onehot_b):
    """nmslib bit_hamming on two one-hot strings: number of differing flags (= 2 x mismatches)."""
    fa, fb = onehot_a.split(" "), onehot_b.split(" ")
    assert len(fa) == len(fb)
    return sum(x != y for x, y in zip(fa, fb))


def py_leven(a, b):
    """nmslib leven: textbook DP."""
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j - 1] + (a[i - 1] != b[j - 1]), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[len(b)]


def np_hamming_matrix(q, db):
    """(nq, L) x (ndb, L) uint8 char arrays -> mismatching-base counts (true units), numpy."""
    return (q[:, None, :] != db[None, :, :]).sum(axis=2).astype(np.int32)


# ------------------------------------------------------------------ exact k-NN (C, OpenMP)
def knn(db, q, L, metric, k, threads=0):
    """Exact knnQueryBatch.  metric 'hamming' -> RAW nmslib units (2 x mismatches); else leven.

    Returns (ids [nq,k] int32 (-1 padded), dists [nq,k] int32 (-1 padded), counts [nq] int32),
    rows ordered by (distance, db index).
    """
    dbb, ndb = _rows(db, L)
    qb, nq = _rows(q, L)
    ids = np.empty((nq, k), np.int32)
    dists = np.empty((nq, k), np.int32)
    counts = np.empty((nq,), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    rc = lib().orc_knn(dbb, ndb, qb, nq, L, 0 if metric == "hamming" else 1, k,
                       p(ids), p(dists), p(counts), threads)
    if rc != 0:
        raise ValueError("oracle: bad input (non-ACGT base or bad argument)")
    return ids, dists, counts


def knn_tuned(db, q, L, k, threads=0, isa=-1):
    """The same answer as knn(..., 'hamming', ...) -- exact, (distance, db index) order, RAW nmslib units -- from
    gm_oracle_tuned.c (bit planes, 16 targets per AVX-512 VPOPCNTDQ instruction / AVX2 / scalar, chosen at run time;
    isa forces a slower path: 0 scalar, 1 AVX2).  Pinned against knn() by tests/test_oracle.py; used where a WHOLE
    genome-sized pool is checked (tests/test_gpu_fullsize.py) and as bench.py's cpu_baseline kind "tuned"."""
    dbb, ndb = _rows(db, L)
    qb, nq = _rows(q, L)
    ids = np.empty((nq, k), np.int32)
    dists = np.empty((nq, k), np.int32)
    counts = np.empty((nq,), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    rc = lib().orc_knn_hamming_tuned(dbb, ndb, qb, nq, L, k, p(ids), p(dists), p(counts), threads, isa)
    if rc != 0:
        raise ValueError("oracle: bad input (non-ACGT base or bad argument)")
    return ids, dists, counts


def knn_leven_tuned(db, q, L, k, threads=0, isa=-1):
    """The same answer as knn(..., 'leven', ...) -- exact, (distance, db index) order -- from gm_oracle_tuned.c (Myers /
    Hyyro bit-vector recurrence, 16 targets per AVX-512 instruction / 8 per AVX2 / scalar; isa forces a slower path: 0
    scalar, 1 AVX2).  Pinned against knn() by tests/test_oracle.py; used where a WHOLE genome-sized Levenshtein result is
    checked (tests/test_gpu_fullsize.py)."""
    dbb, ndb = _rows(db, L)
    qb, nq = _rows(q, L)
    ids = np.empty((nq, k), np.int32)
    dists = np.empty((nq, k), np.int32)
    counts = np.empty((nq,), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    rc = lib().orc_knn_leven_tuned(dbb, ndb, qb, nq, L, k, p(ids), p(dists), p(counts), threads, isa)
    if rc != 0:
        raise ValueError("oracle: bad input (non-ACGT base or bad argument)")
    return ids, dists, counts


def tuned_isa():
    return {0: "scalar popcnt", 1: "AVX2", 2: "AVX-512 VPOPCNTDQ"}[lib().orc_tuned_isa()]


def dist_matrix(db, q, L, metric):
    dbb, ndb = _rows(db, L)
    qb, nq = _rows(q, L)
    out = np.empty((nq, ndb), np.int32)
    rc = lib().orc_dist_matrix(dbb, ndb, qb, nq, L, 0 if metric == "hamming" else 1,
                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc != 0:
        raise ValueError("oracle: bad input")
    return out


# ------------------------------------------------------------------ CFD
def calc_cfd(wt, off, mm):
    """cfd_score_calculator.py:60-85 on the 20x4x4 table (tests/golden/cfd_table.json)."""
    assert len(wt) == len(off), "The lengths wt and off differ: wt = {}, off = {}".format(
        str(len(wt)), str(len(off)))
    guidelen = len(wt)
    score = 1.
    comp = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}
    for i, sl in enumerate(off):
        if (guidelen - 20 - i) <= 0:
            if wt[i] != sl:
                pos = 20 + i + 1 - guidelen
                score *= mm[pos - 1]["ACGT".index(wt[i])]["ACGT".index(comp[sl])]
    return score


def cfd_batch(wts, offs, L, mm):
    table = np.ascontiguousarray(np.asarray(mm, dtype=np.float64).reshape(-1))
    assert table.size == 320
    wb, n = _rows(wts, L)
    ob, n2 = _rows(offs, L)
    assert n == n2
    out = np.empty((n,), np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    lib().orc_cfd_batch(wb, ob, n, L, table.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


# ------------------------------------------------------------------ the reference's operator level
class OracleTargetProcessor:
    """Restatement of TargetProcessor's neighbour-search methods (core.py:295-633) on plain lists.

    targets            list[str], frame order (duplicates allowed)
    isseedduplicated   list[bool] or None  (find_unique_near_pam output, core.py:416)
    hasrestrictionsite list[bool] or None  (check_restriction_enzymes output, core.py:375-377)
    """

    def __init__(self, targets, dtype="hamming", editdist=2, knum=2,
                 isseedduplicated=None, hasrestrictionsite=None):
        self.targets = list(targets)
        self.dtype = dtype
        self.editdist = editdist
        self.knum = knum
        n = len(self.targets)
        self.isseedduplicated = list(isseedduplicated) if isseedduplicated is not None else [False] * n
        self.hasrestrictionsite = list(hasrestrictionsite) if hasrestrictionsite is not None else [False] * n
        self.db = None
        self.neighbors = {}
        self.L = len(self.targets[0])

    def create_index(self):
        # core.py:446 dedups with set(); the oracle keeps first-occurrence order
        self.db = list(dict.fromkeys(self.targets))

    def query_list(self):
        # core.py:495 -- note the OR
        return [t for t, sd, rs in zip(self.targets, self.isseedduplicated, self.hasrestrictionsite)
                if (sd == False) or (rs == False)]

    def knn_raw(self, queries, k, threads=0):
        metric = "hamming" if self.dtype == "hamming" else "leven"
        return knn(self.db, queries, self.L, metric, k, threads)

    def get_neighbors(self, threads=0):
        unique_targets = self.query_list()
        ids, dists, counts = self.knn_raw(unique_targets, self.knum, threads)
        neighbor_dict = {}
        for i, queryseq in enumerate(unique_targets):
            c = int(counts[i])
            hitseqidx = ids[i, :c].tolist()
            editdist = dists[i, :c].tolist()
            if len(editdist) < 2:
                raise IndexError("list index out of range")     # core.py:512 editdist[1]
            if self.dtype == "hamming":
                if editdist[1] >= 2 * self.editdist:              # core.py:512
                    neighbor_dict[queryseq] = {
                        "target": queryseq,
                        "neighbors": {"seqs": [self.db[x] for x in hitseqidx],
                                      "dist": [int(x / 2) for x in editdist]}}   # core.py:514
            else:
                if editdist[1] >= self.editdist:                  # core.py:518
                    neighbor_dict[queryseq] = {
                        "target": queryseq,
                        "neighbors": {"seqs": [self.db[x] for x in hitseqidx],
                                      "dist": [int(x) for x in editdist]}}
        self.neighbors = neighbor_dict

    def nearest_raw(self, seqs, threads=0):
        """core.py:603-606: knnQueryBatch(k=2) then i[1][0]."""
        _, dists, _ = self.knn_raw(seqs, 2, threads)
        return dists[:, 0]

    def control_select(self, seqs, n, threads=0):
        """core.py:603-617 for one batch of candidate control sequences (the RNG draw is the caller's).

        Returns (sort_seq, sort_dist, minimum_hmdist)."""
        distlist = self.nearest_raw(seqs, threads).tolist()
        zipped = list(zip(seqs, distlist))
        dist_seqs = sorted(zipped, reverse=True, key=lambda x: x[1])   # stable, like the reference
        sort_seq = [item[0] for item in dist_seqs][0:n]
        if self.dtype == "hamming":
            sort_dist = [item[1] / 2 for item in dist_seqs][0:n]
        else:
            sort_dist = [item[1] for item in dist_seqs][0:n]
        return sort_seq, sort_dist, int(min(sort_dist))

    @staticmethod
    def control_stats(sort_dist):
        return min(sort_dist), statistics.median(sort_dist)


def format_neighbor_columns(neighbors, guides):
    """Annotation._format_guide_table's two closures restated (reference core.py:913-921, applied at :930-931):
    get_off_target_seqs -> ";".join(seqs), get_off_target_score -> ";".join(str(i) for i in dist).
    neighbors: the dict get_neighbors builds.  -> (similar guides list, similar guide distances list)"""
    sims, dsts = [], []
    for seq in guides:
        slist = neighbors[seq]["neighbors"]["seqs"]
        sims.append(";".join(slist))
        dlist = neighbors[seq]["neighbors"]["dist"]
        dsts.append(";".join([str(i) for i in dlist]))
    return sims, dsts


def cfd_score_rows(guides, similar_guides, mm):
    """core.py:1129-1148 on plain lists: returns (list of list[str], list of max)."""
    out, mx = [], []
    for g, sim in zip(guides, similar_guides):
        vals = [calc_cfd(g, item, mm) for item in sim.split(';')]
        out.append([str(v) for v in vals])
        mx.append(sorted(float(x) for x in out[-1])[-1])
    return out, mx


# ------------------------------------------------------------------------------------------------
# N2 filters (test checker for gm_site_flags / gm_seed_duplicated)
# ------------------------------------------------------------------------------------------------
_IUPAC_VALUES = {
    "A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "R": "AG", "W": "AT", "S": "CG", "Y": "CT",
    "K": "GT", "V": "ACG", "H": "ACT", "D": "AGT", "B": "CGT", "X": "GATC", "N": "GATC",
}
_IUPAC_COMPLEMENT = str.maketrans("ACGTMRWSYKVHDBXN", "TGCAKYWSRMBDHVXN")


def restriction_flags(targets, restriction_enzyme_list):
    """check_restriction_enzymes restated (reference core.py:354-377): every enzyme and its reverse
    complement is expanded to its concrete sequences (core.py:1093-1124), the alternation of all of them
    is searched in each target.  -> list[bool]"""
    import itertools
    import re
    alternatives = []
    for record in set(restriction_enzyme_list):
        record = record.upper()
        for letter in record:
            assert letter in _IUPAC_VALUES
        for site in (record, record.translate(_IUPAC_COMPLEMENT)[::-1]):
            alternatives += ["".join(p) for p in itertools.product(*[_IUPAC_VALUES[b] for b in site])]
    if not alternatives:
        return [False] * len(targets)
    rx = re.compile("|".join(alternatives))
    return [rx.search(t) is not None for t in targets]


def seed_duplicated(targets, lsr, five_prime):
    """find_unique_near_pam restated (reference core.py:388-416): seed = whole target (lsr == 0), the
    first lsr bases (5prime: PAM in front) or s[(len(s) - lsr):] (3prime, with Python's negative-index
    behaviour when lsr > len(s)); flag = the seed was seen at an earlier row (pandas duplicated()).
    -> (seeds list[str], flags list[bool])"""
    seeds, flags, seen = [], [], set()
    for s in targets:
        if lsr == 0:
            seed = s
        elif five_prime:
            seed = s[0:lsr]
        else:
            seed = s[(len(s) - lsr):]
        seeds.append(seed)
        flags.append(seed in seen)
        seen.add(seed)
    return seeds, flags


# ------------------------------------------------------------------------------------------------
# Reference-ALGORITHM CPU baseline: HNSW with the shipped parameters (hnsw_baseline.cpp; not the parity oracle)
# ------------------------------------------------------------------------------------------------
_HNSW = None


def _hnsw_lib():
    global _HNSW
    if _HNSW is None:
        path = os.path.join(_HERE, "libhnsw_baseline.so")
        if not os.path.exists(path):
            build()
        H = ctypes.CDLL(path)
        H.hb_build.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_int, ctypes.c_uint64]
        H.hb_build.restype = ctypes.c_void_p
        H.hb_query.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p]
        H.hb_query.restype = ctypes.c_int
        H.hb_free.argtypes = [ctypes.c_void_p]
        H.hb_free.restype = None
        H.hb_max_threads.restype = ctypes.c_int
        _HNSW = H
    return _HNSW


class HnswBaseline:
    """HNSW index over 2-bit packed targets with the reference's nmslib parameters (config_default.yaml:3-6):
    M 16, efConstruction 10, post 1; query beam max(ef 9, k).  packed: uint64, base p in bits [2p, 2p+1]."""

    def __init__(self, packed, L, metric="hamming", M=16, efc=10, post=1, threads=0, seed=1):
        self.packed = np.ascontiguousarray(packed, dtype=np.uint64)
        self._h = _hnsw_lib().hb_build(self.packed.ctypes.data, self.packed.shape[0], L,
                                       0 if metric == "hamming" else 1, M, efc, post, threads, seed)
        if not self._h:
            raise ValueError("hnsw baseline: bad arguments")

    def query(self, q_packed, k, ef=9, threads=0):
        """-> (ids [nq,k] int32, dists [nq,k] int32 in TRUE units), -1 padded, rows ordered by (distance, id)."""
        q = np.ascontiguousarray(q_packed, dtype=np.uint64)
        ids = np.empty((q.shape[0], k), np.int32)
        dists = np.empty((q.shape[0], k), np.int32)
        rc = _hnsw_lib().hb_query(self._h, q.ctypes.data, q.shape[0], k, ef, threads, ids.ctypes.data, dists.ctypes.data)
        if rc != 0:
            raise ValueError("hnsw baseline: bad query")
        return ids, dists

    def close(self):
        if self._h:
            _hnsw_lib().hb_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def top2_distance_recall(result_dists, truth_dists):
    """The reference's own quality measure (tests/GridOptimization.ipynb:175-190, `recall`): the fraction of queries
    whose first two k-NN DISTANCES equal the exact search's first two ("we really care that the algorithm estimates
    the correct distance not the exact value of the neighbor")."""
    r = np.asarray(result_dists)[:, 0:2]
    t = np.asarray(truth_dists)[:, 0:2]
    assert r.shape == t.shape
    return float((r == t).all(axis=1).mean()) if r.shape[0] else 1.0
