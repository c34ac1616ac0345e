"""A real genome of BASELINE size through the whole mirrored flow: P. aeruginosa PAO1 (NC_002516, 6 264 404 bp, GC 66.6 %;
SURVEY.md section 8 "offline real data"), the genome the reference ships as data.

tests/golden/pao1_summary.json holds what the REFERENCE's own find_targets / check_restriction_enzymes /
find_unique_near_pam produced on it in the build container (tests/golden/make_pao1.py): row counts, SHA-256 of every
column, first and last rows.  The sequence travels as tests/golden/pao1_genome.2bit.gz.

  * CPU: the host PAM scanner reproduces the reference's frame (1 171 798 rows) column by column.
  * GPU: the GPU PAM scan and the two GPU filters do; then create_index / get_neighbors on the 1 159 446 unique targets
    (repeat families, GC-skewed block buckets -- what the synthetic stand-ins do not have): the pigeonhole scan against
    the plain scan on the whole result, against the CPU oracle on a row sample, and the reference-level contract
    (retained keys, neighbour sequences) on the TargetProcessor.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_json
from guidemaker_amd import _lib, core, workloads


class Rec:
    def __init__(self, seq, rid):
        self.seq, self.id = seq, rid


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _sha_strings(values):
    return hashlib.sha256(("\n".join(values) + "\n").encode()).hexdigest()


def _genome():
    s = load_json("pao1_summary.json")
    seq = workloads.read_2bit(os.path.join(GOLDEN, "pao1_genome.2bit.gz"), s["record"]["length"])
    assert hashlib.sha256(seq.encode()).hexdigest() == s["record"]["sha256_ascii"]
    return s, seq


def _check_frame(s, df):
    assert len(df) == s["n_targets"]
    assert df["target"].nunique() == s["n_unique"]
    h = s["sha256"]
    assert _sha_strings(df["target"].tolist()) == h["target"]
    assert _sha_strings([str(x) for x in df["exact_pam"].tolist()]) == h["exact_pam"]
    assert _sha_strings(df["target_seq30"].tolist()) == h["target_seq30"]
    assert _sha(df["start"].to_numpy().astype(np.uint32)) == h["start_u32"]
    assert _sha(df["stop"].to_numpy().astype(np.uint32)) == h["stop_u32"]
    assert _sha(df["strand"].to_numpy().astype(np.uint8)) == h["strand_u8"]
    for want, r in zip(s["first_rows"], df.head(5).itertuples()):
        assert [r.target, str(r.exact_pam), int(r.start), int(r.stop), bool(r.strand), r.target_seq30] == want[:6]


def test_pao1_find_targets_host_scanner():
    """reference core.py:83-292 on a 6.26 Mb record, host scanner (no GPU in the CPU suite)"""
    s, seq = _genome()
    p = s["params"]
    res = core.scan_targets(seq, p["pam"], p["orientation"], p["target_len"])
    assert res["start"].shape[0] == s["n_targets"]
    assert _sha(res["start"].astype(np.uint32)) == s["sha256"]["start_u32"]
    assert _sha(res["stop"].astype(np.uint32)) == s["sha256"]["stop_u32"]
    assert _sha(res["strand"].astype(np.uint8)) == s["sha256"]["strand_u8"]
    assert _sha_strings(core.codes_to_strings(res["codes"])) == s["sha256"]["target"]
    pool = workloads.make_pool("pao1")
    assert pool["packed_all"].shape[0] == s["n_targets"] and pool["packed_db"].shape[0] == s["n_unique"]


@pytest.mark.gpu
def test_pao1_flow_gpu(oracle):
    s, seq = _genome()
    p = s["params"]
    pt = core.PamTarget(p["pam"], p["orientation"], "hamming")
    df = pt.find_targets([Rec(seq, s["record"]["id"])], p["target_len"])          # GPU PAM scan (gm_pam_*)
    _check_frame(s, df)
    tp = core.TargetProcessor(df, lsr=p["lsr"], editdist=2, knum=10)
    tp.check_restriction_enzymes(p["restriction"])                                # gm_site_flags
    tp.find_unique_near_pam()                                                     # gm_seed_duplicated
    t = tp.targets
    assert _sha(t["isseedduplicated"].to_numpy().astype(np.uint8)) == s["sha256"]["isseedduplicated_u8"]
    assert _sha(t["hasrestrictionsite"].to_numpy().astype(np.uint8)) == s["sha256"]["hasrestrictionsite_u8"]
    assert int((t["isseedduplicated"] == False).sum()) == s["n_seed_unique"]      # noqa: E712
    assert int((t["hasrestrictionsite"] == True).sum()) == s["n_restriction"]     # noqa: E712
    for want, r in zip(s["last_rows"], t.tail(5).itertuples()):
        assert [bool(r.isseedduplicated), bool(r.hasrestrictionsite)] == want[6:8]

    import guidemaker_amd
    tp.create_index(guidemaker_amd.CONFIG_PATH)
    tp.get_neighbors(guidemaker_amd.CONFIG_PATH)
    idx = tp.nmslib_index
    assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")                            # the production kernel on real data
    ids, dists, counts = tp.last_knn
    n_q = ids.shape[0]
    assert n_q == s["n_query"]                                                    # core.py:495 on the reference's flags
    assert (counts == 10).all() and (dists[:, 0] == 0).all() and (np.diff(dists, axis=1) >= 0).all()
    # the whole result against the plain two-phase scan
    qmask = ((t["isseedduplicated"] == False) | (t["hasrestrictionsite"] == False)).to_numpy()   # noqa: E712
    q = core.packed_side_car(df)[qmask]
    idx.set_variant(8)
    i8, d8, c8 = idx.knn(q, 10)
    idx.set_variant(0)
    assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
    # the layout planned from the independent-base model alone (no scanned sample): another layout, the same answer
    idx.set_param("pg_sample", 0)
    i0, d0, c0 = idx.knn(q, 10)
    assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
    rescanned_model = idx.last_evaluated()[1]
    idx.set_param("pg_sample", 1)
    assert (i0 == ids).all() and (d0 == dists).all() and (c0 == counts).all()
    i1, d1, c1 = idx.knn(q, 10)
    assert (i1 == ids).all() and (d1 == dists).all() and (c1 == counts).all()
    # what the sample is for: on this genome the model's radius leaves an eighth of the rows to the plain re-scans; with
    # the measured spread of the k-th neighbour's distance the first K1P level runs the radius the majority needs, the
    # rows it leaves go through deeper K1P levels (wider radius, layouts of their own) and hardly any reach the re-scans
    levels = idx.cascade_info()
    assert len(levels) >= 2 and levels[0]["n_q"] == n_q and levels[1]["n_q"] == levels[0]["left"], levels
    assert all(b["radius"] > a["radius"] for a, b in zip(levels, levels[1:])), levels
    assert levels[-1]["left"] < rescanned_model // 4, (levels, rescanned_model)
    # a row sample against the CPU oracle over the full pool (true units)
    packed_db = core.packed_side_car(df)[np.sort(np.unique(core.packed_side_car(df), return_index=True)[1])]
    assert packed_db.shape[0] == s["n_unique"]
    rows = np.sort(np.random.default_rng(66).choice(n_q, size=2048, replace=False))
    L = p["target_len"]
    asc = np.frombuffer(b"ACGT", np.uint8)
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    db_b = np.ascontiguousarray(asc[((packed_db[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    q_b = np.ascontiguousarray(asc[((q[rows][:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    oi, od, oc = oracle.knn(db_b, q_b, L, "hamming", 10)
    assert (oc == counts[rows]).all() and (od // 2 == dists[rows]).all() and (oi == ids[rows]).all()
    # ... and EVERY row (1.36 10^12 pairs) against the fast exact CPU scan (gm_oracle_tuned.c, pinned to the literal oracle
    # by tests/test_oracle.py): the real genome is where the cascade, the repeats and the id ties are
    for lo in range(0, n_q, 150_000):
        q_s = np.ascontiguousarray(asc[((q[lo:lo + 150_000][:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
        ti, td, tc = oracle.knn_tuned(db_b, q_s, L, 10)
        hi = lo + q_s.shape[0]
        assert (tc == counts[lo:hi]).all() and (td // 2 == dists[lo:hi]).all() and (ti == ids[lo:hi]).all(), lo
    # reference-level contract: retained = candidates whose 2nd hit is at least editdist away (core.py:512), and the
    # neighbour sequences are the pool's
    keep = np.flatnonzero(dists[:, 1] >= 2)
    keys = set(np.asarray(t["target"].to_numpy()[qmask], dtype=object)[keep].tolist())
    assert set(tp.neighbors.keys()) == keys
    some = next(iter(tp.neighbors))
    e = tp.neighbors[some]
    assert e["target"] == some and e["neighbors"]["seqs"][0] == some and e["neighbors"]["dist"][0] == 0


@pytest.mark.gpu
def test_pao1_levenshtein_gpu(oracle):
    """The same genome under BASELINE configs[2]'s parameters (TTTV 5prime, L = 23, Levenshtein, k = 5): the bit-sliced
    scan with its narrow-band first pass against the single pass, against the per-pair Myers kernel, and against the
    oracle's O(nm) DP on a row sample -- real targets cluster (repeats, codon bias), the synthetic pools do not."""
    pool = workloads.make_pool("pao1_leven")
    L, k = pool["guidelen"], pool["k"]
    q, db = pool["packed_all"], pool["packed_db"]
    assert q.shape[0] == 43123 and db.shape[0] == 42455
    idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
    try:
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0].startswith("k_leven_knn_bs")
        assert (counts == k).all() and (dists[:, 0] == 0).all() and (np.diff(dists, axis=1) >= 0).all()
        idx.set_param("lv_two_pass", 0)
        i1, d1, c1 = idx.knn(q, k)
        idx.set_param("lv_two_pass", 1)
        assert (i1 == ids).all() and (d1 == dists).all() and (c1 == counts).all()
        idx.set_variant(2)                                                        # per-pair Myers kernel
        i2, d2, c2 = idx.knn(q, k)
        idx.set_variant(0)
        assert (i2 == ids).all() and (d2 == dists).all() and (c2 == counts).all()
    finally:
        idx.close()
    rows = np.sort(np.random.default_rng(23).choice(q.shape[0], size=2048, replace=False))
    asc = np.frombuffer(b"ACGT", np.uint8)
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    db_b = np.ascontiguousarray(asc[((db[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    q_b = np.ascontiguousarray(asc[((q[rows][:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    oi, od, oc = oracle.knn(db_b, q_b, L, "leven", k)
    assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    # ... and every row against the fast exact CPU scan (Myers' bit-vector recurrence, pinned to the dynamic programme)
    q_all = np.ascontiguousarray(asc[((q[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    oi, od, oc = oracle.knn_leven_tuned(db_b, q_all, L, k)
    assert (oc == counts).all() and (od == dists).all() and (oi == ids).all()


@pytest.mark.gpu
def test_pao1_l23_levenshtein_gpu(oracle):
    """The genome's NGG targets cut at L = 23 under Levenshtein, k = 5 (bench.py's `pao1_l23`: 1 160 561 x 1 160 561, two
    displaced-block partition levels + the plain pass): every 32nd candidate row against the fast exact CPU scan over
    the WHOLE pool (4.2 10^10 pairs), ids, distances and counts bit for bit."""
    pool = workloads.make_pool("pao1_l23")
    L, k = pool["guidelen"], pool["k"]
    q, db = pool["packed_all"], pool["packed_db"]
    idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
    try:
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0] == "k_leven_part_bs"
    finally:
        idx.close()
    assert (counts == k).all() and (dists[:, 0] == 0).all() and (np.diff(dists, axis=1) >= 0).all()
    asc = np.frombuffer(b"ACGT", np.uint8)
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    db_b = np.ascontiguousarray(asc[((db[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    rows = np.arange(0, q.shape[0], 32)
    q_b = np.ascontiguousarray(asc[((q[rows][:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    oi, od, oc = oracle.knn_leven_tuned(db_b, q_b, L, k)
    assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
