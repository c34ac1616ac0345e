"""GPU parity tests: the HIP path (through the C-ABI in libgmhip.so) against the CPU oracle, the
reference's known answers and the golden fixtures.  Bar: bit-exact for distances / ids / counts,
|delta| <= 1e-6 for CFD (observed: bit-exact)."""
import random

import numpy as np
import pytest

from conftest import load_json, load_carsonella
from helpers import Rec, carsonella, random_pool, targets_frame

import guidemaker_amd
from guidemaker_amd import _lib, core

pytestmark = pytest.mark.gpu
CONFIG = guidemaker_amd.CONFIG_PATH


def _oracle_true_units(orc, db, q, L, metric, k):
    ids, dists, counts = orc.knn(db, q, L, metric, k)
    if metric == "hamming":
        dists = np.where(dists >= 0, dists // 2, -1)     # oracle reports nmslib's one-hot units
    return ids, dists, counts


def _gpu_knn(db, q, L, metric, k):
    idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING if metric == "hamming" else _lib.METRIC_LEVEN)
    try:
        return idx.knn(_lib.pack(q, L), k)
    finally:
        idx.close()


# ------------------------------------------------------------------ reference known answers
def test_get_neighbors_three_row():
    # reference tests/test_core.py:116-126
    kat = load_json("knn_kat.json")["three_row"]
    tl = core.TargetProcessor(targets=targets_frame(kat["targets"]), lsr=10, editdist=2, knum=2)
    tl.check_restriction_enzymes(["NRAGCA"])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    tl.get_neighbors(configpath=CONFIG)
    assert tl.neighbors["AAATGGTACGTTATGTGTTA"]["neighbors"]["dist"][1] == 12
    assert tl.neighbors["AAATGGTACGTTATGTGTTA"]["neighbors"]["seqs"] == [
        "AAATGGTACGTTATGTGTTA", "AACAGTAAAATGGTTTAATG"]


def test_export_bed_knum_larger_than_index():
    # reference tests/test_core.py:129-139: knum=10 on a 2-point index
    kat = load_json("knn_kat.json")["three_row"]
    tl = core.TargetProcessor(targets=targets_frame(kat["targets"]), lsr=10, editdist=2, knum=10)
    tl.check_restriction_enzymes(["NRAGCA"])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    tl.get_neighbors(configpath=CONFIG)
    assert tl.neighbors["AAATGGTACGTTATGTGTTA"]["neighbors"]["dist"] == [0, 12]
    assert tl.export_bed().shape == (2, 5)


def test_levin_dist():
    # reference tests/test_core.py:319-347
    kat = load_json("knn_kat.json")["levin"]
    distseq = [Rec(kat["seq"], "distseq")]
    pd_levin = core.PamTarget("NGG", "3prime", "levin").find_targets(distseq, 20)
    pd_hamming = core.PamTarget("NGG", "3prime", "hamming").find_targets(distseq, 20)
    tp_levin = core.TargetProcessor(targets=pd_levin, lsr=0, editdist=1, knum=3)
    tp_hamming = core.TargetProcessor(targets=pd_hamming, lsr=0, editdist=1, knum=3)
    for tp in (tp_levin, tp_hamming):
        tp.find_unique_near_pam()
        tp.check_restriction_enzymes()
        tp.create_index(configpath=CONFIG)
        tp.get_neighbors(configpath=CONFIG)
    assert tp_levin.neighbors['CTAGTCACTAGCTGACAGCA']['neighbors']['dist'] == [0, 1, 2]
    assert tp_hamming.neighbors['CTAGTCACTAGCTGACAGCA']['neighbors']['dist'] == [0, 1, 16]


def test_single_point_index_raises_like_reference():
    # reference core.py:512 editdist[1] -> IndexError when nmslib returns < 2 hits
    tl = core.TargetProcessor(targets=targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0, editdist=1, knum=2)
    tl.find_unique_near_pam()
    tl.check_restriction_enzymes()
    tl.create_index(configpath=CONFIG)
    with pytest.raises(IndexError):
        tl.get_neighbors(configpath=CONFIG)


# ------------------------------------------------------------------ oracle parity, real data
@pytest.mark.parametrize("tag,orient", [("NGG5p", "5prime"), ("NGG3p", "3prime")])
@pytest.mark.parametrize("dtype", ["hamming", "leven"])
def test_carsonella_neighbors_match_oracle(oracle, tag, orient, dtype):
    cols = load_carsonella(tag)
    df = core.PamTarget("NGG", orient, dtype).find_targets(carsonella(), 20)
    tl = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=10)
    tl.check_restriction_enzymes(["NRAGCA"])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    tl.get_neighbors(configpath=CONFIG)

    otp = oracle.OracleTargetProcessor(cols["target"], dtype, 2, 10, cols["isseedduplicated"],
                                       cols["hasrestrictionsite"])
    otp.create_index()
    otp.get_neighbors()
    assert tl._db_targets.tolist() == otp.db
    assert set(tl.neighbors.keys()) == set(otp.neighbors.keys())           # P2 retained set
    for key, val in otp.neighbors.items():
        assert tl.neighbors[key]["neighbors"]["dist"] == val["neighbors"]["dist"]   # P1
        assert tl.neighbors[key]["neighbors"]["seqs"] == val["neighbors"]["seqs"]   # P3
    if dtype == "hamming":
        expect = {"NGG5p": 3808, "NGG3p": 3810}[tag]
        assert len(tl.neighbors) == expect
        ids, dists, _ = tl.last_knn
        want_first = {"NGG5p": [0, 6, 6, 7, 7, 7, 7, 7, 7, 7], "NGG3p": [0, 6, 6, 6, 6, 6, 6, 7, 7, 7]}[tag]
        assert dists[0].tolist() == want_first


# ------------------------------------------------------------------ oracle parity, synthetic pools
@pytest.mark.parametrize("L", [10, 17, 20, 23, 27])
@pytest.mark.parametrize("k", [2, 10, 20])
def test_hamming_knn_matches_oracle(oracle, L, k):
    db = list(dict.fromkeys(random_pool(3000 + 37 * L, L, seed=L * 100 + k)))
    q = db[::3] + random_pool(200, L, seed=5)
    ids, dists, counts = _gpu_knn(db, q, L, "hamming", k)
    oi, od, oc = _oracle_true_units(oracle, db, q, L, "hamming", k)
    assert (counts == oc).all()
    assert (dists == od).all()
    assert (ids == oi).all()


@pytest.mark.parametrize("L", [10, 20, 23, 27])
@pytest.mark.parametrize("k", [2, 5, 20])
def test_leven_knn_matches_oracle(oracle, L, k):
    db = list(dict.fromkeys(random_pool(1500 + 11 * L, L, seed=L * 7 + k, indel=True)))
    q = db[::4] + random_pool(100, L, seed=6, indel=True)
    ids, dists, counts = _gpu_knn(db, q, L, "leven", k)
    oi, od, oc = _oracle_true_units(oracle, db, q, L, "leven", k)
    assert (counts == oc).all()
    assert (dists == od).all()
    assert (ids == oi).all()


def test_ragged_and_edge_sizes(oracle):
    L = 20
    pool = list(dict.fromkeys(random_pool(1300, L, seed=11)))
    for n_db in (1, 2, 3, 63, 64, 65, 511, 512, 513, 1025):
        for n_q in (1, 63, 257):
            for metric in ("hamming", "leven"):
                db, q = pool[:n_db], pool[-n_q:]
                k = 5
                ids, dists, counts = _gpu_knn(db, q, L, metric, k)
                oi, od, oc = _oracle_true_units(oracle, db, q, L, metric, k)
                assert (counts == oc).all() and (dists == od).all() and (ids == oi).all(), (n_db, n_q, metric)
    # empty query set and empty index
    idx = _lib.Index(_lib.pack(pool[:10], L), L, _lib.METRIC_HAMMING)
    ids, dists, counts = idx.knn(np.empty((0,), np.uint64), 3)
    assert ids.shape == (0, 3) and counts.shape == (0,)
    idx.close()
    idx = _lib.Index(np.empty((0,), np.uint64), L, _lib.METRIC_HAMMING)
    ids, dists, counts = idx.knn(_lib.pack(pool[:5], L), 3)
    assert (ids == -1).all() and (dists == -1).all() and (counts == 0).all()
    assert (idx.nearest(_lib.pack(pool[:5], L)) == -1).all()
    idx.close()


def test_duplicate_db_entries_and_ties(oracle):
    # collisions: many identical / equidistant db points -> canonical (distance, id) order
    L = 20
    base = "ACGTACGTACGTACGTACGT"
    db = [base] * 40 + [base[:-1] + "A"] * 40 + random_pool(500, L, seed=2)
    q = [base, base[:-1] + "C", base[:-2] + "AA"]
    for metric in ("hamming", "leven"):
        ids, dists, counts = _gpu_knn(db, q, L, metric, 20)
        oi, od, oc = _oracle_true_units(oracle, db, q, L, metric, 20)
        assert (ids == oi).all() and (dists == od).all() and (counts == oc).all()
    assert ids[0].tolist() == list(range(20))


@pytest.mark.parametrize("metric", ["hamming", "leven"])
def test_nearest_dist_matches_oracle(oracle, metric):
    L = 20
    db = list(dict.fromkeys(random_pool(2500, L, seed=21, indel=True)))
    q = random_pool(700, L, seed=22) + db[:50]
    idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING if metric == "hamming" else _lib.METRIC_LEVEN)
    d1 = idx.nearest(_lib.pack(q, L))
    idx.close()
    _, od, _ = _oracle_true_units(oracle, db, q, L, metric, 2)
    assert (d1 == od[:, 0]).all()


# ------------------------------------------------------------------ size-independent properties
def test_full_size_properties():
    """E. coli-sized pool (BASELINE config 2 shape): properties that need no oracle."""
    L, k, n = 20, 10, 560_000
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    packed = np.unique(core.pack_codes(codes))
    rng.shuffle(packed)
    n = packed.shape[0]
    idx = _lib.Index(packed, L, _lib.METRIC_HAMMING)
    qsel = rng.choice(n, size=70_000, replace=False)
    ids, dists, counts = idx.knn(packed[qsel], k)
    assert (counts == k).all()
    assert (dists[:, 0] == 0).all() and (ids[:, 0] == qsel).all()          # self is the nearest
    assert (np.diff(dists, axis=1) >= 0).all()                              # sorted
    same = np.diff(dists, axis=1) == 0
    assert (np.diff(ids, axis=1)[same] > 0).all()                           # ties by id
    # reported distances are the true distances of the reported ids
    x = packed[qsel][:, None] ^ packed[ids]
    mism = ((x | (x >> np.uint64(1))) & np.uint64(0x5555555555555555))
    pop = np.zeros(mism.shape, np.int32)
    for b in range(0, 64, 2):
        pop += ((mism >> np.uint64(b)) & np.uint64(1)).astype(np.int32)
    assert (pop == dists).all()
    # nearest-distance kernel agrees with the k-NN kernel: query != db point -> compare on a probe set
    probe = core.pack_codes(rng.integers(0, 4, size=(5000, L), dtype=np.uint8))
    d1 = idx.nearest(probe)
    pid, pd_, _ = idx.knn(probe, 2)
    assert (d1 == pd_[:, 0]).all()
    # idempotence: same call, same answer
    ids2, dists2, _ = idx.knn(packed[qsel], k)
    assert (ids2 == ids).all() and (dists2 == dists).all()
    # shard invariance (the multi-GPU decomposition): query shards give the same rows
    for nshard in (2, 8):
        parts = [idx.knn(part, k) for part in np.array_split(packed[qsel], nshard)]
        assert (np.concatenate([p[0] for p in parts]) == ids).all()
        assert (np.concatenate([p[1] for p in parts]) == dists).all()
    idx.close()


def test_full_size_properties_leven(oracle):
    """B. subtilis-sized Levenshtein pool (BASELINE config 3 shape: L=23, k=5): size-independent
    properties + a sampled comparison with the oracle over the FULL pool."""
    L, k, n = 23, 5, 136_000
    rng = np.random.default_rng(7)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    # planted near-duplicates: substitutions and single-base shifts (indel pairs)
    src = rng.integers(0, n, size=4000)
    near = codes[src].copy()
    near[np.arange(4000), rng.integers(0, L, size=4000)] = rng.integers(0, 4, size=4000)
    shifted = np.concatenate([codes[src[:2000], 1:], rng.integers(0, 4, size=(2000, 1), dtype=np.uint8)], axis=1)
    codes = np.concatenate([codes, near, shifted])
    packed = np.unique(core.pack_codes(codes))
    rng.shuffle(packed)
    n = packed.shape[0]
    idx = _lib.Index(packed, L, _lib.METRIC_LEVEN)
    hidx = _lib.Index(packed, L, _lib.METRIC_HAMMING)
    ids, dists, counts = idx.knn(packed, k)
    assert (counts == k).all()
    assert (dists[:, 0] == 0).all() and (ids[:, 0] == np.arange(n)).all()   # self is the nearest
    assert (np.diff(dists, axis=1) >= 0).all()
    same = np.diff(dists, axis=1) == 0
    assert (np.diff(ids, axis=1)[same] > 0).all()                            # ties by id
    # lev <= hamming, column by column of the two k-NN tables
    _, hd, _ = hidx.knn(packed, k)
    assert (dists <= hd).all()
    assert (dists[:, 1] <= 1).sum() >= 3000                                  # the planted pairs are found
    # nearest-distance kernel == first column of a 2-NN on probes that are not pool members
    probe = core.pack_codes(rng.integers(0, 4, size=(3000, L), dtype=np.uint8))
    assert (idx.nearest(probe) == idx.knn(probe, 2)[1][:, 0]).all()
    # bit-sliced DP (default) == per-pair Myers kernel on a query sample
    sel = rng.choice(n, size=20_000, replace=False)
    idx.set_variant(2)
    i2, d2, c2 = idx.knn(packed[sel], k)
    idx.set_variant(0)
    assert (i2 == ids[sel]).all() and (d2 == dists[sel]).all() and (c2 == counts[sel]).all()
    # shard invariance
    parts = [idx.knn(part, k) for part in np.array_split(packed[sel], 3)]
    assert (np.concatenate([p[0] for p in parts]) == ids[sel]).all()
    # oracle over the full pool for a few candidates
    sample = rng.choice(n, size=64, replace=False)
    db_str = _lib.unpack(packed, L)
    oi, od, oc = _oracle_true_units(oracle, db_str, [db_str[i] for i in sample], L, "leven", k)
    assert (oi == ids[sample]).all() and (od == dists[sample]).all() and (oc == counts[sample]).all()
    idx.close()
    hidx.close()


# ------------------------------------------------------------------ controls
def test_get_control_seqs(oracle):
    # reference tests/test_core.py:144-155 (shape), plus exact agreement with the oracle on the draw
    df = core.PamTarget("NGG", "5prime", "hamming").find_targets(carsonella(), 20)
    tl = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=10)
    tl.check_restriction_enzymes(['NRAGCA'])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    np.random.seed(42)
    data = tl.get_control_seqs(carsonella(), length=20, n=100, num_threads=2, configpath=CONFIG)
    assert data[2].shape == (100, 3)
    assert list(data[2].columns) == ["name", "Sequences", "Hamming distance"]
    assert data[0] >= 7 and tl.ncontrolsearched in (1000, 10000, 100000)
    assert abs(tl.gc_percent - 16.56) < 0.01
    # replay the same RNG stream through the oracle's restatement of core.py:586-617
    otp = oracle.OracleTargetProcessor(df["target"].tolist(), "hamming", 2, 10)
    otp.create_index()
    np.random.seed(42)
    gc = tl.gc_percent / 100
    minimum, sm = 0, 0
    while minimum < 7:
        mult = [10, 100, 1000, 10000][sm]
        seqs = core.TargetProcessor._draw_control_seqs(100 * mult, 20, gc)
        sort_seq, sort_dist, minimum = otp.control_select(seqs, 100)
        sm += 1
    assert data[2]["Sequences"].tolist() == sort_seq
    assert data[2]["Hamming distance"].tolist() == sort_dist
    assert (data[0], data[1]) == otp.control_stats(sort_dist)


def test_get_control_seqs_levenshtein(oracle, tmp_path):
    """reference core.py:586-617 on a Levenshtein index (dtype != "hamming", core.py:458-466): the controls are queried as
    raw sequences (:601-602), their nearest EDIT distances are reported as integers (no halving, :615), and the column is
    still called "Hamming distance" (:624).  Same RNG stream through the oracle's restatement."""
    import yaml
    df = core.PamTarget("NGG", "5prime", "leven").find_targets(carsonella(), 20)
    tl = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=5)
    tl.check_restriction_enzymes(['NRAGCA'])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    with open(CONFIG) as fh:
        cfg = yaml.safe_load(fh)
    cfg["CONTROL"]["MINIMUM_HMDIST"] = 7      # (edit distances to a 3 814-target pool: the twenty best of 200 draws reach 6, of 2 000 draws 7)
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    np.random.seed(11)
    mn, med, table = tl.get_control_seqs(carsonella(), length=20, n=20, num_threads=2, configpath=str(cfg_path))
    assert table.shape == (20, 3) and list(table.columns) == ["name", "Sequences", "Hamming distance"]
    assert all(isinstance(d, (int, np.integer)) for d in table["Hamming distance"].tolist())      # integers, not x / 2 floats
    otp = oracle.OracleTargetProcessor(df["target"].tolist(), "leven", 2, 5)
    otp.create_index()
    np.random.seed(11)
    gc = tl.gc_percent / 100
    minimum, sm = 0, 0
    while minimum < 7:
        mult = [10, 100, 1000, 10000][sm]
        seqs = core.TargetProcessor._draw_control_seqs(20 * mult, 20, gc)
        sort_seq, sort_dist, minimum = otp.control_select(seqs, 20)
        sm += 1
    assert sm <= 2      # (keeps the oracle's share of this test to a few seconds)
    assert table["Sequences"].tolist() == sort_seq
    assert table["Hamming distance"].tolist() == sort_dist
    assert (mn, med) == otp.control_stats(sort_dist)
    assert tl.ncontrolsearched == 20 * [10, 100, 1000, 10000][sm - 1]


# ------------------------------------------------------------------ CFD
def test_cfd_golden(cfd_table):
    cases = load_json("cfd_kat.json")["cases"]
    got = core.calc_cfd_pairs([c["guide"] for c in cases], [c["off"] for c in cases])
    want = np.array([float.fromhex(c["cfd_hex"]) for c in cases])
    assert np.abs(got - want).max() <= 1e-6          # north-star tolerance
    assert (got == want).all()                       # and in fact bit-exact
    # reference tests/test_core.py:265-267
    assert abs(core.calc_cfd("GCATGCACAGCTAGCATGCATGCAGCT", "GCATGCACAGCTAGCATGCATGCAGCG") - 0.176470588) < 1e-4


def test_cfd_score_frame(oracle, cfd_table):
    import pandas as pd
    rng = random.Random(9)
    guides = random_pool(300, 20, seed=31)
    sims = [";".join(random_pool(10, 20, seed=100 + i)[:rng.randint(1, 10)]) for i in range(300)]
    df = pd.DataFrame({"Guide sequence": guides, "Similar guides": sims})
    out = core.cfd_score(df)
    want_lists, want_max = oracle.cfd_score_rows(guides, sims, cfd_table)
    assert out["CFD Similar Guides"].tolist() == want_lists
    assert out["Max CFD"].tolist() == want_max
    assert out["Max CFD"][0] > 0                      # reference tests/test_core.py:315


# ------------------------------------------------------------------ every kernel family / dispatch branch
@pytest.mark.parametrize("L", [4, 7, 9])
def test_short_guides_use_popcount_kernel(oracle, L):
    # L < 10 is outside the bit-sliced kernels' range: popcount kernel K1 / plane-based Levenshtein
    db = list(dict.fromkeys(random_pool(900, L, seed=40 + L)))
    q = db[::2] + random_pool(64, L, seed=41)
    for metric in ("hamming", "leven"):
        ids, dists, counts = _gpu_knn(db, q, L, metric, 6)
        oi, od, oc = _oracle_true_units(oracle, db, q, L, metric, 6)
        assert (counts == oc).all() and (dists == od).all() and (ids == oi).all(), (L, metric)


@pytest.mark.parametrize("k", [1, 3, 5, 6, 11, 21, 33, 64])
def test_list_capacity_boundaries(oracle, k):
    # register lists hold k <= 5 / 10 / 20; beyond 20 the lists live in LDS (Hamming) or the
    # plane-based kernel is used (Levenshtein)
    L = 20
    db = list(dict.fromkeys(random_pool(1500, L, seed=77, indel=True)))
    q = db[:300] + random_pool(50, L, seed=78)
    for metric in ("hamming", "leven"):
        ids, dists, counts = _gpu_knn(db, q, L, metric, k)
        oi, od, oc = _oracle_true_units(oracle, db, q, L, metric, k)
        assert (counts == oc).all() and (dists == od).all() and (ids == oi).all(), (k, metric)


def test_kernel_variants_agree(oracle):
    # A/B knobs must not change results: K1 popcount (1), K1L with its lists in LDS (7), plain K1L (8)
    L = 20
    db = list(dict.fromkeys(random_pool(4000, L, seed=90)))
    q = db[::3]
    idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING)
    ref = idx.knn(_lib.pack(q, L), 10)
    for v in (1, 7, 8):
        idx.set_variant(v)
        got = idx.knn(_lib.pack(q, L), 10)
        assert all((a == b).all() for a, b in zip(ref, got)), v
        assert (idx.nearest(_lib.pack(q, L)) == ref[1][:, 0]).all(), v
    idx.close()


@pytest.mark.parametrize("L,variant", [
    (12, 104), (16, 108), (20, 104), (20, 107), (23, 105), (27, 108),      # plain: B blocks, radius B - 1
    (12, 204), (16, 208), (20, 204), (20, 206), (23, 205), (27, 208),      # one expansion block, even lengths
    (12, 304), (16, 306), (20, 306), (23, 305), (27, 308),                 # ... one base longer
    (16, 406), (20, 405), (23, 405), (20, 504), (27, 507),                 # two / three longer expansion blocks
    (20, 605), (20, 704), (24, 706),                                       # two / three, even lengths
    (16, 803), (20, 803), (20, 904), (23, 803), (27, 904), (24, 903)])     # a block with the two-substitution neighbourhood
@pytest.mark.parametrize("hit", [0, 2])      # hit buffer in HBM / register lists + mailbox (gm_set_param "pg_hit")
def test_pigeonhole_layouts_match_oracle(oracle, L, variant, hit):
    """K1P, the pigeonhole-partitioned scan, with a forced layout (gm_set_variant codes of
    run_hamming_pigeon: 100 + B plain, 200..700 + B with one to three expansion blocks).  It must be exact
    for any layout: candidates whose merged lists do not fill are re-scanned by the plain kernel, so a
    small pool with planted near-duplicates exercises the bucket scans, the neighbour-bucket passes, the
    duplicate-dropping merge and the re-scan.  The skewed base composition gives crowded and empty buckets."""
    B = variant
    rng = random.Random(1000 * L + B)
    skew = "AAAAACCGT"
    db = ["".join(rng.choice(skew) for _ in range(L)) for _ in range(5000)]
    for _ in range(2500):      # near-duplicates: 1-4 substitutions
        src = list(db[rng.randrange(len(db))])
        for p in rng.sample(range(L), rng.randint(1, 4)):
            src[p] = rng.choice("ACGT")
        db.append("".join(src))
    db = list(dict.fromkeys(db))
    rng.shuffle(db)
    q = db[::2] + ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(700)] + db[:5]
    idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING)
    idx.set_variant(variant)
    idx.set_param("pg_hit", hit)
    idx.set_param("pg_group", 1 if hit == 0 else 2)      # candidate grouping: one counting sort / two levels (partitions)
    idx.set_param("pg_slots", 1 if (L + variant) % 2 else 2)      # the merge's list of incomplete rows: atomics / workgroup slots
    qp = _lib.pack(q, L)
    for k in (2, 5, 10, 20):
        ids, dists, counts = idx.knn(qp, k)
        assert idx.last_kernel()[0].startswith("k_hamming_scan_pg"), "the forced layout was not used"
        oi, od, oc = _oracle_true_units(oracle, db, q, L, "hamming", k)
        assert (counts == oc).all() and (dists == od).all() and (ids == oi).all(), (L, B, k)
    d1 = idx.nearest(qp)
    assert idx.last_kernel()[0].startswith("k_hamming_scan_pg")
    assert (d1 == _oracle_true_units(oracle, db, q, L, "hamming", 1)[1][:, 0]).all()
    evaluated, rescanned = idx.last_evaluated()
    assert evaluated > 0      # (no saving expected here: tile and chunk padding dominate a 7k pool)
    idx.close()


def test_pigeonhole_repeat_family_overflows_to_rescan(oracle):
    """A family of near-identical targets (a repeat) floods the slot lists of its members: those rows must
    come back exact through the re-scan, the rest through the merge."""
    L, k = 20, 20
    rng = random.Random(77)
    db = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(9000)]
    root = db[0]
    for _ in range(700):
        s = list(root)
        for p in rng.sample(range(L), rng.randint(1, 2)):
            s[p] = rng.choice("ACGT")
        db.append("".join(s))
    db = list(dict.fromkeys(db))
    rng.shuffle(db)
    fam = [t for t in db if sum(a != b for a, b in zip(t, root)) <= 2]
    q = fam[:150] + db[:600]
    idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING)
    idx.set_variant(107)
    ids, dists, counts = idx.knn(_lib.pack(q, L), k)
    assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
    assert idx.last_evaluated()[1] >= 100          # the family rows were re-scanned
    oi, od, oc = _oracle_true_units(oracle, db, q, L, "hamming", k)
    assert (counts == oc).all() and (dists == od).all() and (ids == oi).all()
    idx.close()


def test_pigeonhole_default_path_sampled_oracle(oracle):
    """Pool large enough for the planner to choose K1P on its own: compare a sample with the oracle over
    the full pool, and the whole result with the plain two-phase scan (variant 8)."""
    L, k, n = 20, 10, 300_000
    rng = np.random.default_rng(11)
    packed = np.unique(core.pack_codes(rng.integers(0, 4, size=(n, L), dtype=np.uint8)))
    rng.shuffle(packed)
    idx = _lib.Index(packed, L, _lib.METRIC_HAMMING)
    q = np.concatenate([packed[:60_000], core.pack_codes(rng.integers(0, 4, size=(20_000, L), dtype=np.uint8))])
    ids, dists, counts = idx.knn(q, k)
    assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
    evaluated, rescanned = idx.last_evaluated()
    assert evaluated < 0.5 * q.shape[0] * packed.shape[0]
    idx.set_variant(8)
    i8, d8, c8 = idx.knn(q, k)
    assert idx.last_kernel()[0].startswith("k_hamming_knn_lk")
    assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
    idx.set_variant(0)
    d1 = idx.nearest(q)
    assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg") and (d1 == dists[:, 0]).all()
    sample = rng.choice(q.shape[0], size=48, replace=False)
    db_str = _lib.unpack(packed, L)
    oi, od, oc = _oracle_true_units(oracle, db_str, _lib.unpack(q[sample], L), L, "hamming", k)
    assert (oi == ids[sample]).all() and (od == dists[sample]).all() and (oc == counts[sample]).all()
    idx.close()


def test_many_segments_leven(oracle):
    # few candidates x larger pool -> the Levenshtein scan splits the pool into many segments + merge
    L = 23
    db = list(dict.fromkeys(random_pool(9000, L, seed=5, indel=True)))
    q = db[:70]
    ids, dists, counts = _gpu_knn(db, q, L, "leven", 5)
    oi, od, oc = _oracle_true_units(oracle, db, q, L, "leven", 5)
    assert (counts == oc).all() and (dists == od).all() and (ids == oi).all()


# ------------------------------------------------------------------ N1: PAM scan on the GPU
def test_pam_scan_gpu_matches_reference_fixtures():
    # the GPU scan against the reference's find_targets output (tests/golden, made from the reference)
    kat = load_json("find_targets_kat.json")
    for name, case in kat["cases"].items():
        pt = core.PamTarget(case["pam"], case["orientation"], "hamming")
        df = pt.find_targets([Rec(case["seq"], name)], case["target_len"])      # GPU path (a device is visible)
        got = [[r.target, str(r.exact_pam), int(r.start), int(r.stop), bool(r.strand),
                bool(r.pam_orientation), r.target_seq30] for r in df.itertuples()]
        assert got == case["rows"], name
    for tag, orient in (("NGG5p", "5prime"), ("NGG3p", "3prime")):
        cols = load_carsonella(tag)
        df = core.PamTarget("NGG", orient, "hamming").find_targets(carsonella(), 20)
        assert df["target"].tolist() == cols["target"]
        assert df["start"].tolist() == cols["start"] and df["stop"].tolist() == cols["stop"]
        assert df["strand"].tolist() == cols["strand"]


@pytest.mark.parametrize("pam,orient,L", [("NGG", "3prime", 20), ("TTTV", "5prime", 23), ("NNGRRT", "3prime", 20),
                                          ("NNNNGATT", "5prime", 27), ("NG", "3prime", 10), ("N", "5prime", 12)])
def test_pam_scan_gpu_matches_host_scanner(pam, orient, L):
    # GPU scan vs the numpy scanner (itself pinned to the reference's fixtures in test_host.py) on a
    # genome with N runs, lower-case stretches and record edges
    rng = np.random.default_rng(17)
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=300_000)].copy()
    bases[1000:1040] = ord("N")
    bases[150_000:150_003] = ord("n")
    bases[299_990:] = ord("G")
    bases[:12] = ord("G")
    seq = bases.tobytes().decode()
    host = core.scan_targets(seq, pam, orient, L)
    dev = core.scan_targets_gpu(seq, pam, orient, L)
    assert (dev["packed"] == core.pack_codes(host["codes"])).all()
    for key in ("start", "stop", "strand", "mstart", "mend"):
        assert (dev[key] == host[key]).all(), key
    # edge cases: record shorter than PAM + guide, empty record
    for short in ("", "ACGT", "G" * (L + len(pam) - 1)):
        d = core.scan_targets_gpu(short, pam, orient, L)
        h = core.scan_targets(short, pam, orient, L)
        assert d["packed"].shape[0] == h["codes"].shape[0] == 0


def test_key_packing_device_matches_host():
    import torch
    from guidemaker_amd import distributed
    rng = np.random.default_rng(3)
    n, k = 1000, 10
    ids = rng.integers(0, 1 << 27, size=(n, k)).astype(np.int32)
    dists = rng.integers(0, 28, size=(n, k)).astype(np.int32)
    cnt = rng.integers(0, k + 1, size=n)
    for r in range(n):
        ids[r, cnt[r]:] = -1
        dists[r, cnt[r]:] = -1
    ti, td = torch.from_numpy(ids), torch.from_numpy(dists)
    want = distributed.pack_keys(ti, td)
    got = distributed.pack_keys(ti.cuda(), td.cuda())
    assert (got.cpu() == want).all()
    gi, gd, gc = distributed.unpack_keys(got)
    assert (gi.cpu() == ti).all() and (gd.cpu() == td).all() and (gc.cpu().numpy() == cnt).all()


# ------------------------------------------------------------------ N2: filters on packed words
@pytest.mark.parametrize("tag,five_prime", [("NGG5p", True), ("NGG3p", False)])
def test_filters_gpu_match_reference_flags(tag, five_prime):
    """gm_site_flags / gm_seed_duplicated against the flags the reference produced on Carsonella."""
    cols = load_carsonella(tag)
    L = len(cols["target"][0])
    packed = _lib.pack(cols["target"], L)
    got = _lib.site_flags(packed, L, ["NRAGCA", core.reverse_complement("NRAGCA")])
    assert got.tolist() == cols["hasrestrictionsite"]
    start = 0 if five_prime else L - 10
    assert _lib.seed_duplicated(packed, L, start, 10).tolist() == cols["isseedduplicated"]
    # and through the mirrored TargetProcessor methods
    tl = core.TargetProcessor(targets_frame(cols["target"], "hamming", five_prime),
                              lsr=10, editdist=2, knum=2)
    tl.check_restriction_enzymes(["NRAGCA"])
    tl.find_unique_near_pam()
    assert tl.targets["hasrestrictionsite"].tolist() == cols["hasrestrictionsite"]
    assert tl.targets["isseedduplicated"].tolist() == cols["isseedduplicated"]


@pytest.mark.parametrize("L", [10, 20, 27])
def test_filters_gpu_match_oracle_edge_cases(oracle, L):
    rng = random.Random(L)
    base = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(3000)]
    targets = base + [rng.choice(base) for _ in range(1500)]            # many repeated seeds
    targets += [t[:L // 2] + rng.choice(base)[L // 2:] for t in base[:500]]
    rng.shuffle(targets)
    packed = _lib.pack(targets, L)
    for enzymes in (["GAATTC"], ["NRAGCA", "GGTCTC", "CACNNNGTG"], ["N"], ["A" * (L + 1)], ["XNNX"], ["BDHV"],
                    [""], ["ACGT", "acgt"]):
        sites = []
        for e in set(enzymes):
            sites += [e.upper(), core.reverse_complement(e.upper())]
        assert _lib.site_flags(packed, L, sites).tolist() == oracle.restriction_flags(targets, enzymes), enzymes
    for five_prime in (True, False):
        for lsr in (0, 1, 5, L - 1, L, L + 3, 2 * L, 2 * L + 5):
            tl = core.TargetProcessor(targets_frame(targets, "hamming", five_prime),
                                      lsr=lsr, editdist=2, knum=2)
            tl.find_unique_near_pam()
            seeds, flags = oracle.seed_duplicated(targets, lsr, five_prime)
            assert tl.targets["seedseq"].tolist() == seeds
            assert tl.targets["isseedduplicated"].tolist() == flags, (five_prime, lsr)
    assert _lib.site_flags(packed[:0], L, ["ACGT"]).shape == (0,)
    assert _lib.seed_duplicated(packed[:0], L, 0, 5).shape == (0,)
    with pytest.raises(ValueError):
        _lib.site_flags(packed, L, ["ACGU"])


def test_filters_full_size_properties():
    """S1-sized frame: flags agree with numpy on the packed words (no oracle at this size)."""
    L, n = 20, 600_000
    rng = np.random.default_rng(5)
    packed = core.pack_codes(rng.integers(0, 4, size=(n, L), dtype=np.uint8))
    packed[rng.integers(0, n, size=50_000)] = packed[rng.integers(0, n, size=50_000)]
    for start, count in ((0, 10), (10, 10), (0, 20), (3, 1)):
        seeds = (packed >> np.uint64(2 * start)) & np.uint64((1 << (2 * count)) - 1)
        first = np.ones(n, bool)
        _, idx = np.unique(seeds, return_index=True)       # index of each seed's first occurrence
        first[idx] = False
        assert (_lib.seed_duplicated(packed, L, start, count) == first).all()
    # GAATTC (its own reverse complement): compare with a direct scan of the 2-bit codes
    site = np.array(["ACGT".index(c) for c in "GAATTC"], np.uint64)
    want = np.zeros(n, bool)
    for o in range(L - 6 + 1):
        ok = np.ones(n, bool)
        for j, c in enumerate(site):
            ok &= ((packed >> np.uint64(2 * (o + j))) & np.uint64(3)) == c
        want |= ok
    assert (_lib.site_flags(packed, L, ["GAATTC"]) == want).all() and want.any()


# ------------------------------------------------------------------ output-table columns
@pytest.mark.parametrize("dtype", ["hamming", "leven"])
def test_output_columns_byte_identical(oracle, cfd_table, dtype):
    """'Similar guides' / 'Similar guide distances' (reference core.py:913-931) and the CFD columns built on them
    (core.py:1129-1148) on Carsonella: the vectorised host path over the GPU result against the oracle's restatement
    of the reference's per-row closures -- byte-identical strings."""
    cols = load_carsonella("NGG5p")
    df = core.PamTarget("NGG", "5prime", dtype).find_targets(carsonella(), 20)
    tl = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=10)
    tl.check_restriction_enzymes(["NRAGCA"])
    tl.find_unique_near_pam()
    tl.create_index(configpath=CONFIG)
    tl.get_neighbors(configpath=CONFIG)
    otp = oracle.OracleTargetProcessor(cols["target"], dtype, 2, 10, cols["isseedduplicated"],
                                       cols["hasrestrictionsite"])
    otp.create_index()
    otp.get_neighbors()
    guides = list(otp.neighbors.keys())
    want_s, want_d = oracle.format_neighbor_columns(otp.neighbors, guides)
    got_s, got_d = core.format_neighbor_columns(tl, guides)
    assert got_s == want_s and got_d == want_d
    tab = core.neighbor_table(tl)
    assert set(tab["Guide sequence"]) == set(guides)
    if dtype == "hamming":
        out = core.cfd_score(tab)
        want_lists, want_max = oracle.cfd_score_rows(tab["Guide sequence"].tolist(), tab["Similar guides"].tolist(),
                                                     cfd_table)
        assert out["CFD Similar Guides"].tolist() == want_lists          # lists of str(float)
        assert out["Max CFD"].tolist() == want_max
        # the packed side-car neighbor_table attaches is used only while it describes the frame: a filtered or
        # re-ordered frame goes through the strings and gives the same numbers
        import pandas as pd
        assert isinstance(tab.attrs.get("gm_cfd_token"), str) and core._cfd_side_car(
            tab, tab["Guide sequence"].tolist(), tab["Similar guides"].tolist()) is not None
        sub = tab.iloc[::-2].reset_index(drop=True)[["Guide sequence", "Similar guides"]]
        out2 = core.cfd_score(sub)
        assert out2["CFD Similar Guides"].tolist() == want_lists[::-2] and out2["Max CFD"].tolist() == want_max[::-2]
        # same length, one row in the MIDDLE changed (not a probe row of the old check): the side-car must not be used
        edit = tab[["Guide sequence", "Similar guides"]].copy()
        edit.attrs["gm_cfd_token"] = tab.attrs["gm_cfd_token"]
        r = len(edit) // 2 + 1
        parts = edit.at[r, "Similar guides"].split(";")
        edit.at[r, "Similar guides"] = ";".join(parts[::-1])
        assert core._cfd_side_car(edit, edit["Guide sequence"].tolist(), edit["Similar guides"].tolist()) is None
        out3 = core.cfd_score(edit)
        assert out3["CFD Similar Guides"].tolist()[r] == want_lists[r][::-1] and out3["Max CFD"].tolist() == want_max
        # frames made by neighbor_table concatenate (pandas compares attrs with ==: arrays in there raised)
        assert len(pd.concat([tab, tab], ignore_index=True)) == 2 * len(tab)


# ------------------------------------------------------------------ controls at genome scale (a6)
def test_get_control_seqs_at_genome_scale(tmp_path):
    """reference core.py:586-620 on an E. coli sized pool (workload S1).  The search is exact, so the shipped
    CONTROL.MINIMUM_HMDIST = 7 cannot be met (a random 20-mer has ~18 targets within 6 mismatches): the mirrored loop
    would walk off CONTROL_SEARCH_MULTIPLE after the 10 000 x draw and raise IndexError, as the reference's does -- here
    the same IndexError comes after the first draw, with the config key named.  With a reachable minimum the call returns the (n, 3) table; the nearest distances behind
    it are the same from the pigeonhole-partitioned scan and from the plain scan."""
    import yaml
    from guidemaker_amd import workloads
    pool = workloads.make_pool("s1")
    L = pool["guidelen"]
    tl = core.TargetProcessor(targets=targets_frame(_lib.unpack(pool["packed_db"][:5], L)), lsr=10, editdist=2, knum=10)
    tl._db_targets = np.asarray(["N"] * pool["packed_db"].shape[0], dtype=object)      # (only its length is used)
    tl._guidelen = L
    tl.nmslib_index = _lib.Index(pool["packed_db"], L, _lib.METRIC_HAMMING)
    genome = [Rec("ACGT" * 5000 + "GC" * 160, "g")]                                    # GC content 0.508
    np.random.seed(7)
    import time
    t0 = time.perf_counter()
    with pytest.raises(IndexError, match="MINIMUM_HMDIST") as exc:
        tl.get_control_seqs(genome, configpath=CONFIG, length=L, n=10)
    # by default the loop is the reference's: every multiple is drawn (111 100 sequences) before the list is walked off,
    # and the global RNG is left where the reference would leave it
    assert "after CONTROL.CONTROL_SEARCH_MULTIPLE" in str(exc.value) and time.perf_counter() - t0 < 60.0
    state_after_default = np.random.random_sample()
    np.random.seed(7)
    for mult in (10, 100, 1000, 10000):
        core.TargetProcessor._draw_control_codes(10 * mult, L, tl.gc_percent / 100)
    assert np.random.random_sample() == state_after_default
    with open(CONFIG) as fh:
        cfg = yaml.safe_load(fh)
    # opt-in (GPU: CONTROL_FAIL_FAST): the same IndexError as soon as two consecutive draws (10 n + 100 n sequences) rule
    # the largest draw out
    cfg.setdefault("GPU", {})["CONTROL_FAIL_FAST"] = True
    ff_path = tmp_path / "cfg_ff.yaml"
    ff_path.write_text(yaml.safe_dump(cfg))
    np.random.seed(7)
    with pytest.raises(IndexError, match="MINIMUM_HMDIST") as exc:
        tl.get_control_seqs(genome, configpath=str(ff_path), length=L, n=10)
    assert "first 1100 draws" in str(exc.value) and "CONTROL_FAIL_FAST" in str(exc.value)
    cfg["GPU"]["CONTROL_FAIL_FAST"] = False
    cfg["CONTROL"]["MINIMUM_HMDIST"] = 5
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    np.random.seed(7)
    mn, med, table = tl.get_control_seqs(genome, configpath=str(cfg_path), length=L, n=100)
    assert table.shape == (100, 3) and mn >= 5 and list(table.columns) == ["name", "Sequences", "Hamming distance"]
    assert tl.ncontrolsearched in (1000, 10000, 100000, 1000000)
    tl.nmslib_index.set_variant(8)
    np.random.seed(7)
    mn8, med8, table8 = tl.get_control_seqs(genome, configpath=str(cfg_path), length=L, n=100)
    assert (mn8, med8) == (mn, med) and table8.equals(table)
    # nearest-distance histogram of 200 000 random controls: pigeonhole-partitioned scan == plain scan
    probe = core.pack_codes(np.random.default_rng(3).integers(0, 4, size=(200_000, L), dtype=np.uint8))
    d_plain = tl.nmslib_index.nearest(probe)
    tl.nmslib_index.set_variant(0)
    d_pg = tl.nmslib_index.nearest(probe)
    assert tl.nmslib_index.last_kernel()[0] .startswith("k_hamming_scan_pg")
    assert (np.bincount(d_pg, minlength=21) == np.bincount(d_plain, minlength=21)).all() and (d_pg == d_plain).all()
    with pytest.raises(ValueError):
        tl.get_control_seqs(genome, configpath=str(cfg_path), length=L + 1, n=10)
    tl.nmslib_index.close()


# ------------------------------------------------------------------ randomised differential test (time-boxed)
def test_fuzz_against_oracle_time_boxed(oracle):
    """random guide length 10..27, metric, k, ragged pool / candidate sizes, duplicate
    rate, base skew and (Hamming) a random forced pigeonhole layout, each case compared with the oracle."""
    import fuzz_parity
    n_cases, bad = fuzz_parity.run(seconds=25.0, seed=2)
    assert bad is None, bad
    assert n_cases >= 20


def test_k1p_stress_time_boxed():
    """tools/k1p_stress.py for 20 s: random pool size (40 k .. 400 k: the planner, the prefix re-scan and the id cut-off
    are live), guide length, k, base skew, planted near-duplicates and cut-off setting -- the whole answer against the
    plain scan (variant 8)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "k1p_stress.py"), "--seconds", "20", "--seed", "5"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "MISMATCH" not in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "cases" in out.stdout.splitlines()[-1]


def _random_packed(n, L, seed, planted=0.0):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    if planted > 0:
        m = int(n * planted)
        dst, src = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
        codes[dst] = codes[src]
        for _ in range(3):
            hit = rng.random(m) < 0.67
            codes[dst[hit], rng.integers(0, L, size=int(hit.sum()))] = rng.integers(0, 4, size=int(hit.sum()))
    db = np.unique(core.pack_codes(codes))
    rng.shuffle(db)
    return db


@pytest.mark.parametrize("n,L,k,layout", [(300_000, 20, 10, 727261), (400_000, 27, 10, 72727161), (250_000, 24, 20, 82824140)])
def test_layouts_with_two_radius2_blocks(oracle, n, L, k, layout):
    """The planner may give TWO blocks the two-substitution neighbourhood (long guides on pools of millions: L = 27,
    4.2 M targets 43 -> 32 ms per query).  Forced here on pools the oracle can check: blocks (7,7,6) radii (2,2,1),
    (7,7,7,6) radii (2,2,1,1), (8,8,4,4) radii (2,2,1,0) -- against the plain scan on every row and the oracle on a sample."""
    db = _random_packed(n, L, seed=900 + L + k, planted=0.01)
    q = np.concatenate([db[: n // 3], _random_packed(3000, L, seed=78)])
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    try:
        idx.set_param("pg_layout", layout)
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
        d1 = idx.nearest(q)
        assert (d1 == dists[:, 0]).all()
        idx.set_variant(8)
        i8, d8, c8 = idx.knn(q, k)
        assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
        rows = np.sort(np.random.default_rng(4).choice(q.shape[0], size=400, replace=False))
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", k)
        od = np.where(od >= 0, od // 2, -1)
        assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    finally:
        idx.close()


@pytest.mark.parametrize("n,L,k,planted,layout", [(200_000, 20, 3, 0.0, 61515140), (200_000, 20, 3, 0.02, 61515140),
                                                   (400_000, 20, 6, 0.01, 61515140), (300_000, 20, 10, 0.0, 61515140),
                                                   (260_000, 23, 5, 0.01, 0), (150_000, 17, 3, 0.0, 0)])
def test_prefix_rescan_matches_plain_rescan_and_oracle(oracle, n, L, k, planted, layout):
    """Rows that do not fill within the layout's radius: prefix re-scan (their partial lists + a scan of the pool's first
    part) against the plain re-scan of all of them ("pg_prefix" = 0), the plain scan (variant 8) and the oracle on the
    rows with the farthest k-th neighbours (the re-scanned ones are among them)."""
    db = _random_packed(n, L, seed=400 + L + k, planted=planted)
    rng = np.random.default_rng(9)
    q = np.concatenate([db[: n // 2], _random_packed(5000, L, seed=77)])      # pool members + strangers
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    try:
        if layout:      # blocks (6,5,5,4), radii (1,1,1,0) -- bench.py's S1 layout; on these pools it leaves 1 .. 60 % of the rows
            idx.set_param("pg_layout", layout)
        ids, dists, counts = idx.knn(q, k)
        name, rescanned = idx.last_kernel()[0], idx.last_evaluated()[1]
        if layout:
            assert name .startswith("k_hamming_scan_pg") and rescanned >= 256, (name, rescanned)
        idx.set_param("pg_prefix", 0)
        i0, d0, c0 = idx.knn(q, k)
        assert (i0 == ids).all() and (d0 == dists).all() and (c0 == counts).all()
        idx.set_param("pg_prefix", 1)
        d1 = idx.nearest(q)
        assert (d1 == dists[:, 0]).all()
        idx.set_variant(8)
        i8, d8, c8 = idx.knn(q, k)
        assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
        far = np.flatnonzero(dists[:, k - 1] >= np.sort(dists[:, k - 1])[-2000])[:1500]
        rows = np.unique(np.concatenate([far, rng.choice(q.shape[0], size=300, replace=False)]))
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", k)
        od = np.where(od >= 0, od // 2, -1)
        assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    finally:
        idx.close()


@pytest.mark.parametrize("n,L,k,planted,lay1,lay2", [
    (300_000, 20, 10, 0.01, 717160, 51515151),      # (7,7,6) radii (1,1,0) = r 4, then (5,5,5,5) all radius 1 = r 7
    (300_000, 20, 10, 0.0, 717061, 61515140),       # (7,7,6) radii (1,0,1), then bench.py's S1 layout (r 6)
    (400_000, 20, 5, 0.02, 50505050, 717160),       # four plain blocks (r 3), then r 4
    (250_000, 23, 10, 0.01, 818170, 61616150),      # L = 23: (8,8,7) radii (1,1,0) = r 4, then (6,6,6,5) radii (1,1,1,0) = r 6
    (200_000, 20, 20, 0.01, 717160, 61515140),      # 20 neighbours
])
def test_cascade_of_k1p_levels_is_exact(oracle, n, L, k, planted, lay1, lay2):
    """The rows a K1P pass leaves short go, as a compact candidate set, through a deeper K1P level with a larger radius
    and a layout of its own (gm_set_param "pg_cascade" = 2 forces that whenever >= 64 rows are left; "pg_layout2" = the
    deeper levels' layout), and only what the last level leaves takes the plain re-scan.  Against the same query without
    the cascade, the plain scan and the oracle on the rows with the farthest k-th neighbours + a random sample."""
    db = _random_packed(n, L, seed=1300 + L + k, planted=planted)
    rng = np.random.default_rng(19)
    q = np.concatenate([db[: n // 2], _random_packed(4000, L, seed=79)])
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    try:
        idx.set_param("pg_layout", lay1)
        idx.set_param("pg_cascade", 2)
        idx.set_param("pg_layout2", lay2)
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
        info = idx.cascade_info()
        assert len(info) >= 2 and info[0]["n_q"] == q.shape[0] and info[1]["n_q"] == info[0]["left"] >= 64, info
        assert info[1]["radius"] > info[0]["radius"], info
        d1 = idx.nearest(q)
        assert (d1 == dists[:, 0]).all()
        for hit in (0, 2):      # both hit paths on every level (the default chooses per level by the candidate count)
            idx.set_param("pg_hit", hit)
            ih, dh, ch = idx.knn(q, k)
            assert len(idx.cascade_info()) >= 2
            assert (ih == ids).all() and (dh == dists).all() and (ch == counts).all(), hit
        idx.set_param("pg_hit", 1)
        idx.set_param("pg_cascade", 0)
        i0, d0, c0 = idx.knn(q, k)
        assert len(idx.cascade_info()) == 1
        assert (i0 == ids).all() and (d0 == dists).all() and (c0 == counts).all()
        idx.set_variant(8)
        i8, d8, c8 = idx.knn(q, k)
        assert idx.cascade_info() == []
        assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
        far = np.flatnonzero(dists[:, k - 1] >= np.sort(dists[:, k - 1])[-1500])[:1200]
        rows = np.unique(np.concatenate([far, rng.choice(q.shape[0], size=300, replace=False)]))
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", k)
        od = np.where(od >= 0, od // 2, -1)
        assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    finally:
        idx.close()


@pytest.mark.parametrize("n,L,k,radius,planted", [(60_000, 20, 5, 3, 0.05), (60_000, 23, 5, 4, 0.05), (90_000, 23, 10, 5, 0.03),
                                                   (50_000, 20, 10, 2, 0.1), (70_000, 27, 5, 6, 0.05), (40_000, 16, 20, 3, 0.05)])
def test_leven_partition_is_exact(oracle, n, L, k, radius, planted):
    """K2P, the pigeonhole partition for Levenshtein (gm_set_param "lp_radius" forces its radius; the planner takes it only
    on pools of >= 65 536 targets where a sample says it pays): r + 1 blocks, a target within r carries one of them
    unchanged, displaced by at most min(floor(r / 2), r - j) for the leftmost such block j.  Rows it vouches for (k-th entry
    within r) are final, the others take a second partitioned level of a larger radius ("lp_radius2") and / or the second
    pass under their Hamming seeds.  Against the plain K2B scan ("lp_radius" = -1), the single-pass scan and
    the oracle on a sample that favours rows with near neighbours (planted near-duplicates incl. indels)."""
    rng = np.random.default_rng(1500 + L + k)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    m = int(n * planted)
    dst, src = rng.integers(0, n, size=m), rng.integers(0, n, size=m)
    codes[dst] = codes[src]
    for i in range(m):      # 1 .. 3 edits per planted copy: substitutions and (shift-making) indels
        row = codes[dst[i]].tolist()
        for _ in range(int(rng.integers(1, 4))):
            kind, pos = int(rng.integers(0, 3)), int(rng.integers(0, L))
            if kind == 0:
                row[pos] = int(rng.integers(0, 4))
            elif kind == 1:
                row = row[:pos] + row[pos + 1:] + [int(rng.integers(0, 4))]      # deletion (+ a base at the end)
            else:
                row = (row[:pos] + [int(rng.integers(0, 4))] + row[pos:])[:L]   # insertion (the last base falls off)
        codes[dst[i]] = row
    db = np.unique(core.pack_codes(codes))
    rng.shuffle(db)
    q = np.concatenate([db[: min(len(db), 20_000)], _random_packed(1500, L, seed=80)])
    idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
    try:
        idx.set_param("lp_radius", radius)
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0] == "k_leven_part_bs"
        evaluated, second = idx.last_evaluated()
        assert evaluated > 0      # (pairs incl. the padding of the buckets to whole 256-target tiles)
        info = idx.levpart_info()
        assert info["r1"] == radius and info["r2"] == 0
        # ... and with a second partitioned level of radius + 1 for the rows the first cannot vouch for
        if radius < 7 and L >= 2 * (radius + 2):
            idx.set_param("lp_radius2", radius + 1)
            i2, d2, c2 = idx.knn(q, k)
            info = idx.levpart_info()
            assert info["r1"] == radius and info["r2"] == radius + 1 and info["rows2"] > 0 and info["left2"] <= info["rows2"]
            assert (i2 == ids).all() and (d2 == dists).all() and (c2 == counts).all()
            idx.set_param("lp_radius2", 0)
        idx.set_param("lp_radius", -1)
        i0, d0, c0 = idx.knn(q, k)
        assert idx.last_kernel()[0] == "k_leven_knn_bs"
        assert (i0 == ids).all() and (d0 == dists).all() and (c0 == counts).all()
        idx.set_param("lv_two_pass", 0)
        i1, d1, c1 = idx.knn(q, k)
        assert (i1 == ids).all() and (d1 == dists).all() and (c1 == counts).all()
        near = np.flatnonzero(dists[:, 1] <= radius)[:200]
        rows = np.unique(np.concatenate([near, rng.choice(q.shape[0], size=100, replace=False)]))
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "leven", k)
        assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    finally:
        idx.close()


def test_leven_partition_handle_reuse():
    """One Levenshtein handle, candidate sets of different sizes and k in turn (K2P's plan is cached per (n_q, k), its pool
    copies per radius, its second level has a state of its own): every answer equals the plain K2B scan of a fresh handle
    setting, whatever ran before on the handle."""
    L = 23
    rng = np.random.default_rng(77)
    db = _random_packed(120_000, L, seed=4242, planted=0.04)
    sets = [(db[:30_000], 5), (np.concatenate([db[40_000:49_000], _random_packed(500, L, seed=5)]), 10), (db[:30_000], 5),
            (db[10_000:70_000], 3), (db[:8_200], 20)]
    idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
    ref = _lib.Index(db, L, _lib.METRIC_LEVEN)
    ref.set_param("lp_radius", -1)
    try:
        used = 0
        for step, (q, k) in enumerate(sets):
            if step == 3:
                idx.set_param("lp_radius", 4)      # forced radii from here on (copies rebuilt for another radius)
                idx.set_param("lp_radius2", 6)
            got = idx.knn(q, k)
            used += idx.last_kernel()[0] == "k_leven_part_bs"
            want = ref.knn(q, k)
            assert ref.last_kernel()[0] == "k_leven_knn_bs"
            assert all((a == b).all() for a, b in zip(got, want)), (step, k)
        assert used >= 3      # (the planner's own choice on the 120 k pool, then the forced radii)
        idx.set_param("lp_radius", -1)
        got = idx.knn(sets[0][0], 5)
        assert idx.last_kernel()[0] == "k_leven_knn_bs" and all((a == b).all() for a, b in zip(got, ref.knn(sets[0][0], 5)))
    finally:
        idx.close()
        ref.close()


@pytest.mark.parametrize("n,k,cut", [(500_000, 20, 0), (500_000, 20, 25), (500_000, 20, 4), (300_000, 5, 30), (300_000, 2, 10)])
def test_id_cutoff_is_exact(oracle, n, k, cut):
    """Hits at the layout's radius only from targets with ids below a cut-off ("pg_cut": 0 = the library's choice, else
    that percentage of the ids) -- a tight cut leaves rows short, and those take the re-scan: same answer as without
    (pg_cut = 100), as the plain scan, and as the oracle on the rows with the farthest k-th neighbours."""
    L = 20
    db = _random_packed(n, L, seed=900 + k, planted=0.01)
    rng = np.random.default_rng(4)
    q = np.concatenate([db[: n // 3], _random_packed(4000, L, seed=78)])
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    try:
        idx.set_param("pg_cut", cut)
        ids, dists, counts = idx.knn(q, k)
        assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
        idx.set_param("pg_cut", 100)
        i0, d0, c0 = idx.knn(q, k)
        assert (i0 == ids).all() and (d0 == dists).all() and (c0 == counts).all()
        idx.set_variant(8)
        i8, d8, c8 = idx.knn(q, k)
        assert (i8 == ids).all() and (d8 == dists).all() and (c8 == counts).all()
        far = np.flatnonzero(dists[:, k - 1] >= np.sort(dists[:, k - 1])[-1500])[:1000]
        rows = np.unique(np.concatenate([far, rng.choice(q.shape[0], size=300, replace=False)]))
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", k)
        od = np.where(od >= 0, od // 2, -1)
        assert (oc == counts[rows]).all() and (od == dists[rows]).all() and (oi == ids[rows]).all()
    finally:
        idx.close()


def test_prefix_rescan_nearest_distance(oracle):
    """Nearest-distance entry (controls): random strangers against a genome-sized pool; rows without a hit inside the
    layout's radius take the re-scan -- same distances as the plain scan."""
    L = 20
    db = _random_packed(600_000, L, seed=31)
    q = _random_packed(120_000, L, seed=32)
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    try:
        d1 = idx.nearest(q)
        idx.set_param("pg_prefix", 0)
        d0 = idx.nearest(q)
        idx.set_variant(8)
        d8 = idx.nearest(q)
        assert (d1 == d8).all() and (d0 == d8).all()
        rows = np.random.default_rng(3).choice(q.shape[0], size=200, replace=False)
        _, od, _ = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", 1)
        assert (od[:, 0] // 2 == d1[rows]).all()
    finally:
        idx.close()


@pytest.mark.parametrize("hit", [0, 2])
def test_repeat_flooded_pool_falls_back_to_plain_scan_next_time(oracle, hit):
    """A genome made largely of repeat families (40 segments of 5 kb, 60 copies each at 3 % divergence, in 4.6 Mb --
    tools/repeat_genome_check.py).  With the hit buffer in HBM (pg_hit = 0) the hit lists of the pigeonhole-partitioned
    scan overflow for half the candidates, which then take the plain scan anyway: the first query is exact all the
    same, the handle remembers and the second one goes straight to the plain scan -- same answer.  With the register
    lists (pg_hit = 2) nothing can overflow: the partitioned scan itself answers, both times."""
    from guidemaker_amd import workloads
    L, k, size, seglen = 20, 10, 4_641_652, 5000
    rng = np.random.Generator(np.random.PCG64(99))
    codes = workloads.synth_genome(size, 0.508, rng)
    for _ in range(40):
        src = rng.integers(0, size - seglen)
        seg = codes[src:src + seglen].copy()
        for _ in range(60):
            dst = rng.integers(0, size - seglen)
            cp = seg.copy()
            mut = rng.random(seglen) < 0.03
            cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
            codes[dst:dst + seglen] = cp
    seq = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()
    q = core.scan_targets_gpu(seq, "NGG", "3prime", L)["packed"]
    db = q[np.sort(np.unique(q, return_index=True)[1])]      # first-occurrence order (create_index's dedup)
    idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
    idx.set_param("pg_hit", hit)
    try:
        i1, d1, c1 = idx.knn(q, k)
        first, rescanned = idx.last_kernel()[0], idx.last_evaluated()[1]
        i2, d2, c2 = idx.knn(q, k)
        second = idx.last_kernel()[0]
        assert (i1 == i2).all() and (d1 == d2).all() and (c1 == c2).all()
        if hit == 0:
            assert first .startswith("k_hamming_scan_pg") and rescanned > q.shape[0] // 3, (first, rescanned)
            assert second.startswith("k_hamming_knn_lk"), second
        else:
            assert first .startswith("k_hamming_scan_pg") and second .startswith("k_hamming_scan_pg"), (first, second)
            assert rescanned < q.shape[0] // 3, rescanned
        rows = rng.choice(q.shape[0], size=400, replace=False)
        oi, od, oc = oracle.knn(_packed_to_bytes(db, L), _packed_to_bytes(q[rows], L), L, "hamming", k)
        od = np.where(od >= 0, od // 2, -1)
        assert (oc == c1[rows]).all() and (od == d1[rows]).all() and (oi == i1[rows]).all()
    finally:
        idx.close()


def _packed_to_bytes(packed, L):
    asc = np.frombuffer(b"ACGT", np.uint8)
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    return np.ascontiguousarray(asc[((packed[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])


# ------------------------------------------------------------------ in-library multi-device entry
_MULTI_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from helpers import random_pool
from guidemaker_amd import _lib
from oracle import oracle as orc
assert "torch" not in sys.modules
L, k = 20, 10
db = list(dict.fromkeys(random_pool(6000, L, seed=71)))
q = db[::2] + random_pool(333, L, seed=72)
m = _lib.MultiIndex(_lib.pack(db, L), L, _lib.METRIC_HAMMING, [0])
assert m.devices() == 1 and m.has_comm()          # librccl.so loaded, communicator created
ids, dists, counts = m.knn(_lib.pack(q, L), k)
oi, od, oc = orc.knn(db, q, L, "hamming", k)
od = np.where(od >= 0, od // 2, -1)
assert (ids == oi).all() and (dists == od).all() and (counts == oc).all()
assert m.knn(np.empty((0,), np.uint64), k)[0].shape == (0, k)
m.close()
for bad in ([0, 0], [99]):                        # duplicate ordinal / no such device
    try:
        _lib.MultiIndex(_lib.pack(db, L), L, _lib.METRIC_HAMMING, bad)
        raise SystemExit("expected GmHipError for devices %r" % bad)
    except _lib.GmHipError:
        pass
print("multi ok")
"""


def test_multi_device_entry_matches_single_device():
    """gm_multi_* (one process, n GPUs, ncclCommInitAll + one ncclAllGather of the keys): on this box the device list
    has one entry, which still runs the whole sequence -- shard, scan, key packing, RCCL all-gather, unpack.  Runs in
    a child process without torch: this entry is for hosts that do not use torch.distributed, and one process should
    not drive RCCL through both (PyTorch ships its own copy of the library)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _MULTI_CHILD, root], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "multi ok" in out.stdout, out.stdout + out.stderr


_ORDER_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from guidemaker_amd import _lib
idx = _lib.Index(np.arange(5000, dtype=np.uint64) * 977, 20, _lib.METRIC_HAMMING, 0)
ids, dists, counts = idx.knn(np.arange(10, dtype=np.uint64) * 977, 2)
assert (dists[:, 0] == 0).all()
import torch                                   # AFTER libgmhip.so has initialised the GPU
x = torch.arange(4).cuda() + 1
assert x.cpu().tolist() == [1, 2, 3, 4]
ids2, _, _ = idx.knn(np.arange(10, dtype=np.uint64) * 977, 2)
assert (ids2 == ids).all()
print("order ok")
"""


_VIRTUAL_CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
from helpers import random_pool
from guidemaker_amd import _lib, core
from oracle import oracle as orc
L, k = 20, 10
# (a) a small pool: the plain scan on every shard; candidate counts that do not divide by n, fewer candidates than shards, none
db = list(dict.fromkeys(random_pool(6000, L, seed=81)))
q_all = db[::2] + random_pool(333, L, seed=82)
single = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING, 0)
for n in (2, 4, 8):
    m = _lib.MultiIndex(_lib.pack(db, L), L, _lib.METRIC_HAMMING, [0] * n)
    assert m.devices() == n and not m.has_comm()          # logical shards on one GPU: copies in place of the collective
    for n_q in (len(q_all), 1001, n - 1, 1, 0):
        q = _lib.pack(q_all[:n_q], L) if n_q else np.empty((0,), np.uint64)
        got = m.knn(q, k)
        want = single.knn(q, k)
        assert all((a == b).all() for a, b in zip(got, want)), ("knn", n, n_q)
        d1 = m.nearest(q)
        assert d1.dtype == np.int32 and (d1 == single.nearest(q)).all(), ("nearest", n, n_q)
    if n == 4:
        oi, od, oc = orc.knn(db, q_all, L, "hamming", k)
        got = m.knn(_lib.pack(q_all, L), k)
        assert (got[0] == oi).all() and (got[1] == np.where(od >= 0, od // 2, -1)).all() and (got[2] == oc).all()
    m.close()
single.close()
# (b) a pool on which every shard runs the pigeonhole-partitioned scan (forced layout, both hit paths), 3 shards of 40 001 rows
rng = np.random.default_rng(83)
dbp = np.unique(core.pack_codes(rng.integers(0, 4, size=(120_000, L), dtype=np.uint8)))
rng.shuffle(dbp)
qp = np.concatenate([dbp[:100_003], core.pack_codes(rng.integers(0, 4, size=(20_000, L), dtype=np.uint8))])
single = _lib.Index(dbp, L, _lib.METRIC_HAMMING, 0)
single.set_variant(8)
want = single.knn(qp, k)
want1 = single.nearest(qp)
for hit in (0, 2):
    m = _lib.MultiIndex(dbp, L, _lib.METRIC_HAMMING, [0, 0, 0])
    m.set_param("pg_layout", 61515140)
    m.set_param("pg_hit", hit)
    got = m.knn(qp, k)
    assert all((a == b).all() for a, b in zip(got, want)), ("K1P shards", hit)
    assert (m.nearest(qp) == want1).all()
    m.close()
single.close()
print("virtual ok")
"""


def test_multi_device_fan_out_with_logical_shards():
    """VERDICT r03 missing 2: the n > 1 path of gm_multi_* (one host thread per shard, shard offsets, padding, key
    packing, the gathered table's unpack) had never executed anywhere -- builder and driver boxes have one GPU.
    GMHIP_TEST_VIRTUAL_DEVICES=1 lets the device list repeat an ordinal: 2 / 4 / 8 logical shards on GPU 0, copies in
    place of ncclAllGather, everything else the code a real node runs.  k-NN and the sharded nearest-distance entry
    (gm_multi_nearest_dist: get_control_seqs' query) against one device and the oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GMHIP_TEST_VIRTUAL_DEVICES="1")
    out = subprocess.run([sys.executable, "-c", _VIRTUAL_CHILD, root], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "virtual ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    # without the test switch a repeated ordinal stays an error (the existing child checks that)


def test_library_first_then_torch_in_one_process():
    """`import guidemaker_amd` (libgmhip.so -> HIP runtime) BEFORE `import torch`: PyTorch ships its own copy of the
    runtime under the same SONAME, and with the system copy loaded first torch reported "No HIP GPUs are available"
    (_lib._share_torch_hip_runtime loads torch's copy first when torch is installed)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _ORDER_CHILD, root], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "order ok" in out.stdout, out.stdout + out.stderr


# ------------------------------------------------------------------ every guide length, both metrics
@pytest.mark.parametrize("L", list(range(10, 28)))
def test_every_guide_length_matches_oracle(oracle, L):
    """Myers/Hyyro bit-sliced DP == textbook DP and bit-sliced Hamming == popcount of XOR over the one-hot words, for
    every guide length the CLI accepts (10..27): seeded bounds put most waves on the banded / narrow-band paths, the
    unseeded rows (k larger than the neighbours in reach) on the full-matrix path."""
    db = list(dict.fromkeys(random_pool(2500 + 13 * L, L, seed=500 + L, planted=0.05, indel=True)))
    q = db[::5] + random_pool(150, L, seed=600 + L, indel=True)
    for metric in ("leven", "hamming"):
        for k in (3, 12):
            ids, dists, counts = _gpu_knn(db, q, L, metric, k)
            oi, od, oc = _oracle_true_units(oracle, db, q, L, metric, k)
            assert (counts == oc).all() and (dists == od).all() and (ids == oi).all(), (L, metric, k)


def test_pack_unpack_roundtrip_and_one_hot_equivalence(oracle):
    """gm_pack / gm_unpack round trip for every length, and Hamming on the 2-bit packing == nmslib's bit_hamming on
    the reference's one-hot strings / 2 (core.py:379-386, :510-514)."""
    rng = random.Random(5)
    for L in (1, 2, 9, 10, 20, 27):
        seqs = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(257)]
        packed = _lib.pack(seqs, L)
        assert _lib.unpack(packed, L) == seqs
        assert int(packed.max()) < (1 << (2 * L))
    L = 20
    seqs = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(40)]
    onehot = oracle.one_hot_encode(seqs)
    idx = _lib.Index(_lib.pack(seqs, L), L, _lib.METRIC_HAMMING)
    ids, dists, counts = idx.knn(_lib.pack(seqs, L), len(seqs))
    idx.close()
    for i in range(len(seqs)):
        for j, d in zip(ids[i].tolist(), dists[i].tolist()):
            assert 2 * d == oracle.py_bit_hamming(onehot[i], onehot[j])


# ------------------------------------------------------------------ RCCL path of the one-process-per-GPU driver
def test_sharded_knn_index_over_rccl_single_rank(oracle):
    """guidemaker_amd.distributed on the real backend (nccl = RCCL) with a one-rank group: the device-resident
    sharded_knn_index (Index.knn_dev -> key packing -> all_gather_into_tensor -> unpack) and the asynchronous
    double-buffered gather bench.py uses.  (More than one rank needs more than one GPU; the rank arithmetic is
    covered by the gloo tests.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from guidemaker_amd import distributed
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        L, k = 20, 10
        db = list(dict.fromkeys(random_pool(5000, L, seed=81)))
        q = db[::3] + random_pool(257, L, seed=82)
        idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING)
        qp = _lib.pack(q, L)
        ids, dists, counts = distributed.sharded_knn_index(idx, qp, k, 0)
        oi, od, oc = _oracle_true_units(oracle, db, q, L, "hamming", k)
        assert (ids == oi).all() and (dists == od).all() and (counts == oc).all()
        dev = torch.device("cuda", 0)
        t_ids, t_d = torch.from_numpy(oi).to(dev), torch.from_numpy(od.astype(np.int32)).to(dev)
        bufs = [torch.empty((len(q), k), dtype=torch.int32, device=dev) for _ in range(2)]
        pend = [distributed.all_gather_neighbors_async(t_ids, t_d, len(q), out=b) for b in bufs]
        for p in pend:
            gi, gd, gc = p.result()
            assert (gi.cpu().numpy() == oi).all() and (gd.cpu().numpy() == od).all() and (gc.cpu().numpy() == oc).all()
        idx.close()
    finally:
        dist.destroy_process_group()


def test_timing_knob_is_not_in_the_release_library():
    """gm_set_param("pg_debug") (round 3: skip work for timing, results wrong) answers GM_ERR_ARG on a live handle."""
    db = _random_packed(5000, 20, seed=3)
    idx = _lib.Index(db, 20, _lib.METRIC_HAMMING)
    try:
        with pytest.raises(_lib.GmHipError):
            idx.set_param("pg_debug", 2)
        idx.set_param("pg_hit", 2)      # (the hit path IS a knob: results identical)
        with pytest.raises(_lib.GmHipError):
            idx.set_param("pg_hit", 3)
    finally:
        idx.close()


def test_speculative_follow_ups_are_exact(oracle):
    """gm_set_param "pg_spec": a query of the same shape as the handle's previous one enqueues the follow-ups of the rows
    each K1P level leaves short on PREDICTED counts (no host wait between the kernels) and validates once at the end.
    (a) repeated identical queries: the speculative runs equal the waiting one and the plain scan; (b) the same shape
    with very different content -- five times the rows left short, then hardly any -- must come out exact too (the
    misprediction is caught and the query runs again the waiting way); cascade levels, prefix and plain re-scans."""
    n, L, k = 300_000, 20, 10
    db = _random_packed(n, L, seed=4401, planted=0.01)
    members = db[:150_000]
    foreign = _random_packed(200_000, L, seed=4402)[:150_000]
    mixed = np.concatenate([members[:75_000], foreign[:75_000]])
    plain = _lib.Index(db, L, _lib.METRIC_HAMMING)
    plain.set_variant(8)
    want = {name: plain.knn(q, k) for name, q in (("members", members), ("foreign", foreign), ("mixed", mixed))}
    want1 = plain.nearest(foreign)
    plain.close()
    for setup in ({}, {"pg_layout": 717160, "pg_cascade": 2, "pg_layout2": 61515140}, {"pg_hit": 2},
                  {"pg_group": 2, "pg_slots": 2, "pg_hit": 2, "pg_layout": 717160, "pg_cascade": 2, "pg_layout2": 61515140}):
        idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
        try:
            for name, val in setup.items():
                idx.set_param(name, val)
            how = []
            for name, q in (("members", members), ("members", members), ("members", members), ("foreign", foreign),
                            ("foreign", foreign), ("mixed", mixed), ("members", members), ("members", members)):
                got = idx.knn(q, k)
                assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
                assert all((a == b).all() for a, b in zip(got, want[name])), (setup, name)
                how.append(idx.last_speculation())
            # the first query waits; its repeats run on its counts; a query of the same shape that leaves more rows than
            # provided for is caught (2) and answered the waiting way, as are the next few
            assert how[0] == 0 and how[1] == 1 and how[2] == 1, how
            assert set(how) <= {0, 1, 2}
            assert (idx.nearest(foreign) == want1).all() and (idx.nearest(foreign) == want1).all()
            idx.set_param("pg_spec", 0)
            got = idx.knn(members, k)
            assert idx.last_speculation() == 0
            assert all((a == b).all() for a, b in zip(got, want["members"]))
        finally:
            idx.close()
