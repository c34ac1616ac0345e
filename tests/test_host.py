"""CPU tests of the host side: PAM scan, filters, RNG stream, packing, and the C-ABI library's
exported symbols.  No compute entry point is called here (there is no GPU in this container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import load_json, load_carsonella, ROOT
from helpers import Rec, carsonella

import guidemaker_amd
from guidemaker_amd import _lib, core


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gmhip.h")).read()
    declared = set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", header))
    declared.discard("gm_index")
    assert declared == set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in _lib.lib().gm_version()


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.GmHipError, match="no HIP device"):
        _lib.Index(_lib.pack(["ACGTACGTAC"], 10), 10, _lib.METRIC_HAMMING)
    with pytest.raises(_lib.GmHipError, match="no HIP device"):
        core.calc_cfd("ACGTACGTAC", "ACGTACGTAA")


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "guidemaker_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "core.py" and "oracle" not in text, f


def test_pack_roundtrip_and_layout():
    seqs = ["ACGTACGTAC", "TTTTTTTTTT", "AAAAAAAAAA", "GATTACAGAT"]
    p = _lib.pack(seqs, 10)
    assert p[0] == sum(c << (2 * i) for i, c in enumerate([0, 1, 2, 3, 0, 1, 2, 3, 0, 1]))
    assert p[1] == (1 << 20) - 1 and p[2] == 0
    assert _lib.unpack(p, 10) == seqs
    codes = np.array([["ACGT".index(c) for c in s] for s in seqs], dtype=np.uint8)
    assert (core.pack_codes(codes) == p).all()
    for L in range(1, 28):
        s = ("ACGT" * 7)[:L]
        assert _lib.unpack(_lib.pack([s], L), L) == [s]
    with pytest.raises(ValueError):
        _lib.pack(["ACGTNCGTAC"], 10)
    with pytest.raises(ValueError):
        _lib.pack(["ACGT", "ACG"], 4)
    with pytest.raises(_lib.GmHipError):
        _lib.pack(["A" * 28], 28)


def test_find_targets_known_answers():
    kat = load_json("find_targets_kat.json")
    for name, case in kat["cases"].items():
        pt = core.PamTarget(case["pam"], case["orientation"], "hamming")
        df = pt.find_targets([Rec(case["seq"], name)], case["target_len"])
        got = [[r.target, str(r.exact_pam), int(r.start), int(r.stop), bool(r.strand),
                bool(r.pam_orientation), r.target_seq30] for r in df.itertuples()]
        assert got == case["rows"], name
    # the two assertions of the reference's tests/test_core.py:41-57
    c5 = kat["cases"]["testseq1_NGG_5p_L6"]["rows"]
    assert c5[0][0] == "ATGCAC" and c5[1][0] == "TAACAA"
    c3 = kat["cases"]["testseq1_NGG_3p_L6"]["rows"]
    assert c3[0][0] == "ATGATC" and c3[1][0] == "ATTAGA"


@pytest.mark.parametrize("tag,orient", [("NGG5p", "5prime"), ("NGG3p", "3prime")])
def test_carsonella_targets_and_filters(tag, orient):
    cols = load_carsonella(tag)
    pt = core.PamTarget("NGG", orient, "hamming")
    df = pt.find_targets(carsonella(), 20)
    assert df["target"].tolist() == cols["target"]
    assert df["start"].tolist() == cols["start"]
    assert df["stop"].tolist() == cols["stop"]
    assert df["strand"].tolist() == cols["strand"]
    assert list(df.columns) == ["target", "exact_pam", "start", "stop", "strand", "pam_orientation",
                                "target_seq30", "seqid", "seedseq", "hasrestrictionsite",
                                "isseedduplicated", "dtype"]
    tp = core.TargetProcessor(df, lsr=10, editdist=2, knum=10)
    tp.check_restriction_enzymes(["NRAGCA"])
    tp.find_unique_near_pam()
    assert tp.targets["isseedduplicated"].tolist() == cols["isseedduplicated"]
    assert tp.targets["hasrestrictionsite"].tolist() == cols["hasrestrictionsite"]
    summ = load_json("carsonella_summary.json")["summary"][tag]
    assert len(tp._query_targets()) == summ["n_query"]
    bed = tp.export_bed()
    assert bed.shape == (summ["n_bed"], 5)
    assert list(bed.columns) == ["chrom", "chromstart", "chromend", "name", "strand"]
    # reference core.py:538-542 on the golden columns: seed-unique rows, '+' / '-', ordered by (record, start) -- stable
    keep = [i for i, d in enumerate(cols["isseedduplicated"]) if not d]
    keep.sort(key=lambda i: cols["start"][i])
    assert bed["name"].tolist() == [cols["target"][i] for i in keep]
    assert bed["chromstart"].tolist() == [cols["start"][i] for i in keep]
    assert bed["chromend"].tolist() == [cols["stop"][i] for i in keep]
    assert bed["strand"].tolist() == ["+" if cols["strand"][i] else "-" for i in keep]
    assert bed.index.tolist() == keep and bed["strand"].dtype == object
    assert len(tp) == summ["n_targets"] and "potential PAM targets" in str(tp)


def test_three_row_filters():
    # reference tests/test_core.py:86-102
    from helpers import targets_frame
    kat = load_json("knn_kat.json")["three_row"]
    tl = core.TargetProcessor(targets_frame(kat["targets"]), lsr=10, editdist=2, knum=2)
    tl.check_restriction_enzymes(["NGGTAB"])
    assert tl.targets["hasrestrictionsite"][0] == True
    tl.find_unique_near_pam()
    assert tl.targets[tl.targets["isseedduplicated"] == False].shape == (2, 12)


def test_extend_ambiguous_dna():
    kat = load_json("knn_kat.json")["extend_ambiguous_dna"]
    for k, v in kat.items():
        assert core.extend_ambiguous_dna(k) == v


def test_pam_asserts():
    with pytest.raises(AssertionError):
        core.PamTarget("NGZ", "5prime", "hamming")
    with pytest.raises(AssertionError):
        core.PamTarget("NGG", "sideways", "hamming")
    assert core.PamTarget("ngg", "5prime", "hamming").pam == "NGG"
    assert str(core.PamTarget("NGG", "5prime", "hamming")) == "A PAM object: NGG"


def test_control_draw_consumes_rng_like_np_random_choice():
    # reference core.py:590-592 draws each control with np.random.choice(..., p=...)
    gc, length, count = 0.37, 20, 50
    p = [gc / 2, gc / 2, (1 - gc) / 2, (1 - gc) / 2]
    np.random.seed(123)
    want = ["".join(np.random.choice(a=["G", "C", "A", "T"], size=length, replace=True, p=p))
            for _ in range(count)]
    after_want = np.random.random_sample()
    np.random.seed(123)
    got = core.TargetProcessor._draw_control_seqs(count, length, gc)
    after_got = np.random.random_sample()
    assert got == want and after_got == after_want


def test_cfd_length_assert():
    with pytest.raises(AssertionError, match="lengths wt and off differ"):
        core.calc_cfd("ACGT", "ACG")


def test_neighbor_map_schema_and_duplicates():
    # reference core.py:513-523: dict keyed by candidate string; later duplicates overwrite earlier ones
    ids = np.array([[0, 1, -1], [1, 0, 2], [0, 2, 1], [2, 1, 0]], np.int32)
    dists = np.array([[0, 3, -1], [0, 3, 4], [0, 5, 6], [0, 1, 2]], np.int32)
    counts = np.array([2, 3, 3, 3], np.int32)
    db = np.array(["AA", "CC", "GG"], dtype=object)
    nm = core.NeighborMap(["AA", "CC", "AA", "GG"], [0, 1, 2], ids, dists, counts, db)   # row 3 not retained
    assert len(nm) == 2 and list(nm.keys()) == ["AA", "CC"] and "GG" not in nm
    assert nm["AA"] == {"target": "AA", "neighbors": {"seqs": ["AA", "GG", "CC"], "dist": [0, 5, 6]}}   # last dup
    assert nm["CC"]["neighbors"]["dist"] == [0, 3, 4]
    assert dict(nm)["CC"]["neighbors"]["seqs"] == ["CC", "AA", "GG"]
    assert nm.keys_array.tolist() == ["AA", "CC", "AA"]
    with pytest.raises(KeyError):
        nm["GG"]
    empty = core.NeighborMap([], [], ids[:0], dists[:0], counts[:0], db)
    assert len(empty) == 0 and list(empty) == []


def test_pigeonhole_planner_choices():
    """gm_pigeon_plan is host arithmetic: runs without a GPU.  Layouts for the BASELINE shapes, encoded as
    B + 100 n1 + 1000 len1 + 10000 n2 + 100000 len2 (include/gmhip.h), and the cases where the plain scan must stay
    in charge."""
    from guidemaker_amd import _lib

    def decode(code):
        return dict(B=code % 100, n1=code % 1000 // 100, len1=code % 10000 // 1000, n2=code % 100000 // 10000,
                    len2=code // 100000)

    def radius(code):
        d = decode(code)
        return 3 * d["n2"] + 2 * d["n1"] + (d["B"] - d["n1"] - d["n2"]) - 1

    s1 = decode(_lib.pigeon_plan(599_093, 20, 10))           # S1: blocks 6,5,5,4, neighbour buckets of the first three
    assert (s1["B"], s1["n1"], s1["n2"], s1["len1"]) == (4, 3, 0, 6) and radius(_lib.pigeon_plan(599_093, 20, 10)) == 6
    s4 = decode(_lib.pigeon_plan(3_124_711, 20, 20))         # S4: one block with the two-substitution neighbourhood
    assert (s4["B"], s4["n1"], s4["n2"]) == (3, 2, 1) and s4["len2"] in (7, 8) and radius(_lib.pigeon_plan(3_124_711, 20, 20)) == 6
    assert radius(_lib.pigeon_plan(891_135, 20, 5, 0.2637)) == 6     # S3 (gc 0.383)
    assert _lib.pigeon_plan(3_814, 20, 10) == 0            # Carsonella-sized pool: plain scan
    assert _lib.pigeon_plan(600_000, 20, 21) == 0          # lists beyond the register capacity
    assert _lib.pigeon_plan(600_000, 11, 10) == 0          # too short
    for n in (40_000, 200_000, 1_000_000, 20_000_000):
        for L in range(12, 28):
            for k in (1, 2, 10, 20):
                code = _lib.pigeon_plan(n, L, k)
                d = decode(code)
                assert code == 0 or (2 <= d["B"] <= 8 and d["n1"] + d["n2"] <= d["B"] and d["n1"] <= 5 and d["n2"] <= 2
                                     and d["len1"] <= 8 and d["len2"] <= 8), (n, L, k, code)
    long_big = decode(_lib.pigeon_plan(4_189_098, 27, 10))     # long guides, millions of targets: two radius-2 blocks
    assert long_big["n2"] == 2 and radius(_lib.pigeon_plan(4_189_098, 27, 10)) >= 8
    # a larger pool never needs a wider radius
    assert radius(_lib.pigeon_plan(20_000_000, 20, 10)) <= radius(_lib.pigeon_plan(600_000, 20, 10))


def test_levenshtein_partition_layouts():
    """K2P's layouts (host arithmetic): r + 1 blocks covering the guide; block j is looked up under the displacements
    |d| <= min(floor(r / 2), r - j) (block 0: none) that keep it inside the sequence -- the bound for the LEFTMOST block no
    edit touches --; the block lengths minimise the modelled share of the pool visited (checked against every split for
    the short cases); and the layouts do find every pair within r (random edit scripts, pure-Python edit distance)."""
    import itertools
    import random
    from guidemaker_amd import _lib

    def shift(r, j):
        return 0 if j == 0 else min(r // 2, r - j)

    def copies(L, r, blocks):
        out, pos = [], 0
        for j, b in enumerate(blocks):
            out.append([d for d in range(-shift(r, j), shift(r, j) + 1) if pos + d >= 0 and pos + d + b <= L])
            pos += b
        return out

    def fraction(L, r, blocks):
        return sum(0.25 ** b * len(c) for b, c in zip(blocks, copies(L, r, blocks)))

    for L in range(12, 28):
        for r in range(1, 8):
            lay = _lib.levpart_layout(L, r)
            if lay is None:
                assert L < 2 * (r + 1) or L > 8 * (r + 1)
                continue
            blocks = lay["blocks"]
            assert len(blocks) == r + 1 and sum(blocks) == L and min(blocks) >= 2 and max(blocks) <= 8
            assert lay["copies"] == sum(len(c) for c in copies(L, r, blocks)) <= 56
            assert abs(lay["fraction"] - fraction(L, r, blocks)) < 1e-12
            if (r + 1) <= 5 or L <= 16:      # every split into r + 1 blocks of 2 .. 8 bases: none visits less
                best = min(fraction(L, r, c) for c in itertools.product(range(2, 9), repeat=r + 1) if sum(c) == L)
                assert lay["fraction"] <= best + 1e-12
    f = {r: _lib.levpart_layout(23, r)["fraction"] for r in range(3, 8)}
    assert f[4] < 0.04 and f[5] < 0.10 and 0.2 < f[6] < 0.4 and f[7] > 0.5      # (uniform bases)
    assert _lib.levpart_layout(23, 0) is None and _lib.levpart_layout(23, 8) is None and _lib.levpart_layout(12, 7) is None

    def edit_distance(a, b):
        prev = list(range(len(b) + 1))
        for i in range(1, len(a) + 1):
            cur = [i] + [0] * len(b)
            for jj in range(1, len(b) + 1):
                cur[jj] = min(prev[jj] + 1, cur[jj - 1] + 1, prev[jj - 1] + (a[i - 1] != b[jj - 1]))
            prev = cur
        return prev[-1]

    rng = random.Random(7)
    checked = 0
    for _ in range(60000):
        L, r = rng.randint(12, 27), rng.randint(1, 7)
        lay = _lib.levpart_layout(L, r)
        if lay is None:
            continue
        a = [rng.choice("ACGT") for _ in range(L)]
        s = list(a)
        for _e in range(rng.randint(0, r)):
            t = rng.random()
            if t < 0.34:
                s[rng.randrange(len(s))] = rng.choice("ACGT")
            elif t < 0.67:
                del s[rng.randrange(len(s))]
            else:
                s.insert(rng.randrange(len(s) + 1), rng.choice("ACGT"))
        if len(s) != L or edit_distance(a, s) > r:
            continue
        checked += 1
        pos, hit = 0, False
        for b, ds in zip(lay["blocks"], copies(L, r, lay["blocks"])):
            hit = hit or any(a[pos:pos + b] == s[pos + d:pos + d + b] for d in ds)
            pos += b
        assert hit, (L, r, "".join(a), "".join(s), lay)
    assert checked > 10000


def test_controls_fail_fast_estimate():
    """get_control_seqs gives up after the first draw when the exact nearest-distance histogram rules the largest draw out
    (core._controls_expected_at; the reference's loop, core.py:586-620, would draw 1.1 10^4 n sequences first)."""
    rng = np.random.default_rng(0)
    # E. coli sized pool: random 20-mers sit 3 .. 6 mismatches from their nearest target; MINIMUM_HMDIST 7 is out of reach
    d1 = rng.choice([3, 4, 5, 6], p=[0.02, 0.45, 0.5, 0.03], size=100)
    assert core._controls_expected_at(d1, 7, 20, 10 * 10000, 599_093) < 1e-6
    d1 = rng.choice([3, 4, 5, 6], p=[0.02, 0.45, 0.5, 0.03], size=10_000)
    assert core._controls_expected_at(d1, 7, 20, 1000 * 10000, 599_093) < 1.0
    assert core._controls_expected_at(d1, 5, 20, 1000 * 10000, 599_093) > 1e6            # a reachable minimum: direct estimate
    assert abs(core._controls_expected_at(d1, 6, 20, 10_000, 599_093) - (d1 >= 6).sum()) < 1e-9
    # Carsonella sized pool: the nearest target is 6 .. 9 away, nothing to give up on
    d1 = rng.choice([6, 7, 8, 9], p=[0.1, 0.5, 0.35, 0.05], size=100)
    assert core._controls_expected_at(d1, 7, 20, 100_000, 3_814) > 50_000
    # a GC-rich pool (66 % GC, as PAO1) with controls of the same composition: positions match with
    # probability ~0.28, not 0.25, lambda grows 2.6 x per step instead of 3 x, and the uniform-base extrapolation over two
    # steps puts the expected controls an order of magnitude too low -- enough to give up on a reachable minimum
    L, q = 20, 1.0 - (2 * 0.33 ** 2 + 2 * 0.17 ** 2)
    from math import comb
    cdf = [sum(comb(L, i) * q ** i * (1 - q) ** (L - i) for i in range(r + 1)) for r in range(L + 1)]
    n_db = 1_160_000
    p_ge = [1.0] + [float(np.exp(-n_db * cdf[s - 1])) for s in range(1, L + 1)]      # P(nearest >= s) under the skewed model
    pmf = np.array([p_ge[s] - p_ge[s + 1] for s in range(L)] + [p_ge[L]])
    d1 = np.random.default_rng(5).choice(L + 1, p=pmf / pmf.sum(), size=100_000)
    true_expect = 1e7 * p_ge[7]
    uniform = core._controls_expected_at(d1, 7, L, 10_000_000, n_db)
    measured = core._controls_expected_at(d1, 7, L, 10_000_000, n_db, p_match=1.0 - q)
    assert abs(np.log(measured / true_expect)) < np.log(3.0), (measured, true_expect)
    assert uniform < measured / 5.0, (uniform, measured)
    # no data at all / every control an exact hit
    assert core._controls_expected_at(np.zeros(0, int), 7, 20, 1000, 10) == 1000.0
    assert core._controls_expected_at(np.zeros(50, int), 7, 20, 1000, 10) == 0.0


def test_planner_prices_short_rows_by_a_deeper_level():
    """The planner with a measured sample (host arithmetic, no GPU): per sampled candidate the pool targets within
    d = 0 .. 8.  On a real-genome-like spread of the k-th neighbour's distance (P. aeruginosa PAO1, k = 10: 3: 3 %,
    4: 37 %, 5: 49 %, 6: 10 %, 7: 1 %) a single K1P pass has to run the radius nearly every row needs (6); with deeper
    levels allowed the first level runs the radius the majority needs (5), the rows it leaves short are priced as a
    candidate set of their own, and the modelled cost drops.  On a uniform pool (the k-th neighbour at 6 for 99 % of the
    rows) nothing changes."""
    from guidemaker_amd import _lib

    def sample(kth_probs, rows=1024, growth=4.6, seed=0):
        rng = np.random.default_rng(seed)
        ks = rng.choice(len(kth_probs), size=rows, p=np.array(kth_probs) / sum(kth_probs))
        cum = np.zeros((rows, 9), np.uint32)
        for i, t in enumerate(ks):
            cum[i] = [max(1, int(round(10 * growth ** (d - t)))) for d in range(9)]
        return cum

    def radius(plan):
        return sum(t + 1 for _, t in plan["blocks"]) - 1

    pao = sample([0, 0, 0, .027, .373, .49, .098, .012, 0])
    one = _lib.pigeon_plan_sample(1_159_446, 1_171_798, 20, 10, pao, 0.2816, levels=1)
    three = _lib.pigeon_plan_sample(1_159_446, 1_171_798, 20, 10, pao, 0.2816, levels=3)
    assert one["cascade"] == 0 and radius(one) == 6
    assert three["cascade"] == 1 and radius(three) == 5 and three["cost"] < 0.7 * one["cost"]
    assert sum(l for l, _ in three["blocks"]) == 20 and sum(l for l, _ in one["blocks"]) == 20
    # the rows the first level leaves (k-th neighbour beyond 5) as a candidate set of their own: a wider radius, fewer buckets
    kth = np.array([int(np.argmax(c >= 10)) if (c >= 10).any() else 9 for c in pao])
    sub = _lib.pigeon_plan_sample(1_159_446, int(1_171_798 * (kth > 5).mean()), 20, 10, pao[kth > 5], 0.2816, levels=2)
    assert radius(sub) >= 6 and max(l for l, _ in sub["blocks"]) <= 6
    uni = sample([0, 0, 0, 0, 0, .002, .987, .011, 0])
    for levels in (1, 3):
        p = _lib.pigeon_plan_sample(599_093, 599_093, 20, 10, uni, 0.25, levels=levels)
        assert p["blocks"] == [(6, 1), (5, 1), (5, 1), (4, 0)] and p["cascade"] == 0
    # the share-only form (no hit counts): same first-level choice on the real-genome-like spread
    ps = np.ones(21)
    ps[:9] = [(kth > r).mean() for r in range(9)]
    ps[9:] = ps[8]
    d3 = _lib.pigeon_plan_dist(1_159_446, 1_171_798, 20, 10, ps, 0.2816, levels=3)
    assert d3["cascade"] == 1 and radius(d3) == 5
    assert _lib.pigeon_plan_dist(1_159_446, 1_171_798, 20, 10, ps, 0.2816, levels=1)["cascade"] == 0


def test_format_neighbor_columns_matches_reference_closures(oracle):
    """the vectorised 'Similar guides' / 'Similar guide distances' columns are byte-identical to
    the reference's per-row closures (core.py:913-931, restated in the oracle), including rows with fewer than k
    hits, two-digit distances and duplicate candidate strings (the last occurrence's row wins, as with the dict)."""
    from helpers import random_pool
    L = 20
    db = list(dict.fromkeys(random_pool(300, L, seed=3)))
    q = db[:150] + db[:7]
    for k, pool in ((2, db), (10, db), (20, db), (10, db[:6])):      # the last: k larger than the index
        ids, d, c = oracle.knn(pool, q, L, "hamming", k)
        d = np.where(d >= 0, d // 2, -1)
        keep = np.flatnonzero(d[:, 1] >= 2)
        nm = core.NeighborMap(q, keep, ids, d, c, np.asarray(pool, dtype=object))

        class TL:
            neighbors = nm
        keys = list(nm.keys())
        got = core.format_neighbor_columns(TL, keys + keys[:3])
        want = oracle.format_neighbor_columns({kk: nm[kk] for kk in keys}, keys + keys[:3])
        assert got[0] == want[0] and got[1] == want[1]
        tab = core.neighbor_table(TL)
        assert tab["Guide sequence"].tolist() == keys and tab["Similar guides"].tolist() == want[0][:len(keys)]
    with pytest.raises(KeyError):
        core.format_neighbor_columns(TL, ["A" * L])
    # a plain dict goes through the reference's own loops
    class TD:
        neighbors = {"ACGT": {"target": "ACGT", "neighbors": {"seqs": ["ACGT", "ACGA"], "dist": [0, 1]}}}
    assert core.format_neighbor_columns(TD, ["ACGT"]) == (["ACGT;ACGA"], ["0;1"])


def test_pack_rejects_mixed_lengths_that_sum_right():
    # lengths 19 + 21 sum to 2 * 20 and first / last rows alone would not always catch it
    with pytest.raises(ValueError):
        _lib.pack(["A" * 20, "C" * 19, "G" * 21, "T" * 20], 20)
    assert _lib.pack(["ACGT" * 5, "TTTT" * 5], 20).shape == (2,)


def test_packed_side_car_travels_with_the_frame():
    """find_targets keeps the 2-bit packed 'target' column for the frame (a string token in frame.attrs, the array in
    the module's cache: arrays in attrs break pd.concat); TargetProcessor uses it instead of re-encoding the strings,
    checks it against the strings, and falls back when it does not fit."""
    import pandas as pd
    recs = carsonella()
    df = core.PamTarget("NGG", "5prime", "hamming").find_targets(recs, 20)
    side = core.packed_side_car(df)
    assert all(isinstance(v, (str, int)) for v in df.attrs.values())      # nothing in attrs that `==` chokes on
    both = pd.concat([df, df], ignore_index=True)                        # (raised ValueError with an array in attrs)
    assert len(both) == 2 * len(df)
    assert df.attrs["gm_guidelen"] == 20 and side.dtype == np.uint64 and len(side) == len(df)
    assert (side == _lib.pack(df["target"].tolist(), 20)).all()
    tp = core.TargetProcessor(df, lsr=10)
    pk = tp._packed_targets()
    assert pk[1] == 20 and (pk[0] == side).all()
    assert tp._packed_targets() is pk                                  # cached
    tp.find_unique_near_pam()                                          # deep-copies the frame: cache carried over
    assert tp._packed_targets()[0] is pk[0]
    # a frame whose rows were filtered loses the alignment: the side-car is ignored, the strings are packed
    sub = df[df["strand"]].reset_index(drop=True)
    sub.attrs["gm_packed_token"] = df.attrs["gm_packed_token"]         # stale on purpose
    tp2 = core.TargetProcessor(sub, lsr=10)
    assert (tp2._packed_targets()[0] == _lib.pack(sub["target"].tolist(), 20)).all()
    # same length but different content: caught by the probe rows
    wrong = df.copy()
    wrong.attrs["gm_packed_token"] = core._packed_side_car_put(side[::-1].copy())
    tp3 = core.TargetProcessor(wrong, lsr=10)
    assert (tp3._packed_targets()[0] == side).all()
    # the concatenated frame above carries a token whose side-car has another length: ignored
    tp4 = core.TargetProcessor(both, lsr=10)
    assert (tp4._packed_targets()[0] == np.concatenate([side, side])).all()


def test_real_genome_hook(tmp_path, monkeypatch):
    """a real genome file named by GM_GENOME_<WORKLOAD> replaces the synthetic stand-in.  The reader
    restates the reference's input step (core.py:1065-1090: FASTA or GenBank, plain or gzipped, upper-cased)."""
    import gzip
    from guidemaker_amd import workloads
    recs = carsonella()
    fasta = tmp_path / "g.fasta.gz"
    with gzip.open(fasta, "wt") as fh:
        for r in recs:
            fh.write(">%s some description\n" % r.id)
            s = str(r.seq).lower()
            for i in range(0, len(s), 70):
                fh.write(s[i:i + 70] + "\n")
    got = workloads.read_genome(str(fasta))
    assert [g[0] for g in got] == [r.id for r in recs] and [g[1] for g in got] == [str(r.seq).upper() for r in recs]
    gbk = tmp_path / "g.gbk"
    with open(gbk, "w") as fh:
        for r in recs:
            s = str(r.seq).lower()
            fh.write("LOCUS       %s %d bp DNA\nVERSION     %s.1\nFEATURES             Location/Qualifiers\nORIGIN\n" % (r.id, len(s), r.id))
            for i in range(0, len(s), 60):
                row = s[i:i + 60]
                fh.write("%9d %s\n" % (i + 1, " ".join(row[j:j + 10] for j in range(0, len(row), 10))))
            fh.write("//\n")
    got2 = workloads.read_genome(str(gbk))
    assert [g[1] for g in got2] == [g[1] for g in got] and got2[0][0] == recs[0].id + ".1"
    monkeypatch.setenv("GM_GENOME_TINY", str(fasta))
    pool = workloads.make_pool("tiny")
    assert pool["real_genome"] == str(fasta) and "REAL genome" in pool["label"]
    df = core.PamTarget("NGG", "3prime", "hamming").find_targets(recs, 20)
    assert (pool["packed_all"] == _lib.pack(df["target"].tolist(), 20)).all()
    monkeypatch.delenv("GM_GENOME_TINY")
    assert workloads.make_pool("tiny")["real_genome"] is None


def test_optional_gpu_config_block(tmp_path, monkeypatch):
    """The optional GPU: block of the config selects the device / sharding unless the constructor
    said otherwise; config files without it (the reference's) keep working."""
    import yaml
    from helpers import targets_frame
    monkeypatch.delenv("GM_SHARD_CANDIDATES", raising=False)
    with open(guidemaker_amd.CONFIG_PATH) as fh:
        cfg = yaml.safe_load(fh)
    assert cfg["GPU"] == {"DEVICE": 0, "SHARD_CANDIDATES": False, "CONTROL_FAIL_FAST": False}
    calls = {}

    class FakeIndex:
        def __init__(self, packed, L, metric, device):
            calls["device"] = device

        def close(self):
            pass
    monkeypatch.setattr(_lib, "Index", FakeIndex)
    cfg["GPU"] = {"DEVICE": 3, "SHARD_CANDIDATES": True}
    p = tmp_path / "c.yaml"
    p.write_text(yaml.safe_dump(cfg))
    tp = core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGA"]), lsr=0)
    tp.create_index(str(p))
    assert calls["device"] == 3 and tp.device == 3 and tp.shard_candidates is True
    tp = core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0, device=1, shard_candidates=False)
    tp.create_index(str(p))
    assert calls["device"] == 1 and tp.shard_candidates is False
    del cfg["GPU"]
    p.write_text(yaml.safe_dump(cfg))
    tp = core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0)
    tp.create_index(str(p))
    assert calls["device"] == 0


def test_format_doubles_is_pythons_str_of_float():
    """gm_format_doubles (the 'CFD Similar Guides' strings of core.cfd_score) against str(float): random values over
    every magnitude, the notation switch points, zeros, denormals, infinities."""
    from guidemaker_amd import _lib
    rng = np.random.default_rng(12)
    vals = np.concatenate([rng.random(300_000), rng.random(50_000) * 1e-5, 10.0 ** rng.uniform(-300, 300, size=100_000),
                           -(10.0 ** rng.uniform(-20, 20, size=20_000)), np.round(rng.random(50_000), 3),
                           np.array([0.0, -0.0, 1.0, 0.5, 1e16, 1e15, 9999999999999998.0, 123456789012345678.0, 1e-4, 1e-5,
                                     0.0001234, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, float("inf"),
                                     float("-inf"), float("nan"), 0.176470588, 1e22, 1e21, 0.1 + 0.2, 1 / 3])])
    got = _lib.format_doubles(vals)
    want = [str(x) for x in vals.tolist()]
    assert got == want
    assert _lib.format_doubles(np.array([0.25])) == ["0.25"] and _lib.format_doubles(np.array([])) == []


def test_native_neighbor_columns_equal_the_numpy_path():
    """gm_format_neighbor_rows (used when the index's targets are at hand in packed form) against the numpy path of
    core.format_neighbor_columns, ragged rows (fewer than k hits, none at all) and two-digit distances included."""
    from guidemaker_amd import _lib
    rng = np.random.default_rng(4)
    for L, k, n_db, n in ((20, 10, 5000, 3000), (27, 3, 40, 500), (10, 20, 300, 70_000)):
        db_packed = rng.integers(0, 1 << (2 * L), size=n_db, dtype=np.uint64)
        db = np.asarray(_lib.unpack(db_packed, L), dtype=object)
        q = ["".join(rng.choice(list("ACGT"), size=L)) + str(i) for i in range(n)]      # distinct keys
        ids = rng.integers(0, n_db, size=(n, k)).astype(np.int32)
        dists = rng.integers(0, 28, size=(n, k)).astype(np.int32)
        counts = rng.integers(0, k + 1, size=n).astype(np.int32)
        counts[: n // 2] = k
        for r in range(n):
            ids[r, counts[r]:] = -1
            dists[r, counts[r]:] = -1

        class TP:
            pass
        tp = TP()
        tp.neighbors = core.NeighborMap(q, np.arange(n), ids, dists, counts, db, q_packed=np.zeros(n, np.uint64),
                                        db_packed=db_packed, guidelen=L)
        got = core.format_neighbor_columns(tp, q)
        tp.neighbors = core.NeighborMap(q, np.arange(n), ids, dists, counts, db)
        want = core.format_neighbor_columns(tp, q)
        assert got[0] == want[0] and got[1] == want[1]


def test_release_library_has_no_wrong_result_knob():
    """gm_set_param("pg_debug") made the scan skip work (results WRONG) in the shipped library.  It is
    now a compile-time option of separate timing builds (-DGM_PG_TIMING); the release .so answers GM_ERR_ARG -- checked
    on the exported function without a GPU (the parameter is rejected before any device call)."""
    import ctypes
    from guidemaker_amd import _lib
    lib = _lib.lib()
    lib.gm_set_param.restype = ctypes.c_int
    lib.gm_set_param.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    # a null handle is rejected first; with a handle the name is: build one without touching a device is not possible, so
    # the source is checked as well
    assert lib.gm_set_param(None, b"pg_debug", 1) != 0
    import os
    src = open(os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "gm_pigeon.hip")).read()
    assert 'std::strcmp(name, "pg_debug") == 0)' in src and "return gm_fail(GM_ERR_ARG, \"pg_debug: not in this build" in src
    assert "tune_debug" not in src
    hdr = open(os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "gm_pg_types.h")).read()
    assert "debug" not in hdr
