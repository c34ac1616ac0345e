#!/usr/bin/env python3
"""Randomised differential run of the pigeonhole-partitioned scan at pool sizes where the planner, the prefix re-scan and the
id cut-off are all live: random pool size / guide length / k / base skew / planted near-duplicates / cut-off setting, the
whole answer against the plain scan (variant 8) of the same library.   python tools/k1p_stress.py --seconds 180"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, core  # noqa: E402


def pool(rng, n, L, skew, planted):
    p = np.array([1 + skew, 1 - skew, 1 - skew, 1 + skew], dtype=np.float64)
    codes = rng.choice(4, size=(n, L), p=p / p.sum()).astype(np.uint8)
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


def rand_layout(rng, L, r_lo, r_hi):
    """A random K1P layout for guides of L bases whose radius lies in [r_lo, r_hi], as gm_set_param "pg_layout" digits
    (len, radius per block; at most four blocks: the parameter is a 32-bit int) -- or 0 when none turned up."""
    for _ in range(200):
        B = int(rng.integers(3, 5)) if L <= 24 else 4
        cuts = np.sort(rng.choice(np.arange(1, L), size=B - 1, replace=False))
        lens = np.diff(np.concatenate([[0], cuts, [L]]))
        if lens.min() < 3 or lens.max() > 8:
            continue
        rad = rng.choice([0, 1, 1, 2], size=B)
        rad[lens < 4] = np.minimum(rad[lens < 4], 1)
        order = np.argsort(-rad, kind="stable")      # (larger radii first, as the planner's layouts)
        lens, rad = lens[order], rad[order]
        r = int(rad.sum() + B - 1)
        if not (r_lo <= r <= r_hi) or (rad == 2).sum() > 2:
            continue
        if sum((1 + 3 * l * (t >= 1) + 9 * (l * (l - 1) // 2) * (t >= 2) + 7) // 8 for l, t in zip(lens, rad)) > 64:
            continue
        code = 0
        for l, t in zip(lens, rad):
            code = code * 100 + int(l) * 10 + int(t)
        return code
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--leven", type=float, default=0.15, help="share of Levenshtein cases")
    ap.add_argument("--max-n", type=int, default=400_000, help="largest pool (>= 1.5 M brings the radius-2 layouts in)")
    ap.add_argument("--max-q", type=int, default=120_000, help="largest number of pool members among the candidates")
    ap.add_argument("--cascade", type=float, default=0.35, help="share of Hamming cases with a forced cascade of K1P levels")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0, case, used = time.time(), 0, {"k_hamming_scan_pg": 0}
    while time.time() - t0 < a.seconds:
        case += 1
        n = int(rng.integers(40_000, a.max_n))
        L = int(rng.integers(14, 25))
        k = int(rng.choice([1, 2, 3, 5, 7, 10, 13, 20]))
        skew = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
        planted = float(rng.choice([0.0, 0.01, 0.05]))
        cut = int(rng.choice([0, 0, 0, 10, 30, 60, 100]))
        db = pool(rng, n, L, skew, planted)
        q = np.concatenate([db[: int(rng.integers(1000, min(len(db), a.max_q)))], pool(rng, 3000, L, skew, 0.0)])
        if rng.random() < a.leven:      # Levenshtein: two-pass scan (narrow band first) against the single pass
            if rng.random() < 0.3:      # (guide lengths up to the maximum; the Hamming cases stay within the layouts' range)
                L = int(rng.integers(14, 28))
                db = pool(rng, n, L, skew, planted)
            db = db[: int(rng.integers(33_000, min(len(db), 90_000)))]
            q = np.concatenate([db[: int(rng.integers(4100, min(len(db), 30_000)))], pool(rng, 500, L, skew, 0.0)])
            k = int(rng.choice([2, 3, 5, 10, 13, 20]))
            idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
            try:
                lpr = int(rng.choice([0, 0, 2, 3, 4, 5]))      # K2P (the Levenshtein partition) with a forced radius
                lp2 = ""
                if lpr:
                    idx.set_param("lp_radius", lpr)
                    if rng.random() < 0.5 and L >= 2 * (lpr + 2):      # ... and a forced second level for the rows it leaves
                        r2 = int(rng.integers(lpr + 1, min(7, L // 2 - 1) + 1))
                        idx.set_param("lp_radius2", r2)
                        lp2 = " K2P %d -> %d" % (lpr, r2)
                got = idx.knn(q, k)
                if lp2:
                    lp2 += " (%d rows, %d left)" % (idx.levpart_info()["rows2"], idx.levpart_info()["left2"])
                name, resc = idx.last_kernel()[0], idx.last_evaluated()[1]
                idx.set_param("lp_radius", -1)
                idx.set_param("lv_two_pass", 0)
                want = idx.knn(q, k)
            finally:
                idx.close()
            used[name] = used.get(name, 0) + 1
            ok = all((x == y).all() for x, y in zip(got, want))
            print("case %3d leven n_db=%d n_q=%d L=%d k=%d skew=%.1f planted=%.2f second pass %d rows%s %s" %
                  (case, len(db), len(q), L, k, skew, planted, resc, lp2, "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                sys.exit(1)
            continue
        idx = _lib.Index(db, L, _lib.METRIC_HAMMING)
        try:
            idx.set_param("pg_cut", cut)
            hit = int(rng.choice([0, 1, 2, 2]))      # hit buffer / chosen per level / register lists on every level
            idx.set_param("pg_hit", hit)
            idx.set_param("pg_group", int(rng.choice([0, 1, 2])))      # candidate grouping: chosen / one level / two levels
            idx.set_param("pg_slots", int(rng.choice([0, 1, 2])))      # the merge's list of incomplete rows: chosen / atomics / slots
            if rng.random() < 0.2:      # (default: the planner scans a sample of the candidates when there are >= 32768)
                idx.set_param("pg_sample", 0)
            casc = ""
            if rng.random() < a.cascade:      # a first level that leaves many rows short + forced deeper levels
                lay1, lay2 = rand_layout(rng, L, 2, 4), rand_layout(rng, L, 4, 8)
                if lay1 and lay2:
                    idx.set_param("pg_layout", lay1)
                    idx.set_param("pg_layout2", lay2)
                    idx.set_param("pg_cascade", 2)
                    casc = " cascade %d -> %d" % (lay1, lay2)
            if k == 1 and rng.random() < 0.5:
                got = (idx.nearest(q),)
                name = idx.last_kernel()[0]
                idx.set_variant(8)
                want = (idx.nearest(q),)
            else:
                got = idx.knn(q, k)
                name = idx.last_kernel()[0]
                levels = idx.cascade_info()
                if casc:
                    casc += " levels " + "/".join("%d@r%d" % (lv["n_q"], lv["radius"]) for lv in levels)
                    used["cascade levels >= 2"] = used.get("cascade levels >= 2", 0) + (len(levels) >= 2)
                idx.set_variant(8)
                want = idx.knn(q, k)
            resc = idx.last_evaluated()[1]
        finally:
            idx.close()
        used[name] = used.get(name, 0) + 1
        ok = all((x == y).all() for x, y in zip(got, want))
        print("case %3d n_db=%d n_q=%d L=%d k=%d skew=%.1f planted=%.2f cut=%d hit=%d %s%s %s" % (case, len(db), len(q), L, k, skew, planted, cut, hit, name, casc, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            sys.exit(1)
    print("cases", case, used)


if __name__ == "__main__":
    main()
