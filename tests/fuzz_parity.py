#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the oracle (test infrastructure, like tests/):
random guide length, metric, k, pool / candidate sizes (including tiny and tile-boundary sizes), duplicate
rate, base skew, and -- for Hamming -- a random forced K1P layout (plain or expansion).
   python tests/fuzz_parity.py --seconds 120 [--seed 1]"""
import argparse
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def run(seconds, seed, max_cases=None):
    """Runs random cases for `seconds` (or max_cases); returns (cases run, description of the first mismatch or None)."""
    orc.build()
    rng = random.Random(seed)
    t_end = time.time() + seconds
    n_cases = 0
    while time.time() < t_end and (max_cases is None or n_cases < max_cases):
        L = rng.randint(10, 27)
        metric = rng.choice(["hamming", "hamming", "leven"])
        n_db = rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 511, 513, rng.randint(3, 3000), rng.randint(3000, 9000)])
        n_q = rng.choice([1, 63, 64, 65, rng.randint(2, 700)])
        k = rng.choice([1, 2, 5, 6, 10, 11, 20, 21, rng.randint(1, 30)])
        alphabet = rng.choice(["ACGT", "AAACGT", "AAAAAACT", "ACGGGGGT"])
        base = ["".join(rng.choice(alphabet) for _ in range(L)) for _ in range(n_db)]
        dup = rng.random() * 0.5
        for i in range(n_db):      # near-duplicates
            if rng.random() < dup:
                s = list(base[rng.randrange(n_db)])
                for p in rng.sample(range(L), rng.randint(0, 4)):
                    s[p] = rng.choice("ACGT")
                base[i] = "".join(s)
        db = list(dict.fromkeys(base))
        q = [rng.choice(db) if rng.random() < 0.5 else "".join(rng.choice(alphabet) for _ in range(L))
             for _ in range(n_q)]
        variant = 0
        if metric == "hamming":
            B = rng.randint(2, 8)
            variant = rng.choice([0, 8, 100 + B, 200 + B, 300 + B, 400 + B, 600 + B, 700 + B, 800 + B, 900 + B])
        idx = _lib.Index(_lib.pack(db, L), L, _lib.METRIC_HAMMING if metric == "hamming" else _lib.METRIC_LEVEN)
        idx.set_variant(variant)
        hit = -1
        if metric == "hamming":
            hit = rng.choice([0, 1, 2])      # K1P's hit path: hit buffer / per level / register lists
            idx.set_param("pg_hit", hit)
            idx.set_param("pg_group", rng.choice([0, 1, 2]))      # candidate grouping: chosen / one level / two levels
            idx.set_param("pg_slots", rng.choice([0, 1, 2]))      # the merge's list of incomplete rows: chosen / atomics / slots
        qp = _lib.pack(q, L)
        ids, dists, counts = idx.knn(qp, k)
        d1 = idx.nearest(qp)
        kern = idx.last_kernel()[0]
        idx.close()
        oi, od, oc = orc.knn(db, q, L, metric, k)
        if metric == "hamming":
            od = np.where(od >= 0, od // 2, -1)
        ok = (ids == oi).all() and (dists == od).all() and (counts == oc).all() and (d1 == od[:, 0]).all()
        if not ok:
            what = [name for name, a, b in (("ids", ids, oi), ("dists", dists, od), ("counts", counts, oc), ("nearest", d1, od[:, 0]))
                    if not (a == b).all()]
            rows = np.flatnonzero((ids != oi).any(axis=1) | (dists != od).any(axis=1) | (d1 != od[:, 0]))[:4].tolist()
            return n_cases, ("MISMATCH: L=%d metric=%s n_db=%d n_q=%d k=%d variant=%d pg_hit=%d kernel=%s seed=%d case=%d: %s differ, rows %s; "
                             "row %s: got ids %s dists %s nearest %s, want ids %s dists %s"
                             % (L, metric, len(db), n_q, k, variant, hit, kern, seed, n_cases, what, rows, rows[:1],
                                ids[rows[:1]].tolist(), dists[rows[:1]].tolist(), d1[rows[:1]].tolist(), oi[rows[:1]].tolist(),
                                od[rows[:1]].tolist()))
        n_cases += 1
    return n_cases, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    n_cases, bad = run(a.seconds, a.seed)
    if bad:
        print(bad)
        sys.exit(1)
    print("fuzz: %d cases ok (seed %d)" % (n_cases, a.seed))


if __name__ == "__main__":
    main()
