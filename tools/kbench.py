#!/usr/bin/env python3
"""Kernel micro-benchmark: random pools, variants A/B-interleaved in ONE process (HIP-event times).

  python tools/kbench.py --n 600000 --k 10 --variants 0,1,2 --rounds 5 [--metric leven] [--nearest]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, core  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=600000)
    ap.add_argument("--nq", type=int, default=0)
    ap.add_argument("--L", type=int, default=20)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="hamming")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--nearest", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare every variant's output with variant 0")
    ap.add_argument("--params", default="", help="gm_set_param sweeps, e.g. pg_sub=2:4:8,pg_run=16:256 (all "
                    "combinations, each with variant 0)")
    ap.add_argument("--workload", default="", help="use guidemaker_amd.workloads.make_pool(NAME) instead of a random pool")
    ap.add_argument("--planted", type=float, default=0.0,
                    help="fraction of pool entries replaced by near-duplicates (1-3 substitutions) of others")
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 4, size=(a.n, a.L), dtype=np.uint8)
    if a.planted > 0:      # SURVEY.md 8(d) "direct pool mode": planted near-duplicates exercise thresholds and ties
        m = int(a.n * a.planted)
        dst, src = rng.integers(0, a.n, size=m), rng.integers(0, a.n, size=m)
        codes[dst] = codes[src]
        for _ in range(3):
            hit = rng.random(m) < 0.67
            codes[dst[hit], rng.integers(0, a.L, size=int(hit.sum()))] = rng.integers(0, 4, size=int(hit.sum()))
    db = np.unique(core.pack_codes(codes))
    rng.shuffle(db)
    q = db if a.nq == 0 else db[:a.nq]
    if a.workload:
        from guidemaker_amd import workloads
        pool = workloads.make_pool(a.workload)
        db, q, a.L, a.k, a.metric = pool["packed_db"], pool["packed_all"], pool["guidelen"], pool["k"], pool["metric"]
    idx = _lib.Index(db, a.L, _lib.METRIC_HAMMING if a.metric == "hamming" else _lib.METRIC_LEVEN)
    variants = [int(v) for v in a.variants.split(",")]
    if a.params:      # parameter sweep: label = index into the list of combinations
        import itertools
        keys, vals = [], []
        for item in a.params.split(","):
            kname, vs = item.split("=")
            keys.append(kname)
            vals.append([int(x) for x in vs.split(":")])
        combos = list(itertools.product(*vals))
        variants = list(range(len(combos)))
    times = {v: [] for v in variants}
    ktimes = {v: [] for v in variants}
    names = {}
    ref = None
    for r in range(a.rounds + 1):
        for v in variants:
            if a.params:
                for kname, val in zip(keys, combos[v]):
                    idx.set_param(kname, val)
            else:
                idx.set_variant(v)
            if a.nearest:
                out = (idx.nearest(q),)
            else:
                out = idx.knn(q, a.k)
            name, ms, pairs = idx.last_kernel()
            ev, inc = idx.last_evaluated()
            names[v] = name if ev == pairs else "%s[eval %.1f%%, %d re-scanned]" % (name, 100.0 * ev / max(pairs, 1), inc)
            if r > 0:
                times[v].append(idx.last_query_ms())     # whole query: all phases + merge, device time
                ktimes[v].append(ms)
            if a.check:
                if ref is None:
                    ref = out
                else:
                    assert all((x == y).all() for x, y in zip(ref, out)), "variant %d differs" % v
    for v in variants:
        if a.params:
            names[v] += " " + " ".join("%s=%d" % kv for kv in zip(keys, combos[v]))
        t = np.array(times[v])
        print("variant %3d %-40s n_db=%d n_q=%d k=%d: query median %.3f ms min %.3f ms (dominant kernel %.3f ms)"
              "  -> %.3e pairs/s"
              % (v, names[v], len(db), len(q), a.k, np.median(t), t.min(), np.median(ktimes[v]),
                 len(db) * len(q) / (np.median(t) * 1e-3)))


if __name__ == "__main__":
    main()
