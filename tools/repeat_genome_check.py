#!/usr/bin/env python3
"""K1P on a genome with repeat families (rRNA-operon / IS-element like): 4.6 Mb random genome in which
--families segments of --seglen bp occur --copies times each with --div per-base divergence.  Checks that
the default scan equals the plain scan (variant 8) and reports the re-scanned rows and the timings.
   python tools/repeat_genome_check.py"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, core, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4_641_652)
    ap.add_argument("--families", type=int, default=12)
    ap.add_argument("--copies", type=int, default=25)
    ap.add_argument("--seglen", type=int, default=5000)
    ap.add_argument("--div", type=float, default=0.01)
    ap.add_argument("--k", type=int, default=10)
    a = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(99))
    codes = workloads.synth_genome(a.size, 0.508, rng)
    for _ in range(a.families):
        src = rng.integers(0, a.size - a.seglen)
        seg = codes[src:src + a.seglen].copy()
        for _ in range(a.copies):
            dst = rng.integers(0, a.size - a.seglen)
            cp = seg.copy()
            mut = rng.random(a.seglen) < a.div
            cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
            codes[dst:dst + a.seglen] = cp
    seq = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode()
    res = core.scan_targets_gpu(seq, "NGG", "3prime", 20)
    allp = res["packed"]
    db = allp[np.sort(np.unique(allp, return_index=True)[1])]      # first-occurrence order
    print("targets %d, unique %d" % (allp.shape[0], db.shape[0]))
    idx = _lib.Index(db, 20, _lib.METRIC_HAMMING)
    out = {}
    for v in (0, 8, 0, 8):
        idx.set_variant(v)
        out[v] = idx.knn(allp, a.k)
        name, ms, pairs = idx.last_kernel()
        ev, inc = idx.last_evaluated()
        print("variant %d: %-28s query %.3f ms, evaluated %.1f %%, re-scanned rows %d"
              % (v, name, idx.last_query_ms(), 100.0 * ev / max(pairs, 1), inc))
    assert all((x == y).all() for x, y in zip(out[0], out[8])), "scans differ"
    d = out[0][1]
    print("identical results; rows with a 2nd neighbour at distance <= 1: %d" % int((d[:, 1] <= 1).sum()))
    idx.close()


if __name__ == "__main__":
    main()
