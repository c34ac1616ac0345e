#!/usr/bin/env python3
"""Levenshtein k-NN on a LARGE real pool: the PAO1 fixture's NGG targets cut at L = 23 (1.17 M targets -- ten times the
BASELINE Levenshtein configuration), two-pass scan against the single pass, plus an oracle sample.
   python tools/real_genome_leven.py [--rows 200000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, core, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=0, help="candidates (0 = all)")
    ap.add_argument("--oracle-rows", type=int, default=64)
    a = ap.parse_args()
    w = workloads.WORKLOADS["pao1"]
    seq = workloads.read_2bit(workloads.fixture_path(w["fixture"]), w["records"][0])
    L, k = 23, 5
    packed = core.pack_codes(core.scan_targets(seq, "NGG", "3prime", L)["codes"])
    db = packed[np.sort(np.unique(packed, return_index=True)[1])]
    q = packed if a.rows <= 0 else packed[:: max(1, packed.shape[0] // a.rows)]
    print("pool %d, candidates %d, L=%d k=%d" % (db.shape[0], q.shape[0], L, k), flush=True)
    idx = _lib.Index(db, L, _lib.METRIC_LEVEN)
    try:
        got = None
        # K2P (the displaced-block partition, radius from a sample) -> the two-pass K2B scan -> the plain single pass
        for label, lpr, lpr2, two in (("K2P", 0, 0, 1), ("K2P one level", 0, -1, 1), ("K2B two-pass", -1, -1, 1),
                                      ("K2B single pass", -1, -1, 0)):
            idx.set_param("lp_radius", lpr)
            idx.set_param("lp_radius2", lpr2)
            idx.set_param("lv_two_pass", two)
            idx.knn(q[:70000], k)      # warm
            idx.knn(q, k)              # (first full query: plan, pool copies)
            t0 = time.perf_counter()
            res = idx.knn(q, k)
            dt = time.perf_counter() - t0
            print("%-16s %.3f s wall, device query %.1f ms, kernel %s (%.1f ms), plain second-pass rows %d, %.3g pairs/s, K2P levels %s"
                  % (label + ":", dt, idx.last_query_ms(), idx.last_kernel()[0], idx.last_kernel()[1], idx.last_evaluated()[1],
                     q.shape[0] * db.shape[0] / (idx.last_query_ms() * 1e-3), idx.levpart_info()), flush=True)
            if got is None:
                got = res
            else:
                assert all((x == y).all() for x, y in zip(got, res)), "%s differs from K2P" % label
    finally:
        idx.close()
    if a.oracle_rows <= 0:
        print("identical to the two-pass and the single-pass K2B scan")
        return
    from oracle import oracle as orc
    orc.build()
    rows = np.sort(np.random.default_rng(3).choice(q.shape[0], size=a.oracle_rows, replace=False))
    asc = np.frombuffer(b"ACGT", np.uint8)
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    db_b = np.ascontiguousarray(asc[((db[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    q_b = np.ascontiguousarray(asc[((q[rows][:, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
    t0 = time.perf_counter()
    oi, od, oc = orc.knn(db_b, q_b, L, "leven", k)
    print("oracle: %d rows in %.1f s" % (rows.size, time.perf_counter() - t0))
    assert (oc == got[2][rows]).all() and (od == got[1][rows]).all() and (oi == got[0][rows]).all()
    print("identical to the two-pass scan, the single pass and the oracle sample")


if __name__ == "__main__":
    main()
