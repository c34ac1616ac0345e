#!/usr/bin/env python3
"""get_control_seqs at genome scale: wall clock, and equality of the default scan (K1P where the planner
picks it) with the plain two-phase scan (variant 8) on the same random draws.
   python tools/controls_check.py --workload s1 --n 1000"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import guidemaker_amd  # noqa: E402
from guidemaker_amd import core, workloads  # noqa: E402
from e2e import Rec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="s1")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--min-dist", type=int, default=5)
    a = ap.parse_args()
    w = workloads.WORKLOADS[a.workload]
    rng = np.random.Generator(np.random.PCG64(w["seed"]))
    recs = []
    if "fixture" in w:      # a real genome (tests/golden/*.2bit.gz)
        recs.append(Rec(workloads.read_2bit(workloads.fixture_path(w["fixture"]), w["records"][0]), a.workload))
    else:
        for i, n in enumerate(w["records"]):
            codes = workloads.synth_genome(n, w["gc"], rng)
            recs.append(Rec(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode(), "rec%d" % i))
    pt = core.PamTarget(w["pam"], w["orientation"], w["metric"])
    tp = core.TargetProcessor(pt.find_targets(recs, w["guidelen"]), lsr=10, editdist=w["editdist"], knum=w["k"])
    tp.check_restriction_enzymes(["NRAGCA"])
    tp.find_unique_near_pam()
    tp.create_index(guidemaker_amd.CONFIG_PATH)
    # With an exact search a 20-mer at distance >= 7 (the shipped MINIMUM_HMDIST) from EVERY target of a
    # 4.6 Mb genome does not exist; the reference's loop then walks off CONTROL_SEARCH_MULTIPLE and raises
    # IndexError (core.py:586-617, mirrored).  Use a reachable minimum here.
    import tempfile
    import yaml
    with open(guidemaker_amd.CONFIG_PATH) as fh:
        cfg = yaml.safe_load(fh)
    cfg["CONTROL"]["MINIMUM_HMDIST"] = a.min_dist
    tmp = tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False)
    yaml.safe_dump(cfg, tmp)
    tmp.close()
    out = {}
    for variant in (0, 8):
        tp.nmslib_index.set_variant(variant)
        np.random.seed(12345)
        t0 = time.perf_counter()
        res = tp.get_control_seqs(recs, tmp.name, length=w["guidelen"], n=a.n)
        dt = time.perf_counter() - t0
        out[variant] = res
        print("variant %d: %.3f s, min %.1f median %.1f, searched %d, last kernel %s"
              % (variant, dt, res[0], res[1], tp.ncontrolsearched, tp.nmslib_index.last_kernel()[0]))
    a0, a8 = out[0][2], out[8][2]
    assert a0.equals(a8), "control tables differ between the scans"
    print("control tables identical (%d rows)" % len(a0))


if __name__ == "__main__":
    main()
