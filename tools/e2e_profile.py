#!/usr/bin/env python3
"""cProfile of the host side of the mirrored flow (tools/e2e.py) -- where the wall clock goes around the
GPU scans.   python tools/e2e_profile.py --workload s1 [--stage get_neighbors]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import guidemaker_amd  # noqa: E402
from guidemaker_amd import core, workloads  # noqa: E402
from e2e import Rec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="s1")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--output", action="store_true", help="include the output columns and cfd_score")
    ap.add_argument("--sort", default="cumulative")
    a = ap.parse_args()
    w = workloads.WORKLOADS[a.workload]
    rng = np.random.Generator(np.random.PCG64(w["seed"]))
    recs = []
    for i, n in enumerate(w["records"]):
        codes = workloads.synth_genome(n, w["gc"], rng)
        recs.append(Rec(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode(), "rec%d" % i))
    # warm the GPU context so the profile shows steady-state host cost
    core.PamTarget(w["pam"], w["orientation"], w["metric"]).find_targets(recs[:1], w["guidelen"])

    def flow():
        pt = core.PamTarget(w["pam"], w["orientation"], w["metric"])
        df = pt.find_targets(recs, w["guidelen"])
        tp = core.TargetProcessor(df, lsr=10, editdist=w["editdist"], knum=w["k"])
        tp.check_restriction_enzymes(["NRAGCA"])
        tp.find_unique_near_pam()
        tp.create_index(guidemaker_amd.CONFIG_PATH)
        tp.get_neighbors(guidemaker_amd.CONFIG_PATH)
        if a.output:
            frame = core.neighbor_table(tp)
            core.cfd_score(frame)
        return tp

    pr = cProfile.Profile()
    pr.enable()
    flow()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats(a.sort).print_stats(a.top)


if __name__ == "__main__":
    main()
