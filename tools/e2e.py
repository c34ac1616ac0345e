#!/usr/bin/env python3
"""End-to-end wall clock of the mirrored GuideMaker flow on a synthetic genome (host + GPU):
find_targets -> check_restriction_enzymes -> find_unique_near_pam -> create_index -> get_neighbors
(-> cfd_score on the retained guides).   python tools/e2e.py --workload s1 [--cfd]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import guidemaker_amd  # noqa: E402
from guidemaker_amd import core, workloads  # noqa: E402


class Rec:
    def __init__(self, seq, id):
        self.seq, self.id = seq, id

    def __len__(self):
        return len(self.seq)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="s1")
    ap.add_argument("--cfd", action="store_true")
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
    t = {}
    # warm the GPU context first (HIP initialisation is paid once per process, not per genome)
    core.PamTarget(w["pam"], w["orientation"], w["metric"]).find_targets([Rec(recs[0].seq[:100000], "warm")], w["guidelen"])
    t0 = time.perf_counter()
    pt = core.PamTarget(w["pam"], w["orientation"], w["metric"])
    df = pt.find_targets(recs, w["guidelen"])
    t["find_targets"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    tp = core.TargetProcessor(df, lsr=10, editdist=w["editdist"], knum=w["k"])
    tp.check_restriction_enzymes(["NRAGCA"])
    tp.find_unique_near_pam()
    t["filters"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    tp.create_index(guidemaker_amd.CONFIG_PATH)
    t["create_index"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    tp.get_neighbors(guidemaker_amd.CONFIG_PATH)
    t["get_neighbors"] = time.perf_counter() - t0
    name, ms, pairs = tp.nmslib_index.last_kernel()
    print("targets %d, unique %d, queries %d, retained %d" % (len(df), len(tp._db_targets), len(tp.last_knn[2]),
                                                           len(tp.neighbors)))
    print("scan kernel %s: %.2f ms for %.3e pairs" % (name, ms, pairs))
    t0 = time.perf_counter()
    frame = core.neighbor_table(tp)      # 'Similar guides' / 'Similar guide distances' (reference core.py:913-931)
    t["output columns(%d rows)" % len(frame)] = time.perf_counter() - t0
    if a.cfd:
        t0 = time.perf_counter()
        core.cfd_score(frame)
        t["cfd_score(%d rows)" % len(frame)] = time.perf_counter() - t0
    for k, v in t.items():
        print("%-28s %8.3f s" % (k, v))
    print("%-28s %8.3f s" % ("find_targets .. get_neighbors", sum(t[x] for x in ("find_targets", "filters", "create_index", "get_neighbors"))))


if __name__ == "__main__":
    main()
