#!/usr/bin/env python3
"""Run-down of the one hang the round-2 GPU suite saw (test_library_first_then_torch_in_one_process: libgmhip first, then
`import torch`, in a fresh child): the child N times in a row, each under its own timeout; a child that does not return is
killed (its own PID only) and re-run once with AMD_LOG_LEVEL=3 so that the runtime's last calls are on record.
   python tools/hang_hunt.py --runs 200 --timeout 60 > gpurun_out/hang_hunt.txt"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
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


def run_once(timeout, env=None):
    t0 = time.perf_counter()
    p = subprocess.Popen([sys.executable, "-c", CHILD, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err, time.perf_counter() - t0
    except subprocess.TimeoutExpired:
        p.kill()                                 # this child only
        out, err = p.communicate()
        return None, out, err, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    ap.add_argument("--timeout", type=float, default=60.0)
    a = ap.parse_args()
    times, bad = [], 0
    for i in range(a.runs):
        rc, out, err, dt = run_once(a.timeout)
        times.append(dt)
        if rc != 0 or "order ok" not in out:
            bad += 1
            print("run %d: %s after %.1f s\nstdout: %s\nstderr tail: %s" % (i, "HUNG (killed)" if rc is None else "rc %s" % rc,
                                                                           dt, out[-300:], err[-1500:]), flush=True)
            rc2, out2, err2, dt2 = run_once(a.timeout, dict(os.environ, AMD_LOG_LEVEL="3"))
            print("  re-run with AMD_LOG_LEVEL=3: %s after %.1f s; last runtime lines:\n%s" %
                  ("HUNG" if rc2 is None else "rc %s" % rc2, dt2, "\n".join(err2.splitlines()[-25:])), flush=True)
        if i % 20 == 19:
            print("%d runs, %d bad, child wall clock min / median / max = %.2f / %.2f / %.2f s" %
                  (i + 1, bad, min(times), sorted(times)[len(times) // 2], max(times)), flush=True)
    print("done: %d runs, %d bad" % (a.runs, bad))


if __name__ == "__main__":
    main()
