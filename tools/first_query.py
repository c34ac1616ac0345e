#!/usr/bin/env python3
"""What a one-shot user pays (SURVEY.md 8d "wall-clock for genome-wide k-NN"): index build from host arrays, the FIRST k-NN
query (layout plan from a sample, block-sorted pool copies, hit buffer) and a second one, host pointers in and out.
   python tools/first_query.py [s1|s3|s4|pao1|s2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from guidemaker_amd import _lib, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "s1"
pool = workloads.make_pool(name)
metric = _lib.METRIC_HAMMING if pool["metric"] == "hamming" else _lib.METRIC_LEVEN
warm = _lib.Index(pool["packed_db"][:70000], pool["guidelen"], metric)      # (HIP context + code objects: once per process)
warm.knn(pool["packed_all"][:1000], pool["k"])
warm.close()
for rep in range(2):
    t0 = time.perf_counter()
    idx = _lib.Index(pool["packed_db"], pool["guidelen"], metric)
    t1 = time.perf_counter()
    idx.knn(pool["packed_all"], pool["k"])
    t2 = time.perf_counter()
    dev1 = idx.last_query_ms()      # (read right after the FIRST query: ADVICE r03 -- it used to be printed as 0.0)
    idx.knn(pool["packed_all"], pool["k"])
    t3 = time.perf_counter()
    print("%s: index %.1f ms, first query %.1f ms (device part %.2f), second query %.1f ms (device part %.2f)"
          % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, dev1, (t3 - t2) * 1e3, idx.last_query_ms()), flush=True)
    idx.close()
