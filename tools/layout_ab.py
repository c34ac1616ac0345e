#!/usr/bin/env python3
"""A/B of one K1P knob on a workload, device-resident: ms/step, scan kernel ms, evaluated pairs, re-scanned rows, and
whether the result equals the default's.
   python tools/layout_ab.py pao1 61515140 727161            # forced layouts (gm_set_param "pg_layout")
   DBG_PARAM=pg_cut python tools/layout_ab.py pao1 100 20 35  # id cut-off percentages"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from guidemaker_amd import _lib, workloads
name = sys.argv[1]
pool = workloads.make_pool(name)
dev = torch.device("cuda", 0)
d_db = torch.from_numpy(pool["packed_db"].view(np.int64)).to(dev)
d_q = torch.from_numpy(pool["packed_all"].view(np.int64)).to(dev)
n_q, k = d_q.shape[0], pool["k"]
ids = torch.empty((n_q, k), dtype=torch.int32, device=dev); dists = torch.empty_like(ids); counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
PARAM = os.environ.get("DBG_PARAM", "pg_layout")
# clock settle: the first configuration must not run at colder clocks than the others (half a second of the default one)
_idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], _lib.METRIC_HAMMING, 0, st)
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < 0.6:
    _idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize()
_idx.close()
for lay in [0] + [int(x) for x in sys.argv[2:]]:
    idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], _lib.METRIC_HAMMING, 0, st)
    if lay: idx.set_param(PARAM, lay)
    try:
        for _ in range(3):
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        ev, resc = idx.last_evaluated()
        same = ""
        if ref is None: ref = (ids.clone(), dists.clone())
        else: same = "same" if bool((ids == ref[0]).all()) and bool((dists == ref[1]).all()) else "DIFFERENT"
        lv = " | ".join("L%d %d rows r=%d %.3g pairs -> %d left" % (i, c["n_q"], c["radius"], c["pairs"], c["left"]) for i, c in enumerate(idx.cascade_info()))
        print("layout %8d: %7.2f ms/step, scan %.2f ms, evaluated %.3g, rescanned %d %s  [%s]" % (lay, dt, idx.last_kernel()[1], ev, resc, same, lv), flush=True)
    except Exception as e:
        print("layout", lay, "failed:", e, flush=True)
    idx.close()
