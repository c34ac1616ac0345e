#!/usr/bin/env python3
"""A/B of gm_set_param configurations on a workload, device-resident, in ONE process: ms/step, scan kernel ms,
evaluated pairs, re-scanned rows, cascade levels, and whether the result equals the first configuration's.
   python tools/param_ab.py pao1 pg_hit=0 pg_hit=1 pg_hit=1,pg_per=22
A configuration is a comma-separated list of name=value (or "default")."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from guidemaker_amd import _lib, workloads  # noqa: E402

name = sys.argv[1]
steps = int(os.environ.get("AB_STEPS", "20"))
pool = workloads.make_pool(name)
metric = _lib.METRIC_HAMMING if pool["metric"] == "hamming" else _lib.METRIC_LEVEN
dev = torch.device("cuda", 0)
d_db = torch.from_numpy(pool["packed_db"].view(np.int64)).to(dev)
d_q = torch.from_numpy(pool["packed_all"].view(np.int64)).to(dev)
n_q, k = d_q.shape[0], pool["k"]
ids = torch.empty((n_q, k), dtype=torch.int32, device=dev)
dists = torch.empty_like(ids)
counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
# clock settle: the first configuration must not run at colder clocks than the others
_idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], metric, 0, st)
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < 0.6:
    _idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize()
_idx.close()
for cfg in sys.argv[2:]:
    idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], metric, 0, st)
    if cfg != "default":
        for item in cfg.split(","):
            pname, val = item.split("=")
            idx.set_param(pname, int(val))
    try:
        for _ in range(3):
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        ev, resc = idx.last_evaluated()
        same = ""
        if ref is None:
            ref = (ids.clone(), dists.clone(), counts.clone())
        else:
            ok = bool((ids == ref[0]).all()) and bool((dists == ref[1]).all()) and bool((counts == ref[2]).all())
            same = "same" if ok else "DIFFERENT"
        lv = " | ".join("L%d %d rows r=%d %.3g pairs -> %d left" % (i, c["n_q"], c["radius"], c["pairs"], c["left"])
                        for i, c in enumerate(idx.cascade_info()))
        print("%-28s: %7.3f ms/step, scan %.3f ms, evaluated %.3g, rescanned %d %s  [%s]"
              % (cfg, dt, idx.last_kernel()[1], ev, resc, same, lv), flush=True)
    except Exception as e:  # noqa: BLE001
        print(cfg, "failed:", e, flush=True)
    idx.close()
