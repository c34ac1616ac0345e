"""S2 (BASELINE configs[2] stand-in, Levenshtein): ms/step with K2P chosen / forced (lp_radius from the command line) / off,
and whether the answers agree.   python tools/s2_ab.py [radius ...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, workloads
pool = workloads.make_pool("s2")
dev = torch.device("cuda", 0)
d_db = torch.from_numpy(pool["packed_db"].view(np.int64)).to(dev)
d_q = torch.from_numpy(pool["packed_all"].view(np.int64)).to(dev)
n_q, k = d_q.shape[0], pool["k"]
ids = torch.empty((n_q, k), dtype=torch.int32, device=dev); dists = torch.empty_like(ids); counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
ref = None
for rep in range(2):
  for lpr in [0, -1] + [int(x) for x in sys.argv[1:]]:
    idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], _lib.METRIC_LEVEN, 0, st)
    idx.set_param("lp_radius", lpr)
    for _ in range(5): idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e3
    if ref is None: ref = (ids.clone(), dists.clone(), counts.clone())
    same = bool((ids == ref[0]).all() and (dists == ref[1]).all() and (counts == ref[2]).all())
    print("lp_radius %2d: %.2f ms/step, kernel %s %.2f ms, plain second-pass rows %d, %s, %s" % (lpr, dt, idx.last_kernel()[0], idx.last_kernel()[1], idx.last_evaluated()[1], idx.levpart_info(), "same" if same else "DIFFERENT"), flush=True)
    idx.close()
