import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from guidemaker_amd import _lib, workloads
pool = workloads.make_pool("s2")
dev = torch.device("cuda", 0)
d_db = torch.from_numpy(pool["packed_db"].view(np.int64)).to(dev)
d_q = torch.from_numpy(pool["packed_all"].view(np.int64)).to(dev)
n_q, k = d_q.shape[0], pool["k"]
ids = torch.empty((n_q, k), dtype=torch.int32, device=dev); dists = torch.empty_like(ids); counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
  for lpr in (0, -1):
    idx = _lib.Index.from_device(d_db.data_ptr(), d_db.shape[0], pool["guidelen"], _lib.METRIC_LEVEN, 0, st)
    idx.set_param("lp_radius", lpr)
    for _ in range(5): idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
    torch.cuda.synchronize(); print("lp_radius", lpr, (time.perf_counter() - t0) / 30 * 1e3, "ms/step", idx.last_kernel()[0], idx.last_kernel()[1])
    idx.close()
