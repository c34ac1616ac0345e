"""Child of tests/test_gpu_fullsize.py::test_sharded_knn_index_world2_full_s1_on_one_gpu (run under torch.distributed.run)."""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
from guidemaker_amd import _lib, distributed, workloads
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks on GPU 0: a one-GPU box cannot host two RCCL ranks
pool = workloads.make_pool("s1")
L, k = pool["guidelen"], pool["k"]
idx = _lib.Index(pool["packed_db"], L, _lib.METRIC_HAMMING, 0)
got = distributed.sharded_knn_index(idx, pool["packed_all"], k, 0)      # this rank scans ITS shard only, one all-gather
lo, hi, per = distributed.shard_range(pool["packed_all"].shape[0], world, rank)
assert idx.last_kernel()[0] .startswith("k_hamming_scan_pg")
if rank == 0:
    want = idx.knn(pool["packed_all"], k)                              # the single-rank table of all 599 093 candidates
    for g, w in zip(got, want):
        assert g.shape == w.shape and (g == w).all()
    print("shard ok", got[0].shape, "shard rows", hi - lo)
dist.barrier()
idx.close()
dist.destroy_process_group()
