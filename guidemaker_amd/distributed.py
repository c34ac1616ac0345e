"""Multi-GPU decomposition of the neighbour search: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards embarrassingly by candidate guide: every rank holds the whole
target pool (replicated index), scans a contiguous, equally sized (padded) shard of the candidates,
and ONE all-gather per result tensor (ids, dists, counts) gives every rank the full neighbour lists.
Nothing is reduced and there is no second exchange.
"""
import numpy as np


def shard_range(n, world, rank):
    """Contiguous shard [lo, hi) of rank; all shards have ceil(n/world) rows except a short/empty tail."""
    per = -(-n // world) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi, per


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed():
    try:
        dist = _dist()
    except ImportError:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_gather_neighbors(ids, dists, counts, n_total, group=None):
    """ids/dists: [per, k] int32 tensors (rows past the shard's end are padding), counts: [per].
    Returns the gathered tensors trimmed to n_total rows.  One all_gather_into_tensor per tensor."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    per, k = ids.shape
    g_ids = torch.empty((world * per, k), dtype=ids.dtype, device=ids.device)
    g_dists = torch.empty((world * per, k), dtype=dists.dtype, device=dists.device)
    g_counts = torch.empty((world * per,), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(g_ids, ids.contiguous(), group=group)
    dist.all_gather_into_tensor(g_dists, dists.contiguous(), group=group)
    dist.all_gather_into_tensor(g_counts, counts.contiguous(), group=group)
    return g_ids[:n_total], g_dists[:n_total], g_counts[:n_total]


def sharded_knn(local_knn, q_packed, k, device=None, group=None):
    """k-NN of all candidates with the work split over the ranks of the process group.

    local_knn(q_shard_uint64, k) -> (ids [n,k] int32, dists [n,k] int32, counts [n] int32) numpy arrays
    (Index.knn of this rank's GPU).  q_packed: the FULL candidate array, identical on every rank.
    Returns full (ids, dists, counts) numpy arrays on every rank, identical to the single-GPU result.
    """
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q_packed = np.ascontiguousarray(q_packed, dtype=np.uint64)
    n = int(q_packed.shape[0])
    lo, hi, per = shard_range(n, world, rank)
    ids = np.full((per, k), -1, np.int32)
    dists = np.full((per, k), -1, np.int32)
    counts = np.zeros((per,), np.int32)
    if hi > lo:
        i, d, c = local_knn(q_packed[lo:hi], k)
        ids[:hi - lo], dists[:hi - lo], counts[:hi - lo] = i, d, c
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = [torch.from_numpy(a).to(device) for a in (ids, dists, counts)]
    g = all_gather_neighbors(t[0], t[1], t[2], n, group)
    return tuple(x.cpu().numpy() for x in g)
