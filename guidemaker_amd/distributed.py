"""Multi-GPU decomposition of the neighbour search: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The path shards embarrassingly by candidate guide: every rank holds the whole
target pool (replicated index), scans a contiguous, equally sized (padded) shard of the candidates,
and ONE all-gather gives every rank the full neighbour lists.  Nothing is reduced and there is no
second exchange.  What crosses xGMI is one 32-bit key per neighbour, (dist << 27) | id with
0xffffffff = "no neighbour" (the kernels' own list format): 4 bytes per neighbour in a single tensor
instead of ids + dists + counts.
"""
import numpy as np

ID_BITS = 27
ID_MASK = (1 << ID_BITS) - 1
EMPTY = -1            # 0xffffffff seen as int32


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
    # a process group can only exist if the caller imported torch.distributed already: never pay the
    # ~1 s `import torch` on the single-GPU path just to learn that there is none
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def pack_keys(ids, dists):
    """[n, k] int32 ids / dists (-1 = none) -> [n, k] int32 keys (bit pattern of the uint32 key)."""
    import torch
    if ids.is_cuda:
        from . import _lib
        keys = torch.empty_like(ids)
        _lib.pack_keys_dev(ids.data_ptr(), dists.data_ptr(), ids.shape[0], ids.shape[1], keys.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
        return keys
    keys = (dists.to(torch.int64) << ID_BITS) | ids.to(torch.int64)
    keys = torch.where(ids < 0, torch.full_like(keys, 0xFFFFFFFF), keys)
    return (keys & 0xFFFFFFFF).to(torch.int64).sub(((keys >> 31) & 1) << 32).to(torch.int32)


def unpack_keys(keys):
    """[n, k] int32 keys -> (ids [n,k], dists [n,k], counts [n]) int32."""
    import torch
    n, k = keys.shape
    if keys.is_cuda:
        from . import _lib
        ids = torch.empty_like(keys)
        dists = torch.empty_like(keys)
        counts = torch.empty((n,), dtype=torch.int32, device=keys.device)
        _lib.unpack_keys_dev(keys.data_ptr(), n, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
        return ids, dists, counts
    u = keys.to(torch.int64) & 0xFFFFFFFF
    ok = u != 0xFFFFFFFF
    ids = torch.where(ok, u & ID_MASK, torch.full_like(u, -1)).to(torch.int32)
    dists = torch.where(ok, u >> ID_BITS, torch.full_like(u, -1)).to(torch.int32)
    return ids, dists, ok.sum(dim=1).to(torch.int32)


def all_gather_neighbors(ids, dists, counts, n_total, group=None):
    """ids/dists: [per, k] int32 tensors of this rank's shard (rows past the shard's end are padding with
    -1 everywhere), counts: [per] (implied by the -1 padding, not transmitted).  Returns the gathered
    (ids, dists, counts) trimmed to n_total rows.  ONE all_gather_into_tensor of packed keys."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    per, k = ids.shape
    keys = pack_keys(ids.contiguous(), dists.contiguous())
    gathered = torch.empty((world * per, k), dtype=torch.int32, device=keys.device)
    dist.all_gather_into_tensor(gathered, keys, group=group)
    g_ids, g_dists, g_counts = unpack_keys(gathered)
    return g_ids[:n_total], g_dists[:n_total], g_counts[:n_total]


class PendingGather:
    """An all-gather of neighbour lists in flight (RCCL runs it on its own stream).  result() makes the
    current stream wait for it and unpacks the keys: (ids, dists, counts) trimmed to n_total rows."""

    def __init__(self, work, gathered, n_total):
        self._work, self._gathered, self._n = work, gathered, n_total

    def result(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        g_ids, g_dists, g_counts = unpack_keys(self._gathered)
        return g_ids[:self._n], g_dists[:self._n], g_counts[:self._n]


def all_gather_neighbors_async(ids, dists, n_total, group=None, out=None):
    """Like all_gather_neighbors, but returns at once with a PendingGather: the collective overlaps whatever
    the caller enqueues next (the following candidate batch's scan).  `out`: optional [world * per, k] int32
    tensor to gather into (double buffering); the caller must not reuse it before result()."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    per, k = ids.shape
    keys = pack_keys(ids.contiguous(), dists.contiguous())
    gathered = out if out is not None else torch.empty((world * per, k), dtype=torch.int32, device=keys.device)
    work = dist.all_gather_into_tensor(gathered, keys, group=group, async_op=True)
    return PendingGather(work, gathered, n_total)


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


def sharded_nearest(local_nearest, q_packed, device=None, group=None):
    """Nearest-target distance of all candidates with the work split over the ranks (get_control_seqs' k = 2 batch,
    reference core.py:603: every rank draws the SAME random sequences -- same seed, same stream -- scans its contiguous
    shard, and ONE all-gather of int32 distances gives every rank the whole vector).

    local_nearest(q_shard_uint64) -> [n] int32 numpy array (Index.nearest of this rank's GPU).  q_packed: the FULL
    candidate array, identical on every rank.  Returns the full [n] int32 array on every rank."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q_packed = np.ascontiguousarray(q_packed, dtype=np.uint64)
    n = int(q_packed.shape[0])
    lo, hi, per = shard_range(n, world, rank)
    d1 = np.full((per,), -1, np.int32)
    if hi > lo:
        d1[:hi - lo] = local_nearest(q_packed[lo:hi])
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.from_numpy(d1).to(device)
    gathered = torch.empty((world * per,), dtype=torch.int32, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return gathered[:n].cpu().numpy()


def sharded_nearest_index(index, q_packed, device_ordinal=0, group=None):
    """sharded_nearest on this rank's guidemaker_amd._lib.Index; on the RCCL backend the shard's query and its distances
    stay on the GPU from the scan (Index.nearest_dev) through the all-gather."""
    import torch
    dist = _dist()
    if dist.get_backend(group) != "nccl":
        _check_same_candidates(q_packed, "cpu", group)
        return sharded_nearest(index.nearest, q_packed, device="cpu", group=group)
    dev = torch.device("cuda", device_ordinal)
    _check_same_candidates(q_packed, dev, group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q_packed = np.ascontiguousarray(q_packed, dtype=np.uint64)
    n = int(q_packed.shape[0])
    lo, hi, per = shard_range(n, world, rank)
    with torch.cuda.device(dev):
        d1 = torch.full((per,), -1, dtype=torch.int32, device=dev)
        if hi > lo:
            d_q = torch.from_numpy(q_packed[lo:hi].view(np.int64)).to(dev)
            index.nearest_dev(d_q.data_ptr(), hi - lo, d1.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        gathered = torch.empty((world * per,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered, d1, group=group)
        return gathered[:n].cpu().numpy()


def _check_same_candidates(q_packed, device, group=None):
    """Every rank must hold the same candidate array: compare (n, xor, sum) fingerprints across the ranks BEFORE the
    shapes of the collective depend on them (ranks working on different frames would hang or mix results)."""
    import torch
    dist = _dist()
    q = np.ascontiguousarray(q_packed, dtype=np.uint64)
    fp = np.array([q.shape[0], int(np.bitwise_xor.reduce(q)) & 0x7FFFFFFFFFFFFFFF if q.size else 0,
                   int(q.sum(dtype=np.uint64)) & 0x7FFFFFFFFFFFFFFF if q.size else 0], dtype=np.int64)
    mine = torch.from_numpy(fp).to(device)
    world = dist.get_world_size(group)
    allfp = torch.empty((world, 3), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allfp, mine.reshape(1, 3), group=group)
    if not bool((allfp == mine.reshape(1, 3)).all().item()):
        raise RuntimeError("sharded_knn: the ranks hold different candidate sets (n / fingerprint per rank: %s); "
                           "sharding needs the same targets frame on every rank" % allfp.cpu().tolist())


def sharded_knn_index(index, q_packed, k, device_ordinal=0, group=None):
    """k-NN of all candidates with the work split over the ranks; `index` is this rank's guidemaker_amd._lib.Index.
    On the RCCL backend the shard's query, its neighbour lists and the gathered table stay on the GPU from the scan
    (Index.knn_dev) through the all-gather; only the final table comes back to the host.  Other backends (gloo, CPU
    rehearsal) go through host arrays."""
    import torch
    dist = _dist()
    if dist.get_backend(group) != "nccl":
        _check_same_candidates(q_packed, "cpu", group)
        return sharded_knn(index.knn, q_packed, k, device="cpu", group=group)
    dev = torch.device("cuda", device_ordinal)
    _check_same_candidates(q_packed, dev, group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    q_packed = np.ascontiguousarray(q_packed, dtype=np.uint64)
    n = int(q_packed.shape[0])
    lo, hi, per = shard_range(n, world, rank)
    with torch.cuda.device(dev):
        ids = torch.full((per, k), -1, dtype=torch.int32, device=dev)
        dists = torch.full((per, k), -1, dtype=torch.int32, device=dev)
        counts = torch.zeros((per,), dtype=torch.int32, device=dev)
        if hi > lo:
            d_q = torch.from_numpy(q_packed[lo:hi].view(np.int64)).to(dev)
            st = torch.cuda.current_stream(dev).cuda_stream
            index.knn_dev(d_q.data_ptr(), hi - lo, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        g = all_gather_neighbors(ids, dists, counts, n, group)
        return tuple(x.cpu().numpy() for x in g)
