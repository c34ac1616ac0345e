"""World-size-2 tests of the multi-GPU decomposition on CPU (gloo).  The per-rank scan is replaced by
the oracle (no GPU here); what is under test is the sharding, padding, the single all-gather per
tensor, and that the gathered result equals the unsharded one (shard-count invariance)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import random_pool

from guidemaker_amd import _lib, distributed


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_q, k, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        L = 20
        db = list(dict.fromkeys(random_pool(700, L, seed=3)))
        q = (db + random_pool(400, L, seed=4))[:n_q]
        assert len(q) == n_q
        calls = []

        def local_knn(q_shard, kk):     # stand-in for Index.knn on this rank's GPU
            calls.append(len(q_shard))
            ids, dists, counts = orc.knn(db, _lib.unpack(q_shard, L), L, "hamming", kk)
            return ids, np.where(dists >= 0, dists // 2, -1), counts

        assert distributed.is_distributed()
        got = distributed.sharded_knn(local_knn, _lib.pack(q, L), k)
        lo, hi, per = distributed.shard_range(n_q, world, rank)
        assert calls == ([hi - lo] if hi > lo else [])
        want = local_knn(_lib.pack(q, L), k) if n_q else (np.empty((0, k), np.int32),) * 2 + (np.empty(0, np.int32),)
        for g, w in zip(got, want):
            assert g.shape == w.shape and (g == w).all()
        # the asynchronous form (bench.py overlaps it with the next batch's scan): two gathers in flight into
        # caller-owned buffers, results taken in order
        if n_q:
            shard = [torch.from_numpy(np.ascontiguousarray(a)) for a in
                     (np.full((per, k), -1, np.int32), np.full((per, k), -1, np.int32))]
            if hi > lo:
                i, d, _ = local_knn(_lib.pack(q, L)[lo:hi], k)
                shard[0][:hi - lo], shard[1][:hi - lo] = torch.from_numpy(i), torch.from_numpy(d.astype(np.int32))
            bufs = [torch.empty((world * per, k), dtype=torch.int32) for _ in range(2)]
            pend = [distributed.all_gather_neighbors_async(shard[0], shard[1], n_q, out=b) for b in bufs]
            for p in pend:
                gi, gd, gc = p.result()
                assert (gi.numpy() == want[0]).all() and (gd.numpy() == want[1]).all() and (gc.numpy() == want[2]).all()
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_q,k", [(1000, 10), (777, 3), (1, 2), (0, 2)])
def test_sharded_knn_world2(tmp_path, n_q, k):
    from oracle import oracle as orc
    orc.build()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_q, k, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok_%d.npy" % r)) for r in range(world))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000, 1001):
        for world in (1, 2, 4, 8):
            spans = [distributed.shard_range(n, world, r) for r in range(world)]
            per = spans[0][2]
            assert all(s[2] == per for s in spans) and per * world >= n
            covered = [i for lo, hi, _ in spans for i in range(lo, hi)]
            assert covered == list(range(n))
            assert all(hi - lo <= per for lo, hi, _ in spans)


def test_not_distributed_by_default():
    assert not distributed.is_distributed()


def test_key_packing_roundtrip_cpu():
    ids = torch.tensor([[0, 5, -1], [(1 << 27) - 1, 7, 9], [-1, -1, -1]], dtype=torch.int32)
    dists = torch.tensor([[0, 3, -1], [27, 16, 20], [-1, -1, -1]], dtype=torch.int32)
    keys = distributed.pack_keys(ids, dists)
    assert keys.dtype == torch.int32
    assert (keys.to(torch.int64) & 0xFFFFFFFF).tolist() == [
        [0, (3 << 27) | 5, 0xFFFFFFFF], [(27 << 27) | ((1 << 27) - 1), (16 << 27) | 7, (20 << 27) | 9],
        [0xFFFFFFFF] * 3]
    i2, d2, c2 = distributed.unpack_keys(keys)
    assert (i2 == ids).all() and (d2 == dists).all() and c2.tolist() == [2, 3, 0]


class _OracleIndex:
    """Stand-in for guidemaker_amd._lib.Index on a rank without a GPU: .knn through the oracle."""

    def __init__(self, db, L):
        self.db, self.L = db, L

    def knn(self, q_shard, kk):
        from oracle import oracle as orc
        ids, dists, counts = orc.knn(self.db, _lib.unpack(q_shard, self.L), self.L, "hamming", kk)
        return ids, np.where(dists >= 0, dists // 2, -1), counts

    def nearest(self, q_shard):
        self.nearest_calls = getattr(self, "nearest_calls", []) + [int(len(q_shard))]
        if len(q_shard) == 0:
            return np.empty((0,), np.int32)
        return self.knn(q_shard, 1)[1][:, 0].astype(np.int32)


def _worker_index(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, k = 20, 5
        db = list(dict.fromkeys(random_pool(500, L, seed=8)))
        q = _lib.pack(db[:301], L)
        index = _OracleIndex(db, L)
        # the entry point TargetProcessor.get_neighbors uses when shard_candidates is set: same frame on all ranks
        got = distributed.sharded_knn_index(index, q, k)
        want = index.knn(q, k)
        for g, w in zip(got, want):
            assert (g == w).all()
        # strong scaling as bench.py --scaling strong shards it: every rank scans ceil(n / world) rows (the last
        # shard padded with copies of its last row), the gathered table trimmed to n equals the unsharded result
        shard = -(-q.shape[0] // world)
        mine = q[rank * shard:(rank + 1) * shard]
        if mine.shape[0] < shard:
            mine = np.concatenate([mine, np.tile(mine[-1:], shard - mine.shape[0])])
        i, d, c = index.knn(mine, k)
        gi, gd, gc = distributed.all_gather_neighbors(torch.from_numpy(i), torch.from_numpy(d.astype(np.int32)),
                                                      torch.from_numpy(c), q.shape[0])
        assert (gi.numpy() == want[0]).all() and (gd.numpy() == want[1]).all() and (gc.numpy() == want[2]).all()
        # ranks that hold DIFFERENT candidate sets must fail loudly on every rank instead of entering the gather
        other = q if rank == 0 else q[:-3]
        with pytest.raises(RuntimeError, match="different candidate sets"):
            distributed.sharded_knn_index(index, other, k)
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_sharded_knn_index_and_strong_scaling_world2(tmp_path):
    from oracle import oracle as orc
    orc.build()
    world = 2
    mp.spawn(_worker_index, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok_%d.npy" % r)) for r in range(world))


def _worker_controls(rank, world, port, out_dir):
    """get_control_seqs' random query batch is split over the ranks like the candidates of get_neighbors."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import targets_frame
        from guidemaker_amd import core
        L = 20
        db = list(dict.fromkeys(random_pool(400, L, seed=21)))
        index = _OracleIndex(db, L)
        # (1) the building block: one all-gather of int32 distances, any length (odd, 1, 0)
        for n in (301, 1, 0, 64):
            q = _lib.pack((db + random_pool(100, L, seed=22))[:n], L) if n else np.empty((0,), np.uint64)
            got = distributed.sharded_nearest_index(index, q)
            want = index.nearest(q)
            assert got.dtype == np.int32 and got.shape == want.shape and (got == want).all(), n
        other = _lib.pack(db[:50] if rank == 0 else db[:49], L)
        with pytest.raises(RuntimeError, match="different candidate sets"):
            distributed.sharded_nearest_index(index, other)
        # (2) TargetProcessor.get_control_seqs with shard_candidates: every rank draws the same stream (same seed),
        # scans only its shard of each draw, and ends with the table the unsharded search gives
        import types
        import yaml
        cfg = {"CONTROL": {"MINIMUM_HMDIST": 6, "CONTROL_SEARCH_MULTIPLE": [10, 100, 1000]}, "NMSLIB": {"ef": 9}}
        cfg_path = os.path.join(out_dir, "cfg_%d.yaml" % rank)
        with open(cfg_path, "w") as fh:
            yaml.safe_dump(cfg, fh)
        rec = [types.SimpleNamespace(seq="ACGT" * 500 + "GGCC" * 100)]
        tables = []
        for shard in (True, False):
            tp = core.TargetProcessor(targets_frame(db), lsr=0, shard_candidates=shard)
            tp.nmslib_index = _OracleIndex(db, L)
            tp._db_targets, tp._guidelen = db, L
            np.random.seed(7)
            mn, med, table = tp.get_control_seqs(rec, cfg_path, length=L, n=10)
            tables.append((mn, med, table, tp.nmslib_index.nearest_calls, tp.ncontrolsearched))
        (mn_s, med_s, t_s, calls_s, searched_s), (mn_u, med_u, t_u, calls_u, searched_u) = tables
        assert mn_s == mn_u and med_s == med_u and searched_s == searched_u
        assert t_s.equals(t_u)
        # the sharded run scanned half of every draw on this rank, the unsharded one all of it
        assert calls_s == [c // world for c in calls_u], (calls_s, calls_u)
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_sharded_controls_world2(tmp_path):
    from oracle import oracle as orc
    orc.build()
    world = 2
    mp.spawn(_worker_controls, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok_%d.npy" % r)) for r in range(world))


def test_target_processor_does_not_shard_unless_asked(monkeypatch):
    """an initialised process group alone must not switch get_neighbors to the sharded path."""
    import pandas as pd
    from helpers import targets_frame
    from guidemaker_amd import core
    tp = core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0)
    assert tp.shard_candidates is False
    monkeypatch.setenv("GM_SHARD_CANDIDATES", "1")
    assert core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0).shard_candidates is True
    assert core.TargetProcessor(targets_frame(["ACGTACGTACGTACGTACGT"]), lsr=0, shard_candidates=False).shard_candidates is False
