#!/usr/bin/env python3
"""K1P on pools beyond every BASELINE configuration ("only if pools >= 10^7 matter"): random unique guides,
candidates = pool members + foreign guides, the whole answer against the plain scan of the same library, the reported
distances against numpy, and the steady-state time per query.
   python tools/huge_pool_check.py --targets 16777216 --candidates 1000000 [--guidelen 20] [--k 10]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib  # noqa: E402


def hamming(a, b, L):
    x = a ^ b
    m = (x | (x >> np.uint64(1))) & np.uint64(0x5555555555555555)
    out = np.zeros(a.shape, np.int32)
    for p in range(L):
        out += ((m >> np.uint64(2 * p)) & np.uint64(1)).astype(np.int32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--targets", type=int, default=1 << 24)
    ap.add_argument("--candidates", type=int, default=1_000_000)
    ap.add_argument("--guidelen", type=int, default=20)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch
    rng = np.random.default_rng(a.seed)
    L, k = a.guidelen, a.k
    db = np.unique(rng.integers(0, 1 << (2 * L), size=a.targets, dtype=np.uint64))
    rng.shuffle(db)
    n_mem = a.candidates * 3 // 4
    q = np.concatenate([db[rng.choice(db.shape[0], size=n_mem, replace=False)],
                        rng.integers(0, 1 << (2 * L), size=a.candidates - n_mem, dtype=np.uint64)])
    rng.shuffle(q)
    print("pool %d unique guides of %d bases, %d candidates (%d pool members), k = %d" % (db.shape[0], L, q.shape[0], n_mem, k), flush=True)
    dev = torch.device("cuda", 0)
    d_db = torch.from_numpy(db.view(np.int64)).to(dev)
    d_q = torch.from_numpy(q.view(np.int64)).to(dev)
    n_q = q.shape[0]
    ids = torch.empty((n_q, k), dtype=torch.int32, device=dev)
    dists = torch.empty_like(ids)
    counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    idx = _lib.Index.from_device(d_db.data_ptr(), db.shape[0], L, _lib.METRIC_HAMMING, 0, st)
    try:
        t0 = time.perf_counter()
        idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(a.steps):
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        ev, resc = idx.last_evaluated()
        print("K1P: first query %.3f s (plan + pool copies), then %.2f ms per query = %.3e resolved pairs/s; kernel %s %.2f ms; "
              "evaluated %.3e pairs (%.4f of all), levels %s" % (first, dt * 1e3, float(n_q) * db.shape[0] / dt, idx.last_kernel()[0],
                                                               idx.last_kernel()[1], ev, ev / (float(n_q) * db.shape[0]),
                                                               idx.cascade_info()), flush=True)
        i1, d1, c1 = ids.cpu().numpy(), dists.cpu().numpy(), counts.cpu().numpy()
        idx.set_variant(8)
        t0 = time.perf_counter()
        idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), st)
        torch.cuda.synchronize()
        print("plain scan (%s): %.1f ms" % (idx.last_kernel()[0], (time.perf_counter() - t0) * 1e3), flush=True)
        i8, d8, c8 = ids.cpu().numpy(), dists.cpu().numpy(), counts.cpu().numpy()
        same = bool((i1 == i8).all() and (d1 == d8).all() and (c1 == c8).all())
        true_d = hamming(np.repeat(q[:, None], k, axis=1), db[i1], L)
        ok = same and bool((true_d == d1).all()) and bool((np.diff(d1, axis=1) >= 0).all()) and bool((c1 == k).all())
        members = np.isin(q, db)
        ok = ok and bool((d1[members, 0] == 0).all()) and bool((db[i1[members, 0]] == q[members]).all())
        print("equal to the plain scan: %s; distances true, sorted, full lists, self hits: %s" % (same, ok), flush=True)
        if not ok:
            raise SystemExit(1)
    finally:
        idx.close()


if __name__ == "__main__":
    main()
