#!/usr/bin/env python3
"""CFD kernel (k_cfd) against its roof: an HBM-streaming kernel, 24 algorithmic bytes per pair (two packed
operands in, one float64 out).  Device-resident inputs, torch events on the launch stream.
   python tools/cfd_bench.py --pairs 50000000 --L 20"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from guidemaker_amd import _lib, core  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=50_000_000)
    ap.add_argument("--L", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--mismatches", type=int, default=4)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.randint(0, 1 << (2 * a.L), (a.pairs,), dtype=torch.int64, device=dev)
    o = g.clone()
    for _ in range(a.mismatches):      # off-targets = guides with up to `mismatches` substituted positions
        o ^= torch.randint(1, 4, (a.pairs,), dtype=torch.int64, device=dev) << (2 * torch.randint(0, a.L, (a.pairs,), device=dev))
    mm = torch.from_numpy(core._load_cfd_table().reshape(-1)).to(dev)
    out = torch.empty((a.pairs,), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L = _lib.lib()

    def launch():
        rc = L.gm_cfd_batch_dev(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(o.data_ptr()), a.pairs, a.L,
                                ctypes.c_void_p(mm.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st))
        assert rc == 0, L.gm_last_error()

    launch()
    torch.cuda.synchronize()
    times = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.median(times))
    out_host = out.cpu().numpy()
    # spot check against the host mirror of calc_cfd on a few pairs
    idx = np.random.default_rng(0).choice(a.pairs, 64, replace=False)
    gs = _lib.unpack(g[idx].cpu().numpy().astype(np.uint64), a.L)
    os_ = _lib.unpack(o[idx].cpu().numpy().astype(np.uint64), a.L)
    want = np.array([core.calc_cfd(x, y) for x, y in zip(gs, os_)])
    assert np.abs(out[idx].cpu().numpy() - want).max() <= 1e-6
    gbs = a.pairs * 24 / (ms * 1e-3) / 1e9
    out = {"kernel": "k_cfd", "pairs": a.pairs, "guidelen": a.L, "mismatches": a.mismatches, "ms": ms,
           "pairs_per_s": a.pairs / (ms * 1e-3),
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                        "algorithmic_bytes_per_pair": 24, "traffic": None}}
    # HBM-side bytes per launch from a rocprofv3 PMC pass of this command, if one is registered (pmc_summary.py --register cfd)
    reg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "counters.json")
    if os.path.exists(reg):
        with open(reg) as fh:
            ent = json.load(fh).get("cfd", {})
        if ent.get("hbm_side_bytes_per_launch"):
            out["roofline"]["traffic"] = ent["hbm_side_bytes_per_launch"]
            out["roofline"]["traffic_over_algorithmic"] = ent["hbm_side_bytes_per_launch"] / (a.pairs * 24.0)
    if not a.no_cpu:
        # CPU beside it: the oracle's calc_cfd restatement, all host threads, on a bounded sample
        import time
        from oracle import oracle as orc
        orc.build()
        n_s = min(a.pairs, 4_000_000)
        gs_all = np.frombuffer("".join(_lib.unpack(g[:n_s].cpu().numpy().astype(np.uint64), a.L)).encode(), np.uint8).reshape(-1, a.L)
        os_all = np.frombuffer("".join(_lib.unpack(o[:n_s].cpu().numpy().astype(np.uint64), a.L)).encode(), np.uint8).reshape(-1, a.L)
        mm_host = core._load_cfd_table()
        orc.cfd_batch(gs_all[:1000], os_all[:1000], a.L, mm_host)
        t0 = time.perf_counter()
        ref = orc.cfd_batch(gs_all, os_all, a.L, mm_host)
        dt = time.perf_counter() - t0
        got = out_host[:n_s]
        assert np.abs(got - ref).max() <= 1e-6 and (got == ref).all()
        out["cpu_baseline"] = {"value": n_s / dt, "unit": "pairs/s", "cores": orc.lib().orc_max_threads(), "kind": "port",
                               "sample": "%d pairs, oracle/gm_oracle.c orc_cfd_batch (calc_cfd restated), OpenMP" % n_s}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
