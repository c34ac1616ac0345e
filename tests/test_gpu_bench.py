"""bench.py end to end on the GPU box: the single-GPU line and a two-rank rehearsal of the multi-GPU path (both ranks
on GPU 0, gloo collectives: a one-GPU box cannot host two RCCL ranks).  Guards the contract fields and -- above all --
that every rank leaves with exit code 0 (a sanity check that only held on rank 0 once broke the multi-rank runs)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


def test_bench_single_gpu_line():
    out = subprocess.run([sys.executable, "bench.py", "--workload", "tiny", "--steps", "3", "--warmup", "1",
                          "--cpu-seconds", "1", "--extra", "s2,pao1,cfd", "--extra-steps", "3"], cwd=ROOT,
                         capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_evaluated", "rccl_world"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "valu" and 0.0 < r["frac"] <= 1.0 and r["useful"]["frac"] <= r["frac"] + 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "hnsw" in cb
    assert 0.0 <= cb["hnsw"]["top2_distance_recall_vs_exact"] <= 1.0
    # the further workloads of the same process (the driver's default run carries s2, s3, s4, pao1 and cfd)
    ex = d["extra"]["workloads"]
    assert set(ex) == {"s2", "pao1", "cfd"}, ex.keys()
    for name in ("s2", "pao1"):
        w = ex[name]
        assert "error" not in w, w
        for key in ("ms_per_step", "kernel", "kernel_ms_avg", "issued_frac", "useful_frac", "traffic", "pairs_evaluated",
                    "candidates_rescanned", "k1p_levels", "roofline", "value", "value_evaluated"):
            assert key in w, (name, key)
        assert w["steps"] == 3 and w["ms_per_step"] > 0 and 0.0 < w["useful_frac"] <= 1.0
        assert w["roofline"]["counters"] in (None, "stale") or isinstance(w["roofline"]["counters"], dict)
    assert ex["pao1"]["data"] == "real genome" and ex["pao1"]["kernel"] .startswith("k_hamming_scan_pg")
    assert ex["cfd"]["roofline"]["bound"] == "hbm" and 0.0 < ex["cfd"]["roofline"]["frac"] <= 1.0


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_rank_rehearsal(scaling):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GM_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                          "--workload", "tiny", "--steps", "2", "--warmup", "1", "--scaling", scaling],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]      # EVERY rank must pass its checks
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["scaling"] == scaling and d["collective_backend"] == "gloo"
    assert d["value"] > 0 and "cpu_baseline" not in d and "extra" not in d
    # every rank's own figures travel with the line (a scaling curve has to be attributable), and every rank has checked
    # the gathered table against its own scan of every shard before the timed region
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    for p in pr:
        assert p["kernel_ms_avg"] > 0 and p["gather_ms_sync"] > 0 and p["ms_per_step"] > 0
    assert d["gather_bytes_per_rank"] == d["config"]["n_candidates_per_gpu"] * d["config"]["k"] * 4
