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
                          "--cpu-seconds", "1"], cwd=ROOT, capture_output=True, text=True, timeout=400)
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
    assert d["value"] > 0 and "cpu_baseline" not in d
