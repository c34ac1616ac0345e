#!/usr/bin/env python3
"""bench.py -- guide-pair distance comparisons/s of the MI355X neighbour search.

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: the k-NN scan of every candidate guide of the
workload against the whole target pool (TargetProcessor.get_neighbors' knnQueryBatch), inputs
already resident in HBM, outputs left in HBM.  Workload at N=1 = BASELINE.json configs[1] (E. coli
NGG L=20 hamming k=10) as the synthetic stand-in S1 (no real genome exists offline).  For N>1 every
rank holds the full pool (replicated db) and scans its own, equally large shard of candidates (weak
scaling: rank r>0 gets the candidates of a further S1-like genome), followed by ONE RCCL
all-gather of the per-shard neighbour lists (ids, dists, counts).

Prints one JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
B_ALG = {"hamming": 16, "leven": 16}     # algorithmic bytes per guide pair: two 8-byte packed operands
VALU_LANE_OPS_PER_S = 256 * 128 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz


def lookup_ops_per_pair(L):
    """VALU ops per guide pair of the look-up Hamming kernel K1L (gm_lookup.hip.h): per group of 64
    pool targets and per half (32 targets) 2 ops per full/half adder of the carry-save tree over the L
    match words + 5 for the threshold comparator, plus 1 v_cmp per group.  The match words themselves
    come from LDS (L ds_read_b64 per group), not from the VALU."""
    def col(n):          # (ops, carries) to reduce n same-weight words to one
        if n <= 1:
            return 0, 0
        if n == 2:
            return 2, 1
        t, rem = divmod(n, 3)
        ops, car = col(t + rem)
        return 2 * t + ops, t + car
    ops, n = 0, L
    while n > 0:
        o, n = col(n)
        ops += o
    return (2 * (ops + 5) + 1) / 64.0


def _tree_ops(L):
    """Full/half-adder ops of one carry-save count of L one-bit words (gm_bitslice.hip.h)."""
    def col(n):
        if n <= 1:
            return 0, 0
        if n == 2:
            return 2, 1
        t, rem = divmod(n, 3)
        ops, car = col(t + rem)
        return 2 * t + ops, t + car
    ops, n = 0, L
    while n > 0:
        o, n = col(n)
        ops += o
    return ops


def bitsliced_leven_ops_per_pair(L):
    """VALU lane-ops per guide pair of K2B (gm_leven_bs.hip.h) on its BANDED path (every lane's bound within
    the band, the common case once the bounds are seeded): per 32-target word 5 ops per DP cell of the band
    |i - j| <= (3L + 4) // 8, one count tree over the diagonal, the 5-op threshold compare and one address
    bump per row per group.  Groups that take the full-matrix path execute more (5 L^2 + two trees), so the
    fraction derived from this figure is a lower bound."""
    band = (3 * L + 4) // 8
    cells = L * L - (L - band - 1) * (L - band)
    per_word = 5.0 * cells + _tree_ops(L) + 5 + L / 2.0
    return per_word / 32.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="s1", choices=["s1", "s2", "s3", "s4", "tiny"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(pool, seconds):
    """The oracle's exact scan (restatement of nmslib brute_force + bit_hamming/leven), all host
    threads, on a bounded sample: the first q_s candidates against the FULL pool."""
    from oracle import oracle as orc
    from guidemaker_amd import _lib
    orc.build()
    L, metric, k = pool["guidelen"], pool["metric"], pool["k"]
    db = np.frombuffer("".join(_lib.unpack(pool["packed_db"], L)).encode(), np.uint8).reshape(-1, L)
    threads = orc.lib().orc_max_threads()
    q_s = min(db.shape[0], 256 if metric == "hamming" else 16)
    for _ in range(4):      # grow the sample until it costs about `seconds` of CPU time
        t0 = time.perf_counter()
        orc.knn(db, db[:q_s], L, metric, k, threads)
        dt = max(time.perf_counter() - t0, 1e-4)
        if dt >= 0.6 * seconds or q_s == db.shape[0]:
            break
        q_s = int(min(db.shape[0], max(q_s + 1, q_s * seconds / dt)))
    try:      # time the real NMSLib HNSW only if the timing box happens to have it
        import nmslib  # noqa: F401
        have_nmslib = "nmslib importable on this box but not timed (exact scan is the stated baseline)"
    except ImportError:
        have_nmslib = "nmslib unavailable on the timing box"
    return {"value": q_s * db.shape[0] / dt, "unit": "guide-pair comparisons/s", "cores": threads,
            "kind": "port", "nmslib": have_nmslib,
            "sample": "%d candidates x %d targets (full pool), %s, k=%d, %.1f s, OpenMP exact scan "
                      "(oracle/gm_oracle.c: nmslib brute_force restated)" % (q_s, db.shape[0], metric, k, dt)}


def wall_clock_knn(pool, metric_code, device):
    """Wall clock of the genome-wide k-NN from host arrays: index build (H2D + table
    kernels = create_index) + gm_knn_query with host pointers (H2D + kernels + D2H = get_neighbors'
    search).  PCIe-inclusive, reported next to `value`, never as `value`."""
    from guidemaker_amd import _lib
    t0 = time.perf_counter()
    idx = _lib.Index(pool["packed_db"], pool["guidelen"], metric_code, device)
    idx.knn(pool["packed_all"], pool["k"])
    dt = time.perf_counter() - t0
    idx.close()
    return dt


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d"
                             % (args.gpus, args.gpus))
    import torch.distributed as dist
    from guidemaker_amd import _lib, distributed, workloads

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libgmhip has no CPU fallback")
    # GM_BENCH_BACKEND=gloo rehearses the multi-process plumbing with every rank on GPU 0 (a one-GPU
    # box cannot host two RCCL ranks); the driver's runs use the default, nccl = RCCL over xGMI
    backend = os.environ.get("GM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pool = workloads.make_pool(args.workload)
    L, k = pool["guidelen"], pool["k"]
    metric_code = _lib.METRIC_HAMMING if pool["metric"] == "hamming" else _lib.METRIC_LEVEN
    db_np = pool["packed_db"]
    n_db = db_np.shape[0]
    if args.scaling == "weak":
        n_q = pool["packed_all"].shape[0]
        if rank == 0:
            q_np = pool["packed_all"]
        else:   # a further genome of the same kind supplies this rank's candidates
            other = workloads.make_pool(args.workload, seed_offset=1000 * rank)["packed_all"]
            reps = -(-n_q // other.shape[0])
            q_np = np.tile(other, reps)[:n_q]
    else:
        shard = -(-pool["packed_all"].shape[0] // world)
        n_q = shard
        q_np = pool["packed_all"][rank * shard:(rank + 1) * shard]
        if q_np.shape[0] < shard:       # pad the last shard so the all-gather is regular
            q_np = np.concatenate([q_np, np.tile(q_np[-1:], shard - q_np.shape[0])])

    d_db = torch.from_numpy(db_np.view(np.int64)).to(dev)
    d_q = torch.from_numpy(np.ascontiguousarray(q_np).view(np.int64)).to(dev)
    # two result buffers: with N > 1 the all-gather of one step (RCCL, its own stream) overlaps the scan of the
    # next step; every gather is waited for and unpacked inside the timed region
    nbuf = 2 if world > 1 and backend == "nccl" else 1
    ids_b = [torch.empty((n_q, k), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    dists_b = [torch.empty((n_q, k), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    counts_b = [torch.empty((n_q,), dtype=torch.int32, device=dev) for _ in range(nbuf)]
    gather_b = [torch.empty((world * n_q, k), dtype=torch.int32, device=dev) for _ in range(nbuf)] if nbuf == 2 else []
    pending = [None] * nbuf
    gathered = [None, None, None]
    ids, dists, counts = ids_b[0], dists_b[0], counts_b[0]
    stream = torch.cuda.current_stream().cuda_stream
    idx = _lib.Index.from_device(d_db.data_ptr(), n_db, L, metric_code, local_rank, stream)
    step_no = [0]
    use_async = [True]

    def drain(b):
        if pending[b] is not None:
            gathered[:] = pending[b].result()
            pending[b] = None

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        drain(b)                          # the gather that last used this buffer pair
        idx.knn_dev(d_q.data_ptr(), n_q, k, ids_b[b].data_ptr(), dists_b[b].data_ptr(), counts_b[b].data_ptr(), stream)
        if world > 1:   # the one exchange of the path: per-shard neighbour lists over RCCL/xGMI
            if backend == "nccl" and use_async[0]:
                try:
                    pending[b] = distributed.all_gather_neighbors_async(ids_b[b], dists_b[b], world * n_q,
                                                                        out=gather_b[b])
                except Exception as e:      # an RCCL build without async support: gather in line from now on
                    print("bench: async all-gather unavailable (%s); using the blocking form" % e, file=sys.stderr)
                    use_async[0] = False
                    gathered[:] = distributed.all_gather_neighbors(ids_b[b], dists_b[b], counts_b[b], world * n_q)
            elif backend == "nccl":
                gathered[:] = distributed.all_gather_neighbors(ids_b[b], dists_b[b], counts_b[b], world * n_q)
            else:
                gathered[:] = distributed.all_gather_neighbors(ids_b[b].cpu(), dists_b[b].cpu(), counts_b[b].cpu(),
                                                               world * n_q)

    def finish():
        for b in range(nbuf):
            drain(b)

    for _ in range(args.warmup):
        step()
    finish()
    idx.timing_reserve(max(1, args.steps))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = idx.timing_read()
    kname = idx.last_kernel()[0]
    query_ms = idx.last_query_ms()

    # sanity: every candidate of rank 0 is in the pool, so its nearest hit is itself at distance 0
    if rank == 0:
        c = counts.cpu().numpy()
        d0 = dists[:, 0].cpu().numpy()
        i0 = ids[:, 0].cpu().numpy()
        assert (c == min(k, n_db)).all() and (d0 == 0).all(), "bench sanity check failed"
        uniq_first = np.unique(q_np, return_index=True)[1]
        assert (db_np[i0[uniq_first]] == q_np[uniq_first]).all(), "bench sanity check failed (ids)"

    if rank == 0:
        pairs_per_step = float(world) * n_q * n_db
        value = pairs_per_step * args.steps / elapsed
        kern_s = float(np.mean(kernel_ms)) * 1e-3
        # pairs evaluated by ONE launch of the dominant kernel (the seeded phase of the two-phase scan
        # covers all pool segments but the first), as counted by the library
        launch_pairs = float(idx.last_kernel()[2])
        # K1P resolves launch_pairs (all candidate x target pairs) but evaluates only those sharing a block
        evaluated_pairs, rescanned = idx.last_evaluated()
        evaluated_pairs = float(evaluated_pairs)
        achieved = launch_pairs * B_ALG[pool["metric"]] / kern_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get(args.workload, {}).get("bytes_per_launch")
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                "kernel_ms_avg": kern_s * 1e3, "launches_timed": int(len(kernel_ms)),
                "query_device_ms": query_ms,
                "algorithmic_bytes_per_pair": B_ALG[pool["metric"]], "pairs_per_launch": launch_pairs,
                "note": "algorithmic (no-reuse) operand bytes / HIP-event kernel time; the scan is "
                        "VALU-bound, not HBM-bound, so frac > 1 means LDS/register reuse beats the "
                        "streaming model (SURVEY.md 8d)"}
        if pool["metric"] != "hamming" and 10 <= L <= 27:
            if "knn_bs" in kname:      # K2B: 5 ops per DP cell per 32 pairs + counts, compare, address bumps
                band = (3 * L + 4) // 8
                opp, reads = bitsliced_leven_ops_per_pair(L), (L * L - (L - band - 1) * (L - band)) / 32.0
            else:                      # K2L: 11 VALU ops per text position + distance/threshold tail
                opp, reads = 11.0 * L + 12.0, float(L)
            ops = launch_pairs * opp / kern_s
            roof["valu"] = {"achieved": ops / 1e12, "peak": VALU_LANE_OPS_PER_S / 1e12,
                            "unit": "T lane-ops/s", "frac": ops / VALU_LANE_OPS_PER_S, "ops_per_pair": opp,
                            "lds_reads_per_pair": reads,
                            "note": "the binding roof for the Levenshtein scan: int32 VALU issue "
                                    "(256 CUs x 128 lanes/clk x 2.4 GHz); the HBM figure above is not meaningful "
                                    "for Levenshtein (SURVEY.md 8d)"}
        if pool["metric"] == "hamming" and 10 <= L <= 27:
            L_eval = L
            if "scan_pg" in kname and idx.pigeon_blocks() > 0:      # K1P evaluates L - L // B positions
                roof["pigeon_blocks"] = idx.pigeon_blocks()
                L_eval = max(10, L - L // idx.pigeon_blocks())
            opp = lookup_ops_per_pair(L_eval)
            ops = evaluated_pairs * opp / kern_s
            roof["pairs_evaluated"] = evaluated_pairs
            roof["candidates_rescanned"] = int(rescanned)
            roof["valu"] = {"achieved": ops / 1e12, "peak": VALU_LANE_OPS_PER_S / 1e12,
                            "unit": "T lane-ops/s", "frac": ops / VALU_LANE_OPS_PER_S,
                            "ops_per_pair": opp,
                            "lds_reads_per_pair": L_eval / 64.0,
                            "note": "the binding roofs: int32 VALU issue (256 CUs x 128 lanes/clk x 2.4 GHz) "
                                    "and LDS read issue (L ds_read_b64 per 64 pairs per lane); computed on the "
                                    "pairs the launch EVALUATED (pairs_evaluated; the pigeonhole-partitioned "
                                    "scan resolves pairs_per_launch pairs by evaluating a fraction of them)"}
        # the ALGORITHMIC op model (4 int32 ops per Hamming pair for the popcount form,
        # 17 L per Levenshtein pair for per-pair Myers) against 256 CUs x 64 lanes/clk x 2.4 GHz; > 1 means
        # the bit-sliced kernels need fewer instructions than that model assumes
        alg_ops = 4.0 if pool["metric"] == "hamming" else 17.0 * L
        roof["alg_valu"] = {"ops_per_pair": alg_ops, "achieved": launch_pairs * alg_ops / kern_s / 1e12,
                            "peak": 39.3216, "unit": "T int32 ops/s",
                            "frac": launch_pairs * alg_ops / kern_s / 39.3216e12}
        out = {
            "metric": "guide-pair distance comparisons/sec (whole node)",
            "value": value, "unit": "guide-pair comparisons/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": pool["label"], "n_candidates_per_gpu": int(n_q), "n_targets": int(n_db),
                       "guidelen": L, "k": k, "distance": pool["metric"],
                       "parallelism": "candidate shards x%d, replicated pool, 1 RCCL all-gather per step"
                                      "%s" % (world, " (overlapping the next step's scan)" if nbuf == 2 and use_async[0] else "")},
            "roofline": roof,
        }
        if world == 1:
            out["genome_wide_knn_wall_s"] = wall_clock_knn(pool, metric_code, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pool, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    idx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
