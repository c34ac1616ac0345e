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
VALU_LANE_OPS_PER_S = 256 * 128 * 2.4e9  # 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T int32 lane-ops/s
LDS_CYCLES_PER_READ = 2.4        # measured ds_read_b64 issue cost per CU (profiles/r01/lds_rate_microbench.txt)
COUNTERS_JSON = os.path.join(ROOT, "profiles", "counters.json")


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


# VALU instructions of ONE 64-target group evaluation of K2B's band-only kernel, counted in the ISA (hipcc 7.2,
# `make asm LEN=23`: the fully unrolled block of k_leven_knn_bs<23, 10, false> with 149 / 187 ds_read_b64 -- DP cells of both
# halves, streaming diagonal counter, threshold compare, address bumps): (guide length, band half-width) -> instructions.
# The round-2 model (5 ops per cell + a count tree) over-counts by 7 %: the first and the last in-band cell of every column
# fold two ops each (constant deltas entering the band).
K2B_ISA_VALU_PER_GROUP = {(23, 3): 1472, (23, 4): 1855}
# ... and of k_leven_part_bs<23, 10, false> (K2P: the same evaluation inside the chunk loop; its bounds are clamped to the
# layout's radius r, so the band is max(2, ceil((r - 1) / 2)): 109 / 149 / 187 ds_read_b64 per group)
K2P_ISA_VALU_PER_GROUP = {(23, 2): 1078, (23, 3): 1479, (23, 4): 1855}


def bitsliced_leven_ops_per_pair(L, narrow=False):
    """USEFUL VALU lane-ops per guide pair of K2B (gm_leven_bs.hip.h) on its banded path (every lane's bound within
    2 BAND + 2, the common case once the bounds are seeded), band half-width BAND = ((3L + 4) // 8) // 2 (one less for the
    narrow-band first pass of the two-pass scan): the static instruction count of the unrolled group evaluation in the ISA
    where it was counted (L = 23), else the cell model scaled by the same folding (0.932)."""
    band = ((3 * L + 4) // 8) // 2 - (1 if narrow else 0)
    if (L, band) in K2B_ISA_VALU_PER_GROUP:
        return K2B_ISA_VALU_PER_GROUP[(L, band)] / 64.0
    cells = L * (2 * band + 1) - band * (band + 1)
    per_word = 5.0 * cells + _tree_ops(L) + 5
    return 0.932 * per_word / 32.0


def roofline_block(workload, pool, kname, kern_s, n_timed, query_ms, launch_pairs, evaluated_pairs, rescanned,
                   pigeon_blocks, n_q, n_db, block_info=None, lp_radius=0):
    """Roofline of the dominant scan kernel.  The scans are integer-VALU bound (O(N) bytes for O(N^2) pair work, all
    of it on bit-sliced words in registers / LDS), so the roof is the int32 VALU issue rate: 256 CUs x 4 SIMD-32 x
    2.4 GHz = 78.6 T lane-ops/s.
      achieved = wave-level VALU instructions the kernel ISSUED per launch (SQ_INSTS_VALU from the rocprofv3 PMC pass
                 registered in profiles/counters.json for this workload) x 64 lanes / HIP-event kernel time measured
                 here -> frac <= 1 by construction;
      useful   = the modelled adder-tree + comparator lane-ops of the pairs the launch EVALUATED / the same time;
      traffic  = HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction applied),
                 next to the compulsory bytes 8 (N_q + N_db) + 8 k N_q."""
    L, k, metric = pool["guidelen"], pool["k"], pool["metric"]
    reg = {}
    if os.path.exists(COUNTERS_JSON):
        with open(COUNTERS_JSON) as fh:
            reg = json.load(fh).get(workload, {})
    if reg and reg.get("kernel", "").split("<")[0].strip() != kname.split("(")[0]:
        reg = {}      # counters of another kernel (e.g. the other K1P hit path): do not mix
    # The registered counters describe ONE launch shape: the layout the planner picked on the profiled box (the pairs it
    # evaluated, the positions per block).  A live launch of another shape (another layout, cut-off or cascade split)
    # must not be priced with them: then the line says "stale" and carries the modelled figure only.
    stale = None
    if reg:
        live_blocks = [int(pos) for pos, _ in (block_info or [])]
        reg_pairs, reg_blocks = reg.get("pairs_evaluated"), reg.get("block_positions")
        if reg_pairs is None:
            stale = "the registered PMC pass does not record the launch shape"
        elif abs(float(reg_pairs) - evaluated_pairs) > 0.005 * max(evaluated_pairs, 1.0):
            stale = "registered for %.6g evaluated pairs per launch, this launch evaluated %.6g" % (reg_pairs, evaluated_pairs)
        elif reg_blocks is not None and list(reg_blocks) != live_blocks:
            stale = "registered for block positions %s, this launch ran %s" % (reg_blocks, live_blocks)
    ctr = reg.get("counters", {}) if not stale else {}
    peak = VALU_LANE_OPS_PER_S / 1e12
    if metric == "hamming":
        if block_info and sum(p for _, p in block_info) > 0:
            # K1P: every copy's table leaves its own block out; pair-weighted over the blocks of the layout
            tot = float(sum(p for _, p in block_info))
            useful_opp = sum(lookup_ops_per_pair(pos) * p for pos, p in block_info) / tot
            L_eval = sum(pos * p for pos, p in block_info) / tot
        else:
            L_eval = L
            useful_opp = lookup_ops_per_pair(L)
        model = ("2 v_bitop3 per adder of the carry-save tree over the match words (%.2f positions per pair on average) "
                 "+ 5 comparator ops per 32 pairs + 1 v_cmp per 64" % L_eval)
    elif "part_bs" in kname:
        # K2P: evaluated_pairs = the 64 x 64 groups its chunks compare; band from the layout's radius
        band = max(2, (max(lp_radius, 1) - 1 + 1) // 2)
        if (L, band) in K2P_ISA_VALU_PER_GROUP:
            useful_opp = K2P_ISA_VALU_PER_GROUP[(L, band)] / 64.0
        else:
            useful_opp = 0.932 * (5.0 * (L * (2 * band + 1) - band * (band + 1)) + _tree_ops(L) + 5) / 32.0
        model = ("static VALU instruction count of the unrolled banded group evaluation in the ISA (band half-width %d for the "
                 "partition's radius %d) / 64 pairs; pairs = the 64 x 64 groups the chunks compare" % (band, lp_radius))
    elif "knn_bs" in kname:
        # candidates re-scanned > 0 <=> the timed launch was the narrow-band first pass of the two-pass scan
        useful_opp = bitsliced_leven_ops_per_pair(L, narrow=rescanned > 0)
        model = ("static VALU instruction count of the unrolled banded group evaluation in the ISA (DP cells of the band, "
                 "diagonal counter, comparator) / 64 pairs (banded path%s)" % (", narrow-band first pass" if rescanned > 0 else ""))
    else:
        useful_opp = 11.0 * L + 12.0
        model = "per-pair Myers: 11 ops per text position + tail"
    useful = evaluated_pairs * useful_opp / kern_s / 1e12
    roof = {"bound": "valu", "unit": "T lane-ops/s", "peak": peak, "kernel": kname, "kernel_ms_avg": kern_s * 1e3,
            "launches_timed": int(n_timed), "query_device_ms": query_ms,
            "pairs_per_launch": launch_pairs, "pairs_evaluated": evaluated_pairs,
            "candidates_rescanned": int(rescanned)}
    if pigeon_blocks:
        roof["pigeon_blocks"] = pigeon_blocks
        roof["positions_evaluated"] = L_eval if metric == "hamming" else None
        roof["blocks"] = [{"positions": pos, "pairs_evaluated": p} for pos, p in (block_info or [])]
    if "SQ_INSTS_VALU" in ctr:
        issued = ctr["SQ_INSTS_VALU"] * 64.0 / kern_s / 1e12
        roof["achieved"] = issued
        roof["frac"] = issued / peak
        roof["counters"] = {"source": reg.get("source"), "kernel": reg.get("kernel"),
                            "SQ_INSTS_VALU_per_launch": ctr["SQ_INSTS_VALU"],
                            "SQ_INSTS_LDS_per_launch": ctr.get("SQ_INSTS_LDS"),
                            "SQ_WAVES_per_launch": ctr.get("SQ_WAVES"),
                            "kernel_ms_in_pmc_pass": reg.get("mean_ns_in_pmc_passes", 0.0) / 1e6}
        if ctr.get("SQ_INSTS_LDS"):
            # LDS instructions are issued per CU: share of the kernel's CU-cycles the LDS pipe is busy issuing them
            roof["lds_issue_frac"] = ctr["SQ_INSTS_LDS"] * LDS_CYCLES_PER_READ / (256 * 2.4e9 * kern_s)
    else:
        roof["achieved"] = useful
        roof["frac"] = useful / peak
        roof["counters"] = "stale" if stale else None
        roof["note_counters"] = ("registered PMC pass not used (%s): achieved = the modelled useful ops" % stale) if stale else \
            "no PMC pass registered for this workload/kernel: achieved = the modelled useful ops"
    if "achieved" in roof and useful > roof["achieved"]:
        # (K2B: part of the waves run the narrower band, i.e. fewer cells than the model's band: the kernel cannot do
        # more useful work than it issues)
        useful = roof["achieved"]
        model += "; capped at the issued rate"
    roof["useful"] = {"achieved": useful, "frac": useful / peak, "ops_per_pair": useful_opp, "model": model}
    compulsory = 8.0 * (n_q + n_db) + 8.0 * k * n_q
    traffic = reg.get("hbm_side_bytes_per_launch") if not stale else None
    roof["traffic"] = traffic
    roof["compulsory_bytes"] = compulsory
    if traffic:
        roof["traffic_over_compulsory"] = traffic / compulsory
        roof["hbm_frac"] = traffic / kern_s / 1e9 / HBM_PEAK_GBS
    roof["streaming_model_note"] = ("SURVEY.md 8(d)(i): %d operand bytes per resolved pair would be %.3g GB/s -- a "
                                    "no-reuse figure the kernel never moves (operands are reused from LDS / registers); "
                                    "reported for reference only, not a roofline fraction"
                                    % (B_ALG[metric], launch_pairs * B_ALG[metric] / kern_s / 1e9))
    return roof


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: 200 steps of ~1 ms -- a timed region of tens of milliseconds measures the clock ramp as much as the kernels:
    # S1 runs 0.96 ms/step over 10 steps and 0.87 over 200 or 1000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="s1", choices=["s1", "s2", "s3", "s4", "pao1", "pao1_leven", "pao1_l23", "tiny"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", default="auto",
                    help="further workloads measured in the same process after the headline (N = 1 only), under "
                         "extra.workloads: comma list of s2,s3,s4,pao1,cfd; 'auto' = all of them when the headline is s1, "
                         "'none' = skip")
    ap.add_argument("--extra-steps", type=int, default=50)
    ap.add_argument("--clock-settle-s", type=float, default=0.5,
                    help="seconds of untimed steps before the warm-up steps, so that the timed region runs at settled clocks (0 = none)")
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
    out = {"value": q_s * db.shape[0] / dt, "unit": "guide-pair comparisons/s", "cores": threads, "kind": "port",
           "sample": "%d candidates x %d targets (full pool), %s, k=%d, %.1f s, OpenMP exact scan "
                     "(oracle/gm_oracle.c: nmslib brute_force restated)" % (q_s, db.shape[0], metric, k, dt)}
    # A second figure, so that nobody divides the GPU's resolved-pair rate by a deliberately literal port: the SAME exact
    # answer from two bit planes per sequence and 16 targets per AVX-512 VPOPCNTDQ instruction (oracle/gm_oracle_tuned.c,
    # pinned to the port by tests/test_oracle.py; chosen at run time: AVX-512 / AVX2 / scalar)
    if metric == "hamming":
        try:
            q_t = min(db.shape[0], 4096)
            for _ in range(4):
                t0 = time.perf_counter()
                orc.knn_tuned(db, db[:q_t], L, k, threads)
                dt_t = max(time.perf_counter() - t0, 1e-4)
                if dt_t >= 0.3 * seconds or q_t == db.shape[0]:
                    break
                q_t = int(min(db.shape[0], max(q_t + 1, 0.5 * q_t * seconds / dt_t)))
            out["tuned"] = {"value": q_t * db.shape[0] / dt_t, "unit": "guide-pair comparisons/s", "cores": threads, "kind": "tuned",
                            "isa": orc.tuned_isa(),
                            "sample": "%d candidates x %d targets (full pool), hamming, k=%d, %.1f s, OpenMP exact scan on bit "
                                      "planes (oracle/gm_oracle_tuned.c)" % (q_t, db.shape[0], k, dt_t)}
        except Exception as e:      # the baseline must never break the bench line
            out["tuned"] = {"error": repr(e)}
    # the reference's search ALGORITHM on the same cores: HNSW with the shipped parameters (M 16, efConstruction 10,
    # post 1, ef 9; config_default.yaml:3-6), restated from the published algorithm because nmslib itself is not
    # installable (oracle/hnsw_baseline.cpp); build on the FULL pool + a bounded query sample, with the notebook's
    # top-2-distance recall against the exact result of the same sample
    try:
        import nmslib  # noqa: F401
        out["nmslib"] = "nmslib importable on this box but not timed (hnsw_baseline.cpp is the stated stand-in)"
    except ImportError:
        out["nmslib"] = "nmslib unavailable on the timing box"
    try:
        t0 = time.perf_counter()
        h = orc.HnswBaseline(pool["packed_db"], L, metric, M=16, efc=10, post=1, threads=threads, seed=1)
        build_s = time.perf_counter() - t0
        n_s = int(min(pool["packed_all"].shape[0], 100_000 if metric == "hamming" else 20_000))
        sel = np.random.default_rng(0).choice(pool["packed_all"].shape[0], size=n_s, replace=False)
        qs = pool["packed_all"][sel]
        t0 = time.perf_counter()
        _, hd = h.query(qs, k, ef=9, threads=threads)
        query_s = time.perf_counter() - t0
        h.close()
        n_t = min(n_s, 2000)
        asc = np.frombuffer(b"ACGT", np.uint8)
        shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
        qt = np.ascontiguousarray(asc[((qs[:n_t, None] >> shifts) & np.uint64(3)).astype(np.uint8)])
        _, td, _ = orc.knn(db, qt, L, metric, k, threads)
        if metric == "hamming":
            td = td // 2
        n_all = pool["packed_all"].shape[0]
        est = build_s + query_s * n_all / n_s
        notebook = None
        try:      # how close the restatement is to nmslib on the one data set the reference published recall for
            import gzip
            rows = [ln.rstrip("\n").split("\t") for ln in
                    gzip.open(os.path.join(ROOT, "tests", "golden", "carsonella_targets_NGG5p.tsv.gz"), "rt")]
            ti = rows[0].index("target")
            uniq = list(dict.fromkeys(r[ti] for r in rows[1:]))
            cp = _lib.pack(uniq, 20)
            cc = np.frombuffer("".join(uniq).encode(), np.uint8).reshape(-1, 20)
            _, cod, _ = orc.knn(cc, cc, 20, "hamming", 3, 4)
            notebook = {"data": "Carsonella ruddii NGG 5prime L=20, %d unique targets queried against themselves, k=3, 4 "
                                "threads, recall on the first two distances (tests/GridOptimization.ipynb:147-233)" % len(uniq),
                        "settings": []}
            for (M_, efc_, ef_), theirs in (((16, 64, 256), 1.0), ((16, 13, 13), 0.987415), ((16, 16, 13), 0.990037)):
                hh = orc.HnswBaseline(cp, 20, "hamming", M=M_, efc=efc_, post=1, threads=4, seed=1)
                _, cd = hh.query(cp, 3, ef=ef_, threads=4)
                hh.close()
                notebook["settings"].append({"M": M_, "efConstruction": efc_, "ef": ef_, "post": 1,
                                             "recall_restated": orc.top2_distance_recall(cd, cod // 2),
                                             "recall_nmslib_notebook": theirs})
        except Exception as e:
            notebook = {"error": repr(e)}
        out["hnsw"] = {"kind": "reference algorithm restated (not nmslib)", "M": 16, "efConstruction": 10, "post": 1,
                       "recall_vs_notebook": notebook,
                       "ef": 9, "cores": threads, "build_s_full_pool": build_s, "query_s_sample": query_s,
                       "n_query_sample": n_s, "genome_wide_s_estimate": est,
                       "value": float(n_all) * db.shape[0] / est, "unit": "guide-pair comparisons/s (resolved, approximate)",
                       "top2_distance_recall_vs_exact": orc.top2_distance_recall(hd[:n_t], td),
                       "recall_sample": n_t}
    except Exception as e:      # the baseline must never break the bench line
        out["hnsw"] = {"error": repr(e)}
    return out


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


def measure_workload(name, steps, warmup, dev, stream):
    """One further workload in the same process (N = 1): the device-resident k-NN of all its candidates against its
    pool, `steps` timed steps after `warmup` (wall clock around synchronised steps, like the headline), the HIP-event
    time of the dominant kernel, and its roofline block with the same counter guard as the headline."""
    from guidemaker_amd import _lib, workloads
    pool = workloads.make_pool(name)
    L, k = pool["guidelen"], pool["k"]
    metric_code = _lib.METRIC_HAMMING if pool["metric"] == "hamming" else _lib.METRIC_LEVEN
    d_db = torch.from_numpy(pool["packed_db"].view(np.int64)).to(dev)
    d_q = torch.from_numpy(np.ascontiguousarray(pool["packed_all"]).view(np.int64)).to(dev)
    n_q, n_db = d_q.shape[0], d_db.shape[0]
    ids = torch.empty((n_q, k), dtype=torch.int32, device=dev)
    dists = torch.empty_like(ids)
    counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
    idx = _lib.Index.from_device(d_db.data_ptr(), n_db, L, metric_code, dev.index or 0, stream)
    try:
        def step():
            idx.knn_dev(d_q.data_ptr(), n_q, k, ids.data_ptr(), dists.data_ptr(), counts.data_ptr(), stream)
        for _ in range(warmup):
            step()
        idx.timing_reserve(max(1, steps))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = idx.timing_read()
        kname = idx.last_kernel()[0]
        c = counts.cpu().numpy()
        d_all = dists.cpu().numpy()
        assert (c == min(k, n_db)).all() and (np.diff(d_all, axis=1) >= 0).all() and (d_all[:, 0] == 0).all(), \
            "bench sanity check failed (%s)" % name
        kern_s = float(np.mean(kernel_ms)) * 1e-3
        evaluated, rescanned = idx.last_evaluated()
        roof = roofline_block(name, pool, kname, kern_s, len(kernel_ms), idx.last_query_ms(), float(idx.last_kernel()[2]),
                              float(evaluated), rescanned, idx.pigeon_blocks() if "scan_pg" in kname else 0, n_q, n_db,
                              idx.pigeon_block_info() if "scan_pg" in kname else None,
                              lp_radius=idx.levpart_info()["r1"] if pool["metric"] != "hamming" else 0)
        pairs = float(n_q) * n_db
        return {"workload": pool["label"], "data": "real genome" if pool.get("real_genome") else "synthetic",
                "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                "value": pairs * steps / elapsed, "value_evaluated": float(evaluated) * steps / elapsed,
                "n_candidates": int(n_q), "n_targets": int(n_db), "guidelen": L, "k": k, "distance": pool["metric"],
                "kernel": kname, "kernel_ms_avg": kern_s * 1e3, "issued_frac": roof.get("frac") if roof.get("counters") not in (None, "stale") else None,
                "useful_frac": roof["useful"]["frac"], "traffic": roof.get("traffic"),
                "pairs_evaluated": float(evaluated), "candidates_rescanned": int(rescanned),
                "k1p_levels": idx.cascade_info(),
                "k2p_levels": idx.levpart_info() if pool["metric"] != "hamming" else None, "roofline": roof}
    finally:
        idx.close()


def measure_cfd(dev, stream, pairs=50_000_000, L=20, mismatches=4, rounds=20):
    """k_cfd on device-resident pairs (what core.cfd_score batches): an HBM-streaming kernel, 24 algorithmic bytes per
    pair (two packed operands in, one float64 out).  Torch events on the launch stream; checked on a sample against
    the host mirror of calc_cfd."""
    import ctypes
    from guidemaker_amd import _lib, core
    g = torch.randint(0, 1 << (2 * L), (pairs,), dtype=torch.int64, device=dev)
    o = g.clone()
    for _ in range(mismatches):
        o ^= torch.randint(1, 4, (pairs,), dtype=torch.int64, device=dev) << (2 * torch.randint(0, L, (pairs,), device=dev))
    mm = torch.from_numpy(core._load_cfd_table().reshape(-1)).to(dev)
    out = torch.empty((pairs,), dtype=torch.float64, device=dev)
    lib = _lib.lib()

    def launch():
        rc = lib.gm_cfd_batch_dev(ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(o.data_ptr()), pairs, L,
                                  ctypes.c_void_p(mm.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream))
        assert rc == 0, lib.gm_last_error()

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    times = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = float(np.mean(times))
    sel = np.random.default_rng(0).choice(pairs, 64, replace=False)
    gs = _lib.unpack(g[sel].cpu().numpy().astype(np.uint64), L)
    os_ = _lib.unpack(o[sel].cpu().numpy().astype(np.uint64), L)
    want = np.array([core.calc_cfd(x, y) for x, y in zip(gs, os_)])
    assert np.abs(out[sel].cpu().numpy() - want).max() <= 1e-6, "cfd sanity check failed"
    gbs = pairs * 24 / (ms * 1e-3) / 1e9
    res = {"workload": "CFD scores of %d device-resident guide pairs, L=%d, %d substituted positions" % (pairs, L, mismatches),
           "kernel": "k_cfd", "pairs": pairs, "ms_per_step": ms, "kernel_ms_avg": ms, "value": pairs / (ms * 1e-3),
           "unit": "guide pairs scored/s", "dtype": "f64",
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_pair": 24, "traffic": None}}
    if os.path.exists(COUNTERS_JSON):
        with open(COUNTERS_JSON) as fh:
            ent = json.load(fh).get("cfd", {})
        if ent.get("hbm_side_bytes_per_launch") and ent.get("pairs", pairs) == pairs:
            res["roofline"]["traffic"] = ent["hbm_side_bytes_per_launch"]
            res["roofline"]["traffic_over_algorithmic"] = ent["hbm_side_bytes_per_launch"] / (pairs * 24.0)
    return res


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
    def rank_candidates(r):
        """Candidates of rank r (every rank can form every rank's set: the self-check below scans them all)."""
        if args.scaling == "weak":
            if r == 0:
                return pool["packed_all"]
            # the same kind of work on every rank: this rank's candidates are the genome's targets again, in an order
            # of their own (pool members, as in get_neighbors -- candidates from a foreign genome have no self hit and
            # leave twice as many rows to the re-scan, which would show up as a scaling loss that is not one)
            return pool["packed_all"][np.random.default_rng(1000 + r).permutation(pool["packed_all"].shape[0])]
        shard = -(-pool["packed_all"].shape[0] // world)
        q = pool["packed_all"][r * shard:(r + 1) * shard]
        if q.shape[0] < shard:       # pad the last shard so the all-gather is regular
            q = np.concatenate([q, np.tile(q[-1:] if q.shape[0] else pool["packed_all"][-1:], shard - q.shape[0])])
        return q

    q_np = rank_candidates(rank)
    n_q = q_np.shape[0]

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
    last_buf, gathered_buf = [0], [0]
    stream = torch.cuda.current_stream().cuda_stream
    idx = _lib.Index.from_device(d_db.data_ptr(), n_db, L, metric_code, local_rank, stream)
    step_no = [0]
    # async collectives: decided ONCE, before any step, and agreed across the ranks (a rank-local fallback would
    # make the ranks issue different collectives); any later RCCL error propagates
    use_async = False
    if world > 1 and backend == "nccl":
        ok = 1
        try:
            probe_in = torch.zeros((8, k), dtype=torch.int32, device=dev)
            probe_out = torch.empty((world * 8, k), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(probe_out, probe_in, async_op=True).wait()
        except Exception as e:
            print("bench: async all-gather unavailable on rank %d (%s)" % (rank, e), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_async = bool(flag.item())

    def drain(b):
        if pending[b] is not None:
            gathered[:] = pending[b].result()
            pending[b] = None
            gathered_buf[0] = b

    def step():
        b = step_no[0] % nbuf
        step_no[0] += 1
        drain(b)                          # the gather that last used this buffer pair
        idx.knn_dev(d_q.data_ptr(), n_q, k, ids_b[b].data_ptr(), dists_b[b].data_ptr(), counts_b[b].data_ptr(), stream)
        if world > 1:   # the one exchange of the path: per-shard neighbour lists over RCCL/xGMI
            if use_async:
                pending[b] = distributed.all_gather_neighbors_async(ids_b[b], dists_b[b], world * n_q, out=gather_b[b])
            elif backend == "nccl":
                gathered[:] = distributed.all_gather_neighbors(ids_b[b], dists_b[b], counts_b[b], world * n_q)
            else:
                gathered[:] = distributed.all_gather_neighbors(ids_b[b].cpu(), dists_b[b].cpu(), counts_b[b].cpu(),
                                                               world * n_q)
            if not use_async:
                gathered_buf[0] = b
        last_buf[0] = b

    def finish():
        for b in range(nbuf):
            drain(b)

    # Clock settle (untimed, before the W warm-up steps): the chip's clocks keep rising through the first ~100 ms of load
    # (the same S1 step reads 0.96 ms in the first 10 ms of work and 0.885 after 200 ms, profiles/r02/NOTES.md), and at the
    # driver's flags warm-up + timed region are 25 steps = 25 ms.  Half a second of the same steps first makes the timed
    # steps those of a GPU that is already working, which is what a deployment sees; reported as clock_settle_steps.
    settle_steps = 0
    if args.clock_settle_s > 0 and world > 1:
        # (every step issues a collective: all ranks must run the SAME number -- a fixed count, not a wall-clock loop)
        for _ in range(256):
            step()
        finish()
        torch.cuda.synchronize()
        settle_steps = 256
    elif args.clock_settle_s > 0:
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.clock_settle_s:
            for _ in range(8):
                step()
            finish()
            torch.cuda.synchronize()
            settle_steps += 8
    for _ in range(args.warmup):
        step()
    finish()
    gather_ms = None
    if world > 1:
        # Outside the timed region, once: (1) the gathered table of one step equals this rank's OWN scan of every rank's
        # candidates (the replicated pool makes every rank able to check every shard), (2) what one all-gather of a step's
        # neighbour lists costs on its own (synchronous; in the timed loop it overlaps the next step's scan).
        if args.warmup == 0:
            step()
            finish()
        g_ids, g_dists, g_counts = gathered
        chk_ids = torch.empty((n_q, k), dtype=torch.int32, device=dev)
        chk_dists = torch.empty_like(chk_ids)
        chk_counts = torch.empty((n_q,), dtype=torch.int32, device=dev)
        for r in range(world):
            d_qr = d_q if r == rank else torch.from_numpy(np.ascontiguousarray(rank_candidates(r)).view(np.int64)).to(dev)
            idx.knn_dev(d_qr.data_ptr(), n_q, k, chk_ids.data_ptr(), chk_dists.data_ptr(), chk_counts.data_ptr(), stream)
            lo = r * n_q
            assert (g_ids[lo:lo + n_q].to(dev) == chk_ids).all() and (g_dists[lo:lo + n_q].to(dev) == chk_dists).all() \
                and (g_counts[lo:lo + n_q].to(dev) == chk_counts).all(), \
                "rank %d: the gathered rows of rank %d differ from this rank's own scan of them" % (rank, r)
        del chk_ids, chk_dists, chk_counts
        idx.knn_dev(d_q.data_ptr(), n_q, k, ids_b[0].data_ptr(), dists_b[0].data_ptr(), counts_b[0].data_ptr(), stream)
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            dist.barrier()
            t_g = time.perf_counter()
            if backend == "nccl":
                distributed.all_gather_neighbors(ids_b[0], dists_b[0], counts_b[0], world * n_q)
                torch.cuda.synchronize()
            else:
                distributed.all_gather_neighbors(ids_b[0].cpu(), dists_b[0].cpu(), counts_b[0].cpu(), world * n_q)
            times.append((time.perf_counter() - t_g) * 1e3)
        gather_ms = float(min(times))
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
    kernel_ms = idx.timing_read()
    per_rank = None
    if world > 1:
        # every rank's own figures (a poor scaling curve must be attributable): wall clock per step, dominant-kernel
        # time, the synchronous all-gather measured above, rows its K1P pass handed on
        mine = {"rank": rank, "ms_per_step": elapsed / args.steps * 1e3, "kernel_ms_avg": float(np.mean(kernel_ms)),
                "gather_ms_sync": gather_ms, "query_device_ms": idx.last_query_ms(),
                "candidates_rescanned": int(idx.last_evaluated()[1])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kname = idx.last_kernel()[0]
    query_ms = idx.last_query_ms()

    # sanity (every rank): full lists, sorted by distance; the nearest hit is the candidate itself at distance 0 (the
    # candidates are pool members on every rank); and the gathered table holds this rank's rows at its shard offset (a
    # double-buffer or gather regression shows here)
    ids, dists, counts = ids_b[last_buf[0]], dists_b[last_buf[0]], counts_b[last_buf[0]]
    c = counts.cpu().numpy()
    d_all = dists.cpu().numpy()
    assert (c == min(k, n_db)).all() and (np.diff(d_all, axis=1) >= 0).all(), "bench sanity check failed"
    if True:      # (every rank's candidates are pool members under both scaling modes)
        i0 = ids[:, 0].cpu().numpy()
        assert (d_all[:, 0] == 0).all(), "bench sanity check failed (self hits)"
        uniq_first = np.unique(q_np, return_index=True)[1]
        assert (db_np[i0[uniq_first]] == q_np[uniq_first]).all(), "bench sanity check failed (ids)"
    if world > 1:
        g_ids, g_dists, g_counts = gathered
        gb, lo = gathered_buf[0], rank * n_q
        assert (g_ids[lo:lo + n_q].to(dev) == ids_b[gb]).all() and (g_dists[lo:lo + n_q].to(dev) == dists_b[gb]).all() \
            and (g_counts[lo:lo + n_q].to(dev) == counts_b[gb]).all(), "gathered rows differ from the local result"

    if rank == 0:
        # weak: every rank scans n_q candidates; strong: the real candidate total (padding rows are not work)
        pairs_per_step = (float(world) * n_q if args.scaling == "weak" else float(pool["packed_all"].shape[0])) * n_db
        value = pairs_per_step * args.steps / elapsed
        kern_s = float(np.mean(kernel_ms)) * 1e-3
        # pairs evaluated by ONE launch of the dominant kernel (the seeded phase of the two-phase scan
        # covers all pool segments but the first), as counted by the library
        launch_pairs = float(idx.last_kernel()[2])
        # K1P resolves launch_pairs (all candidate x target pairs) but evaluates only those sharing a block
        evaluated_pairs, rescanned = idx.last_evaluated()
        evaluated_pairs = float(evaluated_pairs)
        roof = roofline_block(args.workload, pool, kname, kern_s, len(kernel_ms), query_ms, launch_pairs,
                              evaluated_pairs, rescanned, idx.pigeon_blocks() if "scan_pg" in kname else 0, n_q, n_db,
                              idx.pigeon_block_info() if "scan_pg" in kname else None,
                              lp_radius=idx.levpart_info()["r1"] if pool["metric"] != "hamming" else 0)
        out = {
            "metric": "guide-pair distance comparisons/sec (whole node)",
            "value": value, "unit": "guide-pair comparisons/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "real genome" if pool.get("real_genome") else "synthetic",
            "value_note": "value = candidate x target pairs RESOLVED per second (every pair's membership in the exact "
                          "k-NN answer is decided; BASELINE.md section 4's definition N_q x N_db / seconds); "
                          "value_evaluated = distance evaluations actually EXECUTED per second (the pigeonhole "
                          "partition proves the rest cannot be among the k nearest)",
            "value_evaluated": float(world) * evaluated_pairs * args.steps / elapsed,      # (rank 0's count x ranks)
            "rccl_world": (dist.get_world_size() if world > 1 else 1),
            "collective_backend": (backend if world > 1 else None),
            "config": {"workload": pool["label"], "n_candidates_per_gpu": int(n_q), "n_targets": int(n_db),
                       "guidelen": L, "k": k, "distance": pool["metric"],
                       "parallelism": "candidate shards x%d, replicated pool, 1 RCCL all-gather per step"
                                      "%s" % (world, " (overlapping the next step's scan)" if nbuf == 2 and use_async else "")},
            "roofline": roof,
        }
        out["k1p_levels"] = idx.cascade_info()
        out["clock_settle_steps"] = settle_steps
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["gather_bytes_per_rank"] = int(n_q) * k * 4
        if world == 1:
            out["genome_wide_knn_wall_s"] = wall_clock_knn(pool, metric_code, local_rank)
        want = args.extra
        if want == "auto":
            want = "s2,s3,s4,pao1,pao1_l23,cfd" if args.workload == "s1" else "none"
        if world == 1 and want != "none":
            # the other BASELINE configurations (and the one real genome) in the same process, so that the driver's run
            # times them too; the contract fields above stay the S1 headline's
            extra = {}
            for name in [w for w in want.split(",") if w]:
                try:
                    if name == "cfd":
                        extra[name] = measure_cfd(dev, stream)
                    elif name == "pao1_l23":      # 0.14 s per query: ten timed steps
                        extra[name] = measure_workload(name, min(10, args.extra_steps), 3, dev, stream)
                    else:
                        extra[name] = measure_workload(name, args.extra_steps, max(5, args.warmup), dev, stream)
                except Exception as e:      # a further workload must never break the headline line
                    extra[name] = {"error": repr(e)}
            out["extra"] = {"workloads": extra}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pool, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    idx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
