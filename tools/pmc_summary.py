#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile.sh) into the per-kernel tables committed under profiles/.

  python tools/pmc_summary.py gpurun_out/TAG --out profiles/r02/TAG_pmc.json [--stats-out profiles/r02/TAG_kernel_stats.csv]
                              [--register s1 --kernel k_hamming_scan_pg [--counters-json profiles/counters.json]]

Reads every *_counter_collection.csv below the directory (one row per dispatch x counter), groups the
dispatches by kernel (template arguments kept, argument lists dropped) and writes, per kernel and counter, the
number of dispatches and the mean value per dispatch; plus the mean dispatch duration seen in the PMC passes
(profiled passes run at a lower clock than un-profiled ones -- never compare across).  FETCH_SIZE / WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so hbm_bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE
KiB (MI355X_MICROARCH.md, HBM section); Infinity-Cache hits are included, i.e. this is the traffic on the L2's
memory side, an upper bound on DRAM traffic.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil


def short_name(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:                      # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    s = "".join(out).strip()
    s = s.replace("(anonymous namespace)::", "").replace("gm::", "")
    if s.startswith("rocprim::") or "rocprim::" in s[:40]:
        m = re.search(r"detail::(\w+)", s)
        s = "rocprim::" + (m.group(1) if m else "kernel")
    return s[:120]


def collect(root, dispatch_min_ns=0.0, split=()):
    """split: substrings of kernel names whose dispatches are told apart by their ORDER inside a query (since the query's
    first kernel, k_to_planes): 'k_leven_part_bs<23, 10, false> #0' is the first partitioned level of every query, '#1'
    the second -- the same instantiation launched twice per step with different work."""
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    regs = {}
    for path in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        seen = set()
        with open(path, newline="") as fh:
            rows = list(csv.DictReader(fh))
        order = {}      # Dispatch_Id -> occurrence index of its kernel inside its query
        if split:
            first = {}
            for row in rows:
                first.setdefault(int(row["Dispatch_Id"]), short_name(row["Kernel_Name"]))
            seen_in_query = collections.Counter()
            for did in sorted(first):
                if "k_to_planes" in first[did]:
                    seen_in_query.clear()
                if any(t in first[did] for t in split):
                    order[did] = seen_in_query[first[did]]
                    seen_in_query[first[did]] += 1
        if True:
            for row in rows:
                if int(row["End_Timestamp"]) - int(row["Start_Timestamp"]) < dispatch_min_ns:
                    continue      # e.g. one-pair spot-check launches of the same kernel
                k = short_name(row["Kernel_Name"])
                if int(row["Dispatch_Id"]) in order:
                    k += " #%d" % order[int(row["Dispatch_Id"])]
                per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                regs[k] = {"vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                           "lds_bytes": int(row["LDS_Block_Size"]), "scratch": int(row["Scratch_Size"]),
                           "workgroup": int(row["Workgroup_Size"])}
                key = (path, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {}
    for k in per:
        ent = {"resources": regs[k], "mean_ns_in_pmc_passes": sum(dur[k]) / max(1, len(dur[k])), "counters": {}}
        for c, vals in per[k].items():
            ent["counters"][c] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
        cs = ent["counters"]
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            f = cs.get("FETCH_SIZE", {}).get("mean_per_launch", 0.0)
            w = cs.get("WRITE_SIZE", {}).get("mean_per_launch", 0.0)
            ent["hbm_side_bytes_per_launch"] = int(2 * f * 1024 + w * 1024)
        out[k] = ent
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--out", required=True)
    ap.add_argument("--stats-out")
    ap.add_argument("--min-ns", type=float, default=0.0, help="drop kernels whose mean dispatch is shorter")
    ap.add_argument("--dispatch-min-ns", type=float, default=0.0, help="ignore single dispatches shorter than this")
    ap.add_argument("--split-kernel", action="append", default=[],
                    help="kernel-name substring: tell its dispatches apart by their order inside a query (see collect)")
    ap.add_argument("--counters-json", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "profiles", "counters.json"))
    ap.add_argument("--register", help="workload key: record the dominant kernel's counters in --counters-json, where "
                                       "bench.py looks them up for its roofline block")
    ap.add_argument("--kernel", help="substring of the dominant kernel's short name (for --register)")
    ap.add_argument("--bench-line", help="file whose last JSON line is the bench.py line of the profiled command "
                                         "(tools/profile.sh: gpurun_out/TAG/stats.log): the launch shape it reports "
                                         "(pairs evaluated, positions per block) is registered with the counters, and "
                                         "bench.py uses the counters only for a live launch of the same shape")
    a = ap.parse_args()
    table = collect(a.root, a.dispatch_min_ns, tuple(a.split_kernel))
    table = {k: v for k, v in table.items() if v["mean_ns_in_pmc_passes"] >= a.min_ns}
    with open(a.out, "w") as fh:
        json.dump({"_source": a.root, "_note": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": table},
                  fh, indent=1, sort_keys=True)
    print("wrote", a.out, "(%d kernels)" % len(table))
    if a.stats_out:
        cand = glob.glob(os.path.join(a.root, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if cand:
            shutil.copyfile(cand[0], a.stats_out)
            print("wrote", a.stats_out)
    if a.register and a.kernel:
        hits = sorted(k for k in table if a.kernel in k)
        if hits:
            # the dominant "kernel" may be several instantiations launched once each per step (the pigeonhole scan runs
            # one launch per number of evaluated positions): their counters and durations add up
            launches = [min((v["launches"] for v in table[k]["counters"].values()), default=0) for k in hits]
            ref = max(launches) if launches else 0
            hits = [k for k, n in zip(hits, launches) if ref and n >= ref // 2]      # (drop stray variants with few launches)
            counters = collections.defaultdict(float)
            for k in hits:
                for c, v in table[k]["counters"].items():
                    counters[c] += v["mean_per_launch"]
            total = None
            if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
                total = int(2 * counters.get("FETCH_SIZE", 0.0) * 1024 + counters.get("WRITE_SIZE", 0.0) * 1024)
            data = {}
            if os.path.exists(a.counters_json):
                with open(a.counters_json) as fh:
                    data = json.load(fh)
            data["_comment"] = ("per workload: PMC counters of bench.py's dominant kernel, mean per launch (summed over its "
                                "instantiations when it runs as several launches per step), from rocprofv3 passes "
                                "(tools/profile.sh + tools/pmc_summary.py --register).  FETCH_SIZE / WRITE_SIZE in KiB; "
                                "hbm_side_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, "
                                "MI355X_MICROARCH.md HBM section; includes Infinity-Cache hits).")
            shape = {}
            if a.bench_line and os.path.exists(a.bench_line):
                with open(a.bench_line) as fh:
                    lines = [ln for ln in fh.read().splitlines() if ln.startswith("{")]
                if lines:
                    line = json.loads(lines[-1])
                    roof = line.get("roofline", {})
                    if a.register in line.get("extra", {}).get("workloads", {}):
                        roof = line["extra"]["workloads"][a.register].get("roofline", {})
                    if "pairs_evaluated" in roof:
                        shape["pairs_evaluated"] = roof["pairs_evaluated"]
                        shape["block_positions"] = [int(b["positions"]) for b in roof.get("blocks", [])]
                        shape["k1p_levels"] = line.get("k1p_levels")
                    if a.register == "cfd":
                        shape["pairs"] = line.get("pairs")
            data[a.register] = {"kernel": " + ".join(hits), "source": a.out, **shape,
                                "resources": {k: table[k]["resources"] for k in hits},
                                "mean_ns_in_pmc_passes": sum(table[k]["mean_ns_in_pmc_passes"] for k in hits),
                                "hbm_side_bytes_per_launch": total, "counters": dict(counters)}
            with open(a.counters_json, "w") as fh:
                json.dump(data, fh, indent=1, sort_keys=True)
            print("registered", a.register, " + ".join(hits), "in", a.counters_json)


if __name__ == "__main__":
    main()
