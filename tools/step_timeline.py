#!/usr/bin/env python3
"""Timeline of ONE steady-state bench step from a rocprofv3 --kernel-trace CSV: start offset, gap to the previous kernel,
duration and name of every dispatch between two consecutive occurrences of a marker kernel (default: the query's first
kernel, k_to_planes).  Shows where a step's time goes BETWEEN the kernels too (host round trips, launch gaps).
   python tools/step_timeline.py gpurun_out/TAG/stats [--marker k_to_planes] [--which -3]"""
import argparse
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--marker", default="k_to_planes")
    ap.add_argument("--which", type=int, default=-3, help="index of the marker occurrence that starts the step")
    a = ap.parse_args()
    path = glob.glob(os.path.join(a.root, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    i0, i1 = marks[a.which], marks[a.which + 1]
    t0, prev = int(rows[i0]["Start_Timestamp"]), None
    busy = 0
    cut = False
    for r in rows[i0:i1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("gm::", "").replace("void ", "")
        name = name.split("(")[0][:60]
        if name.startswith("__amd_rocclr") or name.startswith("at::"):      # the caller's read-back of the results: the step is over
            cut = True
            break
        print("%9.1f us  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, name))
        prev = e
        busy += e - s
    if cut:
        total = prev - t0
        print("step (up to the caller's result read-back): %.1f us, %.1f us inside kernels, %.1f us between them (launch gaps of the "
              "timed kernel's HIP events included)" % (total / 1e3, busy / 1e3, (total - busy) / 1e3))
    else:
        total = int(rows[i1]["Start_Timestamp"]) - t0
        print("step: %.1f us from marker to marker, %.1f us inside kernels, %.1f us between them" % (total / 1e3, busy / 1e3, (total - busy) / 1e3))


if __name__ == "__main__":
    main()
