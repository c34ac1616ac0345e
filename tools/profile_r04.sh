#!/bin/bash
# tools/profile_r04.sh W...  -- ON THE GPU BOX: tools/profile.sh for every listed workload, condensed into gpurun_out/r04/
# (p_<w>_pmc.json, p_<w>_kernel_stats.csv, p_<w>_bench.json, p_<w>_step_timeline.txt) and registered in a copy of
# profiles/counters.json (gpurun_out/r04/counters.json) with the launch shape of the profiled command.
set -o pipefail
mkdir -p gpurun_out/r04
cp profiles/counters.json gpurun_out/r04/counters.json
for W in "$@"; do
  case $W in
    s1) KERN="k_hamming_scan_pg4<16";; s3) KERN="k_hamming_scan_pg4<14";; s4) KERN="k_hamming_scan_pg4<14";;
    pao1) KERN="k_hamming_scan_pg4<14";; s2) KERN="k_leven_part_bs";; pao1_l23) KERN="k_leven_part_bs<23, 10, false> #0";;
    cfd) KERN="k_cfd";;
  esac
  timeout -k 10 600 bash tools/profile.sh r04_$W $W > gpurun_out/r04/p_${W}_profile.log 2>&1 || { echo "$W: profile failed"; tail -3 gpurun_out/r04/p_${W}_profile.log; continue; }
  grep '^{' gpurun_out/r04_$W/stats.log | tail -1 > gpurun_out/r04/p_${W}_bench.json
  SPLIT=""; [ "$W" = "pao1_l23" ] && SPLIT="--split-kernel k_leven_part_bs"
  python3 tools/pmc_summary.py gpurun_out/r04_$W --out gpurun_out/r04/p_${W}_pmc.json --stats-out gpurun_out/r04/p_${W}_kernel_stats.csv \
      --register $W --kernel "$KERN" --counters-json gpurun_out/r04/counters.json --bench-line gpurun_out/r04_$W/stats.log $SPLIT | tail -2
  [ "$W" != "cfd" ] && python3 tools/step_timeline.py gpurun_out/r04_$W/stats --which -3 > gpurun_out/r04/p_${W}_step_timeline.txt 2>&1
  rm -rf gpurun_out/r04_$W
done
ls gpurun_out/r04
