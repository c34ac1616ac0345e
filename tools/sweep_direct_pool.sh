#!/bin/bash
# Direct pool mode of SURVEY.md 8(d): uniform random pools with 1 % planted near-duplicates, plain two-phase scan
# (variant 8) against the default path (variant 0), results compared (--check).  Run ON THE GPU BOX:
#   bash tools/sweep_direct_pool.sh > gpurun_out/sweep_direct_pool.txt
set -e -o pipefail
K="python3 tools/kbench.py --planted 0.01 --rounds 2 --check"
for n in 65536 262144 1048576 4194304; do
  for k in 2 10 20; do
    $K --n $n --L 20 --k $k --metric hamming --variants 8,0 2>&1 | grep -E "^variant|MISMATCH|differ"
  done
done
for L in 10 23 27; do
  $K --n 1048576 --L $L --k 10 --metric hamming --variants 8,0 2>&1 | grep -E "^variant|MISMATCH|differ"
done
for n in 65536 262144; do
  for L in 23 27; do
    $K --n $n --L $L --k 10 --metric leven --variants 2,0 2>&1 | grep -E "^variant|MISMATCH|differ"
  done
done
