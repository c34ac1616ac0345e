set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative" > gpurun_out/r4_t12.log 2>&1; tail -3 gpurun_out/r4_t12.log
for w in s1 pao1; do
timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/r4_tl_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 3 --no-cpu-baseline --extra none > gpurun_out/r4_tl_$w.log 2>&1
python3 tools/step_timeline.py gpurun_out/r4_tl_$w > gpurun_out/r4_tl_$w.txt 2>&1
rm -rf gpurun_out/r4_tl_$w
done
cat gpurun_out/r4_tl_s1.txt
