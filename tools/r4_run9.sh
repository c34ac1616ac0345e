set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t9.log; tail -4 gpurun_out/r4_t9.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench9.json 2> gpurun_out/r4_bench9.err; tail -c 3000 gpurun_out/r4_bench9.json
