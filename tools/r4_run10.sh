set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t10.log; tail -5 gpurun_out/r4_t10.log
