set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or cascade or prefix_rescan or id_cutoff or pigeonhole or repeat or list_capacity or layouts or nearest" > gpurun_out/r4_t14.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t14.log; tail -4 gpurun_out/r4_t14.log
bash tools/r4_run13.sh 2>&1 | grep -E "==|merge|scan_pg"
for w in s1 pao1 s3; do AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids; done
