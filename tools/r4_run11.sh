set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or cascade or prefix_rescan or id_cutoff or pigeonhole_default or repeat" > gpurun_out/r4_t11.log 2>&1; echo "rc=$?" >> gpurun_out/r4_t11.log; tail -5 gpurun_out/r4_t11.log
for w in s1 pao1 s3 s4; do AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w pg_spec=1 pg_spec=0 pg_spec=1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_ab11.log; done
cat gpurun_out/r4_ab11.log
