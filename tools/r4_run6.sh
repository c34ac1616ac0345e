set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pigeonhole or layouts or cascade or id_cutoff or prefix_rescan or list_capacity or every_guide_length or hamming_knn or nearest" > gpurun_out/r4_t6.log 2>&1; tail -3 gpurun_out/r4_t6.log
for w in s1 pao1 s3 s4; do AB_STEPS=10 timeout -k 10 150 python tools/param_ab.py $w pg_hit=1 pg_hit=0 pg_hit=1,pg_per=22 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_ab6.log; 
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_w75.so AB_STEPS=10 timeout -k 10 150 python tools/param_ab.py $w pg_hit=1 2>&1 | grep -v amdgpu.ids | sed 's/^/w75 /' >> gpurun_out/r4_ab6.log; done
cat gpurun_out/r4_ab6.log
