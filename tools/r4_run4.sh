set -o pipefail
for t in 1 4 5 2; do
echo "== timing build $t (1 = no list insertion, 4 = no record gather, 5 = neither, 2 = no hits)" >> gpurun_out/r4_abl2.log
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_t$t.so AB_STEPS=10 timeout -k 10 120 python tools/param_ab.py s1 pg_hit=1,pg_debug=4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_abl2.log
done
echo "== release" >> gpurun_out/r4_abl2.log
AB_STEPS=10 timeout -k 10 120 python tools/param_ab.py s1 pg_hit=1 pg_hit=0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_abl2.log
cat gpurun_out/r4_abl2.log
