set -o pipefail
export TMPDIR=/tmp
echo "== stats build" >> gpurun_out/r4_stats.log
for w in pao1 s1; do
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_st.so AB_STEPS=1 timeout -k 10 120 python tools/param_ab.py $w pg_hit=1 pg_hit=1,pg_per=22 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' >> gpurun_out/r4_stats.log
done
echo "== no-hit timing build (pg_debug=4 keeps the flooded-pool memory off)" >> gpurun_out/r4_stats.log
for w in pao1 s1; do
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_t2.so AB_STEPS=5 timeout -k 10 120 python tools/param_ab.py $w pg_hit=1,pg_debug=4 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4_stats.log
done
cat gpurun_out/r4_stats.log
