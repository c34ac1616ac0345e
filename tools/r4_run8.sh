for w in pao1 s4; do
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_st.so AB_STEPS=1 timeout -k 10 120 python tools/param_ab.py $w pg_hit=1 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' | tail -5
done
