for w in pao1 s3 s4; do AB_STEPS=20 timeout -k 10 200 python tools/param_ab.py $w pg_hit=1 pg_hit=2 pg_hit=2,pg_per=8 pg_hit=0 2>&1 | grep -v amdgpu.ids; done
