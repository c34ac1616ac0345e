set -o pipefail
mkdir -p gpurun_out/r27
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_pao1.py -x -q -m gpu -k "pigeon or cascade or prefix or speculative or pao1 or layout or control or repeat or k1p or K1P or nearest" > gpurun_out/r27/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r27/t.log
for w in s1 s3 pao1; do
AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-100 | sed "s/^/$w /"
GM_GROUP_LDS_RATIO=1.0 AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-100 | sed "s/^/$w ratio1 /"
GM_GROUP_LDS_RATIO=100 AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-100 | sed "s/^/$w nolds /"
done
export TMPDIR=/tmp
for w in s1 pao1; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r27/st_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r27/st_$w.log 2>&1
python3 tools/step_timeline.py gpurun_out/r27/st_$w --which -3 > gpurun_out/r27/tl_$w.txt 2>&1
rm -rf gpurun_out/r27/st_$w
done
