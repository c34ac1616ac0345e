set -o pipefail
mkdir -p gpurun_out/r35
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_pao1.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r35/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r35/t.log
[ $rc -eq 0 ] || exit 1
export TMPDIR=/tmp
for w in s4 pao1; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r35/st_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r35/st_$w.log 2>&1
python3 tools/step_timeline.py gpurun_out/r35/st_$w --which -3 > gpurun_out/r35/tl_$w.txt 2>&1
rm -rf gpurun_out/r35/st_$w
done
grep -A6 "k_merge_pg3<20" gpurun_out/r35/tl_s4.txt | head -8; tail -12 gpurun_out/r35/tl_pao1.txt | head -10
for w in s1 s3 pao1 s4; do
AB_STEPS=20 timeout -k 10 200 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-110 | sed "s/^/$w /"
done
