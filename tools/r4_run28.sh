set -o pipefail
mkdir -p gpurun_out/r28
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_pao1.py -x -q -m gpu > gpurun_out/r28/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r28/t.log
[ $rc -eq 0 ] || exit 1
for w in s1 s3 pao1 s4; do
AB_STEPS=20 timeout -k 10 150 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-100 | sed "s/^/$w /"
done
export TMPDIR=/tmp
for w in s1 pao1; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r28/st_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r28/st_$w.log 2>&1
python3 tools/step_timeline.py gpurun_out/r28/st_$w --which -3 > gpurun_out/r28/tl_$w.txt 2>&1
rm -rf gpurun_out/r28/st_$w
done
