set -o pipefail
mkdir -p gpurun_out/r37
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_pao1.py -x -q -m gpu > gpurun_out/r37/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r37/t.log
[ $rc -eq 0 ] || exit 1
export TMPDIR=/tmp
for w in s1 s4 pao1 s3; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r37/st_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --extra none > gpurun_out/r37/st_$w.log 2>&1
python3 tools/step_timeline.py gpurun_out/r37/st_$w --which -3 > gpurun_out/r37/tl_$w.txt 2>&1
rm -rf gpurun_out/r37/st_$w
echo $w; grep "k_merge_pg3" gpurun_out/r37/tl_$w.txt
done
