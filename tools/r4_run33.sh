set -o pipefail
mkdir -p gpurun_out/r33
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r33/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r33/t.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/k1p_stress.py --seconds 240 --seed 20261103 > gpurun_out/r33/stress.txt 2>&1; echo "stress rc=$?"; tail -2 gpurun_out/r33/stress.txt
timeout -k 10 300 python tests/fuzz_parity.py --seconds 120 --seed 78 > gpurun_out/r33/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r33/fuzz.txt
for w in s1 s3 pao1 s4; do
AB_STEPS=20 timeout -k 10 200 python tools/param_ab.py $w default 2>&1 | grep -v amdgpu.ids | cut -c1-110 | sed "s/^/$w /"
done
