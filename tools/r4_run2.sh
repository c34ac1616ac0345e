set -o pipefail
export TMPDIR=/tmp
for t in t2 t5 t1; do
  echo "== timing build $t (results wrong by design)" >> gpurun_out/r4_abl.log
  GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_$t.so AB_STEPS=10 timeout -k 10 120 python tools/param_ab.py pao1 pg_hit=1 >> gpurun_out/r4_abl.log 2>&1
  GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_$t.so AB_STEPS=10 timeout -k 10 120 python tools/param_ab.py s1 pg_hit=1 >> gpurun_out/r4_abl.log 2>&1
done
CMD="python3 bench.py --workload pao1 --steps 5 --warmup 2 --no-cpu-baseline --extra none"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d gpurun_out/r4_pmc_a/pmc1 -o p --output-format csv -- $CMD > gpurun_out/r4_pmc_a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES -d gpurun_out/r4_pmc_a/pmc2 -o p --output-format csv -- $CMD >> gpurun_out/r4_pmc_a.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r4_pmc_a --out gpurun_out/r4_pmc_a.json > /dev/null 2>&1
cat gpurun_out/r4_abl.log
