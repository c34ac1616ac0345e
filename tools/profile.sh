#!/bin/bash
# tools/profile.sh TAG WORKLOAD|cfd [args...]  -- run ON THE GPU BOX (through gpurun).
# One rocprofv3 --kernel-trace --stats pass and four PMC passes (SQ instruction counters + GRBM; SQ wave / wait states;
# FETCH_SIZE; WRITE_SIZE -- FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC
# slots") of
#   python3 bench.py --workload WORKLOAD --steps 5 --warmup 2 --no-cpu-baseline [args]
# Raw output goes to gpurun_out/TAG/; tools/pmc_summary.py condenses it into profiles/.
set -e -o pipefail
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$WL" = "cfd" ]; then      # the CFD kernel: tools/cfd_bench.py (50 M device-resident pairs)
  CMD="python3 tools/cfd_bench.py --rounds 5 --no-cpu $*"
else
  CMD="python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --extra none $*"
fi
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT/stats.log 2>&1
echo "[$TAG] stats pass done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
echo "[$TAG] SQ pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_wait -o p --output-format csv -- $CMD > $OUT/pmc_wait.log 2>&1
echo "[$TAG] wait-state pass done"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
echo "[$TAG] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
echo "[$TAG] WRITE_SIZE pass done"
tail -1 $OUT/stats.log | cut -c1-400
