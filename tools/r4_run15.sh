export TMPDIR=/tmp
for t in 8 16 24; do
GMHIP_LIB=$PWD/guidemaker_amd/libgmhip_t$t.so timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_st15_$t -o p --output-format csv -- python3 tools/param_ab.py s1 pg_spec=0 > gpurun_out/r4_st15_$t.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r4_st15_$t/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_merge_pg3' in r['Name']: print("timing $t: merge_pg3 avg_us %.1f calls %s"%(float(r['AverageNs'])/1e3, r['Calls']))
PY
rm -rf gpurun_out/r4_st15_$t
done
