set -o pipefail
export TMPDIR=/tmp
for h in 1 0; do
GM_PG_HIT=$h timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_st7_$h -o p --output-format csv -- python3 tools/param_ab.py pao1 pg_hit=$h > gpurun_out/r4_st7_$h.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r4_st7_$h/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== pg_hit=$h")
for r in rows[:14]:
    print("%-70s calls %6s avg_us %9.1f pct %5s"%(r['Name'][:70],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
done
