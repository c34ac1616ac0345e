set -o pipefail
export TMPDIR=/tmp
for w in pao1 s1; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r4_st13_$w -o p --output-format csv -- python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --extra none > gpurun_out/r4_st13_$w.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/r4_st13_$w/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== $w")
for r in rows[:16]:
    print("%-64s calls %5s avg_us %9.1f pct %5s"%(r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:64],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage']))
PY
rm -rf gpurun_out/r4_st13_$w
done
