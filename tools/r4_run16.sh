export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d gpurun_out/r4_pmc16 -o p --output-format csv -- python3 tools/param_ab.py s1 pg_spec=0 > gpurun_out/r4_pmc16.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/r4_pmc16/**/p_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_merge_pg3' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        acc['dur'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
        acc['vgpr'].append(int(r['VGPR_Count'])); acc['lds'].append(int(r['LDS_Block_Size'])); acc['sgpr'].append(int(r['SGPR_Count']))
for k,v in acc.items(): print(k, sum(v)/len(v))
PY
rm -rf gpurun_out/r4_pmc16
