cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof5 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 620 --warmup 62 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python3 - <<'PY'
import csv, glob, os
f=sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof5/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'], r['Percentage'])
t=sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof5/*/*kernel_trace.csv'))[-1]
rows=sorted(csv.DictReader(open(t)), key=lambda r:int(r['Start_Timestamp']))
rk=[i for i,r in enumerate(rows) if 'mstep_rows' in r['Kernel_Name']]
import statistics
gaps1=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in rk[100:600] if 'rows_finish' in rows[i+1]['Kernel_Name']]
gaps2=[int(rows[i]['Start_Timestamp'])-int(rows[i-1]['End_Timestamp']) for i in rk[100:600] if 'rows_finish' in rows[i-1]['Kernel_Name']]
print('gap rows->finish med', statistics.median(gaps1), 'gap finish->rows med', statistics.median(gaps2))
per=[int(rows[rk[j+1]]['Start_Timestamp'])-int(rows[rk[j]]['Start_Timestamp']) for j in range(100,600)]
print('step period med', statistics.median(per))
PY
