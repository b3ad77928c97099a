# extra PMC passes for profiles/ (one derived metric per pass; eager launches so that every kernel is seen)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_extra
mkdir -p $OUT
for m in LDSBankConflict VALUBusy MemUnitBusy MemUnitStalled L2CacheHit OccupancyPercent; do
  INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $m --output-format csv -d $OUT/$m -- python3 $R/bench.py --steps 62 --warmup 31 --no-cpu-baseline > /dev/null 2> $OUT/$m.log
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_extra'
with open(out+'/pmc_extra_summary.csv','w') as f:
    f.write('kernel,counter,launches,mean_value\n')
    for d in sorted(glob.glob(out+'/*/')):
        fs=glob.glob(d+'*/*counter_collection.csv')
        if not fs: print('no counters in', d); continue
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if 'rows' in r['Kernel_Name'] or 'estep' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.3f}\n'); print(k, len(v), round(sum(v)/len(v),3))
PY
