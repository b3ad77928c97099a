# tools/pmc_extra.sh -- a few more counters for the step's two kernels (GPU box; separate --pmc passes, eager launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_extra
mkdir -p $OUT
for c in "L2CacheHit" "MemUnitBusy" "MemUnitStalled" "WriteUnitStalled" "VALUBusy" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr" "TCC_EA_ATOMIC_sum"; do
  n=$(echo $c | tr ' ' '_')
  INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- python3 $R/bench.py --steps 155 --warmup 155 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/$n.log || echo "failed: $c"
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_extra'
with open(out+'/summary.csv','w') as f:
    f.write('kernel,counter,launches,mean_value\n')
    for d in sorted(glob.glob(out+'/*/')):
        fs=glob.glob(d+'*/*counter_collection.csv')
        if not fs: continue
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if 'rows' in r['Kernel_Name']:
                acc[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(acc.items()):
            f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.3f}\n')
print(open(out+'/summary.csv').read())
PY
