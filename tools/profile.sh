# rocprofv3 summaries for profiles/ (run on the GPU box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 310 --warmup 31 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 62 --warmup 31 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.log
INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 62 --warmup 31 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.log
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_r01'
for name in ('pmc_fetch','pmc_write'):
    fs=glob.glob(out+'/'+name+'/*/*counter_collection.csv')
    if not fs: print(name,'no counter file', glob.glob(out+'/'+name+'/*/*')); continue
    rows=list(csv.DictReader(open(fs[0])))
    d=collections.defaultdict(list)
    for r in rows: d[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
    with open(out+'/'+name+'_summary.csv','w') as f:
        f.write('kernel,counter,launches,mean_value\n')
        for k,v in sorted(d.items()):
            f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.1f}\n')
            print(name, k, len(v), sum(v)/len(v))
PY
ls $OUT $OUT/stats/* | head -30
