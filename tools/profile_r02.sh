# rocprofv3 summaries for profiles/r02 (run on the GPU box through gpurun: bash tools/profile_r02.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
B="--steps 1550 --warmup 155 --no-cpu-baseline --no-extras"
# per-kernel time of the bench command (graph replays are opaque to the tracer: what it sees are the eagerly issued
# epochs -- set-up, event-timed epoch -- and the E-steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
# the same command with every step issued eagerly, so that each launch is traced
INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python3 $R/bench.py $B > $OUT/bench_eager_under_rocprof.json 2> $OUT/stats_eager.log
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  INVPREF_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 155 --warmup 155 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_$n.log
  ROOFLINE_REPS=2 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/large_pmc_$n -- python3 $R/tools/roofline_large.py > $OUT/large_under_pmc_$n.jsonl 2> $OUT/large_pmc_$n.log
done
python3 $R/tools/roofline_large.py > $OUT/roofline_large_launch.jsonl 2> $OUT/roofline_large.err
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_r02'
for name in ('pmc_fetch','pmc_write','large_pmc_fetch','large_pmc_write'):
    fs=glob.glob(out+'/'+name+'/*/*counter_collection.csv')
    if not fs: print(name,'no counter file', glob.glob(out+'/'+name+'/*/*')); continue
    rows=list(csv.DictReader(open(fs[0])))
    d=collections.defaultdict(list)
    for r in rows:
        key=(r['Kernel_Name'][:60], r['Counter_Name'])
        if name.startswith('large'):   # one line per launch shape: grid size tells the points apart
            key=(r['Kernel_Name'][:60]+' grid='+r.get('Grid_Size', r.get('Workgroup_Size','?')), r['Counter_Name'])
        d[key].append(float(r['Counter_Value']))
    with open(out+'/'+name+'_summary.csv','w') as f:
        f.write('kernel,counter,launches,mean_value\n')
        for k,v in sorted(d.items()):
            f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.1f}\n')
PY
ls $OUT
