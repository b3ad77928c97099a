# rocprofv3 summaries for profiles/r04 (run on the GPU box through gpurun: timeout 2400 bash tools/profile_r04.sh)
# Every rocprofv3 run sits under its own `timeout`; counters are collected in their own passes (--kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
B="--steps 1550 --warmup 155 --no-cpu-baseline --no-extras"
ML=6040x3706x8x128x65536
MIND=50000x51283x16x256x262144
# per-kernel time of the bench command: the tracer sees the kernel nodes of the replayed graphs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
# the same command with every step issued eagerly
INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python3 $R/bench.py $B > $OUT/bench_eager_under_rocprof.json 2> $OUT/stats_eager.log
# L2 <-> fabric traffic of the step's two kernels (eager launches: one record per launch), and of the other shapes
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 155 --warmup 155 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_$n.log
  for shape in 400000x100000x4x64x1048576 400000x100000x8x128x1048576 400000x100000x16x256x1048576 $ML $MIND; do
    PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/large_pmc_${n}_$shape -- python3 $R/tools/step_probe.py > $OUT/large_${n}_$shape.log 2>&1
  done
done
# the Yahoo step's other counters
for c in "VALUBusy" "MemUnitStalled" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr"; do
  n=$(echo $c | tr ' ' '_')
  INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/extra_$n -- python3 $R/bench.py --steps 155 --warmup 155 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/extra_$n.log || echo "failed: $c"
done
# the wide-row instances (MovieLens: 16 lanes x 2 float4, E = 8; MIND: 32 lanes x 2, E = 16) and a cache-exceeding launch
for shape in $ML $MIND 400000x100000x16x256x1048576; do
  for c in "VALUBusy MemUnitStalled" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "MfmaUtil" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/wide_${shape}_$n -- python3 $R/tools/step_probe.py > $OUT/wide_${shape}_$n.log 2>&1 || echo "failed: $shape $c"
  done
done
# per-kernel durations at the other shapes
rm -f $R/gpurun_out/prof_probe.log
bash $R/tools/prof_probe.sh default,PROBE_SHAPE=$ML,PROBE_EAGER=1 default,PROBE_SHAPE=$MIND,PROBE_EAGER=1 default,PROBE_SHAPE=50000x51283x16x256x32768,PROBE_EAGER=1 default,PROBE_SHAPE=15400x1000x4x64x250154,PROBE_ZIPF=1,PROBE_EAGER=1 default,PROBE_SHAPE=400000x100000x4x64x1048576,PROBE_EAGER=1 default,PROBE_SHAPE=400000x100000x8x128x1048576,PROBE_EAGER=1 default,PROBE_SHAPE=400000x100000x16x256x1048576,PROBE_EAGER=1 > /dev/null 2>&1
cp $R/gpurun_out/prof_probe.log $OUT/other_shapes_kernel_durations.txt
# phase stamps of one full-size step, the loop benchmark, the wide shapes' stamps and loop figures, the E-step's tie-break path
PROBE_STAMPS=1 PROBE_TOP=6 timeout 120 python3 $R/tools/step_probe.py > $OUT/stamps_default_plan.txt 2>&1
timeout 120 python3 $R/tools/step_probe.py > $OUT/loop_benchmark.txt 2>&1
for shape in $ML $MIND; do
  PROBE_SHAPE=$shape timeout 200 python3 $R/tools/step_probe.py >> $OUT/loop_benchmark_wide.txt 2>&1
  PROBE_STAMPS=1 PROBE_SHAPE=$shape PROBE_STEPS=3 timeout 200 python3 $R/tools/step_probe.py >> $OUT/stamps_wide.txt 2>&1
done
timeout 200 python3 $R/tools/estep_rs.py > $OUT/estep_random_sort.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ['GRAFT_REPO_ROOT']; out=root+'/gpurun_out/prof_r04'
def summarise(dirs, dest, keep=lambda k: True, by_dir=False):
    with open(dest,'w') as f:
        f.write('kernel,counter,launches,mean_value\n')
        for d in dirs:
            fs=glob.glob(d+'/*/*counter_collection.csv')
            if not fs: f.write(f'"{os.path.basename(d)}: no counter file",,,\n'); continue
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(fs[0])):
                if keep(r['Kernel_Name']):
                    k=r['Kernel_Name']
                    tag=(os.path.basename(d)+' ' if by_dir else '')+(k.split('(')[1].split('::')[-1][:56] if k.startswith('void (') else k[:60])
                    acc[(tag, r['Counter_Name'])].append(float(r['Counter_Value']))
            for k,v in sorted(acc.items()):
                f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.1f}\n')
for n in ('fetch','write'):
    summarise([out+'/pmc_'+n], out+f'/pmc_{n}_summary.csv', keep=lambda k: 'mstep_' in k or 'estep' in k)
    summarise(sorted(d for d in glob.glob(out+f'/large_pmc_{n}_*') if os.path.isdir(d)), out+f'/large_pmc_{n}_summary.csv', keep=lambda k: 'mstep_' in k, by_dir=True)
summarise(sorted(d for d in glob.glob(out+'/extra_*') if os.path.isdir(d)), out+'/pmc_extra_summary.csv', keep=lambda k: 'mstep_' in k or 'estep' in k)
summarise(sorted(d for d in glob.glob(out+'/wide_*') if os.path.isdir(d)), out+'/pmc_wide_instances_summary.csv', keep=lambda k: 'mstep_' in k, by_dir=True)
for src,dst in (('stats','rocprofv3_kernel_stats_bench_graph.csv'),('stats_eager','rocprofv3_kernel_stats_bench_eager_steps.csv')):
    fs=glob.glob(out+'/'+src+'/*/*kernel_stats.csv')
    if fs: open(out+'/'+dst,'w').write(open(fs[0]).read())
PY
ls $OUT | head -60
