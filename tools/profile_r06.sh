# rocprofv3 summaries for profiles/r06 (GPU box, through gpurun: timeout 2400 bash tools/profile_r06.sh)
# Every rocprofv3 run sits under its own `timeout`; counters are collected in their own passes (--kernel-trace + --pmc only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
B="--steps 1550 --warmup 155 --no-cpu-baseline --no-extras"
BQ="--steps 155 --warmup 155 --no-cpu-baseline --no-extras --no-parity-gate"   # counter passes: no gate epoch in the counters
# per-kernel time of the bench command: the tracer sees the kernel nodes of the replayed graphs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
# the same command with every launch issued eagerly (the alternating launches, eager)
INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager -- python3 $R/bench.py $B > $OUT/bench_eager_under_rocprof.json 2> $OUT/stats_eager.log
# the two-launch form (INVPREF_ALT=0), same command: the A/B of the round
INVPREF_ALT=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_two_launch -- python3 $R/bench.py $B > $OUT/bench_two_launch_under_rocprof.json 2> $OUT/stats_two_launch.log
# L2 <-> fabric traffic of the step's kernel (eager launches: one record per launch)
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | cut -d_ -f1 | tr A-Z a-z)
  INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py $BQ > /dev/null 2> $OUT/pmc_$n.log
  # SURVEY 8(d)'s roofline launch at ITS size: 2^24 interactions, D = 64 (the two-launch form: the shape is not Yahoo-class ... it is, E = 4: both forms exist; the bench's sweep runs the two-launch form)
  PROBE_EAGER=1 PROBE_SHAPE=400000x100000x4x64x16777216 PROBE_STEPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/large24_pmc_${n}_D64 -- python3 $R/tools/step_probe.py > $OUT/large24_${n}.log 2>&1
done
PROBE_EAGER=1 PROBE_SHAPE=400000x100000x4x64x16777216 PROBE_STEPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/large24_stats_D64 -- python3 $R/tools/step_probe.py > $OUT/large24_stats.log 2>&1
# the alternating launch's other counters
for c in "VALUBusy" "MemUnitStalled" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_')
  INVPREF_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/extra_$n -- python3 $R/bench.py $BQ > /dev/null 2> $OUT/extra_$n.log || echo "failed: $c"
done
# the E-step kernel to kernel: the fused form (one launch, epilogue fold) vs the fold as a second kernel vs the two-launch form
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/estep_stats -- python3 $R/tools/estep_prof.py > $OUT/estep_prof.log 2>&1
(cd $R && bash tools/estep_ab6.sh) > $OUT/estep_kernel_ab.txt 2>&1
(cd $R && bash tools/ab_estep_blocks.sh) > $OUT/estep_blocks.txt 2>&1
# evaluation at the three implicit test shapes; where evaluate() spends its time at the MIND shape
timeout 300 python3 $R/tools/eval_timing.py > $OUT/eval_timing.json 2> /dev/null
timeout 300 python3 $R/tools/eval_phases.py > $OUT/eval_phases_mind.txt 2>&1
# the alternating form against the two-launch form: parity + loop figures, launches by side, phase stamps
timeout 200 python3 $R/tools/alt_probe.py > $OUT/alt_probe.txt 2>&1
(cd $R && bash tools/alt_trace.sh) > $OUT/alt_launches_by_side.txt 2>&1
(cd $R && python3 -c "from invpref_kdd_2022_amd import build; print(build.build_variant('altstamps', ['-DALT_STAMPS']))") > $OUT/altstamps_build.log 2>&1
INVPREF_LIB=$R/invpref_kdd_2022_amd/variants/altstamps.so PROBE_STAMPS=1 PROBE_TOP=6 timeout 200 python3 $R/tools/alt_probe.py > $OUT/alt_stamps.txt 2>&1
timeout 120 python3 $R/tools/step_probe.py > $OUT/loop_benchmark_two_launch.txt 2>&1
SOAK_BENCH=1 SOAK_INTERVALS=170 timeout 600 python3 $R/tools/alt_soak.py > $OUT/alt_soak.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections, re
root=os.environ['GRAFT_REPO_ROOT']; out=root+'/gpurun_out/prof_r06'
def tag_of(k):
    m=re.search(r'((?:mstep|estep|stat_envs|eps_unrank)\w*)(<[^>]*>)?', k)
    return (m.group(1)+(m.group(2) or '')) if m else k[:60]
def summarise(dirs, dest, keep=lambda k: True, prefix=lambda d: ''):
    with open(dest,'w') as f:
        f.write('kernel,counter,launches,mean_value\n')
        for d in dirs:
            fs=glob.glob(d+'/*/*counter_collection.csv')
            if not fs: f.write(f'"{os.path.basename(d)}: no counter file",,,\n'); continue
            acc=collections.defaultdict(list)
            for r in csv.DictReader(open(fs[0])):
                if keep(r['Kernel_Name']):
                    acc[(prefix(d)+tag_of(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
            for k,v in sorted(acc.items()):
                f.write(f'"{k[0]}",{k[1]},{len(v)},{sum(v)/len(v):.1f}\n')
for n in ('fetch','write'):
    summarise([out+'/pmc_'+n], out+f'/alt_pmc_{n}_summary.csv', keep=lambda k: 'mstep_' in k or 'estep' in k)
summarise([out+'/large24_pmc_fetch_D64', out+'/large24_pmc_write_D64'], out+'/large24_pmc_summary.csv', keep=lambda k: 'mstep_' in k, prefix=lambda d: 'D64 ')
summarise(sorted(d for d in glob.glob(out+'/extra_*') if os.path.isdir(d)), out+'/pmc_extra_summary.csv', keep=lambda k: 'mstep_' in k or 'estep' in k)
for src,dst in (('stats','rocprofv3_kernel_stats_bench_graph.csv'),('stats_eager','rocprofv3_kernel_stats_bench_eager_steps.csv'),
                ('stats_two_launch','rocprofv3_kernel_stats_bench_two_launch_form.csv'),('estep_stats','rocprofv3_kernel_stats_estep.csv'),
('large24_stats_D64','rocprofv3_kernel_stats_large24_D64.csv')):
    fs=glob.glob(out+'/'+src+'/*/*kernel_stats.csv')
    if fs: open(out+'/'+dst,'w').write(open(fs[0]).read())
fs=glob.glob(out+'/large24_stats_D64/*/*kernel_stats.csv')
if fs:
    with open(out+'/large24_kernel_durations.csv','w') as f:
        f.write('shape,kernel,calls,mean_us\n')
        for r in csv.DictReader(open(fs[0])):
            if 'mstep_' in r['Name']:
                f.write(f'D64,"{tag_of(r["Name"])}",{r["Calls"]},{float(r["AverageNs"])/1e3:.2f}\n')
PY
ls $OUT | head -80
# the raw rocprofv3 output directories stay on the box (gpurun merges at most 64 MiB back): the summaries above are what is kept
for d in stats stats_eager stats_two_launch pmc_fetch pmc_write large24_pmc_fetch_D64 large24_pmc_write_D64 large24_stats_D64 estep_stats; do rm -rf $OUT/$d; done
rm -rf $OUT/extra_*/ 2>/dev/null
(cd $R && git rev-parse HEAD 2>/dev/null || true) > /dev/null
du -sh $OUT
