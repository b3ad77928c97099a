#!/bin/bash
# tools/alt_trace.sh [variant.so ...] -- per-launch durations of the alternating form under rocprofv3 (kernel trace of
# tools/alt_probe.py's graph replays), user-side and item-side launches apart (they differ in grid size).  GPU box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in default "$@"; do
  out=gpurun_out/alt_trace_$(basename $lib .so)
  rm -rf $out; mkdir -p $out
  if [ "$lib" = default ]; then unset INVPREF_LIB; else export INVPREF_LIB=$PWD/invpref_kdd_2022_amd/variants/$lib.so; fi
  PROBE_REPS=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/alt_probe.py > $out/probe.log 2>&1
  python3 - "$out" "$lib" <<'PY'
import csv, glob, sys, collections
out, lib = sys.argv[1], sys.argv[2]
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    import re
    m = re.search(r'(mstep_\w+)(<[^>]*>)?', r['Kernel_Name'])
    if not m: continue
    short = m.group(1) + (m.group(2) or '')
    grid = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    agg[(short, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f'== {lib}')
for (k, g), v in sorted(agg.items(), key=lambda kv: -len(kv[1]))[:12]:
    v = sorted(v); n = len(v)
    print(f'  {k:45s} grid {g:5d}  n {n:6d}  median {v[n // 2]:7.2f} us  mean {sum(v) / n:7.2f}  p90 {v[int(n * .9)]:7.2f}')
PY
done
