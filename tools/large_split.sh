#!/bin/bash
# tools/large_split.sh -- rocprofv3 durations of the two launches of the 2^24-interaction step at D = 64 / 128 / 256 (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in 400000x100000x4x64x16777216 400000x100000x8x128x16777216 400000x100000x16x256x16777216; do
  O=$R/gpurun_out/large_split_$shape
  rm -rf $O
  PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/step_probe.py > /dev/null 2>&1
  echo "== $shape"
  python3 - "$O" <<'PY'
import csv, glob, sys
fs = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')
for r in csv.DictReader(open(fs[0])):
    if 'mstep_' in r['Name']:
        print('  %-60s calls %3s  mean %9.1f us' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $O
done | tee $R/gpurun_out/large_split.txt
