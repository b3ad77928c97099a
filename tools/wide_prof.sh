#!/bin/bash
# tools/wide_prof.sh SHAPE... -- rocprofv3 per-kernel durations of the eager step probe at the given shapes + phase stamps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wide_prof; rm -rf $O; mkdir -p $O
for shape in "$@"; do
  PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$shape -- python3 $R/tools/step_probe.py > $O/$shape.log 2>&1
  f=$(ls $O/$shape/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $shape" >> $O/summary.txt
  [ -n "$f" ] && python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mstep' in r['Name']:
        print('  %-70s calls %4s avg %9.2f us  min %9.2f  max %9.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  PROBE_STAMPS=1 PROBE_SHAPE=$shape PROBE_STEPS=3 timeout 300 python3 $R/tools/step_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "^shape" >> $O/summary.txt
done
cat $O/summary.txt
