#!/bin/bash
# tools/prof_probe.sh SPEC... -- rocprofv3 per-kernel durations of tools/step_probe.py once per SPEC (see tools/ab.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for spec in "$@"; do
  IFS=',' read -ra parts <<< "$spec"
  v=${parts[0]}
  lib=$R/invpref_kdd_2022_amd/variants/$v.so
  [ "$v" = default ] && lib=$R/invpref_kdd_2022_amd/libinvpref_hip.so
  d=/tmp/prof_$$_$RANDOM
  env INVPREF_LIB=$lib "${parts[@]:1}" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/step_probe.py > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" 2>/dev/null | head -1)
  echo "== $spec" >> $R/gpurun_out/prof_probe.log
  if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'mstep_' in n or 'estep' in n or 'adam' in n:
        print('  %-34s calls %6s avg %8.2f us  min %8.2f  max %8.2f' % (n.split('(')[1].split('::')[-1][:34] if n.startswith('void (') else n[:34], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
" >> $R/gpurun_out/prof_probe.log; else echo "no stats file" >> $R/gpurun_out/prof_probe.log; fi
done
cat $R/gpurun_out/prof_probe.log
