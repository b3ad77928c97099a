#!/bin/bash
# tools/pmc_ab.sh VARIANT... -- fabric traffic (WRITE_SIZE, FETCH_SIZE; KiB, FETCH x 2 on gfx950) of the step's kernels per library
# variant (invpref_kdd_2022_amd/variants/NAME.so, "default" = the shipped one) at PROBE_SHAPE (default: the MIND shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${PROBE_SHAPE:-50000x51283x16x256x262144}
for v in "$@"; do
  lib=$R/invpref_kdd_2022_amd/variants/$v.so; [ $v = default ] && lib=$R/invpref_kdd_2022_amd/libinvpref_hip.so
  for c in ${COUNTERS:-WRITE_SIZE FETCH_SIZE}; do
    rm -rf /tmp/pm_${v}_$c
    INVPREF_LIB=$lib PROBE_EAGER=1 PROBE_SHAPE=$SHAPE PROBE_STEPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm_${v}_$c -- python3 $R/tools/step_probe.py > /dev/null 2>&1
    python3 - /tmp/pm_${v}_$c $v $c <<'PY'
import csv,glob,sys,collections,re
fs=glob.glob(sys.argv[1]+'/*/*counter_collection.csv'); acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'mstep_' in r['Kernel_Name']: acc[re.search(r'mstep_\w+', r['Kernel_Name']).group(0) + ' grid ' + r['Grid_Size']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print('%-12s %-11s %-42s n=%d mean %.1f' % (sys.argv[2],sys.argv[3],k,len(v),sum(v)/len(v)))
PY
  done
done
