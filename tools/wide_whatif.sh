#!/bin/bash
# tools/wide_whatif.sh "VARIANT..." SHAPE... -- rocprofv3 per-kernel durations of the eager step probe for each library
# variant (tools/build_variant.sh; "default" = the shipped library) at each shape: what a what-if build moves, per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wide_whatif; rm -rf $O; mkdir -p $O
variants=$1; shift
for shape in "$@"; do
  for spec in $variants; do   # VARIANT[,ENV=VALUE...]
    IFS=',' read -ra parts <<< "$spec"
    v=$spec
    lib=$R/invpref_kdd_2022_amd/variants/${parts[0]}.so
    [ "${parts[0]}" = default ] && lib=$R/invpref_kdd_2022_amd/libinvpref_hip.so
    export INVPREF_LIB=$lib PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=${PROBE_STEPS:-4}
    for kv in "${prev_kv[@]}"; do unset "${kv%%=*}"; done   # (the previous spec's switches do not leak into this one)
    prev_kv=("${parts[@]:1}")
    for kv in "${parts[@]:1}"; do export "$kv"; done
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v-$shape -- python3 $R/tools/step_probe.py > $O/$v-$shape.log 2>&1
    f=$(ls $O/$v-$shape/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $v $shape" >> $O/summary.txt
    grep -a "us per step\|per step" $O/$v-$shape.log | tail -2 | cut -c1-200 >> $O/summary.txt
    [ -n "$f" ] && python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mstep' in r['Name']:
        print('  %-70s calls %4s avg %9.2f us  min %9.2f  max %9.2f' % (r['Name'][:70].replace('void (anonymous namespace)::',''), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
    rm -rf $O/$v-$shape
  done
done
cat $O/summary.txt
