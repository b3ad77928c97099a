#!/bin/bash
# tools/estep_ab.sh -- kernel-to-kernel: estep_assign_kernel plain vs with the reference's default tie-break (per-dispatch
# durations of tools/estep_prof.py's replays: the first manager's are the plain ones), pinned-host indices vs copied first
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/estep_ab; rm -rf $O; mkdir -p $O
for mode in pinned copy; do
  [ $mode = copy ] && export INVPREF_EPS_PINNED=0 || unset INVPREF_EPS_PINNED
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -- python3 $R/tools/estep_prof.py > $O/$mode.log 2>&1
  f=$(ls $O/$mode/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== $mode" >> $O/summary.txt
  [ -n "$f" ] && python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'estep_assign_kernel' in r['Kernel_Name']]
h = len(d) // 2
for name, part in (('plain', d[:h]), ('tie-break', d[h:])):
    print('  %-10s n %3d  median %7.2f us  mean %7.2f  min %7.2f  max %7.2f' % (name, len(part), statistics.median(part), statistics.mean(part), min(part), max(part)))
cp = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'copyBuffer' in r['Kernel_Name']]
print('  (copy kernels in the trace: %d, median %.2f us)' % (len(cp), statistics.median(cp) if cp else 0))
PY
  rm -rf $O/$mode
done
timeout 200 python3 $R/tools/estep_rs.py >> $O/summary.txt 2>&1
cat $O/summary.txt
