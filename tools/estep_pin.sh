cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pin in 1 0; do
  export INVPREF_EPS_PINNED=$pin
  O=$R/gpurun_out/estep_pin$pin; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/estep_prof.py > $O.log 2>&1
  f=$(ls $O/*/*kernel_trace.csv | head -1)
  python3 - "$f" $pin <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'estep_assign_kernel' in r['Kernel_Name']]
h = len(d) // 2
print('PINNED=%s  plain median %.2f  tie-break median %.2f' % (sys.argv[2], statistics.median(d[:h]), statistics.median(d[h:])))
PY
  rm -rf $O
done
