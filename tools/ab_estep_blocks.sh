#!/bin/bash
# tools/ab_estep_blocks.sh -- round 6: the fused E-step kernel's duration against its workgroup count (INVPREF_ESTEP_BLOCKS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/estep_blocks; rm -rf $O; mkdir -p $O
for nb in 2048 1536 1024 768 512; do
  export INVPREF_ESTEP_BLOCKS=$nb
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/b$nb -- python3 $R/tools/estep_prof.py > $O/b$nb.log 2>&1
  f=$(ls $O/b$nb/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" $nb >> $O/summary.txt <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'estep_assign_kernel' in r['Kernel_Name']]
h = len(d) // 2
print('blocks %5s  plain median %7.2f us   tie-break median %7.2f us' % (sys.argv[2], statistics.median(d[:h]), statistics.median(d[h:])))
PY
  rm -rf $O/b$nb
done
cat $O/summary.txt
