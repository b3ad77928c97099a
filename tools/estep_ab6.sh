#!/bin/bash
# tools/estep_ab6.sh -- round 6: per-dispatch kernel durations of the E-step cycle at the Yahoo shape, the fused form (ONE launch:
# estep_assign_kernel with the stat_envs fold as its epilogue) against the two-launch form (INVPREF_ESTEP_FUSED=0:
# estep_assign_kernel + stat_envs_kernel), plain and with the reference's default tie-break
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/estep_ab6; rm -rf $O; mkdir -p $O
for fused in 1 k 0; do
  export INVPREF_ESTEP_FUSED=$fused
  unset INVPREF_ESTEP_FOLD
  [ $fused = k ] && export INVPREF_ESTEP_FUSED=1 INVPREF_ESTEP_FOLD=kernel
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/f$fused -- python3 $R/tools/estep_prof.py > $O/f$fused.log 2>&1
  f=$(ls $O/f$fused/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== INVPREF_ESTEP_FUSED=$fused (k: fused entry point, the fold as a second one-workgroup kernel)" >> $O/summary.txt
  [ -n "$f" ] && python3 - "$f" >> $O/summary.txt <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for kn in ('estep_assign_kernel', 'estep_fold_kernel', 'stat_envs_kernel', 'sample_weights_kernel'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if kn in r['Kernel_Name']]
    if not d:
        continue
    h = len(d) // 2
    for name, part in (('plain', d[:h]), ('tie-break', d[h:])):
        print('  %-22s %-10s n %3d  median %7.2f us  mean %7.2f  min %7.2f  max %7.2f' % (kn, name, len(part), statistics.median(part), statistics.mean(part), min(part), max(part)))
PY
  rm -rf $O/f$fused
done
unset INVPREF_ESTEP_FUSED INVPREF_ESTEP_FOLD
cat $O/summary.txt
