#!/bin/bash
# tools/large_pmc.sh -- what bounds launch 1 of the 2^24-interaction step: VALU busy share and instruction counts (GPU box; counters
# in passes of their own, --kernel-trace + --pmc only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in 400000x100000x4x64x16777216 400000x100000x8x128x16777216; do
  for c in "VALUBusy" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_')
    O=$R/gpurun_out/large_pmc_${shape}_$n
    rm -rf $O
    PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -- python3 $R/tools/step_probe.py > /dev/null 2>&1
    python3 - "$O" "$shape" <<'PY'
import csv, glob, sys, collections, re
fs = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')
if not fs:
    print(sys.argv[2], 'no counter file'); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'mstep_' in r['Kernel_Name']:
        acc[(re.search(r'mstep_\w+', r['Kernel_Name']).group(0), r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%s  %-48s %-18s n %d mean %.4g' % (sys.argv[2], k[0], k[1], len(v), sum(v) / len(v)))
PY
    rm -rf $O
  done
done | tee $R/gpurun_out/large_pmc.txt
