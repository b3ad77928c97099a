#!/bin/bash
# tools/wide_pmc.sh SHAPE... -- PMC passes (one counter group per run) of the eager step probe at the given shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/wide_pmc; rm -rf $O; mkdir -p $O
for shape in "$@"; do
  for c in "VALUBusy MemUnitStalled" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "MfmaUtil" "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $c | tr ' ' '_')
    PROBE_EAGER=1 PROBE_SHAPE=$shape PROBE_STEPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${shape}_$n -- python3 $R/tools/step_probe.py > $O/${shape}_$n.log 2>&1 || echo "failed: $c" >> $O/summary.txt
  done
done
python3 - <<'PY' >> $O/summary.txt
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/wide_pmc'
for d in sorted(glob.glob(out+'/*/')):
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs: continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'mstep_' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('::')[-1][:44]
            acc[(k, r['Counter_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        print('%-60s %-46s grid %-8s %-22s n=%d mean %.1f' % (os.path.basename(d.rstrip('/'))[:60], k[0], k[2], k[1], len(v), sum(v)/len(v)))
PY
cat $O/summary.txt
