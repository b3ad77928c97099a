#!/bin/bash
# tools/ab_pushpull.sh -- push vs pull form of the two-launch step at Yahoo-class tables by minibatch size, and at the MovieLens
# shape with longer item slices (round 6: the pull records sit at the slot, launch 2 reads them front to back)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $2 PROBE_SHAPE=$1 PROBE_STEPS=3 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/lanes [0-9]*, //; s/stream rows.*workgroups/wgs/' | tr '\n' ' ')
  echo "$1 [$2]: $r"
}
for b in 1024 4096 8192 32768 65536 131072 250154; do
  for p in 1 0; do run 15400,1000,4,64,$b "INVPREF_PLAN_PUSH=$p"; done
done | tee gpurun_out/ab_pushpull.txt
for ips in 2 4 8 16 32; do run 6040,3706,8,128,65536 "INVPREF_PLAN_PUSH=0 INVPREF_PLAN_ITEM_PER_SLICE=$ips"; done | tee -a gpurun_out/ab_pushpull.txt
for b in 16384 65536; do for p in 1 0; do run 6040,3706,4,64,$b "INVPREF_PLAN_PUSH=$p"; done; done | tee -a gpurun_out/ab_pushpull.txt
for shape in 400000,100000,4,64,1048576 400000,100000,8,128,1048576 100000,20000,4,64,1048576; do for p in 1 0; do run $shape "INVPREF_PLAN_PUSH=$p"; done; run $shape "X=1"; done | tee -a gpurun_out/ab_pushpull.txt
