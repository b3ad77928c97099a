#!/bin/bash
# tools/ab_plan_shapes.sh -- plan parameters at the MovieLens- / MIND- / Yahoo-B=N-shaped steps after round 6's kernel changes
# (write-through rows, weights by environment, records at the slot): do round 4-5's choices still hold?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $2 PROBE_SHAPE=$1 PROBE_STEPS=3 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/lanes [0-9]*, //; s/stream rows.*workgroups/wgs/' | tr '\n' ' ')
  echo "$1 [$2]: $r"
}
for shape in 6040,3706,8,128,65536 50000,51283,16,256,262144 50000,51283,16,256,32768 15400,1000,4,64,250154; do
  run $shape "X=1"
  for ps in 4 6 8 12 16 24 32 64; do run $shape "INVPREF_PLAN_PER_SLICE=$ps"; done
  for ips in 4 8 16 32 64; do run $shape "INVPREF_PLAN_ITEM_PER_SLICE=$ips"; done
  for rt in 1 2 4; do run $shape "INVPREF_PLAN_ROUNDS=$rt"; done
  for rt in 1 2 4 8; do run $shape "INVPREF_PLAN_ITEM_ROUNDS=$rt"; done
  run $shape "INVPREF_PLAN_PUSH=0"
  run $shape "INVPREF_PLAN_PUSH=1"
  run $shape "X=2"
done | tee gpurun_out/ab_plan_shapes.txt
