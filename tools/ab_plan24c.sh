#!/bin/bash
# tools/ab_plan24c.sh -- rounds per task around the new slice length at 2^24 (see ab_plan24b.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $2 PROBE_SHAPE=$1 PROBE_STEPS=1 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/stream rows.*workgroups/wgs/' | tr '\n' ' ')
  echo "$1 [$2]: $r"
}
for rt in 2 4 8 16; do run 400000,100000,16,256,16777216 "INVPREF_PLAN_ROUNDS=$rt"; done | tee gpurun_out/ab_plan24c.txt
for rt in 4 8; do run 400000,100000,16,256,16777216 "INVPREF_PLAN_ITEM_ROUNDS=$rt"; done | tee -a gpurun_out/ab_plan24c.txt
for shape in 400000,100000,4,64,16777216 400000,100000,8,128,16777216; do
  for rt in 2 3 4 6; do run $shape "INVPREF_PLAN_ROUNDS=$rt"; done
  for rt in 2 4; do run $shape "INVPREF_PLAN_ITEM_ROUNDS=$rt"; done
  run $shape "INVPREF_PLAN_ITEM_PER_SLICE=64 INVPREF_PLAN_ITEM_ROUNDS=2"
done | tee -a gpurun_out/ab_plan24c.txt
