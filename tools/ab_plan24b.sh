#!/bin/bash
# tools/ab_plan24b.sh -- user-slice length at the cache-exceeding launches, finer (see ab_plan24.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $2 PROBE_SHAPE=$1 PROBE_STEPS=1 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/lanes 16, //; s/stream rows.*workgroups/wgs/' | tr '\n' ' ')
  echo "$1 [$2]: $r"
}
for shape in 400000,100000,4,64,16777216 400000,100000,8,128,16777216; do
  for ps in 40 48 56 64 96; do
    for rt in 4 8; do run $shape "INVPREF_PLAN_PER_SLICE=$ps INVPREF_PLAN_ROUNDS=$rt"; done
  done
  run $shape "INVPREF_PLAN_PER_SLICE=64 INVPREF_PLAN_ROUNDS=4 INVPREF_PLAN_ITEM_ROUNDS=2"
  run $shape "INVPREF_PLAN_PER_SLICE=64 INVPREF_PLAN_ROUNDS=2"
done | tee gpurun_out/ab_plan24b.txt
for shape in 400000,100000,16,256,16777216; do
  for ps in 24 48 64; do run $shape "INVPREF_PLAN_PER_SLICE=$ps"; done
done | tee -a gpurun_out/ab_plan24b.txt
for shape in 400000,100000,4,64,4194304 400000,100000,8,128,2097152 400000,100000,16,256,1048576; do
  for ps in 0 8 12 16 24 32; do
    if [ $ps = 0 ]; then run $shape "X=1"; else run $shape "INVPREF_PLAN_PER_SLICE=$ps"; fi
  done
done | tee -a gpurun_out/ab_plan24b.txt
