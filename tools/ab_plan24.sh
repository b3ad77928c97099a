#!/bin/bash
# tools/ab_plan24.sh -- plan parameters of the 2^24-interaction launch pair after the slot-ordered records (launch 2 got cheaper:
# does the split between the launches still sit where round 5's sweep left it?)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for shape in 400000,100000,4,64,16777216 400000,100000,8,128,16777216; do
  for v in "X=1" "INVPREF_PLAN_PER_SLICE=12" "INVPREF_PLAN_PER_SLICE=16" "INVPREF_PLAN_PER_SLICE=32" "INVPREF_PLAN_PER_SLICE=48" \
           "INVPREF_PLAN_ITEM_PER_SLICE=8" "INVPREF_PLAN_ITEM_PER_SLICE=16" "INVPREF_PLAN_ITEM_PER_SLICE=32" "INVPREF_PLAN_ITEM_PER_SLICE=64" \
           "INVPREF_PLAN_ROUNDS=2" "INVPREF_PLAN_ROUNDS=4" "INVPREF_PLAN_ROUNDS=16" "INVPREF_PLAN_ITEM_ROUNDS=2" "INVPREF_PLAN_ITEM_ROUNDS=8" \
           "INVPREF_PLAN_STREAM_SPLIT=0.3" "INVPREF_PLAN_STREAM_SPLIT=0.7" "INVPREF_PLAN_PUSH=1"; do
    r=$(env $v PROBE_SHAPE=$shape PROBE_STEPS=1 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | tr '\n' ' ')
    echo "$shape [$v]: $r"
  done
done | tee gpurun_out/ab_plan24.txt
