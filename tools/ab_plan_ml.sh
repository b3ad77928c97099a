#!/bin/bash
# tools/ab_plan_ml.sh -- the remaining plan knobs at the MovieLens- and MIND-shaped steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $2 PROBE_SHAPE=$1 PROBE_STEPS=3 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/lanes [0-9]*, //' | tr '\n' ' ')
  echo "$1 [$2]: $r"
}
for shape in 6040,3706,8,128,65536 50000,51283,16,256,262144; do
  run $shape "X=1"
  for v in "INVPREF_PLAN_STREAM_SPLIT=0" "INVPREF_PLAN_STREAM_SPLIT=0.25" "INVPREF_PLAN_STREAM_SPLIT=0.5" "INVPREF_PLAN_STREAM_SPLIT=0.75" "INVPREF_PLAN_STREAM_SPLIT=1" \
           "INVPREF_PLAN_STREAM_ROWS=8" "INVPREF_PLAN_STREAM_ROWS=16" "INVPREF_PLAN_STREAM_ROWS=64" "INVPREF_PLAN_STREAM_ROWS2=8" "INVPREF_PLAN_STREAM_ROWS2=64" \
           "INVPREF_PLAN_SNAKE=0" "INVPREF_PLAN_SNAKE=16" "INVPREF_PLAN_SNAKE=64" "INVPREF_PLAN_PER_SLICE=10" "INVPREF_PLAN_PER_SLICE=14" \
           "INVPREF_PLAN_CLASSES=1" "INVPREF_PLAN_TARGET_WGS=1024" "INVPREF_PLAN_TARGET_WGS=2048" "INVPREF_PLAN_EVAL_COST=2" "INVPREF_PLAN_EVAL_COST=4"; do
    run $shape "$v"
  done
  run $shape "X=2"
done | tee gpurun_out/ab_plan_ml.txt
