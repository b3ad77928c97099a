#!/bin/bash
# tools/sweep_plan.sh -- plan-parameter sweeps at the cache-exceeding shapes and MIND (one box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/sweep_plan.log
run() { echo "== $*" >> gpurun_out/sweep_plan.log; env "$@" timeout 600 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230 >> gpurun_out/sweep_plan.log; }
for s in 400000x100000x4x64x4194304 400000x100000x8x128x2097152 400000x100000x16x256x1048576 50000x51283x16x256x262144; do
  S="PROBE_SHAPE=$s PROBE_STEPS=2"
  run $S
  for t in 768 1024 2304 3072; do run $S INVPREF_PLAN_TARGET_WGS=$t; done
  for ips in 8 16 32; do run $S INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
  for ir in 1 2 4 8 16; do run $S INVPREF_PLAN_ITEM_ROUNDS=$ir; done
  for sr in 8 16 64; do run $S INVPREF_PLAN_STREAM_ROWS=$sr; done
done
cat gpurun_out/sweep_plan.log
