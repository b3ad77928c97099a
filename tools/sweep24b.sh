#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/sweep24.log
run() { echo "== $*" >> gpurun_out/sweep24.log; env "$@" timeout 600 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230 >> gpurun_out/sweep24.log; }
for s in 400000x100000x16x256x16777216 400000x100000x8x128x16777216 400000x100000x4x64x16777216; do
  run PROBE_SHAPE=$s PROBE_STEPS=1 INVPREF_PLAN_PER_SLICE=24 INVPREF_PLAN_ROUNDS=8
  run PROBE_SHAPE=$s PROBE_STEPS=1 INVPREF_PLAN_PER_SLICE=20
  run PROBE_SHAPE=$s PROBE_STEPS=1 INVPREF_PLAN_PER_SLICE=28
done
for s in 400000x100000x16x256x1048576 400000x100000x8x128x2097152 400000x100000x4x64x4194304 400000x100000x8x128x1048576; do
  run PROBE_SHAPE=$s PROBE_STEPS=2
  run PROBE_SHAPE=$s PROBE_STEPS=2 INVPREF_PLAN_ROUNDS=8
  run PROBE_SHAPE=$s PROBE_STEPS=2 INVPREF_PLAN_ROUNDS=4
done
cat gpurun_out/sweep24.log
