#!/bin/bash
# tools/sweep_mm.sh -- plan-parameter sweep of the D = 256 launches with the MFMA-classifier launch 1 (csrc/step_wide_mm.hpp)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/sweep_mm.log
run() { echo "== $*" >> gpurun_out/sweep_mm.log; env "$@" PROBE_STEPS=${PROBE_STEPS:-2} timeout 600 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230 >> gpurun_out/sweep_mm.log; }
for K in PROBE_SHAPE=50000x51283x16x256x262144 PROBE_SHAPE=400000x100000x16x256x16777216; do
  run $K
  for ps in 4 8 16 24 32; do run $K INVPREF_PLAN_PER_SLICE=$ps; done
  for r in 2 4 16 32; do run $K INVPREF_PLAN_ROUNDS=$r; done
done
cat gpurun_out/sweep_mm.log
