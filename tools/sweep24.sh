#!/bin/bash
# tools/sweep24.sh -- plan-parameter sweep of the 2^24-interaction launches (cheap since the native plan builder)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/sweep24.log
run() { echo "== $*" >> gpurun_out/sweep24.log; env "$@" PROBE_STEPS=${PROBE_STEPS:-1} timeout 600 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230 >> gpurun_out/sweep24.log; }
K=PROBE_SHAPE=400000x100000x16x256x16777216
run $K
for ps in 8 12 24 32; do run $K INVPREF_PLAN_PER_SLICE=$ps; done
for r in 2 4 8; do run $K INVPREF_PLAN_ROUNDS=$r; done
for ir in 1 4 16; do run $K INVPREF_PLAN_ITEM_ROUNDS=$ir; done
for ips in 8 16 32; do run $K INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
L=PROBE_SHAPE=400000x100000x8x128x16777216
run $L
for ps in 8 12 24 32; do run $L INVPREF_PLAN_PER_SLICE=$ps; done
for r in 2 4 8 16; do run $L INVPREF_PLAN_ROUNDS=$r; done
run $L INVPREF_PLAN_PUSH=1
J=PROBE_SHAPE=400000x100000x4x64x16777216
run $J
for ps in 8 12 24 32; do run $J INVPREF_PLAN_PER_SLICE=$ps; done
for r in 2 4 8 16; do run $J INVPREF_PLAN_ROUNDS=$r; done
cat gpurun_out/sweep24.log
