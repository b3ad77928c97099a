#!/bin/bash
# tools/wide_sweep.sh -- plan-parameter sweep of the wide kernels (loop benchmark), one box, one call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/wide_sweep.log
run() { echo "== $*" >> gpurun_out/wide_sweep.log; env "$@" timeout 300 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/wide_sweep.log; }
M=PROBE_SHAPE=6040x3706x8x128x65536
run $M
for ps in 8 10 16; do run $M INVPREF_PLAN_PER_SLICE=$ps; done
for ips in 2 8 16; do run $M INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
run $M INVPREF_PLAN_PUSH=0
run $M INVPREF_PLAN_PER_SLICE=8 INVPREF_PLAN_ITEM_PER_SLICE=8
run $M INVPREF_PLAN_ROUNDS=2 INVPREF_PLAN_PER_SLICE=6
N=PROBE_SHAPE=50000x51283x16x256x262144
run $N
run $N INVPREF_PLAN_ITEM_ROUNDS=8
run $N INVPREF_PLAN_PER_SLICE=16
run $N INVPREF_PLAN_PER_SLICE=8
run $N INVPREF_PLAN_ROUNDS=2
run $N INVPREF_PLAN_ROUNDS=8
run $N INVPREF_PLAN_ITEM_PER_SLICE=6
run $N INVPREF_PLAN_ITEM_PER_SLICE=16
L=PROBE_SHAPE=400000x100000x16x256x1048576
run $L
run $L INVPREF_PLAN_ITEM_ROUNDS=2
run $L INVPREF_PLAN_ROUNDS=4
run $L INVPREF_PLAN_ROUNDS=8 INVPREF_PLAN_ITEM_ROUNDS=2
run $L INVPREF_PLAN_PER_SLICE=3 INVPREF_PLAN_ITEM_ROUNDS=2
cat gpurun_out/wide_sweep.log
