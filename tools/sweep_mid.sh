#!/bin/bash
# tools/sweep_mid.sh -- plan-parameter sweeps at the MovieLens / Yahoo-large-batch / MIND-rank-share shapes (one box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/sweep_mid.log
run() { echo "== $*" >> gpurun_out/sweep_mid.log; env "$@" timeout 600 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-230 >> gpurun_out/sweep_mid.log; }
M="PROBE_SHAPE=6040x3706x8x128x65536"
run $M
for ps in 8 10 11 13 14 16 20; do run $M INVPREF_PLAN_PER_SLICE=$ps; done
for ips in 4 8 12 24 32; do run $M INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
run $M INVPREF_PLAN_SNAKE=0
run $M INVPREF_PLAN_SNAKE=16
run $M INVPREF_PLAN_SNAKE=64
Y="PROBE_SHAPE=15400x1000x4x64x250154 PROBE_ZIPF=1 PROBE_STEPS=2"
run $Y
for ps in 8 12 24 32; do run $Y INVPREF_PLAN_PER_SLICE=$ps; done
for r in 1 2 4; do run $Y INVPREF_PLAN_ROUNDS=$r; done
for ips in 8 16 32; do run $Y INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
R="PROBE_SHAPE=50000x51283x16x256x32768"
run $R
for ps in 2 3 4 6 8; do run $R INVPREF_PLAN_PER_SLICE=$ps; done
for r in 1 2 4 8; do run $R INVPREF_PLAN_ROUNDS=$r; done
cat gpurun_out/sweep_mid.log
