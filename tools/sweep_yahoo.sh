#!/bin/bash
# tools/sweep_yahoo.sh -- plan-parameter sweep at the headline shape (loop benchmark, one box)
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout 300 python tools/step_probe.py 2>&1 | grep "us per step" | cut -c1-100; }
run X=0
for ps in 1 3 4; do run INVPREF_PLAN_PER_SLICE=$ps; done
for ips in 2 3 6 8; do run INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
for sr in 16 24 48 64; do run INVPREF_PLAN_STREAM_ROWS=$sr; done
for sp in 0.6 0.8 0.9; do run INVPREF_PLAN_STREAM_SPLIT=$sp; done
run INVPREF_PLAN_FILL=0
run INVPREF_PLAN_PUSH=0
for sr2 in 16 64; do run INVPREF_PLAN_STREAM_ROWS2=$sr2; done
run INVPREF_PLAN_ITEM_ROUNDS=2
run INVPREF_PLAN_SNAKE=32
run X=0
