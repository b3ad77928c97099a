#!/bin/bash
# tools/sweep_rounds32.sh -- the 32-lane instance at MIND's tables, several minibatch sizes: default plan vs fixed rounds per task
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout 600 python tools/step_probe.py 2>&1 | grep "us per step" | cut -c1-100; }
for B in 32768 65536 131072 262144; do
  S="PROBE_SHAPE=50000x51283x16x256x$B"
  run $S
  for r in ${ROUNDS:-2 4 6 8}; do run $S INVPREF_PLAN_ROUNDS=$r; done
done
