#!/bin/bash
# tools/shape_sweep.sh U I E D B -- plan-parameter sweep of the planned step at one shape (tools/kbench.py), GPU box
cd "$(dirname "$0")/.."
for dense in ${DENSES:-64 128 256}; do
  echo "== dense_per_task=$dense"
  INVPREF_PLAN_DENSE=$dense KB_COMBOS=${COMBOS:-4:1:64,8:1:128,16:1:256,32:1:512,8:1:1000000000,16:2:256,4:2:64} python tools/kbench.py "$@" 2>&1 | grep "^rows"
done
