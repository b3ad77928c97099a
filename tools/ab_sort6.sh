#!/bin/bash
# tools/ab_sort6.sh -- what the position-indexed accesses are worth at the cache-exceeding launch (2^24 interactions): the
# minibatch as generated / in user order (launch 1's environment read and record store sequential) / in item order (launch 2's
# record reads sequential)
cd "$(dirname "$0")/.."
for shape in 400000,100000,4,64,16777216 400000,100000,8,128,16777216; do
  for v in "X=1" "PROBE_USORT=1" "PROBE_ISORT=1"; do
    r=$(env $v PROBE_SHAPE=$shape PROBE_STEPS=1 timeout 600 python tools/step_probe.py 2>/dev/null | grep "us per step" | sed 's/.*: //')
    echo "$shape [$v]: $r"
  done
done | tee -a gpurun_out/ab_sort6.txt
