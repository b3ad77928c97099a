#!/bin/bash
# tools/ab_pushpull2.sh -- push vs pull at the Yahoo tables on SKEWED ids (Zipf, and the Yahoo-like generator's own popularity):
# ab_pushpull.sh ran on uniform ids
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() {
  r=$(env $3 $2 PROBE_STEPS=2 timeout 300 python tools/step_probe.py 2>/dev/null | grep -E "us per step|^shape" | sed 's/.*: //' | sed 's/lanes [0-9]*, //; s/stream rows.*workgroups/wgs/' | tr '\n' ' ')
  echo "$1 [$3]: $r"
}
for b in 32768 65536 131072 250154; do
  for p in 1 0; do run "zipf B=$b" "PROBE_SHAPE=15400,1000,4,64,$b PROBE_ZIPF=1" "INVPREF_PLAN_PUSH=$p"; done
  for p in 1 0; do run "yahoo_like B=$b" "PROBE_B=$b" "INVPREF_PLAN_PUSH=$p"; done
done | tee gpurun_out/ab_pushpull2.txt
