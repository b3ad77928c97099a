#!/bin/bash
# tools/ab_tl6b.sh NAME... -- the 2^24-interaction cache-exceeding launches per library variant
cd "$(dirname "$0")/.."
for shape in 400000,100000,4,64,16777216 400000,100000,8,128,16777216; do
  for v in "$@"; do
    lib=invpref_kdd_2022_amd/variants/$v.so
    [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
    r=$(INVPREF_LIB=$PWD/$lib PROBE_SHAPE=$shape PROBE_STEPS=1 timeout 600 python tools/step_probe.py 2>/dev/null | grep "us per step" | sed 's/.*: //')
    echo "$shape $v: $r"
  done
done | tee -a gpurun_out/ab_tl6.txt
