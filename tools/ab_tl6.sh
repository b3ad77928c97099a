#!/bin/bash
# tools/ab_tl6.sh NAME... -- round 6 A/B of the two-launch forms (tools/step_probe.py: steps in one HIP graph, HIP events) per library
# variant at the MovieLens-, MIND- and Yahoo-B=N-shaped steps and the 2^22-interaction cache-exceeding launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for shape in 6040,3706,8,128,65536 50000,51283,16,256,262144 15400,1000,4,64,250154 400000,100000,4,64,4194304; do
  for v in "$@"; do
    lib=invpref_kdd_2022_amd/variants/$v.so
    [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
    r=$(INVPREF_LIB=$PWD/$lib PROBE_SHAPE=$shape PROBE_STEPS=3 timeout 300 python tools/step_probe.py 2>/dev/null | grep "us per step" | sed 's/.*: //')
    echo "$shape $v: $r"
  done
done | tee -a gpurun_out/ab_tl6.txt
