#!/bin/bash
# tools/ab_r06.sh NAME... -- A/B of library variants (invpref_kdd_2022_amd/variants/NAME.so, made by build.build_variant) on the
# (AB_BENCH_FLAGS=--no-parity-gate for what-if builds whose results are wrong on purpose) bench's own timed loop: per-step time of the timed region and HIP-event time per launch.  "default" = the regular library.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  lib=invpref_kdd_2022_amd/variants/$v.so
  [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
  [ "$v" = foldkernel ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so && export INVPREF_ESTEP_FOLD=kernel
  for rep in 1 2; do
    INVPREF_LIB=$PWD/$lib timeout 300 python bench.py --no-extras --no-cpu-baseline $AB_BENCH_FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v rep $rep: ms_per_step %.5f  launch %.5f  estep %.4f  value %.1f M/s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['detail']['estep_ms'], d['value']/1e6))
"
  done
  unset INVPREF_ESTEP_FOLD
done | tee -a gpurun_out/ab_r06.txt
