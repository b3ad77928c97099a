#!/bin/bash
# tools/ab.sh SPEC... -- the loop benchmark (tools/step_probe.py) once per SPEC (GPU box).
# SPEC = VARIANT[,ENV=VALUE...]: an A/B build of the library (tools/build_variant.sh; "default" = the shipped one) and
# environment variables for that run (plan parameters INVPREF_PLAN_*, PROBE_SHAPE, PROBE_STAMPS ...).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for spec in "$@"; do
  IFS=',' read -ra parts <<< "$spec"
  v=${parts[0]}
  lib=invpref_kdd_2022_amd/variants/$v.so
  [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
  echo "== $spec" >> gpurun_out/ab.log
  env INVPREF_LIB=$PWD/$lib "${parts[@]:1}" python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
