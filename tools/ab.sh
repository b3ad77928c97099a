#!/bin/bash
# tools/ab.sh VARIANT... -- the loop benchmark (tools/kb3.py) once per A/B build of the library (GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  lib=invpref_kdd_2022_amd/variants/$v.so
  [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
  echo "== $v" >> gpurun_out/ab.log
  INVPREF_LIB=$PWD/$lib KB3_SHORT=1 python tools/kb3.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
