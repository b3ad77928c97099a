#!/bin/bash
# tools/ab2.sh "ENV=..;ENV=.." VARIANT... -- kb3 per library variant and per plan-parameter environment
cd "$(dirname "$0")/.."
IFS=';' read -ra envs <<< "$1"; shift
for v in "$@"; do
  lib=invpref_kdd_2022_amd/variants/$v.so
  [ "$v" = default ] && lib=invpref_kdd_2022_amd/libinvpref_hip.so
  for e in "${envs[@]}"; do
    echo "== $v [$e]"
    env $e INVPREF_LIB=$PWD/$lib KB3_SHORT=1 python tools/kb3.py 2>&1 | grep -v amdgpu.ids | tail -2
  done
done
