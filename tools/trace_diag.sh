#!/bin/bash
# tools/trace_diag.sh SHAPE... -- -DWIDE_DIAG_TRACE build ON THE BOX + tools/wide_trace.py per shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/build_variant.sh trace_diag -DWIDE_DIAG_TRACE > /dev/null 2>&1   # (a path of its own, selected through INVPREF_LIB)
for s in "$@"; do
  echo "== $s"
  INVPREF_LIB=$PWD/invpref_kdd_2022_amd/variants/trace_diag.so PROBE_SHAPE=$s timeout 300 python tools/wide_trace.py 2>&1 | grep -v amdgpu.ids | head -${TRACE_LINES:-70}
done > gpurun_out/trace_diag.log 2>&1
cat gpurun_out/trace_diag.log
