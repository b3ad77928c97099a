#!/bin/bash
# tools/trace_diag.sh SHAPE... -- -DWIDE_DIAG_TRACE build ON THE BOX + tools/wide_trace.py per shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
INVPREF_HIPCC_EXTRA="-DWIDE_DIAG_TRACE" python -c "from invpref_kdd_2022_amd import build; build.build(force=True)" > /dev/null 2>&1
for s in "$@"; do
  echo "== $s"
  PROBE_SHAPE=$s timeout 300 python tools/wide_trace.py 2>&1 | grep -v amdgpu.ids | head -${TRACE_LINES:-70}
done > gpurun_out/trace_diag.log 2>&1
cat gpurun_out/trace_diag.log
