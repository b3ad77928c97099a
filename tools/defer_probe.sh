#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/defer_probe.log
echo "== PROBE_DEFER=0 loop" >> gpurun_out/defer_probe.log
timeout 200 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/defer_probe.log
echo "== PROBE_DEFER=1 loop" >> gpurun_out/defer_probe.log
PROBE_DEFER=1 timeout 200 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/defer_probe.log
for pend in 0 1 3; do
  echo "== PROBE_DEFER=1 stamps, every row $pend steps behind" >> gpurun_out/defer_probe.log
  PROBE_DEFER=1 PROBE_DEFER_PEND=$pend PROBE_STAMPS=1 timeout 200 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/defer_probe.log
done
cat gpurun_out/defer_probe.log
