#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/yahoo_check.log
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_manager_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -4 >> gpurun_out/yahoo_check.log
timeout 200 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/yahoo_check.log
PROBE_STAMPS=1 timeout 200 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/yahoo_check.log
cat gpurun_out/yahoo_check.log
