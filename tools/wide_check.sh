#!/bin/bash
# tools/wide_check.sh -- parity of the wide-row kernels (csrc/step_wide.hpp) and their step times at the bench shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/wide_*.log
INVPREF_FUZZ=${FUZZ:-60} timeout 1500 python -m pytest tests/test_edge_cases_gpu.py tests/test_hip_parity.py tests/test_large_traj_gpu.py -x -q > gpurun_out/wide_tests.log 2>&1
tail -25 gpurun_out/wide_tests.log
for shape in 6040x3706x8x128x65536 50000x51283x16x256x262144 50000x51283x16x256x32768 400000x100000x8x128x1048576 400000x100000x16x256x1048576 6040x3706x8x64x65536; do
  echo "== $shape" >> gpurun_out/wide_probe.log
  PROBE_SHAPE=$shape timeout 300 python tools/step_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/wide_probe.log
done
cat gpurun_out/wide_probe.log
