#!/bin/bash
# tools/profile_r05b.sh -- the second half of profiles/r05 (GPU box): launch 1 of the D = 256 instance, the MFMA-classifier form
# against the per-interaction form (rocprofv3 kernel durations per shape), and the E-step kernel to kernel (plain vs tie-break,
# per dispatch, pinned indices vs copied first)
cd "$(dirname "$0")/.."
PROBE_STEPS=3 tools/wide_whatif.sh "default default,INVPREF_WIDE_MM=0" 50000x51283x16x256x262144 400000x100000x16x256x16777216 400000x100000x16x256x1048576 50000x51283x16x256x32768 > /dev/null 2>&1
cp gpurun_out/wide_whatif/summary.txt gpurun_out/mm_launch1_ab.txt
tools/estep_ab.sh > /dev/null 2>&1
cp gpurun_out/estep_ab/summary.txt gpurun_out/estep_kernel_ab.txt
cat gpurun_out/mm_launch1_ab.txt gpurun_out/estep_kernel_ab.txt
