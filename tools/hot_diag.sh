#!/bin/bash
# tools/hot_diag.sh -- what-if / diagnostic builds (DIAG="-D...", SHAPES=..., SKIP_DEFAULT=1): the wide kernels with every partner-row gather hitting the same 16 rows (-DWIDE_DIAG_HOT):
# what launch 1 costs without gather latency.  Builds the variant ON THE BOX under invpref_kdd_2022_amd/variants/ (INVPREF_LIB selects it; the in-tree library is never replaced).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
shapes=${SHAPES:-"50000x51283x16x256x262144 400000x100000x16x256x1048576 400000x100000x8x128x1048576 6040x3706x8x128x65536"}
probe() {
  for s in $shapes; do
    echo "== $1 $s"
    PROBE_SHAPE=$s timeout 300 python tools/step_probe.py 2>&1 | grep "us per step"
    PROBE_STAMPS=1 PROBE_STEPS=3 PROBE_SHAPE=$s timeout 300 python tools/step_probe.py 2>&1 | grep "launch 1\|launch 2\|job phases" | cut -c1-220
  done
}
[ -z "$SKIP_DEFAULT" ] && probe default > gpurun_out/hot_diag.log 2>&1
DIAG=${DIAG:--DWIDE_DIAG_HOT}
[ -n "$SKIP_DEFAULT" ] && : > gpurun_out/hot_diag.log
# (a numerically WRONG what-if kernel: built to a path of its own and selected through INVPREF_LIB, never over the in-tree library)
bash tools/build_variant.sh hot_diag $DIAG > /dev/null 2>&1
INVPREF_LIB=$PWD/invpref_kdd_2022_amd/variants/hot_diag.so probe hot >> gpurun_out/hot_diag.log 2>&1
cat gpurun_out/hot_diag.log
