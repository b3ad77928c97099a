cd /root/repo
run() { echo "== $*"; env "$@" timeout 600 python tools/step_probe.py 2>&1 | grep "us per step" | cut -c1-100; }
K="PROBE_SHAPE=400000x100000x16x256x1048576 PROBE_STEPS=2"
run $K
for ps in 3 4 8 12; do run $K INVPREF_PLAN_PER_SLICE=$ps; done
for r in 12 24 32; do run $K INVPREF_PLAN_ROUNDS=$r; done
L="PROBE_SHAPE=400000x100000x8x128x2097152 PROBE_STEPS=2"
run $L
for ps in 3 4 8 12; do run $L INVPREF_PLAN_PER_SLICE=$ps; done
J="PROBE_SHAPE=400000x100000x4x64x4194304 PROBE_STEPS=2"
run $J
for ps in 4 6 8 12 16; do run $J INVPREF_PLAN_PER_SLICE=$ps; done
