cd /root/repo
run() { echo "== $*"; env "$@" timeout 300 python tools/step_probe.py 2>&1 | grep "us per step" | cut -c1-100; }
run X=0
for ips in 2 3 5 6 8; do run INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
run X=0
Y="PROBE_SHAPE=15400x1000x4x64x250154 PROBE_ZIPF=1 PROBE_STEPS=2"
run $Y
for ips in 8 16 24; do run $Y INVPREF_PLAN_ITEM_PER_SLICE=$ips; done
