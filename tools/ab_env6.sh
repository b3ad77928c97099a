#!/bin/bash
# tools/ab_env6.sh "ENV=VAL ENV2=VAL" ... -- the bench's timed loop under environment settings (plan parameters etc.)
cd "$(dirname "$0")/.."
for setting in "$@"; do
  for rep in 1 2; do
    env $setting timeout 300 python bench.py --no-extras --no-cpu-baseline --no-parity-gate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$setting] rep $rep: ms_per_step %.5f  launch %.5f  estep %.4f  value %.1f M/s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['detail']['estep_ms'], d['value']/1e6))
"
  done
done | tee -a gpurun_out/ab_env6.txt
