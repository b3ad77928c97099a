#!/bin/bash
# tools/defer_ab.sh -- deferred dense Adam on/off, same box, same call: parity tests then the bench line twice each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_deferred_gpu.py -x -q > gpurun_out/defer_tests.log 2>&1
tail -15 gpurun_out/defer_tests.log
for i in 1 2; do
  for d in 1 0; do
    echo "== INVPREF_DEFER=$d" >> gpurun_out/defer_ab.log
    INVPREF_DEFER=$d timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print('value %.1f M/s  ms_per_step %.5f  dev step %.5f  estep_ms %.4f' % (j['value']/1e6, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['detail']['estep_ms']), {k: round(v, 5) for k, v in j['detail'].items() if 'eager' in k})
" >> gpurun_out/defer_ab.log
  done
done
cat gpurun_out/defer_ab.log
