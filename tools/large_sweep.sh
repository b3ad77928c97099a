#!/bin/bash
# tools/large_sweep.sh -- plan-parameter sweep on the cache-exceeding points (tools/roofline_large.py), GPU box
cd "$(dirname "$0")/.."
for e in "X=1" "INVPREF_PLAN_ROUNDS=2" "INVPREF_PLAN_ROUNDS=4" "INVPREF_PLAN_ROUNDS=8" "INVPREF_PLAN_DENSE=512" "INVPREF_PLAN_DENSE=128" "INVPREF_PLAN_STREAM_ROWS=256" "INVPREF_PLAN_ROUNDS=4 INVPREF_PLAN_STREAM_ROWS=256 INVPREF_PLAN_DENSE=512"; do
  echo "== $e"
  env $e ROOFLINE_REPS=2 ROOFLINE_POINTS=${POINTS:-0,2} python tools/roofline_large.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('  D=%d %-14s %.3f ms  %.3f' % (d['D'], d['form'][:14], d['ms'], d['frac_of_8TBs']))
"
done
