#!/bin/bash
# tools/packed_check.sh -- the packed exchange on the GPU box: kernel + manager tests, then bench.py's N-rank flow rehearsed
# with two ranks on the one GPU (gloo; never a measurement), then the default bench line (plan build times at 2^24)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_manager_gpu.py -x -q -k "packed or pack_rows or sharded_epochs" > gpurun_out/packed_tests.log 2>&1
tail -5 gpurun_out/packed_tests.log
INVPREF_BENCH_BACKEND=gloo INVPREF_BENCH_SAME_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 310 --warmup 31 > gpurun_out/rehearsal2.json 2> gpurun_out/rehearsal2.err
tail -c 2500 gpurun_out/rehearsal2.json
tail -3 gpurun_out/rehearsal2.err
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'])
rl = d['roofline_large']
for k, v in rl['sweep'].items():
    print(k, {kk: (round(x, 4) if isinstance(x, float) else x) for kk, x in v.items() if kk != 'shape'})
for k, v in rl['sweep_r03_sizes'].items():
    print('r03', k, {kk: (round(x, 4) if isinstance(x, float) else x) for kk, x in v.items() if kk != 'shape'})
print({k: (v.get('frac'), v.get('ms_per_step')) for k, v in d['detail']['configs'].items() if isinstance(v, dict)})
print(d['detail']['estep_random_sort'])
PY
