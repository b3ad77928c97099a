#!/usr/bin/env python3
"""Kernel micro-bench (GPU box): times the M-step variants back to back with HIP events around a
long loop (GPU-bound), for plan-parameter sweeps.  Not part of the product or the tests."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth

dev = torch.device('cuda:0')
U, I, E, D, B = 15400, 1000, 4, 64, 8192
if len(sys.argv) > 1:
    U, I, E, D, B = [int(x) for x in sys.argv[1:6]]
data = synth.interactions(1, U, I, B, implicit=True)
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [torch.zeros_like(p) for p in P]
G = [torch.zeros_like(p) for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
u, v = (torch.from_numpy(np.ascontiguousarray(data[:, i])).to(dev) for i in (0, 1))
y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, B).astype(np.int64)).to(dev)
w = torch.rand(B, device=dev)
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3, (time.perf_counter() - t0) / iters * 1e6


print('atomic grad: %.1f us gpu  (%.1f us wall)' % timeit(lambda: ops.mstep_grad(P, G, u, v, e, y, w, B, coefs, flags, losses, ws)))
flat = torch.zeros(sum(p.numel() for p in P) // 4 * 4 + 64, device=dev)
print('adam flat:   %.1f us gpu  (%.1f us wall)' % timeit(lambda: ops.adam_(flat, flat.clone() if False else flat, flat, flat, 1, 0.005)))
def graph_time(fn, inner=20, reps=20):
    """GPU-only time per call: capture `inner` calls in a HIP graph and replay it."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner):
                fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * inner) * 1e3


print('graph: atomic grad %.1f us, adam %.1f us' % (
    graph_time(lambda: ops.mstep_grad(P, G, u, v, e, y, w, B, coefs, flags, losses, ws)),
    graph_time(lambda: ops.adam_(flat, flat, flat, flat, 1, 0.005))))
combos = [tuple(int(x) for x in c.split(':')) for c in os.environ['KB_COMBOS'].split(',')] if os.environ.get('KB_COMBOS') else ((1, 1, 16), (2, 1, 16), (2, 1, 10 ** 9))
for per_slice, rpt, hot in combos:
    pl = planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, per_slice=per_slice, rounds_per_task=rpt,
                                hot_threshold=hot)
    dp = planlib.upload(pl, dev)
    tg = graph_time(lambda: ops.mstep_rows_grad(P, G, dp, e, y, w, B, coefs, flags, losses, ws))
    tf = graph_time(lambda: ops.mstep_rows_adam(P, P2, M, V, dp, e, y, w, B, coefs, flags, losses, 5, 0.005, ws))
    print(f'rows per_slice={per_slice} rounds/task={rpt} hot>{hot}: tasks {dp.n_tasks:5d} rounds {dp.n_rounds:5d}  grad {tg:.1f} us  fused-adam {tf:.1f} us')
