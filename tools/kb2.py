#!/usr/bin/env python3
"""Fused rows-kernel timing at Yahoo scale and at a large batch (GPU box; experiments only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from invpref_kdd_2022_amd import ops, plan as planlib, synth
dev = torch.device('cuda:0')
def graph_time(fn, inner=10, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner): fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): g.replay()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * inner) * 1e3
for (U, I, E, D, B, zipf) in ((15400, 1000, 4, 64, 8192, True), (50000, 51283, 4, 64, 1 << 19, False)):
    data = synth.interactions(1, U, I, B, implicit=True, zipf=zipf)
    tabs = synth.tables(2, U, I, E, D)
    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
    P2 = [torch.zeros_like(p) for p in P]; M = [torch.zeros_like(p) for p in P]; V = [torch.zeros_like(p) for p in P]
    y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
    e = torch.from_numpy(np.random.RandomState(3).randint(0, E, B).astype(np.int64)).to(dev)
    w = torch.rand(B, device=dev); ws = ops.Workspace(dev); losses = torch.zeros(6, device=dev)
    coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9); flags = ops.flags_of(True, False, True, True, False)
    for hot in (16, 10 ** 9, -1):
        dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, hot_threshold=hot), dev)
        t = graph_time(lambda: ops.mstep_rows_adam(P, P2, M, V, dp, e, y, w, B, coefs, flags, losses, 5, 0.005, ws))
        print(f'B={B:8d} hot>{hot:<10d} tasks {dp.n_tasks:6d}  fused step {t:8.1f} us   {B / t:8.1f} M inter/s')
