#!/usr/bin/env python3
"""Large single-launch roofline points (SURVEY §8(d) "Roofline launch"): one launch over B uniform-random
interactions with MIND-sized tables, per kernel, timed inside a HIP graph.  Prints one JSON line per point.
GPU box only; not part of the product or the tests."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth

dev = torch.device('cuda:0')
U, I = 50000, 51283


def graph_time(fn, inner=5, reps=5):
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
    return a.elapsed_time(b) / (reps * inner) * 1e-3


for E, D, logB in ((4, 64, 20), (4, 64, 22), (8, 128, 20), (16, 256, 20)):
    B = 1 << logB
    data = synth.interactions(1, U, I, B, implicit=True, zipf=False)
    tabs = synth.tables(2, U, I, E, D, std=0.1)
    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
    G = [torch.zeros_like(p) for p in P]
    u, v = (torch.from_numpy(np.ascontiguousarray(data[:, i])).to(dev) for i in (0, 1))
    y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
    e = torch.from_numpy(np.random.RandomState(3).randint(0, E, B).astype(np.int64)).to(dev)
    w = torch.rand(B, device=dev)
    ws = ops.Workspace(dev)
    losses = torch.zeros(6, device=dev)
    coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
    flags = ops.flags_of(True, False, True, True, False)
    npar = sum(p.numel() for p in P)
    table_mb = npar * 4 / 1e6
    t = graph_time(lambda: ops.estep(P, u, v, y, True, e, ws, new_envs=e))
    print(json.dumps(dict(kernel='estep_assign_kernel(+stat_envs)', E=E, D=D, B=B, tables_MB=round(table_mb, 1), ms=t * 1e3,
                          algorithmic_GBs=B * (28 + 16 * D) / t / 1e9, frac_of_8TBs=B * (28 + 16 * D) / t / 8e12)))
    t = graph_time(lambda: ops.mstep_grad(P, G, u, v, e, y, w, B, coefs, flags, losses, ws))
    print(json.dumps(dict(kernel='mstep_atomic_kernel(+finish)', E=E, D=D, B=B, ms=t * 1e3,
                          algorithmic_GBs=B * (32 + 32 * D) / t / 1e9, frac_of_8TBs=B * (32 + 32 * D) / t / 8e12)))
    if logB <= 20 and E * (1 if D <= 64 else 2 if D <= 128 else 4) <= 4:
        dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I), dev)
        P2 = [torch.zeros_like(p) for p in P]; M = [torch.zeros_like(p) for p in P]; V = [torch.zeros_like(p) for p in P]
        t = graph_time(lambda: ops.mstep_rows_adam(P, P2, M, V, dp, e, y, w, B, coefs, flags, losses, 3, 0.005, ws))
        nb = B * (32 + 16 * D) + 24 * npar
        print(json.dumps(dict(kernel='mstep_rows_kernel(+finish), Adam fused', E=E, D=D, B=B, ms=t * 1e3,
                              algorithmic_GBs=nb / t / 1e9, frac_of_8TBs=nb / t / 8e12)))
