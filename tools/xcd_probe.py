#!/usr/bin/env python3
"""Diagnostic (GPU box): does a row's Adam state stay in the L2 of the XCD that wrote it, from one step's launch to the
next?  Pure fused-Adam streaming (every row untouched, 50 MB per step, ping-pong parameter buffers) with (a) the same
row -> workgroup assignment every step and (b) the assignment rotated by one 64-row task per step (block b and b+1 sit
on different XCDs under round-robin placement)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth

dev = torch.device('cuda:0')
U, I, E, D = 15400, 1000, 4, 64
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)
e = torch.zeros(1, dtype=torch.int64, device=dev)
y = torch.zeros(1, device=dev)
w = torch.ones(1, device=dev)


def plan_rot(rot):
    pl = planlib.build_row_plan(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32), U, I, n_classes=1)
    nu = pl['n_stream_user']
    su, si = pl['stream_rows'][:nu], pl['stream_rows'][nu:]
    pl['stream_rows'] = np.concatenate([np.roll(su, 64 * rot), np.roll(si, 64 * rot)])
    return planlib.upload(pl, dev)


def graph_time(plans, steps=32, reps=20):
    def seq():
        a, b = P, P2
        for k in range(steps):
            ops.mstep_rows_adam(a, b, M, V, plans[k % len(plans)], e, y, w, 1, coefs, flags, losses, 5, 0.005, ws)
            a, b = b, a
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        seq(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            seq()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * steps) * 1e3


print('same assignment every step      : %.2f us/step' % graph_time([plan_rot(0)]))
print('rotated by 1 task per step      : %.2f us/step' % graph_time([plan_rot(r) for r in range(8)]))
print('rotated by 8 tasks per step     : %.2f us/step' % graph_time([plan_rot(8 * r) for r in range(8)]))
print('rotated by 4 tasks per step     : %.2f us/step' % graph_time([plan_rot(4 * r) for r in range(8)]))
