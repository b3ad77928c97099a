#!/usr/bin/env python3
"""Cache-exceeding single-launch roofline points of the planned M-step (SURVEY section 8(d) "Roofline launch"; BASELINE
configs 2, 3 and 5): one launch over B uniform-random interactions, tables sized so that the five flat buffers are far
beyond the 256 MiB Infinity Cache, both forms of the step -- the fused pass (mstep_rows_kernel + finish, Adam inside)
and the gradient pass + flat Adam (what multi-GPU runs and rows of more than 128 floats use).  One JSON line per point;
`tools/profile_r02.sh` runs the same program under rocprofv3 for the FETCH_SIZE / WRITE_SIZE counters.
GPU box only; not part of the product or the tests."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth

dev = torch.device('cuda:0')
POINTS = [  # E, D, users, items, log2(B), note
    (4, 64, 400000, 100000, 20, 'Yahoo-class kernel instance (one row chunk, E <= 4)'),
    (8, 128, 200000, 100000, 20, 'MovieLens-class instance (two row chunks, E = 8)'),
    (16, 256, 50000, 51283, 20, 'MIND-class instance and MIND-sized tables (four row chunks, E = 16)'),
]
only = os.environ.get('ROOFLINE_POINTS')
reps = int(os.environ.get('ROOFLINE_REPS', '3'))


def timed(fn):
    ts = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts[1:]) * 1e-3


for idx, (E, D, U, I, logB, note) in enumerate(POINTS):
    if only and str(idx) not in only.split(','):
        continue
    B = 1 << logB
    rs = np.random.RandomState(1)
    u, v = rs.randint(0, U, B), rs.randint(0, I, B)
    yv = (rs.random_sample(B) < 0.5).astype(np.float32)
    tabs = synth.tables(2, U, I, E, D, std=0.1)
    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
    del tabs
    P2, M, V, G = ([torch.zeros_like(p) for p in P] for _ in range(4))
    t0 = time.perf_counter()
    dp = planlib.upload(planlib.build_row_plan(u, v, yv, U, I), dev)
    plan_s = time.perf_counter() - t0
    y = torch.from_numpy(yv).to(dev)
    e = torch.from_numpy(rs.randint(0, E, B).astype(np.int64)).to(dev)
    w = torch.rand(B, device=dev)
    ws = ops.Workspace(dev)
    losses = torch.zeros(6, device=dev)
    coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
    flags = ops.flags_of(True, False, True, True, False)
    npar = sum(p.numel() for p in P)
    flat = torch.cat([p.reshape(-1) for p in P])
    fg, fm, fv = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    base = dict(E=E, D=D, users=U, items=I, B=B, flat_buffer_MB=round(npar * 4 / 1e6, 1), plan_build_s=round(plan_s, 2), note=note)
    st = [0]

    def fused():
        st[0] += 1
        a, b = (P, P2) if st[0] & 1 else (P2, P)
        ops.mstep_rows_adam(a, b, M, V, dp, e, y, w, B, coefs, flags, losses, st[0], 0.005, ws)
    t = timed(fused)
    nb = B * (32 + 16 * D) + 24 * npar
    print(json.dumps(dict(base, form='fused: mstep_rows_kernel + rows_finish_kernel (Adam inside)', ms=t * 1e3,
                          algorithmic_bytes=nb, algorithmic_GBs=nb / t / 1e9, frac_of_8TBs=nb / t / 8e12)), flush=True)
    t = timed(lambda: ops.mstep_rows_grad(P, G, dp, e, y, w, B, coefs, flags, losses, ws))
    nbg = B * (32 + 16 * D) + 4 * npar        # rows read once per interaction side, every gradient row written once
    print(json.dumps(dict(base, form='gradient pass: mstep_rows_kernel + rows_finish_kernel (gradient rows stored)', ms=t * 1e3,
                          algorithmic_bytes=nbg, algorithmic_GBs=nbg / t / 1e9, frac_of_8TBs=nbg / t / 8e12)), flush=True)
    t = timed(lambda: ops.adam_(flat, fg, fm, fv, 3, 0.005, zero_grad=False))
    nba = 28 * npar
    print(json.dumps(dict(base, form='flat Adam: adam_kernel (p, g, m, v read; p, m, v written)', ms=t * 1e3,
                          algorithmic_bytes=nba, algorithmic_GBs=nba / t / 1e9, frac_of_8TBs=nba / t / 8e12)), flush=True)
    del P, P2, M, V, G, flat, fg, fm, fv, dp
    torch.cuda.empty_cache()
