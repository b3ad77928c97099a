#!/usr/bin/env python3
"""Diagnostic: phase time stamps of mstep_rows_kernel (GPU box).  Not part of the product or tests."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device('cuda:0')
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
os.environ['INVPREF_STAMPS'] = hex(stamps.data_ptr())
from invpref_kdd_2022_amd import ops, plan as planlib, synth

U, I, E, D, B = 15400, 1000, 4, 64, 8192
data = synth.interactions(1, U, I, B, implicit=True)
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [torch.zeros_like(p) for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, B).astype(np.int64)).to(dev)
w = torch.rand(B, device=dev)
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)
for per_slice, rpt, hot in ((1, 1, -1), (1, 1, 16)):
    pl = planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, per_slice=per_slice, rounds_per_task=rpt,
                                hot_threshold=hot)
    dp = planlib.upload(pl, dev)
    for _ in range(5):
        stamps.zero_()
        ops.mstep_rows_adam(P, P2, M, V, dp, e, y, w, B, coefs, flags, losses, 5, 0.005, ws)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8)[:dp.n_tasks].astype(np.int64)
    ni = pl['n_item_rounds'] // rpt
    t0 = st[:, 0].min()
    print(f'== per_slice={per_slice} rpt={rpt} hot>{hot} tasks={dp.n_tasks} (item {ni}); 100 MHz ticks = 10 ns')
    print('kernel span (first start -> last end): %.2f us' % ((st[:, 6:8].max() - t0) / 100))
    for name, sl in (('item', slice(0, ni)), ('user', slice(ni, None))):
        s = st[sl]
        if len(s) == 0:
            continue
        start = (s[:, 0] - t0) / 100
        print(f' {name}: WG start  min {start.min():.2f} med {np.median(start):.2f} max {start.max():.2f} us')
        names = ['prologue(stage+sync)', 'desc load', 'own rows', 'interactions', 'lds combine', 'row finish(adam)', 'dense flush']
        for i, nm in enumerate(names):
            if i == 6 and name == 'item':
                continue
            dlt = (s[:, i + 1] - s[:, i]) / 100
            print(f'   {nm:22s} med {np.median(dlt):6.2f}  p90 {np.quantile(dlt, .9):6.2f}  max {dlt.max():6.2f} us')
        life = (s[:, 7 if name == 'user' else 6] - s[:, 0]) / 100
        print(f'   WG lifetime            med {np.median(life):6.2f}  p90 {np.quantile(life, .9):6.2f}  max {life.max():6.2f} us')
        end = (s[:, 7 if name == 'user' else 6] - t0) / 100
        print(f'   WG end time            med {np.median(end):6.2f}  p90 {np.quantile(end, .9):6.2f}  max {end.max():6.2f} us; starts after 5us: {(start > 5).sum()}')
