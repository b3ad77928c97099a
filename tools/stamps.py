#!/usr/bin/env python3
"""Diagnostic: phase time stamps of mstep_rows_kernel (GPU box).  Not part of the product or tests.
The stamped launch is the LAST of a run of steps over different minibatch plans with ping-pong parameter buffers
(the cache state of the training loop), STAMPS_STEPS of them (default 12)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device('cuda:0')
stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
os.environ['INVPREF_STAMPS'] = hex(stamps.data_ptr())
from invpref_kdd_2022_amd import ops, plan as planlib, synth

U, I, E, D, B = 15400, 1000, 4, 64, 8192
nsteps = int(os.environ.get('STAMPS_STEPS', '12'))
if os.environ.get('STAMPS_SHAPE'):   # "U,I,E,D,B": another shape (uniform synthetic interactions)
    U, I, E, D, B = [int(x) for x in os.environ['STAMPS_SHAPE'].split(',')]
    nsteps = min(nsteps, 3)
    data = synth.interactions(1, U, I, nsteps * B, implicit=True)
else:
    data = synth.yahoo_like()[:nsteps * B]
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [torch.zeros_like(p) for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, len(data)).astype(np.int64)).to(dev)
w = torch.rand(len(data), device=dev)
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)
if os.environ.get('STAMPS_DRAIN') != '1':
    os.environ['INVPREF_STAMPS_NODRAIN'] = '1'
combos = [tuple(int(x) for x in c.split(',')) for c in os.environ.get('STAMPS_COMBOS', '2,1,16,32').split(';')]
for per_slice, rpt, hot, dense in combos:
    os.environ['INVPREF_PLAN_DENSE'] = str(dense)
    pls = [planlib.build_row_plan(data[k * B:(k + 1) * B, 0], data[k * B:(k + 1) * B, 1], data[k * B:(k + 1) * B, 2], U, I,
                                  per_slice=per_slice, rounds_per_task=rpt, hot_threshold=hot) for k in range(nsteps)]
    dps = [planlib.upload(pl, dev) for pl in pls]
    for rep in range(3):
        a, b = P, P2
        for k in range(nsteps):
            if k == nsteps - 1:
                stamps.zero_()
            sl = slice(k * B, (k + 1) * B)
            ops.mstep_rows_adam(a, b, M, V, dps[k], e[sl], y[sl], w[sl], B, coefs, flags, losses, 5 + k, 0.005, ws)
            a, b = b, a
    torch.cuda.synchronize()
    pl, dp = pls[-1], dps[-1]
    ncls, cls, spt = pl['n_classes'], pl['cls'], pl['rows_per_stream_task']
    nd_all = -(-B // dense)
    per_class = dp.n_tasks // ncls
    nfin = 0   # (blocks beyond the task grid: an experiment carried the finish in the last blocks of the launch)
    st = stamps.cpu().numpy().reshape(-1, 8)[:dp.n_tasks + nfin].astype(np.int64)
    kind = np.full(dp.n_tasks + nfin, 'pad   ', dtype=object)
    for b in range(dp.n_tasks):          # the kernel's own dispatch arithmetic
        c, j = b % ncls, b // ncls
        ndc = max(0, -(-(nd_all - c) // ncls))
        ti, tu = -(-int(cls[c, 1]) // rpt), -(-int(cls[c, 3]) // rpt)
        ts = -(-int(cls[c, 5]) // spt) + -(-int(cls[c, 7]) // spt)
        kind[b] = 'dense' if j < ndc else 'item' if j < ndc + ti else 'user' if j < ndc + ti + tu else 'stream' if j < ndc + ti + tu + ts else 'pad'
    kind[dp.n_tasks:] = 'finish'
    live = st[:, 0] > 0
    t0 = st[live, 0].min()
    end = np.where(st[:, 7] > 0, st[:, 7], st[:, 6])
    print(f'== per_slice={per_slice} hot>{hot} dense/task={dense}: grid {dp.n_tasks} + {nfin} finish blocks; ' + ', '.join(f'{k} {int((kind == k).sum())}' for k in ('dense', 'item', 'user', 'stream', 'pad', 'finish')) + f'; n_hot {len(pl["hot_rows"])}; span {(end[live].max() - t0) / 100:.2f} us')
    for name in ('dense', 'item', 'user', 'stream', 'finish'):
        sel = (kind == name) & live
        if not sel.any():
            continue
        s0, e0 = (st[sel, 0] - t0) / 100, (end[sel] - t0) / 100
        life = e0 - s0
        print(f'  {name:6s} n={sel.sum():4d} start med {np.median(s0):5.2f} p90 {np.quantile(s0, .9):5.2f} max {s0.max():5.2f} | life med {np.median(life):5.2f} p90 {np.quantile(life, .9):5.2f} max {life.max():5.2f} | end med {np.median(e0):5.2f} p90 {np.quantile(e0, .9):5.2f} max {e0.max():5.2f}')
    for name in ('item', 'user'):
        j = st[(kind == name) & live].astype(np.float64)
        ph = np.diff(j[:, :7], axis=1) / 100
        print('  %s job phases (us, median): stage issue %.2f | descriptor %.2f | gathers+sync %.2f | interactions %.2f | slice meet %.2f | adam+store %.2f' % ((name,) + tuple(np.median(ph, axis=0))))
    d = st[(kind == 'dense') & live].astype(np.float64)
    print('  dense (us, median / max): start -> staged %.2f/%.2f | staged -> end of loop %.2f/%.2f | end of loop -> arrived %.2f/%.2f'
          % tuple(x for a_, b_ in ((0, 1), (1, 6), (6, 7)) for x in (np.median(d[:, b_] - d[:, a_]) / 100, (d[:, b_] - d[:, a_]).max() / 100)))
    fs = stamps.cpu().numpy().reshape(-1, 8)[12288:12288 + 256].astype(np.int64)   # rows_finish_kernel's blocks
    fl = fs[:, 0] > 0
    if fl.any():
        main_end = end[live].max()
        slab = fl & (fs[:, 2] > 0)
        hotb = fl & (fs[:, 2] == 0) & (fs[:, 3] > 0)
        print(f'  finish kernel: {int(fl.sum())} stamped blocks; first block starts {(fs[fl, 0].min() - main_end) / 100:.2f} us after the last '
              f'workgroup of the main kernel ended; last block ends {(fs[fl, 3].max() - main_end) / 100:.2f} us after it')
        for nm, sel in (('slab', slab), ('hot ', hotb)):
            if sel.any():
                x = fs[sel].astype(np.float64)
                print(f'    {nm} blocks n={int(sel.sum())}: start {np.median(x[:, 0] - main_end) / 100:.2f} | loads landed +{np.median(x[:, 1] - x[:, 0]) / 100:.2f}'
                      + (f' | adam + stores issued and drained +{np.median(x[:, 2] - x[:, 1]) / 100:.2f} | end +{np.median(x[:, 3] - x[:, 2]) / 100:.2f} (max end {(x[:, 3].max() - main_end) / 100:.2f})'
                         if nm == 'slab' else f' | end +{np.median(x[:, 3] - x[:, 1]) / 100:.2f} (max end {(x[:, 3].max() - main_end) / 100:.2f})'))
    f = st[(kind == 'finish') & live].astype(np.float64)
    if len(f):
        print('  finish blocks (us, median / max): waiting %.2f/%.2f | finish work %.2f/%.2f | counters %.2f/%.2f'
              % tuple(x for a_, b_ in ((0, 1), (1, 6), (6, 7)) for x in (np.median(f[:, b_] - f[:, a_]) / 100, (f[:, b_] - f[:, a_]).max() / 100)))
