#!/usr/bin/env python3
"""tools/wide_trace.py -- with a -DWIDE_DIAG_TRACE build (INVPREF_LIB): shader-clock stamps of the first steps of one wave
of a wide launch-1 task (workgroup 40): where a lock-step iteration spends its time.
tags: 1 step entry | 2 slot rows + env row arrived | (inside the evaluation: 10 row sums | 11 loss chains | 12 class dot
products | 13 butterfly + softmax, gz written | 14 gz read back) | 3 evaluation done | 4 stores / updates issued | 5 MFMAs issued |
6 refill issued"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device('cuda:0')
stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
os.environ['INVPREF_STAMPS'] = hex(stamps.data_ptr())
os.environ['INVPREF_STAMPS_NODRAIN'] = '1'
from invpref_kdd_2022_amd import ops, plan as planlib, synth  # noqa: E402

U, I, E, D, B = [int(x) for x in os.environ.get('PROBE_SHAPE', '6040x3706x8x128x65536').split('x')]
data = synth.interactions(1, U, I, B, implicit=True)
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
y = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, B).astype(np.int64)).to(dev)
w = torch.rand(B, device=dev)
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
plan = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D, env_num=E), dev)
for _ in range(3):
    stamps.zero_()
    ops.mstep_rows_adam(P, P2, M, V, plan, e, y, w, B, (3.35, 9.99, 9.06, 3.13, 0.49, 1.9), ops.flags_of(True, False, True, True, False),
                        losses, 5, 0.005, ws)
torch.cuda.synchronize()
raw = stamps.cpu().numpy()[100000:100120]
skip = int(os.environ.get('TRACE_SKIP', '0'))
raw = raw[raw != 0]
tag, t = (raw >> 56) & 0xff, raw & ((1 << 56) - 1)
t = t - t[0]
names = {1: 'entry', 2: 'rows arrived', 3: 'evaluated (gx done)', 4: 'updates/stores', 5: 'mfma issued', 6: 'refill issued',
         30: '  S1 rows in, x/qi out', 31: '  S1 row sums', 32: '  S1 loss chains', 33: '  S1 meta', 34: '  S1 ge/o/push/reports',
         20: 'S1 entry', 21: ' S3 softmax done', 22: ' S3 products + gx out', 23: ' S1 done', 24: 'barrier', 25: ' S2 products + P out', 26: ' S5 done', 27: 'barrier',
         10: '  row sums', 11: '  loss chains', 12: '  class dots', 13: '  softmax, gz out', 14: '  gz back'}
prev = 0
for k, (g, x) in enumerate(zip(tag, t)):
    print('%3d %-16s t=%8d clk  +%6d' % (k, names.get(int(g), str(g)), x, x - prev))
    prev = x
