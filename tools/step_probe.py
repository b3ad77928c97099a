#!/usr/bin/env python3
"""Diagnostic (GPU box; not part of the product or tests): the planned two-launch step in the cache state of the
training loop -- ping-pong parameter buffers, a different plan every step.
  * per-step time of NB steps captured in one HIP graph (what bench.py times, without the manager around it);
  * with PROBE_STAMPS=1: phase time stamps of the LAST step of a run, per task kind, for both launches.
Shape: PROBE_SHAPE="U,I,E,D,B" (default the Yahoo shape on synth.yahoo_like()); plan parameters through the
INVPREF_PLAN_* environment variables (plan.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device('cuda:0')
want_stamps = os.environ.get('PROBE_STAMPS') == '1'
if want_stamps:
    stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    os.environ['INVPREF_STAMPS'] = hex(stamps.data_ptr())
    if os.environ.get('PROBE_DRAIN') != '1':
        os.environ['INVPREF_STAMPS_NODRAIN'] = '1'
from invpref_kdd_2022_amd import ops, plan as planlib, synth  # noqa: E402

U, I, E, D, B = 15400, 1000, 4, 64, 8192
nb = int(os.environ.get('PROBE_STEPS', '31'))
if os.environ.get('PROBE_SHAPE'):
    U, I, E, D, B = [int(x) for x in os.environ['PROBE_SHAPE'].replace('x', ',').split(',')]
    nb = min(nb, int(os.environ.get('PROBE_STEPS', '4')))
    data = synth.interactions(1, U, I, nb * B, implicit=True, zipf=os.environ.get('PROBE_ZIPF') == '1')
else:
    if os.environ.get('PROBE_B'):          # the Yahoo-like data (its own popularity skew) at another minibatch size
        B = int(os.environ['PROBE_B'])
        nb = max(1, min(nb, len(synth.yahoo_like()) // B))
    data = synth.yahoo_like()[:nb * B]
# what-if orders of a minibatch's interactions (stable sorts): PROBE_USORT=1 by user -- launch 1's position-indexed accesses
# (environment, weight, record store) become sequential along the user list; PROBE_ISORT=1 by item -- launch 2's record reads do
for key, col in (('PROBE_USORT', 0), ('PROBE_ISORT', 1)):
    if os.environ.get(key) == '1':
        for k in range(nb):
            blk = data[k * B:(k + 1) * B]
            data[k * B:(k + 1) * B] = blk[np.argsort(blk[:, col], kind='stable')]
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
N = nb * B
y = torch.from_numpy(data[:N, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, N).astype(np.int64)).to(dev)
w = torch.rand(N, device=dev)
BY_ENV = os.environ.get('PROBE_BY_ENV', '1') == '1'   # weights as the managers' epochs take them: class_weights[env] in the kernel
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)
if BY_ENV:
    from invpref_kdd_2022_amd import _capi
    flags |= _capi.WEIGHTS_BY_ENV
    _, cw_env, _ = ops.stat_envs(e, E, ops.Workspace(dev), want_sample_weights=False)


def wts(sl):
    return cw_env if BY_ENV else w[sl]
pls = [planlib.build_row_plan(data[k * B:(k + 1) * B, 0], data[k * B:(k + 1) * B, 1], data[k * B:(k + 1) * B, 2], U, I,
                              factor_num=D, env_num=E) for k in range(nb)]
plans = [planlib.upload(p, dev) for p in pls]
Pn = sum(p.numel() for p in P)
nbytes = B * (32 + 16 * D) + 24 * Pn


def run_steps(first=0, last=None, flush=True, reset=True):
    a, b = P, P2
    for k in range(first, nb if last is None else last):
        sl = slice(k * B, (k + 1) * B)
        ops.mstep_rows_adam(a, b, M, V, plans[k], e[sl], y[sl], wts(sl), B, coefs, flags, losses, 5 + k, 0.005, ws)
        a, b = b, a
    return nb


def graph_time(reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run_steps(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            n = run_steps(reset=False)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                g.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / (reps * n) * 1e3)
    return best


p0 = pls[0]
print(f'shape U={U} I={I} E={E} D={D} B={B}: lanes {p0["lanes_per_group"]}, per_slice {p0["per_slice"]}/{p0["item_per_slice"]}, '
      f'rounds/task {p0["user_rounds_per_task"]}/{p0["item_rounds_per_task"]}, stream rows/task {p0["rows_per_stream_task"]}, '
      f'split {p0["stream_split"]:.2f}; workgroups launch 1 {planlib.launch_workgroups(p0, 0)}, launch 2 {planlib.launch_workgroups(p0, 1)}')
if os.environ.get('PROBE_EAGER') == '1':   # (profiling passes: every launch issued eagerly, a few times)
    for _ in range(3):
        run_steps()
    torch.cuda.synchronize()
    print(f'algorithmic bytes per step {nbytes}')
    sys.exit(0)
if not want_stamps:
    us = graph_time()
    print(f'{os.environ.get("INVPREF_LIB", "default")}: {us:.2f} us per step = {nbytes / us / 1e3:.0f} GB/s of algorithmic bytes = '
          f'{nbytes / us / 1e3 / 8000:.3f} of 8 TB/s')
    sys.exit(0)

ks = int(os.environ.get('PROBE_STAMP_STEP', str(max(0, nb - 2))))   # (the last minibatch of an epoch is a short one)
for rep in range(3):
    run_steps(0, ks)
    stamps.zero_()
    a, b = (P, P2) if ks % 2 == 0 else (P2, P)
    sl = slice(ks * B, (ks + 1) * B)
    ops.mstep_rows_adam(a, b, M, V, plans[ks], e[sl], y[sl], wts(sl), B, coefs, flags, losses, 5 + ks, 0.005, ws)
torch.cuda.synchronize()
pl = pls[ks]
ncls, cls = pl['n_classes'], np.asarray(pl['cls'])
raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.int64)
t0 = None
for launch, name in ((0, 'launch 1 (eval)'), (1, 'launch 2 (apply)')):
    rpt = pl['user_rounds_per_task'] if launch == 0 else pl['item_rounds_per_task']
    spt = pl['rows_per_stream_task'] if launch == 0 else pl['rows_per_stream_task2']
    wg = planlib.launch_workgroups(pl, launch)
    extra = 0 if launch == 0 else 64
    st = raw[launch * 8192: launch * 8192 + wg + extra]
    kind = np.full(len(st), 'pad', dtype=object)
    for bk in range(wg):
        c, j = bk % ncls, bk // ncls
        tj = -(-int(cls[c, 4 * launch + 1]) // rpt)
        kind[bk] = 'job' if j < tj else ('stream' if (j - tj) * spt < cls[c, 4 * launch + 3] else 'pad')
    kind[wg:] = 'fold'
    live = st[:, 0] > 0
    if t0 is None:
        t0 = st[live, 0].min()
    end = np.where(st[:, 7] > 0, st[:, 7], st[:, 6])
    print(f'== {name}: grid {wg}; ' + ', '.join(f'{k} {int(((kind == k) & live).sum())}' for k in ('job', 'stream', 'fold'))
          + f'; first start {(st[live, 0].min() - t0) / 100:.2f} us, last end {(end[live].max() - t0) / 100:.2f} us')
    for nm in ('job', 'stream', 'fold'):
        sel = (kind == nm) & live
        if not sel.any():
            continue
        s0, e0 = (st[sel, 0] - t0) / 100, (end[sel] - t0) / 100
        life = e0 - s0
        print(f'  {nm:6s} n={sel.sum():4d} start med {np.median(s0):5.2f} p90 {np.quantile(s0, .9):5.2f} max {s0.max():5.2f} | life med '
              f'{np.median(life):5.2f} p90 {np.quantile(life, .9):5.2f} max {life.max():5.2f} | end med {np.median(e0):5.2f} p90 '
              f'{np.quantile(e0, .9):5.2f} max {e0.max():5.2f}')
    j = st[(kind == 'job') & live].astype(np.float64)
    if len(j):
        ph = np.diff(j[:, :7], axis=1) / 100
        tail = (j[:, 7] - j[:, 6]) / 100 if launch == 0 else np.zeros(len(j))
        print('  job phases (us, median): stage issue %.2f | descriptor %.2f | gathers+sync %.2f | interactions %.2f | slice meet %.2f | '
              'adam+store %.2f | partial slab %.2f' % (tuple(np.median(ph, axis=0)) + (np.median(tail),)))
        if os.environ.get('PROBE_TOP'):
            print('  job phases (us, p90):    ' + ' | '.join('%.2f' % x for x in np.quantile(ph, .9, axis=0)) +
                  ' ; slowest tenth of the jobs (median): ' +
                  ' | '.join('%.2f' % x for x in np.median(ph[(j[:, 6] - j[:, 0]) >= np.quantile(j[:, 6] - j[:, 0], .9)], axis=0)))
            desc = pl['user_desc'] if launch == 0 else pl['item_desc']
            idx = np.flatnonzero((kind == 'job') & live)
            order = idx[np.argsort(-(end[idx] - st[idx, 0]))][:int(os.environ['PROBE_TOP'])]
            for bk in order:
                c, jj = bk % ncls, bk // ncls
                rnd = int(cls[c, 4 * launch]) + jj * rpt
                meta = desc[rnd, :, 1]
                cnts = (meta >> 9)[desc[rnd, :, 0] >= 0]
                print('    wg %4d start %.2f life %.2f phases %s | round %d: slices %d, rows %s' % (
                    bk, (st[bk, 0] - t0) / 100, (end[bk] - st[bk, 0]) / 100,
                    ' '.join('%.2f' % x for x in np.diff(st[bk, :7].astype(np.float64)) / 100), rnd, (meta[0] >> 1) & 31,
                    sorted(set(cnts.tolist()), reverse=True)[:4]))
