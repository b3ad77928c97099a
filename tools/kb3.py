#!/usr/bin/env python3
"""Diagnostic (GPU box): why an epoch of the training loop costs more per step than the same
kernel pair replayed on one minibatch -- ping-pong parameter buffers, 31 different plans."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth

dev = torch.device('cuda:0')
U, I, E, D, B = 15400, 1000, 4, 64, 8192
data = synth.yahoo_like()
nb = 31
tabs = synth.tables(2, U, I, E, D)
P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
P2 = [p.clone() for p in P]
M = [torch.zeros_like(p) for p in P]
V = [torch.zeros_like(p) for p in P]
N = nb * B
y = torch.from_numpy(data[:N, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, N).astype(np.int64)).to(dev)
w = torch.rand(N, device=dev)
ws = ops.Workspace(dev)
losses = torch.zeros(6, device=dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
flags = ops.flags_of(True, False, True, True, False)
plans = [planlib.upload(planlib.build_row_plan(data[k * B:(k + 1) * B, 0], data[k * B:(k + 1) * B, 1],
                                               data[k * B:(k + 1) * B, 2], U, I), dev) for k in range(nb)]


def graph_time(seq, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        seq(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            n = seq()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * n) * 1e3


import ctypes as C
from invpref_kdd_2022_amd import _capi
L = _capi.lib()
host = np.zeros((8192, 8), np.float32)
L.invpref_adam_schedule_fill(host.ctypes.data, 1, 8192, 0.005, 0.9, 0.999, 1e-8)
s_table = torch.from_numpy(host).to(dev)
st16 = np.zeros(32, np.int32); st16[16:18] = 1; st16[18:26] = host[0].view(np.int32)
s_state0 = torch.from_numpy(st16).to(dev); s_state = s_state0.clone()
s_struct = _capi.AdamSchedule(s_state.data_ptr(), s_table.data_ptr(), 8192)
tabs_c = {id(P): _capi.make_tables(P), id(P2): _capi.make_tables(P2)}
tm, tv = _capi.make_tables(M), _capi.make_tables(V)
cf = _capi.Coefs(*coefs)
ows = ws.get_zeroed(max(L.invpref_rows_workspace_bytes(C.byref(tabs_c[id(P)]), C.byref(p.struct)) for p in plans))


def seq_sched(steps=30):
    def run():
        a, b = P, P2
        st = torch.cuda.current_stream().cuda_stream
        for k in range(steps):
            s_struct.slot = (k + 1) & 1
            rc = L.invpref_mstep_rows_adam_sched_hip(C.byref(tabs_c[id(a)]), C.byref(tabs_c[id(b)]), C.byref(tm), C.byref(tv),
                                                     C.byref(plans[k].struct), e.data_ptr() + 8 * k * B, y.data_ptr() + 4 * k * B,
                                                     w.data_ptr() + 4 * k * B, B, C.byref(cf), flags, losses.data_ptr(),
                                                     C.byref(s_struct), ows.data_ptr(), ows.numel(), st)
            assert rc == 0
            a, b = b, a
        s_state.copy_(s_state0)   # keep the step inside the table over many replays
        return steps
    return run


import copy
def no_stream(dp):
    st = planlib.RowPlanStruct.from_buffer_copy(dp.struct)
    st.n_stream_user = 0; st.n_stream_item = 0
    d = copy.copy(dp); d.struct = st
    return d
plans_ns = [no_stream(p) for p in plans]


def seq(pingpong, multiplan, steps=30, pl=None):
    pl = pl or plans
    def run():
        a, b = P, P2
        for k in range(steps):
            kk = k if multiplan else 0
            sl = slice(kk * B, (kk + 1) * B)
            ops.mstep_rows_adam(a, b, M, V, pl[kk], e[sl], y[sl], w[sl], B, coefs, flags, losses, 5, 0.005, ws)
            if pingpong:
                a, b = b, a
        return steps
    return run


combos = ((False, False), (True, True)) if os.environ.get('KB3_SHORT') else ((False, False), (False, True), (True, False), (True, True))
empty = planlib.upload(planlib.build_row_plan(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32), U, I), dev)
def seq_stream_only(steps=30):
    def run():
        a, b = P, P2
        for k in range(steps):
            ops.mstep_rows_adam(a, b, M, V, empty, e[:1], y[:1], w[:1], 1, coefs, flags, losses, 5, 0.005, ws)
            a, b = b, a
        return steps
    return run
try:
    print('stream tasks only (every row untouched: pure fused-Adam streaming, 50 MB) pp=1: %.2f us, tasks %d' % (graph_time(seq_stream_only()), empty.n_tasks))
except Exception as ex:
    print('stream-only failed:', ex)
print('without stream tasks pp=1 mp=1: %.2f us' % graph_time(seq(True, True, pl=plans_ns)))
print('sched variant pp=1 mp=1: %.2f us' % graph_time(seq_sched()))
print(os.environ.get('INVPREF_LIB', 'default'), ' '.join(f'pp={int(pp)} mp={int(mp)}: {graph_time(seq(pp, mp)):.2f} us' for pp, mp in combos))
