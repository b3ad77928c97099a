#!/usr/bin/env python3
"""Diagnostic (GPU box): where ImplicitTestManager.evaluate() spends its time at the MIND test shape (50 000 test users x
51 283 items, D = 256, top-k 5/10/20/40): predict, top-k selection, read-back + numpy metrics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from invpref_kdd_2022_amd.evaluate import ImplicitTestManager, recall_precision_ndcg
from invpref_kdd_2022_amd.models import InvPrefImplicit
dev = torch.device('cuda:0')
nu, ni, nd, n_test, tb, topk = 50000, 51283, 256, 50000, 256, [5, 10, 20, 40]
loader = bench.test_loader(nu, ni, n_test, 60, 10)
model = InvPrefImplicit(nu, ni, 16, nd).to(dev)
tm = ImplicitTestManager(model, loader, test_batch_size=tb, top_k_list=list(topk))
tm.evaluate(); torch.cuda.synchronize()
t0 = time.perf_counter(); tm.evaluate(); torch.cuda.synchronize(); print('evaluate %.1f ms' % ((time.perf_counter() - t0) * 1e3))
step = max(tb, min(n_test, (1 << 28) // ni))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tp = tk = th = 0.0
for lo in range(0, n_test, step):
    hi = min(lo + step, n_test)
    ev[0].record()
    r = model.predict(tm._users[lo:hi].contiguous())
    ev[1].record()
    _, hits = tm.topk(lo, hi)          # (predict again inside + the selection)
    ev[2].record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    h = hits.cpu().numpy()
    for k in topk:
        recall_precision_ndcg(h, tm._truth_len[lo:hi], k)
    th += time.perf_counter() - t1
    tp += ev[0].elapsed_time(ev[1]); tk += ev[1].elapsed_time(ev[2])
print('batches of %d users: predict %.1f ms, predict + top-k %.1f ms (top-k alone %.1f), read-back + numpy metrics %.1f ms' % (step, tp, tk, tk - tp, th * 1e3))
