#!/usr/bin/env python3
"""Diagnostic (GPU box): replays of the captured E-step at the Yahoo shape, the plain instance and the reference's default
random tie-break one -- run under `rocprofv3 --kernel-trace --stats` for the kernel-to-kernel comparison
(tools/profile_r05.sh).  INVPREF_EPS_PINNED=0: the permutation indices copied to the device first instead of read from
pinned host memory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
for rs in (False, True):
    mgr = bench.build_manager(dev, 0, 1, random_sort=rs)
    mgr.train_epochs(1)
    for _ in range(40):
        mgr.cluster_and_stat_envs(sync=False)
    torch.cuda.synchronize()
    print('random_sort', rs, 'done')
