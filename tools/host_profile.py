#!/usr/bin/env python3
"""Diagnostic (GPU box): where the host time of the eager sharded step sequence goes (cProfile over 10 epochs)."""
import cProfile
import os
import pstats
import sys

os.environ['INVPREF_FORCE_SHARDED_PATH'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from invpref_kdd_2022_amd import synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager

import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29555')
dist.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda:0')
U, I, E, D = 15400, 1000, 4, 64
data = synth.yahoo_like()


class Stub:
    def evaluate(self):
        return {}


model = InvPrefImplicit(U, I, E, D, reg_only_embed=True, reg_env_embed=False)
np.random.seed(1)
mgr = ImplicitTrainManager(model=model, evaluator=Stub(), device=dev, training_data=torch.from_numpy(data).to(dev),
                           batch_size=8192, epochs=10 ** 9, cluster_interval=5, evaluate_interval=10 ** 9, lr=0.005,
                           invariant_coe=3.35, env_aware_coe=9.99, env_coe=9.06, L2_coe=3.13, L1_coe=0.49, alpha=1.9,
                           use_class_re_weight=True, use_recommend_re_weight=False, cluster_use_random_sort=False)
mgr.stat_envs()
mgr.train_epochs(2)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
mgr.train_epochs(10, sync=False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)
dist.destroy_process_group()
