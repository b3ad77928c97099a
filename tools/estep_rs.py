#!/usr/bin/env python3
"""tools/estep_rs.py -- cluster(sync=False) at the Yahoo shape with and without cluster_use_random_sort (the reference's
default, train.py:24, :192-196): device time per E-step (HIP events; the permutation indices drawn beforehand, so that the
host's numpy draws -- the reference's own np.random.randint calls, 1 per minibatch -- are timed apart) and wall time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device('cuda:0')
print(bench.estep_random_sort_timing(dev))
