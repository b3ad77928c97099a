"""Seeded inputs of the g11 goldens (cluster() with cluster_use_random_sort=True): shared by
tests/golden/gen_goldens.py (which runs the reference on them) and the tests."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invpref_kdd_2022_amd import synth  # noqa: E402


def random_sort_case(E):
    """Inputs of g11 (shared with the tests): an explicit model in which HALF of the users are predicted exactly
    (p + q_e - y = 0 for every environment: an exact tie that only the eps permutation rows can break, train.py:86-92,
    :192-196) and the other half generically."""
    U, I, D, n, bs = 40, 30, 8, 900, 256
    rs = np.random.RandomState(1100 + E)
    users, items = rs.randint(0, U, n), rs.randint(0, I, n)
    cu = rs.randint(1, 6, U)                               # the score of an exactly predicted user
    scores = np.where(users < U // 2, cu[users], rs.randint(1, 6, n))
    data = np.stack([users, items, scores], axis=1).astype(np.int64)
    tabs = synth.tables(1200 + E, U, I, E, D, std=0.3)
    pu = tabs['embed_user_invariant.weight']
    qi = tabs['embed_item_invariant.weight']
    qi[:, :] = 0.0
    qi[:, 0] = 1.0
    pu[:U // 2, :] = 0.0
    pu[:U // 2, 0] = cu[:U // 2]
    tabs['embed_user_env_aware.weight'][:U // 2] = 0.0     # q_e = 0 for those users
    return (U, I, D, n, bs), data, tabs
