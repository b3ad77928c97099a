"""Seeded inputs of the g7 PureMF goldens: shared by tests/golden/gen_goldens.py (which runs the
reference on them) and the tests (which run the oracle / the HIP path on them)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from invpref_kdd_2022_amd import synth  # noqa: E402

LOSS_KEYS = ['score_loss', 'L2_reg', 'L1_reg', 'loss']  # reference train.py:399-404


def pure_mf_inputs(kind):
    U, I, D, n, bs, epochs = 400, 250, 24, 12000, 2048, 6
    data = synth.interactions(4242 if kind == 'implicit' else 4343, U, I, n, implicit=(kind == 'implicit'))
    rs = np.random.RandomState(77)
    init = {'user_emb.weight': (rs.standard_normal((U, D)) * 0.1).astype(np.float32),
            'item_emb.weight': (rs.standard_normal((I, D)) * 0.1).astype(np.float32)}
    return (U, I, D, n, bs, epochs), data, init, dict(lr=0.01, L2_coe=0.05, L1_coe=0.01)
