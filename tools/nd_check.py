"""Diagnostic (GPU box): run-to-run reproducibility of the manager over 1..5 epochs, graph vs eager."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from invpref_kdd_2022_amd import synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager, LOSS_KEYS
from oracle import oracle as O
import test_manager_gpu as T
z = np.load(os.path.join(T.G, 'g4_yahoo_like_traj.npz'))
U, I, E, D, bs, epochs, seed = [int(x) for x in z['meta']]
data = synth.yahoo_like(seed)[:40000]
tabs = synth.tables(seed + 7, U, I, E, D, std=0.05)


def run(n):
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=False, reg_env_embed=True)
    model.load_state_dict({k: torch.from_numpy(tabs[k]) for k in O.PARAM_NAMES})
    np.random.seed(seed)
    mgr = T._mgr(ImplicitTrainManager, model, data, z, use_class_re_weight=True, use_recommend_re_weight=True)
    mgr.stat_envs()
    tr = [mgr.train_a_epoch() for _ in range(n)]
    return np.array([[e[k] for k in LOSS_KEYS] for e in tr]), {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


for n in (1, 2, 4):
    a, b = run(n), run(n)
    print(os.environ.get('INVPREF_NO_GRAPH', '0'), 'epochs', n, 'loss rel per epoch', np.abs(a[0] / b[0] - 1).max(axis=1),
          'param max diff', {k.split('.')[0][6:]: float(np.abs(a[1][k] - b[1][k]).max()) for k in O.PARAM_NAMES[:4]},
          'absmax', float(np.abs(a[1][O.PARAM_NAMES[0]]).max()))
