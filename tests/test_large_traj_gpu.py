"""GPU: the drop-in manager against the large-configuration trajectories recorded from the reference -- MIND-shaped
(g13: E=16, D=256, minibatch 262 144: the gradient-pass + flat-Adam sequence, four row chunks, one class per lane) and
MovieLens at full size (g14: 6 040 x 3 706, E=8, D=128, minibatch 65 536, scheduled alpha, graph replay).
Tolerances: tests/large_traj_fixture.py (1e-5 on the data-loss terms; report terms bounded by the reference's own
measured error / thread spread); E-step: bit-exact vs the oracle on the same tables, near-tie rule vs the reference."""
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager
from large_traj_fixture import check_losses, exact_reg_terms, load
from oracle import oracle as O

# share of the rows an E-step may assign differently from the reference after the recorded trajectory (every such row has a
# relative distance gap < 2e-5, checked on a sample): 2 x the measured share (INVPREF_TOL_REPORT=1 prints it)
MAX_MISMATCH_SHARE = {'g13': 3e-5, 'g14': 6.5e-3}     # measured: 8 of 786 432 rows = 1.0e-5; 3 240 of 1 048 576 = 3.1e-3

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


class StubEvaluator:
    def evaluate(self):
        return {}


@pytest.mark.parametrize('case', ['g14', 'g13'])
def test_manager_vs_reference_large_trajectory(case):
    z, (U, I, E, D, bs, epochs, seed, n), data, tabs, cf = load(case)
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=False, reg_env_embed=True)
    model.load_state_dict({k: torch.from_numpy(tabs[k]) for k in O.PARAM_NAMES})
    np.random.seed(seed)
    alpha = None if np.isnan(cf[5]) else float(cf[5])
    mgr = ImplicitTrainManager(model=model, evaluator=StubEvaluator(), device=DEV, training_data=torch.from_numpy(data).to(DEV),
                               batch_size=bs, epochs=epochs, cluster_interval=epochs, evaluate_interval=10 ** 9,
                               lr=float(cf[6]), invariant_coe=float(cf[0]), env_aware_coe=float(cf[1]), env_coe=float(cf[2]),
                               L2_coe=float(cf[3]), L1_coe=float(cf[4]), alpha=alpha, use_class_re_weight=True,
                               use_recommend_re_weight=True, cluster_use_random_sort=False)
    env0 = z['env0'].astype(np.int64)
    np.testing.assert_array_equal(mgr.envs.cpu().numpy(), env0)
    assert mgr.use_plan and not mgr._unfused
    mgr.stat_envs()
    steps, ep = [], []
    for _ in range(epochs):
        d = mgr.train_epochs(1)[0]
        ep.append([d[k] for k in ('invariant_loss', 'env_aware_loss', 'envs_loss', 'L2_reg', 'L1_reg', 'loss')])
        steps.append(mgr._epoch_losses[0].cpu().numpy().astype(np.float64))
    ex0 = exact_reg_terms(tabs, data[:bs, 0], data[:bs, 1], env0[:bs], E, D)
    check_losses(np.concatenate(steps), z, 'step_losses', cf, ex0)
    check_losses(np.array(ep), z, 'epoch_losses', cf, ex0)
    if epochs > 1:
        assert mgr._graphs
    diff = mgr.cluster()
    cnt = mgr.stat_envs()
    got = mgr.envs.cpu().numpy()
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    # HIP == oracle, bit for bit, on the tables the manager ended with (a sample of rows keeps the CPU side short)
    rs = np.random.RandomState(1)
    sel = np.sort(rs.choice(n, 65536, replace=False))
    on, _, _, dist = O.estep(O.Tables(sd), data[sel, 0], data[sel, 1], data[sel, 2], True, want_dist=True)
    np.testing.assert_array_equal(got[sel], on)
    ref = z['env_after'].astype(np.int64)
    mm = np.nonzero(got[sel] != ref[sel])[0]
    if len(mm):
        assert ((dist[mm, ref[sel][mm]] - dist[mm, on[mm]]) / dist[mm, on[mm]]).max() < 2e-5
    total_mm = int((got != ref).sum())
    if os.environ.get('INVPREF_TOL_REPORT'):
        print(f'TOL large trajectory {case}: rows assigned differently from the reference {total_mm} of {n} = {total_mm / n:.2e}')
    assert total_mm < MAX_MISMATCH_SHARE[case] * n and abs(diff - int(z['diff_num'])) <= total_mm
    assert np.abs(np.array([cnt[e] for e in range(E)]) - z['counts']).sum() <= 2 * total_mm
    lr = float(cf[6])
    for k in O.PARAM_NAMES:
        arr = sd[k]
        ref_p = z['final_' + k]
        g = arr[z['urows']] if 'user' in k else (arr[z['irows']] if 'item' in k else arr)
        assert np.abs(g - ref_p).max() < 0.05 * lr + 50 * float(z['spread_' + k][0]), k
