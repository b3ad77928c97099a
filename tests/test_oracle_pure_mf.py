"""CPU: the PureMF baselines as the degenerate case of the InvPref step, pinned by
goldens recorded from the reference's own PureMatrixFactorization / PureExplicitMatrixFactorization
under Basic{Implicit,Explicit}TrainManager (tests/golden/gen_goldens.py g7)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from pure_mf_fixture import pure_mf_inputs

G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('kind', ['implicit', 'explicit'])
def test_first_batch_terms_and_gradients(kind):
    z = np.load(os.path.join(G, f'g7_pure_mf_{kind}.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs(kind)
    params = O.pure_mf_params(init['user_emb.weight'], init['item_emb.weight'])
    for prec, tag, tol in (('f64', 'f64', 1e-12), ('f32', 'f32', 2e-6)):
        tab = O.Tables(params, prec)
        grads, losses = O.mstep(tab, data[:bs, 0], data[:bs, 1], np.zeros(bs, np.int64), data[:bs, 2], None,
                                O.pure_mf_coefs(cfg['L2_coe'], cfg['L1_coe']),
                                O.flags_of(kind == 'implicit', False, False, True, False))
        np.testing.assert_allclose(O.pure_mf_losses(losses), z[f'step_losses_{tag}'], rtol=1e-12 if prec == 'f64' else 1e-5)
        for g, name in ((grads[0], 'user'), (grads[1], 'item')):
            ref = z[f'step_g_{name}_{tag}']
            assert np.abs(g - ref).max() <= tol * np.abs(ref).max(), (prec, name)
        for g in grads[2:]:  # the zero tables get exactly-zero gradients
            assert not g.any()
    inv, _, _ = O.forward(O.Tables(params), data[:512, 0], data[:512, 1], np.zeros(512, np.int64), kind == 'implicit')
    np.testing.assert_allclose(inv, z['step_scores'], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('kind', ['implicit', 'explicit'])
def test_trajectory(kind):
    z = np.load(os.path.join(G, f'g7_pure_mf_{kind}.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs(kind)
    tr = O.pure_mf_trainer(init['user_emb.weight'], init['item_emb.weight'], data, implicit=(kind == 'implicit'),
                           batch_size=bs, lr=cfg['lr'], L2_coe=cfg['L2_coe'], L1_coe=cfg['L1_coe'])
    trace = O.pure_mf_losses(np.stack([tr.train_a_epoch() for _ in range(epochs)]))
    np.testing.assert_allclose(trace, z['traj'], rtol=1e-5)
    for arr, key in ((tr.tab.arrs[0], 'final_user_emb.weight'), (tr.tab.arrs[1], 'final_item_emb.weight')):
        assert np.abs(arr - z[key]).max() < 2e-4 * cfg['lr'] / 0.01 + 1e-5
    for arr in tr.tab.arrs[2:]:
        assert not arr.any()  # the degenerate tables never move
