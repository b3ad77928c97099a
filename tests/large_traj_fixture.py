"""Shared pieces of the large-configuration trajectory tests (goldens g13 / g14, recorded from the reference with 1 and
with 8 torch threads): how inputs are regenerated and how a loss trace is held against the reference's own spread."""
import os

import numpy as np

from invpref_kdd_2022_amd import synth

G = os.path.join(os.path.dirname(__file__), 'golden')
CASES = {'g13': 'g13_mind_like_traj.npz', 'g14': 'g14_movielens_full_traj.npz'}


def load(case):
    z = np.load(os.path.join(G, CASES[case]))
    U, I, E, D, bs, epochs, seed, n = [int(x) for x in z['meta']]
    data = synth.interactions(seed, U, I, n, implicit=True, zipf=True)
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.05)
    cf = z['coefs']   # invariant, env_aware, env, L2, L1, alpha (NaN: the alpha schedule of train.py:214-217), lr
    return z, (U, I, E, D, bs, epochs, seed, n), data, tabs, cf


def exact_reg_terms(tabs, u, v, e, E, D):
    """L2_reg / L1_reg of one minibatch (models.py:328-391, reg_only_embed=False, reg_env_embed=True) in float64"""
    n = float(len(u) * D)
    out = []
    for p in (2, 1):
        def nrm(name, idx):
            return (np.abs(tabs[name][idx].astype(np.float64)) ** p).sum()
        out.append((nrm('embed_user_invariant.weight', u) + nrm('embed_user_env_aware.weight', u)
                    + nrm('embed_item_invariant.weight', v) + nrm('embed_item_env_aware.weight', v)) / (2 * n)
                   + nrm('embed_env.weight', e) / n + nrm('env_classifier.linear_map.weight', slice(None)) / (D * E)
                   + nrm('env_classifier.linear_map.bias', slice(None)) / E)
    return out


def check_losses(got, z, key, cf, exact0=None):
    """got [k, 6] (invariant, env-aware, envs, L2_reg, L1_reg, loss) against the reference's trace z[key].

    * the three data-loss terms: north_star's 1e-5, against the 1-thread AND the 8-thread run of the reference;
    * L2_reg / L1_reg REPORTS: the reference forms them with fp32 norm() over B*D terms (8 M ... 67 M here) and is
      itself off -- its 1-thread and 8-thread runs differ by up to 6e-3 in L1, and both are off from the exact sum by
      what `exact0` shows for the first step (float64 recomputation from the same tables).  Bound: 1e-5, or 4x the
      reference's own thread spread, or 3x the reference's own error on the first step, whichever is largest;
    * the total loss: 1e-5 once the report-term differences are taken out (loss = sum coef_i * term_i)."""
    got = np.asarray(got, np.float64)
    ref1, ref8 = z[key], z[key + '_t8']
    np.testing.assert_allclose(got[:, :3], ref1[:, :3], rtol=1e-5)
    np.testing.assert_allclose(got[:, :3], ref8[:, :3], rtol=1e-5)
    spread = np.abs(ref1 - ref8) / np.abs(ref8)
    own = np.zeros(6)
    if exact0 is not None:
        s1, s8 = z['step_losses'], z['step_losses_t8']
        for col, ex in zip((3, 4), exact0):
            if key == 'step_losses':
                assert abs(got[0, col] - ex) < 1e-6 * ex, (col, got[0, col], ex)   # ours IS the exact sum
            own[col] = max(abs(s1[0, col] - ex), abs(s8[0, col] - ex)) / ex         # the reference's own error
    for col in (3, 4):
        tol = np.maximum(1e-5, np.maximum(4.0 * spread[:, col], 3.0 * own[col]))
        err = np.abs(got[:, col] - ref8[:, col]) / np.abs(ref8[:, col])
        assert (err <= tol).all(), (col, err.max(), tol.max())
    l2c, l1c = float(cf[3]), float(cf[4])
    slack = l2c * np.abs(got[:, 3] - ref1[:, 3]) + l1c * np.abs(got[:, 4] - ref1[:, 4])
    assert (np.abs(got[:, 5] - ref1[:, 5]) <= 1e-5 * np.abs(ref1[:, 5]) + slack).all()
