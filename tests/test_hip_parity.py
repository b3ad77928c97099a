"""GPU: the HIP kernels (through the C ABI) against the CPU oracle on the same inputs, and against
the golden vectors recorded from the reference.

Tolerances: E-step assignments, counts, diff: bit exact (integers).  E-step / forward values: bit
exact vs the oracle (shared canonical arithmetic).  M-step losses: 1e-5 relative
(north_star).  Gradients: 2e-5 of the table's max |g| (float atomics reorder the sums)."""
import glob
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import ops, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')
G1 = sorted(glob.glob(os.path.join(G, 'g1_*.npz')))
DEV = torch.device('cuda:0')


def dev_params(params):
    return [torch.from_numpy(np.ascontiguousarray(params[k], dtype=np.float32)).to(DEV) for k in ops.PARAM_NAMES]


def t64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(DEV)


def t32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def _load(path):
    z = np.load(path)
    U, I, E, D, B, roe, ree, cls_w, rec_w = [int(x) for x in z['meta']]
    kind = 'implicit' if '_implicit_' in path else 'explicit'
    params = {k: z['p_' + k] for k in O.PARAM_NAMES}
    return z, kind == 'implicit', params, (roe, ree, cls_w, rec_w)


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('path', G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_forward_bit_exact_vs_oracle(path):
    z, implicit, params, _ = _load(path)
    P = dev_params(params)
    inv, env, out = ops.forward(P, t64(z['u']), t64(z['v']), t64(z['e']), implicit)
    oi, oe, oo = O.forward(O.Tables(params), z['u'], z['v'], z['e'], implicit)
    np.testing.assert_array_equal(inv.cpu().numpy(), oi)
    np.testing.assert_array_equal(env.cpu().numpy(), oe)
    np.testing.assert_array_equal(out.cpu().numpy(), oo)
    # and within fp32 rounding of the reference's own output
    assert _relerr(inv.cpu().numpy(), z['inv_f32']) < 2e-6
    assert _relerr(out.cpu().numpy(), z['envout_f32']) < 2e-6


@pytest.mark.parametrize('path', G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_mstep_grad_and_adam(path):
    z, implicit, params, (roe, ree, cls_w, rec_w) = _load(path)
    P = dev_params(params)
    Gd = [torch.zeros_like(p) for p in P]
    losses = torch.zeros(6, device=DEV)
    ws = ops.Workspace(DEV)
    flags = ops.flags_of(implicit, rec_w, cls_w, roe, ree)
    B = len(z['u'])
    ops.mstep_grad(P, Gd, t64(z['u']), t64(z['v']), t64(z['e']), t32(z['y']), t32(z['w']), B, z['coefs'], flags,
                   losses, ws)
    og, ol = O.mstep(O.Tables(params), z['u'], z['v'], z['e'], z['y'], z['w'], z['coefs'],
                     O.flags_of(implicit, rec_w, cls_w, roe, ree))
    np.testing.assert_allclose(losses.cpu().numpy(), ol, rtol=1e-5)
    np.testing.assert_allclose(losses.cpu().numpy(), z['losses_f32'], rtol=1e-5)
    for k, g, o in zip(O.PARAM_NAMES, Gd, og):
        assert _relerr(g.cpu().numpy(), o) < 2e-5, k
        assert _relerr(g.cpu().numpy(), z['g_f32_' + k]) < 2e-5, k
    # Adam on a flat copy: bit exact vs the oracle when fed the SAME gradient
    for p0, o in zip(params.values(), og):
        p = t32(p0).reshape(-1).clone()
        n = (p.numel() + 3) // 4 * 4
        pp = torch.zeros(n, device=DEV); pp[:p.numel()] = p
        gg = torch.zeros(n, device=DEV); gg[:p.numel()] = t32(o).reshape(-1)
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        po = np.ascontiguousarray(p0, np.float32).reshape(-1).copy()
        mo, vo = np.zeros_like(po), np.zeros_like(po)
        for step in (1, 2, 3):
            ops.adam_(pp, gg.clone(), m, v, step, float(z['coefs'][6]))
            O.adam(po, np.ascontiguousarray(o, np.float32).reshape(-1), mo, vo, step, float(z['coefs'][6]))
        np.testing.assert_array_equal(pp[:p.numel()].cpu().numpy(), po)
        np.testing.assert_array_equal(m[:p.numel()].cpu().numpy(), mo)


@pytest.mark.parametrize('path', G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_estep_bit_exact_small(path):
    z, implicit, params, _ = _load(path)
    P = dev_params(params)
    ws = ops.Workspace(DEV)
    old = t64(z['e'])
    new, counts, diff, cw, sw = ops.estep(P, t64(z['u']), t64(z['v']), t32(z['y']), implicit, old, ws)
    on, oc, od, _ = O.estep(O.Tables(params), z['u'], z['v'], z['y'], implicit, old_envs=z['e'])
    np.testing.assert_array_equal(new.cpu().numpy(), on)
    np.testing.assert_array_equal(counts.cpu().numpy(), oc)
    assert int(diff.item()) == od
    _, ocw, osw = O.stat_envs(on, len(oc))
    np.testing.assert_array_equal(cw.cpu().numpy(), ocw)
    np.testing.assert_array_equal(sw.cpu().numpy(), osw)


@pytest.mark.parametrize('kind', ['implicit', 'explicit'])
def test_estep_yahoo_shape_bit_exact_and_golden(kind):
    z = np.load(os.path.join(G, f'g2_estep_{kind}.npz'))
    U, I, E, D, n, seed = [int(x) for x in z['meta']]
    data = synth.interactions(seed, U, I, n, implicit=(kind == 'implicit'))
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.3 if kind == 'implicit' else 0.15)
    P = dev_params(tabs)
    ws = ops.Workspace(DEV)
    old = z['old_envs'].astype(np.int64)
    new, counts, diff, cw, sw = ops.estep(P, t64(data[:, 0]), t64(data[:, 1]), t32(data[:, 2]), kind == 'implicit',
                                          t64(old), ws)
    on, oc, od, _ = O.estep(O.Tables(tabs), data[:, 0], data[:, 1], data[:, 2], kind == 'implicit', old_envs=old)
    new = new.cpu().numpy()
    np.testing.assert_array_equal(new, on)                      # HIP == oracle, all 250 154 rows
    np.testing.assert_array_equal(counts.cpu().numpy(), oc)
    assert int(diff.item()) == od
    ref = z['new_envs'].astype(np.int64)
    mism = np.nonzero(new != ref)[0]
    assert set(mism.tolist()) <= set(z['low_margin_rows'].tolist())  # vs reference: only rounding-level rows
    assert len(mism) <= 40


def test_stat_envs_and_sample_weights():
    rs = np.random.RandomState(0)
    envs = rs.randint(0, 7, 100003).astype(np.int64)
    ws = ops.Workspace(DEV)
    counts, cw, sw = ops.stat_envs(t64(envs), 7, ws)
    oc, ocw, osw = O.stat_envs(envs, 7)
    np.testing.assert_array_equal(counts.cpu().numpy(), oc)
    np.testing.assert_array_equal(cw.cpu().numpy(), ocw)
    np.testing.assert_array_equal(sw.cpu().numpy(), osw)
    cw2, sw2 = ops.sample_weights(t64(envs[:5000]), counts, len(envs), 7)
    np.testing.assert_array_equal(cw2.cpu().numpy(), ocw)
    np.testing.assert_array_equal(sw2.cpu().numpy(), osw[:5000])


def test_mstep_g5_mind_shape():
    z = np.load(os.path.join(G, 'g5_mind_like_step.npz'))
    U, I, E, D, B, seed = [int(x) for x in z['meta']]
    data = synth.interactions(seed, U, I, B, implicit=True, zipf=True)
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.05)
    P = dev_params(tabs)
    env0 = z['env0'].astype(np.int64)
    ws = ops.Workspace(DEV)
    _, _, sw = ops.stat_envs(t64(env0), E, ws)
    Gd = [torch.zeros_like(p) for p in P]
    losses = torch.zeros(6, device=DEV)
    ops.mstep_grad(P, Gd, t64(data[:, 0]), t64(data[:, 1]), t64(env0), t32(data[:, 2]), sw, B, z['coefs'],
                   ops.flags_of(True, True, True, False, True), losses, ws)
    L = losses.cpu().numpy()
    np.testing.assert_allclose(L[:3], z['losses'][:3], rtol=1e-5)
    for k, g in zip(O.PARAM_NAMES, Gd):
        g = g.cpu().numpy()
        gn = np.sqrt((g.astype(np.float64) ** 2).sum())
        assert abs(gn - float(z['gnorm_' + k])) < 1e-4 * float(z['gnorm_' + k]), k
