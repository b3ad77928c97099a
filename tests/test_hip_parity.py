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

# g5 (MIND shape, one step against the reference's recorded gradients): embed_env's gradient -- per environment a sum over
# ~16 000 interactions in fp32 against the reference's fp32 autograd -- is the one entry above 2e-5 of the table's largest:
# measured 1.45e-4 (INVPREF_TOL_REPORT=1 prints every table's error; the classifier's is 8e-8), bound = 2 x that.
TOL_G5_EMBED_ENV = 3e-4

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
    # vs the reference's own assignments on this fixture (healthy margins: SURVEY 8(c) G2): every one of the 250 154 rows
    # (measured: HIP == oracle bitwise, oracle == reference on all rows, both kinds)
    assert len(mism) == 0


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


# ------------------------------------------------------------------ planned, atomic-free rows path
PER_SLICE = 2
from invpref_kdd_2022_amd import plan as planlib  # noqa: E402


@pytest.mark.parametrize('path', G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_rows_grad_matches_oracle_and_is_reproducible(path):
    z, implicit, params, (roe, ree, cls_w, rec_w) = _load(path)
    U, I, E, D, B = [int(x) for x in z['meta'][:5]]
    P = dev_params(params)
    ws = ops.Workspace(DEV)
    flags = ops.flags_of(implicit, rec_w, cls_w, roe, ree)
    dp = planlib.upload(planlib.build_row_plan(z['u'], z['v'], z['y'], U, I, factor_num=D, per_slice=PER_SLICE), DEV)
    outs = []
    for _ in range(2):
        Gd = [torch.full_like(p, 7.0) for p in P]  # garbage: the kernel must overwrite every row
        losses = torch.zeros(6, device=DEV)
        ops.mstep_rows_grad(P, Gd, dp, t64(z['e']), t32(z['y']), t32(z['w']), B, z['coefs'], flags, losses, ws)
        outs.append(([g.cpu().numpy() for g in Gd], losses.cpu().numpy()))
    og, ol = O.mstep(O.Tables(params), z['u'], z['v'], z['e'], z['y'], z['w'], z['coefs'],
                     O.flags_of(implicit, rec_w, cls_w, roe, ree))
    np.testing.assert_allclose(outs[0][1], ol, rtol=1e-5)
    np.testing.assert_allclose(outs[0][1], z['losses_f32'], rtol=1e-5)
    for k, g, o in zip(O.PARAM_NAMES, outs[0][0], og):
        assert _relerr(g, o) < 2e-5, k
        assert _relerr(g, z['g_f32_' + k]) < 2e-5, k
    # no float atomics anywhere on this path (registers, plain stores, fixed-order sums): every gradient and every
    # loss term is bitwise reproducible run to run
    for a, b in zip(outs[0][0], outs[1][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('path', G1[::2], ids=[os.path.basename(p)[3:-4] for p in G1[::2]])
def test_rows_fused_adam_equals_grad_then_adam(path):
    """fused (M-step + Adam in one pass) == planned gradient followed by the stand-alone Adam kernel,
    to a few ulp; three steps, parameters double-buffered."""
    z, implicit, params, (roe, ree, cls_w, rec_w) = _load(path)
    U, I, E, D, B = [int(x) for x in z['meta'][:5]]
    lr = float(z['coefs'][6])
    ws = ops.Workspace(DEV)
    flags = ops.flags_of(implicit, rec_w, cls_w, roe, ree)
    dp = planlib.upload(planlib.build_row_plan(z['u'], z['v'], z['y'], U, I, factor_num=D, per_slice=PER_SLICE), DEV)
    e, y, w = t64(z['e']), t32(z['y']), t32(z['w'])
    # path A: fused, ping-pong buffers
    A = [dev_params(params), [torch.zeros_like(p) for p in dev_params(params)]]
    mA = [torch.zeros_like(p) for p in A[0]]; vA = [torch.zeros_like(p) for p in A[0]]
    # path B: grad + adam in place (per-table adam on 16-byte aligned clones)
    Bp = dev_params(params)
    mB = [torch.zeros_like(p) for p in Bp]; vB = [torch.zeros_like(p) for p in Bp]
    la, lb = torch.zeros(6, device=DEV), torch.zeros(6, device=DEV)
    for step in (1, 2, 3):
        cur, nxt = A[(step - 1) % 2], A[step % 2]
        la.zero_(); lb.zero_()
        ops.mstep_rows_adam(cur, nxt, mA, vA, dp, e, y, w, B, z['coefs'], flags, la, step, lr, ws)
        Gd = [torch.empty_like(p) for p in Bp]
        ops.mstep_rows_grad(Bp, Gd, dp, e, y, w, B, z['coefs'], flags, lb, ws)
        for p, g, m, v in zip(Bp, Gd, mB, vB):
            n = p.numel()
            pad = (n + 3) // 4 * 4
            bufs = [torch.zeros(pad, device=DEV) for _ in range(4)]
            for bsrc, bdst in zip((p, g, m, v), bufs):
                bdst[:n] = bsrc.reshape(-1)
            ops.adam_(bufs[0], bufs[1], bufs[2], bufs[3], step, lr, zero_grad=False)
            p.copy_(bufs[0][:n].view_as(p)); m.copy_(bufs[2][:n].view_as(m)); v.copy_(bufs[3][:n].view_as(v))
        np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-6)
        for i, (k, pa, pb) in enumerate(zip(O.PARAM_NAMES, nxt, Bp)):
            # the fused pass uses the hardware sqrt/rcp units in its Adam (~1 ulp each); an element whose
            # gradient is ~eps-sized can differ by a fraction of lr (see the oracle-vs-golden Adam test)
            dlt = np.abs(pa.cpu().numpy() - pb.cpu().numpy())
            assert dlt.max() < 0.02 * lr, f'{k} step {step}'
            assert np.quantile(dlt, 0.99) < 1e-6 * max(1.0, float(pa.abs().max())), f'{k} step {step}'
            pb.copy_(pa); mB[i].copy_(mA[i]); vB[i].copy_(vA[i])  # keep both paths on identical state
    # and against the reference's parameters after three Adam steps (same tolerance as the oracle test)
    for k, pa in zip(O.PARAM_NAMES, A[1]):
        assert np.abs(pa.cpu().numpy() - z['adam3_f32_' + k]).max() < 0.05 * lr, k


def test_rows_path_mind_shape_g5():
    """Planned path at the MIND-shaped fixture (E=16, D=256, B=262 144: 64-lane groups, the E x D sums through LDS
    records, item rows with thousands of interactions) against the reference's recorded step."""
    z = np.load(os.path.join(G, 'g5_mind_like_step.npz'))
    U, I, E, D, B, seed = [int(x) for x in z['meta']]
    data = synth.interactions(seed, U, I, B, implicit=True, zipf=True)
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.05)
    P = dev_params(tabs)
    env0 = z['env0'].astype(np.int64)
    ws = ops.Workspace(DEV)
    _, _, sw = ops.stat_envs(t64(env0), E, ws)
    dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D), DEV)
    Gd = [torch.full_like(p, 3.0) for p in P]
    losses = torch.zeros(6, device=DEV)
    ops.mstep_rows_grad(P, Gd, dp, t64(env0), t32(data[:, 2]), sw, B, z['coefs'],
                        ops.flags_of(True, True, True, False, True), losses, ws)
    L = losses.cpu().numpy()
    np.testing.assert_allclose(L[:3], z['losses'][:3], rtol=1e-5)
    for k, g in zip(O.PARAM_NAMES, Gd):
        g = g.cpu().numpy()
        gn = np.sqrt((g.astype(np.float64) ** 2).sum())
        assert abs(gn - float(z['gnorm_' + k])) < 1e-4 * float(z['gnorm_' + k]), k
        if 'user' in k:
            ref, got = z['grows_' + k], g[z['urows']]
        elif 'item' in k:
            ref, got = z['grows_' + k], g[z['irows']]
        else:
            ref, got = z['g_' + k], g
        tol = TOL_G5_EMBED_ENV if 'embed_env.' in k else 2e-5
        if os.environ.get('INVPREF_TOL_REPORT'):
            print(f'TOL g5 {k}: measured {np.abs(got - ref).max() / np.abs(ref).max():.3e} (bound {tol:.1e})')
        assert np.abs(got - ref).max() < tol * np.abs(ref).max() + 1e-9, k


def test_rows_path_movielens_shape_default_plan():
    """MovieLens-class step (U=6 040, I=3 706, E=8, D=128, B=65 536) through the planned fused
    pass with the DEFAULT plan parameters for that minibatch size (16-lane groups of two float4 per lane,
    csrc/step_wide.hpp) against oracle gradient + oracle Adam."""
    U, I, E, D, B = 6040, 3706, 8, 128, 65536
    data = synth.interactions(31, U, I, B, implicit=True, zipf=False)
    tabs = synth.tables(32, U, I, E, D, std=0.1)
    envs = np.random.RandomState(33).randint(0, E, B).astype(np.int64)
    pl = planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D)
    assert pl['lanes_per_group'] == 16 and len(pl['item_desc']) > 0
    dp = planlib.upload(pl, DEV)
    P = dev_params(tabs)
    P2, M, V = ([torch.zeros_like(p) for p in P] for _ in range(3))
    ws = ops.Workspace(DEV)
    _, _, sw = ops.stat_envs(t64(envs), E, ws)
    coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
    flags = ops.flags_of(True, True, True, False, True)
    losses = torch.zeros(6, device=DEV)
    lr = 0.005
    ops.mstep_rows_adam(P, P2, M, V, dp, t64(envs), t32(data[:, 2]), sw, B, coefs, flags, losses, 1, lr, ws)
    tab = O.Tables(tabs)
    _, _, osw = O.stat_envs(envs, E)
    og, ol = O.mstep(tab, data[:, 0], data[:, 1], envs, data[:, 2], osw, coefs, O.flags_of(True, True, True, False, True))
    np.testing.assert_allclose(losses.cpu().numpy(), ol, rtol=1e-5)
    for k, p0, g, p2 in zip(O.PARAM_NAMES, tab.arrs, og, P2):
        po = p0.reshape(-1).copy()
        O.adam(po, g.reshape(-1), np.zeros_like(po), np.zeros_like(po), 1, lr)
        dlt = np.abs(p2.cpu().numpy().reshape(-1) - po)
        assert dlt.max() < 0.05 * lr and np.quantile(dlt, 0.99) < 1e-5, k


def test_plan_parameters_change_nothing_but_the_order_of_sums():
    """The same minibatch under different plans (slice lengths, rounds per task, class order, share of the streamed rows
    per launch): every result agrees to the reordering of float sums, each plan is bitwise reproducible, all match the
    oracle."""
    z, implicit, params, (roe, ree, cls_w, rec_w) = _load(G1[0])
    U, I, E, D, B = [int(x) for x in z['meta'][:5]]
    P = dev_params(params)
    ws = ops.Workspace(DEV)
    flags = ops.flags_of(implicit, rec_w, cls_w, roe, ree)
    lr = float(z['coefs'][6])
    outs = []
    for kw in (dict(per_slice=1, item_per_slice=1, n_classes=1, stream_split=0.0, push=False),
               dict(per_slice=2, item_per_slice=3, rounds_per_task=3, item_rounds_per_task=2, stream_split=1.0, push=True),
               dict(per_slice=64, item_per_slice=64, n_classes=3, push=False),
               dict(per_slice=3, item_per_slice=5, push=True),
               dict(env_num=E)):   # the plan's own choices, launch 1 filled to its residency where that applies
        dp = planlib.upload(planlib.build_row_plan(z['u'], z['v'], z['y'], U, I, factor_num=D, **kw), DEV)
        rep = []
        for _ in range(2):
            Q = [torch.zeros_like(p) for p in P]
            M = [torch.zeros_like(p) for p in P]
            V = [torch.zeros_like(p) for p in P]
            Gd = [torch.full_like(p, 7.0) for p in P]
            la, lb = torch.zeros(6, device=DEV), torch.zeros(6, device=DEV)
            ops.mstep_rows_grad(P, Gd, dp, t64(z['e']), t32(z['y']), t32(z['w']), B, z['coefs'], flags, la, ws)
            ops.mstep_rows_adam(P, Q, M, V, dp, t64(z['e']), t32(z['y']), t32(z['w']), B, z['coefs'], flags, lb, 1, lr, ws)
            rep.append(([g.cpu().numpy() for g in Gd], [q.cpu().numpy() for q in Q], [m.cpu().numpy() for m in M],
                        la.cpu().numpy(), lb.cpu().numpy()))
        for x, y in zip(rep[0][:3], rep[1][:3]):
            for a, b in zip(x, y):
                np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(rep[0][3], rep[1][3])
        np.testing.assert_array_equal(rep[0][4], rep[1][4])
        outs.append(rep[0])
    og, ol = O.mstep(O.Tables(params), z['u'], z['v'], z['e'], z['y'], z['w'], z['coefs'], O.flags_of(implicit, rec_w, cls_w, roe, ree))
    for form in outs:
        np.testing.assert_allclose(form[3], ol, rtol=1e-5)
        np.testing.assert_allclose(form[4], ol, rtol=1e-5)
        for k, g, o, g0 in zip(O.PARAM_NAMES, form[0], og, outs[0][0]):
            assert _relerr(g, o) < 2e-5, k
            assert _relerr(g, g0) < 1e-5, k
