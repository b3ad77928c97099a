"""GPU: the ``torch.ops.invpref.*`` custom operators (SURVEY §8(b) level 3) called directly -- schema, fake
tensors and aliasing via ``torch.library.opcheck``, results against the CPU oracle on the same inputs.

Tolerances as in test_hip_parity.py: integers / forward values bit exact, losses 1e-5, gradients 2e-5 of max."""
import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth, torch_ops
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
U, I, E, D, B = 300, 120, 4, 64, 1024
COEFS = [3.35, 9.99, 9.06, 3.13, 0.49, 1.9]


def _setup(implicit=True, seed=0):
    data = synth.interactions(seed + 1, U, I, B, implicit=implicit)
    tabs = synth.tables(seed + 2, U, I, E, D, std=0.2)
    envs = np.random.RandomState(seed + 3).randint(0, E, B).astype(np.int64)
    w = np.random.RandomState(seed + 4).rand(B).astype(np.float32)
    P = [torch.from_numpy(tabs[k]).to(DEV) for k in ops.PARAM_NAMES]
    t = dict(u=torch.from_numpy(np.ascontiguousarray(data[:, 0])).to(DEV),
             v=torch.from_numpy(np.ascontiguousarray(data[:, 1])).to(DEV),
             y=torch.from_numpy(data[:, 2].astype(np.float32)).to(DEV), e=torch.from_numpy(envs).to(DEV),
             w=torch.from_numpy(w).to(DEV))
    return data, tabs, envs, w, P, t


def _ws(nbytes=1 << 26, zero=False):
    return (torch.zeros if zero else torch.empty)(nbytes, dtype=torch.uint8, device=DEV)


def _relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_registered_names_and_no_cpu_kernel():
    for n in torch_ops.NAMES:
        assert hasattr(torch.ops.invpref, n), n
    for n in ('train_step_fused', 'estep_assign', 'adam_dense_'):      # the three SURVEY §8(b) names
        assert n in torch_ops.NAMES
    x = torch.zeros(8)
    with pytest.raises((NotImplementedError, RuntimeError)):            # CPU tensors: no kernel, no fallback
        torch.ops.invpref.adam_dense_(x, x.clone(), x.clone(), x.clone(), 1, 0.1, 0.9, 0.999, 1e-8, True)


@pytest.mark.parametrize('implicit', [True, False])
def test_train_step_fused_and_adam_vs_oracle(implicit):
    data, tabs, envs, w, P, t = _setup(implicit)
    G = [torch.zeros_like(p) for p in P]
    losses = torch.zeros(6, device=DEV)
    flags = ops.flags_of(implicit, True, True, False, True)
    torch.ops.invpref.train_step_fused(P, G, t['u'], t['v'], t['e'], t['y'], t['w'], B, COEFS, flags, losses, _ws())
    og, ol = O.mstep(O.Tables(tabs), data[:, 0], data[:, 1], envs, data[:, 2], w, COEFS,
                     O.flags_of(implicit, True, True, False, True))
    np.testing.assert_allclose(losses.cpu().numpy(), ol, rtol=1e-5)
    for k, g, o in zip(ops.PARAM_NAMES, G, og):
        assert _relerr(g.cpu().numpy(), o) < 2e-5, k
    # adam_dense_: bit exact vs the oracle's Adam fed the same gradient; zero_grad clears the gradient
    p = P[0].reshape(-1).clone()
    g = torch.from_numpy(np.ascontiguousarray(og[0], np.float32).reshape(-1)).to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    po = tabs[ops.PARAM_NAMES[0]].reshape(-1).copy()
    mo, vo = np.zeros_like(po), np.zeros_like(po)
    for step in (1, 2):
        gg = g.clone()
        torch.ops.invpref.adam_dense_(p, gg, m, v, step, 0.005, 0.9, 0.999, 1e-8, True)
        O.adam(po, np.ascontiguousarray(og[0], np.float32).reshape(-1), mo, vo, step, 0.005)
        assert float(gg.abs().max()) == 0.0
    np.testing.assert_array_equal(p.cpu().numpy(), po)
    np.testing.assert_array_equal(v.cpu().numpy(), vo)


def test_adam_dense_unaligned_ranges():
    """ranges that start anywhere (a user-sharded rank's rows of a table with factor_num % 4 != 0)"""
    n = 4096 + 7
    rs = np.random.RandomState(0)
    base = [rs.randn(n + 8).astype(np.float32) for _ in range(4)]
    base[3] = np.abs(base[3])
    for off in (0, 1, 2, 3, 5):
        for ln in (1, 2, 3, 4, 5, 1000, 1001, n - 3):
            dev = [torch.from_numpy(b.copy()).to(DEV) for b in base]
            torch.ops.invpref.adam_dense_(dev[0][off:off + ln], dev[1][off:off + ln], dev[2][off:off + ln],
                                          dev[3][off:off + ln], 3, 0.01, 0.9, 0.999, 1e-8, True)
            host = [b.copy() for b in base]
            O.adam(host[0][off:off + ln], host[1][off:off + ln].copy(), host[2][off:off + ln], host[3][off:off + ln],
                   3, 0.01)
            host[1][off:off + ln] = 0
            for d, h in zip(dev, host):
                np.testing.assert_array_equal(d.cpu().numpy(), h)    # bit exact, nothing outside the range touched
    # buffers misaligned differently from one another: the scalar form
    dev = [torch.from_numpy(b.copy()).to(DEV) for b in base]
    offs = (0, 1, 2, 3)
    torch.ops.invpref.adam_dense_(*[d[o:o + 1000] for d, o in zip(dev, offs)], 3, 0.01, 0.9, 0.999, 1e-8, False)
    host = [b.copy() for b in base]
    O.adam(host[0][0:1000], host[1][1:1001].copy(), host[2][2:1002], host[3][3:1003], 3, 0.01)
    for d, h in zip(dev, host):
        np.testing.assert_array_equal(d.cpu().numpy(), h)


@pytest.mark.parametrize('implicit', [True, False])
def test_estep_assign_functional_and_inplace(implicit):
    data, tabs, envs, w, P, t = _setup(implicit, seed=5)
    new, counts, diff = torch.ops.invpref.estep_assign(P, t['u'], t['v'], t['y'], t['e'], implicit, None, _ws())
    on, oc, od, _ = O.estep(O.Tables(tabs), data[:, 0], data[:, 1], data[:, 2], implicit, old_envs=envs)
    np.testing.assert_array_equal(new.cpu().numpy(), on)
    np.testing.assert_array_equal(counts.cpu().numpy(), oc)
    assert int(diff.item()) == od
    np.testing.assert_array_equal(t['e'].cpu().numpy(), envs)          # functional form: input untouched
    e2 = t['e'].clone()
    c2, d2, cw, sw = torch.ops.invpref.estep_assign_(P, t['u'], t['v'], t['y'], e2, implicit, None, True, _ws())
    np.testing.assert_array_equal(e2.cpu().numpy(), on)
    np.testing.assert_array_equal(c2.cpu().numpy(), oc)
    assert int(d2.item()) == od
    _, ocw, osw = O.stat_envs(on, E)
    np.testing.assert_array_equal(cw.cpu().numpy(), ocw)
    np.testing.assert_array_equal(sw.cpu().numpy(), osw)
    cs, cws, sws = torch.ops.invpref.stat_envs(e2, E, True, _ws())
    np.testing.assert_array_equal(cs.cpu().numpy(), oc)
    np.testing.assert_array_equal(sws.cpu().numpy(), osw)


def test_planned_ops_vs_oracle():
    """train_step_planned_grad_ / train_step_planned_adam_ (the path bench.py times) through torch.ops"""
    data, tabs, envs, w, P, t = _setup(True, seed=9)
    dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D), DEV)
    flags = ops.flags_of(True, False, True, True, False)
    og, ol = O.mstep(O.Tables(tabs), data[:, 0], data[:, 1], envs, data[:, 2], w, COEFS,
                     O.flags_of(True, False, True, True, False))
    G = [torch.full_like(p, 7.0) for p in P]      # every row is overwritten
    losses = torch.zeros(6, device=DEV)
    ws = _ws(zero=True)
    torch.ops.invpref.train_step_planned_grad_(P, G, dp.buf, dp.meta, t['e'], t['y'], t['w'], B, COEFS, flags, losses,
                                               None, None, 0, ws)
    np.testing.assert_allclose(losses.cpu().numpy(), ol, rtol=1e-5)
    for k, g, o in zip(ops.PARAM_NAMES, G, og):
        assert _relerr(g.cpu().numpy(), o) < 2e-5, k
    P2 = [torch.zeros_like(p) for p in P]
    M = [torch.zeros_like(p) for p in P]
    V = [torch.zeros_like(p) for p in P]
    l2 = torch.zeros(6, device=DEV)
    torch.ops.invpref.train_step_planned_adam_(P, P2, M, V, dp.buf, dp.meta, t['e'], t['y'], t['w'], B, COEFS, flags, l2,
                                               1, 0.005, 0.9, 0.999, 1e-8, None, None, 0, ws)
    np.testing.assert_allclose(l2.cpu().numpy(), ol, rtol=1e-5)
    for k, p0, g, p2, m in zip(ops.PARAM_NAMES, P, og, P2, M):
        po = p0.cpu().numpy().reshape(-1).copy()
        mo, vo = np.zeros_like(po), np.zeros_like(po)
        O.adam(po, np.ascontiguousarray(g, np.float32).reshape(-1), mo, vo, 1, 0.005)
        # first Adam step moves every parameter by ~lr*sign(g): compare where the gradient is not noise-level
        big = np.abs(np.asarray(g).reshape(-1)) > 1e-4 * np.abs(g).max()
        assert np.abs(p2.cpu().numpy().reshape(-1) - po)[big].max() < 0.02 * 0.005, k
        assert _relerr(m.cpu().numpy().reshape(-1), mo) < 2e-5, k


def test_opcheck():
    """schema (aliasing / mutation annotations), fake-tensor implementation, autograd registration, AOT dispatch"""
    data, tabs, envs, w, P, t = _setup(True, seed=11)
    flags = ops.flags_of(True, True, True, False, True)
    oc = torch.library.opcheck
    G = [torch.zeros_like(p) for p in P]
    oc(torch.ops.invpref.train_step_fused.default,
       (P, G, t['u'], t['v'], t['e'], t['y'], t['w'], B, COEFS, flags, torch.zeros(6, device=DEV), _ws()))
    oc(torch.ops.invpref.estep_assign.default, (P, t['u'], t['v'], t['y'], t['e'], True, None, _ws()))
    oc(torch.ops.invpref.estep_assign_.default, (P, t['u'], t['v'], t['y'], t['e'].clone(), True, None, True, _ws()))
    n = 1024
    oc(torch.ops.invpref.adam_dense_.default,
       (torch.randn(n, device=DEV), torch.randn(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV),
        1, 0.01, 0.9, 0.999, 1e-8, True))
    oc(torch.ops.invpref.adam_ranges_.default,
       (torch.randn(n, device=DEV), torch.randn(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV),
        [0, 512], [256, 128], 1, 0.01, 0.9, 0.999, 1e-8, False, None, None, 0))
    oc(torch.ops.invpref.forward.default, (P, t['u'], t['v'], t['e'], True))
    oc(torch.ops.invpref.stat_envs.default, (t['e'], E, True, _ws()))
    oc(torch.ops.invpref.predict.default, (P[0], P[1], t['u'][:16], True))
    dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D), DEV)
    oc(torch.ops.invpref.train_step_planned_grad_.default,
       (P, G, dp.buf, dp.meta, t['e'], t['y'], t['w'], B, COEFS, flags, torch.zeros(6, device=DEV), None, None, 0,
        _ws(zero=True)))
    oc(torch.ops.invpref.train_step_planned_adam_.default,
       (P, [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P],
        dp.buf, dp.meta, t['e'], t['y'], t['w'], B, COEFS, flags, torch.zeros(6, device=DEV), 1, 0.005, 0.9, 0.999,
        1e-8, None, None, 0, _ws(zero=True)))


def test_ops_capture_in_hip_graph():
    """current-stream enqueue, no host sync: the operators record into a HIP graph and replay"""
    data, tabs, envs, w, P, t = _setup(True, seed=13)
    flags = ops.flags_of(True, True, True, False, True)
    G = [torch.zeros_like(p) for p in P]
    losses = torch.zeros(6, device=DEV)
    ws = _ws()
    args = (P, G, t['u'], t['v'], t['e'], t['y'], t['w'], B, COEFS, flags, losses, ws)
    torch.ops.invpref.train_step_fused(*args)
    torch.cuda.synchronize()
    ref = [g.clone() for g in G]
    ref_l = losses.clone()
    for g in G:
        g.zero_()
    losses.zero_()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            torch.ops.invpref.train_step_fused(*args)
        gr.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses.cpu().numpy(), ref_l.cpu().numpy(), rtol=1e-6)
    for a, b in zip(G, ref):
        assert _relerr(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
