"""GPU: deferred dense Adam on untouched user rows (include/invpref_hip.h: invpref_mstep_rows_adam_deferred_hip,
invpref_flush_deferred_hip) is the dense optimiser of the reference (torch.optim.Adam, train.py:41, :155-157) BIT FOR BIT:
N deferred steps + flush leave exactly the parameters, moments and loss terms N dense steps leave."""
import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
YAHOO = dict(lr=0.005, invariant_coe=3.351991776096847, env_aware_coe=9.988658447411407,
             env_coe=9.06447753571379, L2_coe=3.1351402017943117, L1_coe=0.4935216278026648)


class StubEvaluator:
    def evaluate(self):
        return {}


def _manager(monkeypatch, defer: bool, U, I, E, D, N, B, alpha, seed=11):
    monkeypatch.setenv('INVPREF_DEFER', '1' if defer else '0')
    monkeypatch.setenv('INVPREF_ALT', '0')   # (the dense TWO-LAUNCH form is what the deferred form must equal bit for bit)
    data = synth.interactions(seed, U, I, N, implicit=True, zipf=True)
    tabs = synth.tables(seed + 7, U, I, E, D)
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=True, reg_env_embed=False)
    model.load_state_dict({k: torch.from_numpy(tabs[k]) for k in O.PARAM_NAMES})
    np.random.seed(seed)
    mgr = ImplicitTrainManager(model=model, evaluator=StubEvaluator(), device=DEV,
                               training_data=torch.from_numpy(data).to(DEV), batch_size=B, epochs=10 ** 9,
                               cluster_interval=5, evaluate_interval=10 ** 9, use_class_re_weight=True,
                               use_recommend_re_weight=False, cluster_use_random_sort=False, alpha=alpha,
                               rank=0, world_size=1, **YAHOO)
    mgr.stat_envs()
    return mgr


def _run(mgr):
    """one eager epoch, then replayed runs with an E-step in between: 1 + 5 + 4 + 2 epochs"""
    out = [mgr.train_epochs(1, sync=False)]
    out.append(mgr.train_epochs(5, sync=False))
    diff = mgr.cluster(sync=False)
    cnt = mgr.stat_envs(sync=False)
    out.append(mgr.train_epochs(4, sync=False))
    out.append(mgr.train_epochs(2, sync=False))
    torch.cuda.synchronize()
    st = mgr.state
    return (torch.cat(out).cpu().numpy(), st.param.cpu().numpy().copy(), st.exp_avg.cpu().numpy().copy(),
            st.exp_avg_sq.cpu().numpy().copy(), int(diff.item()), cnt.cpu().numpy(), mgr.envs.cpu().numpy().copy())


@pytest.mark.parametrize('alpha', [1.9053711444718746, None], ids=['fixed_alpha', 'alpha_schedule'])
def test_deferred_steps_plus_flush_equal_dense_steps_bitwise(monkeypatch, alpha):
    # 6 000 users, 2 048 interactions per minibatch: about three quarters of the user rows are untouched by a step, some for
    # a whole run; 20 minibatches x 12 epochs = 240 optimiser steps with an E-step after the sixth epoch
    shape = dict(U=6000, I=400, E=4, D=64, N=40000, B=2048)
    a = _manager(monkeypatch, True, alpha=alpha, **shape)
    got = _run(a)
    assert a._defer and a.graphs_enabled() and a.state.step == 240
    assert sum(int(x) for dp in a._plans for x in dp.struct.defer_tail) > 1000
    b = _manager(monkeypatch, False, alpha=alpha, **shape)
    want = _run(b)
    assert not b._defer and b.state.step == 240
    for x, y, name in zip(got, want, ('epoch losses', 'parameters', 'exp_avg', 'exp_avg_sq', 'diff_num', 'env counts',
                                      'envs')):
        np.testing.assert_array_equal(x, y, err_msg=name)
    assert np.isfinite(got[0]).all() and got[0][:, 5].min() > 0
    # every row stamp is at the run's last step: nothing is left deferred outside a replayed run
    assert int(a._last_step.min()) == int(a._last_step.max()) == 240
    # the module surface sees the flushed parameters (state_dict aliases the current flat buffer)
    np.testing.assert_array_equal(a.model.state_dict()['embed_user_invariant.weight'].cpu().numpy(),
                                  b.model.state_dict()['embed_user_invariant.weight'].cpu().numpy())


def test_deferred_step_through_the_c_abi_against_dense(monkeypatch):
    """the C-ABI sequence itself: k schedule-driven deferred steps on in-place user tables + one flush == the same k
    schedule-driven dense steps on ping-pong buffers (parameters, moments, loss sums)."""
    import ctypes as C
    from invpref_kdd_2022_amd import _capi
    U, I, E, D, B, K = 3000, 300, 4, 64, 1024, 9
    L = _capi.lib()
    data = synth.interactions(3, U, I, K * B, implicit=True, zipf=True)
    tabs = synth.tables(4, U, I, E, D)
    rs = np.random.RandomState(5)
    e = torch.from_numpy(rs.randint(0, E, K * B).astype(np.int64)).to(DEV)
    y = torch.from_numpy(data[:, 2].astype(np.float32)).to(DEV)
    w = torch.rand(K * B, device=DEV)
    plans = [planlib.upload(planlib.build_row_plan(data[k * B:(k + 1) * B, 0], data[k * B:(k + 1) * B, 1],
                                                   data[k * B:(k + 1) * B, 2], U, I, factor_num=D, env_num=E, push=True), DEV)
             for k in range(K)]
    coefs = (3.35, 9.99, 9.06, 3.13, 0.49, 1.9)
    flags = ops.flags_of(True, False, True, True, False)
    host = np.zeros((64, 8), np.float32)
    _capi.check(L.invpref_adam_schedule_fill(host.ctypes.data, 1, 64, 0.005, 0.9, 0.999, 1e-8), 'fill')

    def run(defer):
        P = [torch.from_numpy(tabs[k]).to(DEV) for k in ops.PARAM_NAMES]
        P2 = [torch.zeros_like(p) for p in P]
        M = [torch.zeros_like(p) for p in P]
        V = [torch.zeros_like(p) for p in P]
        table = torch.from_numpy(host).to(DEV)
        st32 = np.zeros(32, np.int32)
        st32[16:18] = 1, 1                               # slot 1 = step 1, base 1
        st32[18:26] = host[0].view(np.int32)
        state = torch.from_numpy(st32).to(DEV)
        last = torch.zeros(U, dtype=torch.int32, device=DEV)
        losses = torch.zeros(K, 6, device=DEV)
        ws = ops.Workspace(DEV)
        a, b = P, P2
        for k in range(K):
            sl = slice(k * B, (k + 1) * B)
            pa, pb = list(a), list(b)
            if defer:
                pa[0] = pb[0] = P[0]; pa[2] = pb[2] = P[2]
            ops.mstep_rows_adam(pa, pb, M, V, plans[k], e[sl], y[sl], w[sl], B, coefs, flags, losses[k], k + 1, 0.005, ws,
                                sched=(state, table, (k + 1) & 1), last_step=last if defer else None)
            a, b = b, a
        if defer:
            home = list(a); home[0], home[2] = P[0], P[2]
            ops.flush_deferred(home, a, M, V, last, (state, table, (K + 1) & 1))
            assert int(last.min()) == K
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in list(a) + M + V + [losses]]

    for x, z in zip(run(True), run(False)):
        np.testing.assert_array_equal(x, z)
