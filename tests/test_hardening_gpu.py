"""GPU: round-2 hardening -- NaN distances in the E-step, the env-classifier's own methods, top-k over item
counts beyond the LDS-staged form, mixing planned epochs with plan-free train_a_batch calls."""
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import ops, synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager, LOSS_KEYS
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _dev(tabs):
    return [torch.from_numpy(np.ascontiguousarray(tabs[k], np.float32)).to(DEV) for k in ops.PARAM_NAMES]


class StubEvaluator:
    def evaluate(self):
        return {}


@pytest.mark.parametrize('implicit', [True, False])
def test_estep_nan_distances_follow_torch_argmin(implicit):
    """torch.argmin (train.py:199) returns the first NaN if a row of distances holds one; the kernel and the oracle
    do the same and never index outside [0, E) (round 1: index 99 when every distance was NaN)."""
    U, I, E, D, B = 40, 30, 5, 64, 512
    rs = np.random.RandomState(3)
    tabs = synth.tables(4, U, I, E, D, std=0.3)
    u, v = rs.randint(0, U, B), rs.randint(0, I, B)
    y = (rs.randint(0, 2, B) if implicit else rs.randint(1, 6, B)).astype(np.float32)
    old = rs.randint(0, E, B)
    # NaN in ONE env row -> that env's distance is NaN for every interaction; NaN in a user's env-aware row -> all
    # of that user's distances are NaN; NaN in an invariant item row -> all distances of its interactions
    tabs['embed_env.weight'][3, 5] = np.nan
    t2 = {k: a.copy() for k, a in tabs.items()}
    t2['embed_env.weight'][3, 5] = 0.1
    t2['embed_user_env_aware.weight'][7, :] = np.nan
    t2['embed_item_invariant.weight'][2, 0] = np.nan
    ws = ops.Workspace(DEV)
    for tb, expect_all in ((tabs, 3), (t2, None)):
        P = _dev(tb)
        to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(DEV)  # noqa: E731
        new, counts, diff, _, _ = ops.estep(P, to(u, np.int64), to(v, np.int64), to(y, np.float32), implicit,
                                            to(old, np.int64), ws)
        on, oc, od, dist = O.estep(O.Tables(tb), u, v, y, implicit, old_envs=old, want_dist=True)
        got = new.cpu().numpy()
        assert got.min() >= 0 and got.max() < E
        np.testing.assert_array_equal(got, on)
        np.testing.assert_array_equal(counts.cpu().numpy(), oc)
        assert int(diff.item()) == od and int(counts.sum().item()) == B
        # the specification itself: torch.argmin on the oracle's distance rows
        np.testing.assert_array_equal(on, torch.argmin(torch.from_numpy(dist), dim=1).numpy())
        if implicit:
            continue   # (the BCE clamps turn a NaN score into the distance 100: nothing NaN reaches the argmin)
        if expect_all is not None:
            assert (got == expect_all).all()
        else:
            assert (got[u == 7] == 0).all() and (got[v == 2] == 0).all()


def test_env_classifier_methods():
    """LinearLogSoftMaxEnvClassifier.forward / get_L1_reg / get_L2_reg (models.py:206-217) on their own, values and
    gradients, against the same formulas in torch on the CPU (fp32: 1e-6 / 1e-5)."""
    E, D, B = 6, 48, 37
    model = InvPrefImplicit(20, 10, E, D).to(DEV)
    rs = np.random.RandomState(1)
    x0 = rs.randn(B, D).astype(np.float32)
    x = torch.from_numpy(x0).to(DEV).requires_grad_(True)
    up = torch.from_numpy(rs.randn(B, E).astype(np.float32))
    cls = model.env_classifier
    out = cls(x)
    (out * up.to(DEV)).sum().backward()
    W = cls.linear_map.weight.detach().cpu().clone().requires_grad_(True)
    b = cls.linear_map.bias.detach().cpu().clone().requires_grad_(True)
    xc = torch.from_numpy(x0).requires_grad_(True)
    ref = torch.log_softmax(xc @ W.t() + b, dim=1)
    (ref * up).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(cls.linear_map.weight.grad.cpu().numpy(), W.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cls.linear_map.bias.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5)
    assert cls(torch.zeros(0, D, device=DEV)).shape == (0, E)
    for p in (cls.linear_map.weight, cls.linear_map.bias):
        p.grad = None
    l1, l2 = cls.get_L1_reg(), cls.get_L2_reg()
    (2.0 * l1 + 3.0 * l2).backward()      # the upstream scalars are applied on the device (no .item())
    W2 = W.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True)
    r1 = torch.norm(W2, 1) / (D * E) + torch.norm(b2, 1) / E
    r2 = torch.norm(W2, 2).pow(2) / (D * E) + torch.norm(b2, 2).pow(2) / E
    (2.0 * r1 + 3.0 * r2).backward()
    np.testing.assert_allclose(float(l1), float(r1), rtol=1e-5)
    np.testing.assert_allclose(float(l2), float(r2), rtol=1e-5)
    np.testing.assert_allclose(cls.linear_map.weight.grad.cpu().numpy(), W2.grad.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(cls.linear_map.bias.grad.cpu().numpy(), b2.grad.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('n_items', [10240, 10241, 51283])
def test_topk_large_item_counts(n_items):
    """evaluate.py:88-120 beyond the LDS-staged form (MIND: 51 283 items): ids exact against numpy."""
    import ctypes as C
    from invpref_kdd_2022_amd._capi import check, lib, ptr, stream_ptr
    rs = np.random.RandomState(n_items)
    n, k = 9, 7
    ratings = rs.rand(n, n_items).astype(np.float32)
    ratings[:, ::97] = 0.5                                   # ties: lowest id first
    mask = [np.sort(rs.choice(n_items, rs.randint(0, 400), replace=False)) for _ in range(n)]
    pool = [np.sort(rs.choice(n_items, rs.randint(1, 300), replace=False)) for _ in range(n)]
    truth = [np.sort(rs.choice(n_items, rs.randint(1, 50), replace=False)) for _ in range(n)]

    def csr(lists):
        p = np.zeros(len(lists) + 1, np.int32)
        p[1:] = np.cumsum([len(a) for a in lists])
        return torch.from_numpy(p).to(DEV), torch.from_numpy(np.concatenate(lists).astype(np.int32)).to(DEV)
    (mp, mi), (hp, hi), (tp, ti) = csr(mask), csr(pool), csr(truth)
    r = torch.from_numpy(ratings).to(DEV)
    for use_pool in (False, True):
        items = torch.empty(n, k, dtype=torch.int32, device=DEV)
        hits = torch.empty(n, k, dtype=torch.float32, device=DEV)
        check(lib().invpref_eval_topk_hip(ptr(r), n, n_items, ptr(mp), ptr(mi), ptr(hp) if use_pool else None,
                                          ptr(hi) if use_pool else None, ptr(tp), ptr(ti), k, ptr(items), ptr(hits),
                                          stream_ptr()), 'invpref_eval_topk_hip')
        for j in range(n):
            row = ratings[j].copy()
            row[mask[j]] = -1024.0
            if use_pool:
                row[pool[j]] += 1024.0
            order = np.argsort(-row, kind='stable')[:k]
            np.testing.assert_array_equal(items[j].cpu().numpy(), order)
            np.testing.assert_array_equal(hits[j].cpu().numpy(), np.isin(order, truth[j]).astype(np.float32))
    np.testing.assert_array_equal(r.cpu().numpy(), ratings)   # the rating matrix is not modified


@pytest.mark.parametrize('D', [64, 256])
def test_train_a_batch_after_planned_epochs(monkeypatch, D):
    """The planned gradient pass leaves its gradient in the flat buffer (its Adam does not zero it); a plan-free
    train_a_batch() right after must not add to it.  Against the oracle trainer: one epoch +
    one extra minibatch."""
    monkeypatch.setenv('INVPREF_FORCE_SHARDED_PATH', '1')
    monkeypatch.setenv('INVPREF_SHARD', 'rows')
    U, I, E, n, bs = 200, 150, 4, 3000, 1024
    data = synth.interactions(21, U, I, n, implicit=True)
    tabs = synth.tables(22, U, I, E, D, std=0.05)
    cf = dict(invariant_coe=3.35, env_aware_coe=9.99, env_coe=9.06, L2_coe=3.13, L1_coe=0.49, alpha=1.9)
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=False, reg_env_embed=True)
    model.load_state_dict({k: torch.from_numpy(tabs[k]) for k in O.PARAM_NAMES})
    np.random.seed(5)
    mgr = ImplicitTrainManager(model=model, evaluator=StubEvaluator(), device=DEV,
                               training_data=torch.from_numpy(data).to(DEV), batch_size=bs, epochs=4, cluster_interval=2,
                               evaluate_interval=10 ** 9, lr=0.005, use_class_re_weight=True,
                               use_recommend_re_weight=True, cluster_use_random_sort=False, **cf)
    env0 = mgr.envs.cpu().numpy().copy()
    mgr.stat_envs()
    ep = mgr.train_a_epoch()
    got = mgr.train_a_batch(mgr.users_tensor[:bs], mgr.items_tensor[:bs], mgr.scores_tensor[:bs], mgr.envs[:bs],
                            mgr.sample_weights[:bs], mgr.alpha)
    tr = O.Trainer(tabs, data, env0, implicit=True, batch_size=bs, coefs=[cf[k] for k in (
        'invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')], lr=0.005, reweight_rec=True,
        reweight_cls=True, reg_only_embed=False, reg_env_embed=True)
    tr.stat_envs()
    oep = tr.train_a_epoch()
    ob = tr.train_a_batch(0, bs)
    np.testing.assert_allclose([ep[k] for k in LOSS_KEYS], oep, rtol=2e-5)
    np.testing.assert_allclose([got[k] for k in LOSS_KEYS], ob, rtol=2e-5)
