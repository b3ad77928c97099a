"""GPU: round 6 -- (1) cluster() + stat_envs() as ONE launch (invpref_estep_fused_hip: the assignment kernel's epilogue folds
counts, diff_num and class weights; train.py:235-259, :268-280) against the oracle and against the two-launch entry points,
bit for bit, over repeated calls (ticket reset, ring rows) and under the reference's default tie-break with the ready-made
permutation table; (2) INVPREF_WEIGHTS_BY_ENV: the planned M-step forming sample_weights[i] = class_weights[envs[i]]
(train.py:278) itself must give EXACTLY the results of the same step fed the gathered array -- every kernel family: the
alternating form, the two-launch small instance, the wide rows (16 / 32 lanes, MFMA classifier) and the element-wise
instances; (3) the managers' lazy `sample_weights` attribute and the semantics of cluster() without stat_envs()."""
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import _capi, ops, plan as planlib, synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager, _unrank_permutations
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _dev(tabs):
    return [torch.from_numpy(np.ascontiguousarray(tabs[k], np.float32)).to(DEV) for k in ops.PARAM_NAMES]


def _t(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(DEV)


@pytest.mark.parametrize('U,I,E,D,N,implicit', [
    (15400, 1000, 4, 64, 250154, True),      # the Yahoo shape: 2 048 workgroups take tickets
    (15400, 1000, 4, 64, 250154, False),
    (300, 70, 2, 40, 5000, True),
    (50, 20, 7, 30, 777, True),              # E = 7: 5 040 permutation rows
    (40, 30, 16, 256, 3000, False),          # no table form beyond seven environments
    (3, 2, 1, 4, 5, True),                   # one workgroup: it is the last one
])
@pytest.mark.parametrize('random_sort', [False, True])
def test_fused_estep_equals_oracle_and_two_launch_form(U, I, E, D, N, implicit, random_sort):
    seed = 40 + U + E
    data = synth.interactions(seed, U, I, N, implicit=implicit)
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.3 if implicit else 0.15)
    P = _dev(tabs)
    u, v, y = _t(data[:, 0], np.int64), _t(data[:, 1], np.int64), _t(data[:, 2], np.float32)
    rs = np.random.RandomState(seed)
    old = rs.randint(0, E, N).astype(np.int64)
    ws, es = ops.Workspace(DEV), ops.EstepState(E, DEV, ring_cap=4)
    assert (es.perm_table is not None) == (E <= 7)
    eps_base = np.array([1e-10 * (1e-1 ** i) for i in range(E)], np.float32)
    tab = O.Tables(tabs)
    envs = _t(old, np.int64)
    cur = old.copy()
    for call in range(6):                     # repeated calls: ticket back at zero, ring rows in turn (capacity 4: wraps)
        perm, rows = None, None
        if random_sort:
            import math
            idx = rs.randint(0, math.factorial(E), N)
            dt = np.uint8 if E <= 5 else (np.int32 if E <= 12 else np.int64)
            perm = _t(idx.astype(dt), dt)
            rows = _unrank_permutations(idx, eps_base)
        counts = torch.zeros(E, dtype=torch.int64, device=DEV)
        diff = torch.zeros(1, dtype=torch.int64, device=DEV)
        cw = torch.zeros(E, dtype=torch.float32, device=DEV)
        ops.estep_fused(P, u, v, y, implicit, envs, es, ws, perm_index=perm, eps_base=eps_base.tolist() if random_sort else None,
                        counts=counts, diff=diff, class_weights=cw)
        row = es.next_row()
        on, oc, od, _ = O.estep(tab, data[:, 0], data[:, 1], data[:, 2], implicit, old_envs=cur, eps_rows=rows)
        np.testing.assert_array_equal(envs.cpu().numpy(), on)
        np.testing.assert_array_equal(counts.cpu().numpy(), oc)
        assert int(diff.item()) == od
        _, ocw, _ = O.stat_envs(on, E)
        np.testing.assert_array_equal(cw.cpu().numpy(), ocw)
        ring = es.ring[row].cpu().numpy()
        np.testing.assert_array_equal(ring[:E], oc)
        assert int(ring[E]) == od and row == call % 4
        st = es.state.cpu().numpy()
        assert st[0] == 0 and not st[32::32].any() and st[1] == call + 1    # every ticket back at zero; E-steps counted
        # the two-launch entry points on the same input: the same assignments
        new2, c2, d2, cw2, _ = ops.estep(P, u, v, y, implicit, _t(cur, np.int64), ws, perm_index=perm,
                                         eps_base=eps_base.tolist() if random_sort else None)
        np.testing.assert_array_equal(new2.cpu().numpy(), on)
        np.testing.assert_array_equal(cw2.cpu().numpy(), ocw)
        cur = on
        # move the tables a little so that the next call has something to reassign
        for p in P[:4]:
            p.mul_(1.0 + 0.05 * (call + 1))
        tab = O.Tables({k: p.cpu().numpy() for k, p in zip(ops.PARAM_NAMES, P)})


def test_fused_estep_replayed_from_a_graph():
    # the way the managers run it: captured once, replayed; the ring keeps the replays' results apart without a copy
    U, I, E, D, N = 2000, 300, 4, 64, 60000
    data = synth.interactions(3, U, I, N, implicit=True)
    P = _dev(synth.tables(4, U, I, E, D, std=0.3))
    u, v, y = _t(data[:, 0], np.int64), _t(data[:, 1], np.int64), _t(data[:, 2], np.float32)
    envs = torch.zeros(N, dtype=torch.int64, device=DEV)
    ws, es = ops.Workspace(DEV), ops.EstepState(E, DEV)
    cw = torch.zeros(E, device=DEV)
    ops.estep_fused(P, u, v, y, True, envs, es, ws, class_weights=cw)   # (sizes the workspace)
    es.next_row()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.estep_fused(P, u, v, y, True, envs, es, ws, class_weights=cw)
    tab = None
    for rep in range(5):
        for p in P[:4]:
            p.mul_(1.03)
        before = envs.cpu().numpy().copy()
        g.replay()
        row = es.next_row()
        tab = O.Tables({k: p.cpu().numpy() for k, p in zip(ops.PARAM_NAMES, P)})
        on, oc, od, _ = O.estep(tab, data[:, 0], data[:, 1], data[:, 2], True, old_envs=before)
        np.testing.assert_array_equal(envs.cpu().numpy(), on)
        ring = es.ring[row].cpu().numpy()
        np.testing.assert_array_equal(ring[:E], oc)
        assert int(ring[E]) == od
    assert int(es.state[0].item()) == 0 and int(es.state[1].item()) == es.issued == 6


def test_perm_table_fill_is_itertools_order():
    import ctypes as C
    import itertools
    L = _capi.lib()
    for E in range(1, 8):
        rows = len(list(itertools.permutations(range(E))))
        host = np.zeros(rows, np.uint32)
        assert L.invpref_perm_table_fill(E, host.ctypes.data) == rows
        want = np.array([sum(p[pos] << (4 * pos) for pos in range(E)) for p in itertools.permutations(range(E))], np.uint32)
        np.testing.assert_array_equal(host, want)
    assert L.invpref_perm_table_fill(8, np.zeros(1, np.uint32).ctypes.data) == -1
    assert L.invpref_perm_table_fill(3, None) == -1


# ---------------------------------------------------------------------------------------------- weights by environment
def _mstep_pair(U, I, E, D, B, implicit, fl, alt=False, seed=7):
    """the same planned step twice: fed the gathered sample weights, and fed the E class weights under WEIGHTS_BY_ENV"""
    data = synth.interactions(seed, U, I, B, implicit=implicit, zipf=True)
    tabs = synth.tables(seed + 1, U, I, E, D, std=0.1)
    rs = np.random.RandomState(seed)
    envs = rs.randint(0, E, B).astype(np.int64)
    _, cwv, swv = O.stat_envs(envs, E)
    e, y = _t(envs, np.int64), _t(data[:, 2], np.float32)
    cw, sw = _t(cwv, np.float32), _t(swv, np.float32)
    coefs = (2.05, 8.63, 5.1, 7.73, 0.0015, 1.74)
    flags = ops.flags_of(implicit, *fl)
    outs = []
    for wts, fg in ((sw, flags), (cw, flags | _capi.WEIGHTS_BY_ENV)):
        P = _dev(tabs)
        M = [torch.full_like(p, 1e-3) for p in P]
        V = [torch.full_like(p, 1e-5) for p in P]
        losses = torch.zeros(6, device=DEV)
        if alt:
            cur = (data[:, 0], data[:, 1], data[:, 2].astype(np.float32))
            p0 = planlib.build_alt_plan(cur, None, 0, U, I, factor_num=D)
            p1 = planlib.build_alt_plan(None, cur[:2], 1, U, I, factor_num=D, n_partials_prev=p0['n_tasks'])
            aws = ops.AltWorkspace(P, B, p0['n_tasks'] + 1)
            ops.mstep_alt(P, M, V, planlib.upload_alt(p0, DEV), e, wts, B, B, coefs, fg, None, 3, 0.01, aws, 0)
            ops.mstep_alt(P, M, V, planlib.upload_alt(p1, DEV), None, None, B, B, coefs, fg, losses, 3, 0.01, aws, 1)
            assert aws.error() == 0
            res = P
        else:
            P2 = [torch.zeros_like(p) for p in P]
            dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D, env_num=E), DEV)
            ops.mstep_rows_adam(P, P2, M, V, dp, e, y, wts, B, coefs, fg, losses, 3, 0.01, ops.Workspace(DEV))
            res = P2
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in res + M + V] + [losses.cpu().numpy()])
    return outs


@pytest.mark.parametrize('U,I,E,D,B,alt', [
    (3000, 400, 4, 64, 8192, True),        # the alternating form (Yahoo's kernels)
    (3000, 400, 4, 64, 8192, False),       # the two-launch small instance, full rows
    (300, 60, 3, 40, 3000, True),          # vector loads, guarded
    (300, 60, 3, 40, 3000, False),
    (120, 30, 2, 30, 1000, True),          # element-wise instances
    (120, 30, 2, 30, 1000, False),
    (2000, 700, 8, 128, 20000, False),     # MovieLens' instance: 16 lanes x 2, E = 8 (compile-time BYENV)
    (2000, 700, 16, 128, 20000, False),
    (1500, 900, 6, 64, 9000, False),       # 16 lanes x 1, E > 4
    (1500, 900, 16, 256, 12000, False),    # MIND's instance: 32 lanes, the MFMA classifier
    (900, 300, 8, 256, 6000, False),
    (400, 200, 16, 100, 4000, False),      # wide, not a full row: run-time flag
    (400, 200, 5, 200, 4000, False),
])
@pytest.mark.parametrize('fl', [(True, True, False, True), (False, True, True, False), (True, False, False, False)])
def test_weights_by_environment_equal_gathered_weights(U, I, E, D, B, alt, fl):
    a, b = _mstep_pair(U, I, E, D, B, True, fl, alt=alt)
    for x, z in zip(a, b):
        np.testing.assert_array_equal(x, z)     # the same float enters the same arithmetic: bit for bit


def test_wide_mm_forms_by_environment(monkeypatch):
    # both forms of launch 1 for full wide rows (per-interaction classifier / MFMA classifier), forced either way
    for mm in ('0', '1'):
        monkeypatch.setenv('INVPREF_WIDE_MM', mm)
        for (E, D) in ((8, 128), (16, 256)):
            a, b = _mstep_pair(800, 300, E, D, 7000, False, (True, True, False, True))
            for x, z in zip(a, b):
                np.testing.assert_array_equal(x, z)


def test_by_environment_needs_the_class_weights_and_not_pure_mf():
    U, I, D, B = 50, 20, 64, 300
    data = synth.interactions(1, U, I, B, implicit=True)
    P = [torch.zeros(U, D, device=DEV), torch.zeros(I, D, device=DEV)]
    dp = planlib.upload(planlib.build_row_plan(data[:, 0], data[:, 1], data[:, 2], U, I, factor_num=D, env_num=0), DEV)
    with pytest.raises(_capi.InvPrefError):
        ops.mstep_rows_adam(P, [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P],
                            dp, None, _t(data[:, 2], np.float32), None, B, (1., 0., 0., 0.1, 0.1, 0.),
                            ops.flags_of(True, False, False, True, False, dense_reg=False) | _capi.WEIGHTS_BY_ENV,
                            torch.zeros(6, device=DEV), 1, 0.01, ops.Workspace(DEV), pure=True)


# ---------------------------------------------------------------------------------------------- the managers
class _Stub:
    def evaluate(self):
        return {'stub': 0.0}


def _mgr(seed=11, E=4, D=64, n=30000, bs=4096, **kw):
    U, I = 1500, 200
    data = synth.interactions(seed, U, I, n, implicit=True)
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = InvPrefImplicit(U, I, E, D)
    args = dict(model=model, evaluator=_Stub(), device=DEV, training_data=torch.from_numpy(data).to(DEV), batch_size=bs,
                epochs=6, cluster_interval=2, evaluate_interval=10 ** 9, lr=0.01, invariant_coe=2., env_aware_coe=8.,
                env_coe=5., L2_coe=1.0, L1_coe=0.01, alpha=1.5, use_class_re_weight=True, use_recommend_re_weight=True)
    args.update(kw)
    return ImplicitTrainManager(**args), data


def test_manager_lazy_sample_weights_and_by_env_epochs(monkeypatch):
    """train(): the E-steps run fused, the epochs take class_weights[env]; the run must equal -- bit for bit: same kernels, same
    floats -- the run of a manager that gathers the N-length array and reads it per interaction (INVPREF_WEIGHTS_BY_ENV=0,
    INVPREF_ESTEP_FUSED=0), and the lazy `sample_weights` attribute must be class_weights[envs]."""
    res = []
    for lazy in ('1', '0'):
        monkeypatch.setenv('INVPREF_WEIGHTS_BY_ENV', lazy)
        monkeypatch.setenv('INVPREF_ESTEP_FUSED', lazy)
        mgr, _ = _mgr()
        (losses, _), _, (diffs, cnts, ceps) = mgr.train(silent=True, auto=True)
        assert mgr._fused_estep_ok() == (lazy == '1')
        if lazy == '1':
            assert mgr._sw_lazy and mgr._by_env and mgr._es.issued == 3
        sw = mgr.sample_weights                      # materialises
        assert not mgr._sw_lazy and not mgr._by_env  # handed out: by position until the next stat_envs()
        np.testing.assert_array_equal(sw.cpu().numpy(), mgr.class_weights.cpu().numpy()[mgr.envs.cpu().numpy()])
        res.append((np.array([[d[k] for k in d] for d in losses]), diffs, cnts, ceps, mgr.envs.cpu().numpy(),
                    {k: v.detach().cpu().numpy() for k, v in mgr.model.state_dict().items()}))
    a, b = res
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3] == [2, 4, 6]
    np.testing.assert_array_equal(a[4], b[4])
    for k in a[5]:
        np.testing.assert_array_equal(a[5][k], b[5][k])


def test_cluster_without_stat_envs_keeps_the_weights_by_position():
    """Off the beaten path: cluster() alone (train.py:235-259) moves the environments but NOT sample_weights (train.py:67, :278
    only stat_envs() writes them) -- the epoch that follows must weigh interaction i with the OLD class weight of its OLD
    environment, exactly like a manager on the two-launch E-step path does."""
    outs = []
    for fused in ('1', '0'):
        os.environ['INVPREF_ESTEP_FUSED'] = fused
        os.environ['INVPREF_WEIGHTS_BY_ENV'] = fused
        try:
            mgr, _ = _mgr(seed=5)
            mgr.stat_envs()
            mgr.train_epochs(2)
            old_envs = mgr.envs.cpu().numpy().copy()
            old_cw = mgr.class_weights.cpu().numpy().copy()
            d = mgr.cluster()
            assert d > 0 and not mgr._weights_by_env()
            l1 = mgr.train_epochs(1)                      # weights by position: class_w_old[envs_old]
            np.testing.assert_array_equal(mgr._sample_weights.cpu().numpy(), old_cw[old_envs])
            cnt = mgr.stat_envs()
            assert sum(cnt.values()) == len(old_envs)
            l2 = mgr.train_epochs(1)
            outs.append((d, l1, l2, cnt, mgr.class_weights.cpu().numpy()))
        finally:
            os.environ.pop('INVPREF_ESTEP_FUSED', None)
            os.environ.pop('INVPREF_WEIGHTS_BY_ENV', None)
    assert outs[0][0] == outs[1][0] and outs[0][3] == outs[1][3]
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    np.testing.assert_array_equal(outs[0][4], outs[1][4])


# ---------------------------------------------------------------------------------------------- predict on the matrix cores
@pytest.mark.parametrize('U,I,D,n', [
    (300, 1000, 64, 77),          # fewer users than a workgroup's 64; items not a multiple of 16
    (500, 3706, 128, 2048),       # MovieLens' items, its test batch
    (700, 5003, 256, 256),        # MIND's row length and test batch; a ragged last item tile
    (64, 16, 64, 16), (65, 17, 128, 65), (1000, 51283, 256, 33),
    (300, 1000, 40, 77), (120, 333, 30, 50),     # the reference drivers' factor_num 40 / Coat's 30: the vector-ALU sweep
])
@pytest.mark.parametrize('sigmoid', [True, False])
def test_predict_mfma_is_the_canonical_dot_product_bit_for_bit(U, I, D, n, sigmoid):
    """InvPrefImplicit.predict (models.py:393-407): the whole [n, item_num] score matrix.  Rows of 64 / 128 / 256 floats run
    predict_mm_kernel -- v_mfma_f32_16x16x4_f32, one MFMA per (chunk, slot), slot tiles added in the butterfly's order -- and
    must equal the oracle's forward() on every (user, item) pair BIT FOR BIT, exactly like the vector-ALU kernel
    (INVPREF_PREDICT_MM=0, in a child process) does."""
    tabs = synth.tables(21 + D, U, I, 2, D, std=0.3)
    users = np.random.RandomState(n).randint(0, U, n).astype(np.int64)
    got = ops.predict(_t(tabs[ops.PARAM_NAMES[0]], np.float32), _t(tabs[ops.PARAM_NAMES[1]], np.float32), _t(users, np.int64), sigmoid)
    assert got.shape == (n, I)
    rows = np.random.RandomState(1).choice(n, min(n, 24), replace=False)        # (every item of a sample of the users: oracle time)
    rows = np.unique(np.concatenate([rows, [0, n - 1]]))
    uu, ii = users[rows].repeat(I), np.tile(np.arange(I, dtype=np.int64), len(rows))
    inv, _, _ = O.forward(O.Tables(tabs), uu, ii, np.zeros_like(uu), sigmoid)
    if not sigmoid:        # (explicit forward's invariant score IS the dot product)
        pass
    np.testing.assert_array_equal(got.cpu().numpy()[rows].reshape(-1), inv)
    assert np.isfinite(got.cpu().numpy()).all()


def test_predict_mfma_equals_vector_alu_kernel_everywhere(tmp_path):
    import subprocess
    import sys
    script = tmp_path / 'pred.py'
    script.write_text('''
import sys, numpy as np, torch
from invpref_kdd_2022_amd import ops, synth
dev = torch.device('cuda:0')
out = {}
for (U, I, D, n) in ((900, 3706, 128, 513), (400, 5003, 256, 130), (300, 999, 64, 200)):
    tabs = synth.tables(5, U, I, 2, D, std=0.3)
    users = torch.from_numpy(np.random.RandomState(2).randint(0, U, n).astype(np.int64)).to(dev)
    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES[:2]]
    out[f'{D}'] = ops.predict(P[0], P[1], users, True).cpu().numpy()
np.savez(sys.argv[1], **out)
''')
    res = []
    for mm in ('1', '0'):
        f = tmp_path / f'p{mm}.npz'
        env = dict(os.environ, INVPREF_PREDICT_MM=mm, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        r = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res.append(np.load(f))
    for k in res[0].files:
        np.testing.assert_array_equal(res[0][k], res[1][k])


# ---------------------------------------------------------------------------------------------- top-k by radix select
@pytest.mark.parametrize('n_items,k,kind', [
    (51283, 40, 'uniform'), (51283, 40, 'quantised'), (51283, 64, 'equal'), (5000, 30, 'quantised'), (4097, 1, 'uniform'),
    (100000, 20, 'quantised'), (399999, 5, 'two-level'), (6000, 64, 'negative'),
])
def test_topk_radix_select_is_k_argmax_passes(n_items, k, kind):
    """evaluate.py:88-120 for large item counts: the radix select (three histogram passes over the value's key, two over the
    ids when more items tie with the k-th value than fit) must pick exactly what k argmax passes with the lowest id first
    among equal scores pick -- numpy's stable argsort of the masked / highlighted row -- on rows FULL of ties."""
    from invpref_kdd_2022_amd._capi import check, lib, ptr, stream_ptr
    rs = np.random.RandomState(n_items + k)
    n = 6
    r = rs.rand(n, n_items).astype(np.float32)
    if kind == 'quantised':
        r = np.round(r * 8) / 8              # nine distinct values: thousands of ties at the k-th place
    elif kind == 'equal':
        r[:] = 0.5
    elif kind == 'two-level':
        r = (r > 0.99999).astype(np.float32)
    elif kind == 'negative':
        r = -r
        r[:, ::5] = 0.0
        r[:, 1::5] = -0.0
    r = r.astype(np.float32)
    mask = [np.sort(rs.choice(n_items, rs.randint(0, 500), replace=False)) for _ in range(n)]
    pool = [np.sort(rs.choice(n_items, rs.randint(1, 3 * k), replace=False)) for _ in range(n)]
    truth = [np.sort(rs.choice(n_items, rs.randint(1, 50), replace=False)) for _ in range(n)]

    def csr(lists):
        p = np.zeros(len(lists) + 1, np.int32)
        p[1:] = np.cumsum([len(a) for a in lists])
        return torch.from_numpy(p).to(DEV), torch.from_numpy(np.concatenate(lists).astype(np.int32)).to(DEV)
    (mp, mi), (hp, hi), (tp, ti) = csr(mask), csr(pool), csr(truth)
    rd = torch.from_numpy(r).to(DEV)
    for use_pool in (False, True):
        items = torch.full((n, k), -1, dtype=torch.int32, device=DEV)
        hits = torch.full((n, k), -1, dtype=torch.float32, device=DEV)
        check(lib().invpref_eval_topk_hip(ptr(rd), n, n_items, ptr(mp), ptr(mi), ptr(hp) if use_pool else None,
                                          ptr(hi) if use_pool else None, ptr(tp), ptr(ti), k, ptr(items), ptr(hits),
                                          stream_ptr()), 'invpref_eval_topk_hip')
        for j in range(n):
            row = r[j].copy()
            row[mask[j]] = -1024.0
            if use_pool:
                row[pool[j]] += 1024.0
            order = np.argsort(-(row + 0.0), kind='stable')[:k]
            np.testing.assert_array_equal(items[j].cpu().numpy(), order, err_msg=f'user {j} pool {use_pool}')
            np.testing.assert_array_equal(hits[j].cpu().numpy(), np.isin(order, truth[j]).astype(np.float32))
