"""GPU: the "next" rows of SURVEY.md §8(f) beyond evaluation -- PureMF baselines (f2) through the drop-in
modules / managers of invpref_kdd_2022_amd.baseline against the goldens recorded from the reference (g7) and the
oracle; the loader -> evaluator hand-over (f3); the static-popularity manager (f4, golden g9)."""
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import ops, plan as planlib, synth
from invpref_kdd_2022_amd.baseline import (PURE_LOSS_KEYS, BasicExplicitTrainManager, BasicImplicitTrainManager,
                                           BasicUniformImplicitTrainManager, PureExplicitMatrixFactorization,
                                           PureMatrixFactorization)
from oracle import oracle as O
from pure_mf_fixture import pure_mf_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
DEV = torch.device('cuda:0')


class StubEvaluator:
    def evaluate(self):
        return {'stub': 0.0}


def _model(kind, init, U, I, D):
    m = (PureMatrixFactorization if kind == 'implicit' else PureExplicitMatrixFactorization)(U, I, D)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in init.items()})
    return m


@pytest.mark.parametrize('kind', ['implicit', 'explicit'])
def test_fused_step_vs_reference_and_oracle(kind):
    """One fused PureMF step through the C ABI (INVPREF_PURE_MF, absent tables = NULL): loss terms vs the
    reference (1e-5), parameters after the step vs oracle gradient + oracle Adam."""
    z = np.load(os.path.join(G, f'g7_pure_mf_{kind}.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs(kind)
    P = [torch.from_numpy(init[k]).to(DEV) for k in ('user_emb.weight', 'item_emb.weight')]
    P2, M, V = ([torch.zeros_like(p) for p in P] for _ in range(3))
    dp = planlib.upload(planlib.build_row_plan(data[:bs, 0], data[:bs, 1], data[:bs, 2], U, I, factor_num=D), DEV)
    y = torch.from_numpy(data[:bs, 2].astype(np.float32)).to(DEV)
    losses = torch.zeros(6, device=DEV)
    flags = ops.flags_of(kind == 'implicit', False, False, True, False, dense_reg=False)
    coefs = O.pure_mf_coefs(cfg['L2_coe'], cfg['L1_coe'])
    ops.mstep_rows_adam(P, P2, M, V, dp, None, y, None, bs, coefs, flags, losses, 1, cfg['lr'], ops.Workspace(DEV), pure=True)
    np.testing.assert_allclose(O.pure_mf_losses(losses.cpu().numpy()), z['step_losses_f32'], rtol=1e-5)
    tab = O.Tables(O.pure_mf_params(init['user_emb.weight'], init['item_emb.weight']))
    grads, _ = O.mstep(tab, data[:bs, 0], data[:bs, 1], np.zeros(bs, np.int64), data[:bs, 2], None, coefs,
                       O.flags_of(kind == 'implicit', False, False, True, False))
    for p0, g, p2, ref_g in zip(tab.arrs[:2], grads[:2], P2, (z['step_g_user_f32'], z['step_g_item_f32'])):
        assert np.abs(g - ref_g).max() <= 2e-6 * np.abs(ref_g).max()
        po = p0.reshape(-1).copy()
        O.adam(po, g.reshape(-1), np.zeros_like(po), np.zeros_like(po), 1, cfg['lr'])
        # first Adam step moves every touched element by ~lr: compare the moves
        assert np.abs(p2.cpu().numpy().reshape(-1) - po).max() < 0.02 * cfg['lr']


@pytest.mark.parametrize('kind', ['implicit', 'explicit'])
def test_manager_trajectory_vs_reference(kind):
    z = np.load(os.path.join(G, f'g7_pure_mf_{kind}.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs(kind)
    model = _model(kind, init, U, I, D)
    cls = BasicImplicitTrainManager if kind == 'implicit' else BasicExplicitTrainManager
    mgr = cls(model=model, evaluator=StubEvaluator(), device=DEV, training_data=torch.from_numpy(data).to(DEV),
              batch_size=bs, epochs=epochs, evaluate_interval=10 ** 9, lr=cfg['lr'], L2_coe=cfg['L2_coe'],
              L1_coe=cfg['L1_coe'])
    (losses, loss_epochs), (tests, test_epochs) = mgr.train(silent=True)
    assert loss_epochs == list(z['loss_epochs']) and test_epochs == list(z['test_epochs'])
    assert list(losses[0].keys()) == PURE_LOSS_KEYS
    np.testing.assert_allclose([[d[k] for k in PURE_LOSS_KEYS] for d in losses], z['traj'], rtol=2e-5)
    sd = model.state_dict()
    assert set(sd.keys()) == {'user_emb.weight', 'item_emb.weight'}
    for k in sd:
        assert np.abs(sd[k].cpu().numpy() - z['final_' + k]).max() < 1e-3, k
    # epoch by epoch (one read-back each, HIP graphs warm) continues the same trajectory as the oracle
    tr = O.pure_mf_trainer(init['user_emb.weight'], init['item_emb.weight'], data, implicit=(kind == 'implicit'),
                           batch_size=bs, lr=cfg['lr'], L2_coe=cfg['L2_coe'], L1_coe=cfg['L1_coe'])
    for _ in range(epochs):
        tr.train_a_epoch()
    want = O.pure_mf_losses(tr.train_a_epoch())
    got = mgr.train_a_epoch()
    np.testing.assert_allclose([got[k] for k in PURE_LOSS_KEYS], want, rtol=5e-5)


@pytest.mark.parametrize('mode', ['rows', 'users'])
def test_manager_sharded_sequence_vs_reference(monkeypatch, mode):
    """The PureMF managers on the multi-GPU step sequence (planned gradient pass -> RCCL all-reduce -> ranged Adam,
    whole epochs replayed as HIP graphs), run on a 1-rank RCCL group: the reference's trajectory (g7)."""
    import torch.distributed as dist
    monkeypatch.setenv('INVPREF_FORCE_SHARDED_PATH', '1')
    monkeypatch.setenv('INVPREF_SHARD', mode)
    z = np.load(os.path.join(G, 'g7_pure_mf_implicit.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs('implicit')
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29536')
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        model = _model('implicit', init, U, I, D)
        mgr = BasicImplicitTrainManager(model=model, evaluator=StubEvaluator(), device=DEV,
                                        training_data=torch.from_numpy(data).to(DEV), batch_size=bs, epochs=epochs,
                                        evaluate_interval=10 ** 9, lr=cfg['lr'], L2_coe=cfg['L2_coe'], L1_coe=cfg['L1_coe'])
        assert mgr.shard_mode == mode and not mgr._fused_seq()
        (losses, _), _ = mgr.train(silent=True)
        assert mgr._graphs
        np.testing.assert_allclose([[d[k] for k in PURE_LOSS_KEYS] for d in losses], z['traj'], rtol=2e-5)
        sd = model.state_dict()
        for k in sd:
            assert np.abs(sd[k].cpu().numpy() - z['final_' + k]).max() < 1e-3, k
    finally:
        dist.destroy_process_group()


def test_train_a_batch_and_unfused_surface():
    """train_a_batch on caller-supplied tensors (plan built on the fly) equals the first step of the
    reference; the module's unfused surface (forward / get_L*_reg / autograd / predict) matches g7."""
    kind = 'implicit'
    z = np.load(os.path.join(G, f'g7_pure_mf_{kind}.npz'))
    (U, I, D, n, bs, epochs), data, init, cfg = pure_mf_inputs(kind)
    model = _model(kind, init, U, I, D)
    td = torch.from_numpy(data).to(DEV)
    mgr = BasicUniformImplicitTrainManager(model, StubEvaluator(), DEV, td, td[:100], bs, epochs, 10 ** 9, cfg['lr'],
                                           cfg['L2_coe'], cfg['L1_coe'])
    assert mgr.uniform_user.shape == (100,)
    # unfused surface first (parameters still at init)
    u, v, y = td[:bs, 0], td[:bs, 1], td[:bs, 2].float()
    sl = model(u, v, y)
    l2, l1 = model.get_L2_reg(u, v), model.get_L1_reg(u, v)
    loss = sl + l2 * cfg['L2_coe'] + l1 * cfg['L1_coe']
    np.testing.assert_allclose([float(sl), float(l2), float(l1), float(loss)], z['step_losses_f32'], rtol=1e-5)
    loss.backward()
    for p, ref in ((model.user_emb.weight, z['step_g_user_f32']), (model.item_emb.weight, z['step_g_item_f32'])):
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    np.testing.assert_allclose(model(u[:512], v[:512]).detach().cpu().numpy(), z['step_scores'], rtol=2e-6, atol=1e-7)
    pr = model.predict(u[:8])
    assert pr.shape == (8, I)
    np.testing.assert_allclose(pr[0, int(v[0])].item(), z['step_scores'][0], rtol=2e-6)
    # fused train_a_batch
    d = mgr.train_a_batch(u, v, y)
    np.testing.assert_allclose([d[k] for k in PURE_LOSS_KEYS], z['step_losses_f32'], rtol=1e-5)
    with pytest.raises(AttributeError):
        mgr.cluster()


def test_loader_to_evaluator_hand_over():
    """dataloader.py's CSR hand-over gives ImplicitTestManager the same metrics as walking the loader's
    python sets (the reference interface), on the committed fixture data set, with and without item pool."""
    from invpref_kdd_2022_amd import dataloader as dl
    from invpref_kdd_2022_amd.evaluate import ImplicitTestManager

    class SetsOnly:  # hides csr_for_eval: forces the set-walking path
        def __init__(self, ld):
            self._ld = ld
            self.all_test_users_by_sorted_list = ld.all_test_users_by_sorted_list
            self.get_sorted_all_test_users_ground_truth = ld.get_sorted_all_test_users_ground_truth

        def user_mask_items(self, u):
            return self._ld.user_mask_items(u) if u < len(self._ld.user_positive_interaction) else set()

        def user_highlight_items(self, u):
            return self._ld.user_highlight_items(u)

    ld = dl.YahooImplicitBCELossDataLoader(os.path.join(G, 'ds_small', 'implicit'), DEV, has_item_pool_file=True)
    torch.manual_seed(3)
    model = PureMatrixFactorization(ld.user_num, ld.item_num, 16).to(DEV)
    with torch.no_grad():
        model.user_emb.weight.mul_(30.)
    for pool in (False, True):
        a = ImplicitTestManager(model, ld, 16, [3, 5], use_item_pool=pool).evaluate()
        b = ImplicitTestManager(model, SetsOnly(ld), 16, [3, 5], use_item_pool=pool).evaluate()
        assert a == b and set(a) == {'ndcg', 'recall', 'precision'} and 0. < a['recall'][5] <= 1.


def test_static_pop_manager_vs_reference():
    """ImplicitTrainStaticPopularityManager.static_pop / final_cluster_stat on the fixture data set
    with the env assignment of golden g9 (one empty environment -> NaN like np.mean of nothing)."""
    from invpref_kdd_2022_amd import dataloader as dl
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    from invpref_kdd_2022_amd.train import ImplicitTrainStaticPopularityManager
    z = np.load(os.path.join(G, 'g9_static_pop.npz'))
    ld = dl.ImplicitBCELossDataLoaderStaticPopularity(os.path.join(G, 'ds_small', 'implicit'), DEV, has_item_pool_file=True)
    np.testing.assert_array_equal(ld.user_inter_cnt_np, z['user_cnt'])
    np.testing.assert_array_equal(ld.item_inter_cnt_np, z['item_cnt'])
    np.testing.assert_array_equal(ld.user_inter_cnt_normalize_np, z['user_norm'])
    np.testing.assert_array_equal(ld.item_inter_cnt_normalize_np, z['item_norm'])
    E = int(z['E'])
    np.random.seed(1)
    mgr = ImplicitTrainStaticPopularityManager(
        model=InvPrefImplicit(ld.user_num, ld.item_num, E, 8), evaluator=StubEvaluator(), device=DEV, data_loader=ld,
        training_data=torch.from_numpy(ld.train_data_np).to(DEV), batch_size=256, epochs=4, cluster_interval=3,
        evaluate_interval=2, lr=0.01, invariant_coe=1., env_aware_coe=1., env_coe=1., L2_coe=0.1, L1_coe=0.1,
        static_pop_interval=2, alpha=1., cluster_use_random_sort=False)
    mgr.envs.copy_(torch.from_numpy(z['envs']))
    res = mgr.static_pop()
    assert list(res.keys()) == ops.POP_KEYS and list(res[ops.POP_KEYS[0]].keys()) == list(range(E))
    got = np.array([[res[k][e] for k in ops.POP_KEYS] for e in range(E)])
    assert np.isnan(got[3]).all() and np.isnan(z['pop'][3]).all()
    keep = [0, 1, 2, 4]
    np.testing.assert_allclose(got[keep], z['pop'][keep], rtol=1e-13)
    np.testing.assert_array_equal(got[keep][:, [0, 1, 4, 5, 8]], z['pop'][keep][:, [0, 1, 4, 5, 8]])  # integer sums: exact
    uc, ic, un, inn, colors = mgr.final_cluster_stat(['c0', 'c1', 'c2', 'c3', 'c4'])
    np.testing.assert_array_equal(uc, z['fcs_user_cnt'])
    np.testing.assert_array_equal(ic, z['fcs_item_cnt'])
    np.testing.assert_array_equal(un, z['fcs_user_norm'])
    np.testing.assert_array_equal(inn, z['fcs_item_norm'])
    assert [int(c[1]) for c in colors] == list(z['fcs_color_idx'])
    # the outer loop: 4 epochs, evaluate every 2, cluster every 3, statistics every 2 -> four result tuples
    (l, le), (t, te), (d, c, ce), (stat, se) = mgr.train(silent=True)
    assert le == [1, 2, 3, 4] and te == [0, 2, 4] and ce == [3] and se == [2, 4] and len(d) == 1
    assert set(stat.keys()) == set(ops.POP_KEYS) and len(stat[ops.POP_KEYS[0]][0]) == 2
    assert mgr.epochs == 4


@pytest.mark.parametrize('scale', [1e-2, 1e-8])
@pytest.mark.parametrize('E', [2, 4, 5, 8, 13])
def test_estep_unranks_the_permutation_rows_on_the_device(E, scale):
    """cluster_use_random_sort (train.py:86-92, :192-196) with only the drawn permutation INDEX on the device: the E-step's
    assignments, counts and weights equal, bit for bit, those of the E-step fed the gathered N x E rows of
    itertools.permutations order (built here with the manager's host-side unranking, pinned to itertools by a CPU test)."""
    import math
    from invpref_kdd_2022_amd.train import _unrank_permutations
    U, I, D, N = 300, 200, 64, 20000
    rs = np.random.RandomState(E)
    tabs = synth.tables(E + 1, U, I, E, D, std=0.3)
    u, v = rs.randint(0, U, N), rs.randint(0, I, N)
    y = rs.randint(0, 2, N).astype(np.float32)
    # every distance of a row ties (no env-aware part) and the permuted vector is large enough to survive the fp32 sum
    # (the reference's own 1e-10 .. 1e-25 are below one ulp of the distances: SURVEY 7), so the tie-break decides every row
    for k in ('embed_user_env_aware.weight', 'embed_item_env_aware.weight', 'embed_env.weight'):
        tabs[k] = np.zeros_like(tabs[k])
    # (scale 1e-8: the kernel skips the permutation row of an interaction whose smallest distance is above 2^26 max|eps|
    #  = 0.67 -- there the add is a no-op in fp32 -- and looks it up for the others: both kinds occur among these BCE distances)
    base = np.array([scale * (0.5 ** i) for i in range(E)], dtype=np.float32)
    idx = rs.randint(0, math.factorial(E), N)
    dt = np.uint8 if E <= 5 else (np.int32 if E <= 12 else np.int64)
    P = [torch.from_numpy(tabs[k]).to(DEV) for k in ops.PARAM_NAMES]
    ws = ops.Workspace(DEV)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
    old = dev(rs.randint(0, E, N).astype(np.int64))
    rows = dev(_unrank_permutations(idx, base))
    a = ops.estep(P, dev(u), dev(v), dev(y), True, old, ws, eps_rows=rows)
    b = ops.estep(P, dev(u), dev(v), dev(y), True, old, ws, perm_index=dev(idx.astype(dt)), eps_base=base.tolist())
    for x, z in zip(a, b):
        np.testing.assert_array_equal(x.cpu().numpy(), z.cpu().numpy())
    # the tie-break really decided rows: without it everything ties at environment 0
    plain = ops.estep(P, dev(u), dev(v), dev(y), True, old, ws)[0].cpu().numpy()
    assert (plain == 0).all() and (scale < 1e-3 or (a[0].cpu().numpy() != 0).any())


def test_estep_permutation_index_at_a_ragged_large_count():
    """The LDS-table form fetches an interaction's permutation index one pass ahead of its use (no chunk of indices in LDS any
    more): a count that is no multiple of the 16 interactions of a pass, large enough for every workgroup to walk many passes,
    one- and four-byte indices -- assignments, counts and weights equal to the gathered-rows form bit for bit."""
    import math
    from invpref_kdd_2022_amd.train import _unrank_permutations
    for E, dt in ((4, np.uint8), (6, np.int32)):
        U, I, D, N = 900, 400, 64, 600011
        rs = np.random.RandomState(11 * E)
        tabs = synth.tables(E + 3, U, I, E, D, std=0.3)
        for k in ('embed_user_env_aware.weight', 'embed_item_env_aware.weight', 'embed_env.weight'):
            tabs[k] = np.zeros_like(tabs[k])   # every distance of a row ties: the tie-break decides
        u, v = rs.randint(0, U, N), rs.randint(0, I, N)
        y = rs.randint(0, 2, N).astype(np.float32)
        base = np.array([1e-2 * (0.5 ** i) for i in range(E)], dtype=np.float32)
        idx = rs.randint(0, math.factorial(E), N)
        P = [torch.from_numpy(tabs[k]).to(DEV) for k in ops.PARAM_NAMES]
        ws = ops.Workspace(DEV)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)   # noqa: E731
        old = dev(rs.randint(0, E, N).astype(np.int64))
        a = ops.estep(P, dev(u), dev(v), dev(y), True, old, ws, eps_rows=dev(_unrank_permutations(idx, base)))
        b = ops.estep(P, dev(u), dev(v), dev(y), True, old, ws, perm_index=dev(idx.astype(dt)), eps_base=base.tolist())
        for x, z in zip(a, b):
            np.testing.assert_array_equal(x.cpu().numpy(), z.cpu().numpy())
        assert (a[0].cpu().numpy() != 0).any()


def test_estep_row_offsets_32_and_64_bit_agree():
    """estep_assign_kernel addresses rows with 32-bit offsets whenever every table is under 4 GB and with 64-bit products
    otherwise; INVPREF_ESTEP_OFFSETS64=1 (read once per process: a child process here) forces the second form -- the same
    assignments, counts and weights bit for bit, at a row length that takes the float4 path and one that does not."""
    import subprocess
    import sys
    code = (
        "import numpy as np, torch, sys\n"
        "from invpref_kdd_2022_amd import ops, synth\n"
        "dev = torch.device('cuda:0')\n"
        "out = []\n"
        "for D, E in ((64, 4), (30, 3), (256, 16)):\n"
        "    U, I, N = 700, 300, 30000\n"
        "    rs = np.random.RandomState(D)\n"
        "    tabs = synth.tables(D + 1, U, I, E, D, std=0.3)\n"
        "    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]\n"
        "    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)\n"
        "    r = ops.estep(P, t(rs.randint(0, U, N)), t(rs.randint(0, I, N)), t(rs.randint(0, 2, N).astype(np.float32)), True,\n"
        "                  t(rs.randint(0, E, N).astype(np.int64)), ops.Workspace(dev))\n"
        "    out += [x.cpu().numpy() for x in r if x is not None]\n"
        "np.savez(sys.argv[1], *out)\n")
    import tempfile
    res = []
    with tempfile.TemporaryDirectory() as d:
        for flag in ('0', '1'):
            path = os.path.join(d, f'r{flag}.npz')
            env = dict(os.environ, INVPREF_ESTEP_OFFSETS64=flag)
            subprocess.run([sys.executable, '-c', code, path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            z = np.load(path)
            res.append([z[k] for k in z.files])
    assert len(res[0]) == len(res[1]) >= 12
    for a, b in zip(*res):
        np.testing.assert_array_equal(a, b)
