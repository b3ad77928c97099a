"""CPU, world_size 2 over gloo: the sharded training paths -- parallel.UserShard (default: users
partitioned, only the item-side gradient + small tables + loss tail all-reduced, Adam on the owned user rows)
and parallel.RowShard (rows split, everything all-reduced) -- with the FlatState layout, the per-step
exchange, the sharded E-step counts and sync_parameters() run through the REAL manager code with the HIP
ops swapped for the CPU oracle (test infrastructure), against the single-process oracle trajectory."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from invpref_kdd_2022_amd import ops, synth
from invpref_kdd_2022_amd.parallel import RowShard, UserShard
from oracle import oracle as O

U, I, E, D, N, B = 60, 40, 3, 16, 1000, 256
COEFS = dict(invariant_coe=2.0, env_aware_coe=8.6, env_coe=5.1, L2_coe=7.7, L1_coe=0.0015, alpha=1.7)
LR = 0.01


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _oracle_ops(monkey_target):
    """CPU stand-ins with the signatures of ops.mstep_grad / ops.adam_ / ops.estep / ops.stat_envs /
    ops.sample_weights, backed by the oracle."""

    def tables_of(params):
        return O.Tables({k: p.detach().numpy() for k, p in zip(O.PARAM_NAMES, params)})

    def mstep_grad(params, grads, users, items, envs, scores, weights, batch_norm, coefs, flags, losses6, ws):
        dense = bool(flags & 32)
        g, l = O.mstep(tables_of(params), users.numpy(), items.numpy(), envs.numpy(), scores.numpy(),
                       None if weights is None else weights.numpy(), np.asarray(coefs, np.float64), flags & 31,
                       bnorm=batch_norm, include_dense_reg=dense)
        for dst, src in zip(grads, g):
            dst += torch.from_numpy(src)
        losses6 += torch.from_numpy(l.astype(np.float32))

    def adam_(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, zero_grad=True):
        n = param.numel()
        O.adam(param.numpy(), grad[:n].numpy().copy(), m.numpy(), v.numpy(), step, lr)
        if zero_grad:
            grad.zero_()

    def estep(params, users, items, scores, implicit, old_envs, ws, eps_rows=None, new_envs=None, want_weights=True,
              perm_index=None, eps_base=None):
        if perm_index is not None:   # (the device unranks the drawn permutation rows; the stand-in does it with numpy)
            from invpref_kdd_2022_amd.train import _unrank_permutations
            eps_rows = _unrank_permutations(perm_index.numpy().astype(np.int64), np.asarray(eps_base, np.float32))
        new, counts, diff, _ = O.estep(tables_of(params), users.numpy(), items.numpy(), scores.numpy(), implicit,
                                       old_envs=None if old_envs is None else old_envs.numpy(), eps_rows=eps_rows)
        if new_envs is not None:
            new_envs.copy_(torch.from_numpy(new))
        return (new_envs if new_envs is not None else torch.from_numpy(new), torch.from_numpy(counts),
                torch.tensor([diff]), None, None)

    def stat_envs(envs, env_num, ws, want_sample_weights=True):
        c, cw, sw = O.stat_envs(envs.numpy(), env_num)
        return torch.from_numpy(c), torch.from_numpy(cw), torch.from_numpy(sw)

    def sample_weights(envs, counts, n_total, env_num):
        c = counts.numpy()
        cw = (np.minimum(c + 1, n_total - 1) / n_total).astype(np.float32)
        return torch.from_numpy(cw), torch.from_numpy(cw[envs.numpy()])

    def adam_ranges_(param, grad, m, v, offsets, lengths, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, zero_grad=True,
                     sched=None):
        for o, n in zip(offsets, lengths):
            adam_(param[o:o + n], grad[o:o + n], m[o:o + n], v[o:o + n], step, lr, zero_grad=zero_grad)

    def pack_rows(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok=True):
        idx = (row_offsets[:, None] + torch.arange(D)).reshape(-1)
        packed[:idx.numel()] = flat[idx]
        packed[idx.numel():idx.numel() + tail_len] = flat[tail_offset:tail_offset + tail_len]

    def unpack_rows(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok=True):
        idx = (row_offsets[:, None] + torch.arange(D)).reshape(-1)
        flat[idx] = packed[:idx.numel()]
        flat[tail_offset:tail_offset + tail_len] = packed[idx.numel():idx.numel() + tail_len]

    for name, fn in dict(mstep_grad=mstep_grad, adam_=adam_, adam_ranges_=adam_ranges_, estep=estep, stat_envs=stat_envs,
                         sample_weights=sample_weights, pack_rows=pack_rows, unpack_rows=unpack_rows).items():
        setattr(monkey_target, name, fn)


def _make_manager(rank, world, data, tabs):
    from invpref_kdd_2022_amd.models import InvPrefExplicit
    from invpref_kdd_2022_amd.train import ExplicitTrainManager

    class Stub:
        def evaluate(self):
            return {}

    os.environ['INVPREF_NO_PLAN'] = '1'  # the planned path is HIP-only; the sharding logic is the same
    model = InvPrefExplicit(U, I, E, D, reg_only_embed=False, reg_env_embed=True)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tabs.items()})
    np.random.seed(5)
    return ExplicitTrainManager(model=model, evaluator=Stub(), device=torch.device('cpu'),
                                training_data=torch.from_numpy(data), batch_size=B, epochs=2, cluster_interval=1,
                                evaluate_interval=10 ** 9, lr=LR, use_class_re_weight=True,
                                use_recommend_re_weight=True, cluster_use_random_sort=False, rank=rank,
                                world_size=world, **COEFS)


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), INVPREF_SHARD=mode)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _oracle_ops(ops)
        data = synth.interactions(3, U, I, N, implicit=False, zipf=False)
        tabs = synth.tables(4, U, I, E, D, std=0.2)
        mgr = _make_manager(rank, world, data, tabs)
        # this rank holds exactly its share of every minibatch, in order
        ref_shard = RowShard(N, B, rank, world) if mode == 'rows' else UserShard(data[:, 0], N, B, U, rank, world)
        rows = ref_shard.local_rows().numpy()
        assert mgr.shard_mode == mode and type(mgr.shard) is type(ref_shard)
        np.testing.assert_array_equal(mgr.users_tensor.numpy(), data[rows, 0])
        if mode == 'users':
            lo, hi = mgr.shard.user_range(U)
            assert ((data[rows, 0] >= lo) & (data[rows, 0] < hi)).all()
        mgr.stat_envs()
        losses = []
        for _ in range(2):
            for k in range(mgr.batch_num):
                lo, hi = mgr.shard.local_batch_bounds(k)
                losses.append(mgr.train_a_batch(mgr.users_tensor[lo:hi], mgr.items_tensor[lo:hi],
                                                mgr.scores_tensor[lo:hi], mgr.envs[lo:hi],
                                                mgr.sample_weights[lo:hi], mgr.alpha))
            diff = mgr.cluster()
            cnt = mgr.stat_envs()
        stale = mgr.state.param.numpy().copy()
        mgr.sync_parameters()   # user-sharded: foreign user rows were never updated on this rank
        if mode == 'users' and world > 1:
            assert not np.array_equal(stale, mgr.state.param.numpy())
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), param=mgr.state.param.numpy(), offsets=np.array(mgr.state.offsets),
                 losses=np.array([[d[k] for k in d] for d in losses]), diff=diff,
                 counts=np.array([cnt[e] for e in range(E)]), envs=mgr.envs.numpy(), rows=rows)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['users', 'rows'])
def test_sharded_training_matches_single_process(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    r = [np.load(tmp_path / f'rank{i}.npz') for i in range(world)]
    # replicas stay identical
    np.testing.assert_array_equal(r[0]['param'], r[1]['param'])
    np.testing.assert_array_equal(r[0]['losses'], r[1]['losses'])
    assert int(r[0]['diff']) == int(r[1]['diff'])
    np.testing.assert_array_equal(r[0]['counts'], r[1]['counts'])
    # single-process reference: the oracle's own loop on the whole data
    data = synth.interactions(3, U, I, N, implicit=False, zipf=False)
    tabs = synth.tables(4, U, I, E, D, std=0.2)
    np.random.seed(5)
    env0 = np.random.randint(0, E, N)
    cf = [COEFS[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    tr = O.Trainer(tabs, data, env0, implicit=False, batch_size=B, coefs=cf, lr=LR, reweight_rec=True,
                   reweight_cls=True, reg_only_embed=False, reg_env_embed=True)
    tr.stat_envs()
    ref_losses = []
    for _ in range(2):
        for lo in range(0, N, B):
            ref_losses.append(tr.train_a_batch(lo, min(lo + B, N)))
        diff = tr.cluster()
        cnt = tr.stat_envs()
    np.testing.assert_allclose(r[0]['losses'], np.array(ref_losses), rtol=2e-5)
    # sharded parameters == single-process parameters up to fp32 re-association of the shard sums
    from invpref_kdd_2022_amd.models import InvPrefExplicit
    from invpref_kdd_2022_amd.train import FlatState
    fs = FlatState(InvPrefExplicit(U, I, E, D).tables(), torch.device('cpu'))
    if mode == 'users':   # the user tables come first in the flat buffers
        assert r[0]['offsets'][2] < r[0]['offsets'][1] and fs.offsets[1] < fs.offsets[2]
    for arr, off, shp in zip(tr.tab.arrs, r[0]['offsets'], fs.shapes):
        got = r[0]['param'][off:off + arr.size].reshape(shp)
        assert np.abs(got - arr).max() < 0.05 * LR
        assert np.quantile(np.abs(got - arr), 0.99) < 5e-6
    # E-step: assignments of every row, gathered back from the two shards
    envs = np.empty(N, np.int64)
    for x in r:
        envs[x['rows']] = x['envs']
    mism = int((envs != tr.envs).sum())
    assert mism <= 2
    assert abs(int(r[0]['diff']) - diff) <= mism and np.abs(r[0]['counts'] - np.array([cnt[e] for e in range(E)])).sum() <= 2 * mism


def test_user_shard_index_arithmetic():
    rs = np.random.RandomState(0)
    for n, b, users_n, w in ((1000, 256, 60, 2), (250154, 8192, 15400, 8), (7, 3, 5, 4), (5, 10, 3, 3)):
        users = rs.randint(0, users_n, n)
        shards = [UserShard(users, n, b, users_n, r, w) for r in range(w)]
        allrows = np.concatenate([s.local_rows().numpy() for s in shards])
        assert sorted(allrows.tolist()) == list(range(n))           # a partition of the rows
        ranges = [s.user_range(users_n) for s in shards]
        assert ranges[0][0] == 0 and ranges[-1][1] == users_n and all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
        for s in shards:
            lo, hi = s.user_range(users_n)
            rows = s.local_rows().numpy()
            assert ((users[rows] >= lo) & (users[rows] < hi)).all()  # only this rank's users
            assert s.n_local == len(rows)
            for k in range(s.batch_num):
                a, c = s.local_batch_bounds(k)
                g = rows[a:c]
                assert ((g >= k * b) & (g < min((k + 1) * b, n))).all() and (np.diff(g) > 0).all()   # minibatch order kept
                np.testing.assert_array_equal(s.select_in_batch(k, np.arange(k * b, k * b + s.global_batch_len(k))), g)


def test_row_shard_index_arithmetic():
    for n, b, w in ((1000, 256, 2), (250154, 8192, 8), (7, 3, 4), (5, 10, 3)):
        shards = [RowShard(n, b, r, w) for r in range(w)]
        allrows = np.concatenate([s.local_rows().numpy() for s in shards])
        assert sorted(allrows.tolist()) == list(range(n))           # a partition of the rows
        for k in range(shards[0].batch_num):
            parts = [s.global_rows_of_batch(k) for s in shards]
            assert parts[0][0] == k * b and parts[-1][1] == min((k + 1) * b, n)
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))   # contiguous, in rank order
            assert max(p[1] - p[0] for p in parts) - min(p[1] - p[0] for p in parts) <= 1   # balanced
            for s in shards:
                lo, hi = s.local_batch_bounds(k)
                g0, g1 = s.global_rows_of_batch(k)
                np.testing.assert_array_equal(s.local_rows().numpy()[lo:hi], np.arange(g0, g1))


# ---- Yahoo-shaped runs through the EPOCH loop (train_epochs -> _raw_step: what bench.py and train() use), world 4 / 8:
# BASELINE.json configs[3] -- 15 400 x 1 000, 250 154 interactions, global minibatch 8 192 cut `world` ways (1 024 rows per
# rank at world 8), 31 minibatches of which the last is ragged (4 394 rows: uneven slices) -- both shard layouts.
YU, YI, YE, YD, YN, YB = 15400, 1000, 4, 16, 250154, 8192


def _yahoo_worker(rank, world, port, out_dir, mode):
    mode, _, exchange = mode.partition('-')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), INVPREF_SHARD=mode, INVPREF_NO_PLAN='1',
                      OMP_NUM_THREADS='1', INVPREF_EXCHANGE=exchange or 'scatter')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _oracle_ops(ops)
        from invpref_kdd_2022_amd.models import InvPrefImplicit
        from invpref_kdd_2022_amd.train import ImplicitTrainManager, LOSS_KEYS

        class Stub:
            def evaluate(self):
                return {}
        data = synth.interactions(17373331, YU, YI, YN, implicit=True)
        tabs = synth.tables(8, YU, YI, YE, YD, std=0.1)
        model = InvPrefImplicit(YU, YI, YE, YD, reg_only_embed=True, reg_env_embed=False)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in tabs.items()})
        np.random.seed(5)
        mgr = ImplicitTrainManager(model=model, evaluator=Stub(), device=torch.device('cpu'),
                                   training_data=torch.from_numpy(data), batch_size=YB, epochs=2, cluster_interval=1,
                                   evaluate_interval=10 ** 9, lr=0.005, use_class_re_weight=True,
                                   use_recommend_re_weight=False, cluster_use_random_sort=False, rank=rank,
                                   world_size=world, invariant_coe=3.35, env_aware_coe=9.99, env_coe=9.06, L2_coe=3.13,
                                   L1_coe=0.49, alpha=1.9)
        assert mgr.batch_num == 31 and mgr.shard.global_batch_len(30) == YN - 30 * YB
        if mode == 'rows':
            # scatter exchange: this rank applies Adam to ONE slice of whole 256-byte lines, the slices tile the buffer
            st = mgr.state
            assert mgr.exchange == (exchange or 'scatter')
            if mgr.exchange == 'scatter':
                chunk = st.cap // world
                assert st.cap % (world * 64) == 0 and st.cap >= st.n and st.cap - st.n < world * 64
                assert mgr._adam_ranges == [(rank * chunk, min(chunk, st.n - rank * chunk))]
            else:
                assert mgr._adam_ranges == [(0, st.n)]   # the ragged last minibatch: slices differ by at most one row and tile it exactly
            lens = [RowShard(YN, YB, r, world).slice_in_batch(30) for r in range(world)]
            assert lens[0][0] == 0 and lens[-1][1] == YN - 30 * YB and max(b - a for a, b in lens) - min(b - a for a, b in lens) <= 1
        mgr.stat_envs()
        l1 = mgr.train_epochs(1)[0]
        diff = mgr.cluster()
        cnt = mgr.stat_envs()
        l2 = mgr.train_epochs(1)[0]
        mgr.sync_parameters()
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), param=mgr.state.param.numpy(),
                 offsets=np.array(mgr.state.offsets), losses=np.array([[l[k] for k in LOSS_KEYS] for l in (l1, l2)]),
                 diff=diff, counts=np.array([cnt[e] for e in range(YE)]), envs=mgr.envs.numpy(),
                 rows=mgr.shard.local_rows().numpy(), ar_floats=mgr.state.n + 8 - mgr._ar_lo,
                 packed_floats=np.array(getattr(mgr, 'packed_floats', [0])))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world,mode', [(4, 'rows'), (8, 'rows'), (8, 'rows-allreduce'), (8, 'rows-packed'), (8, 'users')])
def test_yahoo_shaped_epochs_world_4_and_8(tmp_path, world, mode):
    mp.spawn(_yahoo_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    r = [np.load(tmp_path / f'rank{i}.npz') for i in range(world)]
    for x in r[1:]:   # replicas agree
        np.testing.assert_array_equal(r[0]['param'], x['param'])
        np.testing.assert_array_equal(r[0]['losses'], x['losses'])
        assert int(r[0]['diff']) == int(x['diff'])
    P = 2 * (YU + YI) * YD + 2 * YE * YD + YE
    if mode.startswith('rows'):
        # BASELINE configs[3] / SURVEY 8(d)-4 as written: every GLOBAL minibatch of 8 192 rows cut `world` ways -- 1 024 rows per
        # rank at eight ranks -- and the ragged last minibatch (250 154 - 30 x 8 192 = 4 394 rows) 549 / 550 per rank
        per = YB // world
        for i, x in enumerate(r):
            rows = np.sort(x['rows'])
            for k in (0, 17, 29):
                assert int(((rows >= k * YB) & (rows < (k + 1) * YB)).sum()) == per
                assert rows[(rows >= k * YB)][0] == k * YB + i * per        # contiguous slices, rank order (utils.py:12-19)
            last = int((rows >= 30 * YB).sum())
            assert last in ((YN - 30 * YB) // world, (YN - 30 * YB) // world + 1)
        assert sum(int((x['rows'] >= 30 * YB).sum()) for x in r) == YN - 30 * YB == 4394
        if world == 8:
            assert per == 1024 and {int((x['rows'] >= 30 * YB).sum()) for x in r} == {549, 550}
    if mode == 'rows-packed':
        # one all-reduce per step over the rows the GLOBAL minibatch touches + the small tables: at this split (8 192
        # interactions per global minibatch over 15 400 users) well under half of the flat gradient
        pf = r[0]['packed_floats']
        assert len(pf) == 31 and 0 < pf.max() < 0.5 * P and pf.min() >= 2 * YE * YD + YE
    elif mode.startswith('rows'):
        assert int(r[0]['ar_floats']) >= P + 8            # the whole flat gradient + the loss tail, one all-reduce
    else:
        assert int(r[0]['ar_floats']) < 2 * YI * YD + 2 * YE * YD + YE + 8 + 5 * 64   # item side + small tables only
    # single process: the oracle's own loop over the whole data
    data = synth.interactions(17373331, YU, YI, YN, implicit=True)
    tabs = synth.tables(8, YU, YI, YE, YD, std=0.1)
    np.random.seed(5)
    env0 = np.random.randint(0, YE, YN)
    tr = O.Trainer(tabs, data, env0, implicit=True, batch_size=YB, coefs=[3.35, 9.99, 9.06, 3.13, 0.49, 1.9], lr=0.005,
                   reweight_rec=False, reweight_cls=True, reg_only_embed=True, reg_env_embed=False)
    tr.stat_envs()
    ref = [tr.train_a_epoch()]
    diff = tr.cluster()
    cnt = tr.stat_envs()
    ref.append(tr.train_a_epoch())
    np.testing.assert_allclose(r[0]['losses'], np.array(ref), rtol=2e-5)
    envs = np.empty(YN, np.int64)
    for x in r:
        envs[x['rows']] = x['envs']
    mism = int((envs != tr.envs).sum())
    assert mism <= 8 and abs(int(r[0]['diff']) - diff) <= mism
    assert np.abs(r[0]['counts'] - np.array([cnt[e] for e in range(YE)])).sum() <= 2 * mism
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    from invpref_kdd_2022_amd.train import FlatState
    fs = FlatState(InvPrefImplicit(YU, YI, YE, YD).tables(), torch.device('cpu'))
    for arr, off, shp in zip(tr.tab.arrs, r[0]['offsets'], fs.shapes):
        got = r[0]['param'][off:off + arr.size].reshape(shp)
        assert np.abs(got - arr).max() < 2.5 * 0.005       # (a noise-level gradient may step the other way, twice)
        if arr.size > 4096:   # the big tables; the E x D ones are sums over every interaction, re-associated 8 ways
            assert np.quantile(np.abs(got - arr), 0.99) < 5e-5
        else:
            assert np.abs(got - arr).max() < 1e-3


# ---- PureMF managers sharded: the planned gradient pass (HIP-only) is stood in for by the oracle
def _pure_worker(rank, world, port, out_dir, mode):
    mode, _, exchange = mode.partition('-')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), INVPREF_SHARD=mode, INVPREF_EXCHANGE=exchange or 'scatter')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _oracle_ops(ops)
        from invpref_kdd_2022_amd import plan as planlib
        from invpref_kdd_2022_amd.baseline import BasicImplicitTrainManager, PureMatrixFactorization

        def mstep_rows_grad(params, grads, dplan, envs, scores, weights, batch_norm, coefs, flags, losses6, ws, sched=None):
            names = [f for f, _ in planlib.RowPlanStruct._fields_]
            meta = dict(zip(names, dplan.meta.tolist()))
            buf, n = dplan.buf.numpy(), meta['n']
            # the minibatch in its own order, from the two sorted lists of the plan: {item, position, label, slot}, {user, slot}
            ul = buf[meta['user_list']:meta['user_list'] + 4 * n].reshape(n, 4)
            il = buf[meta['item_list']:meta['item_list'] + 2 * n].reshape(n, 2)
            u, v = np.zeros(n, np.int64), np.zeros(n, np.int64)
            v[ul[:, 1]] = ul[:, 0]
            u[ul[:, 1]] = il[ul[:, 3], 0]
            tab = O.Tables(O.pure_mf_params(params[0].detach().numpy(), params[1].detach().numpy()))
            g, l = O.mstep(tab, u, v, np.zeros(n, np.int64), scores.numpy(), None, np.asarray(coefs, np.float64),
                           O.flags_of(bool(flags & 1), False, False, True, False), bnorm=batch_norm, include_dense_reg=False)
            grads[0].copy_(torch.from_numpy(g[0]))     # the planned pass OVERWRITES every row
            grads[1].copy_(torch.from_numpy(g[1]))
            losses6 += torch.from_numpy(l.astype(np.float32))
        ops.mstep_rows_grad = mstep_rows_grad

        class Stub:
            def evaluate(self):
                return {}
        data = synth.interactions(3, U, I, N, implicit=True, zipf=False)
        rs = np.random.RandomState(4)
        model = PureMatrixFactorization(U, I, D)
        model.load_state_dict({'user_emb.weight': torch.from_numpy((rs.randn(U, D) * 0.2).astype(np.float32)),
                               'item_emb.weight': torch.from_numpy((rs.randn(I, D) * 0.2).astype(np.float32))})
        mgr = BasicImplicitTrainManager(model, Stub(), torch.device('cpu'), torch.from_numpy(data), B, 3, 10 ** 9, LR,
                                        0.01, 0.001, rank=rank, world_size=world)
        assert mgr.shard_mode == mode and mgr.users_tensor.shape[0] < N
        if exchange:
            assert mgr.exchange == exchange
        if exchange == 'packed':   # two tables, no small ones: rows only, fewer than the whole flat gradient
            assert mgr._packed_tail[1] == 0 and 0 < max(mgr.packed_floats) <= (U + I) * D
        (losses, _), _ = mgr.train(silent=True)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), pu=model.user_emb.weight.detach().numpy(),
                 qi=model.item_emb.weight.detach().numpy(), losses=np.array([[l[k] for k in l] for l in losses]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['users', 'rows', 'rows-packed'])
def test_pure_mf_manager_sharded(tmp_path, mode):
    world = 2
    mp.spawn(_pure_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    r = [np.load(tmp_path / f'rank{i}.npz') for i in range(world)]
    np.testing.assert_array_equal(r[0]['pu'], r[1]['pu'])       # after sync_parameters every rank holds the model
    np.testing.assert_array_equal(r[0]['qi'], r[1]['qi'])
    np.testing.assert_array_equal(r[0]['losses'], r[1]['losses'])
    data = synth.interactions(3, U, I, N, implicit=True, zipf=False)
    rs = np.random.RandomState(4)
    pu0, qi0 = (rs.randn(U, D) * 0.2).astype(np.float32), (rs.randn(I, D) * 0.2).astype(np.float32)
    tr = O.pure_mf_trainer(pu0, qi0, data, implicit=True, batch_size=B, lr=LR, L2_coe=0.01, L1_coe=0.001)
    ref = O.pure_mf_losses(np.stack([tr.train_a_epoch() for _ in range(3)]))
    np.testing.assert_allclose(r[0]['losses'], ref, rtol=2e-5)
    assert np.abs(r[0]['pu'] - tr.tab.arrs[0]).max() < 0.05 * LR and np.abs(r[0]['qi'] - tr.tab.arrs[1]).max() < 0.05 * LR
