"""CPU: invariants of the host-side row plan (plan.py) that the planned M-step kernels rely on."""
import numpy as np
import pytest

from invpref_kdd_2022_amd import plan as planlib, synth


def check_plan(users, items, U, I, D=64, **kw):
    y = (np.arange(len(users)) % 5).astype(np.float32)
    p = planlib.build_row_plan(users, items, y, U, I, factor_num=D, **kw)
    n = len(users)
    lanes = p['lanes_per_group']
    ng = 256 // lanes
    assert lanes == planlib.lanes_of(D) and p['n'] == n
    ud, idd = p['user_desc'], p['item_desc']
    assert ud.shape[1:] == (ng, 8) and idd.shape[1:] == (ng, 8)
    assert len(ud) % p['user_rounds_per_task'] == 0 and len(idd) % p['item_rounds_per_task'] == 0
    assert len(p['user_round_iters']) == len(ud)
    ulist, ilist = p['user_list'].reshape(-1, 4), p['item_list'].reshape(-1, 2)
    assert len(ulist) == n and len(ilist) == n
    # rec_slot: position -> slot = the interaction's index in the item order; the item side names interactions by slot, the user
    # list carries both (word 1 the position, word 3 the slot)
    rec_slot = p['rec_slot']
    assert sorted(rec_slot.tolist()) == list(range(n))
    pos_of_slot = np.empty(n, np.int64)
    pos_of_slot[rec_slot] = np.arange(n)
    np.testing.assert_array_equal(ilist[:, 1], np.arange(n))
    np.testing.assert_array_equal(ulist[:, 3], rec_slot[ulist[:, 1]])
    assert sorted(ulist[:, 1].tolist()) == list(range(n))
    assert (np.diff(users[ulist[:, 1]]) >= 0).all() and (np.diff(items[pos_of_slot]) >= 0).all()   # sorted by own row
    np.testing.assert_array_equal(ulist[:, 0], items[ulist[:, 1]])
    np.testing.assert_array_equal(ulist[:, 2].view(np.float32), y[ulist[:, 1]])
    np.testing.assert_array_equal(ilist[:, 0], users[pos_of_slot])
    # streamed rows: every untouched row exactly once, in launch 1 or launch 2, grouped by class
    sr = p['stream_rows']
    assert len(sr) == p['n_stream']
    su = np.sort(sr[(sr & planlib.ITEM_BIT) == 0])
    si = np.sort(sr[(sr & planlib.ITEM_BIT) != 0] & (planlib.ITEM_BIT - 1))
    np.testing.assert_array_equal(su, np.flatnonzero(np.bincount(users, minlength=U) == 0))
    np.testing.assert_array_equal(si, np.flatnonzero(np.bincount(items, minlength=I) == 0))
    ncls, cls = p['n_classes'], p['cls']
    assert 1 <= ncls <= 8
    ub = ib = sb = 0
    for c in range(ncls):
        assert cls[c, 0] == ub and cls[c, 4] == ib
        assert cls[c, 0] % p['user_rounds_per_task'] == 0 and cls[c, 4] % p['item_rounds_per_task'] == 0
        for desc, lo, cnt_ in ((ud, cls[c, 0], cls[c, 1]), (idd, cls[c, 4], cls[c, 5])):
            rows = desc[lo:lo + cnt_, :, 0].reshape(-1)
            rows = rows[rows >= 0]
            assert (planlib.row_class(rows, ncls) == c).all()
        ub += cls[c, 1]
        ib += cls[c, 5]
    assert ub == len(ud) and ib == len(idd)
    for launch in (0, 1):
        for c in range(ncls):
            lo, cnt_ = cls[c, 2 + 4 * launch], cls[c, 3 + 4 * launch]
            assert lo == sb
            seg = sr[lo:lo + cnt_]
            useg = seg[(seg & planlib.ITEM_BIT) == 0]
            assert (planlib.row_class(useg, ncls) == c).all()
            iseg = seg[(seg & planlib.ITEM_BIT) != 0] & (planlib.ITEM_BIT - 1)
            assert len(si) <= 512 or (planlib.row_class(iseg, ncls) == c).all()
            sb += cnt_
    assert sb == p['n_stream']
    if p['push']:   # push form: the contribution rows go to the same slots; item slices are ranges
        assert p['push_slot'] is rec_slot or np.array_equal(p['push_slot'], rec_slot)
    else:
        assert p['push_slot'] is None
    for side, (own, oth, R, lst, desc, inline, w) in enumerate((
            (users, items, U, ulist, ud, 2, 3), (items, users, I, ilist, idd, 0 if p['push'] else 3, 2))):
        d = desc.reshape(-1, 8)
        act = d[d[:, 0] >= 0]
        leaders = act[(act[:, 1] & 1) == 1]
        touched = np.flatnonzero(np.bincount(own, minlength=R) > 0)
        assert sorted(leaders[:, 0].tolist()) == touched.tolist()     # one job per touched row
        cnt = np.bincount(own, minlength=R)
        np.testing.assert_array_equal(leaders[:, 1] >> 9, cnt[leaders[:, 0]])
        seen = np.zeros(n, np.int32)
        for slot in act:
            row, meta = slot[0], slot[1]
            mode = (meta >> 6) & 7
            if mode == planlib.MODE_LIST:
                a, b = slot[2], slot[3]
                assert b - a > inline
                js = lst[a:b, 1] if side == 0 else pos_of_slot[lst[a:b, 1]]
                assert (own[js] == row).all()
                seen[js] += 1
            else:
                assert mode <= inline
                for q in range(mode):
                    f = slot[2 + q * w: 2 + (q + 1) * w]
                    pos = f[1] if side == 0 else pos_of_slot[f[1]]
                    assert own[pos] == row and oth[pos] == f[0]       # inline copy of the interaction
                    if side == 0:
                        assert np.int32(f[2]).view(np.float32) == y[pos]
                    seen[pos] += 1
        assert (seen == 1).all()                                      # each interaction in exactly one slice per side
        for ri, rd in enumerate(desc):                                # slot layout inside a round
            g = (rd[0, 1] >> 1) & 31
            assert g in (1, 2, 4, 8, 16, 32, 64) and g <= ng and (((rd[:, 1] >> 1) & 31) == g).all()
            longest = 0
            for s0 in range(0, ng, g):
                if rd[s0, 0] < 0:
                    assert (rd[s0:s0 + g, 0] < 0).all()
                    continue
                assert rd[s0, 1] & 1 and (rd[s0:s0 + g, 0] == rd[s0, 0]).all() and ((rd[s0 + 1:s0 + g, 1] & 1) == 0).all()
                for sl in rd[s0:s0 + g]:
                    mode = (sl[1] >> 6) & 7
                    longest = max(longest, sl[3] - sl[2] if mode == planlib.MODE_LIST else mode)
            if side == 0:
                assert p['user_round_iters'][ri] == longest
    return p


def test_plan_yahoo_like_batch():
    d = synth.yahoo_like()[:8192]
    p = check_plan(d[:, 0], d[:, 1], 15400, 1000)
    assert len(p['user_desc']) < 500 and len(p['item_desc']) < 400


def test_default_plan_fits_one_residency_wave():
    """both launches of a Yahoo-shaped step stay inside one residency wave of workgroups: two per CU in general, three
    for launch 1 when the caller names at most four environments -- which the default split then fills with stream tasks"""
    d = synth.yahoo_like()
    for k in (0, 3, 8, 12):
        b = d[k * 8192:(k + 1) * 8192]
        p = planlib.build_row_plan(b[:, 0], b[:, 1], b[:, 2], 15400, 1000)
        assert planlib.launch_workgroups(p, 0) <= 640 and planlib.launch_workgroups(p, 1) <= 640
        assert 0.0 <= p['stream_split'] <= 1.0
        q = check_plan(b[:, 0], b[:, 1], 15400, 1000, env_num=4)
        assert planlib.launch_workgroups(p, 0) < planlib.launch_workgroups(q, 0) <= planlib.RESIDENT_SMALL
        assert q['stream_split'] > 0.9 and planlib.launch_workgroups(q, 1) < planlib.launch_workgroups(p, 1)
        r = planlib.build_row_plan(b[:, 0], b[:, 1], b[:, 2], 15400, 1000, env_num=8)   # other instances: the balanced split
        assert r['stream_split'] == p['stream_split'] and planlib.launch_workgroups(r, 0) == planlib.launch_workgroups(p, 0)
        r, r4 = (planlib.build_row_plan(b[:, 0], b[:, 1], b[:, 2], 15400, 1000, factor_num=128, env_num=e) for e in (None, 4))
        assert r['stream_split'] == r4['stream_split']
    g = synth.interactions(6, 400000, 100000, 1 << 18, implicit=True, zipf=False)   # jobs alone exceed the residency
    q = planlib.build_row_plan(g[:, 0], g[:, 1], g[:, 2], 400000, 100000, env_num=4)
    assert q['stream_split'] == planlib.build_row_plan(g[:, 0], g[:, 1], g[:, 2], 400000, 100000)['stream_split']


@pytest.mark.parametrize('D', [30, 64, 128, 256])
@pytest.mark.parametrize('per_slice,rpt', [(1, 1), (2, 3), (4, 2), (64, 5)])
def test_plan_parameters(per_slice, rpt, D):
    rs = np.random.RandomState(per_slice)
    u, v = rs.randint(0, 37, 500), rs.randint(0, 5, 500)  # item rows with ~100 interactions each
    check_plan(u, v, 40, 7, D=D, per_slice=per_slice, item_per_slice=per_slice, rounds_per_task=rpt,
               item_rounds_per_task=rpt, stream_split=[0.0, 0.3, 1.0][rpt % 3])


def test_plan_single_class_is_the_plain_order():
    d = synth.yahoo_like()[:8192]
    p = check_plan(d[:, 0], d[:, 1], 15400, 1000, n_classes=1, stream_split=1.0)
    sr = p['stream_rows']
    np.testing.assert_array_equal(sr[(sr & planlib.ITEM_BIT) == 0], np.flatnonzero(np.bincount(d[:, 0], minlength=15400) == 0))


@pytest.mark.parametrize('push', [False, True])
def test_plan_push_and_pull_forms(push):
    d = synth.yahoo_like()[:8192]
    p = check_plan(d[:, 0], d[:, 1], 15400, 1000, push=push)
    assert p['push'] == push
    assert planlib.build_row_plan(d[:, 0], d[:, 1], d[:, 2], 15400, 1000)['push']          # Yahoo-class default: push
    m = synth.interactions(5, 6040, 3706, 65536, implicit=True, zipf=False)
    assert planlib.build_row_plan(m[:, 0], m[:, 1], m[:, 2], 6040, 3706, factor_num=128)['push']       # MovieLens-class: cache-resident, push
    g = synth.interactions(6, 400000, 100000, 1 << 20, implicit=True, zipf=False)
    assert not planlib.build_row_plan(g[:, 0], g[:, 1], g[:, 2], 400000, 100000)['push']            # cache-exceeding, rows dominate: pull


def test_plan_empty_and_single():
    check_plan(np.zeros(0, np.int64), np.zeros(0, np.int64), 5, 3)
    check_plan(np.array([2]), np.array([0]), 5, 3)
    check_plan(np.full(300, 1), np.full(300, 2), 4, 4)          # one hot row on both sides: 16 slices of 19
    check_plan(np.full(300, 1), np.full(300, 2), 4, 4, D=256)   # 4 slices of 75


def test_plan_large_shapes():
    d = synth.interactions(5, 6040, 3706, 65536, implicit=True, zipf=False)
    p = check_plan(d[:, 0], d[:, 1], 6040, 3706, D=128)
    assert p['lanes_per_group'] == 16 and 500 < planlib.launch_workgroups(p, 0) < 4000


def test_slice_length_of_mid_sized_launches_follows_the_makespan_estimate():
    """launches of a few residencies: the default slice length is the shortest one within 2 % of the best list-scheduling
    estimate (plan._launch1_makespan); a Yahoo step (jobs resident at once) keeps 2, a huge launch keeps the rounds rule"""
    cnt = np.array([40] * 8 + [3] * 100)
    few = planlib._launch1_makespan(cnt, 16, 16, 512, 8.0, 1.0)
    assert few == 8.0 + 10                                  # everything resident: the longest task (40 = 4 slices of 10)
    assert planlib._launch1_makespan(cnt, 16, 16, 1, 8.0, 1.0) > few   # one slot: the tasks queue
    m = synth.interactions(5, 6040, 3706, 65536, implicit=True, zipf=False)
    p = planlib.build_row_plan(m[:, 0], m[:, 1], m[:, 2], 6040, 3706, factor_num=128, env_num=8)
    ucnt = np.bincount(m[:, 0])
    est = {ps: planlib._launch1_makespan(ucnt, 16, ps, 512, 8.0, 2.1) for ps in (3, 4, 6, 8, 10, 12, 14, 16, 20)}
    assert est[p['per_slice']] <= 1.02 * min(est.values())
    assert all(est[ps] > 1.02 * min(est.values()) for ps in est if ps < p['per_slice'])
    d = synth.yahoo_like()[:8192]
    assert planlib.build_row_plan(d[:, 0], d[:, 1], d[:, 2], 15400, 1000, env_num=4)['per_slice'] == 2


def test_launch_1_rounds_are_in_co_residency_order_for_wide_rows():
    """rows of more than 64 floats (or more than 4 environments), one round per task: every XCD class's rounds come heaviest
    first in rows of 32 (the CUs of an XCD), every other row reversed -- tasks j and j + 32 of a class share a CU, so the
    heaviest round meets the lightest of the next row.  The Yahoo-class instance keeps the plain order."""
    from invpref_kdd_2022_amd import synth
    d = synth.interactions(3, 6040, 3706, 65536, implicit=True)
    pl = planlib.build_row_plan(d[:, 0], d[:, 1], d[:, 2], 6040, 3706, factor_num=128, env_num=8)
    assert pl['user_rounds_per_task'] == 1
    it, cls = np.asarray(pl['user_round_iters']), np.asarray(pl['cls'])
    for c in range(pl['n_classes']):
        its = it[cls[c, 0]:cls[c, 0] + cls[c, 1]]
        assert len(its) > 40
        rows = [its[r:r + 32] for r in range(0, len(its), 32)]
        assert all(np.diff(rows[0]) <= 0) and all(np.diff(rows[1]) >= 0)
        assert rows[0].min() >= rows[1].max()
    plain = planlib.build_row_plan(d[:, 0], d[:, 1], d[:, 2], 6040, 3706, factor_num=128, env_num=8, native=False)
    for k in ('user_desc', 'user_round_iters', 'item_desc'):
        assert np.array_equal(np.asarray(pl[k]), np.asarray(plain[k])), k
    y = synth.yahoo_like()[:8192]
    py = planlib.build_row_plan(y[:, 0], y[:, 1], y[:, 2], 15400, 1000, factor_num=64, env_num=4, _resolve_only=True)
    assert py['snake_user'] == 0
