"""CPU: invariants of the host-side row plan (plan.py) that the planned M-step kernel relies on."""
import numpy as np
import pytest

from invpref_kdd_2022_amd import plan as planlib, synth


def check_plan(users, items, U, I, **kw):
    y = (np.arange(len(users)) % 5).astype(np.float32)
    p = planlib.build_row_plan(users, items, y, U, I, **kw)
    desc, rpt, nir = p['desc'], p['rounds_per_task'], p['n_item_rounds']
    n = len(users)
    assert desc.shape[1:] == (16, 8) and nir % rpt == 0 and 0 <= nir <= len(desc)
    np.testing.assert_array_equal(p['batch_users'], users)
    np.testing.assert_array_equal(p['batch_items'], items)
    assert p['dense_per_task'] > 0
    su, si = p['stream_rows'][:p['n_stream_user']], p['stream_rows'][p['n_stream_user']:]
    assert len(si) == p['n_stream_item']
    np.testing.assert_array_equal(np.sort(su), np.flatnonzero(np.bincount(users, minlength=U) == 0))
    np.testing.assert_array_equal(np.sort(si), np.flatnonzero(np.bincount(items, minlength=I) == 0))
    # XCD-affine order: rounds and streamed rows are grouped by class = (row >> 6) % n_classes, cls[c] delimits them
    ncls, cls = p['n_classes'], p['cls']
    assert 1 <= ncls <= 8
    ib, ub, sb = 0, nir, 0
    for c in range(ncls):
        assert cls[c, 0] == ib and cls[c, 2] == ub and cls[c, 1] % rpt == 0
        for lo, cnt_ in ((cls[c, 0], cls[c, 1]), (cls[c, 2], cls[c, 3])):
            rows = desc[lo:lo + cnt_, :, 0].reshape(-1)
            rows = rows[rows >= 0]
            assert (planlib.row_class(rows, ncls) == c).all()
        ib += cls[c, 1]
        ub += cls[c, 3]
    assert ib == nir and ub == len(desc)
    for c in range(ncls):
        assert cls[c, 4] == sb
        seg = p['stream_rows'][sb:sb + cls[c, 5]]
        assert (planlib.row_class(seg, ncls) == c).all() and (np.diff(seg) > 0).all()
        sb += cls[c, 5]
    assert sb == p['n_stream_user'] and cls[0, 6] == sb and cls[:ncls, 7].sum() == p['n_stream_item']
    for c in range(ncls):   # untouched item rows: by class when there are many, all in class 0 otherwise
        seg = p['stream_rows'][cls[c, 6]:cls[c, 6] + cls[c, 7]]
        assert (np.diff(seg) > 0).all() and (len(si) <= 512 or (planlib.row_class(seg, ncls) == c).all())
    hot = set(p['hot_rows'].tolist())
    icnt = np.bincount(items, minlength=I)
    np.testing.assert_array_equal(p['hot_count'], icnt[p['hot_rows']])
    assert all(p['item_hot_index'][r] == i for i, r in enumerate(p['hot_rows'])) and (p['item_hot_index'] >= 0).sum() == len(hot)
    np.testing.assert_array_equal(p['item_hot_count'], np.where(p['item_hot_index'] >= 0, icnt, 0))
    assert (p['hot_count'] > 0).all()                                  # an untouched row is streamed, never hot
    for side, (own, oth, R, o_key, p_key, rounds) in enumerate((
            (users, items, U, 'other_user', 'pos_user', desc[nir:]),
            (items, users, I, 'other_item', 'pos_item', desc[:nir]))):
        pos = p[p_key]
        assert sorted(pos.tolist()) == list(range(n))                 # a permutation of the minibatch
        assert (np.diff(own[pos]) >= 0).all()                         # sorted by own row
        np.testing.assert_array_equal(p[o_key], oth[pos])
        d = rounds.reshape(-1, 8)
        act = d[d[:, 0] >= 0]
        leaders = act[(act[:, 1] & 1) == 1]
        jobless = (hot | set(si.tolist())) if side == 1 else set(su.tolist())
        assert sorted(leaders[:, 0].tolist()) == [r for r in range(R) if r not in jobless]   # one job per row
        cnt = np.bincount(own, minlength=R)
        np.testing.assert_array_equal(leaders[:, 1] >> 8, cnt[leaders[:, 0]])
        seen = np.zeros(n, np.int32)
        for row, meta, a, b, c, dd, e, f in act:
            mode = (meta >> 6) & 3
            if mode == 3:
                assert a < b
                js = pos[a:b]
                assert (own[js] == row).all() and b - a > 2
                seen[js] += 1
            else:
                for (o_, p_, y_) in ((a, b, c), (dd, e, f))[:mode]:
                    assert own[p_] == row and oth[p_] == o_           # inline copy of the interaction
                    assert np.int32(y_).view(np.float32) == y[p_]
                    seen[p_] += 1
        in_job = np.array([own[i] not in jobless for i in range(n)], bool) if n else np.zeros(0, bool)
        assert (seen == in_job.astype(np.int32)).all()                # each interaction of a job row in exactly one slice
        for rd in rounds:                                             # slot layout inside a round
            g = (rd[0, 1] >> 1) & 31
            assert g in (1, 2, 4, 8, 16) and (((rd[:, 1] >> 1) & 31) == g).all()
            for s0 in range(0, 16, g):
                if rd[s0, 0] < 0:
                    assert (rd[s0:s0 + g, 0] < 0).all()
                    continue
                assert rd[s0, 1] & 1 and (rd[s0:s0 + g, 0] == rd[s0, 0]).all() and ((rd[s0 + 1:s0 + g, 1] & 1) == 0).all()
    return p


def test_plan_yahoo_like_batch():
    d = synth.yahoo_like()[:8192]
    p = check_plan(d[:, 0], d[:, 1], 15400, 1000)
    assert len(p['desc']) < 4000


def test_default_plan_fits_one_residency_wave():
    """the builder lowers the hot-row threshold until the launch fits the 1 024 resident workgroups (a second wave of
    workgroups costs ~1.5 us per step); an explicit threshold is taken as given"""
    d = synth.yahoo_like()
    for k in (0, 3, 8, 12):
        b = d[k * 8192:(k + 1) * 8192]
        p = planlib.build_row_plan(b[:, 0], b[:, 1], b[:, 2], 15400, 1000)
        assert planlib.plan_workgroups(p) <= planlib.RESIDENT_WORKGROUPS and p['hot_threshold'] <= 10
    b = d[8 * 8192:9 * 8192]
    assert planlib.build_row_plan(b[:, 0], b[:, 1], b[:, 2], 15400, 1000, hot_threshold=16)['hot_threshold'] == 16


@pytest.mark.parametrize('hot', [-1, 0, 8, 10 ** 9])
@pytest.mark.parametrize('per_slice,rpt', [(1, 1), (2, 3), (4, 2), (64, 5)])
def test_plan_parameters(per_slice, rpt, hot):
    rs = np.random.RandomState(per_slice)
    u, v = rs.randint(0, 37, 500), rs.randint(0, 5, 500)  # item rows with ~100 interactions each
    check_plan(u, v, 40, 7, per_slice=per_slice, rounds_per_task=rpt, hot_threshold=hot)


def test_plan_single_class_is_the_plain_order():
    d = synth.yahoo_like()[:8192]
    p = check_plan(d[:, 0], d[:, 1], 15400, 1000, n_classes=1)
    np.testing.assert_array_equal(p['stream_rows'][:p['n_stream_user']], np.flatnonzero(np.bincount(d[:, 0], minlength=15400) == 0))


def test_plan_empty_and_single():
    check_plan(np.zeros(0, np.int64), np.zeros(0, np.int64), 5, 3)
    check_plan(np.array([2]), np.array([0]), 5, 3)
    check_plan(np.full(300, 1), np.full(300, 2), 4, 4, hot_threshold=10 ** 9)   # one hot row on both sides: 16 slices of 19
    check_plan(np.full(300, 1), np.full(300, 2), 4, 4, hot_threshold=16)
