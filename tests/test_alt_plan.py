"""CPU: invariants of the alt plans (plan.build_alt_plan; include/invpref_hip.h: InvPrefAltPlan) that the alternating
one-launch-per-step kernels (csrc/step_alt.hpp) rely on."""
import numpy as np
import pytest

from invpref_kdd_2022_amd import plan as planlib, synth



def check_alt(cur, prev, side, U, I, **kw):
    p = planlib.build_alt_plan(cur, prev, side, U, I, **kw)
    NG = p['slots_per_round']
    assert NG == kw.get('slots', 16)
    own_num = U if side == 0 else I
    has_cur, has_prev = cur is not None, prev is not None
    assert p['has_cur'] == int(has_cur) and p['has_prev'] == int(has_prev) and p['side'] == side
    assert p['rounds_per_task'] == 1 and p['lanes_per_group'] == 16
    desc, pend = p['desc'], p['pend']
    assert desc.shape[1:] == (NG, 8) and pend.shape == (len(desc), NG, 4)
    n = p['n']
    if has_cur:
        u, i, y = [np.asarray(x) for x in cur]
        own, oth = (u, i) if side == 0 else (i, u)
        assert n == len(own)
        lst = p['list'].reshape(-1, 4)
        assert sorted(lst[:, 1].tolist()) == list(range(n))
        assert (np.diff(own[lst[:, 1]]) >= 0).all()
        np.testing.assert_array_equal(lst[:, 0], oth[lst[:, 1]])
        np.testing.assert_array_equal(lst[:, 2].view(np.float32), np.asarray(y, np.float32)[lst[:, 1]])
        # push slots: a permutation, slot order = partner-sorted order (stable)
        ps = p['push_slot']
        assert sorted(ps.tolist()) == list(range(n))
        order = np.argsort(ps, kind='stable')
        assert (np.diff(oth[order]) >= 0).all()
        cnt = np.bincount(own, minlength=own_num)
    else:
        assert n == 0
        cnt = np.zeros(own_num, np.int64)
    if has_prev:
        ownp = np.asarray(prev[0] if side == 0 else prev[1])
        cntp = np.bincount(ownp, minlength=own_num)
        assert p['n_prev'] == len(ownp)
    else:
        cntp = np.zeros(own_num, np.int64)
    ptrp = np.concatenate([[0], np.cumsum(cntp)])
    # ---- jobs: every interaction exactly once; slices of a row adjacent, leader first, the row's pending range split over them
    seen_pos, job_rows, pend_cover = [], [], np.zeros(int(ptrp[-1]), np.int64)
    for r in range(len(desc)):
        meta0 = desc[r, 0, 1]
        slices = ((meta0 >> 1) & 31) or 32            # (32 slices are stored as 0)
        assert slices >= 1 and (slices & (slices - 1)) == 0
        flag = bool(meta0 < 0)
        any_pend = False
        for s in range(NG):
            row, meta = int(desc[r, s, 0]), int(desc[r, s, 1])
            assert (((meta >> 1) & 31) or 32) == slices and bool(meta < 0) == flag   # round-uniform: the barriers depend on them
            if row < 0:
                assert (pend[r, s] == 0).all()
                continue
            leader, mode, c = meta & 1, (meta >> 6) & 7, (meta >> 9) & 0x3fffff
            assert leader == (1 if s % slices == 0 else 0)
            assert desc[r, s - s % slices, 0] == row           # slices of one row are adjacent, leader first
            assert c == cnt[row]
            a, b, cp, z = pend[r, s]
            assert cp == cntp[row] and z == 0 and ptrp[row] <= a <= b <= ptrp[row + 1]
            pend_cover[a:b] += 1
            any_pend |= cp > 0
            if leader:
                job_rows.append(row)
            if mode == 7:
                lo, hi = desc[r, s, 2], desc[r, s, 3]
                assert (own[lst[lo:hi, 1]] == row).all() and hi > lo
                seen_pos += lst[lo:hi, 1].tolist()
            else:
                w = desc[r, s, 2:].reshape(2, 3)
                for q in range(mode):
                    assert own[w[q, 1]] == row and oth[w[q, 1]] == w[q, 0] and w[q, 2].view(np.float32) == np.float32(y[w[q, 1]])
                    seen_pos.append(int(w[q, 1]))
        assert flag == (has_prev and any_pend)
    assert sorted(seen_pos) == list(range(n))
    assert len(set(job_rows)) == len(job_rows)
    # ---- stream: every other row of the side exactly once, with its whole pending range
    st = p['stream'].reshape(-1, 4)
    assert len(st) == p['n_stream']
    assert sorted(job_rows + st[:, 0].tolist()) == list(range(own_num))
    assert (cnt[st[:, 0]] == 0).all()
    np.testing.assert_array_equal(st[:, 1], ptrp[st[:, 0]])
    np.testing.assert_array_equal(st[:, 2], ptrp[st[:, 0] + 1])
    np.testing.assert_array_equal(st[:, 3], cntp[st[:, 0]])
    for a, b in st[:, 1:3]:
        pend_cover[a:b] += 1
    assert (pend_cover == 1).all()       # every pending contribution row is summed exactly once
    # ---- classes
    ncls, cls = p['n_classes'], p['cls']
    rb = sb = 0
    for c in range(ncls):
        assert cls[c, 0] == rb and cls[c, 2] == sb
        rows = desc[rb:rb + cls[c, 1], :, 0].reshape(-1)
        assert (planlib.row_class(rows[rows >= 0], ncls) == c).all()
        assert (planlib.row_class(st[sb:sb + cls[c, 3], 0], ncls) == c).all()
        rb += cls[c, 1]
        sb += cls[c, 3]
    assert rb == len(desc) and sb == len(st)
    assert p['n_tasks'] == (len(desc) if has_cur else 0)
    return p


def _mb(seed, n, U, I, zipf=True):
    d = synth.interactions(seed, U, I, n, implicit=True, zipf=zipf)
    return d[:, 0], d[:, 1], d[:, 2].astype(np.float32)


@pytest.mark.parametrize('side', [0, 1])
def test_alt_plan_yahoo_shape(side):
    U, I, B = 15400, 1000, 8192
    d = synth.yahoo_like()
    mb = lambda k: (d[k * B:(k + 1) * B, 0], d[k * B:(k + 1) * B, 1], d[k * B:(k + 1) * B, 2].astype(np.float32))  # noqa: E731
    check_alt(mb(1), None, side, U, I)
    check_alt(mb(1), mb(0)[:2], side, U, I, slots=32)
    p = check_alt(mb(1), mb(0)[:2], side, U, I, n_partials_prev=7)
    assert p['n_partials_prev'] == 7
    check_alt(None, mb(1)[:2], side, U, I)
    # the ragged last minibatch of an epoch and the wrap-around to the first
    last = (d[30 * B:, 0], d[30 * B:, 1], d[30 * B:, 2].astype(np.float32))
    check_alt(last, mb(29)[:2], side, U, I)
    check_alt(mb(0), last[:2], side, U, I)


@pytest.mark.parametrize('seed', range(12))
def test_alt_plan_random_small(seed):
    rs = np.random.RandomState(seed)
    U, I = int(rs.randint(1, 300)), int(rs.randint(1, 120))
    n, n_prev = int(rs.randint(1, 700)), int(rs.randint(1, 700))
    cur, prev = _mb(seed, n, U, I, zipf=bool(seed & 1)), _mb(100 + seed, n_prev, U, I, zipf=bool(seed & 2))
    for side in (0, 1):
        kw = dict(per_slice=int(rs.randint(1, 5)), n_classes=int(rs.choice([1, 2, 8])), slots=int(rs.choice([16, 32])))
        check_alt(cur, None, side, U, I, **kw)
        check_alt(cur, prev[:2], side, U, I, **kw)
        check_alt(None, prev[:2], side, U, I, **kw)


def test_alt_plan_hot_pending_rows_become_jobs():
    # a row without a current interaction but many pending rows gets a sliced job of its own (mode 0)
    U, I = 50, 10
    prev_u = np.arange(40) % U
    prev_i = np.zeros(40, np.int64)            # item 0: 40 pending rows on the item side
    cur = (np.arange(30) % U, 1 + np.arange(30) % (I - 1), np.ones(30, np.float32))   # item 0 untouched now
    p = check_alt(cur, (prev_u, prev_i), 1, U, I)
    rows = p['desc'][:, :, 0]
    sel = rows == 0
    assert sel.any() and (((p['desc'][:, :, 1][sel] >> 6) & 7) == 0).all()
    assert 0 not in p['stream'].reshape(-1, 4)[:, 0]


def test_alt_meta_roundtrip():
    import torch
    U, I = 40, 20
    cur, prev = _mb(3, 100, U, I), _mb(4, 90, U, I)
    p = planlib.build_alt_plan(cur, prev[:2], 0, U, I, n_partials_prev=5)
    dp = planlib.upload_alt(p, torch.device('cpu'))
    st = planlib.alt_struct_from_meta(dp.buf, dp.meta)
    assert (st.side, st.has_prev, st.has_cur, st.n, st.n_prev, st.n_rounds, st.n_partials_prev) == (0, 1, 1, 100, 90, len(p['desc']), 5)
    assert list(st.cls) == np.asarray(p['cls'], np.int32).reshape(-1).tolist()
    assert st.desc and st.pend and st.list and st.push_slot and st.stream
