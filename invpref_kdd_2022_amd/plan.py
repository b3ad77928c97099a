"""Row plans for the planned, atomic-free M-step (include/invpref_hip.h: InvPrefRowPlan).

Built once per training run: the reference's minibatches are static (utils.mini_batch,
utils.py:12-19: contiguous, unshuffled slices), so each minibatch's scatter pattern is inverted
ahead of time.  Host side, numpy, vectorised; the result is a few int32 device arrays per minibatch.

    job   = one table row + the minibatch's interactions that touch it, cut into 1/2/4/8/16 slices
    round = 16 group slots of a workgroup, filled with jobs of one slice count (heaviest first)
    task  = `rounds_per_task` consecutive rounds of one side, run by one workgroup
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

GROUPS = 16


class RowPlanStruct(C.Structure):
    """struct InvPrefRowPlan"""
    _fields_ = [('n_rounds', C.c_int32), ('n_item_rounds', C.c_int32), ('rounds_per_task', C.c_int32),
                ('n', C.c_int32), ('desc', C.c_void_p), ('other_user', C.c_void_p), ('pos_user', C.c_void_p),
                ('other_item', C.c_void_p), ('pos_item', C.c_void_p), ('n_hot', C.c_int32), ('reserved', C.c_int32),
                ('hot_rows', C.c_void_p), ('hot_count', C.c_void_p), ('item_hot_index', C.c_void_p),
                ('n_stream_user', C.c_int32), ('n_stream_item', C.c_int32), ('rows_per_stream_task', C.c_int32),
                ('dense_per_task', C.c_int32), ('stream_rows', C.c_void_p), ('batch_users', C.c_void_p),
                ('batch_items', C.c_void_p), ('n_classes', C.c_int32), ('reserved2', C.c_int32),
                ('cls', C.c_int32 * 64), ('item_hot_count', C.c_void_p)]

OPTIONAL_ARRAYS = ('item_hot_count',)   # may be NULL (meta offset -1): plan['item_hot_count'] = None

N_CLASSES = 8          # XCDs of an MI355X: blocks b and b + 8 of a launch share one (round-robin placement)
CLASS_SHIFT = 6        # rows are dealt to the classes in blocks of 64


def stream_rows_default(factor_num: int) -> int:
    """untouched rows per stream task: two iterations of a workgroup (2 rows per 16-lane group in flight)"""
    return 64


def row_class(rows: np.ndarray, n_classes: int) -> np.ndarray:
    return (np.asarray(rows) >> CLASS_SHIFT) % n_classes


def _side_rounds(own, oth, pos, y, n_rows: int, per_slice: int, pad_to: int, skip=None):
    """desc [n_rounds,16,8] int32 for one side.  own / oth / pos / y are in the side's sorted order.
    skip: boolean mask of rows that get no job (hot item rows)."""
    cnt = np.bincount(own, minlength=n_rows).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    ybits = np.ascontiguousarray(y, np.float32).view(np.int32)
    # slices needed at `per_slice` interactions each, rounded up to a power of two, at most 16;
    # rows hotter than 16*per_slice get longer slices instead
    need = np.maximum(1, -(-cnt // per_slice))
    slices = np.minimum(GROUPS, 1 << np.ceil(np.log2(need)).astype(np.int64))
    sl_len = np.maximum(-(-cnt // slices), 1)
    if skip is not None:
        slices = np.where(skip, 0, slices)
    descs = [np.zeros((0, GROUPS, 8), np.int32)]
    for g in (16, 8, 4, 2, 1):
        rows = np.flatnonzero(slices == g)
        if len(rows) == 0:
            continue
        rows = rows[np.argsort(-cnt[rows], kind='stable')]  # heaviest first
        per_round = GROUPS // g
        n_rounds = -(-len(rows) // per_round)
        d = np.zeros((n_rounds, GROUPS, 8), np.int32)
        d[:, :, 0] = -1
        d[:, :, 1] = g << 1            # idle slots still tell the round's slice count (sync decision)
        i = np.arange(len(rows))
        rnd, first = i // per_round, (i % per_round) * g
        for k in range(g):
            j0 = np.minimum(ptr[rows] + k * sl_len[rows], ptr[rows + 1])
            j1 = np.minimum(j0 + sl_len[rows], ptr[rows + 1])
            m = j1 - j0
            mode = np.where(m <= 2, m, 3)
            meta = (1 if k == 0 else 0) | (g << 1) | (mode << 6) | (cnt[rows] << 8)
            slot = d[rnd, first + k]
            slot[:, 0], slot[:, 1] = rows, meta
            j0c, j1c = np.minimum(j0, max(len(oth) - 1, 0)), np.minimum(j0 + 1, max(len(oth) - 1, 0))
            if len(oth):
                inl = mode <= 2
                slot[:, 2] = np.where(inl, np.where(m >= 1, oth[j0c], 0), j0)
                slot[:, 3] = np.where(inl, np.where(m >= 1, pos[j0c], 0), j1)
                slot[:, 4] = np.where(inl & (m >= 1), ybits[j0c], 0)
                slot[:, 5] = np.where(inl & (m >= 2), oth[j1c], 0)
                slot[:, 6] = np.where(inl & (m >= 2), pos[j1c], 0)
                slot[:, 7] = np.where(inl & (m >= 2), ybits[j1c], 0)
            d[rnd, first + k] = slot
        descs.append(d)
    d = np.concatenate(descs)
    pad = (-len(d)) % pad_to
    if pad:
        idle = np.zeros((pad, GROUPS, 8), np.int32)
        idle[:, :, 0] = -1
        idle[:, :, 1] = 1 << 1
        d = np.concatenate([d, idle])
    return d


RESIDENT_WORKGROUPS = 1024   # 256 CUs x 4 workgroups of mstep_rows_kernel (121 VGPRs): one residency wave


def build_row_plan(users: np.ndarray, items: np.ndarray, scores: np.ndarray, user_num: int, item_num: int,
                   per_slice: int | None = None, rounds_per_task: int | None = None,
                   hot_threshold: int | None = None, user_range=None, n_classes: int | None = None,
                   rows_per_stream_task: int | None = None) -> dict:
    """See _build_row_plan.  With hot_threshold left to the builder (and no INVPREF_PLAN_HOT), a Yahoo-class plan
    (slices of at most two interactions) whose launch would not fit one residency wave is rebuilt with a lower
    threshold -- more rows through the atomics, fewer item jobs: the step time is flat below the default and jumps by
    ~1.5 us the moment a second wave of workgroups is needed (measured, tools/ab2.sh: 19.4 us at > 6 ... > 10, 20.8 at > 12)."""
    kw = dict(per_slice=per_slice, rounds_per_task=rounds_per_task, user_range=user_range, n_classes=n_classes,
              rows_per_stream_task=rows_per_stream_task)
    plan = _build_row_plan(users, items, scores, user_num, item_num, hot_threshold=hot_threshold, **kw)
    if hot_threshold is not None or 'INVPREF_PLAN_HOT' in os.environ or plan['per_slice'] > 2:
        return plan
    for thr in (8, 6, 4, 2, 1):
        if plan_workgroups(plan) <= RESIDENT_WORKGROUPS or plan_workgroups(plan) > 2 * RESIDENT_WORKGROUPS:
            break   # fits -- or is a multi-wave launch anyway
        plan = _build_row_plan(users, items, scores, user_num, item_num, hot_threshold=thr, **kw)
    return plan


def plan_workgroups(plan: dict) -> int:
    """workgroups of the launch: the classes' task lists interleaved, padded to the longest"""
    ncls = int(plan.get('n_classes', 1))
    return ncls * max(class_tasks(plan, np.asarray(plan['cls']), c) for c in range(ncls))


def _build_row_plan(users: np.ndarray, items: np.ndarray, scores: np.ndarray, user_num: int, item_num: int,
                    per_slice: int | None = None, rounds_per_task: int | None = None,
                    hot_threshold: int | None = None, user_range=None, n_classes: int | None = None,
                    rows_per_stream_task: int | None = None) -> dict:
    """users/items/scores: ONE minibatch (or this rank's slice of it); scores as the fp32 labels.
    hot_threshold: item rows with MORE interactions than this get no job; their gradient is added with
    float atomics by the user-side jobs and completed by the finish kernel (-1: every item row).
    user_range: (lo, hi) user rows this rank is responsible for (user-sharded runs): untouched user rows
    outside it are not streamed (nobody reads their gradient or updates them here).
    n_classes: XCD-affine task order (default 8 = the XCDs of an MI355X; INVPREF_PLAN_CLASSES): table rows are dealt
    to the classes in blocks of 64 rows, and every job / streamed row of class c is run by a workgroup with
    blockIdx.x % n_classes == c -- under the round-robin placement of workgroups the SAME XCD step after step, so
    the row's parameters and Adam moments are still in that XCD's L2 when the next step reads them (measured,
    tools/xcd_probe.py: a pure streaming step takes 8.4 us with a stable assignment and 11.5 us when the assignment
    moves to another XCD every step).  Speed only: any order gives the same results."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    # Defaults follow the minibatch size (measured, tools/kbench.py): a Yahoo step (8 192 interactions) wants many
    # short tasks -- about one resident set of workgroups, so the step is one latency chain -- while a
    # MovieLens-sized one (65 536) is throughput-bound and wants fewer, longer ones.
    scale = max(1, len(users) // 8192)
    if per_slice is None:
        per_slice = int(os.environ.get('INVPREF_PLAN_PER_SLICE', str(min(32, 2 * scale))))
    if rounds_per_task is None:
        rounds_per_task = int(os.environ.get('INVPREF_PLAN_ROUNDS', '1'))
    if hot_threshold is None:
        # (Yahoo-class plans: 10 -- measured with the XCD-affine order, tools/ab2.sh: the per-class padding of the
        #  job rounds must not push the launch beyond one residency wave of 1 024 workgroups)
        hot_threshold = int(os.environ.get('INVPREF_PLAN_HOT', str(10 if per_slice <= 2 else 16 * per_slice)))
    n = len(users)
    if n and (cnt_max := max(np.bincount(users).max(), np.bincount(items).max())) >= (1 << 23):
        raise ValueError(f'a row with {cnt_max} interactions in one minibatch overflows the job descriptor')
    pu = np.argsort(users, kind='stable')
    pi = np.argsort(items, kind='stable')
    icnt = np.bincount(items, minlength=item_num)
    hot = (icnt > hot_threshold) & (icnt > 0)     # (an untouched row is streamed, never hot)
    hot_rows = np.flatnonzero(hot).astype(np.int32)
    hot_index = np.full(item_num, -1, np.int32)
    hot_index[hot_rows] = np.arange(len(hot_rows), dtype=np.int32)
    # item rounds first (they hold the longest jobs; padded to whole workgroups), user rounds after
    ucnt = np.bincount(users, minlength=user_num)
    untouched_u = ucnt == 0
    if user_range is not None:
        untouched_u[:user_range[0]] = False
        untouched_u[user_range[1]:] = False
    stream_u = np.flatnonzero(untouched_u).astype(np.int32)     # untouched rows: streamed, no job
    stream_i = np.flatnonzero(icnt == 0).astype(np.int32)
    if n_classes is None:
        n_classes = int(os.environ.get('INVPREF_PLAN_CLASSES', str(N_CLASSES)))
    n_classes = max(1, min(8, n_classes))
    ucls, icls = row_class(np.arange(user_num), n_classes), row_class(np.arange(item_num), n_classes)
    di_parts, du_parts, su_parts, si_parts = [], [], [], []
    cls = np.zeros((8, 8), np.int32)
    for c in range(n_classes):
        # per class: item rounds (padded to whole tasks), user rounds, untouched user rows, untouched item rows
        di_parts.append(_side_rounds(items[pi], users[pi], pi, scores[pi], item_num, per_slice, rounds_per_task,
                                     skip=hot | (icnt == 0) | (icls != c)))
        du_parts.append(_side_rounds(users[pu], items[pu], pu, scores[pu], user_num, per_slice, 1,
                                     skip=(ucnt == 0) | (ucls != c)))
        su_parts.append(stream_u[ucls[stream_u] == c])
        # (a few untouched item rows are not worth one tiny task per class: class 0 streams them all then)
        si_parts.append(stream_i[icls[stream_i] == c] if len(stream_i) > 8 * 64 else (stream_i if c == 0 else stream_i[:0]))
    n_item_rounds = sum(len(d) for d in di_parts)
    ib, ub, sb = 0, n_item_rounds, 0
    for c in range(n_classes):
        cls[c, 0], cls[c, 1] = ib, len(di_parts[c]); ib += len(di_parts[c])
        cls[c, 2], cls[c, 3] = ub, len(du_parts[c]); ub += len(du_parts[c])
    for c in range(n_classes):
        cls[c, 4], cls[c, 5] = sb, len(su_parts[c]); sb += len(su_parts[c])
    for c in range(n_classes):
        cls[c, 6], cls[c, 7] = sb, len(si_parts[c]); sb += len(si_parts[c])
    stream_u, stream_i = np.concatenate(su_parts), np.concatenate(si_parts)
    di, du = np.concatenate(di_parts), np.concatenate(du_parts)
    return dict(batch_users=users.astype(np.int32), batch_items=items.astype(np.int32),
                dense_per_task=int(os.environ.get('INVPREF_PLAN_DENSE', str(min(256, 32 * scale)))),
                stream_rows=np.concatenate([stream_u, stream_i]).astype(np.int32), n_stream_user=len(stream_u),
                n_stream_item=len(stream_i),
                rows_per_stream_task=int(os.environ.get('INVPREF_PLAN_STREAM_ROWS', str(rows_per_stream_task or 64))),
                n_classes=n_classes, cls=cls, per_slice=per_slice, hot_threshold=hot_threshold,
                n=n, n_item_rounds=len(di), rounds_per_task=rounds_per_task, desc=np.concatenate([di, du]),
                other_user=items[pu].astype(np.int32), pos_user=pu.astype(np.int32),
                other_item=users[pi].astype(np.int32), pos_item=pi.astype(np.int32),
                hot_rows=hot_rows, hot_count=icnt[hot_rows].astype(np.int32), item_hot_index=hot_index,
                item_hot_count=np.where(hot, icnt, 0).astype(np.int32))


@dataclass
class DevicePlan:
    struct: RowPlanStruct
    arrays: list  # keeps the device tensors alive
    n_tasks: int
    n_rounds: int
    buf: torch.Tensor = None    # the one int32 device buffer behind every array of the plan
    meta: torch.Tensor = None   # CPU int64[len(_fields_)]: struct fields in order, pointers as int32 offsets into buf


_META_LEN = sum(64 if name == 'cls' else 1 for name, _ in RowPlanStruct._fields_)


def struct_from_meta(buf: torch.Tensor, meta: torch.Tensor) -> RowPlanStruct:
    """The InvPrefRowPlan of a plan that travels as (device buffer, CPU meta tensor) -- the form in which
    ``torch.ops.invpref.train_step_planned_*`` take it."""
    if buf.dtype != torch.int32 or not buf.is_contiguous() or meta.dtype != torch.int64 or meta.is_cuda \
            or meta.numel() != _META_LEN:
        raise ValueError('row plan: int32 device buffer + CPU int64 meta tensor expected')
    vals = meta.tolist()
    base, n = buf.data_ptr(), buf.numel()
    args, i = [], 0
    for name, ty in RowPlanStruct._fields_:
        if name == 'cls':
            args.append((C.c_int32 * 64)(*vals[i:i + 64]))
            i += 64
            continue
        v = vals[i]
        i += 1
        if ty is C.c_void_p:
            if v == -1 and name in OPTIONAL_ARRAYS:
                v = None
            elif not 0 <= v <= n:
                raise ValueError('row plan: array offset outside the buffer')
            else:
                v = base + 4 * v
        args.append(v)
    return RowPlanStruct(*args)


def upload(plan: dict, device) -> DevicePlan:
    keys = ('desc', 'other_user', 'pos_user', 'other_item', 'pos_item', 'hot_rows', 'hot_count', 'item_hot_index',
            'stream_rows', 'batch_users', 'batch_items', 'item_hot_count')
    parts, ptrs, off = [], {}, 0
    if 'item_hot_count' not in plan:   # (plans built by hand, tests)
        cnt_of = np.zeros(len(plan['item_hot_index']), np.int32)
        cnt_of[np.asarray(plan['hot_rows'], np.int64)] = plan['hot_count']
        plan = dict(plan, item_hot_count=cnt_of)
    keys = tuple(k for k in keys if not (k in OPTIONAL_ARRAYS and plan[k] is None))
    for k in keys:  # every array starts on a 16-byte boundary of the one device buffer
        a = np.ascontiguousarray(plan[k], np.int32).reshape(-1)
        pad = (-len(a)) % 4
        parts.append(np.concatenate([a, np.zeros(pad, np.int32)]))
        ptrs[k] = off
        off += len(a) + pad
    buf = torch.from_numpy(np.concatenate(parts)).to(device)
    offs = dict(ptrs)
    ptrs = {k: buf.data_ptr() + 4 * o for k, o in ptrs.items()}
    for k in OPTIONAL_ARRAYS:
        if k not in ptrs:
            offs[k], ptrs[k] = -1, None
    nr, rpt = len(plan['desc']), plan['rounds_per_task']
    ncls = int(plan.get('n_classes', 1))
    cls = np.asarray(plan['cls'], np.int32) if 'cls' in plan else np.zeros((8, 8), np.int32)
    if 'cls' not in plan:   # the plain order as one class
        cls[0] = [0, plan['n_item_rounds'], plan['n_item_rounds'], nr - plan['n_item_rounds'], 0, plan['n_stream_user'],
                  plan['n_stream_user'], plan['n_stream_item']]
    st = RowPlanStruct(nr, plan['n_item_rounds'], rpt, plan['n'], ptrs['desc'], ptrs['other_user'],
                       ptrs['pos_user'], ptrs['other_item'], ptrs['pos_item'], len(plan['hot_rows']), 0,
                       ptrs['hot_rows'], ptrs['hot_count'], ptrs['item_hot_index'], plan['n_stream_user'],
                       plan['n_stream_item'], plan['rows_per_stream_task'], plan['dense_per_task'],
                       ptrs['stream_rows'], ptrs['batch_users'], ptrs['batch_items'], ncls, 0,
                       (C.c_int32 * 64)(*cls.reshape(-1).tolist()), ptrs['item_hot_count'])
    n_tasks = ncls * max(class_tasks(plan, cls, c) for c in range(ncls))
    meta = _meta_of(st, offs)
    return DevicePlan(st, [buf], n_tasks, nr, buf, torch.tensor(meta, dtype=torch.int64))


def class_tasks(plan: dict, cls: np.ndarray, c: int) -> int:
    """workgroups of class c: its share of the dense tasks, its item / user job tasks, its stream tasks"""
    ncls, rpt, spt = int(plan.get('n_classes', 1)), plan['rounds_per_task'], plan['rows_per_stream_task']
    nd = -(-plan['n'] // plan['dense_per_task'])
    return (max(0, -(-(nd - c) // ncls)) + -(-int(cls[c, 1]) // rpt) + -(-int(cls[c, 3]) // rpt)
            + -(-int(cls[c, 5]) // spt) + -(-int(cls[c, 7]) // spt))


def _meta_of(st: RowPlanStruct, offs: dict) -> list:
    """the struct as a flat list of integers (array fields expanded), pointers as int32 offsets into the buffer"""
    meta = []
    for name, ty in RowPlanStruct._fields_:
        v = getattr(st, name)
        if ty is C.c_void_p:
            meta.append(offs[name])
        elif hasattr(v, '__len__'):
            meta.extend(list(v))
        else:
            meta.append(v)
    return meta
