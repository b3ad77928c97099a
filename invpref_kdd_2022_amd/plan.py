"""Row plans for the planned, atomic-free M-step (include/invpref_hip.h: InvPrefRowPlan).

Built once per training run: the reference's minibatches are static (utils.mini_batch,
utils.py:12-19: contiguous, unshuffled slices), so each minibatch's scatter pattern is inverted
ahead of time.  Host side, numpy, vectorised; the result is a few int32 device arrays per minibatch.

    group = the lanes that hold one embedding row (16 / 32 / 64 for factor_num <= 64 / 128 / 256);
            NG = 256 / lanes groups per workgroup
    job   = one table row + the minibatch's interactions that touch it, cut into 1 .. NG slices
    round = the NG group slots of a workgroup, filled with jobs of one slice count (heaviest first)
    task  = `*_rounds_per_task` consecutive rounds of one side, run by one workgroup

Launch 1 runs the user jobs (every interaction is evaluated there and leaves a record), launch 2 the item
jobs (they consume the records); rows the minibatch does not touch are streamed through the dense-Adam step
by either launch (`stream_split` balances the two).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

THREADS = 256
ITEM_BIT = 1 << 30     # stream_rows: a row of the item tables
MODE_LIST = 7
MAX_ROW_COUNT = 1 << 22
TARGET_WORKGROUPS = 1536
RESIDENT_SMALL = 768       # launch 1, rows <= 64 floats and <= 4 environments: three workgroups on each of 256 CUs


class RowPlanStruct(C.Structure):
    """struct InvPrefRowPlan"""
    _fields_ = [('n', C.c_int32), ('lanes_per_group', C.c_int32), ('n_user_rounds', C.c_int32),
                ('n_item_rounds', C.c_int32), ('user_rounds_per_task', C.c_int32), ('item_rounds_per_task', C.c_int32),
                ('user_desc', C.c_void_p), ('item_desc', C.c_void_p), ('user_round_iters', C.c_void_p),
                ('user_list', C.c_void_p), ('item_list', C.c_void_p), ('n_stream', C.c_int32),
                ('rows_per_stream_task', C.c_int32), ('stream_rows', C.c_void_p), ('n_classes', C.c_int32),
                ('rows_per_stream_task2', C.c_int32), ('cls', C.c_int32 * 64), ('push_slot', C.c_void_p),
                ('rec_slot', C.c_void_p)]


ARRAYS = ('user_desc', 'item_desc', 'user_round_iters', 'user_list', 'item_list', 'stream_rows', 'rec_slot', 'push_slot')
OPTIONAL_ARRAYS = ('push_slot',)   # may be absent (plan[k] is None; meta offset -1; NULL in the struct)
# rec_slot[position] = the interaction's index in item_list order ("slot"): launch 1 stores what it hands to launch 2 THERE --
# the record (pull form) or the two contribution rows (push form: push_slot is the same array) -- so that launch 2's item
# jobs read their slices of the workspace front to back; user_list's word 3 and item_list's word 1 carry the slot too.

N_CLASSES = 8          # XCDs of an MI355X: blocks b and b + 8 of a launch share one (round-robin placement)
CUS_PER_XCD = 32       # ... and inside an XCD tasks j, j + 32, j + 64 of a class share a CU (tools/probes/cu_map.hip)
CLASS_SHIFT = 6        # rows are dealt to the classes in blocks of 64


def lanes_of(factor_num: int) -> int:
    """lanes that hold one row (invpref_rows_lanes_per_group): 16 lanes x 1 float4 up to 64 floats, 16 x 2 up to 128,
    32 x 2 up to 256 (csrc/step_wide.hpp: an interaction shares its wave with three / one other)"""
    return 16 if factor_num <= 128 else 32


def padded_row(factor_num: int) -> int:
    """floats of a padded row (contribution rows, LDS images): 64 / 128 / 256"""
    return 64 if factor_num <= 64 else (128 if factor_num <= 128 else 256)


def stream_rows_default(factor_num: int) -> int:
    """untouched rows per stream task: ONE iteration of a workgroup (2 rows per group in flight); a second iteration would be a
    second serial round trip -- the loads of an iteration cannot be hoisted above the stores of the one before"""
    return 2 * (THREADS // lanes_of(factor_num))


def row_class(rows: np.ndarray, n_classes: int) -> np.ndarray:
    return (np.asarray(rows) >> CLASS_SHIFT) % n_classes


def _side_rounds(own, cols, n_rows: int, ng: int, per_slice: int, pad_to: int, inline: int, skip, snake: int = 0):
    """(desc [n_rounds, ng, 8] int32, iters [n_rounds]) for one side.  own: the side's row of every interaction in the
    side's sorted order; cols: the int32 columns of the sorted list (user side: partner, position, label bits; item
    side: partner, slot) -- `inline` interactions of a slice travel inside the descriptor, longer slices as a range
    of the list.  skip: rows that get no job."""
    w = len(cols)
    cnt = np.bincount(own, minlength=n_rows).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(cnt)])
    # slices needed at `per_slice` interactions each, rounded up to a power of two, at most ng;
    # rows hotter than ng * per_slice get longer slices instead
    need = np.maximum(1, -(-cnt // per_slice))
    slices = np.minimum(ng, 1 << np.ceil(np.log2(need)).astype(np.int64))
    sl_len = np.maximum(-(-cnt // slices), 1)
    slices = np.where(skip, 0, slices)
    descs, iters = [np.zeros((0, ng, 8), np.int32)], [np.zeros(0, np.int32)]
    g = ng
    while g >= 1:
        rows = np.flatnonzero(slices == g)
        if len(rows):
            rows = rows[np.argsort(-cnt[rows], kind='stable')]  # heaviest first
            per_round = ng // g
            n_rounds = -(-len(rows) // per_round)
            d = np.zeros((n_rounds, ng, 8), np.int32)
            d[:, :, 0] = -1
            d[:, :, 1] = (g & 31) << 1     # idle slots still tell the round's slice count (sync decision); 32 slices: 0
            it = np.zeros(n_rounds, np.int32)
            i = np.arange(len(rows))
            rnd, first = i // per_round, (i % per_round) * g
            for k in range(g):
                j0 = np.minimum(ptr[rows] + k * sl_len[rows], ptr[rows + 1])
                j1 = np.minimum(j0 + sl_len[rows], ptr[rows + 1])
                m = j1 - j0
                mode = np.where(m <= inline, m, MODE_LIST)
                meta = (1 if k == 0 else 0) | ((g & 31) << 1) | (mode << 6) | (cnt[rows] << 9)
                slot = d[rnd, first + k]
                slot[:, 0], slot[:, 1] = rows, meta
                if len(own):
                    inl = mode != MODE_LIST
                    slot[:, 2] = np.where(inl, 0, j0)
                    slot[:, 3] = np.where(inl, 0, j1)
                    for q in range(inline):
                        jq = np.minimum(j0 + q, len(own) - 1)
                        on = inl & (m > q)
                        for c in range(w):
                            slot[:, 2 + q * w + c] = np.where(on, cols[c][jq], slot[:, 2 + q * w + c])
                    # a list slice also carries its LEADING interactions in the descriptor's spare words (4 ..: one on the
                    # user side, two on the item side): their gathers start when the descriptor arrives, one round trip
                    # before the list's entries are there
                    for q in range((8 - 4) // w if inline > 0 else 0):
                        jq = np.minimum(j0 + q, len(own) - 1)
                        on = ~inl & (m > q)
                        for c in range(w):
                            slot[:, 4 + q * w + c] = np.where(on, cols[c][jq], slot[:, 4 + q * w + c])
                d[rnd, first + k] = slot
                np.maximum.at(it, rnd, m.astype(np.int32))
            descs.append(d)
            iters.append(it)
        g >>= 1
    d, it = np.concatenate(descs), np.concatenate(iters)
    if snake and len(it) > 1:
        # Which rounds share a CU: the dispatcher deals the workgroups of a launch to the CUs breadth-first, so tasks j, j + 32,
        # j + 64 ... of one XCD class run on the SAME CU while the launch fits one residency (tools/probes/cu_map.hip) and
        # compete for its SIMDs.  Heaviest rounds first, then every other row of `snake` (= the CUs of an XCD) reversed:
        # the heaviest round shares its CU with the lightest of the next row (measured at the MovieLens shape: launch 1 ends
        # at 40 us instead of 46, the step takes 63.7 us instead of 66.0).
        perm = np.argsort(-it, kind='stable')
        for r in range(1, -(-len(perm) // snake), 2):
            perm[r * snake:(r + 1) * snake] = perm[r * snake:(r + 1) * snake][::-1].copy()
        d, it = d[perm], it[perm]
    pad = (-len(d)) % pad_to
    if pad:
        idle = np.zeros((pad, ng, 8), np.int32)
        idle[:, :, 0] = -1
        idle[:, :, 1] = 1 << 1
        d = np.concatenate([d, idle])
        it = np.concatenate([it, np.zeros(pad, np.int32)])
    return d, it


def _launch1_makespan(cnt, ng: int, per_slice: int, resident: int, c0: float, ti: float) -> float:
    """Estimated end of launch 1's last user job (in units of ~us): the rounds _side_rounds would form at this slice
    length (heaviest first inside each slice count; the XCD classes are ignored), one task per round, c0 per task + ti
    per iteration of its longest slice, list-scheduled in launch order onto `resident` workgroup slots."""
    import heapq
    c = np.sort(cnt[cnt > 0])[::-1].astype(np.int64)
    need = np.maximum(1, -(-c // per_slice))
    slices = np.minimum(ng, 1 << np.ceil(np.log2(need)).astype(np.int64))
    length = -(-c // slices)
    durs = []
    g = ng
    while g >= 1:
        ln = length[slices == g]            # (sorted by count, so by slice length too)
        if len(ln):
            durs.extend((c0 + ti * ln[::ng // g]).tolist())   # a round's longest slice is its first row's
        g >>= 1
    slots = [0.0] * max(1, min(resident, len(durs)))
    heapq.heapify(slots)
    for d in durs:
        heapq.heappush(slots, heapq.heappop(slots) + d)
    return max(slots) if durs else 0.0


class PlanParamsStruct(C.Structure):
    """struct InvPrefPlanParams (include/invpref_plan.h)"""
    _fields_ = [('lanes_per_group', C.c_int32), ('per_slice', C.c_int32), ('item_per_slice', C.c_int32),
                ('rounds_per_task', C.c_int32), ('item_rounds_per_task', C.c_int32), ('n_classes', C.c_int32),
                ('rows_per_stream_task', C.c_int32), ('push', C.c_int32), ('user_lo', C.c_int32), ('user_hi', C.c_int32),
                ('fill_cap', C.c_int32), ('snake_user', C.c_int32), ('stream_split', C.c_double)]


class AltPlanParamsStruct(C.Structure):
    """struct InvPrefAltPlanParams (include/invpref_plan.h)"""
    _fields_ = [('side', C.c_int32), ('per_slice', C.c_int32), ('n_classes', C.c_int32), ('pend_job_min', C.c_int32),
                ('pend_per_slice', C.c_int32), ('slots', C.c_int32)]


_NATIVE = None
_NATIVE_ARRAYS = ('user_desc', 'item_desc', 'user_round_iters', 'user_list', 'item_list', 'stream_rows', 'rec_slot', 'cls')


def _native_lib():
    """the host library with the native plan builder (include/invpref_plan.h), or None when it is not built"""
    global _NATIVE
    if _NATIVE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libinvpref_ingest.so')
        try:
            L = C.CDLL(path)
            L.invpref_plan_build.restype = C.c_void_p
            L.invpref_plan_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                             C.POINTER(PlanParamsStruct)]
            L.invpref_plan_array.restype = C.c_int64
            L.invpref_plan_array.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.POINTER(C.c_int32))]
            L.invpref_plan_free.argtypes = [C.c_void_p]
            L.invpref_plan_row_counts.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
            L.invpref_plan_build_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                                  C.c_int64, C.POINTER(PlanParamsStruct), C.POINTER(C.c_void_p), C.c_int32]
            L.invpref_alt_plan_build.restype = C.c_void_p
            L.invpref_alt_plan_build.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int64, C.c_int64, C.POINTER(AltPlanParamsStruct)]
            L.invpref_alt_plan_build_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64,
                                                      C.POINTER(AltPlanParamsStruct), C.POINTER(C.c_void_p), C.c_int32]
            _NATIVE = L
        except (OSError, AttributeError):
            _NATIVE = False
    return _NATIVE or None


def _params_struct(r: dict) -> PlanParamsStruct:
    return PlanParamsStruct(r['lanes'], r['per_slice'], r['item_per_slice'], r['rounds_per_task'], r['item_rounds_per_task'],
                            r['n_classes'], r['rows_per_stream_task'], int(r['push']), r['user_lo'], r['user_hi'], r['fill_cap'],
                            r['snake_user'], r['stream_split'])


class _NativePlan:
    """owner of a native plan handle: released when the last array that views its memory is gone"""

    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        try:
            if self.h:
                self.L.invpref_plan_free(self.h)
                self.h = None
        except Exception:      # (interpreter shutdown: the library may be gone already)
            pass


def _plan_from_handle(L, h, r: dict) -> dict:
    """the plan dict build_row_plan returns, from a native handle: the large arrays are VIEWS of the handle's memory (no
    copy: a 2^24-interaction plan is 0.7 GB), kept alive by the arrays themselves; the small ones are copied"""
    out = {}
    owner = _NativePlan(L, h)
    for which, name in enumerate(_NATIVE_ARRAYS):
        ptr = C.POINTER(C.c_int32)()
        ln = L.invpref_plan_array(h, which, C.byref(ptr))
        if ln <= 0:
            out[name] = np.zeros(0, np.int32)
        elif ln < (1 << 16):
            out[name] = np.ctypeslib.as_array(ptr, shape=(ln,)).copy()
        else:
            buf = (C.c_int32 * ln).from_address(C.addressof(ptr.contents))
            buf._owner = owner
            out[name] = np.frombuffer(buf, dtype=np.int32)
    ng = THREADS // r['lanes']
    cls = out['cls'].reshape(8, 8)
    sb = int(cls[:, 3].sum() + cls[:, 7].sum())
    return dict(n=r['n'], lanes_per_group=r['lanes'], factor_num=r['factor_num'], per_slice=r['per_slice'],
                item_per_slice=r['item_per_slice'], user_rounds_per_task=r['rounds_per_task'],
                item_rounds_per_task=r['item_rounds_per_task'], user_desc=out['user_desc'].reshape(-1, ng, 8),
                item_desc=out['item_desc'].reshape(-1, ng, 8), user_round_iters=out['user_round_iters'],
                user_list=out['user_list'], item_list=out['item_list'],
                rec_slot=out['rec_slot'], push_slot=(out['rec_slot'] if r['push'] else None), push=r['push'],
                stream_rows=out['stream_rows'], n_stream=sb,
                rows_per_stream_task=r['rows_per_stream_task'], rows_per_stream_task2=r['rows_per_stream_task2'],
                stream_split=(float(cls[:, 3].sum()) / sb if (r['fill_cap'] and sb) else r['stream_split']),
                n_classes=r['n_classes'], cls=cls.copy())


def _native_build(users, items, scores, user_num, item_num, r: dict) -> dict:
    L = _native_lib()
    users, items = np.ascontiguousarray(users, np.int64), np.ascontiguousarray(items, np.int64)
    scores = np.ascontiguousarray(scores, np.float32)
    ps = _params_struct(r)
    h = L.invpref_plan_build(users.ctypes.data, items.ctypes.data, scores.ctypes.data, len(users), user_num, item_num,
                             C.byref(ps))
    if not h:
        raise ValueError('native plan builder: invalid arguments (row ids out of range?)')
    return _plan_from_handle(L, h, r)


def build_row_plans(users: np.ndarray, items: np.ndarray, scores: np.ndarray, offsets, user_num: int, item_num: int,
                    threads: int = 0, **kw) -> list:
    """The plans of many minibatches -- minibatch k = interactions [offsets[k], offsets[k + 1]) of the same arrays, what
    utils.mini_batch (utils.py:12-19) yields -- each with build_row_plan's parameters resolved for it; the arrays are built
    by the native builder on a thread pool (one call).  Falls back to build_row_plan per minibatch without the library."""
    users, items = np.ascontiguousarray(users, np.int64), np.ascontiguousarray(items, np.int64)
    scores = np.ascontiguousarray(scores, np.float32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    nb = len(offsets) - 1
    L = _native_lib()
    if L is None or kw.get('native') is False or os.environ.get('INVPREF_PLAN_NATIVE', '1') == '0':
        return [build_row_plan(users[offsets[k]:offsets[k + 1]], items[offsets[k]:offsets[k + 1]],
                               scores[offsets[k]:offsets[k + 1]], user_num, item_num, **kw) for k in range(nb)]
    kw.pop('native', None)
    res = [build_row_plan(users[offsets[k]:offsets[k + 1]], items[offsets[k]:offsets[k + 1]], scores[offsets[k]:offsets[k + 1]],
                          user_num, item_num, _resolve_only=True, **kw) for k in range(nb)]
    params = (PlanParamsStruct * nb)(*[_params_struct(r) for r in res])
    handles = (C.c_void_p * nb)()
    rc = L.invpref_plan_build_many(users.ctypes.data, items.ctypes.data, scores.ctypes.data, offsets.ctypes.data, nb, user_num,
                                   item_num, params, handles, int(threads))
    if rc != 0:
        for h in handles:
            if h:
                L.invpref_plan_free(h)
        raise ValueError('native plan builder: invalid arguments (row ids out of range?)')
    return [_plan_from_handle(L, handles[k], res[k]) for k in range(nb)]


def build_row_plan(users: np.ndarray, items: np.ndarray, scores: np.ndarray, user_num: int, item_num: int,
                   factor_num: int = 64, per_slice: int | None = None, item_per_slice: int | None = None,
                   rounds_per_task: int | None = None, item_rounds_per_task: int | None = None, user_range=None,
                   n_classes: int | None = None, rows_per_stream_task: int | None = None,
                   stream_split: float | None = None, push: bool | None = None, env_num: int | None = None,
                   native: bool | None = None, _resolve_only: bool = False) -> dict:
    """users/items/scores: ONE minibatch (or this rank's slice of it); scores as the fp32 labels.
    factor_num: decides the row layout the plan is built for (lanes_of).
    per_slice / item_per_slice: interactions one group walks for a user / an item row (more interactions: more slices).
    user_range: (lo, hi) user rows this rank is responsible for (user-sharded runs): untouched user rows
    outside it are not streamed (nobody reads their gradient or updates them here).
    n_classes: XCD-affine task order (default 8 = the XCDs of an MI355X; INVPREF_PLAN_CLASSES): table rows are dealt
    to the classes in blocks of 64 rows, and every job / streamed row of class c is run by a workgroup with
    blockIdx.x % n_classes == c -- under the round-robin placement of workgroups the SAME XCD step after step, so
    the row's parameters and Adam moments are still in that XCD's L2 when the next step reads them (measured,
    tools/xcd_probe.py: a pure streaming step takes 8.4 us with a stable assignment and 11.5 us when the assignment
    moves to another XCD every step).  Speed only: any order gives the same results.
    env_num: the model's environment count, if the caller knows it: rows of up to 64 floats with up to four environments
    run launch 1's smallest instance, which fits THREE workgroups per CU (52 KB of LDS, <= 168 registers) -- the default
    stream split then fills launch 1 up to that residency with untouched rows (measured at Yahoo shape: 18.9 us per step
    with all of them in launch 1 against 19.5-20.4 us at the balanced split; tools/ab.sh).
    stream_split: share of the untouched rows that launch 1 streams (the rest goes to launch 2); default: what
    balances the two launches' row traffic.
    push: the "push" form of the item side (InvPrefRowPlan.push_slot): launch 1 stores every interaction's two
    contribution rows to its item's gradient at the interaction's item-sorted slot and launch 2 sums contiguous rows
    instead of gathering partner rows + records.  One extra row write + read per interaction and table; default
    (INVPREF_PLAN_PUSH): on while the step's working set is cache-resident or that traffic is a small share of the
    step's bytes (the item jobs become one burst of contiguous loads -- no hot-row tail)."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32)
    lanes = lanes_of(factor_num)
    ng = THREADS // lanes
    n = len(users)
    # the latency-tuned kernel instance (rows of up to 64 floats, up to four environments): three workgroups per CU
    small = factor_num <= 64 and env_num is not None and env_num <= 4
    # Defaults: a launch wants about TARGET_WORKGROUPS job workgroups -- enough to fill 256 CUs a few times over, few
    # enough that the per-workgroup partial slabs of launch 1 stay a small share of the step's bytes.  A Yahoo step
    # (8 192 interactions) is one latency chain and gets the shortest slices (2 interactions); larger minibatches are
    # throughput-bound and get long slices (measured at 65 536 .. 1 048 576 interactions: 16 per slice beats 2 .. 8 -- a
    # slice walks its interactions with the next gathers in flight, while every further slot, round and workgroup pays
    # its own start-up round trips), then more rounds per task.
    target = int(os.environ.get('INVPREF_PLAN_TARGET_WGS', str(TARGET_WORKGROUPS)))
    _cnt = {}
    def counts(side):          # interactions per row of a side (one pass over the minibatch each, on first use)
        if side not in _cnt:
            rows, n_rows = (users, user_num) if side == 'u' else (items, item_num)
            L = _native_lib() if (n >= (1 << 20) and rows.flags.c_contiguous) else None
            if L is not None:       # (the same numbers as np.bincount, counted on threads)
                out = np.empty(n_rows, np.int64)
                if L.invpref_plan_row_counts(rows.ctypes.data, n, n_rows, out.ctypes.data) != 0:
                    raise ValueError('row ids out of range')
                _cnt[side] = out
            else:
                _cnt[side] = np.bincount(rows, minlength=n_rows)
        return _cnt[side]
    def rounds_for(cnt, ps):   # group slots / ng: the rounds a side needs at `ps` interactions per slice
        c = cnt[cnt > 0]
        need = np.maximum(1, -(-c // ps))
        return int(np.minimum(ng, 1 << np.ceil(np.log2(need)).astype(np.int64)).sum()) // ng + 1
    if per_slice is None and 'INVPREF_PLAN_PER_SLICE' in os.environ:
        per_slice = int(os.environ['INVPREF_PLAN_PER_SLICE'])
    if per_slice is None:
        # shortest slices while launch 1's jobs are resident at once (one latency chain: a Yahoo step); otherwise the
        # shortest slice that costs no more than 5 % more rounds than 16 per slice does (measured at 32 768 Yahoo-shaped
        # interactions: 6 per slice 33.5 us against 37.5 at 2 and 37.2 at 16; at 250 154: 16 per slice 118 us, 8: 135)
        ucnt0 = counts('u')
        resident = RESIDENT_SMALL if small else 512
        if rounds_for(ucnt0, 2) <= resident:
            per_slice = 2
        else:
            r16 = rounds_for(ucnt0, 16)
            per_slice = next((ps for ps in (3, 4, 6, 8, 12) if rounds_for(ucnt0, ps) <= 1.05 * r16), 16)
            # rows with dozens of interactions each (2^24 interactions over 400 000 users: 42 per row): 24 per slice halves
            # the slices again (measured at that size, D = 64 / 128 / 256: +3 % each over 16; 20, 28 and 32 are behind)
            if per_slice == 16 and rounds_for(ucnt0, 24) <= 0.75 * r16:
                per_slice = 24
                # Round 6, with the records at the interaction's slot (launch 2 reads them front to back, launch 1 no longer pays
                # a random slot read per interaction): the shortest slice that leaves (nearly) every row in ONE slice -- no slice
                # meet, half the descriptors -- wins on launches of many residencies.  2^24 interactions, 42 per user row, same
                # box: D = 64 4 173 us at 24 per slice, 3 989 at 48, 3 909-3 919 at 56 / 64, 3 887 at 96 (with 4 rounds per task);
                # D = 128 9 004 / 8 506 / 8 371-8 456 / 8 475; D = 256 22.6 ms at 24, 21.8 at 48 (tools/ab_plan24b.sh).
                if r16 > 6 * resident:
                    fewest = rounds_for(ucnt0, 96)
                    per_slice = next((ps for ps in (24, 32, 40, 48, 56, 64) if rounds_for(ucnt0, ps) <= 1.05 * fewest), 96)
            if r16 <= 6 * resident and os.environ.get('INVPREF_PLAN_SIMULATE', '1') == '1':
                # a launch of a few residencies: where its last workgroup ends depends on how the task lengths pack
                # into the resident slots.  Estimate that for each slice length (list scheduling of the tasks in launch
                # order) and take the shortest slice within 2 % of the best (measured: MovieLens-shaped steps 85.5 us at 12
                # per slice -- two full residencies -- against 87.9 at 16 and 97 at 10; the model ranks them the same way)
                ti = 1.0 if (env_num is None or env_num <= 4) else (2.1 if env_num <= 8 else 3.7)
                est = {ps: _launch1_makespan(ucnt0, ng, ps, resident, 8.0, ti) for ps in (3, 4, 6, 8, 10, 12, 14, 16, 20)}
                best = min(est.values())
                per_slice = min(ps for ps, t in est.items() if t <= 1.02 * best)
    if push is None:
        env = os.environ.get('INVPREF_PLAN_PUSH')
        if env is not None:
            push = env == '1'
        else:
            # measured (tools/ab.sh, push vs pull): push wins wherever the step's working set sits in the 256 MiB Infinity
            # Cache (Yahoo 19.4 vs 20.4 us, Yahoo with B = N 134 vs 178, MovieLens 100 vs 111) and wherever the contribution
            # rows are a small share of the bytes (a 1/8 MIND minibatch 457 vs 534); it loses on cache-exceeding launches
            # whose rows dominate (2^20 interactions over 500 000 rows: 768 vs 732)
            p_floats = 2 * (user_num + item_num) * factor_num
            contrib = n * padded_row(factor_num) * 16               # two padded rows written and read per interaction
            total = n * (32 + 16 * factor_num) + 24 * p_floats       # algorithmic bytes of the step
            resident = 20 * p_floats + contrib // 2 <= 200e6       # p, p', m, v, g-free: five flat buffers + the rows
            push = n > 0 and (resident or contrib <= 0.2 * total)
            # (Round 6, records at the interaction's slot: on UNIFORM ids at the Yahoo tables the pull form overtakes push from
            #  65 536 interactions on -- B = N 82.5 vs 87.3 us -- but on skewed ids, Zipf or the Yahoo-like popularity, push stays
            #  12-21 % ahead at every size (B = N 100-103 vs 118-125 us): its hot items sum contiguous rows.  The rule stands;
            #  tools/ab_pushpull.sh, ab_pushpull2.sh)
    if lanes == 32:
        # rows of more than 128 floats: embed_env's outer product runs in launch 2's item jobs, which need the partner
        # rows and the records -- the pull form (csrc/step_wide.hpp, EVL2)
        push = False
    if item_per_slice is None:
        # (push form: a slice's contribution rows are contiguous and leave in one burst of up to four: measured 4 > 2, 3)
        # (wide rows, push form, minibatches of several residencies: longer item slices, fewer item workgroups -- MovieLens-
        #  shaped steps 66.7 -> 64.9 us at 16 per slice instead of 4)
        floor = (16 if (factor_num > 64 and n > 16384) else 4) if push else 2
        item_per_slice = int(os.environ.get('INVPREF_PLAN_ITEM_PER_SLICE',
                                            str(min(32, max(floor, -(-n // (ng * 2 * target)))))))

    if rounds_per_task is None:
        # (at most 8 rounds per task for rows on 16 lanes -- measured on cache-exceeding launches of 2^20 .. 2^24 interactions:
        #  D = 64 +4 %, D = 128 +2 .. +5 % over 16 -- and 16 for rows on 32 lanes, whose tasks stage 32 KB of tables first: -3 % at 8)
        # (long slices, round 6: about 200 interactions per group and task -- 4 rounds at 48-56 per slice: +2-3 % over 8 at 2^24)
        #  (rows on 32 lanes, 56 per slice at 2^24: 4 rounds 20.6 ms, 8 20.8, 16 21.1, 2 21.0 -- tools/ab_plan24c.sh)
        cap = (16 if per_slice < 40 else 4) if lanes == 32 else min(8, max(2, 224 // max(per_slice, 1)))
        # (rows on 32 lanes: about 800 tasks -- not quite two residencies of 512 -- instead of 1 536: measured at MIND's tables
        #  with minibatches of 32 768 .. 262 144, best at 4 / 6 / 6 / 4-8 rounds per task: +5 % at the rank share of eight GPUs)
        tgt = target * 800 // TARGET_WORKGROUPS if lanes == 32 else target
        rounds_per_task = int(os.environ.get('INVPREF_PLAN_ROUNDS', '0')) or \
            min(cap, max(1, round(rounds_for(counts('u'), per_slice) / tgt)))
    if item_rounds_per_task is None:
        # (rows of up to 1 KB, pull form: an item task first stages two [E, D] tables of up to 16 KB each, holds only eight
        #  rows and leaves a 16 KB partial slab of embed_env's gradient -- fewer, longer tasks: MIND-shaped steps 898 -> 769 us
        #  at 8 rounds per task instead of 1)
        few = lanes == 32
        item_rounds_per_task = int(os.environ.get('INVPREF_PLAN_ITEM_ROUNDS', '0')) or \
            min(16, max(1, round(rounds_for(counts('i'), item_per_slice) / (target // 2 if few else 4 * target))))
    if rows_per_stream_task is None:
        rows_per_stream_task = int(os.environ.get('INVPREF_PLAN_STREAM_ROWS', str(stream_rows_default(factor_num))))
    rows_per_stream_task2 = int(os.environ.get('INVPREF_PLAN_STREAM_ROWS2', str(rows_per_stream_task)))  # one iteration of a workgroup
    ucnt, icnt = counts('u'), counts('i')
    if n and max(ucnt.max(), icnt.max()) >= MAX_ROW_COUNT:
        raise ValueError(f'a row with {max(ucnt.max(), icnt.max())} interactions in one minibatch overflows the job descriptor')
    if max(user_num, item_num) >= ITEM_BIT:
        raise ValueError('too many rows')
    untouched_u = ucnt == 0
    if user_range is not None:
        untouched_u[:user_range[0]] = False
        untouched_u[user_range[1]:] = False
    stream_u = np.flatnonzero(untouched_u).astype(np.int32)     # untouched rows: streamed, no job
    stream_i = np.flatnonzero(icnt == 0).astype(np.int32)
    if n_classes is None:
        n_classes = int(os.environ.get('INVPREF_PLAN_CLASSES', str(N_CLASSES)))
    n_classes = max(1, min(8, n_classes))
    if stream_split is None and 'INVPREF_PLAN_STREAM_SPLIT' in os.environ:
        stream_split = float(os.environ['INVPREF_PLAN_STREAM_SPLIT'])
    n_stream = len(stream_u) + len(stream_i)
    fill_cap = 0   # launch 1 residency (workgroups) the default split fills with stream tasks; 0 = plain balance
    if stream_split is None:
        if small and os.environ.get('INVPREF_PLAN_FILL', '1') == '1':
            fill_cap = RESIDENT_SMALL
        # row moves (one row read or written in both tables of a side = 2): a touched or streamed row costs 12 (p, m, v
        # in, p', m', v' out), an interaction 4 gathered rows per launch; launch 1 also evaluates, hence the bias
        tu, ti = int((ucnt > 0).sum()), int((icnt > 0).sum())
        bias = float(os.environ.get('INVPREF_PLAN_EVAL_COST', '3.0'))
        s1 = (12.0 * ti - 12.0 * tu - bias * n + 12.0 * n_stream) / 24.0
        stream_split = min(1.0, max(0.0, s1 / n_stream)) if n_stream else 0.0
    # wide-row instances, one round per task: launch 1's rounds in co-residency order (_side_rounds: snake)
    snake_user = int(os.environ.get('INVPREF_PLAN_SNAKE', str(CUS_PER_XCD if (not small and rounds_per_task == 1) else 0)))
    # ---- every parameter is resolved: the arrays themselves come from the native builder (csrc/invpref_plan.cpp, the same
    # arrays byte for byte: tests/test_plan_native.py) unless INVPREF_PLAN_NATIVE=0 or `_resolve_only`
    resolved = dict(lanes=lanes, per_slice=int(per_slice), item_per_slice=int(item_per_slice),
                    rounds_per_task=int(rounds_per_task), item_rounds_per_task=int(item_rounds_per_task),
                    n_classes=int(n_classes), rows_per_stream_task=int(rows_per_stream_task),
                    rows_per_stream_task2=int(rows_per_stream_task2), stream_split=float(stream_split), fill_cap=int(fill_cap),
                    snake_user=int(snake_user),
                    push=bool(push), user_lo=0 if user_range is None else int(user_range[0]),
                    user_hi=int(user_num) if user_range is None else int(user_range[1]), factor_num=factor_num, n=n)
    if _resolve_only:
        return resolved
    if native is None:
        native = os.environ.get('INVPREF_PLAN_NATIVE', '1') != '0'
    if native and _native_lib() is not None:
        return _native_build(users, items, scores, user_num, item_num, resolved)
    pu = np.argsort(users, kind='stable')
    pi = np.argsort(items, kind='stable')
    ybits = scores.view(np.int32)
    slot = np.empty(n, np.int32)            # position -> index in the item order
    slot[pi] = np.arange(n, dtype=np.int32)
    ucols = (items[pu].astype(np.int32), pu.astype(np.int32), ybits[pu])
    icols = (users[pi].astype(np.int32), np.arange(n, dtype=np.int32))
    ucls, icls = row_class(np.arange(user_num), n_classes), row_class(np.arange(item_num), n_classes)
    du_parts, it_parts, di_parts, s1_parts, s2_parts = [], [], [], [], []
    cls = np.zeros((8, 8), np.int32)
    for c in range(n_classes):
        d, it = _side_rounds(users[pu], ucols, user_num, ng, per_slice, rounds_per_task, 2,
                             skip=(ucnt == 0) | (ucls != c), snake=snake_user)
        du_parts.append(d)
        it_parts.append(it)
        d, _ = _side_rounds(items[pi], icols, item_num, ng, item_per_slice, item_rounds_per_task, 0 if push else 3,
                            skip=(icnt == 0) | (icls != c))
        di_parts.append(d)
        # (a few untouched item rows are not worth one tiny task per class: class 0 streams them all then)
        si = stream_i[icls[stream_i] == c] if len(stream_i) > 8 * 64 else (stream_i if c == 0 else stream_i[:0])
        rows = np.concatenate([stream_u[ucls[stream_u] == c], si | ITEM_BIT]).astype(np.int32)
        k = int(round(stream_split * len(rows)))
        if fill_cap:   # (per class: the grid is n_classes x the longest class)
            room = fill_cap // n_classes - -(-len(du_parts[-1]) // rounds_per_task)
            k = min(len(rows), max(k, room * rows_per_stream_task))
        # inside each launch's share: item rows first, user rows last
        for part, li in ((rows[:k], 0), (rows[k:], 1)):
            is_user = (part & ITEM_BIT) == 0
            part = np.concatenate([part[~is_user], part[is_user]])
            (s1_parts if li == 0 else s2_parts).append(part)
    ub = ib = sb = 0
    for c in range(n_classes):
        cls[c, 0], cls[c, 1] = ub, len(du_parts[c]); ub += len(du_parts[c])
        cls[c, 4], cls[c, 5] = ib, len(di_parts[c]); ib += len(di_parts[c])
    for c in range(n_classes):
        cls[c, 2], cls[c, 3] = sb, len(s1_parts[c]); sb += len(s1_parts[c])
    for c in range(n_classes):
        cls[c, 6], cls[c, 7] = sb, len(s2_parts[c]); sb += len(s2_parts[c])
    return dict(n=n, lanes_per_group=lanes, factor_num=factor_num, per_slice=per_slice, item_per_slice=item_per_slice,
                user_rounds_per_task=rounds_per_task, item_rounds_per_task=item_rounds_per_task,
                user_desc=np.concatenate(du_parts), item_desc=np.concatenate(di_parts),
                user_round_iters=np.concatenate(it_parts),
                user_list=np.stack([ucols[0], ucols[1], ucols[2], slot[pu]], axis=1).reshape(-1),
                item_list=np.stack([icols[0], icols[1]], axis=1).reshape(-1),
                rec_slot=slot, push_slot=(slot if push else None), push=bool(push),
                stream_rows=np.concatenate(s1_parts + s2_parts).astype(np.int32), n_stream=sb,
                rows_per_stream_task=rows_per_stream_task, rows_per_stream_task2=rows_per_stream_task2,
                stream_split=(sum(len(x) for x in s1_parts) / sb if (fill_cap and sb) else stream_split),
                n_classes=n_classes, cls=cls)


def launch_workgroups(plan: dict, launch: int) -> int:
    """task workgroups of launch 0 / 1: the classes' task lists interleaved, padded to the longest"""
    ncls, cls = int(plan['n_classes']), np.asarray(plan['cls'])
    rpt = plan['user_rounds_per_task'] if launch == 0 else plan['item_rounds_per_task']
    spt = plan['rows_per_stream_task'] if launch == 0 else plan.get('rows_per_stream_task2', plan['rows_per_stream_task'])
    return ncls * max(-(-int(cls[c, 4 * launch + 1]) // rpt) + -(-int(cls[c, 4 * launch + 3]) // spt)
                      for c in range(ncls))


def plan_workgroups(plan: dict) -> int:
    """workgroups of the larger of the two launches"""
    return max(launch_workgroups(plan, 0), launch_workgroups(plan, 1))


@dataclass
class DevicePlan:
    struct: RowPlanStruct
    arrays: list  # keeps the device tensors alive
    n_tasks: int
    n_rounds: int
    buf: torch.Tensor = None    # the one int32 device buffer behind every array of the plan
    meta: torch.Tensor = None   # CPU int64[len(_fields_)]: struct fields in order, pointers as int32 offsets into buf


_ARRAY_FIELDS = {'cls': 64}
_META_LEN = sum(_ARRAY_FIELDS.get(name, 1) for name, _ in RowPlanStruct._fields_)


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
        if name in _ARRAY_FIELDS:
            k = _ARRAY_FIELDS[name]
            args.append((C.c_int32 * k)(*vals[i:i + k]))
            i += k
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
    parts, offs, off = [], {}, 0
    for k in ARRAYS:  # every array starts on a 16-byte boundary of the one device buffer
        if k in OPTIONAL_ARRAYS and plan.get(k) is None:
            offs[k] = -1
            continue
        if k == 'push_slot' and plan[k] is plan['rec_slot']:   # (the same array: one copy on the device)
            offs[k] = offs['rec_slot']
            continue
        a = np.ascontiguousarray(plan[k], np.int32).reshape(-1)
        parts.append((off, a))
        offs[k] = off
        off += len(a) + (-len(a)) % 4
    total = off + 4                        # (never an empty buffer)
    if total < (1 << 18):                  # a small plan: one host buffer, one copy
        host = np.zeros(total, np.int32)
        for o, a in parts:
            host[o:o + len(a)] = a
        buf = torch.from_numpy(host).to(device)
    else:                                  # a large one (up to GBs): every array straight from where the builder left it
        buf = torch.zeros(total, dtype=torch.int32, device=device)
        for o, a in parts:
            if len(a):
                buf[o:o + len(a)].copy_(torch.from_numpy(a))
    ptrs = {k: (buf.data_ptr() + 4 * o if o >= 0 else None) for k, o in offs.items()}
    cls = np.asarray(plan['cls'], np.int32)
    st = RowPlanStruct(plan['n'], plan['lanes_per_group'], len(plan['user_desc']), len(plan['item_desc']),
                       plan['user_rounds_per_task'], plan['item_rounds_per_task'], ptrs['user_desc'], ptrs['item_desc'],
                       ptrs['user_round_iters'], ptrs['user_list'], ptrs['item_list'], plan['n_stream'],
                       plan['rows_per_stream_task'], ptrs['stream_rows'], int(plan['n_classes']),
                       int(plan.get('rows_per_stream_task2', 0)),
                       (C.c_int32 * 64)(*cls.reshape(-1).tolist()), ptrs['push_slot'], ptrs['rec_slot'])
    meta = _meta_of(st, offs)
    return DevicePlan(st, [buf], plan_workgroups(plan), len(plan['user_desc']) + len(plan['item_desc']), buf,
                      torch.tensor(meta, dtype=torch.int64))


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


# ---------------------------------------------------------------------------------------------------------------------
# Alt plans: ONE launch per optimiser step, the evaluating side alternates (include/invpref_hip.h: InvPrefAltPlan,
# csrc/step_alt.hpp).  A plan describes one launch: the CURRENT minibatch seen from side S (S's rows own the jobs, the
# other side's rows are the partners) + the contribution rows the PREVIOUS launch pushed for S's rows.
class AltPlanStruct(C.Structure):
    """struct InvPrefAltPlan"""
    _fields_ = [('side', C.c_int32), ('has_prev', C.c_int32), ('has_cur', C.c_int32), ('n', C.c_int32), ('n_prev', C.c_int32),
                ('lanes_per_group', C.c_int32), ('slots_per_round', C.c_int32), ('n_rounds', C.c_int32),
                ('rounds_per_task', C.c_int32), ('desc', C.c_void_p), ('pend', C.c_void_p), ('list', C.c_void_p), ('push_slot', C.c_void_p),
                ('n_stream', C.c_int32), ('rows_per_stream_task', C.c_int32), ('stream', C.c_void_p),
                ('n_classes', C.c_int32), ('cls', C.c_int32 * 32), ('n_partials_prev', C.c_int32)]


ALT_ARRAYS = ('desc', 'pend', 'list', 'push_slot', 'stream')
ALT_PEND_JOB_MIN = 9       # rows WITHOUT a current interaction and at least this many pending rows get a (sliced) job
ALT_PEND_PER_SLICE = 4


def alt_supported(factor_num: int, env_num: int) -> bool:
    return factor_num <= 64 and env_num <= 4


def build_alt_plan(cur, prev, side: int, user_num: int, item_num: int, factor_num: int = 64, per_slice: int | None = None,
                   rounds_per_task: int = 1, n_classes: int | None = None, rows_per_stream_task: int | None = None,
                   n_partials_prev: int = 0, native: bool | None = None, slots: int = 16) -> dict:
    """cur: (users, items, scores) of the minibatch this launch evaluates, or None (a flush launch); prev: (users, items)
    of the minibatch the previous launch evaluated (from the OTHER side), or None (first launch of a run); side: 0 = the
    user tables evaluate / are updated, 1 = the item tables.  n_partials_prev: job tasks of the previous launch (its
    plan's 'n_tasks')."""
    lanes = 16
    if slots not in (16, 32):
        raise ValueError('slots per round: 16 (workgroups of 256 threads) or 32 (512 threads)')
    ng = slots
    own_num = user_num if side == 0 else item_num
    if per_slice is None:
        per_slice = int(os.environ.get('INVPREF_ALT_PER_SLICE_I' if side else 'INVPREF_ALT_PER_SLICE_U', '2'))
    if n_classes is None:
        n_classes = int(os.environ.get('INVPREF_PLAN_CLASSES', str(N_CLASSES)))
    n_classes = max(1, min(8, n_classes))
    if rows_per_stream_task is None:
        rows_per_stream_task = int(os.environ.get('INVPREF_PLAN_STREAM_ROWS', str(stream_rows_default(factor_num))))
    has_cur, has_prev = cur is not None, prev is not None
    if native is None:
        native = os.environ.get('INVPREF_PLAN_NATIVE', '1') != '0'
    if native and _native_lib() is not None and rounds_per_task == 1:
        L = _native_lib()
        z64 = np.zeros(0, np.int64)
        cu, ci = (np.ascontiguousarray(cur[0], np.int64), np.ascontiguousarray(cur[1], np.int64)) if has_cur else (z64, z64)
        cy = np.ascontiguousarray(cur[2], np.float32) if has_cur else np.zeros(0, np.float32)
        pu, pi = (np.ascontiguousarray(prev[0], np.int64), np.ascontiguousarray(prev[1], np.int64)) if has_prev else (z64, z64)
        ps = AltPlanParamsStruct(side, per_slice, n_classes, ALT_PEND_JOB_MIN, ALT_PEND_PER_SLICE, slots)
        h = L.invpref_alt_plan_build(cu.ctypes.data, ci.ctypes.data, cy.ctypes.data, len(cu),
                                     pu.ctypes.data if has_prev else None, pi.ctypes.data if has_prev else None, len(pu),
                                     user_num, item_num, C.byref(ps))
        if not h:
            raise ValueError('native alt plan builder: invalid arguments (row ids out of range?)')
        return _alt_from_handle(L, h, side, has_prev, has_cur, len(cu), len(pu), per_slice, n_classes, rows_per_stream_task,
                                n_partials_prev, slots)
    if has_cur:
        u, i, y = (np.asarray(cur[0], np.int64), np.asarray(cur[1], np.int64), np.asarray(cur[2], np.float32))
        own, oth = (u, i) if side == 0 else (i, u)
        n = len(own)
        po = np.argsort(own, kind='stable')
        cols = (oth[po].astype(np.int32), po.astype(np.int32), y.view(np.int32)[po])
        own_sorted = own[po]
        cnt = np.bincount(own, minlength=own_num).astype(np.int64)
        push_slot = np.argsort(np.argsort(oth, kind='stable'), kind='stable').astype(np.int32)
        lst = np.stack([cols[0], cols[1], cols[2], np.zeros(n, np.int32)], axis=1).reshape(-1)
    else:
        n = 0
        cols, own_sorted = (np.zeros(0, np.int32),) * 3, np.zeros(0, np.int64)
        cnt = np.zeros(own_num, np.int64)
        push_slot, lst = np.zeros(0, np.int32), np.zeros(0, np.int32)
    if has_prev:
        ownp = np.asarray(prev[0] if side == 0 else prev[1], np.int64)
        n_prev = len(ownp)
        cntp = np.bincount(ownp, minlength=own_num).astype(np.int64)
    else:
        n_prev = 0
        cntp = np.zeros(own_num, np.int64)
    if n and max(cnt.max(), cntp.max()) >= MAX_ROW_COUNT:
        raise ValueError('a row with too many interactions in one minibatch overflows the job descriptor')
    ptrp = np.concatenate([[0], np.cumsum(cntp)])
    pjob = (cnt == 0) & (cntp >= ALT_PEND_JOB_MIN)          # pending-only rows worth slicing
    rcls = row_class(np.arange(own_num), n_classes)
    d_parts, p_parts, s_parts = [], [], []
    cls = np.zeros((8, 4), np.int32)
    ownp_sorted = np.sort(ownp, kind='stable') if has_prev else np.zeros(0, np.int64)
    for c in range(n_classes):
        parts = []
        if has_cur:
            d, _ = _side_rounds(own_sorted, cols, own_num, ng, per_slice, 1, 2, skip=(cnt == 0) | (rcls != c))
            parts.append(d)
        if pjob.any():
            # jobs without interactions: slices over the pending rows (mode 7 ranges of the previous S-sorted order)
            d, _ = _side_rounds(ownp_sorted, (), own_num, ng, ALT_PEND_PER_SLICE, 1, 0, skip=~pjob | (rcls != c))
            parts.append(d)
        d = np.concatenate(parts) if parts else np.zeros((0, ng, 8), np.int32)
        d = d[(d[:, :, 0] >= 0).any(axis=1)] if len(d) else d            # (drop the padding rounds of the parts)
        pend = np.zeros((len(d), ng, 4), np.int32)
        if len(d):
            row = d[:, :, 0].astype(np.int64)
            act = row >= 0
            rowc = np.where(act, row, 0)
            g = (d[:, :, 1] >> 1) & 31
            g = np.where(g == 0, 32, g)            # (32 slices are stored as 0)
            k = np.arange(ng)[None, :] % np.maximum(g, 1)
            cp = np.where(act, cntp[rowc], 0)
            ln = -(-cp // np.maximum(g, 1))
            a0 = ptrp[rowc] + np.minimum(k * ln, cp)
            b0 = ptrp[rowc] + np.minimum((k + 1) * ln, cp)
            pend_only = act & (cnt[rowc] == 0)
            pend[:, :, 0] = np.where(act, a0, 0)
            pend[:, :, 1] = np.where(act, b0, 0)
            pend[:, :, 2] = cp
            # pending-only jobs: no interactions (mode 0, count 0), leader / slices bits kept
            meta = d[:, :, 1]
            meta = np.where(pend_only, meta & 0x3f, meta)
            d[:, :, 1] = meta
            d[:, :, 2:] = np.where(pend_only[:, :, None], 0, d[:, :, 2:])
            has = (cp > 0).any(axis=1)
            d[:, :, 1] |= np.where(has, np.int32(-2 ** 31), np.int32(0))[:, None]
        pad = (-len(d)) % rounds_per_task
        if pad:
            idle = np.zeros((pad, ng, 8), np.int32)
            idle[:, :, 0] = -1
            idle[:, :, 1] = 1 << 1
            d = np.concatenate([d, idle])
            pend = np.concatenate([pend, np.zeros((pad, ng, 4), np.int32)])
        d_parts.append(d)
        p_parts.append(pend)
        rows = np.flatnonzero((cnt == 0) & ~pjob & (rcls == c))
        s = np.stack([rows, ptrp[rows], ptrp[rows + 1], cntp[rows]], axis=1).astype(np.int32)
        s_parts.append(s)
    rb = sb = 0
    for c in range(n_classes):
        cls[c] = rb, len(d_parts[c]), sb, len(s_parts[c])
        rb += len(d_parts[c])
        sb += len(s_parts[c])
    desc = np.concatenate(d_parts)
    return dict(side=side, has_prev=int(has_prev), has_cur=int(has_cur), n=n, n_prev=n_prev, lanes_per_group=lanes,
                slots_per_round=ng, rounds_per_task=rounds_per_task, desc=desc, pend=np.concatenate(p_parts), list=lst, push_slot=push_slot,
                stream=np.concatenate(s_parts).reshape(-1), n_stream=sb, rows_per_stream_task=rows_per_stream_task,
                n_classes=n_classes, cls=cls, n_partials_prev=int(n_partials_prev),
                n_tasks=(len(desc) // rounds_per_task if has_cur else 0), per_slice=per_slice)


def _alt_from_handle(L, h, side, has_prev, has_cur, n, n_prev, per_slice, n_classes, rows_per_stream_task, n_partials_prev,
                     slots=16) -> dict:
    out = []
    try:
        for which in range(6):
            ptr = C.POINTER(C.c_int32)()
            ln = L.invpref_plan_array(h, which, C.byref(ptr))
            out.append(np.ctypeslib.as_array(ptr, shape=(ln,)).copy() if ln > 0 else np.zeros(0, np.int32))
    finally:
        L.invpref_plan_free(h)
    desc, pend, lst, push_slot, stream, cls = out
    desc = desc.reshape(-1, slots, 8)
    return dict(side=side, has_prev=int(has_prev), has_cur=int(has_cur), n=int(n), n_prev=int(n_prev), lanes_per_group=16,
                slots_per_round=slots, rounds_per_task=1, desc=desc, pend=pend.reshape(-1, slots, 4), list=lst, push_slot=push_slot, stream=stream,
                n_stream=len(stream) // 4, rows_per_stream_task=rows_per_stream_task, n_classes=n_classes,
                cls=cls.reshape(8, 4).copy(), n_partials_prev=int(n_partials_prev), n_tasks=(len(desc) if has_cur else 0),
                per_slice=per_slice)


def alt_slots_for(own_rows: np.ndarray, n_rows: int, per_slice: int, n_classes: int = N_CLASSES) -> int:
    """Group slots per round for a side's evaluating launches: 32 (workgroups of 512 threads, one per CU: a hot row's
    interactions spread over up to 32 slices) while such a launch still fits ONE residency of the 256 CUs, else 16 (256
    threads, three per CU).  own_rows: the side's row of every interaction of a representative minibatch."""
    forced = os.environ.get('INVPREF_ALT_SLOTS')
    if forced in ('16', '32'):
        return int(forced)
    cnt = np.bincount(np.asarray(own_rows, np.int64), minlength=n_rows)
    c = cnt[cnt > 0]
    if len(c) == 0:
        return 16
    need = np.maximum(1, -(-c // per_slice))
    slices = np.minimum(32, 1 << np.ceil(np.log2(need)).astype(np.int64))
    rounds = sum(-(-int((slices == g).sum()) // (32 // g)) for g in (1, 2, 4, 8, 16, 32))
    # (per class the round lists are rounded up: allow one round per class and slice count in use)
    rounds += n_classes * len(np.unique(slices)) // 2
    return 32 if rounds + 48 <= 256 else 16


def build_alt_plans(users: np.ndarray, items: np.ndarray, scores: np.ndarray, specs, user_num: int, item_num: int,
                    factor_num: int = 64, per_slice_u: int | None = None, per_slice_i: int | None = None,
                    n_classes: int | None = None, rows_per_stream_task: int | None = None, threads: int = 0,
                    slots_u: int = 16, slots_i: int = 16) -> list:
    """Many alt plans over the same interaction arrays in one native call (a thread pool): specs = [(cur, prev, side)] with
    cur / prev = (lo, n) ranges of the arrays or None.  Falls back to build_alt_plan per spec without the library."""
    users, items = np.ascontiguousarray(users, np.int64), np.ascontiguousarray(items, np.int64)
    scores = np.ascontiguousarray(scores, np.float32)
    if per_slice_u is None:
        per_slice_u = int(os.environ.get('INVPREF_ALT_PER_SLICE_U', '2'))
    if per_slice_i is None:
        per_slice_i = int(os.environ.get('INVPREF_ALT_PER_SLICE_I', '2'))
    if n_classes is None:
        n_classes = int(os.environ.get('INVPREF_PLAN_CLASSES', str(N_CLASSES)))
    n_classes = max(1, min(8, n_classes))
    if rows_per_stream_task is None:
        rows_per_stream_task = int(os.environ.get('INVPREF_PLAN_STREAM_ROWS', str(stream_rows_default(factor_num))))

    def rng(r):
        return None if r is None else (users[r[0]:r[0] + r[1]], items[r[0]:r[0] + r[1]], scores[r[0]:r[0] + r[1]])
    L = _native_lib()
    if L is not None and os.environ.get('INVPREF_PLAN_NATIVE', '1') != '0' and (per_slice_u != per_slice_i or slots_u != slots_i):
        # the sides differ in slice length or slots per round (the Yahoo shape: rounds of 16 slots for the users' launches, of
        # 32 for the items'): ONE threaded native call per side instead of one serial call per plan
        out = [None] * len(specs)
        for sd in (0, 1):
            idx = [j for j, sp in enumerate(specs) if sp[2] == sd]
            if idx:
                ps, sl = (per_slice_i, slots_i) if sd else (per_slice_u, slots_u)
                for j, pl in zip(idx, build_alt_plans(users, items, scores, [specs[j] for j in idx], user_num, item_num,
                                                      factor_num=factor_num, per_slice_u=ps, per_slice_i=ps, n_classes=n_classes,
                                                      rows_per_stream_task=rows_per_stream_task, threads=threads, slots_u=sl,
                                                      slots_i=sl)):
                    out[j] = pl
        return out
    if L is None or os.environ.get('INVPREF_PLAN_NATIVE', '1') == '0':
        return [build_alt_plan(rng(c), None if pv is None else rng(pv)[:2], side, user_num, item_num, factor_num=factor_num,
                               per_slice=per_slice_i if side else per_slice_u, n_classes=n_classes,
                               rows_per_stream_task=rows_per_stream_task, slots=slots_i if side else slots_u)
                for c, pv, side in specs]
    k = len(specs)
    cur_lo = np.array([0 if c is None else c[0] for c, _, _ in specs], np.int64)
    cur_n = np.array([0 if c is None else c[1] for c, _, _ in specs], np.int64)
    prev_lo = np.array([0 if pv is None else pv[0] for _, pv, _ in specs], np.int64)
    prev_n = np.array([-1 if pv is None else pv[1] for _, pv, _ in specs], np.int64)
    sides = np.array([sd for _, _, sd in specs], np.int32)
    ps = AltPlanParamsStruct(0, per_slice_u, n_classes, ALT_PEND_JOB_MIN, ALT_PEND_PER_SLICE, slots_u)
    handles = (C.c_void_p * k)()
    rc = L.invpref_alt_plan_build_many(users.ctypes.data, items.ctypes.data, scores.ctypes.data, cur_lo.ctypes.data,
                                       cur_n.ctypes.data, prev_lo.ctypes.data, prev_n.ctypes.data, sides.ctypes.data, k,
                                       user_num, item_num, C.byref(ps), handles, int(threads))
    if rc != 0:
        for h in handles:
            if h:
                L.invpref_plan_free(h)
        raise ValueError('native alt plan builder: invalid arguments (row ids out of range?)')
    return [_alt_from_handle(L, handles[j], specs[j][2], specs[j][1] is not None, specs[j][0] is not None, int(cur_n[j]),
                             max(0, int(prev_n[j])), per_slice_u, n_classes, rows_per_stream_task, 0, slots_u) for j in range(k)]


def alt_workgroups(plan: dict) -> int:
    ncls, cls = int(plan['n_classes']), np.asarray(plan['cls'])
    rpt, spt = plan['rounds_per_task'], plan['rows_per_stream_task']
    return ncls * max(-(-int(cls[c, 1]) // rpt) + -(-int(cls[c, 3]) // spt) for c in range(ncls))


@dataclass
class DeviceAltPlan:
    struct: AltPlanStruct
    buf: torch.Tensor
    n_tasks: int
    n: int
    side: int
    has_prev: int
    has_cur: int
    meta: torch.Tensor = None   # CPU int64: the struct's fields in order, pointers as int32 offsets into buf (-1: NULL)


def alt_struct_from_meta(buf: torch.Tensor, meta: torch.Tensor) -> AltPlanStruct:
    if buf.dtype != torch.int32 or not buf.is_contiguous() or meta.dtype != torch.int64 or meta.is_cuda:
        raise ValueError('alt plan: int32 device buffer + CPU int64 meta tensor expected')
    vals = meta.tolist()
    base, nb = buf.data_ptr(), buf.numel()
    args, i = [], 0
    for name, ty in AltPlanStruct._fields_:
        if name == 'cls':
            args.append((C.c_int32 * 32)(*vals[i:i + 32]))
            i += 32
            continue
        v = vals[i]
        i += 1
        if ty is C.c_void_p:
            if v < 0:
                v = None
            elif v > nb:
                raise ValueError('alt plan: array offset outside the buffer')
            else:
                v = base + 4 * v
        args.append(v)
    if i != len(vals):
        raise ValueError('alt plan: meta length')
    return AltPlanStruct(*args)


def alt_with_partials(dp: DeviceAltPlan, n_partials_prev: int) -> DeviceAltPlan:
    """the same plan behind a launch that left another number of partial slabs (the arrays are shared)"""
    if int(dp.meta[-1]) == int(n_partials_prev):
        return dp
    meta = dp.meta.clone()
    meta[-1] = int(n_partials_prev)
    return DeviceAltPlan(alt_struct_from_meta(dp.buf, meta), dp.buf, dp.n_tasks, dp.n, dp.side, dp.has_prev, dp.has_cur, meta)


def upload_alt(plan: dict, device) -> DeviceAltPlan:
    parts, offs, off = [], {}, 0
    for k in ALT_ARRAYS:
        a = np.ascontiguousarray(plan[k], np.int32).reshape(-1)
        if len(a) == 0:
            offs[k] = -1
            continue
        parts.append((off, a))
        offs[k] = off
        off += len(a) + (-len(a)) % 4
    host = np.zeros(off + 4, np.int32)
    for o, a in parts:
        host[o:o + len(a)] = a
    buf = torch.from_numpy(host).to(device)
    meta = []
    for name, ty in AltPlanStruct._fields_:
        if name == 'cls':
            meta.extend(np.asarray(plan['cls'], np.int32).reshape(-1).tolist())
        elif ty is C.c_void_p:
            meta.append(offs[name])
        elif name == 'n_rounds':
            meta.append(len(plan['desc']))
        else:
            meta.append(int(plan[name]))
    meta = torch.tensor(meta, dtype=torch.int64)
    return DeviceAltPlan(alt_struct_from_meta(buf, meta), buf, plan['n_tasks'], plan['n'], plan['side'], plan['has_prev'],
                         plan['has_cur'], meta)
