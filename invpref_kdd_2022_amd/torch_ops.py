"""``torch.ops.invpref.*`` -- the PyTorch custom-op surface of the HIP library (level 3).

Every operator is declared with ``torch.library`` (schema with mutable-argument aliasing, ``Tensor(a!)``), has ONE
implementation, registered for the CUDA (= ROCm/HIP) dispatch key, that forwards to the C ABI of
``include/invpref_hip.h`` on torch's current stream without any host synchronisation, and a fake (meta)
implementation so that fake-tensor tracing / ``torch.compile`` see shapes and aliasing.  There is no CPU kernel:
calling an operator on CPU tensors fails in the dispatcher ("no kernel for the CPU backend").

What replaces what in the reference:

=============================  =====================================================================================
``train_step_fused``           forward + 3 losses + 2 regularisers + ``loss.backward()`` of ``train_a_batch``
                               (train.py:108-156; models.py:307-391; functions.py:4-16): ADDS into ``grads``/``losses6``
``train_step_planned_grad_``   the same on a row plan (plan.py): atomic-free, OVERWRITES every gradient row
``train_step_planned_adam_``   the same + ``optimizer.step()`` in one pass (train.py:94-157 entire)
``train_step_alt_``            ONE launch of the alternating form: a whole ``train_a_batch`` step per launch, the evaluating
                               side (users / items) alternating; tables and moments in place (train.py:94-157, :41)
``adam_dense_``                ``optimizer.zero_grad()`` + ``torch.optim.Adam.step()`` (train.py:41, :155-157)
``adam_ranges_``               the same over up to four pieces of the flat buffers (user-sharded ranks)
``pack_rows_`` / ``unpack_``   the touched rows of the flat gradient into / out of one buffer (row-sharded ranks' exchange)
``estep_assign``               ``cluster_a_batch`` / ``cluster`` (train.py:169-202, :235-259), functional
``estep_assign_``              ``cluster()`` updating ``envs`` in place + the ``stat_envs()`` that follows (train.py:330)
``estep_fused_``               the same as ONE launch: counts, ``diff_num`` and class weights from the kernel's epilogue
``stat_envs``                  ``stat_envs()`` (train.py:268-280)
``sample_weights``             its weight half from global counts (multi-GPU)
``forward`` / ``backward``     ``InvPref*.forward`` values (models.py:307-326, :448-467) and its backward
``predict``                    ``InvPrefImplicit.predict`` (models.py:393-407)
=============================  =====================================================================================

Tensors are borrowed for the call and never retained.  ``workspace`` arguments are caller-owned scratch (uint8),
declared mutable.  A row plan travels as two tensors: its int32 device buffer and a small CPU int64 ``meta`` tensor
(``plan.DevicePlan.meta``: the scalar fields and the array offsets of ``InvPrefRowPlan``).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi
from ._capi import Coefs, InvPrefError, check, lib, make_pure_tables, make_tables, ptr, stream_ptr

_LIB = torch.library.Library('invpref', 'DEF')
NAMES = []


def _define(schema: str):
    _LIB.define(schema)
    NAMES.append(schema.split('(')[0])


def _impl(name: str):
    def deco(fn):
        _LIB.impl(name, fn, 'CUDA')
        return fn
    return deco


def _fake(name: str):
    return torch.library.register_fake(f'invpref::{name}', lib=_LIB)


def _ids(t, name):
    _capi._req(t, torch.int64, name)
    return t


def _f32(t, name):
    _capi._req(t, torch.float32, name)
    return t


def _tables(ts):
    return make_pure_tables(ts) if len(ts) == 2 else make_tables(ts)


def _coefs(coefs) -> Coefs:
    if len(coefs) < 6:
        raise InvPrefError('coefs = [invariant_coe, env_aware_coe, env_coe, L2_coe, L1_coe, alpha]')
    return Coefs(*[float(c) for c in coefs[:6]])


def _plan_struct(plan_buf: torch.Tensor, plan_meta: torch.Tensor):
    from .plan import struct_from_meta
    return struct_from_meta(plan_buf, plan_meta)


# ------------------------------------------------------------------------------------------------ forward / backward
_define('forward(Tensor[] tables, Tensor users, Tensor items, Tensor envs, bool implicit) -> (Tensor, Tensor, Tensor)')


@_impl('forward')
def _forward(tables, users, items, envs, implicit):
    t = make_tables(tables)
    B = users.numel()
    dev = users.device
    inv = torch.empty(B, dtype=torch.float32, device=dev)
    env = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty(B, t.env_num, dtype=torch.float32, device=dev)
    check(lib().invpref_forward_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                    ptr(_ids(envs, 'envs')), B, _capi.IMPLICIT if implicit else 0, ptr(inv), ptr(env),
                                    ptr(out), stream_ptr()), 'invpref_forward_hip')
    return inv, env, out


@_fake('forward')
def _forward_fake(tables, users, items, envs, implicit):
    B, E = users.numel(), tables[4].shape[0]
    f = dict(dtype=torch.float32, device=users.device)
    return torch.empty(B, **f), torch.empty(B, **f), torch.empty(B, E, **f)


_define('backward(Tensor[] tables, Tensor(a!)[] grads, Tensor users, Tensor items, Tensor envs, bool implicit, '
        'float alpha, Tensor? d_inv, Tensor? d_env, Tensor? d_out, Tensor(b!) workspace) -> ()')


@_impl('backward')
def _backward(tables, grads, users, items, envs, implicit, alpha, d_inv, d_env, d_out, workspace):
    t, g = make_tables(tables), make_tables(grads)
    B = users.numel()
    for n, x in (('d_inv', d_inv), ('d_env', d_env), ('d_out', d_out)):
        _capi._req(x, torch.float32, n)
    check(lib().invpref_backward_hip(C.byref(t), C.byref(g), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                     ptr(_ids(envs, 'envs')), B, _capi.IMPLICIT if implicit else 0, float(alpha),
                                     ptr(d_inv), ptr(d_env), ptr(d_out), ptr(workspace), workspace.numel(),
                                     stream_ptr()), 'invpref_backward_hip')


@_fake('backward')
def _backward_fake(tables, grads, users, items, envs, implicit, alpha, d_inv, d_env, d_out, workspace):
    return None


# ------------------------------------------------------------------------------------------------ M-step
_define('train_step_fused(Tensor[] tables, Tensor(a!)[] grads, Tensor users, Tensor items, Tensor envs, Tensor scores, '
        'Tensor? sample_weights, int batch_norm, float[] coefs, int flags, Tensor(b!) losses6, Tensor(c!) workspace) -> ()')


@_impl('train_step_fused')
def _train_step_fused(tables, grads, users, items, envs, scores, sample_weights, batch_norm, coefs, flags, losses6,
                      workspace):
    t, g = make_tables(tables), make_tables(grads)
    B = users.numel()
    _f32(scores, 'scores'); _f32(sample_weights, 'sample_weights'); _f32(losses6, 'losses6')
    cf = _coefs(coefs)
    check(lib().invpref_mstep_grad_hip(C.byref(t), C.byref(g), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                       ptr(_ids(envs, 'envs')), ptr(scores), ptr(sample_weights), B, int(batch_norm),
                                       C.byref(cf), int(flags), ptr(losses6), ptr(workspace), workspace.numel(),
                                       stream_ptr()), 'invpref_mstep_grad_hip')


@_fake('train_step_fused')
def _train_step_fused_fake(tables, grads, users, items, envs, scores, sample_weights, batch_norm, coefs, flags, losses6,
                           workspace):
    return None


_define('train_step_planned_grad_(Tensor[] tables, Tensor(a!)[] grads, Tensor plan_buf, Tensor plan_meta, Tensor? envs, '
        'Tensor scores, Tensor? sample_weights, int batch_norm, float[] coefs, int flags, Tensor(b!) losses6, '
        'Tensor? sched_state, Tensor? sched_table, int sched_slot, Tensor(c!) workspace) -> ()')


def _sched_struct(sched_state, sched_table, sched_slot):
    _capi._req(sched_state, torch.int32, 'sched_state')
    _f32(sched_table, 'sched_table')
    if sched_table is None or sched_state.numel() < 32 or sched_table.dim() != 2 or sched_table.shape[1] != 8:
        raise InvPrefError('sched_state int32[32] and sched_table float32[n, 8] go together')
    return _capi.AdamSchedule(sched_state.data_ptr(), sched_table.data_ptr(), sched_table.shape[0], int(sched_slot) & 1)


@_impl('train_step_planned_grad_')
def _planned_grad(tables, grads, plan_buf, plan_meta, envs, scores, sample_weights, batch_norm, coefs, flags, losses6,
                  sched_state, sched_table, sched_slot, workspace):
    t, g = _tables(tables), _tables(grads)
    _f32(scores, 'scores'); _f32(sample_weights, 'sample_weights'); _f32(losses6, 'losses6')
    cf = _coefs(coefs)
    ps = _plan_struct(plan_buf, plan_meta)
    if sched_state is not None:   # graph replay: a scheduled alpha comes from the device-side schedule
        sc = _sched_struct(sched_state, sched_table, sched_slot)
        check(lib().invpref_mstep_rows_grad_sched_hip(C.byref(t), C.byref(g), C.byref(ps),
                                                      ptr(None if envs is None else _ids(envs, 'envs')), ptr(scores),
                                                      ptr(sample_weights), int(batch_norm), C.byref(cf), int(flags),
                                                      ptr(losses6), C.byref(sc), ptr(workspace), workspace.numel(),
                                                      stream_ptr()), 'invpref_mstep_rows_grad_sched_hip')
        return
    check(lib().invpref_mstep_rows_grad_hip(C.byref(t), C.byref(g), C.byref(ps),
                                             ptr(None if envs is None else _ids(envs, 'envs')), ptr(scores),
                                             ptr(sample_weights), int(batch_norm), C.byref(cf), int(flags),
                                             ptr(losses6), ptr(workspace), workspace.numel(), stream_ptr()),
          'invpref_mstep_rows_grad_hip')


@_fake('train_step_planned_grad_')
def _planned_grad_fake(tables, grads, plan_buf, plan_meta, envs, scores, sample_weights, batch_norm, coefs, flags,
                       losses6, sched_state, sched_table, sched_slot, workspace):
    return None


_define('train_step_planned_adam_(Tensor[] tables, Tensor(a!)[] new_tables, Tensor(b!)[] exp_avg, Tensor(c!)[] exp_avg_sq, '
        'Tensor plan_buf, Tensor plan_meta, Tensor? envs, Tensor scores, Tensor? sample_weights, int batch_norm, '
        'float[] coefs, int flags, Tensor(d!) losses6, int step, float lr, float beta1, float beta2, float eps, '
        'Tensor(e!)? sched_state, Tensor? sched_table, int sched_slot, Tensor(f!) workspace) -> ()')


@_impl('train_step_planned_adam_')
def _planned_adam(tables, new_tables, exp_avg, exp_avg_sq, plan_buf, plan_meta, envs, scores, sample_weights,
                  batch_norm, coefs, flags, losses6, step, lr, beta1, beta2, eps, sched_state, sched_table, sched_slot,
                  workspace):
    t, tn, tm, tv = _tables(tables), _tables(new_tables), _tables(exp_avg), _tables(exp_avg_sq)
    _f32(scores, 'scores'); _f32(sample_weights, 'sample_weights'); _f32(losses6, 'losses6')
    cf = _coefs(coefs)
    ps = _plan_struct(plan_buf, plan_meta)
    pe = ptr(None if envs is None else _ids(envs, 'envs'))
    if sched_state is not None:
        # Adam scalars (and a scheduled alpha) come from the device-side schedule: graph replay freezes arguments
        sc = _sched_struct(sched_state, sched_table, sched_slot)
        check(lib().invpref_mstep_rows_adam_sched_hip(C.byref(t), C.byref(tn), C.byref(tm), C.byref(tv), C.byref(ps), pe,
                                                      ptr(scores), ptr(sample_weights), int(batch_norm), C.byref(cf),
                                                      int(flags), ptr(losses6), C.byref(sc), ptr(workspace),
                                                      workspace.numel(), stream_ptr()),
              'invpref_mstep_rows_adam_sched_hip')
        return
    check(lib().invpref_mstep_rows_adam_hip(C.byref(t), C.byref(tn), C.byref(tm), C.byref(tv), C.byref(ps), pe,
                                             ptr(scores), ptr(sample_weights), int(batch_norm), C.byref(cf), int(flags),
                                             ptr(losses6), int(step), float(lr), float(beta1), float(beta2), float(eps),
                                             ptr(workspace), workspace.numel(), stream_ptr()),
          'invpref_mstep_rows_adam_hip')


@_fake('train_step_planned_adam_')
def _planned_adam_fake(tables, new_tables, exp_avg, exp_avg_sq, plan_buf, plan_meta, envs, scores, sample_weights,
                       batch_norm, coefs, flags, losses6, step, lr, beta1, beta2, eps, sched_state, sched_table,
                       sched_slot, workspace):
    return None


_define('train_step_alt_(Tensor(a!)[] tables, Tensor(b!)[] exp_avg, Tensor(c!)[] exp_avg_sq, Tensor plan_buf, Tensor plan_meta, '
        'Tensor? envs, Tensor? sample_weights, int batch_norm, int batch_norm_prev, float[] coefs, int flags, '
        'Tensor(d!)? losses6_prev, int step, float lr, float beta1, float beta2, float eps, Tensor(e!)? sched_state, '
        'Tensor? sched_table, int sched_slot, Tensor(f!) workspace, int n_cap, int partials_cap, int parity) -> ()')


@_impl('train_step_alt_')
def _step_alt(tables, exp_avg, exp_avg_sq, plan_buf, plan_meta, envs, sample_weights, batch_norm, batch_norm_prev, coefs,
              flags, losses6_prev, step, lr, beta1, beta2, eps, sched_state, sched_table, sched_slot, workspace, n_cap,
              partials_cap, parity):
    # include/invpref_hip.h: invpref_mstep_alt_hip.  The plan (plan.DeviceAltPlan) travels like a row plan: its int32 device
    # buffer and a CPU int64 meta tensor
    from .plan import alt_struct_from_meta
    t, tm, tv = _tables(tables), _tables(exp_avg), _tables(exp_avg_sq)
    _f32(sample_weights, 'sample_weights'); _f32(losses6_prev, 'losses6_prev')
    cf = _coefs(coefs)
    ps = alt_struct_from_meta(plan_buf, plan_meta)
    pe = ptr(None if envs is None else _ids(envs, 'envs'))
    sc = None if sched_state is None else C.byref(_sched_struct(sched_state, sched_table, sched_slot))
    check(lib().invpref_mstep_alt_hip(C.byref(t), C.byref(tm), C.byref(tv), C.byref(ps), pe, ptr(sample_weights),
                                      int(batch_norm), int(batch_norm_prev), C.byref(cf), int(flags), ptr(losses6_prev),
                                      int(step), float(lr), float(beta1), float(beta2), float(eps), sc, ptr(workspace),
                                      workspace.numel(), int(n_cap), int(partials_cap), int(parity), stream_ptr()),
          'invpref_mstep_alt_hip')


@_fake('train_step_alt_')
def _step_alt_fake(tables, exp_avg, exp_avg_sq, plan_buf, plan_meta, envs, sample_weights, batch_norm, batch_norm_prev, coefs,
                   flags, losses6_prev, step, lr, beta1, beta2, eps, sched_state, sched_table, sched_slot, workspace, n_cap,
                   partials_cap, parity):
    return None




# ------------------------------------------------------------------------------------------------ Adam
_define('adam_dense_(Tensor(a!) param, Tensor(b!) grad, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, int step, float lr, '
        'float beta1, float beta2, float eps, bool zero_grad) -> ()')


def _adam_check(param, grad, exp_avg, exp_avg_sq):
    for n, t in (('param', param), ('grad', grad), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
        _f32(t, n)
    n = param.numel()
    if not (grad.numel() >= n and exp_avg.numel() == n and exp_avg_sq.numel() == n):
        raise InvPrefError('adam: buffer sizes differ')
    return n


@_impl('adam_dense_')
def _adam_dense(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, zero_grad):
    n = _adam_check(param, grad, exp_avg, exp_avg_sq)
    check(lib().invpref_adam_hip(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), n, int(step), float(lr),
                                 float(beta1), float(beta2), float(eps), int(bool(zero_grad)), stream_ptr()),
          'invpref_adam_hip')


@_fake('adam_dense_')
def _adam_dense_fake(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, zero_grad):
    return None


# ------------------------------------------------------------------------------------------------ packed exchange
_define('pack_rows_(Tensor flat, Tensor row_offsets, int D, int tail_offset, int tail_len, Tensor(a!) packed, bool vec_ok) -> ()')
_define('unpack_rows_(Tensor(a!) flat, Tensor row_offsets, int D, int tail_offset, int tail_len, Tensor packed, bool vec_ok) -> ()')


def _pack_check(flat, row_offsets, D, tail_offset, tail_len, packed):
    _f32(flat, 'flat')
    _f32(packed, 'packed')
    if row_offsets.dtype != torch.int64 or not row_offsets.is_contiguous() or row_offsets.device != flat.device:
        raise InvPrefError('pack_rows: row_offsets must be a contiguous int64 tensor on the device of `flat`')
    n = row_offsets.numel()
    if packed.numel() < n * D + tail_len or tail_offset + tail_len > flat.numel():
        raise InvPrefError('pack_rows: buffer too small')
    return n


@_impl('pack_rows_')
def _pack_rows(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok):
    n = _pack_check(flat, row_offsets, D, tail_offset, tail_len, packed)
    check(lib().invpref_pack_rows_hip(ptr(flat), ptr(row_offsets), n, int(D), int(tail_offset), int(tail_len), ptr(packed),
                                      int(bool(vec_ok)), stream_ptr()), 'invpref_pack_rows_hip')


@_impl('unpack_rows_')
def _unpack_rows(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok):
    n = _pack_check(flat, row_offsets, D, tail_offset, tail_len, packed)
    check(lib().invpref_unpack_rows_hip(ptr(flat), ptr(row_offsets), n, int(D), int(tail_offset), int(tail_len), ptr(packed),
                                        int(bool(vec_ok)), stream_ptr()), 'invpref_unpack_rows_hip')


@_fake('pack_rows_')
def _pack_rows_fake(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok):
    return None


@_fake('unpack_rows_')
def _unpack_rows_fake(flat, row_offsets, D, tail_offset, tail_len, packed, vec_ok):
    return None


_define('adam_ranges_(Tensor(a!) param, Tensor(b!) grad, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, int[] offsets, '
        'int[] lengths, int step, float lr, float beta1, float beta2, float eps, bool zero_grad, '
        'Tensor(e!)? sched_state, Tensor? sched_table, int sched_slot) -> ()')


@_impl('adam_ranges_')
def _adam_ranges(param, grad, exp_avg, exp_avg_sq, offsets, lengths, step, lr, beta1, beta2, eps, zero_grad,
                 sched_state, sched_table, sched_slot):
    n = _adam_check(param, grad, exp_avg, exp_avg_sq)
    k = len(offsets)
    if k != len(lengths) or not 1 <= k <= 4 or any(o < 0 or ln <= 0 or o + ln > n for o, ln in zip(offsets, lengths)):
        raise InvPrefError('adam_ranges_: 1..4 (offset, length) pieces inside the buffers')
    offs, lens = (C.c_int64 * k)(*offsets), (C.c_int64 * k)(*lengths)
    if sched_state is not None:   # graph replay: scalars from the device-side schedule, which this launch moves on
        sc = _sched_struct(sched_state, sched_table, sched_slot)
        check(lib().invpref_adam_ranges_sched_hip(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), offs, lens, k,
                                                  C.byref(sc), int(bool(zero_grad)), stream_ptr()),
              'invpref_adam_ranges_sched_hip')
        return
    check(lib().invpref_adam_ranges_hip(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), offs, lens, k, int(step),
                                        float(lr), float(beta1), float(beta2), float(eps), int(bool(zero_grad)),
                                        stream_ptr()), 'invpref_adam_ranges_hip')


@_fake('adam_ranges_')
def _adam_ranges_fake(param, grad, exp_avg, exp_avg_sq, offsets, lengths, step, lr, beta1, beta2, eps, zero_grad,
                      sched_state, sched_table, sched_slot):
    return None


# ------------------------------------------------------------------------------------------------ E-step
_PERM_BYTES = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}


def _estep_call(tables, users, items, scores, implicit, eps_rows, old_envs, new_envs, want_weights, workspace,
                perm_index=None, eps_base=None):
    t = make_tables(tables)
    N = users.numel()
    dev = users.device
    _f32(scores, 'scores'); _f32(eps_rows, 'eps_rows')
    if perm_index is not None:
        # train.py:192-196 with the permutation row unranked on the device (invpref_estep_perm_hip)
        if eps_rows is not None or eps_base is None or len(eps_base) != t.env_num or perm_index.dtype not in _PERM_BYTES \
                or perm_index.numel() != N:
            raise InvPrefError('perm_index: one uint8 / int32 / int64 permutation row per interaction + eps_base[env_num]')
        if not (perm_index.is_cuda or (perm_index.is_pinned() and t.env_num <= 7)) or not perm_index.is_contiguous():
            raise InvPrefError('perm_index: a contiguous device tensor, or (up to 7 environments) pinned host memory')
        if old_envs is not None:
            _ids(old_envs, 'old_envs')
        counts = torch.empty(t.env_num, dtype=torch.int64, device=dev)
        diff = torch.zeros(1, dtype=torch.int64, device=dev)
        cw = torch.empty(t.env_num if want_weights else 0, dtype=torch.float32, device=dev)
        sw = torch.empty(N if want_weights else 0, dtype=torch.float32, device=dev)
        base = (C.c_float * t.env_num)(*[float(x) for x in eps_base])
        check(lib().invpref_estep_perm_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')), ptr(scores), N,
                                           _capi.IMPLICIT if implicit else 0, ptr(perm_index), _PERM_BYTES[perm_index.dtype],
                                           base, ptr(old_envs), ptr(new_envs), ptr(counts), ptr(diff),
                                           ptr(cw) if want_weights else None, ptr(sw) if want_weights else None,
                                           ptr(workspace), workspace.numel(), stream_ptr()), 'invpref_estep_perm_hip')
        return counts, diff, cw, sw
    if old_envs is not None:
        _ids(old_envs, 'old_envs')
    counts = torch.empty(t.env_num, dtype=torch.int64, device=dev)
    diff = torch.zeros(1, dtype=torch.int64, device=dev)
    cw = torch.empty(t.env_num if want_weights else 0, dtype=torch.float32, device=dev)
    sw = torch.empty(N if want_weights else 0, dtype=torch.float32, device=dev)
    check(lib().invpref_estep_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')), ptr(scores), N,
                                  _capi.IMPLICIT if implicit else 0, ptr(eps_rows), ptr(old_envs), ptr(new_envs),
                                  ptr(counts), ptr(diff), ptr(cw) if want_weights else None,
                                  ptr(sw) if want_weights else None, ptr(workspace), workspace.numel(), stream_ptr()),
          'invpref_estep_hip')
    return counts, diff, cw, sw


_define('estep_assign(Tensor[] tables, Tensor users, Tensor items, Tensor scores, Tensor? old_envs, bool implicit, '
        'Tensor? eps_rows, Tensor(a!) workspace, Tensor? perm_index=None, float[]? eps_base=None) -> (Tensor, Tensor, Tensor)')


@_impl('estep_assign')
def _estep_assign(tables, users, items, scores, old_envs, implicit, eps_rows, workspace, perm_index=None, eps_base=None):
    new_envs = torch.empty(users.numel(), dtype=torch.int64, device=users.device)
    counts, diff, _, _ = _estep_call(tables, users, items, scores, implicit, eps_rows, old_envs, new_envs, False, workspace,
                                     perm_index, eps_base)
    return new_envs, counts, diff


@_fake('estep_assign')
def _estep_assign_fake(tables, users, items, scores, old_envs, implicit, eps_rows, workspace, perm_index=None, eps_base=None):
    i = dict(dtype=torch.int64, device=users.device)
    return torch.empty(users.numel(), **i), torch.empty(tables[4].shape[0], **i), torch.empty(1, **i)


_define('estep_assign_(Tensor[] tables, Tensor users, Tensor items, Tensor scores, Tensor(a!) envs, bool implicit, '
        'Tensor? eps_rows, bool want_weights, Tensor(b!) workspace, Tensor? perm_index=None, float[]? eps_base=None) '
        '-> (Tensor, Tensor, Tensor, Tensor)')


@_impl('estep_assign_')
def _estep_assign_inplace(tables, users, items, scores, envs, implicit, eps_rows, want_weights, workspace, perm_index=None,
                          eps_base=None):
    # envs is read (old assignment of row i) and written (new assignment of row i) by the same lane
    return _estep_call(tables, users, items, scores, implicit, eps_rows, _ids(envs, 'envs'), envs, want_weights,
                       workspace, perm_index, eps_base)


@_fake('estep_assign_')
def _estep_assign_inplace_fake(tables, users, items, scores, envs, implicit, eps_rows, want_weights, workspace,
                               perm_index=None, eps_base=None):
    E, N, dev = tables[4].shape[0], users.numel(), users.device
    return (torch.empty(E, dtype=torch.int64, device=dev), torch.empty(1, dtype=torch.int64, device=dev),
            torch.empty(E if want_weights else 0, dtype=torch.float32, device=dev),
            torch.empty(N if want_weights else 0, dtype=torch.float32, device=dev))


_define('estep_fused_(Tensor[] tables, Tensor users, Tensor items, Tensor scores, Tensor(a!) envs, bool implicit, '
        'Tensor? perm_index, float[]? eps_base, Tensor? perm_table, Tensor(b!) state, Tensor(c!)? ring, Tensor(d!)? counts, '
        'Tensor(e!)? diff, Tensor(f!)? class_weights, Tensor(g!) workspace) -> ()')


@_impl('estep_fused_')
def _estep_fused(tables, users, items, scores, envs, implicit, perm_index, eps_base, perm_table, state, ring, counts, diff,
                 class_weights, workspace):
    # cluster() + stat_envs() as ONE launch (include/invpref_hip.h: invpref_estep_fused_hip; train.py:235-259, :268-280)
    t = _tables(tables)
    N = users.numel()
    _f32(scores, 'scores'); _f32(class_weights, 'class_weights')
    _ids(envs, 'envs')
    _capi._req(state, torch.int32, 'state')
    for x, nm in ((ring, 'ring'), (counts, 'counts'), (diff, 'diff')):
        _capi._req(x, torch.int64, nm)
    if state.numel() < 32 + 32 * 32 or (ring is not None and (ring.dim() != 2 or ring.shape[1] != t.env_num + 1)) \
            or (counts is not None and counts.numel() < t.env_num) or (class_weights is not None and class_weights.numel() < t.env_num):
        raise InvPrefError('estep_fused_: state int32[INVPREF_ESTEP_STATE_INTS], ring int64[cap, env_num + 1], counts / class_weights [env_num]')
    base, nbytes = None, 0
    if perm_index is not None:
        if eps_base is None or len(eps_base) != t.env_num or perm_index.dtype not in _PERM_BYTES or perm_index.numel() != N:
            raise InvPrefError('perm_index: one uint8 / int32 / int64 permutation row per interaction + eps_base[env_num]')
        if not (perm_index.is_cuda or (perm_index.is_pinned() and t.env_num <= 7)) or not perm_index.is_contiguous():
            raise InvPrefError('perm_index: a contiguous device tensor, or (up to 7 environments) pinned host memory')
        base, nbytes = (C.c_float * t.env_num)(*[float(x) for x in eps_base]), _PERM_BYTES[perm_index.dtype]
    if perm_table is not None:
        _capi._req(perm_table, torch.int32, 'perm_table')
    check(lib().invpref_estep_fused_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')), ptr(scores), N,
                                        _capi.IMPLICIT if implicit else 0, ptr(perm_index), nbytes, base, ptr(perm_table),
                                        ptr(envs), ptr(state), ptr(ring), 0 if ring is None else int(ring.shape[0]),
                                        ptr(counts), ptr(diff), ptr(class_weights), ptr(workspace), workspace.numel(),
                                        stream_ptr()), 'invpref_estep_fused_hip')


@_fake('estep_fused_')
def _estep_fused_fake(tables, users, items, scores, envs, implicit, perm_index, eps_base, perm_table, state, ring, counts, diff,
                      class_weights, workspace):
    return None


_define('stat_envs(Tensor envs, int env_num, bool want_sample_weights, Tensor(a!) workspace) -> (Tensor, Tensor, Tensor)')


@_impl('stat_envs')
def _stat_envs(envs, env_num, want_sample_weights, workspace):
    N, dev = envs.numel(), envs.device
    counts = torch.empty(env_num, dtype=torch.int64, device=dev)
    cw = torch.empty(env_num, dtype=torch.float32, device=dev)
    sw = torch.empty(N if want_sample_weights else 0, dtype=torch.float32, device=dev)
    check(lib().invpref_stat_envs_hip(ptr(_ids(envs, 'envs')), N, int(env_num), ptr(counts), ptr(cw),
                                      ptr(sw) if want_sample_weights else None, ptr(workspace), workspace.numel(),
                                      stream_ptr()), 'invpref_stat_envs_hip')
    return counts, cw, sw


@_fake('stat_envs')
def _stat_envs_fake(envs, env_num, want_sample_weights, workspace):
    dev = envs.device
    return (torch.empty(env_num, dtype=torch.int64, device=dev), torch.empty(env_num, dtype=torch.float32, device=dev),
            torch.empty(envs.numel() if want_sample_weights else 0, dtype=torch.float32, device=dev))


_define('sample_weights(Tensor envs, Tensor counts, int n_total, int env_num) -> (Tensor, Tensor)')


@_impl('sample_weights')
def _sample_weights(envs, counts, n_total, env_num):
    N, dev = envs.numel(), envs.device
    _capi._req(counts, torch.int64, 'counts')
    cw = torch.empty(env_num, dtype=torch.float32, device=dev)
    sw = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib().invpref_sample_weights_hip(ptr(_ids(envs, 'envs')), N, ptr(counts), int(n_total), int(env_num), ptr(cw),
                                           ptr(sw), stream_ptr()), 'invpref_sample_weights_hip')
    return cw, sw


@_fake('sample_weights')
def _sample_weights_fake(envs, counts, n_total, env_num):
    f = dict(dtype=torch.float32, device=envs.device)
    return torch.empty(env_num, **f), torch.empty(envs.numel(), **f)


# ------------------------------------------------------------------------------------------------ predict
_define('predict(Tensor user_table, Tensor item_table, Tensor users, bool sigmoid) -> Tensor')


@_impl('predict')
def _predict(user_table, item_table, users, sigmoid):
    _f32(user_table, 'user_table'); _f32(item_table, 'item_table')
    n, (I, D) = users.numel(), item_table.shape
    out = torch.empty(n, I, dtype=torch.float32, device=users.device)
    check(lib().invpref_predict_hip(ptr(user_table), ptr(item_table), ptr(_ids(users, 'users')), n, I, D,
                                    int(bool(sigmoid)), ptr(out), stream_ptr()), 'invpref_predict_hip')
    return out


@_fake('predict')
def _predict_fake(user_table, item_table, users, sigmoid):
    return torch.empty(users.numel(), item_table.shape[0], dtype=torch.float32, device=users.device)
