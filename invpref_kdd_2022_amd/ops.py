"""Operator layer: convenience wrappers over the ``torch.ops.invpref.*`` custom operators (torch_ops.py),
which in turn forward to the C ABI (include/invpref_hip.h).

Each function enqueues HIP kernels on torch's current stream and returns without syncing.
Reference semantics are cited per function; there is no PyTorch-eager implementation behind
any of them (the operators are registered for the CUDA/HIP dispatch key only).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _capi
from . import torch_ops  # noqa: F401  (registers torch.ops.invpref.*)
from ._capi import (DENSE_REG, IMPLICIT, REG_ENV_EMBED, REG_ONLY_EMBED, REWEIGHT_CLS, REWEIGHT_REC, WEIGHTS_BY_ENV, Coefs,
                    InvPrefError, check, lib, make_tables, ptr, stream_ptr)

PARAM_NAMES = [
    'embed_user_invariant.weight', 'embed_item_invariant.weight',
    'embed_user_env_aware.weight', 'embed_item_env_aware.weight',
    'embed_env.weight', 'env_classifier.linear_map.weight', 'env_classifier.linear_map.bias',
]


def flags_of(implicit: bool, reweight_rec: bool, reweight_cls: bool, reg_only_embed: bool, reg_env_embed: bool,
             dense_reg: bool = True) -> int:
    return (IMPLICIT * bool(implicit) | REWEIGHT_REC * bool(reweight_rec) | REWEIGHT_CLS * bool(reweight_cls)
            | REG_ONLY_EMBED * bool(reg_only_embed) | REG_ENV_EMBED * bool(reg_env_embed)
            | DENSE_REG * bool(dense_reg))


def _ids(t: torch.Tensor, name: str):
    _capi._req(t, torch.int64, name)
    return t


class Workspace:
    """Device scratch for the per-workgroup partial slabs; grown on demand, reused across calls."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(0, dtype=torch.uint8, device=device)
        # A buffer that is outgrown is RETIRED, not freed: a captured HIP graph bakes the address of the scratch it was
        # recorded with, and replays must keep finding live memory there (scratch carries nothing between calls, and
        # every user is ordered on the stream, so a replay working in a retired buffer is as good as in the new one).
        self._retired = []

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf.numel() < nbytes:
            if self.buf.numel():
                self._retired.append(self.buf)
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf

    def get_zeroed(self, nbytes: int) -> torch.Tensor:
        """the scratch of the planned M-step (records + partial slabs), a buffer of its own"""
        z = getattr(self, 'zbuf', None)
        if z is None or z.numel() < nbytes:
            if z is not None:
                self._retired.append(z)
            self.zbuf = z = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        return z


def _o():
    return torch.ops.invpref


def _gpu(*tensors):
    """The operators exist for the CUDA/HIP dispatch key only; say so in this package's own words before the
    dispatcher does ("no kernel for the CPU backend")."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise InvPrefError(f'tensor on {t.device}: the InvPref operators run on the GPU only '
                               '(the HIP path has no CPU fallback)')


def forward(params: Sequence[torch.Tensor], users, items, envs, implicit: bool):
    """InvPref{Implicit,Explicit}.forward (models.py:307-326 / :448-467), values only."""
    _gpu(users, *params)
    return _o().forward(list(params), users, items, envs, bool(implicit))


def mstep_grad(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], users, items, envs, scores,
               sample_weights: Optional[torch.Tensor], batch_norm: int, coefs: Sequence[float], flags: int,
               losses6: torch.Tensor, workspace: Workspace) -> None:
    """Forward + losses + regularisers + backward of train_a_batch (train.py:94-156): ADDS the
    gradients into `grads` and the six loss terms into `losses6` (device fp32[6])."""
    _gpu(users, *params)
    t = make_tables(params)
    ws = workspace.get(lib().invpref_mstep_workspace_bytes(C.byref(t), users.numel()))
    _o().train_step_fused(list(params), list(grads), users, items, envs, scores, sample_weights, int(batch_norm),
                          [float(c) for c in coefs[:6]], int(flags), losses6, ws)


def adam_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
          lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = True) -> None:
    """optimizer.zero_grad() + torch.optim.Adam.step() (train.py:41,155-157) on flat buffers."""
    _gpu(param, grad, exp_avg, exp_avg_sq)
    _o().adam_dense_(param, grad, exp_avg, exp_avg_sq, int(step), float(lr), float(beta1), float(beta2), float(eps),
                     bool(zero_grad))


def adam_ranges_(param, grad, exp_avg, exp_avg_sq, offsets, lengths, step: int, lr: float, beta1: float = 0.9,
                 beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = True, sched=None) -> None:
    """the same rule over up to four (offset, length) pieces of the flat buffers in one launch.
    sched = (state, table, slot): scalars from the device-side schedule (graph replay); the launch moves it on."""
    s_state, s_table, s_slot = sched if sched is not None else (None, None, 0)
    _o().adam_ranges_(param, grad, exp_avg, exp_avg_sq, [int(o) for o in offsets], [int(n) for n in lengths], int(step),
                      float(lr), float(beta1), float(beta2), float(eps), bool(zero_grad), s_state, s_table, int(s_slot))


def pack_rows(flat: torch.Tensor, row_offsets: torch.Tensor, D: int, tail_offset: int, tail_len: int, packed: torch.Tensor,
              vec_ok: bool = True) -> None:
    """the packed exchange of a row-sharded step: rows `row_offsets` (first float of each, D floats) of the flat gradient,
    then its tail [tail_offset, tail_offset + tail_len), copied into `packed` for ONE all-reduce."""
    _o().pack_rows_(flat, row_offsets, int(D), int(tail_offset), int(tail_len), packed, bool(vec_ok))


def unpack_rows(flat: torch.Tensor, row_offsets: torch.Tensor, D: int, tail_offset: int, tail_len: int, packed: torch.Tensor,
                vec_ok: bool = True) -> None:
    """the reverse copy: the all-reduced rows and tail back into the flat gradient"""
    _o().unpack_rows_(flat, row_offsets, int(D), int(tail_offset), int(tail_len), packed, bool(vec_ok))


def estep(params: Sequence[torch.Tensor], users, items, scores, implicit: bool, old_envs: Optional[torch.Tensor],
          workspace: Workspace, eps_rows: Optional[torch.Tensor] = None, new_envs: Optional[torch.Tensor] = None,
          want_weights: bool = True, perm_index: Optional[torch.Tensor] = None, eps_base=None):
    """cluster() + stat_envs() (train.py:235-259, :268-280) over all given interactions.
    Returns (new_envs int64[N], counts int64[E], diff int64[1], class_w fp32[E], sample_w fp32[N]).
    new_envs may be the same tensor as old_envs (in-place update, `estep_assign_`).
    eps_rows: the tie-break rows of train.py:192-196 already gathered per interaction ([N, E] fp32) -- or perm_index
    (uint8 / int32 / int64 [N]: the permutation row drawn for every interaction) + eps_base (the E floats that are
    permuted): the row is unranked on the device."""
    _gpu(users, *params)
    t = make_tables(params)
    ws = workspace.get(lib().invpref_estep_workspace_bytes(C.byref(t), users.numel()))
    if new_envs is not None and old_envs is not None and new_envs.data_ptr() == old_envs.data_ptr():
        counts, diff, cw, sw = _o().estep_assign_(list(params), users, items, scores, new_envs, bool(implicit), eps_rows,
                                                  bool(want_weights), ws, perm_index, eps_base)
        return new_envs, counts, diff, (cw if want_weights else None), (sw if want_weights else None)
    out, counts, diff = _o().estep_assign(list(params), users, items, scores, old_envs, bool(implicit), eps_rows, ws,
                                          perm_index, eps_base)
    if new_envs is not None:
        new_envs.copy_(out)
        out = new_envs
    cw = sw = None
    if want_weights:
        cw, sw = _o().sample_weights(out, counts, users.numel(), t.env_num)
    return out, counts, diff, cw, sw


class EstepState:
    """what invpref_estep_fused_hip keeps between calls: the ticket / ring-position words (zero before the first call), the ring
    of {counts, diff} rows the replayed E-steps write to, and the E! permutation rows of train.py:86-92 (built once)."""

    def __init__(self, env_num: int, device, ring_cap: int = 256):
        self.env_num, self.ring_cap = int(env_num), int(ring_cap)
        self.state = torch.zeros(32 + 32 * 32, dtype=torch.int32, device=device)   # INVPREF_ESTEP_STATE_INTS
        self.ring = torch.zeros(ring_cap, env_num + 1, dtype=torch.int64, device=device)
        self.issued = 0          # host mirror of state[1]: E-steps issued so far
        self.perm_table = None
        if env_num <= 7:
            import numpy as np
            rows = 1
            for k in range(2, env_num + 1):
                rows *= k
            host = np.zeros(rows, np.uint32)
            if lib().invpref_perm_table_fill(env_num, host.ctypes.data) != rows:
                raise InvPrefError('invpref_perm_table_fill failed')
            self.perm_table = torch.from_numpy(host.view(np.int32)).to(device)

    def next_row(self) -> int:
        """ring row the NEXT issued E-step writes (call when issuing / replaying one)"""
        r = self.issued % self.ring_cap
        self.issued += 1
        return r


def estep_fused(params: Sequence[torch.Tensor], users, items, scores, implicit: bool, envs: torch.Tensor, es: EstepState,
                workspace: Workspace, perm_index: Optional[torch.Tensor] = None, eps_base=None,
                counts: Optional[torch.Tensor] = None, diff: Optional[torch.Tensor] = None,
                class_weights: Optional[torch.Tensor] = None, use_ring: bool = True) -> None:
    """cluster() + stat_envs() (train.py:235-259, :268-280) as ONE launch over all given interactions, `envs` updated in
    place; counts / diff / class weights come out of the kernel's epilogue (to the given tensors and / or the next row of
    es.ring: the caller advances es.next_row() per issued or replayed call).  No N-length sample_weights: the planned M-step
    looks class_weights[env] up itself (flags | WEIGHTS_BY_ENV)."""
    _gpu(users, envs, *params)
    t = make_tables(params)
    ws = workspace.get(lib().invpref_estep_workspace_bytes(C.byref(t), users.numel()))
    _o().estep_fused_(list(params), users, items, scores, envs, bool(implicit), perm_index,
                      None if eps_base is None else [float(x) for x in eps_base], es.perm_table, es.state,
                      es.ring if use_ring else None, counts, diff, class_weights, ws)


def stat_envs(envs: torch.Tensor, env_num: int, workspace: Workspace, want_sample_weights: bool = True):
    """stat_envs() (train.py:268-280): (counts int64[E], class_w fp32[E], sample_w fp32[N])."""
    _gpu(envs)
    ws = workspace.get(4 * (env_num + 1) * 2048)
    counts, cw, sw = _o().stat_envs(envs, int(env_num), bool(want_sample_weights), ws)
    return counts, cw, (sw if want_sample_weights else None)


def sample_weights(envs: torch.Tensor, counts: torch.Tensor, n_total: int, env_num: int):
    """Weight half of stat_envs (train.py:274-278) from global counts: (class_w[E], sample_w[N_local])."""
    return _o().sample_weights(envs, counts, int(n_total), int(env_num))


def backward(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], users, items, envs, implicit: bool,
             alpha: float, d_inv, d_env, d_out, workspace: Workspace) -> None:
    """Backward of forward() incl. the gradient-reversal layer (functions.py:7-16): ADDS into grads."""
    t = make_tables(params)
    ws = workspace.get(lib().invpref_mstep_workspace_bytes(C.byref(t), users.numel()))
    _o().backward(list(params), list(grads), users, items, envs, bool(implicit), float(alpha), d_inv, d_env, d_out, ws)


def predict(user_table: torch.Tensor, item_table: torch.Tensor, users: torch.Tensor, sigmoid: bool) -> torch.Tensor:
    """InvPrefImplicit.predict (models.py:393-407): [n_users, item_num] scores."""
    _gpu(user_table, item_table, users)
    return _o().predict(user_table, item_table, users, bool(sigmoid))


def rows_workspace(params, dplan, workspace: Workspace, pure: bool = False) -> torch.Tensor:
    """the scratch of the planned M-step (per-interaction records + per-workgroup partial slabs)"""
    t = (_capi.make_pure_tables if (pure or len(params) == 2) else make_tables)(params)
    return workspace.get_zeroed(lib().invpref_rows_workspace_bytes(C.byref(t), C.byref(dplan.struct)))


def mstep_rows_grad(params, grads, dplan, envs, scores, sample_weights, batch_norm: int, coefs, flags: int,
                     losses6: torch.Tensor, workspace: Workspace, sched=None) -> None:
    """Planned, atomic-free M-step gradient of one minibatch (plan.py): OVERWRITES every row of grads.
    sched = (state, table, slot): a scheduled alpha is read from the device-side schedule (graph replay)."""
    ws = rows_workspace(params, dplan, workspace)
    s_state, s_table, s_slot = sched if sched is not None else (None, None, 0)
    _o().train_step_planned_grad_(list(params), list(grads), dplan.buf, dplan.meta, envs, scores, sample_weights,
                                  int(batch_norm), [float(c) for c in coefs[:6]], int(flags), losses6, s_state, s_table,
                                  int(s_slot), ws)


def mstep_rows_adam(params, new_params, exp_avg, exp_avg_sq, dplan, envs, scores, sample_weights, batch_norm: int,
                    coefs, flags: int, losses6: torch.Tensor, step: int, lr: float, workspace: Workspace,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, pure: bool = False,
                    sched=None, mid_event=None) -> None:
    """M-step + Adam in one pass: reads params, writes new_params, updates the moments in place.
    pure=True: PureMF step (INVPREF_PURE_MF): the table lists hold [user table, item table] only, envs /
    sample_weights may be None.  sched = (state int32[32], table fp32[n, 8], slot): per-step scalars from the
    device-side schedule (graph replay) instead of (step, lr, betas, eps).
    mid_event (profiling, bench.py): a torch.cuda.Event recorded on the stream BETWEEN the step's two launches; the call
    goes straight to the C ABI then (invpref_mstep_rows_adam_profiled_hip)."""
    if pure:
        flags |= _capi.PURE_MF
    ws = rows_workspace(params, dplan, workspace, pure)
    if mid_event is not None:
        mk = _capi.make_pure_tables if pure else make_tables
        t, tn, tm, tv = mk(params), mk(new_params), mk(exp_avg), mk(exp_avg_sq)
        cf = _capi.Coefs(*[float(c) for c in coefs[:6]])
        mid_event.record()   # (creates the underlying hipEvent_t; re-recorded by the library between the launches)
        check(lib().invpref_mstep_rows_adam_profiled_hip(
            C.byref(t), C.byref(tn), C.byref(tm), C.byref(tv), C.byref(dplan.struct), ptr(envs), ptr(scores),
            ptr(sample_weights), int(batch_norm), C.byref(cf), int(flags), ptr(losses6), int(step), float(lr),
            float(beta1), float(beta2), float(eps), ptr(ws), ws.numel(), stream_ptr(), C.c_void_p(mid_event.cuda_event)),
            'invpref_mstep_rows_adam_profiled_hip')
        return
    s_state, s_table, s_slot = sched if sched is not None else (None, None, 0)
    _o().train_step_planned_adam_(list(params), list(new_params), list(exp_avg), list(exp_avg_sq), dplan.buf, dplan.meta,
                                  envs, scores, sample_weights, int(batch_norm), [float(c) for c in coefs[:6]],
                                  int(flags), losses6, int(step), float(lr), float(beta1), float(beta2), float(eps),
                                  s_state, s_table, int(s_slot), ws)


class AltWorkspace:
    """scratch of a run of alternating launches (include/invpref_hip.h: invpref_mstep_alt_hip): two halves of
    {contribution rows, partial slabs} + the fold flags.  Never reallocated while a captured graph may replay into it."""

    def __init__(self, params, n_cap: int, partials_cap: int, pure: bool = False):
        t = (_capi.make_pure_tables if pure else make_tables)(params)
        self.n_cap, self.partials_cap = int(n_cap), int(partials_cap)
        nbytes = lib().invpref_alt_workspace_bytes(C.byref(t), self.n_cap, self.partials_cap)
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=params[0].device)
        self.err_off = lib().invpref_alt_error_offset(C.byref(t), self.n_cap, self.partials_cap)

    def error(self) -> int:
        """1 if a job workgroup ever gave up waiting for the small tables of its step (host sync).  The word is sticky:
        no launch clears it, only reset_error() does."""
        return int(self.buf[self.err_off:self.err_off + 4].view(torch.int32).item())

    def reset_error(self) -> None:
        self.buf[self.err_off:self.err_off + 4].zero_()


def alt_supported(params) -> bool:
    t = (_capi.make_pure_tables if len(params) == 2 else make_tables)(params)
    return bool(lib().invpref_alt_supported(C.byref(t)))


def mstep_alt(params, exp_avg, exp_avg_sq, aplan, envs, sample_weights, batch_norm: int, batch_norm_prev: int, coefs,
              flags: int, losses6_prev, step: int, lr: float, aws: AltWorkspace, parity: int, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, pure: bool = False, sched=None) -> None:
    """ONE launch of the alternating form (include/invpref_hip.h: invpref_mstep_alt_hip): the plan's side applies the
    previous step's pending update, evaluates the current minibatch, applies its own update and pushes for the other side.
    params / moments are updated in place.  sched = (state, table, slot) as for mstep_rows_adam."""
    if pure:
        flags |= _capi.PURE_MF
    _gpu(*params, envs, sample_weights, losses6_prev)
    s_state, s_table, s_slot = sched if sched is not None else (None, None, 0)
    _o().train_step_alt_(list(params), list(exp_avg), list(exp_avg_sq), aplan.buf, aplan.meta, envs, sample_weights,
                         int(batch_norm), int(batch_norm_prev), [float(c) for c in coefs[:6]], int(flags), losses6_prev,
                         int(step), float(lr), float(beta1), float(beta2), float(eps), s_state, s_table, int(s_slot), aws.buf,
                         aws.n_cap, aws.partials_cap, int(parity))


POP_KEYS = ['users_cnt_weight_result', 'items_cnt_weight_result', 'users_normalize_cnt_weight_result',
            'items_normalize_cnt_weight_result', 'users_cnt_result', 'items_cnt_result', 'users_normalize_cnt_result',
            'items_normalize_cnt_result', 'pair_cnt_add_result', 'pair_normalize_cnt_multiply_result']  # train.py:558-569


def static_pop(users, items, envs, env_num: int, user_cnt, item_cnt, user_norm, item_norm, workspace: Workspace):
    """-> float64 [env_num, 10] per-environment popularity means in POP_KEYS order (train.py:509-571)."""
    for t, n in ((users, 'users'), (items, 'items'), (envs, 'envs'), (user_cnt, 'user_cnt'), (item_cnt, 'item_cnt')):
        _ids(t, n)
    _capi._req(user_norm, torch.float64, 'user_norm')
    _capi._req(item_norm, torch.float64, 'item_norm')
    U, I = user_cnt.numel(), item_cnt.numel()
    if user_norm.numel() != U or item_norm.numel() != I or not (users.numel() == items.numel() == envs.numel()):
        raise InvPrefError('static_pop: inconsistent sizes')
    out = torch.empty(env_num, 10, dtype=torch.float64, device=users.device)
    ws = workspace.get(lib().invpref_static_pop_workspace_bytes(U, I, env_num))
    check(lib().invpref_static_pop_hip(ptr(users), ptr(items), ptr(envs), users.numel(), U, I, env_num, ptr(user_cnt),
                                       ptr(item_cnt), ptr(user_norm), ptr(item_norm), ptr(out), ptr(ws), ws.numel(),
                                       stream_ptr()), 'invpref_static_pop_hip')
    return out
