"""Operator layer: thin, validated wrappers over the C ABI (include/invpref_hip.h).

Each function enqueues HIP kernels on torch's current stream and returns without syncing.
Reference semantics are cited per function; there is no PyTorch-eager implementation behind
any of them.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _capi
from ._capi import (DENSE_REG, IMPLICIT, REG_ENV_EMBED, REG_ONLY_EMBED, REWEIGHT_CLS, REWEIGHT_REC, Coefs,
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

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf.numel() < nbytes:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self.buf

    def get_zeroed(self, nbytes: int) -> torch.Tensor:
        """the owner path's replica slabs: zero-filled before first use, left zeroed by every call"""
        z = getattr(self, 'zbuf', None)
        if z is None or z.numel() < nbytes:
            self.zbuf = z = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        return z


def forward(params: Sequence[torch.Tensor], users, items, envs, implicit: bool):
    """InvPref{Implicit,Explicit}.forward (models.py:307-326 / :448-467), values only."""
    t = make_tables(params)
    B = users.numel()
    dev = users.device
    inv = torch.empty(B, dtype=torch.float32, device=dev)
    env = torch.empty(B, dtype=torch.float32, device=dev)
    out = torch.empty(B, t.env_num, dtype=torch.float32, device=dev)
    check(lib().invpref_forward_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                    ptr(_ids(envs, 'envs')), B, IMPLICIT if implicit else 0, ptr(inv), ptr(env),
                                    ptr(out), stream_ptr()), 'invpref_forward_hip')
    return inv, env, out


def mstep_grad(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], users, items, envs, scores,
               sample_weights: Optional[torch.Tensor], batch_norm: int, coefs: Sequence[float], flags: int,
               losses6: torch.Tensor, workspace: Workspace) -> None:
    """Forward + losses + regularisers + backward of train_a_batch (train.py:94-156): ADDS the
    gradients into `grads` and the six loss terms into `losses6` (device fp32[6])."""
    t, g = make_tables(params), make_tables(grads)
    B = users.numel()
    _capi._req(scores, torch.float32, 'scores')
    _capi._req(sample_weights, torch.float32, 'sample_weights')
    _capi._req(losses6, torch.float32, 'losses6')
    cf = Coefs(*[float(c) for c in coefs[:6]])
    need = lib().invpref_mstep_workspace_bytes(C.byref(t), B)
    ws = workspace.get(need)
    check(lib().invpref_mstep_grad_hip(C.byref(t), C.byref(g), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                       ptr(_ids(envs, 'envs')), ptr(scores), ptr(sample_weights), B, int(batch_norm),
                                       C.byref(cf), flags, ptr(losses6), ptr(ws), ws.numel(), stream_ptr()),
          'invpref_mstep_grad_hip')


def adam_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
          lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = True) -> None:
    """optimizer.zero_grad() + torch.optim.Adam.step() (train.py:41,155-157) on flat buffers."""
    for n, t in (('param', param), ('grad', grad), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
        _capi._req(t, torch.float32, n)
    n = param.numel()
    if not (grad.numel() >= n and exp_avg.numel() == n and exp_avg_sq.numel() == n):
        raise InvPrefError('adam_: buffer sizes differ')
    check(lib().invpref_adam_hip(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), n, int(step), float(lr),
                                 float(beta1), float(beta2), float(eps), int(bool(zero_grad)), stream_ptr()),
          'invpref_adam_hip')


def estep(params: Sequence[torch.Tensor], users, items, scores, implicit: bool, old_envs: Optional[torch.Tensor],
          workspace: Workspace, eps_rows: Optional[torch.Tensor] = None, new_envs: Optional[torch.Tensor] = None,
          want_weights: bool = True):
    """cluster() + stat_envs() (train.py:235-259, :268-280) over all given interactions.
    Returns (new_envs int64[N], counts int64[E], diff int64[1], class_w fp32[E], sample_w fp32[N])."""
    t = make_tables(params)
    N = users.numel()
    dev = users.device
    _capi._req(scores, torch.float32, 'scores')
    _capi._req(eps_rows, torch.float32, 'eps_rows')
    if old_envs is not None:
        _ids(old_envs, 'old_envs')
    if new_envs is None:
        new_envs = torch.empty(N, dtype=torch.int64, device=dev)
    counts = torch.empty(t.env_num, dtype=torch.int64, device=dev)
    diff = torch.zeros(1, dtype=torch.int64, device=dev)
    cw = torch.empty(t.env_num, dtype=torch.float32, device=dev) if want_weights else None
    sw = torch.empty(N, dtype=torch.float32, device=dev) if want_weights else None
    ws = workspace.get(lib().invpref_estep_workspace_bytes(C.byref(t), N))
    check(lib().invpref_estep_hip(C.byref(t), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')), ptr(scores), N,
                                  IMPLICIT if implicit else 0, ptr(eps_rows), ptr(old_envs), ptr(new_envs),
                                  ptr(counts), ptr(diff), ptr(cw), ptr(sw), ptr(ws), ws.numel(), stream_ptr()),
          'invpref_estep_hip')
    return new_envs, counts, diff, cw, sw


def stat_envs(envs: torch.Tensor, env_num: int, workspace: Workspace, want_sample_weights: bool = True):
    """stat_envs() (train.py:268-280): (counts int64[E], class_w fp32[E], sample_w fp32[N])."""
    N = envs.numel()
    dev = envs.device
    counts = torch.empty(env_num, dtype=torch.int64, device=dev)
    cw = torch.empty(env_num, dtype=torch.float32, device=dev)
    sw = torch.empty(N, dtype=torch.float32, device=dev) if want_sample_weights else None
    ws = workspace.get(4 * (env_num + 1) * 2048)
    check(lib().invpref_stat_envs_hip(ptr(_ids(envs, 'envs')), N, env_num, ptr(counts), ptr(cw), ptr(sw), ptr(ws),
                                      ws.numel(), stream_ptr()), 'invpref_stat_envs_hip')
    return counts, cw, sw


def sample_weights(envs: torch.Tensor, counts: torch.Tensor, n_total: int, env_num: int):
    """Weight half of stat_envs (train.py:274-278) from global counts: (class_w[E], sample_w[N_local])."""
    N = envs.numel()
    dev = envs.device
    _capi._req(counts, torch.int64, 'counts')
    cw = torch.empty(env_num, dtype=torch.float32, device=dev)
    sw = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib().invpref_sample_weights_hip(ptr(_ids(envs, 'envs')), N, ptr(counts), int(n_total), env_num, ptr(cw),
                                           ptr(sw), stream_ptr()), 'invpref_sample_weights_hip')
    return cw, sw


def backward(params: Sequence[torch.Tensor], grads: Sequence[torch.Tensor], users, items, envs, implicit: bool,
             alpha: float, d_inv, d_env, d_out, workspace: Workspace) -> None:
    """Backward of forward() incl. the gradient-reversal layer (functions.py:7-16): ADDS into grads."""
    t, g = make_tables(params), make_tables(grads)
    B = users.numel()
    for n, x in (('d_inv', d_inv), ('d_env', d_env), ('d_out', d_out)):
        _capi._req(x, torch.float32, n)
    ws = workspace.get(lib().invpref_mstep_workspace_bytes(C.byref(t), B))
    check(lib().invpref_backward_hip(C.byref(t), C.byref(g), ptr(_ids(users, 'users')), ptr(_ids(items, 'items')),
                                     ptr(_ids(envs, 'envs')), B, IMPLICIT if implicit else 0, float(alpha), ptr(d_inv),
                                     ptr(d_env), ptr(d_out), ptr(ws), ws.numel(), stream_ptr()),
          'invpref_backward_hip')


def predict(user_table: torch.Tensor, item_table: torch.Tensor, users: torch.Tensor, sigmoid: bool) -> torch.Tensor:
    """InvPrefImplicit.predict (models.py:393-407): [n_users, item_num] scores."""
    _capi._req(user_table, torch.float32, 'user_table')
    _capi._req(item_table, torch.float32, 'item_table')
    n, (I, D) = users.numel(), item_table.shape
    out = torch.empty(n, I, dtype=torch.float32, device=users.device)
    check(lib().invpref_predict_hip(ptr(user_table), ptr(item_table), ptr(_ids(users, 'users')), n, I, D,
                                    int(bool(sigmoid)), ptr(out), stream_ptr()), 'invpref_predict_hip')
    return out


def mstep_rows_grad(params, grads, dplan, envs, scores, sample_weights, batch_norm: int, coefs, flags: int,
                     losses6: torch.Tensor, workspace: Workspace) -> None:
    """Planned, atomic-free M-step gradient of one minibatch (plan.py): OVERWRITES every row of grads."""
    t, g = make_tables(params), make_tables(grads)
    _capi._req(scores, torch.float32, 'scores')
    _capi._req(sample_weights, torch.float32, 'sample_weights')
    cf = Coefs(*[float(c) for c in coefs[:6]])
    ws = workspace.get_zeroed(lib().invpref_rows_workspace_bytes(C.byref(t), C.byref(dplan.struct)))
    check(lib().invpref_mstep_rows_grad_hip(C.byref(t), C.byref(g), C.byref(dplan.struct), ptr(_ids(envs, 'envs')),
                                             ptr(scores), ptr(sample_weights), int(batch_norm), C.byref(cf), flags,
                                             ptr(losses6), ptr(ws), ws.numel(), stream_ptr()),
          'invpref_mstep_rows_grad_hip')


def mstep_rows_adam(params, new_params, exp_avg, exp_avg_sq, dplan, envs, scores, sample_weights, batch_norm: int,
                    coefs, flags: int, losses6: torch.Tensor, step: int, lr: float, workspace: Workspace,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, pure: bool = False) -> None:
    """M-step + Adam in one pass: reads params, writes new_params, updates the moments in place.
    pure=True: PureMF step (INVPREF_PURE_MF): the table lists hold [user table, item table] only, envs /
    sample_weights may be None."""
    mk = _capi.make_pure_tables if pure else make_tables
    t, tn = mk(params), mk(new_params)
    tm, tv = mk(exp_avg), mk(exp_avg_sq)
    _capi._req(scores, torch.float32, 'scores')
    _capi._req(sample_weights, torch.float32, 'sample_weights')
    if pure:
        flags |= _capi.PURE_MF
    cf = Coefs(*[float(c) for c in coefs[:6]])
    ws = workspace.get_zeroed(lib().invpref_rows_workspace_bytes(C.byref(t), C.byref(dplan.struct)))
    check(lib().invpref_mstep_rows_adam_hip(C.byref(t), C.byref(tn), C.byref(tm), C.byref(tv), C.byref(dplan.struct),
                                             ptr(None if envs is None else _ids(envs, 'envs')), ptr(scores),
                                             ptr(sample_weights), int(batch_norm), C.byref(cf), flags, ptr(losses6),
                                             int(step), float(lr), float(beta1), float(beta2), float(eps), ptr(ws),
                                             ws.numel(), stream_ptr()), 'invpref_mstep_rows_adam_hip')


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
