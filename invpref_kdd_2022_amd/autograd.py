"""Autograd glue for the UNFUSED module surface (InvPref*.forward / get_L*_reg / predict), so that a
caller who builds the loss with torch ops -- e.g. the reference's untouched train.py -- still
runs on the HIP kernels: forward values from ``invpref_forward_hip``, gradients from
``invpref_backward_hip`` / the regulariser mode of ``invpref_mstep_grad_hip``.  The train managers
of this package do not come through here (they use the fused M-step)."""
from __future__ import annotations

import torch

from . import _capi, ops

_ws = {}


def _workspace(device) -> ops.Workspace:
    key = (device.type, device.index)
    if key not in _ws:
        _ws[key] = ops.Workspace(device)
    return _ws[key]


class InvPrefForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, users, items, envs, alpha, implicit, *tables):
        users, items, envs = users.contiguous(), items.contiguous(), envs.contiguous()
        data = [t.detach() for t in tables]
        inv, env, out = ops.forward(data, users, items, envs, implicit)
        ctx.save_for_backward(users, items, envs, *tables)
        ctx.alpha, ctx.implicit = alpha, implicit
        return inv, env, out

    @staticmethod
    def backward(ctx, d_inv, d_env, d_out):
        users, items, envs, *tables = ctx.saved_tensors
        data = [t.detach() for t in tables]
        grads = [torch.zeros_like(t) for t in data]

        def c(x):
            return None if x is None else x.contiguous().float()

        ops.backward(data, grads, users, items, envs, ctx.implicit, ctx.alpha, c(d_inv), c(d_env), c(d_out),
                     _workspace(users.device))
        return (None, None, None, None, None, *grads)


class InvPrefReg(torch.autograd.Function):
    """get_L2_reg / get_L1_reg (models.py:328-391): the fused kernel in regulariser-only mode."""

    @staticmethod
    def forward(ctx, users, items, envs, norm, reg_only_embed, reg_env_embed, *tables):
        users, items, envs = users.contiguous(), items.contiguous(), envs.contiguous()
        data = [t.detach() for t in tables]
        B = users.numel()
        dev = users.device
        flags = ops.flags_of(True, False, False, reg_only_embed, reg_env_embed) | _capi.NO_GRAD
        losses = torch.zeros(6, dtype=torch.float32, device=dev)
        zeros = torch.zeros(B, dtype=torch.float32, device=dev)
        ops.mstep_grad(data, data, users, items, envs, zeros, None, B, (0., 0., 0., 0., 0., 0.), flags, losses,
                       _workspace(dev))
        ctx.save_for_backward(users, items, envs, *tables)
        ctx.cfg = (norm, reg_only_embed, reg_env_embed)
        return losses[3 if norm == 2 else 4].clone()

    @staticmethod
    def backward(ctx, gs):
        users, items, envs, *tables = ctx.saved_tensors
        norm, roe, ree = ctx.cfg
        data = [t.detach() for t in tables]
        grads = [torch.zeros_like(t) for t in data]
        B = users.numel()
        dev = users.device
        # the regulariser's gradient is linear in the upstream scalar: the kernel forms it for a unit
        # coefficient and the scalar is applied on the device (no read-back, no host sync)
        coefs = (0., 0., 0., 1., 0., 0.) if norm == 2 else (0., 0., 0., 0., 1., 0.)
        losses = torch.zeros(6, dtype=torch.float32, device=dev)
        zeros = torch.zeros(B, dtype=torch.float32, device=dev)
        ops.mstep_grad(data, grads, users, items, envs, zeros, None, B, coefs,
                       ops.flags_of(True, False, False, roe, ree), losses, _workspace(dev))
        scale = gs.detach().to(torch.float32)
        for g in grads:
            g.mul_(scale)
        return (None, None, None, None, None, None, *grads)


def classifier_log_softmax(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """LinearLogSoftMaxEnvClassifier.forward (models.py:206-209) on the HIP forward / backward kernels: the rows of
    `x` stand in for the user table (one "user" per row), a row of ones for the item table, so the kernels'
    classifier input Pu*Qi is x itself; alpha = -1 turns the gradient-reversal factor into +1, so d x is the plain
    gradient of the linear map."""
    if x.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ops.InvPrefError('env_classifier: input must be [B, factor_dim]')
    B, D = x.shape
    dev = x.device
    if B == 0:
        return torch.zeros(0, weight.shape[0], dtype=torch.float32, device=dev)
    ids = torch.arange(B, dtype=torch.int64, device=dev)
    zid = torch.zeros(B, dtype=torch.int64, device=dev)
    ones = torch.ones(1, D, dtype=torch.float32, device=dev)
    z1 = torch.zeros(1, D, dtype=torch.float32, device=dev)
    zb = torch.zeros(B, D, dtype=torch.float32, device=dev)       # env-aware user rows: unused values
    _, _, out = InvPrefForward.apply(ids, zid, zid, -1.0, False, x.contiguous().float(), ones, zb, z1,
                                     torch.zeros_like(weight), weight, bias)
    return out


def classifier_reg(norm: int, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """LinearLogSoftMaxEnvClassifier.get_L1_reg / get_L2_reg (models.py:211-217): the regulariser mode of the
    fused kernel over zero stand-in embedding rows, which leaves exactly the classifier's share."""
    dev, D = weight.device, weight.shape[1]
    zid = torch.zeros(1, dtype=torch.int64, device=dev)
    z1 = torch.zeros(1, D, dtype=torch.float32, device=dev)
    zE = torch.zeros_like(weight)
    return InvPrefReg.apply(zid, zid, zid, norm, False, False, z1, z1, z1, z1, zE, weight, bias)


def predict_all_items(user_table, item_table, users, sigmoid: bool):
    return ops.predict(user_table.contiguous(), item_table.contiguous(), users.contiguous(), sigmoid)
