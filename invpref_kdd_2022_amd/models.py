"""Drop-in ``InvPrefImplicit`` / ``InvPrefExplicit`` modules (reference models.py:272-411, :414-543).

Same constructor signature, attribute names and parameter names as the reference, so
``state_dict``s interchange.  The arithmetic is NOT done with torch ops: ``forward`` runs the
hand-written HIP forward kernel and, when autograd needs it, the HIP backward kernel
(``invpref_backward_hip``); the train managers bypass this unfused surface entirely and call the
fused M-step.  The parameters are ordinary ``nn.Parameter``s so ``.to(device)``, ``.parameters()``
and optimisers keep working.
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops


class LinearLogSoftMaxEnvClassifier(nn.Module):
    """The D->E environment classifier (reference models.py:197-220).  Inside InvPref*.forward it is fused into
    the forward / M-step kernels; called on its own it runs on the same HIP kernels (autograd.py)."""

    def __init__(self, factor_dim: int, env_num: int):
        super().__init__()
        self.linear_map = nn.Linear(factor_dim, env_num)
        nn.init.xavier_uniform_(self.linear_map.weight)
        self.elements_num = float(factor_dim * env_num)
        self.bias_num = float(env_num)

    def forward(self, invariant_preferences: torch.Tensor) -> torch.Tensor:  # models.py:206-209
        from .autograd import classifier_log_softmax
        return classifier_log_softmax(invariant_preferences, self.linear_map.weight, self.linear_map.bias)

    def get_L1_reg(self) -> torch.Tensor:  # models.py:211-213
        from .autograd import classifier_reg
        return classifier_reg(1, self.linear_map.weight, self.linear_map.bias)

    def get_L2_reg(self) -> torch.Tensor:  # models.py:215-217
        from .autograd import classifier_reg
        return classifier_reg(2, self.linear_map.weight, self.linear_map.bias)


class _InvPrefBase(nn.Module):
    implicit = True

    def __init__(self, user_num: int, item_num: int, env_num: int, factor_num: int, reg_only_embed: bool = False,
                 reg_env_embed: bool = True):
        super().__init__()
        self.user_num, self.item_num, self.env_num, self.factor_num = user_num, item_num, env_num, factor_num
        self.embed_user_invariant = nn.Embedding(user_num, factor_num)
        self.embed_item_invariant = nn.Embedding(item_num, factor_num)
        self.embed_user_env_aware = nn.Embedding(user_num, factor_num)
        self.embed_item_env_aware = nn.Embedding(item_num, factor_num)
        self.embed_env = nn.Embedding(env_num, factor_num)
        self.env_classifier = LinearLogSoftMaxEnvClassifier(factor_num, env_num)
        self.reg_only_embed = reg_only_embed
        self.reg_env_embed = reg_env_embed
        self._init_weight()

    def _init_weight(self):  # models.py:300-305
        for emb in (self.embed_user_invariant, self.embed_item_invariant, self.embed_user_env_aware,
                    self.embed_item_env_aware, self.embed_env):
            nn.init.normal_(emb.weight, std=0.01)

    # ---- the seven tensors in state_dict order
    def tables(self):
        return [self.embed_user_invariant.weight, self.embed_item_invariant.weight,
                self.embed_user_env_aware.weight, self.embed_item_env_aware.weight, self.embed_env.weight,
                self.env_classifier.linear_map.weight, self.env_classifier.linear_map.bias]

    def _data(self):
        return [p.detach() for p in self.tables()]

    def forward(self, users_id, items_id, envs_id, alpha):
        """-> (invariant_score[B], env_aware_score[B], env_outputs[B, E])   models.py:307-326 / :448-467"""
        from .autograd import InvPrefForward
        return InvPrefForward.apply(users_id, items_id, envs_id, float(alpha), self.implicit, *self.tables())

    def cluster_predict(self, users_id, items_id, envs_id) -> torch.Tensor:  # models.py:409-411
        _, env_aware_score, _ = self.forward(users_id, items_id, envs_id, 0.)
        return env_aware_score

    def get_L2_reg(self, users_id, items_id, envs_id):  # models.py:368-379
        from .autograd import InvPrefReg
        return InvPrefReg.apply(users_id, items_id, envs_id, 2, self.reg_only_embed, self.reg_env_embed,
                                *self.tables())

    def get_L1_reg(self, users_id, items_id, envs_id):  # models.py:381-391
        from .autograd import InvPrefReg
        return InvPrefReg.apply(users_id, items_id, envs_id, 1, self.reg_only_embed, self.reg_env_embed,
                                *self.tables())


class InvPrefImplicit(_InvPrefBase):
    """reference models.py:272-411"""
    implicit = True

    def predict(self, users_id):  # models.py:393-407: sigmoid(Pu[users] @ Qi^T) -> [n, item_num]
        from .autograd import predict_all_items
        return predict_all_items(self.embed_user_invariant.weight.detach(), self.embed_item_invariant.weight.detach(),
                                 users_id, sigmoid=True)


class InvPrefExplicit(_InvPrefBase):
    """reference models.py:414-543"""
    implicit = False

    def predict(self, users_id, items_id):  # models.py:534-539
        envs = torch.zeros_like(users_id)
        inv, _, _ = ops.forward(self._data(), users_id.contiguous(), items_id.contiguous(), envs, False)
        return inv.reshape(-1)
