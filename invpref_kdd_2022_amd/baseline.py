"""PureMF baselines on the same fused HIP step.

Drop-in for the reference's ``PureMatrixFactorization`` / ``PureExplicitMatrixFactorization``
(baseline_models.py:12-69, :652-704) and ``Basic{Implicit,Explicit}TrainManager`` /
``BasicUniform*TrainManager`` (train.py:345-481, :1022-1157): same constructor signatures, attribute and
parameter names (``user_emb.weight``, ``item_emb.weight``), loss-dict keys and return shapes.

PureMF is the degenerate case of the InvPref step: with the env-aware tables, ``embed_env`` and the
classifier absent (``INVPREF_PURE_MF``: never loaded, never stored), one environment and coefficients
``(1, 0, 0, 2*L2_coe, 2*L1_coe, alpha=0)`` the InvPref loss IS ``score_loss + L2_coe*L2_reg + L1_coe*L1_reg``
of train.py:389-397 (the InvPref regularisers are normalised by 2*B*D, PureMF's by B*D; see
oracle/oracle.py ``pure_mf_*`` for the CPU statement of the same mapping, pinned by goldens g7).  The
managers reuse the epoch engine of ``train.py`` (row plans, fused Adam, HIP-graph replay, deferred read-backs).
"""
from __future__ import annotations

import math
import os

import torch
from torch import nn

from . import _capi, ops, plan as planlib
from .parallel import RowShard, UserShard
from .train import FlatState, _InvPrefTrainManager, transfer_loss_dict_to_line_str

PURE_LOSS_KEYS = ['score_loss', 'L2_reg', 'L1_reg', 'loss']  # train.py:399-404


class _PureMFBase(nn.Module):
    implicit = True

    def __init__(self, user_num: int, item_num: int, factor_num: int):
        super().__init__()
        self.user_num, self.item_num, self.factor_num = user_num, item_num, factor_num
        self.user_emb = nn.Embedding(user_num, factor_num)
        self.item_emb = nn.Embedding(item_num, factor_num)
        nn.init.normal_(self.user_emb.weight, std=0.01)  # baseline_models.py:23-25
        nn.init.normal_(self.item_emb.weight, std=0.01)
        self._absent = None

    def tables(self):
        return [self.user_emb.weight, self.item_emb.weight]

    # ---- unfused, autograd-capable surface (what the reference's untouched Basic*TrainManager calls).
    # It goes through the InvPref forward / backward kernels with zero stand-ins for the absent tables
    # (allocated on first use; the managers below never need them).
    def _seven(self):
        w = self.user_emb.weight
        if self._absent is None or self._absent[0].device != w.device:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=w.device)  # noqa: E731
            D = self.factor_num
            self._absent = [z(self.user_num, D), z(self.item_num, D), z(1, D), z(1, D), z(1)]
        return [self.user_emb.weight, self.item_emb.weight] + self._absent

    def _scores(self, users_id, items_id):
        from .autograd import InvPrefForward
        inv, _, _ = InvPrefForward.apply(users_id, items_id, torch.zeros_like(users_id), 0., self.implicit, *self._seven())
        return inv

    def forward(self, users_id, items_id, ground_truth=None):  # baseline_models.py:27-37 / :665-674
        final_ratings = self._scores(users_id, items_id)
        if ground_truth is not None:
            return self.loss_func(final_ratings, ground_truth)
        return final_ratings

    def _reg(self, users_id, items_id, norm: int):
        # InvPref's regulariser over (Pu, Pa=0) and (Qi, Qa=0) is normalised by 2*B*D: PureMF's is twice that
        from .autograd import InvPrefReg
        return 2. * InvPrefReg.apply(users_id, items_id, torch.zeros_like(users_id), norm, True, False, *self._seven())

    def get_L1_reg(self, users_id, items_id):  # baseline_models.py:59-60
        return self._reg(users_id, items_id, 1)

    def get_L2_reg(self, users_id, items_id):  # baseline_models.py:62-63
        return self._reg(users_id, items_id, 2)


class PureMatrixFactorization(_PureMFBase):
    """baseline_models.py:12-69"""
    implicit = True

    def __init__(self, user_num: int, item_num: int, factor_num: int):
        super().__init__(user_num, item_num, factor_num)
        self.output_func = nn.Sigmoid()
        self.loss_func = nn.BCELoss()

    def predict(self, users_id):  # baseline_models.py:65-69: sigmoid(Pu[users] @ Qi^T)
        from .autograd import predict_all_items
        return predict_all_items(self.user_emb.weight.detach(), self.item_emb.weight.detach(), users_id, sigmoid=True)


class PureExplicitMatrixFactorization(_PureMFBase):
    """baseline_models.py:652-704"""
    implicit = False

    def __init__(self, user_num: int, item_num: int, factor_num: int):
        super().__init__(user_num, item_num, factor_num)
        self.loss_func = nn.MSELoss()

    def predict(self, users_id, items_id):  # baseline_models.py:703-704
        with torch.no_grad():
            return self._scores(users_id, items_id).reshape(-1)


class _BasicTrainManager(_InvPrefTrainManager):
    """Basic{Implicit,Explicit}TrainManager (train.py:345-461, :1022-1138) on the fused PureMF step."""
    _pure = True
    _make_tables = staticmethod(_capi.make_pure_tables)

    def __init__(self, model, evaluator, device: torch.device, training_data: torch.Tensor, batch_size: int,
                 epochs: int, evaluate_interval: int, lr: float, L2_coe: float, L1_coe: float,
                 test_begin_epoch: int = 0, *, rank=None, world_size=None, process_group=None):
        """rank / world_size / process_group (keyword-only, beyond the reference's signature): one process per GPU,
        sharded like the InvPref managers (parallel.py; INVPREF_SHARD=users|rows): every rank keeps its share of
        every minibatch, one all-reduce per optimiser step, every mean() over the GLOBAL minibatch."""
        if model.implicit != self.implicit:
            raise TypeError(f'{type(self).__name__} needs an {"implicit" if self.implicit else "explicit"} model')
        self.model, self.evaluator, self.device = model, evaluator, torch.device(device)
        self.process_group = process_group
        if world_size is None:
            import torch.distributed as dist
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
            rank = dist.get_rank(process_group) if world_size > 1 else 0
        self.rank, self.world_size = int(rank or 0), int(world_size)
        n_total = training_data.shape[0]
        self.n_total, self.batch_size = n_total, batch_size
        self.batch_num = math.ceil(n_total / batch_size)
        forced = os.environ.get('INVPREF_FORCE_SHARDED_PATH', '0') == '1'
        self.shard_mode = os.environ.get('INVPREF_SHARD', 'rows') if (self.world_size > 1 or forced) else 'rows'
        if self.shard_mode == 'users':
            self.shard = UserShard(training_data[:, 0].cpu().numpy(), n_total, batch_size, model.user_num, self.rank,
                                   self.world_size)
        else:
            self.shard = RowShard(n_total, batch_size, self.rank, self.world_size)
        training_data_full = training_data
        if self.world_size > 1:
            training_data = training_data.index_select(0, self.shard.local_rows().to(training_data.device))
        self.users_tensor = training_data[:, 0].contiguous().to(self.device)
        self.items_tensor = training_data[:, 1].contiguous().to(self.device)
        self.scores_tensor = training_data[:, 2].float().contiguous().to(self.device)
        self.evaluate_interval, self.epochs, self.lr = evaluate_interval, epochs, lr
        self.L2_coe, self.L1_coe = L2_coe, L1_coe
        self.epoch_cnt = 0
        self.test_begin_epoch = test_begin_epoch
        # no environments: the engine's env / weight pointers are never dereferenced under INVPREF_PURE_MF
        self.envs = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.sample_weights = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.alpha, self.update_alpha = 0., False
        self.cluster_interval = 1 << 62
        self.model.to(self.device)
        # (row-sharded: reduce-scatter / slice Adam / all-gather exchange like the InvPref managers, train.py)
        self.exchange = os.environ.get('INVPREF_EXCHANGE', 'allreduce') if self.shard_mode == 'rows' else 'allreduce'
        self.state = FlatState(model.tables(), self.device,     # [user table | item table]: the shared part is last
                               chunks=self.world_size if self.exchange == 'scatter' else 1)
        self._setup_ranges(model)
        if self.exchange == 'packed':
            self._setup_packed(training_data_full[:, 0], training_data_full[:, 1])
        self.workspace = ops.Workspace(self.device)
        self._flags = ops.flags_of(self.implicit, False, False, True, False, dense_reg=False) | _capi.PURE_MF
        self.use_plan, self._plans = True, None
        self._batch_plans, self.planned_batch_steps = {}, 0
        self.use_graph = os.environ.get('INVPREF_NO_GRAPH', '0') != '1'
        import torch.distributed as _dist
        self._force_sharded_path, self._unfused = forced, False
        self._collective_ok = _dist.is_available() and _dist.is_initialized()
        self._graphs, self._graph_warm = {}, False
        self._estep_graphs = {}
        self._grad_stale = False
        self._sched, self._sched_synced = None, False
        self._alt = None

    def _coefs(self, alpha):
        return (1., 0., 0., 2. * self.L2_coe, 2. * self.L1_coe, 0.)

    @staticmethod
    def loss_dicts(dev_losses: torch.Tensor) -> list:
        """InvPref's six outputs -> the four PureMF terms (the regulariser reports are twice InvPref's)."""
        return [dict(zip(PURE_LOSS_KEYS, (v[0], 2. * v[3], 2. * v[4], v[5]))) for v in dev_losses.tolist()]

    def train_a_batch(self, batch_users_tensor, batch_items_tensor, batch_scores_tensor, *args) -> dict:
        """train.py:379-405 on caller-supplied tensors: the row plan of this one minibatch is built on the
        host first (the epoch loop uses the plans prepared once for the static minibatches instead).
        Single-process form; sharded runs go through train_epochs() / train()."""
        if self.world_size > 1:
            raise NotImplementedError('train_a_batch on caller-supplied tensors is single-process; use train_epochs()')
        u = batch_users_tensor.detach().cpu().numpy()
        v = batch_items_tensor.detach().cpu().numpy()
        y = batch_scores_tensor.detach().float().contiguous()
        dp = planlib.upload(planlib.build_row_plan(u, v, y.cpu().numpy(), self.model.user_num, self.model.item_num,
                                                   factor_num=self.model.factor_num, env_num=0), self.device)
        st = self.state
        st.losses6.zero_()
        st.step += 1
        self._sched_synced = False
        ops.mstep_rows_adam(st.p_views, st.p_views_alt, st.m_views, st.v_views, dp, None, y.to(self.device), None,
                            len(u), self._coefs(0.), self._flags, st.losses6, st.step, self.lr, self.workspace,
                            pure=True)
        st.swap()
        return self.loss_dicts(st.losses6[None])[0]

    def train(self, silent: bool = False, auto: bool = False):
        """train.py:428-461: ((loss dicts, epochs), (test results, epochs))."""
        test_result_list, test_epoch_list, loss_result_list, train_epoch_index_list = [], [], [], []

        def evaluate():
            self.sync_parameters()
            res = self.evaluator.evaluate()
            test_result_list.append(res)
            test_epoch_list.append(self.epoch_cnt)
            if not silent and not auto:
                print('test at epoch:', self.epoch_cnt)
                print(transfer_loss_dict_to_line_str(res))

        evaluate()
        defer = bool(silent or auto)
        while self.epoch_cnt < self.epochs:
            first = self.epoch_cnt + 1
            run = self.train_epochs(self._epochs_to_next_event(), sync=not defer)
            for i in range(len(run)):
                train_epoch_index_list.append(first + i)
                loss_result_list.append(run[i])
                if not defer:
                    print('train epoch:', first + i)
                    print(transfer_loss_dict_to_line_str(run[i]))
            if (self.epoch_cnt % self.evaluate_interval) == 0 and self.epoch_cnt >= self.test_begin_epoch:
                evaluate()
        self.sync_parameters()
        self._check_alt_error()   # (PureMF launches wait for nothing inside a launch: the word can only be set by another user of the workspace)
        if defer and loss_result_list:
            loss_result_list = self.loss_dicts(torch.stack(loss_result_list))
        return (loss_result_list, train_epoch_index_list), (test_result_list, test_epoch_list)

    # the InvPref-only parts of the engine do not exist here
    def cluster(self, *a, **k):
        raise AttributeError('PureMF has no environments')

    stat_envs = cluster_a_batch = update_each_env_count = cluster


class BasicImplicitTrainManager(_BasicTrainManager):
    """reference train.py:345-461 (BCELoss)"""
    implicit = True


class BasicExplicitTrainManager(_BasicTrainManager):
    """reference train.py:1022-1138 (MSELoss)"""
    implicit = False


class _UniformMixin:
    def _keep_uniform(self, uniform_data):  # train.py:478-481 / :1154-1157: stored, not used by the loop
        self.uniform_user = uniform_data[:, 0].to(self.device).long()
        self.uniform_item = uniform_data[:, 1].to(self.device).long()
        self.uniform_score = uniform_data[:, 2].to(self.device).float()


class BasicUniformImplicitTrainManager(BasicImplicitTrainManager, _UniformMixin):
    """reference train.py:464-481"""

    def __init__(self, model, evaluator, device, training_data, uniform_data, batch_size, epochs, evaluate_interval,
                 lr, L2_coe, L1_coe, test_begin_epoch: int = 0):
        super().__init__(model, evaluator, device, training_data, batch_size, epochs, evaluate_interval, lr, L2_coe,
                         L1_coe, test_begin_epoch)
        self._keep_uniform(uniform_data)


class BasicUniformExplicitTrainManager(BasicExplicitTrainManager, _UniformMixin):
    """reference train.py:1140-1157"""

    def __init__(self, model, evaluator, device, training_data, uniform_data, batch_size, epochs, evaluate_interval,
                 lr, L2_coe, L1_coe, test_begin_epoch: int = 0):
        super().__init__(model, evaluator, device, training_data, batch_size, epochs, evaluate_interval, lr, L2_coe,
                         L1_coe, test_begin_epoch)
        self._keep_uniform(uniform_data)
