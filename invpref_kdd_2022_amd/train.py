"""Drop-in ``ImplicitTrainManager`` / ``ExplicitTrainManager`` (reference train.py:16-342, :693-1019).

Same constructor signature, public attributes, method names and return shapes as the reference;
the work inside is re-designed for MI355X:

* the seven parameter tensors are re-homed as views of ONE flat fp32 buffer (with flat gradient
  and Adam-moment twins), so the optimiser is a single fused HIP launch and multi-GPU needs a
  single RCCL all-reduce per step;
* ``train_a_batch`` = fused M-step HIP kernel (forward + 3 losses + 2 regularisers + analytic
  backward + scatter-add) -> [all-reduce] -> fused zero_grad+Adam kernel;
* ``cluster`` + ``stat_envs`` = one E-step launch over all local interactions + one finish launch;
* no per-batch host syncs: ``train_a_epoch`` reads the loss terms back once per epoch.

Sharding (parallel.py): with ``world_size`` G > 1 every rank keeps only its share of
every minibatch ``[kB,(k+1)B)`` -- by default (round 6: what BASELINE.json's north_star names) a contiguous row
slice, parameters replicated, ONE all-reduce of the flat gradient per optimiser step (INVPREF_EXCHANGE=scatter |
packed: the same sums as reduce-scatter + all-gather / over the touched rows only); with INVPREF_SHARD=users the
interactions of the users it owns (only the item-side gradient is all-reduced).
Every mean() keeps the GLOBAL batch length as denominator, so summing the per-rank gradient
buffers reproduces the single-GPU gradient.
"""
from __future__ import annotations

import itertools
import math
import os
import time
from typing import Optional

import numpy as np
import torch

import ctypes as C

from . import _capi, ops
from . import plan as planlib
from .parallel import (RowShard, UserShard, all_gather_chunks_, all_reduce_min_, all_reduce_sum_,
                       reduce_scatter_sum_)

LOSS_KEYS = ('invariant_loss', 'env_aware_loss', 'envs_loss', 'L2_reg', 'L1_reg', 'loss')
_ALIGN = 64  # floats; every table starts on a 256-byte boundary of the flat buffer


def transfer_loss_dict_to_line_str(d: dict) -> str:
    return ', '.join(f'{k}: {v}' for k, v in d.items())  # utils.py:254-260


class FlatState:
    """params / grads / exp_avg / exp_avg_sq as four flat buffers + per-table views."""

    def __init__(self, tabs, device, order=None, chunks: int = 1):
        """tabs: the model's parameter tensors (InvPref: the seven of state_dict order; PureMF: two).
        order: physical placement of the tables inside the flat buffers (default: as listed).  A user-sharded
        run puts the two user tables first, so that everything the ranks share is ONE contiguous range.
        chunks: the buffers' capacity `cap` is padded so that it splits into that many equal pieces of whole
        256-byte lines (the reduce-scatter / all-gather exchange of a row-sharded run); `n` stays the used length."""
        tabs = list(tabs)
        self.shapes = [tuple(p.shape) for p in tabs]
        self.offsets = [0] * len(tabs)
        off = 0
        for i in (range(len(tabs)) if order is None else order):
            self.offsets[i] = off
            off += (tabs[i].numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n = off
        piece = max(1, chunks) * _ALIGN
        self.cap = cap = (off + piece - 1) // piece * piece
        self.param_full = torch.zeros(cap, dtype=torch.float32, device=device)
        self.param = self.param_full[:self.n]
        # +8: the six loss partials ride at the tail of the gradient buffer so that ONE all-reduce
        # carries everything; they sit behind the padding: [0, n) gradient, [cap, cap + 6) losses
        self.grad_full = torch.zeros(cap + 8, dtype=torch.float32, device=device)
        self.grad_ext = self.grad_full          # gradient | padding (zeros) | loss tail: one contiguous range
        self.grad = self.grad_full[:self.n]
        self.losses6 = self.grad_full[cap:cap + 6]
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=device)
        # double buffer for the fused M-step+Adam pass (reads old rows, writes new rows)
        self.param_alt = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.p_views = self._views(self.param)
        self.p_views_alt = self._views(self.param_alt)
        self.g_views = self._views(self.grad)
        self.m_views = self._views(self.exp_avg)
        self.v_views = self._views(self.exp_avg_sq)
        self._tabs = tabs
        with torch.no_grad():
            for p, view in zip(tabs, self.p_views):
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view  # the module's parameters now alias the flat buffer
        self.step = 0

    def swap(self):
        """the fused pass wrote the new parameters into the alternate buffer: make it current and
        re-point the module's parameters at it (metadata only, no copy)."""
        self.param, self.param_alt = self.param_alt, self.param
        self.p_views, self.p_views_alt = self.p_views_alt, self.p_views
        for p, view in zip(self._tabs, self.p_views):
            p.data = view

    def _views(self, flat):
        return [flat[o:o + int(np.prod(s))].view(s) for o, s in zip(self.offsets, self.shapes)]


class _InvPrefTrainManager:
    implicit = True
    _pure = False                       # PureMF managers (baseline.py) reuse the epoch engine below
    _make_tables = staticmethod(_capi.make_tables)

    def __init__(
            self, model, evaluator, device: torch.device, training_data: torch.Tensor, batch_size: int,
            epochs: int, cluster_interval: int, evaluate_interval: int, lr: float,
            invariant_coe: float, env_aware_coe: float, env_coe: float, L2_coe: float, L1_coe: float,
            alpha: float = None, use_class_re_weight: bool = False, test_begin_epoch: int = 0,
            begin_cluster_epoch: int = None, stop_cluster_epoch: int = None, cluster_use_random_sort: bool = True,
            use_recommend_re_weight: bool = True, *, rank: Optional[int] = None, world_size: Optional[int] = None,
            process_group=None,
    ):
        if model.implicit != self.implicit:
            raise TypeError(f'{type(self).__name__} needs an {"implicit" if self.implicit else "explicit"} model')
        self.model = model
        self.evaluator = evaluator
        self.envs_num: int = model.env_num
        self.device = torch.device(device)
        self.process_group = process_group
        if world_size is None:
            import torch.distributed as dist
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
            rank = dist.get_rank(process_group) if world_size > 1 else 0
        self.rank, self.world_size = int(rank or 0), int(world_size)

        n_total = training_data.shape[0]
        self.n_total = n_total
        self.batch_size = batch_size
        self.batch_num = math.ceil(n_total / batch_size)
        # multi-GPU: the literal row split of SURVEY 8(e) / north_star by default (parallel.RowShard, parameters replicated);
        # INVPREF_SHARD=users partitions the users instead (parallel.UserShard: only the item-side gradient is all-reduced)
        # (testing aid INVPREF_FORCE_SHARDED_PATH=1: the sharded step sequence, either layout, on one rank)
        forced = os.environ.get('INVPREF_FORCE_SHARDED_PATH', '0') == '1'
        self.shard_mode = os.environ.get('INVPREF_SHARD', 'rows') if (self.world_size > 1 or forced) else 'rows'
        if self.shard_mode not in ('users', 'rows'):
            raise ValueError('INVPREF_SHARD must be "users" or "rows"')
        if self.shard_mode == 'users':
            self.shard = UserShard(training_data[:, 0].cpu().numpy(), n_total, batch_size, model.user_num, self.rank,
                                   self.world_size)
        else:
            self.shard = RowShard(n_total, batch_size, self.rank, self.world_size)
        rows = self.shard.local_rows().to(training_data.device)
        td = training_data.index_select(0, rows) if self.world_size > 1 else training_data
        # full tensors keep the reference attribute names; *_local are what the kernels read
        self.users_tensor = td[:, 0].contiguous().to(self.device)
        self.items_tensor = td[:, 1].contiguous().to(self.device)
        self.scores_tensor = td[:, 2].float().contiguous().to(self.device)
        # initial environments: one numpy draw over ALL rows, like the reference (train.py:34),
        # then this rank keeps its rows -> identical to the single-GPU run for any world size
        envs_all = torch.from_numpy(np.random.randint(0, self.envs_num, n_total).astype(np.int64))
        self.envs = (envs_all.index_select(0, rows.cpu()) if self.world_size > 1 else envs_all).to(self.device)

        self.cluster_interval, self.evaluate_interval, self.epochs = cluster_interval, evaluate_interval, epochs
        self.lr = lr
        self.invariant_coe, self.env_aware_coe, self.env_coe = invariant_coe, env_aware_coe, env_coe
        self.L2_coe, self.L1_coe = L2_coe, L1_coe
        self.epoch_cnt = 0
        self.each_env_count = dict()
        if alpha is None:
            self.alpha, self.update_alpha = 0., True
        else:
            self.alpha, self.update_alpha = alpha, False
        self.use_class_re_weight = use_class_re_weight
        self.use_recommend_re_weight = use_recommend_re_weight
        # sample_weights[i] = class_weights[envs[i]] (train.py:274-278).  The N-length array is kept LAZILY on one GPU: stat_envs()
        # refreshes class_weights and the planned M-step kernels look the weight up by environment (INVPREF_WEIGHTS_BY_ENV);
        # the array is materialised when somebody reads the `sample_weights` attribute (property below).
        self._sample_weights = torch.zeros(self.users_tensor.shape[0], dtype=torch.float32, device=self.device)
        self._sw_lazy = False        # True: _sample_weights is behind (it equals class_weights[envs] once materialised)
        self._by_env = False         # True: class_weights[envs] IS what the reference's sample_weights would hold right now
        self._epoch_by_env = False   # what the epochs being issued / captured use (fixed per _enqueue_epochs call)
        self._counts_dev = torch.zeros(self.envs_num, dtype=torch.int64, device=self.device)
        self._es = None              # ops.EstepState of the fused E-step (one GPU)
        self.class_weights = torch.zeros(self.envs_num, dtype=torch.float32, device=self.device)
        self.test_begin_epoch = test_begin_epoch
        self.begin_cluster_epoch, self.stop_cluster_epoch = begin_cluster_epoch, stop_cluster_epoch
        self.cluster_use_random_sort = cluster_use_random_sort
        self._eps_base = np.array([1e-10 * (1e-1 ** i) for i in range(self.envs_num)], dtype=np.float32)
        self._eps_rows_cnt = math.factorial(self.envs_num)

        self.model.to(self.device)
        # user-sharded: [Pu | Pa | Qi | Qa | Ev | W | b]: the replicated part (items + small tables) and the loss
        # tail form one contiguous range for the all-reduce, the owned user rows two ranges for Adam
        # row-sharded exchange (INVPREF_EXCHANGE): "scatter" = reduce-scatter of the flat gradient, Adam on this rank's 1/G
        # slice of the flat buffers, all-gather of the new parameters (same bytes on the wire as the all-reduce, the dense
        # Adam stream cut G-fold); "allreduce" = SURVEY 8(e) as written: every rank reduces and updates everything; "packed" =
        # the all-reduce over the rows the GLOBAL minibatch touches only (+ the small tables): every other gradient row is
        # zero on every rank, so nothing else needs the wire (_setup_packed)
        self.exchange = os.environ.get('INVPREF_EXCHANGE', 'allreduce') if self.shard_mode == 'rows' else 'allreduce'
        if self.exchange not in ('scatter', 'allreduce', 'packed'):
            raise ValueError('INVPREF_EXCHANGE must be "scatter", "allreduce" or "packed"')
        self.state = FlatState(model.tables(), self.device,
                               order=[0, 2, 1, 3, 4, 5, 6] if self.shard_mode == 'users' else None,
                               chunks=self.world_size if self.exchange == 'scatter' else 1)
        self._setup_ranges(model)
        if self.exchange == 'packed':
            self._setup_packed(training_data[:, 0], training_data[:, 1])
        self.workspace = ops.Workspace(self.device)
        self._flags = ops.flags_of(self.implicit, use_recommend_re_weight, use_class_re_weight,
                                   model.reg_only_embed, model.reg_env_embed, dense_reg=(self.rank == 0))
        # atomic-free planned M-step (plan.py) unless INVPREF_NO_PLAN=1 (then: float-atomic scatter-add)
        self.use_plan = os.environ.get('INVPREF_NO_PLAN', '0') != '1'
        # (measured, tools/kbench.py, planned fused step vs plan-free gradient + Adam: Yahoo class 19 vs 78 us,
        #  MovieLens class -- E = 8, D = 128, 65 536 interactions -- 110 vs 181 us, MIND class -- E = 16, D = 256,
        #  262 144 interactions -- 1.1 vs 1.66 ms: the plan wins everywhere, so it is the default for every shape)
        # INVPREF_UNFUSED=1: gradient pass + flat Adam kernel instead of the fused pass on one GPU (the sequence a sharded
        # rank runs, without the exchange); every row length takes the fused pass by default
        self._unfused = self.use_plan and os.environ.get('INVPREF_UNFUSED', '') == '1'
        self._plans = None
        self._batch_plans, self.planned_batch_steps = {}, 0   # train_a_batch on caller tensors: see _cached_batch_plan
        self._batch_plans_foreign = {}
        # runs of whole epochs as one HIP graph launch (single GPU, planned path); INVPREF_NO_GRAPH=1 disables
        self.use_graph = os.environ.get('INVPREF_NO_GRAPH', '0') != '1'
        # testing aid: run the multi-GPU step sequence (gradient pass -> all-reduce -> stand-alone Adam) on one rank
        self._force_sharded_path = os.environ.get('INVPREF_FORCE_SHARDED_PATH', '0') == '1'
        import torch.distributed as _dist
        self._collective_ok = _dist.is_available() and _dist.is_initialized()
        self._graphs, self._graph_warm = {}, False
        self._estep_graphs = {}
        self._grad_stale = False
        self._sched = None
        self._sched_synced = False
        self._alt = None

    # ------------------------------------------------------------------ sample weights (train.py:67, :274-278)
    @property
    def sample_weights(self) -> torch.Tensor:
        """The reference's attribute.  Read from outside, the array is brought up to date first; and because the caller may
        write into it (the reference's loop only slices it), the epochs fall back to reading it per interaction until the next
        stat_envs() re-establishes sample_weights == class_weights[envs]."""
        self._materialise_sample_weights()
        self._by_env = False
        return self._sample_weights

    @sample_weights.setter
    def sample_weights(self, value: torch.Tensor):
        self._sample_weights, self._sw_lazy, self._by_env = value, False, False

    def _materialise_sample_weights(self):
        if self._sw_lazy:
            _, sw = ops.sample_weights(self.envs, self._counts_dev, self.n_total, self.envs_num)   # class_weights[envs]
            self._sample_weights.copy_(sw)
            self._sw_lazy = False

    def _step_weights(self, bw):
        """(weights argument, flags) of one of the epochs' planned launches: the minibatch's slice of the sample-weight array, or
        -- INVPREF_WEIGHTS_BY_ENV -- the E class weights the kernel indexes by environment"""
        if bw is not None and self._epoch_by_env:
            return self.class_weights, self._flags | _capi.WEIGHTS_BY_ENV
        return bw, self._flags

    def _weights_by_env(self) -> bool:
        """may the epochs' launches take an interaction's weight as class_weights[env]? (one GPU, planned M-step, weights
        consistent with the environments: stat_envs() ran since they last changed, nobody was handed the array since)"""
        return bool(self._by_env and self.world_size == 1 and self.use_plan and not self._pure
                    and os.environ.get('INVPREF_WEIGHTS_BY_ENV', '1') == '1')

    def _setup_ranges(self, model):
        """What one optimiser step exchanges and updates on this rank: `_ar_lo` = first float of the flat
        gradient that is all-reduced (through the loss tail), `_adam_ranges` = (offset, length) pieces of the
        flat buffers this rank applies Adam to."""
        st = self.state
        # tables indexed by user: InvPref keeps two (invariant, env-aware), PureMF one
        self._user_tables = (0,) if self._pure else (0, 2)
        if self.shard_mode == 'users':
            lo, hi = self.shard.user_range(model.user_num)
            D = model.factor_num
            self._ar_lo = st.offsets[1]                      # the item table that follows the user tables: start of the replicated part
            self._adam_ranges = [(st.offsets[i] + lo * D, (hi - lo) * D) for i in self._user_tables] + \
                                [(self._ar_lo, st.n - self._ar_lo)]
            self._adam_ranges = [r for r in self._adam_ranges if r[1] > 0]
        else:
            self._ar_lo = 0
            self._adam_ranges = [(0, st.n)]
            if self.exchange == 'scatter' and self.world_size > 1:
                chunk = st.cap // self.world_size          # whole 256-byte lines (FlatState.cap)
                lo = min(self.rank * chunk, st.n)
                self._adam_ranges = [(lo, min(chunk, st.n - lo))] if st.n > lo else []

    def _setup_packed(self, users_all, items_all):
        """The packed exchange of a row-sharded run (INVPREF_EXCHANGE=packed): the minibatches are static (utils.mini_batch,
        utils.py:12-19), so the rows of the four big tables that the GLOBAL minibatch k touches -- on any rank -- are known
        up front.  Every other row's gradient is zero everywhere (the planned pass writes zeros there, the plan-free one
        leaves the zeroed buffer alone; the dense regulariser only reaches the classifier), so the step all-reduces
        [touched rows | small tables] through one packed buffer instead of the whole flat gradient.  Same sums, same Adam:
        results equal the "allreduce" exchange up to the order in which the collective adds the ranks' terms.
        Worth it when a global minibatch touches a small share of the rows (a strong-scaling split of one 8 192-row Yahoo
        minibatch: 2.7 x fewer bytes); a weak-scaling run at 8 ranks touches nearly every user row."""
        st, D = self.state, self.model.factor_num
        u = users_all.cpu().numpy().astype(np.int64)
        v = items_all.cpu().numpy().astype(np.int64)
        n_tabs = len(st.shapes)
        user_tabs = self._user_tables
        item_tabs = tuple(i + 1 for i in user_tabs)                     # (state_dict order: every user table is followed by its item table)
        big = sorted(user_tabs + item_tabs)
        small = [i for i in range(n_tabs) if i not in big]
        if small and min(st.offsets[i] for i in small) < max(st.offsets[i] for i in big):
            raise ValueError('packed exchange: the small tables must follow the big ones in the flat buffers')
        tail_off = min((st.offsets[i] for i in small), default=st.n)
        self._packed_tail = (int(tail_off), int(st.n - tail_off))
        self._packed_rows, most = [], 0
        for k in range(self.batch_num):
            lo, hi = k * self.batch_size, min((k + 1) * self.batch_size, self.n_total)
            tu, ti = np.unique(u[lo:hi]), np.unique(v[lo:hi])
            offs = np.concatenate([st.offsets[i] + tu * D for i in user_tabs] + [st.offsets[i] + ti * D for i in item_tabs])
            self._packed_rows.append(torch.from_numpy(np.sort(offs).astype(np.int64)).to(self.device))
            most = max(most, len(offs))
        self._packed_vec = D % 4 == 0 and all(st.offsets[i] % 4 == 0 for i in big)
        self._packed_buf = torch.zeros(most * D + self._packed_tail[1] + 4, dtype=torch.float32, device=self.device)
        self.packed_floats = [int(r.numel()) * D + self._packed_tail[1] for r in self._packed_rows]   # on the wire, per minibatch

    def _exchange_gradient(self, k: Optional[int] = None):
        """the step's exchange in front of Adam (also on the 1-rank group of the forced sharded path: the same
        collectives, captured and replayed like on N ranks); k: the minibatch (the packed exchange's row list)"""
        st = self.state
        if self.exchange == 'scatter':
            reduce_scatter_sum_(st.grad_full[:st.cap], self.rank, self.world_size, self.process_group)
        elif self.exchange == 'packed' and k is not None:
            rows, (t_off, t_len) = self._packed_rows[k], self._packed_tail
            buf = self._packed_buf[:self.packed_floats[k]]
            D = self.model.factor_num
            ops.pack_rows(st.grad, rows, D, t_off, t_len, buf, self._packed_vec)
            all_reduce_sum_(buf, self.process_group)
            ops.unpack_rows(st.grad, rows, D, t_off, t_len, buf, self._packed_vec)
        else:
            all_reduce_sum_(st.grad[self._ar_lo:], self.process_group)

    def _exchange_parameters(self):
        """the step's exchange behind Adam (scatter mode: every rank updated its own slice of the flat buffer)"""
        if self.exchange == 'scatter':
            all_gather_chunks_(self.state.param_full, self.rank, self.world_size, self.process_group)

    def sync_parameters(self) -> None:
        """User-sharded runs: bring every rank's copy of the user tables up to date (each rank contributes its
        own rows; one all-reduce per table).  Called before evaluations and at the end of train(); a no-op
        otherwise.  The Adam moments of foreign rows are not exchanged (nothing reads them)."""
        if self.world_size == 1 or self.shard_mode != 'users':
            return
        lo, hi = self.shard.user_range(self.model.user_num)
        for i in self._user_tables:
            view = self.state.p_views[i]
            tmp = torch.zeros_like(view)
            tmp[lo:hi] = view[lo:hi]
            all_reduce_sum_(tmp, self.process_group)
            view.copy_(tmp)

    def _all_reduce_step(self):
        """the one exchange of an optimiser step: the shared part of the flat gradient + the loss tail"""
        all_reduce_sum_(self.state.grad_ext[self._ar_lo:], self.process_group)

    # ------------------------------------------------------------------ M-step
    def _coefs(self, alpha):
        return (self.invariant_coe, self.env_aware_coe, self.env_coe, self.L2_coe, self.L1_coe, alpha)

    def _step(self, users, items, scores, envs, weights, alpha, batch_norm: int, losses6=None):
        """one optimiser step on explicit tensors; the six loss terms of this step are ADDED into
        `losses6` (default: state.losses6, zeroed first)."""
        st = self.state
        if losses6 is None:
            losses6 = st.losses6
            losses6.zero_()
        if self._grad_stale:
            # a planned gradient pass (multi-GPU / D > 128 path) OVERWRITES its rows and its Adam does not zero
            # them: the buffer still holds that step's (all-reduced) gradient, and the plan-free kernel below ADDS
            st.grad.zero_()
            self._grad_stale = False
        ops.mstep_grad(st.p_views, st.g_views, users, items, envs, scores, weights, batch_norm, self._coefs(alpha),
                       self._flags, losses6, self.workspace)
        if self.world_size > 1:
            self._all_reduce_step()
        st.step += 1
        self._sched_synced = False
        for o, n in self._adam_ranges:
            ops.adam_(st.param[o:o + n], st.grad[o:o + n], st.exp_avg[o:o + n], st.exp_avg_sq[o:o + n], st.step, self.lr,
                      zero_grad=True)
        if self.world_size > 1:
            if self.exchange == 'scatter':
                st.grad.zero_()             # (only this rank's slice was cleared by its Adam)
            self._exchange_parameters()     # scatter mode: every rank updated its own slice of the flat buffer

    def train_a_batch(self, batch_users_tensor, batch_items_tensor, batch_scores_tensor, batch_envs_tensor,
                      batch_sample_weights, alpha) -> dict:
        """train.py:94-167.  In a sharded run the arguments are this rank's share of the batch (user-sharded:
        the interactions of the users this rank owns)."""
        assert batch_users_tensor.shape == batch_items_tensor.shape == batch_scores_tensor.shape \
            == batch_envs_tensor.shape
        bn = batch_users_tensor.shape[0]
        if self.world_size > 1:
            t = torch.tensor([bn], dtype=torch.int64, device=self.device)
            all_reduce_sum_(t, self.process_group)
            bn = int(t.item())
        dp = self._cached_batch_plan(batch_users_tensor, batch_items_tensor, batch_scores_tensor)
        if dp is not None:
            # the reference's own loop (`for batch in mini_batch(...)`, train.py:204-233) hands over the same slices of
            # the same resident tensors every epoch: from their second sighting they run the planned fused step
            st = self.state
            st.losses6.zero_()
            st.step += 1
            self._sched_synced = False
            ops.mstep_rows_adam(st.p_views, st.p_views_alt, st.m_views, st.v_views, dp, batch_envs_tensor.contiguous(),
                                batch_scores_tensor.float().contiguous(), batch_sample_weights.contiguous(), bn,
                                self._coefs(alpha), self._flags, st.losses6, st.step, self.lr, self.workspace,
                                pure=self._pure)
            st.swap()
            self.planned_batch_steps += 1
        else:
            self._step(batch_users_tensor.contiguous(), batch_items_tensor.contiguous(),
                       batch_scores_tensor.float().contiguous(), batch_envs_tensor.contiguous(),
                       batch_sample_weights.contiguous(), alpha, bn)
        vals = self.state.losses6.tolist()
        return dict(zip(LOSS_KEYS, vals))

    # row plans of caller-supplied minibatches, keyed by the identity of the tensors they were built from: (data_ptr,
    # length, _version) of users / items / scores -- a slice of a resident tensor that nobody has written to keeps all three
    _BATCH_PLAN_CACHE_MAX = 4096
    _BATCH_PLAN_CACHE_MAX_FOREIGN = 128

    def _cached_batch_plan(self, users, items, scores):
        if not (self.use_plan and self.world_size == 1 and not self._unfused and not self._force_sharded_path
                and users.is_cuda and users.numel() > 0 and users.is_contiguous() and items.is_contiguous()
                and os.environ.get('INVPREF_NO_BATCH_PLAN_CACHE', '0') != '1'):
            return None
        key = tuple(x for t in (users, items, scores) for x in (t.data_ptr(), t.numel(), t._version, t.dtype))
        # The identity of a tensor (address, length, version counter) names its CONTENT only while the memory is the
        # manager's own: a slice of the resident interaction arrays (what the reference's loop hands over, train.py:204-233)
        # cannot be freed and reallocated behind the manager's back.  Any other tensor -- a freshly allocated batch comes
        # back from the caching allocator at the same address with version 0 -- is keyed by an order-sensitive checksum of its
        # ids and scores as well (three small kernels and a read-back per step), is planned at its SECOND sighting only,
        # and shares a small cache.
        resident = all(any(t.untyped_storage().data_ptr() == r.untyped_storage().data_ptr()
                           for r in (self.users_tensor, self.items_tensor, self.scores_tensor)) for t in (users, items, scores))
        if not resident:
            n = users.numel()
            pos = getattr(self, '_batch_pos', None)
            if pos is None or pos.numel() < n:
                pos = self._batch_pos = torch.arange(max(n, 1 << 16), dtype=torch.int64, device=users.device) * 2654435761 + 1
            # (a multiplicative mix of the three words per interaction, xor-folded: not linear in (u, i) as u * a + i * b was)
            mix = (users.to(torch.int64) * -7046029254386353131) ^ (items.to(torch.int64) * -4417276706812531889 + 0x2545F491) ^ \
                (scores.float().contiguous().view(torch.int32).to(torch.int64) * 40503)
            mix = mix ^ (mix >> 29)
            # (the content IS the key: addresses and version counters of foreign tensors say nothing)
            key = ('foreign', n, users.dtype, items.dtype, int((mix * pos[:n]).sum().item()))
        # (foreign minibatches have a small cache of their own -- filling it never evicts the resident slices' plans -- and a
        #  hit is trusted only after the ids and labels kept with the plan compare equal: the checksum is a filter, not an
        #  identity -- ADVICE r05)
        cache = self._batch_plans if resident else self._batch_plans_foreign
        hit = cache.get(key)
        if hit is not None and not resident and hit != 1:
            hit, ku, ki, ky = hit
            if not (torch.equal(ku, users) and torch.equal(ki, items) and torch.equal(ky, scores.float())):
                hit = 1                         # a checksum collision: plan this content afresh, replacing the entry
        if hit is None:
            if len(cache) >= (self._BATCH_PLAN_CACHE_MAX if resident else self._BATCH_PLAN_CACHE_MAX_FOREIGN):
                cache.clear()
            # first sighting: with the native builder a plan costs about a millisecond (and the ids' trip to the host), so it
            # is made at once -- INVPREF_BATCH_PLAN_AT=2 waits for the second sighting (a caller that never repeats a
            # minibatch pays nothing for plans it would not reuse) and runs the first one plan-free
            if os.environ.get('INVPREF_BATCH_PLAN_AT', '1') != '1' or planlib._native_lib() is None or not resident:
                cache[key] = 1
                return None
            hit = 1
        if hit == 1:                            # invert the scatter pattern once (host side)
            hit = planlib.upload(planlib.build_row_plan(users.cpu().numpy(), items.cpu().numpy(),
                                                        scores.float().cpu().numpy(), self.model.user_num,
                                                        self.model.item_num, factor_num=self.model.factor_num,
                                                        env_num=getattr(self.model, 'env_num', 0)),
                                 self.device)
            cache[key] = hit if resident else (hit, users.clone(), items.clone(), scores.float().clone())
        return hit

    # ---- the epoch loop: every per-minibatch argument (views of the resident interaction arrays, the row plan,
    #      the zero-initialised scratch) is prepared once; a step is one or two torch.ops.invpref.* calls
    #      (+ the collective in a sharded run) and nothing is read back
    def _raw_setup(self):
        st = self.state
        L = _capi.lib()
        # one graph replays up to _graph_epochs epochs (fewer, longer launches: the GPU idles ~60 us
        # between two replays); each epoch of a replay writes its own [batch_num, 6] slice of the loss buffer
        self._graph_epochs = max(1, min(8, 2048 // self.batch_num))
        self._epoch_losses = torch.zeros(self._graph_epochs, self.batch_num, 6, dtype=torch.float32, device=self.device)
        self._loss_slot = 0
        self._raw_ptrs = (self.users_tensor.data_ptr(), self.items_tensor.data_ptr(), self.envs.data_ptr(),
                          self.scores_tensor.data_ptr(), self._sample_weights.data_ptr())
        self._raw_batches = []
        for k in range(self.batch_num):
            lo, hi = self.shard.local_batch_bounds(k)
            v = None if self._pure else (self.envs[lo:hi], self._sample_weights[lo:hi])
            self._raw_batches.append((lo, hi - lo, self.shard.global_batch_len(k), self.users_tensor[lo:hi],
                                      self.items_tensor[lo:hi], self.scores_tensor[lo:hi],
                                      None if self._pure else v[0], None if self._pure else v[1]))
        if not hasattr(self, '_adam_ranges'):
            self._adam_ranges = [(0, st.n)]
        rg = self._adam_ranges
        self._adam_ranges_ok = len(rg) <= 4 and all(o % 4 == 0 and n % 4 == 0 for o, n in rg)   # one ranged launch can do it
        self._adam_ranges_one = len(rg) > 1 and self._adam_ranges_ok
        if self.use_plan and self._plans is None:
            t0 = time.perf_counter()
            u, v = self.users_tensor.cpu().numpy(), self.items_tensor.cpu().numpy()
            y = self.scores_tensor.cpu().numpy()
            # every minibatch's plan in one call: parameters resolved per minibatch, the arrays built natively on a thread
            # pool (csrc/invpref_plan.cpp; plan.py's numpy builder is the reference implementation, INVPREF_PLAN_NATIVE=0)
            offs = np.array([b[0] for b in self._raw_batches] + [self._raw_batches[-1][0] + self._raw_batches[-1][1]], np.int64)
            contiguous = all(offs[k] + self._raw_batches[k][1] == offs[k + 1] for k in range(len(self._raw_batches)))
            kw = dict(factor_num=self.model.factor_num, user_range=self.shard.user_range(self.model.user_num),
                      env_num=getattr(self.model, 'env_num', 0))
            if contiguous:
                pls = planlib.build_row_plans(u, v, y, offs, self.model.user_num, self.model.item_num, **kw)
            else:
                pls = [planlib.build_row_plan(u[lo:lo + n], v[lo:lo + n], y[lo:lo + n], self.model.user_num,
                                              self.model.item_num, **kw) for lo, n, *_ in self._raw_batches]
            self._plans = [planlib.upload(pl, self.device) for pl in pls]
            self.plan_build_s = time.perf_counter() - t0     # host-side, once per run (reported by bench.py)
        self._alt_setup()
        if self.use_plan and self.users_tensor.is_cuda:
            # every plan shares ONE scratch (records + partial slabs; nothing carries over between steps): size it for
            # the largest BEFORE any graph capture bakes its address in
            t = self._make_tables(st.p_views)
            self.workspace.get_zeroed(max(L.invpref_rows_workspace_bytes(C.byref(t), C.byref(dp.struct))
                                          for dp in self._plans))

    # ---- the alternating form: ONE launch per optimiser step (include/invpref_hip.h: invpref_mstep_alt_hip; csrc/step_alt.hpp).
    # Inside a run of whole epochs launch i evaluates minibatch i % batch_num from the users' side (even i) or the items'
    # (odd i): the side applies the previous step's pending update to its rows, evaluates, applies its own update and
    # pushes contribution rows for the other side; a flush launch ends the run.  Nothing outside a run ever sees the
    # intermediate state; parameters and moments are updated in place (no buffer swap).
    def _alt_setup(self):
        st = self.state
        prev, self._alt = self._alt, None
        # (eagerly issued epochs run the same launches as captured ones: graph replay == eager launches, bit for bit;
        #  INVPREF_ALT_EAGER=0 keeps eagerly issued epochs on the two-launch form)
        self._alt_eager = os.environ.get('INVPREF_ALT_EAGER', '1') == '1'
        if not (self.use_plan and self._plans and self.users_tensor.is_cuda and self._fused_seq()
                and os.environ.get('INVPREF_ALT', '1') == '1' and ops.alt_supported(st.p_views)):
            return
        n_cap = max(b[1] for b in self._raw_batches)
        host = getattr(self, '_alt_host', None)
        if host is None:
            host = self._alt_host = (self.users_tensor.cpu().numpy(), self.items_tensor.cpu().numpy(),
                                     self.scores_tensor.cpu().numpy().astype(np.float32))
        # The alternating form EVALUATES from the items' side every other step: a row's interactions are spread over at most 32
        # slices and each slice evaluates its share one after the other (~1.3 us each).  With 8 192-row Yahoo minibatches the
        # hottest item has 60-77 interactions -- three per slice; with one minibatch per epoch (SURVEY 8(d)-4's B = N variant:
        # 250 interactions per item on average) the chains are tens of evaluations long and the two-launch form, whose item
        # side only SUMS contribution rows, is twice as fast (measured: 199 vs 101 us per step).  Decided on the first
        # minibatch's heaviest row; INVPREF_ALT_MAX_CHAIN overrides the bound.
        lo0, n0 = self._raw_batches[0][0], self._raw_batches[0][1]
        heavy = max(int(np.bincount(host[0][lo0:lo0 + n0]).max(initial=0)), int(np.bincount(host[1][lo0:lo0 + n0]).max(initial=0)))
        if -(-heavy // 32) > int(os.environ.get('INVPREF_ALT_MAX_CHAIN', '6')):
            return
        self._alt = dict(plans={}, host_plans={}, n_cap=n_cap, partials_cap=n_cap // 4 + n_cap // 8 + 128, ws=None,
                         build_s=0.0)
        if prev is not None and prev['n_cap'] == n_cap:
            # (the plans depend on the ids and the minibatch bounds only -- not on the environment / weight arrays whose
            #  re-homing brought us here: keep them, their slot choice and the workspace)
            self._alt.update({k: prev[k] for k in ('plans', 'ws', 'slots') if k in prev})

    def _alt_keys(self, n: int):
        """(k_prev, k, side) of every launch of a run of n epochs, the flush (k = None) last"""
        bn, total = self.batch_num, n * self.batch_num
        keys = [(None, 0, 0)] + [((i - 1) % bn, i % bn, i & 1) for i in range(1, min(total, 2 * bn + 1))]
        return keys + [((total - 1) % bn, None, total & 1)]     # (the plan sequence is periodic after the first launch)

    def _alt_prepare(self, n: int):
        """builds (natively, one call: csrc/invpref_plan.cpp) and uploads every plan a run of n epochs needs, and the run's
        workspace -- BEFORE a graph capture starts: host-to-device copies and allocations are not capturable"""
        A = self._alt
        need = [key for key in dict.fromkeys(self._alt_keys(n)) if key not in A['plans']]
        if need:
            t0 = time.perf_counter()
            u, v, y = self._alt_host
            rng = lambda k: None if k is None else (self._raw_batches[k][0], self._raw_batches[k][1])   # noqa: E731
            if 'slots' not in A:   # group slots per round of either side's launches, from the first minibatch
                lo, n0 = self._raw_batches[0][0], self._raw_batches[0][1]
                ps = (int(os.environ.get('INVPREF_ALT_PER_SLICE_U', '2')), int(os.environ.get('INVPREF_ALT_PER_SLICE_I', '2')))
                A['slots'] = (planlib.alt_slots_for(u[lo:lo + n0], self.model.user_num, ps[0]),
                              planlib.alt_slots_for(v[lo:lo + n0], self.model.item_num, ps[1]))
            hps = planlib.build_alt_plans(u, v, y, [(rng(k), rng(kp), side) for kp, k, side in need], self.model.user_num,
                                          self.model.item_num, factor_num=self.model.factor_num, slots_u=A['slots'][0],
                                          slots_i=A['slots'][1])
            for key, hp in zip(need, hps):
                if hp['n_tasks'] > A['partials_cap']:
                    raise _capi.InvPrefError('alt plan: more job tasks than the workspace was sized for')
                A['plans'][key] = planlib.upload_alt(hp, self.device)
            A['build_s'] += time.perf_counter() - t0
        if A['ws'] is None:
            A['ws'] = ops.AltWorkspace(self.state.p_views, A['n_cap'], A['partials_cap'], pure=self._pure)

    def _alt_plan(self, k_prev, k, side: int, partials_prev: int):
        """device plan of the launch that evaluates minibatch k (None: a flush) from `side` after a launch that evaluated
        minibatch k_prev (None: first launch of a run) and left `partials_prev` partial slabs"""
        return planlib.alt_with_partials(self._alt['plans'][(k_prev, k, side)], partials_prev)

    def _issue_epochs_alt(self, sched: bool, n: int):
        st, A, bn = self.state, self._alt, self.batch_num
        total = n * bn
        if not torch.cuda.is_current_stream_capturing():
            self._alt_prepare(n)
        k_prev, tasks_prev, alpha = None, 0, self.alpha
        for i in range(total):
            j, k = divmod(i, bn)
            self._loss_slot = j
            dp = self._alt_plan(k_prev, k, i & 1, tasks_prev)
            lo, nloc, bnorm, bu, bi, by, be, bw = self._raw_batches[k]
            alpha = self._alpha_for(k)
            st.step += 1
            sc = (self._sched['state'], self._sched['table'], st.step & 1) if sched else None
            lp = self._epoch_losses[(i - 1) // bn, (i - 1) % bn] if i else None
            wts, flags = self._step_weights(bw)
            ops.mstep_alt(st.p_views, st.m_views, st.v_views, dp, be, wts, bnorm,
                          self._raw_batches[k_prev][2] if k_prev is not None else bnorm, self._coefs(alpha), flags, lp,
                          st.step, self.lr, A['ws'], i & 1, pure=self._pure, sched=sc)
            k_prev, tasks_prev = k, dp.n_tasks
        # the flush: the other side's rows take the last step's update, the last fold (in the LAST step's schedule slot)
        dp = self._alt_plan(k_prev, None, total & 1, tasks_prev)
        sc = (self._sched['state'], self._sched['table'], st.step & 1) if sched else None
        bnorm = self._raw_batches[k_prev][2]
        ops.mstep_alt(st.p_views, st.m_views, st.v_views, dp, None, None, bnorm, bnorm, self._coefs(alpha), self._flags,
                      self._epoch_losses[(total - 1) // bn, (total - 1) % bn], st.step, self.lr, A['ws'], total & 1,
                      pure=self._pure, sched=sc)
        self._loss_slot = 0

    def alt_error(self) -> int:
        """1 if a workgroup of an alternating launch ever gave up waiting for its step's small tables (host sync; never in a
        healthy run).  Sticky on the device until _check_alt_error() has raised for it."""
        A = getattr(self, '_alt', None)
        return 0 if not A or A['ws'] is None else A['ws'].error()

    def _check_alt_error(self) -> None:
        """Called wherever the host reads results of a run back anyway (train_epochs(sync=True), loss_dicts of a deferred
        run, the end of train()): a job workgroup that timed out staged NaN tables, so the run's numbers are void --
        say so loudly instead of handing them back (csrc/step_alt.hpp: alt_stage_finish)."""
        if self.alt_error():
            self._alt['ws'].reset_error()
            raise _capi.InvPrefError(
                'alternating M-step: a job workgroup gave up waiting for the fold blocks of its launch (ALT_POLL_MAX polls); '
                'the parameters of this run are invalid (NaN small tables were staged).  Re-create the manager; '
                'INVPREF_ALT=0 selects the two-launch form, which has no in-launch dependency.')

    def _raw_step(self, k: int, alpha: float, stream=None, mid_event=None, sched=False):
        st = self.state
        if not sched:
            self._sched_synced = False
        lo, n, bn, bu, bi, by, be, bw = self._raw_batches[k]
        coefs = self._coefs(alpha)
        multi = self.world_size > 1 or self._force_sharded_path or self._unfused
        # every step's six loss terms go straight into the epoch's loss buffer (this rank's partial sums in a
        # sharded run: they are all-reduced once per epoch, not per step)
        lp = self._epoch_losses[self._loss_slot, k]
        if self.use_plan and not multi:
            # fused M-step + Adam: one pass, gradient never stored, parameters ping-pong
            st.step += 1
            sc = (self._sched['state'], self._sched['table'], st.step & 1) if sched else None
            wts, flags = self._step_weights(bw)
            ops.mstep_rows_adam(st.p_views, st.p_views_alt, st.m_views, st.v_views, self._plans[k], be, by, wts, bn,
                                coefs, flags, lp, st.step, self.lr, self.workspace, pure=self._pure, sched=sc,
                                mid_event=mid_event)   # (profiling: recorded between the step's two launches)
            st.swap()
            return
        st.step += 1
        # graph capture: per-step scalars (Adam bias corrections, a scheduled alpha) come from the device-side
        # schedule, read by the gradient pass and moved on by the ranged Adam launch that ends the step
        sc = (self._sched['state'], self._sched['table'], st.step & 1) if sched else None
        if self.use_plan:
            wts, flags = self._step_weights(bw)
            ops.mstep_rows_grad(st.p_views, st.g_views, self._plans[k], be, by, wts, bn, coefs, flags, lp,
                                self.workspace, sched=sc)
        else:
            ops.mstep_grad(st.p_views, st.g_views, bu, bi, be, by, bw, bn, coefs, self._flags, lp, self.workspace)
        if multi:
            if self.world_size > 1 or self._collective_ok:
                self._exchange_gradient(k)   # all-reduce (whole or packed), or reduce-scatter (this rank keeps its slice of the sum)
        if mid_event is not None:
            mid_event.record()
        # the planned gradient pass overwrites every row it is responsible for, so the gradient buffer needs no zeroing
        zero = not self.use_plan
        if self.use_plan:
            self._grad_stale = True   # see _step()
        if self._adam_ranges_one or sched:   # every piece in one launch
            ops.adam_ranges_(st.param, st.grad, st.exp_avg, st.exp_avg_sq, [o for o, _ in self._adam_ranges],
                             [ln for _, ln in self._adam_ranges], st.step, self.lr, zero_grad=zero, sched=sc)
        else:
            for o, ln in self._adam_ranges:
                ops.adam_(st.param[o:o + ln], st.grad[o:o + ln], st.exp_avg[o:o + ln], st.exp_avg_sq[o:o + ln], st.step,
                          self.lr, zero_grad=zero)
        if multi and (self.world_size > 1 or self._collective_ok):
            if zero and self.exchange == 'scatter' and self.world_size > 1:
                st.grad.zero_()   # plan-free gradients ADD: the slices this rank's Adam did not clear must start from zero too
            self._exchange_parameters()

    def _alpha_for(self, k: int) -> float:
        if self.update_alpha:  # train.py:214-217
            self.alpha = self._scheduled_alpha(self.epoch_cnt, k)
        return self.alpha

    # ---- device-side Adam schedule for graph replay (include/invpref_hip.h: InvPrefAdamSchedule)
    _SCHED_N = 8192

    def _scheduled_alpha(self, epoch_cnt: int, k: int) -> float:
        """train.py:214-217 for minibatch k of the epoch that follows `epoch_cnt` completed ones"""
        p = float(k + (epoch_cnt + 1) * self.batch_num) / float((epoch_cnt + 1) * self.batch_num)
        return 2. / (1. + np.exp(-10. * p)) - 1.

    def _sched_prepare(self, steps_ahead: int):
        """Device-side schedule for graph replay (InvPrefAdamSchedule): one row per optimiser step with the Adam
        scalars and, under the alpha schedule of train.py:214-217, that step's alpha.  Called at an epoch
        boundary: row j belongs to minibatch j % batch_num of epoch epoch_cnt + j // batch_num."""
        st, L = self.state, _capi.lib()
        if self._sched is None:
            table = torch.zeros(self._SCHED_N, 8, dtype=torch.float32, device=self.device)
            state = torch.zeros(32, dtype=torch.int32, device=self.device)
            self._sched = dict(table=table, state=state, base=-(10 ** 9), host=None,
                               struct=_capi.AdamSchedule(state.data_ptr(), table.data_ptr(), self._SCHED_N))
        sc = self._sched
        first = st.step + 1
        stale_alpha = self.update_alpha and not self._sched_synced   # the step -> (epoch, minibatch) mapping moved
        if first < sc['base'] or first + steps_ahead > sc['base'] + self._SCHED_N or stale_alpha:
            host = np.zeros((self._SCHED_N, 8), np.float32)
            _capi.check(L.invpref_adam_schedule_fill(host.ctypes.data, first, self._SCHED_N, self.lr, 0.9, 0.999, 1e-8),
                        'invpref_adam_schedule_fill')
            if self.update_alpha:
                j = np.arange(self._SCHED_N)
                e1 = (self.epoch_cnt + j // self.batch_num + 1).astype(np.float64) * self.batch_num
                host[:, 6] = (2. / (1. + np.exp(-10. * ((j % self.batch_num) + e1) / e1)) - 1.).astype(np.float32)
            sc['table'].copy_(torch.from_numpy(host))
            sc['base'], sc['host'] = first, host
            self._sched_synced = False
        if not self._sched_synced:  # the device counter follows the host's after eager steps / refills
            st32, o = np.zeros(32, np.int32), 16 * (first & 1)  # the slot of the step about to run
            st32[o:o + 2] = first, sc['base']
            st32[o + 2:o + 10] = sc['host'][first - sc['base']].view(np.int32)
            sc['state'].copy_(torch.from_numpy(st32))
            self._sched_synced = True

    def _issue_epochs(self, stream, sched: bool, n: int):
        self._epoch_losses[:n].zero_()
        st = self.state
        if getattr(self, '_alt', None) is not None and (sched or self._alt_eager):
            self._issue_epochs_alt(sched, n)
            return
        for j in range(n):
            self._loss_slot = j
            for k in range(self.batch_num):
                self._raw_step(k, self._alpha_for(k), stream, sched=sched)
        self._loss_slot = 0

    def train_a_epoch(self) -> dict:
        """train.py:204-233, without the per-batch host syncs (one read-back per epoch)."""
        return self.train_epochs(1)[0]

    def train_epochs(self, n: int, sync: bool = True):
        """n consecutive train_a_epoch() calls with ONE host read-back at the end: the epochs are
        enqueued back to back, so the GPU does not idle between them while the host fetches losses
        (train() uses this between two evaluate/cluster events).  Returns the n loss dicts in order;
        with sync=False nothing is read back and the device tensor [n, 6] (LOSS_KEYS order) is returned
        instead -- loss_dicts() turns it into the dicts later."""
        dev, left = [], int(n)
        while left > 0:
            dev.append(self._enqueue_epochs(left))
            left -= dev[-1].shape[0]
        out = torch.cat(dev) if dev else torch.zeros(0, 6, device=self.device)
        if sync:
            self._check_alt_error()
        return self.loss_dicts(out) if sync else out

    @staticmethod
    def loss_dicts(dev_losses: torch.Tensor) -> list:
        return [dict(zip(LOSS_KEYS, v)) for v in dev_losses.tolist()]

    def _enqueue_epochs(self, want: int) -> torch.Tensor:
        """Issues up to `want` epochs on the current stream and returns their mean losses as a device
        tensor [n_issued, 6].  On one GPU with a fixed alpha, runs of whole epochs (batch_num fused steps
        each) are captured once per (parameter-buffer parity, run length) into a HIP graph and replayed:
        one launch per run instead of 2*batch_num per epoch."""
        self.model.train()
        if getattr(self, '_raw_ptrs', None) is None or self._raw_ptrs[2] != self.envs.data_ptr() \
                or self._raw_ptrs[4] != self._sample_weights.data_ptr():
            if getattr(self, '_raw_ptrs', None) is not None and self._raw_ptrs[2] != self.envs.data_ptr():
                self._by_env = False        # (somebody replaced the environments: weights by position until the next stat_envs())
            self._raw_setup()
            self._graphs.clear()
            self._estep_graphs.clear()
        st = self.state
        # per-interaction weights: by environment from the E class weights, or from the N-length array (brought up to date
        # HERE, outside any capture)
        self._epoch_by_env = self._weights_by_env()
        if not self._epoch_by_env:
            self._materialise_sample_weights()
        # the fused single-GPU step, and the gradient-pass -> [all-reduce] -> ranged-Adam sequence of sharded runs and
        # wide rows (RCCL collectives record into a HIP graph like kernels do), are replayed as whole-epoch graphs
        fused_seq = self._fused_seq()
        graph_ok = self.graphs_enabled()
        g = None
        if graph_ok and self._graph_warm:
            n = min(want, self._graph_epochs)
            steps = n * self.batch_num
            self._sched_prepare(steps)
            fresh = self._graph_key(n) not in self._graphs
            try:
                g = self._graph_for(n)
            except Exception as exc:   # e.g. a collective library that cannot record into a graph: stay eager
                import warnings
                warnings.warn(f'HIP-graph capture of the epoch failed ({exc!r}); continuing with eager launches')
                g = None
            if fresh and self.world_size > 1:
                # graph or eager changes the order and the sizes of the collectives that follow (one loss all-reduce per
                # replayed run against one per epoch): a capture that failed on ANY rank puts EVERY rank on eager launches
                ok = torch.tensor([0 if g is None else 1], dtype=torch.int32, device=self.device)
                all_reduce_min_(ok, self.process_group)
                if int(ok.item()) == 0:
                    g = None
                    self._graphs.pop(self._graph_key(n), None)
            if g is None:
                self.use_graph = False
                self._graphs_agreed = None
                self._loss_slot = 0
        if g is not None:
            g.replay()
            st.step += steps
            if steps % 2 and fused_seq and self._alt is None:   # (the alternating form updates in place)
                st.swap()
            if self.world_size > 1:
                all_reduce_sum_(self._epoch_losses[:n], self.process_group)   # per-rank loss partials -> totals
            if self.update_alpha:   # what the reference's loop leaves in self.alpha after these epochs
                self.alpha = self._scheduled_alpha(self.epoch_cnt + n - 1, self.batch_num - 1)
        else:
            n = 1
            self._issue_epochs(None, False, 1)
            if self.world_size > 1:
                all_reduce_sum_(self._epoch_losses[:1], self.process_group)   # per-rank loss partials -> totals
            self._graph_warm = True
            self._sched_synced = False
        self.epoch_cnt += n
        return self._epoch_losses[:n].mean(dim=1)

    def _fused_seq(self) -> bool:
        """one fused M-step + Adam launch per step (single GPU, rows of at most 128 floats), parameters ping-pong"""
        return self.world_size == 1 and not self._unfused and not self._force_sharded_path

    def graphs_enabled(self) -> bool:
        """Are whole epochs replayed as HIP graphs?  (after the first, eagerly issued, epoch)"""
        if getattr(self, '_raw_ptrs', None) is None:
            self._raw_setup()
        mine = bool(self.use_graph and self.use_plan and self.users_tensor.is_cuda
                    and self.batch_num <= self._SCHED_N // 2
                    and (self._fused_seq() or (self._adam_ranges_ok and self._graph_collectives())))
        if self.world_size == 1:
            return mine
        # The decision must be the SAME on every rank (it fixes the sequence of collectives), but its inputs are not:
        # _adam_ranges_ok depends on this rank's row range (user-sharded runs with factor_num % 4 != 0).  Agreed once
        # per manager state with an all-reduce(MIN); a failed capture resets it (see _enqueue_epochs).
        if getattr(self, '_graphs_agreed', None) is None or self._graphs_agreed[0] != mine:
            ok = torch.tensor([1 if mine else 0], dtype=torch.int32, device=self.device)
            all_reduce_min_(ok, self.process_group)
            self._graphs_agreed = (mine, bool(int(ok.item())))
        return self._graphs_agreed[1]

    def _graph_collectives(self) -> bool:
        """May the step's all-reduce be captured?  Yes on the RCCL backend (or when there is none to capture);
        INVPREF_NO_COLLECTIVE_GRAPH=1 keeps the sharded loop eager."""
        if os.environ.get('INVPREF_NO_COLLECTIVE_GRAPH', '0') == '1':
            return False
        if self.world_size == 1 and not self._collective_ok:
            return True
        import torch.distributed as dist
        return dist.get_backend(self.process_group) == 'nccl'

    def _graph_key(self, n: int):
        return (self.state.p_views[0].data_ptr(), n, self.state.step & 1, self._epoch_by_env)

    def _graph_for(self, n: int):
        """The HIP graph of n epochs starting from the current parameter buffer (captured on first use)."""
        st = self.state
        key = self._graph_key(n)
        g = self._graphs.get(key)
        if g is None:
            if self._alt is not None:
                self._alt_prepare(n)
            step0, views0 = st.step, st.p_views
            g = torch.cuda.CUDAGraph()
            try:
                # (thread-local capture mode: the RCCL watchdog thread of a process group may query events meanwhile)
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._issue_epochs(None, True, n)
            finally:
                # capture records, it does not run: put the host-side bookkeeping back
                st.step = step0
                if st.p_views is not views0:
                    st.swap()
            self._graphs[key] = g
        return g

    def prepare_graphs(self, run_lengths) -> None:
        """Captures the epoch graphs for the given run lengths (epochs per replay) and both parameter
        buffers ahead of time, so that no capture happens inside a timed loop.  Needs one eager epoch
        before it (train_epochs(1)); nothing is executed and no state changes."""
        if not self._graph_warm or getattr(self, '_raw_ptrs', None) is None:
            raise RuntimeError('prepare_graphs(): run one epoch first (train_epochs(1))')
        self._sched_prepare(self.batch_num)
        self._epoch_by_env = self._weights_by_env()
        if not self._epoch_by_env:
            self._materialise_sample_weights()
        for n in sorted({min(max(1, int(x)), self._graph_epochs) for x in run_lengths}):
            step0 = self.state.step
            for _ in range(2):  # a fused step swaps the buffers and flips the schedule slot: both move together
                self._graph_for(n)
                if self._fused_seq() and self._alt is None:   # (the alternating form: in place, only the slot parity moves)
                    self.state.swap()
                self.state.step += 1
            self.state.step = step0
        # the E-step graphs of both parameter buffers too (their capture warms up and synchronises the host)
        if self.users_tensor.is_cuda and self.world_size == 1 and not self._pure:
            for _ in range(2):
                if self._fused_estep_ok():
                    self._estep_state()
                    self._estep_graph(self.cluster_use_random_sort, fused=True, combined=True)
                else:
                    self._estep_graph(self.cluster_use_random_sort)
                if self._fused_seq():
                    self.state.swap()

    def _epochs_to_next_event(self) -> int:
        """How many epochs train() may enqueue before the next evaluate / cluster / end of training."""
        r = 1
        while r < 64:
            e = self.epoch_cnt + r
            if e >= self.epochs or e % self.cluster_interval == 0 \
                    or (e % self.evaluate_interval == 0 and e >= self.test_begin_epoch):
                break
            r += 1
        return r

    # ------------------------------------------------------------------ E-step
    def _perm_index_dtype(self):
        """the narrowest type that holds E! - 1: what travels to the device per interaction"""
        return np.uint8 if self.envs_num <= 5 else (np.int32 if self.envs_num <= 12 else np.int64)

    def _eps_index(self) -> np.ndarray:
        """train.py:192-196: per batch, np.random.randint picks one of the E! permutations of the eps vector per row --
        the same draws from the same numpy stream as the reference.  Only the INDEX is kept (1 / 4 / 8 bytes per
        interaction); neither the E! x E table of train.py:86-92 nor an N x E table of gathered rows is built: the E-step
        unranks the row on the device (invpref_estep_perm_hip)."""
        parts = []
        for k in range(self.batch_num):
            glen = self.shard.global_batch_len(k)
            idx = np.random.randint(0, self._eps_rows_cnt, glen)  # same numpy stream as the reference
            parts.append(self.shard.select_in_batch(k, idx))
        return np.concatenate(parts).astype(self._perm_index_dtype(), copy=False)

    def _eps_index_device(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """the drawn indices on the device, through a pinned staging buffer (a plain asynchronous copy)"""
        idx = self._eps_index()
        cuda = self.device.type == 'cuda'
        stage = getattr(self, '_eps_stage', None)
        if stage is None or stage.numel() != len(idx) or stage.dtype != torch.from_numpy(idx[:0]).dtype:
            stage = self._eps_stage = torch.empty(len(idx), dtype=torch.from_numpy(idx[:0]).dtype, pin_memory=cuda)
            self._eps_stage_done = None
        if self._eps_stage_done is not None:
            self._eps_stage_done.synchronize()    # (the previous E-step's copy has left the staging buffer)
        stage.numpy()[:] = idx
        if out is None:
            out = stage.to(self.device, non_blocking=True)
        else:
            out.copy_(stage, non_blocking=True)
        if cuda:
            self._eps_stage_done = torch.cuda.Event()
            self._eps_stage_done.record()
        return out

    # ---- one GPU: cluster() + stat_envs() are ONE launch (include/invpref_hip.h: invpref_estep_fused_hip) -- the assignment
    # kernel's epilogue folds the counts, cluster()'s diff_num and the class weights; nothing N-long is produced (the M-step
    # looks class_weights[env] up itself) and nothing is cloned behind a replay (results land in a ring the kernel advances)
    def _fused_estep_ok(self) -> bool:
        return bool(self.world_size == 1 and self.envs.is_cuda and not self._pure and self.use_plan
                    and os.environ.get('INVPREF_ESTEP_FUSED', '1') == '1')

    def _estep_state(self):
        if self._es is None:
            self._es = ops.EstepState(self.envs_num, self.device)
            self._pend_cw = torch.zeros(self.envs_num, dtype=torch.float32, device=self.device)
            self._pend_counts = torch.zeros(self.envs_num, dtype=torch.int64, device=self.device)
        return self._es

    def _issue_fused_estep(self, eps_buf, combined: bool):
        """enqueues (or records, under capture) the fused E-step.  combined: class weights and counts go straight into the
        manager's own tensors (cluster() and stat_envs() called together, train.py:329-330); otherwise into the buffers a later
        stat_envs() applies."""
        es = self._estep_state()
        ops.estep_fused(self.state.p_views, self.users_tensor, self.items_tensor, self.scores_tensor, self.implicit, self.envs,
                        es, self.workspace, perm_index=eps_buf,
                        eps_base=self._eps_base.tolist() if eps_buf is not None else None,
                        counts=self._counts_dev if combined else self._pend_counts,
                        class_weights=self.class_weights if combined else self._pend_cw)

    def _run_fused_estep(self, combined: bool) -> int:
        """one fused E-step on the current stream -- a graph replay where graphs are on -- and the ring row it writes"""
        es = self._estep_state()
        with_eps = self.cluster_use_random_sort
        if self.use_graph and not torch.cuda.is_current_stream_capturing():
            g, eps_buf, _ = self._estep_graph(with_eps, fused=True, combined=combined)
            self._fill_eps(eps_buf)
            g.replay()
            self._eps_replayed(eps_buf)
        else:
            self._issue_fused_estep(self._eps_index_device() if with_eps else None, combined)
        return es.next_row()

    def cluster_and_stat_envs(self, sync: bool = True):
        """cluster() followed by stat_envs() (train.py:329-330: the reference never calls one without the other inside
        train()) -> (diff_num, {env: count}).  sync=False: no read-back; both come back as device views of the E-step ring
        (valid until ops.EstepState.ring_cap further E-steps have run)."""
        if not self._fused_estep_ok():
            return self.cluster(sync=sync), self.stat_envs(sync=sync)
        self.model.eval()
        row = self._run_fused_estep(combined=True)
        self._pending_stat = None
        self._sw_lazy, self._by_env = True, True      # sample_weights == class_weights[envs] from here on, not materialised
        E, ring = self.envs_num, self._es.ring
        if sync:
            vals = ring[row].tolist()
            return int(vals[E]), {env: int(c) for env, c in enumerate(vals[:E])}
        return ring[row, E:E + 1], ring[row, :E]

    def cluster(self, sync: bool = True):
        """train.py:235-259.  sync=False: no read-back, returns diff_num as a device int64[1] tensor."""
        self.model.eval()
        if self._fused_estep_ok() and not torch.cuda.is_current_stream_capturing():
            # cluster() WITHOUT the stat_envs() that follows it in train(): the reference's sample_weights keep their values BY
            # POSITION (train.py:67, :278) while the environments move -- bring the array up to date with the old environments
            # first, and read it per interaction until stat_envs() runs
            self._materialise_sample_weights()
            row = self._run_fused_estep(combined=False)
            self._by_env = False
            self._pending_stat = 'fused'
            diff = self._es.ring[row, self.envs_num:self.envs_num + 1]
            return int(diff.item()) if sync else diff
        if self.world_size == 1 and self.use_graph and self.envs.is_cuda and not torch.cuda.is_current_stream_capturing():
            # one HIP-graph replay instead of a handful of eager launches behind the Python operator layer (the
            # host-side cost of those was several times the 40 us the kernels take); the permutation indices of
            # train.py:192-196 are host random numbers: they are copied into the buffer the graph reads
            self._materialise_sample_weights()
            counts, diff, cw, sw = self._cluster_replay(self.cluster_use_random_sort)
            self._pending_stat = (counts, cw, sw)
            self._by_env = False
            return int(diff.item()) if sync else diff.clone()
        perm = self._eps_index_device() if self.cluster_use_random_sort else None
        self._materialise_sample_weights()
        self._by_env = False
        # new assignments overwrite self.envs in place (the kernel reads old_envs[i] before writing i)
        new, counts, diff, cw, sw = ops.estep(self.state.p_views, self.users_tensor, self.items_tensor,
                                              self.scores_tensor, self.implicit, self.envs, self.workspace,
                                              new_envs=self.envs, want_weights=(self.world_size == 1),
                                              perm_index=perm, eps_base=self._eps_base.tolist() if perm is not None else None)
        if self.world_size > 1:
            cd = torch.cat([counts, diff])
            all_reduce_sum_(cd, self.process_group)
            counts, diff = cd[:-1].contiguous(), cd[-1:]
            cw, sw = ops.sample_weights(self.envs, counts, self.n_total, self.envs_num)
        self._pending_stat = (counts, cw, sw)
        return int(diff.item()) if sync else diff.clone()

    def _estep_graph(self, with_eps: bool, fused: bool = False, combined: bool = True):
        """The captured E-step for the CURRENT parameter buffer and interaction arrays, captured on first use: (graph, eps
        staging buffer or None, output tensors in the graph's pool -- None for the fused form, whose outputs are the manager's
        own tensors and the ring).  The capture needs a warm-up E-step and a host sync; prepare_graphs() does it ahead of time
        for both parameter buffers, so that an E-step inside a timed loop only replays."""
        key = (self.state.p_views[0].data_ptr(), self.envs.data_ptr(), self.users_tensor.data_ptr(), with_eps, fused, combined)
        ent = self._estep_graphs.get(key)
        if ent is None:
            # (the buffer the captured E-step reads its permutation indices from, 1 / 4 / 8 bytes per interaction: up to
            #  seven environments PINNED HOST memory the kernel reads in place -- no copy in front of the replay --
            #  otherwise a device buffer the staging buffer is copied into)
            eps_buf = None
            if with_eps:
                dt = torch.from_numpy(np.zeros(0, self._perm_index_dtype())).dtype
                n_loc = self.users_tensor.shape[0]
                # INVPREF_EPS_PINNED: 1 (default) = the kernel reads PINNED HOST memory in place (up to seven environments), 0 = a
                # device buffer filled by a copy in front of the replay, "ahead" = a device buffer filled by a copy on a SIDE stream
                # while the epochs enqueued before the E-step still run (_fill_eps).  Round 6 measured "ahead": the kernel 61.9 ->
                # 54.8 us, the replay 69.9 -> 63.8 us -- and the bench's interval 80 us LONGER (15.15 vs 14.63 us per step): the copy's
                # blit kernel and the cross-stream waits sit among launches that fill the chip exactly
                mode = os.environ.get('INVPREF_EPS_PINNED', '1')
                pinned = self.envs_num <= 7 and mode == '1'
                eps_buf = torch.zeros(n_loc, dtype=dt, pin_memory=True) if pinned \
                    else torch.zeros(n_loc, dtype=dt, device=self.device)

            def run():
                if fused:
                    self._issue_fused_estep(eps_buf, combined)
                    return None
                _, counts, diff, cw, sw = ops.estep(self.state.p_views, self.users_tensor, self.items_tensor,
                                                    self.scores_tensor, self.implicit, self.envs, self.workspace,
                                                    new_envs=self.envs, want_weights=True, perm_index=eps_buf,
                                                    eps_base=self._eps_base.tolist() if with_eps else None)
                return counts, diff, cw, sw
            # sizes the workspace outside the capture; the warm-up is not an E-step: the environments, the class weights, the
            # counts and the ring position are put back
            keep = self.envs.clone()
            keep_cw, keep_c = self.class_weights.clone(), self._counts_dev.clone()
            run()
            self.envs.copy_(keep)
            if fused:
                self.class_weights.copy_(keep_cw)
                self._counts_dev.copy_(keep_c)
                self._es.state[1] -= 1
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = run()
            ent = self._estep_graphs[key] = (g, eps_buf, outs)
        return ent

    def _fill_eps(self, eps_buf):
        """this E-step's permutation indices (host numpy stream, like the reference) into the buffer the graph reads"""
        if eps_buf is not None and not eps_buf.is_cuda:
            done = getattr(self, '_eps_read_done', None)
            if done is not None:
                done.synchronize()                 # (the previous replay has read the buffer)
            eps_buf.numpy()[:] = self._eps_index()
        elif eps_buf is not None and os.environ.get('INVPREF_EPS_PINNED', '1') == 'ahead':
            # the draws are host work that overlaps the epochs the GPU is still running; so does their copy: on a side stream,
            # behind the previous replay's read of the buffer, in front of this replay (one event each way)
            idx = self._eps_index()
            stage = getattr(self, '_eps_stage', None)
            if stage is None or stage.numel() != len(idx) or stage.dtype != eps_buf.dtype:
                stage = self._eps_stage = torch.empty(len(idx), dtype=eps_buf.dtype, pin_memory=True)
                self._eps_stage_done = None
            if self._eps_stage_done is not None:
                self._eps_stage_done.synchronize()      # (the previous copy has left the staging buffer)
            stage.numpy()[:] = idx
            side = getattr(self, '_eps_stream', None)
            if side is None:
                side = self._eps_stream = torch.cuda.Stream(device=self.device)
            done = getattr(self, '_eps_read_done', None)
            if done is not None:
                side.wait_event(done)                   # (the previous replay has read the device buffer)
            with torch.cuda.stream(side):
                eps_buf.copy_(stage, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
            self._eps_stage_done = ev
        elif eps_buf is not None:
            self._eps_index_device(out=eps_buf)

    def _eps_replayed(self, eps_buf):
        if eps_buf is not None and (not eps_buf.is_cuda or os.environ.get('INVPREF_EPS_PINNED', '1') == 'ahead'):
            self._eps_read_done = torch.cuda.Event()
            self._eps_read_done.record()

    def _cluster_replay(self, with_eps: bool):
        """One graph per parameter buffer (the fused M-step ping-pongs between two) and per interaction-array set."""
        g, eps_buf, outs = self._estep_graph(with_eps)
        self._fill_eps(eps_buf)
        g.replay()
        self._eps_replayed(eps_buf)
        return outs

    def cluster_a_batch(self, batch_users_tensor, batch_items_tensor, batch_scores_tensor) -> torch.Tensor:
        """train.py:169-202 for one batch (single-rank semantics)."""
        perm = None
        if self.cluster_use_random_sort:
            idx = np.random.randint(0, self._eps_rows_cnt, batch_users_tensor.shape[0])
            perm = torch.from_numpy(idx.astype(self._perm_index_dtype(), copy=False)).to(self.device)
        new, _, _, _, _ = ops.estep(self.state.p_views, batch_users_tensor.contiguous(),
                                    batch_items_tensor.contiguous(), batch_scores_tensor.float().contiguous(),
                                    self.implicit, None, self.workspace, want_weights=False, perm_index=perm,
                                    eps_base=self._eps_base.tolist() if perm is not None else None)
        return new

    def stat_envs(self, sync: bool = True):
        """train.py:268-280.  sync=False: no read-back, returns the per-env counts as a device tensor."""
        pend = getattr(self, '_pending_stat', None)
        self._pending_stat = None
        lazy_ok = self._fused_estep_ok()
        if isinstance(pend, str):                  # a fused cluster() ran: its epilogue left counts and class weights behind
            counts, cw, sw = self._pend_counts, self._pend_cw, None
        elif pend is not None:
            counts, cw, sw = pend
        elif self.world_size == 1:
            counts, cw, sw = ops.stat_envs(self.envs, self.envs_num, self.workspace, want_sample_weights=not lazy_ok)
        else:
            counts, _, _ = ops.stat_envs(self.envs, self.envs_num, self.workspace, want_sample_weights=False)
            all_reduce_sum_(counts, self.process_group)
            cw, sw = ops.sample_weights(self.envs, counts, self.n_total, self.envs_num)
        # in place: the epoch loop (and a captured HIP graph of it) holds pointers into these buffers
        self.class_weights.copy_(cw)
        if lazy_ok:
            # one GPU: the planned M-step takes class_weights[env] (INVPREF_WEIGHTS_BY_ENV); the N-length array is
            # materialised only if somebody asks for the attribute
            self._counts_dev.copy_(counts)
            self._sw_lazy, self._by_env = True, True
        else:
            if sw is None:
                _, sw = ops.sample_weights(self.envs, counts, self.n_total, self.envs_num)
            self._sample_weights.copy_(sw)
            self._sw_lazy = False
            self._by_env = True      # (consistent; whether launches USE it: _weights_by_env())
        return {env: int(c) for env, c in enumerate(counts.tolist())} if sync else counts.clone()

    def update_each_env_count(self):  # train.py:261-266
        counts, _, _ = ops.stat_envs(self.envs, self.envs_num, self.workspace, want_sample_weights=False)
        if self.world_size > 1:
            all_reduce_sum_(counts, self.process_group)
        self.each_env_count.update({e: c for e, c in enumerate(counts)})

    # ------------------------------------------------------------------ outer loop (train.py:282-342)
    def train(self, silent: bool = False, auto: bool = False):
        return self._train_span(silent, auto, initial=True)

    def _train_span(self, silent: bool, auto: bool, initial: bool):
        """The outer loop from the current epoch up to self.epochs; `initial` runs the epoch-0 evaluation and
        stat_envs of train.py:292-301 first."""
        test_result_list, test_epoch_list = [], []
        cluster_diff_num_list, cluster_epoch_list, envs_cnt_list = [], [], []
        loss_result_list, train_epoch_index_list = [], []

        if initial:
            self.sync_parameters()
            temp_eval_result = self.evaluator.evaluate()
            test_result_list.append(temp_eval_result)
            test_epoch_list.append(self.epoch_cnt)
            self.stat_envs()
            if not silent and not auto:
                print('test at epoch:', self.epoch_cnt)
                print(transfer_loss_dict_to_line_str(temp_eval_result))

        # nothing printed -> nothing is read back inside the loop (losses, diff_num and env counts stay on
        # the device until the end, or until an evaluation needs the host anyway): the GPU never waits
        # for the host between epochs and E-steps
        defer = bool(silent or auto)
        while self.epoch_cnt < self.epochs:
            # the epochs up to the next evaluate/cluster event are enqueued together (one read-back);
            # the records and the printed lines are the same, in the same order, as one epoch at a time
            first = self.epoch_cnt + 1
            run = self.train_epochs(self._epochs_to_next_event(), sync=not defer)
            for i in range(len(run)):
                train_epoch_index_list.append(first + i)
                loss_result_list.append(run[i])
                if not defer:
                    print('train epoch:', first + i)
                    print(transfer_loss_dict_to_line_str(run[i]))

            if (self.epoch_cnt % self.evaluate_interval) == 0 and self.epoch_cnt >= self.test_begin_epoch:
                self.sync_parameters()
                temp_eval_result = self.evaluator.evaluate()
                test_result_list.append(temp_eval_result)
                test_epoch_list.append(self.epoch_cnt)
                if not silent and not auto:
                    print('test at epoch:', self.epoch_cnt)
                    print(transfer_loss_dict_to_line_str(temp_eval_result))

            if (self.epoch_cnt % self.cluster_interval) == 0:
                if (self.begin_cluster_epoch is None or self.begin_cluster_epoch <= self.epoch_cnt) \
                        and (self.stop_cluster_epoch is None or self.stop_cluster_epoch > self.epoch_cnt):
                    diff_num, envs_cnt = self.cluster_and_stat_envs(sync=not defer)   # train.py:329-330, one launch
                else:
                    diff_num, envs_cnt = 0, self.stat_envs(sync=not defer)
                cluster_diff_num_list.append(diff_num)
                if defer and self._es is not None and len(cluster_diff_num_list) % (self._es.ring_cap - 8) == 0:
                    # (deferred results are views of the E-step ring: read them out before it wraps)
                    cluster_diff_num_list = [int(d.item()) if torch.is_tensor(d) else d for d in cluster_diff_num_list]
                    envs_cnt_list = [t.tolist() if torch.is_tensor(t) else t for t in envs_cnt_list]
                    envs_cnt = envs_cnt.tolist()
                cluster_epoch_list.append(self.epoch_cnt)
                envs_cnt_list.append(envs_cnt)
                if not defer:
                    print('cluster at epoch:', self.epoch_cnt)
                    print('diff num:', diff_num)
                    print(transfer_loss_dict_to_line_str(envs_cnt))

        self.sync_parameters()  # (user-sharded runs: every rank ends with the complete model)
        self._check_alt_error()
        if defer:  # one read-back for everything
            loss_result_list = [dict(zip(LOSS_KEYS, v)) for v in torch.stack(loss_result_list).tolist()] \
                if loss_result_list else []
            cluster_diff_num_list = [int(d.item()) if torch.is_tensor(d) else d for d in cluster_diff_num_list]
            envs_cnt_list = [{env: int(c) for env, c in enumerate(t.tolist() if torch.is_tensor(t) else t)}
                             for t in envs_cnt_list]

        return (loss_result_list, train_epoch_index_list), \
               (test_result_list, test_epoch_list), \
               (cluster_diff_num_list, envs_cnt_list, cluster_epoch_list)


def _unrank_permutations(idx: np.ndarray, base: np.ndarray) -> np.ndarray:
    """Row idx[i] of ``list(itertools.permutations(base))`` (lexicographic in positions), without
    building the E! table (the reference builds it eagerly, train.py:86-92, and cannot be
    constructed for E >= 11)."""
    E = len(base)
    idx = np.asarray(idx, dtype=np.int64).copy()
    n = len(idx)
    out = np.empty((n, E), np.float32)
    avail = np.tile(np.arange(E, dtype=np.int64), (n, 1))
    for pos in range(E):
        f = math.factorial(E - 1 - pos)
        d = idx // f
        idx = idx % f
        pick = avail[np.arange(n), d]
        out[:, pos] = base[pick]
        # remove the picked element from each row's availability list
        keep = np.ones_like(avail, dtype=bool)
        keep[np.arange(n), d] = False
        avail = avail[keep].reshape(n, -1)
    return out


class ImplicitTrainManager(_InvPrefTrainManager):
    """reference train.py:16-342 (BCELoss)."""
    implicit = True


class ExplicitTrainManager(_InvPrefTrainManager):
    """reference train.py:693-1019 (MSELoss)."""
    implicit = False


class ImplicitTrainStaticPopularityManager(ImplicitTrainManager):
    """reference train.py:484-690: ImplicitTrainManager + per-environment popularity statistics every
    `static_pop_interval` epochs.  The statistics are one device pass over the resident interactions
    (`invpref_static_pop_hip`) instead of numpy boolean masks per environment.  Single GPU."""

    def __init__(self, model, evaluator, device, data_loader, training_data, batch_size, epochs, cluster_interval,
                 evaluate_interval, lr, invariant_coe, env_aware_coe, env_coe, L2_coe, L1_coe, static_pop_interval,
                 alpha=None, use_class_re_weight=False, test_begin_epoch=0, begin_cluster_epoch=None,
                 stop_cluster_epoch=None, cluster_use_random_sort=True, use_recommend_re_weight=True):
        super().__init__(model=model, evaluator=evaluator, device=device, training_data=training_data,
                         batch_size=batch_size, epochs=epochs, cluster_interval=cluster_interval,
                         evaluate_interval=evaluate_interval, lr=lr, invariant_coe=invariant_coe,
                         env_aware_coe=env_aware_coe, env_coe=env_coe, L2_coe=L2_coe, L1_coe=L1_coe, alpha=alpha,
                         use_class_re_weight=use_class_re_weight, test_begin_epoch=test_begin_epoch,
                         begin_cluster_epoch=begin_cluster_epoch, stop_cluster_epoch=stop_cluster_epoch,
                         cluster_use_random_sort=cluster_use_random_sort,
                         use_recommend_re_weight=use_recommend_re_weight, rank=0, world_size=1)
        self.training_np = training_data.cpu().numpy()
        self.static_pop_interval = static_pop_interval
        self.data_loader = data_loader
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)  # noqa: E731
        self._pop_tabs = (dev(data_loader.user_inter_cnt_np, np.int64), dev(data_loader.item_inter_cnt_np, np.int64),
                          dev(data_loader.user_inter_cnt_normalize_np, np.float64),
                          dev(data_loader.item_inter_cnt_normalize_np, np.float64))

    def static_pop(self) -> dict:
        """train.py:509-571: {statistic: {env: mean}}."""
        out = ops.static_pop(self.users_tensor, self.items_tensor, self.envs, self.envs_num, *self._pop_tabs,
                             self.workspace).tolist()
        return {key: {env: out[env][j] for env in range(self.envs_num)} for j, key in enumerate(ops.POP_KEYS)}

    def final_cluster_stat(self, colors_list: list):
        """train.py:573-601: per-interaction popularity values grouped by environment, for the scatter plot."""
        assert len(colors_list) == self.envs_num
        envs = self.envs.cpu().numpy()
        order = np.argsort(envs, kind='stable')           # env 0 rows first, original order inside an env
        u, i = self.training_np[order, 0], self.training_np[order, 1]
        dl = self.data_loader
        colors = [colors_list[e] for e in envs[order].tolist()]
        return (dl.query_users_inter_cnt(u).tolist(), dl.query_items_inter_cnt(i).tolist(),
                dl.query_users_inter_cnt_normalize(u).tolist(), dl.query_items_inter_cnt_normalize(i).tolist(), colors)

    def train(self, silent: bool = False, auto: bool = False):
        """train.py:603-690: the InvPref loop + `static_pop` every static_pop_interval epochs; a fourth result tuple."""
        import json
        stat_dicts, stat_epochs = [], []
        epochs_total = self.epochs
        loss_l, loss_e, test_l, test_e, diff_l, cnt_l, cl_e = [], [], [], [], [], [], []
        first = True
        while first or self.epoch_cnt < epochs_total:
            # run the base loop up to the next statistics epoch (it evaluates at its start only the first time)
            nxt = min(epochs_total, (self.epoch_cnt // self.static_pop_interval + 1) * self.static_pop_interval)
            self.epochs = nxt
            (l, le), (t, te), (d, c, ce) = self._train_span(silent, auto, initial=first)
            first = False
            loss_l += l; loss_e += le; test_l += t; test_e += te; diff_l += d; cnt_l += c; cl_e += ce
            if self.epoch_cnt % self.static_pop_interval == 0 and self.epoch_cnt > 0:
                pop = self.static_pop()
                stat_epochs.append(self.epoch_cnt)
                stat_dicts.append(pop)
                if not silent and not auto:
                    print('pop stat at epoch:', self.epoch_cnt)
                    print(json.dumps(pop, indent=4))
            if self.epoch_cnt >= epochs_total:
                break
        self.epochs = epochs_total
        merged = {key: {env: [d[key][env] for d in stat_dicts] for env in stat_dicts[0][key]} for key in stat_dicts[0]} \
            if stat_dicts else {}
        return (loss_l, loss_e), (test_l, test_e), (diff_l, cnt_l, cl_e), (merged, stat_epochs)
