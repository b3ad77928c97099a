"""Drop-in ``ImplicitTestManager`` / ``ExplicitTestManager`` (reference evaluate.py:59-175, :178-212).

Same constructor signatures and result dictionaries.  The reference walks python lists per test
user (mask / highlight index lists, ``x in groundTrue``) and builds a ``[n*I, D]`` tensor in
``model.predict``; here the per-user sets are turned into CSR arrays ONCE, and a test batch is three HIP
launches: ``predict_kernel`` (rating matrix), ``topk_mask_kernel`` (mask, highlight, top-k, hit labels).
The metric formulas (recall / precision / NDCG sums, evaluate.py:22-56) are restated in numpy float64
on the ``[n, k]`` hit matrix.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._capi import check, lib, ptr, stream_ptr


def _csr(sets, n_items=None) -> tuple[np.ndarray, np.ndarray]:
    ptrs = np.zeros(len(sets) + 1, np.int32)
    items = []
    for i, s in enumerate(sets):
        a = np.sort(np.fromiter(s, dtype=np.int64, count=len(s)))
        items.append(a)
        ptrs[i + 1] = ptrs[i] + len(a)
    flat = np.concatenate(items).astype(np.int32) if items else np.zeros(0, np.int32)
    return ptrs, flat


def recall_precision_ndcg(hits: np.ndarray, truth_len: np.ndarray, k: int):
    """Sums over the batch of recall@k, precision@k, NDCG@k (evaluate.py:22-56)."""
    r = hits[:, :k].astype(np.float64)
    right = r.sum(1)
    recall = float(np.sum(right / truth_len))
    precision = float(np.sum(right / k))
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    length = np.minimum(truth_len, k)
    ideal = (np.arange(k)[None, :] < length[:, None]).astype(np.float64)
    idcg = (ideal * disc).sum(1)
    idcg[idcg == 0.] = 1.
    ndcg = (r * disc).sum(1) / idcg
    ndcg[np.isnan(ndcg)] = 0.
    return recall, precision, float(ndcg.sum())


class ImplicitTestManager:
    def __init__(self, model, data_loader, test_batch_size: int, top_k_list: list, use_item_pool: bool = False):
        self.model = model
        self.data_loader = data_loader
        self.batch_size = test_batch_size
        self.top_k_list = top_k_list
        self.top_k_list.sort(reverse=False)
        self.use_item_pool = use_item_pool
        self._dev = None

    def _prepare(self, device):
        dl = self.data_loader
        users = list(dl.all_test_users_by_sorted_list)
        if hasattr(dl, 'csr_for_eval'):  # this package's loaders hand over CSR arrays (dataloader.py): no sets walked
            ev = dl.csr_for_eval()
            (mp, mi), (tp, ti) = ev['mask'], ev['truth']
            truth_len = np.diff(tp)
            if self.use_item_pool:
                hp, hi = ev['highlight']
        else:                            # any loader with the reference's interface (python sets)
            truth = dl.get_sorted_all_test_users_ground_truth
            mp, mi = _csr([dl.user_mask_items(u) for u in users])
            tp, ti = _csr(truth)
            truth_len = [len(t) for t in truth]
            if self.use_item_pool:
                hp, hi = _csr([dl.user_highlight_items(u) for u in users])
        arrs = dict(mask_ptr=mp, mask_items=mi, truth_ptr=tp, truth_items=ti)
        if self.use_item_pool:
            arrs.update(hl_ptr=hp, hl_items=hi)
        self._dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in arrs.items()}
        for k in ('mask_items', 'truth_items', 'hl_items'):  # a zero-length tensor has no valid pointer
            if k in self._dev and self._dev[k].numel() == 0:
                self._dev[k] = torch.zeros(1, dtype=torch.int32, device=device)
        self._users = torch.as_tensor(np.asarray(users, np.int64)).to(device)
        self._truth_len = np.asarray(truth_len, np.float64)

    def topk(self, lo: int, hi: int):
        """(items int32[n,k], hits fp32[n,k]) for test users [lo, hi) of the sorted list."""
        d = self._dev
        users = self._users[lo:hi].contiguous()
        n, k = hi - lo, max(self.top_k_list)
        ratings = self.model.predict(users)
        items = torch.empty(n, k, dtype=torch.int32, device=users.device)
        hits = torch.empty(n, k, dtype=torch.float32, device=users.device)
        off = lambda t, o: C.c_void_p(t.data_ptr() + 4 * o)  # noqa: E731
        check(lib().invpref_eval_topk_hip(
            ptr(ratings), n, ratings.shape[1], off(d['mask_ptr'], lo), ptr(d['mask_items']),
            off(d['hl_ptr'], lo) if self.use_item_pool else None, ptr(d['hl_items']) if self.use_item_pool else None,
            off(d['truth_ptr'], lo), ptr(d['truth_items']), k, ptr(items), ptr(hits), stream_ptr()),
            'invpref_eval_topk_hip')
        return items, hits

    def evaluate(self) -> dict:
        self.model.eval()
        device = next(self.model.parameters()).device
        if self._dev is None:
            self._prepare(device)
        n_users = self._users.shape[0]
        sums = {m: np.zeros(len(self.top_k_list)) for m in ('ndcg', 'recall', 'precision')}
        # test_batch_size bounds the reference's [n * I, D] temporary (models.py:393-407); here a batch is one score matrix of
        # n x I floats and three launches, and the metrics are sums over users -- the same whatever the batch -- so small
        # batches are merged up to a 1 GiB score matrix (MIND's 256-user batches: 196 launches + read-backs -> 10)
        n_items = int(self.model.item_num) if hasattr(self.model, 'item_num') else 1
        step = max(int(self.batch_size), min(n_users, (1 << 28) // max(1, n_items)))
        for lo in range(0, n_users, step):
            hi = min(lo + step, n_users)
            _, hits = self.topk(lo, hi)
            h = hits.cpu().numpy()
            for i, k in enumerate(self.top_k_list):
                rec, pre, nd = recall_precision_ndcg(h, self._truth_len[lo:hi], k)
                sums['recall'][i] += rec
                sums['precision'][i] += pre
                sums['ndcg'][i] += nd
        return {m: {k: float(v[i] / float(n_users)) for i, k in enumerate(self.top_k_list)} for m, v in sums.items()}


class ExplicitTestManager:
    def __init__(self, model, data_loader):
        self.model = model
        self.data_loader = data_loader

    def evaluate(self) -> dict:
        self.model.eval()
        device = next(self.model.parameters()).device
        pairs = self.data_loader.all_test_pairs_tensor.to(device)
        users, items = pairs[:, 0].reshape(-1).contiguous(), pairs[:, 1].reshape(-1).contiguous()
        target = self.data_loader.all_test_scores_tensor.to(device).float().contiguous()
        pred = self.model.predict(users, items)
        out = torch.empty(2, dtype=torch.float64, device=device)
        check(lib().invpref_eval_error_sums_hip(ptr(pred), ptr(target), pred.numel(), ptr(out), stream_ptr()),
              'invpref_eval_error_sums_hip')
        s2, s1 = out.tolist()
        n = float(pred.numel())
        return {'mse': s2 / n, 'rmse': float(np.sqrt(s2 / n)), 'mae': s1 / n}
