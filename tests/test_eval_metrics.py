"""CPU: the numpy restatement of the reference metric formulas (evaluate.py:22-56) on hand-checked cases."""
import numpy as np

from invpref_kdd_2022_amd.evaluate import _csr, recall_precision_ndcg


def test_metric_formulas():
    hits = np.array([[1, 0, 1], [0, 0, 0], [1, 1, 1]], np.float32)
    tl = np.array([2, 5, 3], np.float64)
    rec, pre, nd = recall_precision_ndcg(hits, tl, 3)
    assert abs(rec - (2 / 2 + 0 + 3 / 3)) < 1e-12 and abs(pre - (2 / 3 + 0 + 1)) < 1e-12
    d = 1 / np.log2(np.arange(2, 5))
    want = (d[0] + d[2]) / (d[0] + d[1]) + 0 + 1.0
    assert abs(nd - want) < 1e-12
    rec, pre, nd = recall_precision_ndcg(hits, tl, 1)
    assert abs(rec - (1 / 2 + 0 + 1 / 3)) < 1e-12 and abs(pre - 2.0) < 1e-12 and abs(nd - 2.0) < 1e-12


def test_csr():
    p, i = _csr([{3, 1}, set(), {7}])
    assert p.tolist() == [0, 2, 2, 3] and i.tolist() == [1, 3, 7]
