"""Synthetic interaction generators for the InvPref hot path.

The GPU box has no datasets, so every bench / parity input is regenerated from a
seed with numpy's legacy ``RandomState`` (bit-stable across numpy versions and
machines).  Shapes follow SURVEY.md §8(d): the Yahoo!R3-implicit training set of
the reference (``dataset/Yahoo_all_data/train.csv``: 15 400 users x 1 000 items,
250 154 rows, 50 % positives, pre-shuffled, Zipf-like item popularity with up to
~100 repeats of one item inside an 8 192-row minibatch).
"""
from __future__ import annotations

import numpy as np

YAHOO_SHAPE = dict(user_num=15400, item_num=1000, n=250154)
MOVIELENS_SHAPE = dict(user_num=6040, item_num=3706, n=1 << 20)
MIND_SHAPE = dict(user_num=50000, item_num=51283, n=1 << 22)


def zipf_probs(num: int, exponent: float, shift: float) -> np.ndarray:
    ranks = np.arange(num, dtype=np.float64)
    w = (ranks + shift) ** (-exponent)
    return w / w.sum()


def interactions(seed: int, user_num: int, item_num: int, n: int, *, implicit: bool = True,
                 zipf: bool = True) -> np.ndarray:
    """Return an ``[n, 3]`` int64 array of (user, item, score) rows, already shuffled.

    implicit: score in {0,1} (50/50); explicit: score in {1..5}.
    """
    rs = np.random.RandomState(seed)
    if zipf:
        uperm = rs.permutation(user_num)
        users = uperm[rs.choice(user_num, size=n, p=zipf_probs(user_num, 0.6, 50.0))].astype(np.int64)
        iperm = rs.permutation(item_num)
        items = iperm[rs.choice(item_num, size=n, p=zipf_probs(item_num, 0.8, 16.0))].astype(np.int64)
    else:
        users = rs.randint(0, user_num, size=n).astype(np.int64)
        items = rs.randint(0, item_num, size=n).astype(np.int64)
    if implicit:
        scores = (rs.random_sample(n) < 0.5).astype(np.int64)
    else:
        scores = rs.randint(1, 6, size=n).astype(np.int64)
    return np.stack([users, items, scores], axis=1)


def yahoo_like(seed: int = 17373331) -> np.ndarray:
    return interactions(seed, **YAHOO_SHAPE, implicit=True, zipf=True)


def tables(seed: int, user_num: int, item_num: int, env_num: int, factor_num: int,
           std: float = 0.01) -> dict:
    """Parameter tables named like the reference's state_dict (models.py:283-291,200)."""
    rs = np.random.RandomState(seed)

    def normal(*shape):
        return (rs.standard_normal(shape) * std).astype(np.float32)

    bound = float(np.sqrt(6.0 / (factor_num + env_num)))  # xavier_uniform_ (models.py:219-220)
    return {
        'embed_user_invariant.weight': normal(user_num, factor_num),
        'embed_item_invariant.weight': normal(item_num, factor_num),
        'embed_user_env_aware.weight': normal(user_num, factor_num),
        'embed_item_env_aware.weight': normal(item_num, factor_num),
        'embed_env.weight': normal(env_num, factor_num),
        'env_classifier.linear_map.weight':
            rs.uniform(-bound, bound, size=(env_num, factor_num)).astype(np.float32),
        'env_classifier.linear_map.bias':
            rs.uniform(-1.0 / np.sqrt(factor_num), 1.0 / np.sqrt(factor_num), size=(env_num,)).astype(np.float32),
    }
