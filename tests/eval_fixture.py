"""Synthetic evaluation fixture shared by tests/golden/gen_goldens.py (which feeds it to the
reference's evaluate.py) and the tests (which feed it to this package's evaluators)."""
import numpy as np
import torch


class StubImplicitLoader:
    """the attributes ImplicitTestManager reads (dataloader.py:186-211), filled with synthetic sets"""

    def __init__(self, users, mask, pool, truth):
        self._users, self._mask, self._pool, self._truth = users, mask, pool, truth

    def user_mask_items(self, u):
        return self._mask[u]

    def user_highlight_items(self, u):
        return self._pool[u]

    @property
    def all_test_users_by_sorted_tensor(self):
        return torch.LongTensor(self._users)

    @property
    def all_test_users_by_sorted_list(self):
        return list(self._users)

    @property
    def get_sorted_all_test_users_ground_truth(self):
        return [self._truth[u] for u in self._users]


def eval_fixture(seed=77, U=400, I=1000, n_test=230):
    rs = np.random.RandomState(seed)
    users = sorted(rs.choice(U, n_test, replace=False).tolist())
    mask = {u: set(rs.choice(I, rs.randint(0, 60), replace=False).tolist()) for u in users}
    truth = {u: set(rs.choice(I, rs.randint(1, 12), replace=False).tolist()) for u in users}
    pool = {u: set(rs.choice(I, rs.randint(20, 200), replace=False).tolist()) | truth[u] for u in users}
    return users, mask, pool, truth
