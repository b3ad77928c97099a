"""ctypes/numpy front-end of the CPU oracle (oracle/invpref_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'liboracle_invpref.so')

F_IMPLICIT, F_REWEIGHT_REC, F_REWEIGHT_CLS, F_REG_ONLY_EMBED, F_REG_ENV_EMBED = 1, 2, 4, 8, 16

PARAM_NAMES = [
    'embed_user_invariant.weight', 'embed_item_invariant.weight',
    'embed_user_env_aware.weight', 'embed_item_env_aware.weight',
    'embed_env.weight', 'env_classifier.linear_map.weight', 'env_classifier.linear_map.bias',
]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, 'invpref_oracle.c')
    twins, twins_src = os.path.join(HERE, 'libinvpref_cpu_abi.so'), os.path.join(HERE, 'invpref_cpu_abi.c')
    stale = (not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src) or not os.path.exists(twins)
             or os.path.getmtime(twins) < max(os.path.getmtime(src), os.path.getmtime(twins_src)))
    if force or stale:   # (the oracle + the CPU twins of the HIP C ABI, invpref_cpu_abi.c: both test infrastructure)
        subprocess.check_call(['make', '-C', HERE, '-s', '-B'])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_cexp.restype = C.c_float
        _lib.oracle_clog.restype = C.c_float
        _lib.oracle_clog1p.restype = C.c_float
        _lib.oracle_cdot2.restype = C.c_float
        _lib.oracle_cdot3.restype = C.c_float
        for f in ('oracle_cexp', 'oracle_clog', 'oracle_clog1p'):
            getattr(_lib, f).argtypes = [C.c_float]
    return _lib


class _Tables(C.Structure):
    _fields_ = [('U', C.c_int64), ('I', C.c_int64), ('E', C.c_int64), ('D', C.c_int64)] + \
               [(n, C.c_void_p) for n in ('Pu', 'Qi', 'Pa', 'Qa', 'Ev', 'W', 'b')]


class _Grads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('Pu', 'Qi', 'Pa', 'Qa', 'Ev', 'W', 'b')]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dt(prec):
    return np.float32 if prec == 'f32' else np.float64


def flags_of(implicit, reweight_rec, reweight_cls, reg_only_embed, reg_env_embed) -> int:
    return (F_IMPLICIT * bool(implicit) | F_REWEIGHT_REC * bool(reweight_rec) | F_REWEIGHT_CLS * bool(reweight_cls)
            | F_REG_ONLY_EMBED * bool(reg_only_embed) | F_REG_ENV_EMBED * bool(reg_env_embed))


class Tables:
    """Holds contiguous copies of the 7 parameter arrays in reference state_dict order."""

    def __init__(self, params: dict, prec: str = 'f32'):
        dt = _dt(prec)
        self.prec = prec
        self.arrs = [np.ascontiguousarray(np.asarray(params[k]), dtype=dt).copy() for k in PARAM_NAMES]
        self.U, self.D = self.arrs[0].shape
        self.I = self.arrs[1].shape[0]
        self.E = self.arrs[4].shape[0]

    def cstruct(self):
        return _Tables(self.U, self.I, self.E, self.D, *[_ptr(a) for a in self.arrs])

    def as_dict(self):
        return dict(zip(PARAM_NAMES, self.arrs))


def _ids(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def forward(tab: Tables, u, v, e, implicit: bool):
    dt = _dt(tab.prec)
    u, v, e = _ids(u), _ids(v), _ids(e)
    B = len(u)
    inv, env, out = np.empty(B, dt), np.empty(B, dt), np.empty((B, tab.E), dt)
    ts = tab.cstruct()
    getattr(lib(), 'oracle_forward_' + tab.prec)(
        C.byref(ts), _ptr(u), _ptr(v), _ptr(e), C.c_int64(B), C.c_uint32(F_IMPLICIT if implicit else 0),
        _ptr(inv), _ptr(env), _ptr(out))
    return inv, env, out


def mstep(tab: Tables, u, v, e, y, w, coefs, flags: int, bnorm: int | None = None, include_dense_reg=True,
          grads=None, losses=None):
    """Returns (grads list in PARAM_NAMES order, losses[6] float64).  Pass grads/losses to accumulate."""
    dt = _dt(tab.prec)
    u, v, e = _ids(u), _ids(v), _ids(e)
    y = np.ascontiguousarray(y, dtype=dt)
    w = None if w is None else np.ascontiguousarray(w, dtype=dt)
    B = len(u)
    if grads is None:
        grads = [np.zeros_like(a) for a in tab.arrs]
    if losses is None:
        losses = np.zeros(6, np.float64)
    ts = tab.cstruct()
    gs = _Grads(*[_ptr(a) for a in grads])
    cf = np.ascontiguousarray(coefs[:6], dtype=np.float64)
    getattr(lib(), 'oracle_mstep_' + tab.prec)(
        C.byref(ts), C.byref(gs), _ptr(u), _ptr(v), _ptr(e), _ptr(y), _ptr(w), C.c_int64(B),
        C.c_int64(B if bnorm is None else bnorm), _ptr(cf), C.c_uint32(flags), C.c_int(bool(include_dense_reg)),
        _ptr(losses))
    return grads, losses


def adam(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, prec='f32'):
    """In-place on contiguous 1-D views p, m, v."""
    n = p.size
    getattr(lib(), 'oracle_adam_' + prec)(
        _ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_int64(n), C.c_int64(step), C.c_double(lr), C.c_double(beta1),
        C.c_double(beta2), C.c_double(eps))


def estep(tab: Tables, u, v, y, implicit: bool, old_envs=None, eps_rows=None, want_dist=False):
    dt = _dt(tab.prec)
    u, v = _ids(u), _ids(v)
    y = np.ascontiguousarray(y, dtype=dt)
    N = len(u)
    new = np.empty(N, np.int64)
    counts = np.zeros(tab.E, np.int64)
    diff = np.zeros(1, np.int64)
    dist = np.empty((N, tab.E), dt) if want_dist else None
    old = None if old_envs is None else _ids(old_envs)
    eps = None if eps_rows is None else np.ascontiguousarray(eps_rows, dtype=dt)
    ts = tab.cstruct()
    getattr(lib(), 'oracle_estep_' + tab.prec)(
        C.byref(ts), _ptr(u), _ptr(v), _ptr(y), C.c_int64(N), C.c_uint32(F_IMPLICIT if implicit else 0),
        _ptr(eps), _ptr(old), _ptr(new), _ptr(counts), _ptr(diff), _ptr(dist))
    return new, counts, int(diff[0]), dist


def stat_envs(envs, E: int, prec='f32'):
    envs = _ids(envs)
    N = len(envs)
    counts = np.zeros(E, np.int64)
    cw = np.zeros(E, _dt(prec))
    sw = np.zeros(N, _dt(prec))
    getattr(lib(), 'oracle_stat_envs_' + prec)(_ptr(envs), C.c_int64(N), C.c_int64(E), _ptr(counts), _ptr(cw),
                                               _ptr(sw))
    return counts, cw, sw


class Trainer:
    """Whole-loop oracle mirroring ImplicitTrainManager/ExplicitTrainManager (train.py:16-342)
    on top of the C functions: used for trajectory parity and as the timed CPU baseline."""

    def __init__(self, params: dict, data: np.ndarray, envs0: np.ndarray, *, implicit: bool, batch_size: int,
                 coefs, lr: float, reweight_rec: bool, reweight_cls: bool, reg_only_embed: bool,
                 reg_env_embed: bool, prec='f32'):
        self.tab = Tables(params, prec)
        self.prec = prec
        dt = _dt(prec)
        self.u, self.v = _ids(data[:, 0]), _ids(data[:, 1])
        self.y = np.ascontiguousarray(data[:, 2], dtype=dt)
        self.envs = _ids(envs0).copy()
        self.N = len(self.u)
        self.implicit = implicit
        self.bs = batch_size
        self.coefs = np.asarray(coefs, np.float64)
        self.lr = lr
        self.flags = flags_of(implicit, reweight_rec, reweight_cls, reg_only_embed, reg_env_embed)
        self.m = [np.zeros_like(a) for a in self.tab.arrs]
        self.vv = [np.zeros_like(a) for a in self.tab.arrs]
        self.step = 0
        self.epoch_cnt = 0
        # alpha = None in the reference (train.py:36-40): alpha follows the training progress (train.py:214-217)
        self.update_alpha = not np.isfinite(self.coefs[5])
        if self.update_alpha:
            self.coefs[5] = 0.0
        self.sample_w = np.zeros(self.N, dt)
        self.counts = None

    def stat_envs(self):
        self.counts, self.class_w, self.sample_w = stat_envs(self.envs, self.tab.E, self.prec)
        return {i: int(c) for i, c in enumerate(self.counts)}

    def train_a_batch(self, lo, hi):
        sl = slice(lo, hi)
        grads, losses = mstep(self.tab, self.u[sl], self.v[sl], self.envs[sl], self.y[sl], self.sample_w[sl],
                              self.coefs, self.flags)
        self.step += 1
        for p, g, m, v in zip(self.tab.arrs, grads, self.m, self.vv):
            adam(p.reshape(-1), g.reshape(-1), m.reshape(-1), v.reshape(-1), self.step, self.lr, prec=self.prec)
        return losses

    def train_a_epoch(self):
        nb = (self.N + self.bs - 1) // self.bs
        ls = []
        for k, lo in enumerate(range(0, self.N, self.bs)):
            if self.update_alpha:  # train.py:214-217
                p = float(k + (self.epoch_cnt + 1) * nb) / float((self.epoch_cnt + 1) * nb)
                self.coefs[5] = 2. / (1. + np.exp(-10. * p)) - 1.
            ls.append(self.train_a_batch(lo, min(lo + self.bs, self.N)))
        self.epoch_cnt += 1
        return np.mean(np.stack(ls), axis=0)

    def cluster(self):
        new, counts, diff, _ = estep(self.tab, self.u, self.v, self.y, self.implicit, old_envs=self.envs)
        self.envs = new
        return diff


# ---------------------------------------------------------------------------------------------
# PureMF baselines (baseline_models.py:12-69 implicit, :652-704 explicit; Basic*TrainManager
# train.py:345-461, :1022-1138) as the degenerate case of the InvPref step:
# with the env-aware tables, embed_env and the classifier all zero, E = 1 and coefficients
# (1, 0, 0, 2*L2_coe, 2*L1_coe, alpha=0), the InvPref loss
#     1*L_inv + (2 L2_coe) * (|Pu[u]|^2 + |Qi[v]|^2) / (2BD) + (2 L1_coe) * (|Pu[u]|_1 + |Qi[v]|_1) / (2BD)
# IS the PureMF loss  BCE|MSE + L2_coe*(|Pu[u]|^2/(BD) + |Qi[v]|^2/(BD)) + L1_coe*(...), the zero tables
# receive exactly-zero gradients (sign(0) = 0, g_q = 0, softmax over one class - onehot = 0) and stay zero
# under Adam.  The REPORTED regularisers are 2x InvPref's.
def pure_mf_params(user_emb, item_emb):
    U, D = user_emb.shape
    I = item_emb.shape[0]
    z = np.zeros
    return {PARAM_NAMES[0]: user_emb, PARAM_NAMES[1]: item_emb, PARAM_NAMES[2]: z((U, D), np.float32),
            PARAM_NAMES[3]: z((I, D), np.float32), PARAM_NAMES[4]: z((1, D), np.float32),
            PARAM_NAMES[5]: z((1, D), np.float32), PARAM_NAMES[6]: z((1,), np.float32)}


def pure_mf_coefs(L2_coe: float, L1_coe: float):
    return np.array([1.0, 0.0, 0.0, 2.0 * L2_coe, 2.0 * L1_coe, 0.0])


def pure_mf_losses(losses6):
    """InvPref's six loss outputs -> PureMF's [score_loss, L2_reg, L1_reg, loss] (train.py:399-404)."""
    l = np.asarray(losses6, np.float64)
    return np.stack([l[..., 0], 2.0 * l[..., 3], 2.0 * l[..., 4], l[..., 5]], axis=-1)


def pure_mf_trainer(user_emb, item_emb, data, *, implicit: bool, batch_size: int, lr: float, L2_coe: float,
                    L1_coe: float, prec='f32') -> Trainer:
    return Trainer(pure_mf_params(user_emb, item_emb), data, np.zeros(len(data), np.int64), implicit=implicit,
                   batch_size=batch_size, coefs=pure_mf_coefs(L2_coe, L1_coe), lr=lr, reweight_rec=False,
                   reweight_cls=False, reg_only_embed=True, reg_env_embed=False, prec=prec)


# ---------------------------------------------------------------------------------------------
# All-core (OpenMP) forms of the timed loops: the `cpu_baseline` leg of bench.py.
# Same arithmetic as the serial functions; the big-table gradients are bit-identical to them.
def omp_max_threads() -> int:
    return int(lib().oracle_omp_max_threads())


def mstep_omp(tab: Tables, u, v, e, y, w, coefs, flags: int, threads: int, bnorm: int | None = None,
              include_dense_reg=True, grads=None, losses=None):
    assert tab.prec == 'f32'
    u, v, e = _ids(u), _ids(v), _ids(e)
    y = np.ascontiguousarray(y, dtype=np.float32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    B = len(u)
    if grads is None:
        grads = [np.zeros_like(a) for a in tab.arrs]
    if losses is None:
        losses = np.zeros(6, np.float64)
    ts = tab.cstruct()
    gs = _Grads(*[_ptr(a) for a in grads])
    cf = np.ascontiguousarray(coefs[:6], dtype=np.float64)
    lib().oracle_mstep_omp_f32(C.byref(ts), C.byref(gs), _ptr(u), _ptr(v), _ptr(e), _ptr(y), _ptr(w), C.c_int64(B),
                               C.c_int64(B if bnorm is None else bnorm), _ptr(cf), C.c_uint32(flags),
                               C.c_int(bool(include_dense_reg)), _ptr(losses), C.c_int(int(threads)))
    return grads, losses


def adam_omp(p, g, m, v, step: int, lr: float, threads: int, zero_grad=True, beta1=0.9, beta2=0.999, eps=1e-8):
    lib().oracle_adam_omp_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), C.c_int64(p.size), C.c_int64(step), C.c_double(lr),
                              C.c_double(beta1), C.c_double(beta2), C.c_double(eps), C.c_int(bool(zero_grad)),
                              C.c_int(int(threads)))


def estep_omp(tab: Tables, u, v, y, implicit: bool, threads: int, old_envs=None):
    assert tab.prec == 'f32'
    u, v = _ids(u), _ids(v)
    y = np.ascontiguousarray(y, dtype=np.float32)
    N = len(u)
    new = np.empty(N, np.int64)
    counts = np.zeros(tab.E, np.int64)
    diff = np.zeros(1, np.int64)
    old = None if old_envs is None else _ids(old_envs)
    ts = tab.cstruct()
    lib().oracle_estep_omp_f32(C.byref(ts), _ptr(u), _ptr(v), _ptr(y), C.c_int64(N),
                               C.c_uint32(F_IMPLICIT if implicit else 0), _ptr(old), _ptr(new), _ptr(counts), _ptr(diff),
                               C.c_int(int(threads)))
    return new, counts, int(diff[0])


class ParallelTrainer(Trainer):
    """Trainer on the all-core functions, with persistent gradient buffers zeroed by the Adam pass (as
    optimizer.zero_grad() + step() do): what bench.py times as the CPU baseline."""

    def __init__(self, *args, threads: int, **kw):
        super().__init__(*args, **kw)
        self.threads = int(threads)
        self.grads = [np.zeros_like(a) for a in self.tab.arrs]

    def train_a_batch(self, lo, hi):
        sl = slice(lo, hi)
        _, losses = mstep_omp(self.tab, self.u[sl], self.v[sl], self.envs[sl], self.y[sl], self.sample_w[sl],
                              self.coefs, self.flags, self.threads, grads=self.grads)
        self.step += 1
        for p, g, m, v in zip(self.tab.arrs, self.grads, self.m, self.vv):
            adam_omp(p.reshape(-1), g.reshape(-1), m.reshape(-1), v.reshape(-1), self.step, self.lr, self.threads)
        return losses

    def cluster(self):
        new, counts, diff = estep_omp(self.tab, self.u, self.v, self.y, self.implicit, self.threads, old_envs=self.envs)
        self.envs = new
        return diff
