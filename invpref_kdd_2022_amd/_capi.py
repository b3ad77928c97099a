"""ctypes binding of libinvpref_hip.so (include/invpref_hip.h).

torch is plumbing here: tensors provide device memory (``data_ptr()``) and the current HIP
stream; every computation happens inside the library's kernels.  There is no fallback: if the
library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('INVPREF_LIB') or os.path.join(PKG, 'libinvpref_hip.so')  # INVPREF_LIB: kernel A/B builds

IMPLICIT, REWEIGHT_REC, REWEIGHT_CLS, REG_ONLY_EMBED, REG_ENV_EMBED, DENSE_REG, NO_GRAD, PURE_MF = 1, 2, 4, 8, 16, 32, 64, 128
WEIGHTS_BY_ENV = 256   # INVPREF_WEIGHTS_BY_ENV: `sample_weights` holds class_weights[env_num], weight of i = class_weights[envs[i]]
ABI_VERSION = 6

EXPORTS = [
    'invpref_abi_version', 'invpref_device_name', 'invpref_forward_hip', 'invpref_mstep_workspace_bytes',
    'invpref_mstep_grad_hip', 'invpref_adam_hip', 'invpref_estep_workspace_bytes', 'invpref_estep_hip',
    'invpref_stat_envs_hip', 'invpref_sample_weights_hip', 'invpref_backward_hip', 'invpref_predict_hip',
    'invpref_rows_workspace_bytes', 'invpref_mstep_rows_grad_hip', 'invpref_mstep_rows_adam_hip',
    'invpref_adam_schedule_fill', 'invpref_mstep_rows_adam_sched_hip', 'invpref_eval_topk_hip',
    'invpref_eval_error_sums_hip', 'invpref_mstep_rows_adam_profiled_hip', 'invpref_static_pop_workspace_bytes',
    'invpref_static_pop_hip', 'invpref_adam_ranges_hip', 'invpref_mstep_rows_grad_sched_hip',
    'invpref_adam_ranges_sched_hip', 'invpref_rows_lanes_per_group', 'invpref_estep_perm_hip',
    'invpref_pack_rows_hip', 'invpref_unpack_rows_hip', 'invpref_alt_workspace_bytes', 'invpref_alt_supported',
    'invpref_mstep_alt_hip', 'invpref_alt_error_offset', 'invpref_estep_fused_hip', 'invpref_perm_table_fill',
]


class InvPrefError(RuntimeError):
    pass


class Tables(C.Structure):
    """struct InvPrefTables"""
    _fields_ = [('user_num', C.c_int64), ('item_num', C.c_int64), ('env_num', C.c_int64), ('factor_num', C.c_int64),
                ('embed_user_invariant', C.c_void_p), ('embed_item_invariant', C.c_void_p),
                ('embed_user_env_aware', C.c_void_p), ('embed_item_env_aware', C.c_void_p),
                ('embed_env', C.c_void_p), ('classifier_weight', C.c_void_p), ('classifier_bias', C.c_void_p)]


class Coefs(C.Structure):
    """struct InvPrefCoefs"""
    _fields_ = [(n, C.c_float) for n in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]


class AdamSchedule(C.Structure):
    """struct InvPrefAdamSchedule"""
    _fields_ = [('state', C.c_void_p), ('table', C.c_void_p), ('n', C.c_int32), ('slot', C.c_int32)]


_lib = None


def lib():
    """Load the HIP library; fail loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise InvPrefError(
                f'{LIB_PATH} is missing: build it with `python -m invpref_kdd_2022_amd.build` '
                '(or __graft_entry__.build()); the InvPref hot path has no fallback implementation')
        L = C.CDLL(LIB_PATH)
        L.invpref_mstep_workspace_bytes.restype = C.c_size_t
        L.invpref_estep_workspace_bytes.restype = C.c_size_t
        L.invpref_mstep_workspace_bytes.argtypes = [C.POINTER(Tables), C.c_int64]
        L.invpref_estep_workspace_bytes.argtypes = [C.POINTER(Tables), C.c_int64]
        vp, i64, u32, f64 = C.c_void_p, C.c_int64, C.c_uint32, C.c_double
        L.invpref_forward_hip.argtypes = [C.POINTER(Tables), vp, vp, vp, i64, u32, vp, vp, vp, vp]
        L.invpref_mstep_grad_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), vp, vp, vp, vp, vp, i64, i64,
                                             C.POINTER(Coefs), u32, vp, vp, C.c_size_t, vp]
        L.invpref_adam_hip.argtypes = [vp, vp, vp, vp, i64, i64, f64, f64, f64, f64, C.c_int, vp]
        L.invpref_pack_rows_hip.argtypes = [vp, vp, i64, C.c_int32, i64, i64, vp, C.c_int, vp]
        L.invpref_unpack_rows_hip.argtypes = [vp, vp, i64, C.c_int32, i64, i64, vp, C.c_int, vp]
        L.invpref_estep_hip.argtypes = [C.POINTER(Tables), vp, vp, vp, i64, u32, vp, vp, vp, vp, vp, vp, vp, vp,
                                        C.c_size_t, vp]
        L.invpref_estep_perm_hip.argtypes = [C.POINTER(Tables), vp, vp, vp, i64, u32, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                             vp, C.c_size_t, vp]
        L.invpref_stat_envs_hip.argtypes = [vp, i64, i64, vp, vp, vp, vp, C.c_size_t, vp]
        L.invpref_sample_weights_hip.argtypes = [vp, i64, vp, i64, i64, vp, vp, vp]
        L.invpref_backward_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), vp, vp, vp, i64, u32, C.c_float, vp,
                                           vp, vp, vp, C.c_size_t, vp]
        L.invpref_predict_hip.argtypes = [vp, vp, vp, i64, i64, i64, C.c_int, vp, vp]
        L.invpref_rows_workspace_bytes.restype = C.c_size_t
        L.invpref_rows_workspace_bytes.argtypes = [C.POINTER(Tables), vp]
        L.invpref_mstep_rows_grad_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), vp, vp, vp, vp, i64,
                                                   C.POINTER(Coefs), u32, vp, vp, C.c_size_t, vp]
        L.invpref_mstep_rows_adam_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), C.POINTER(Tables),
                                                   C.POINTER(Tables), vp, vp, vp, vp, i64, C.POINTER(Coefs), u32, vp,
                                                   i64, f64, f64, f64, f64, vp, C.c_size_t, vp]
        L.invpref_adam_ranges_hip.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, i64, f64, f64, f64, f64, C.c_int, vp]
        L.invpref_adam_schedule_fill.argtypes = [vp, i64, i64, f64, f64, f64, f64]
        L.invpref_mstep_rows_grad_sched_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), vp, vp, vp, vp, i64,
                                                         C.POINTER(Coefs), u32, vp, C.POINTER(AdamSchedule), vp,
                                                         C.c_size_t, vp]
        L.invpref_adam_ranges_sched_hip.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, C.POINTER(AdamSchedule), C.c_int, vp]
        L.invpref_mstep_rows_adam_sched_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), C.POINTER(Tables),
                                                        C.POINTER(Tables), vp, vp, vp, vp, i64, C.POINTER(Coefs), u32,
                                                        vp, C.POINTER(AdamSchedule), vp, C.c_size_t, vp]
        L.invpref_eval_topk_hip.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp]
        L.invpref_eval_error_sums_hip.argtypes = [vp, vp, i64, vp, vp]
        L.invpref_mstep_rows_adam_profiled_hip.argtypes = L.invpref_mstep_rows_adam_hip.argtypes + [vp]
        L.invpref_rows_lanes_per_group.argtypes = [C.POINTER(Tables)]
        L.invpref_static_pop_workspace_bytes.argtypes = [i64, i64, i64]
        L.invpref_static_pop_workspace_bytes.restype = C.c_size_t
        L.invpref_static_pop_hip.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        L.invpref_device_name.argtypes = [C.c_char_p, C.c_size_t]
        L.invpref_alt_workspace_bytes.restype = C.c_size_t
        L.invpref_alt_workspace_bytes.argtypes = [C.POINTER(Tables), C.c_int32, C.c_int32]
        L.invpref_alt_error_offset.restype = C.c_size_t
        L.invpref_alt_error_offset.argtypes = [C.POINTER(Tables), C.c_int32, C.c_int32]
        L.invpref_alt_supported.argtypes = [C.POINTER(Tables)]
        L.invpref_mstep_alt_hip.argtypes = [C.POINTER(Tables), C.POINTER(Tables), C.POINTER(Tables), vp, vp, vp, i64, i64,
                                            C.POINTER(Coefs), u32, vp, i64, f64, f64, f64, f64, C.POINTER(AdamSchedule), vp,
                                            C.c_size_t, C.c_int32, C.c_int32, C.c_int32, vp]
        L.invpref_estep_fused_hip.argtypes = [C.POINTER(Tables), vp, vp, vp, i64, u32, vp, C.c_int, vp, vp, vp, vp, vp,
                                              C.c_int32, vp, vp, vp, vp, C.c_size_t, vp]
        L.invpref_perm_table_fill.argtypes = [C.c_int32, vp]
        if L.invpref_abi_version() != ABI_VERSION:
            raise InvPrefError('libinvpref_hip.so ABI version mismatch')
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: 'invalid argument', -2: 'unsupported factor_num/env_num', -3: 'workspace too small'}.get(
            rc, f'hipError_t {rc}' if rc > 0 else f'error {rc}')
        raise InvPrefError(f'{what} failed: {kind}')


def ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype, name: str):
    if t is None:
        return
    if not t.is_cuda:
        raise InvPrefError(f'{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback')
    if t.dtype != dtype:
        raise InvPrefError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise InvPrefError(f'{name} must be contiguous')


def make_tables(tensors) -> Tables:
    """tensors: the 7 parameter (or gradient) tensors in state_dict order."""
    pu, qi, pa, qa, ev, w, b = tensors
    for n, t in zip(('Pu', 'Qi', 'Pa', 'Qa', 'Ev', 'W', 'b'), tensors):
        _req(t, torch.float32, n)
    U, D = pu.shape
    I = qi.shape[0]
    E = ev.shape[0]
    if pa.shape != (U, D) or qa.shape != (I, D) or ev.shape != (E, D) or w.shape != (E, D) or b.shape != (E,):
        raise InvPrefError('inconsistent table shapes')
    return Tables(U, I, E, D, *[t.data_ptr() for t in tensors])


def make_pure_tables(tensors) -> Tables:
    """tensors: [user table, item table] of a PureMF model (INVPREF_PURE_MF: the other five tables are absent)."""
    pu, qi = tensors
    for n, t in zip(('user_emb', 'item_emb'), tensors):
        _req(t, torch.float32, n)
    U, D = pu.shape
    if qi.shape[1] != D:
        raise InvPrefError('inconsistent table shapes')
    return Tables(U, qi.shape[0], 1, D, pu.data_ptr(), qi.data_ptr(), None, None, None, None, None)


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(lib().invpref_device_name(buf, 256), 'invpref_device_name')
    return buf.value.decode()
