"""CPU: libinvpref_hip.so loads (no GPU needed for dlopen) and exports every function that
include/invpref_hip.h declares; the ctypes mirrors of the ABI structs have the C layout; argument
validation returns error codes without touching a device; a missing library fails loudly."""
import ctypes as C
import os
import re

import pytest

from invpref_kdd_2022_amd import _capi, build, plan as planlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, 'include', 'invpref_hip.h')).read()


@pytest.fixture(scope='module')
def lib():
    build.build()
    return C.CDLL(_capi.LIB_PATH)


def declared_functions():
    code = re.sub(r'/\*.*?\*/', '', HEADER, flags=re.S)
    return sorted(set(re.findall(r'\b(invpref_\w+)\s*\(', code)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/invpref_hip.h but not exported'
    assert set(_capi.EXPORTS) <= set(names)
    assert lib.invpref_abi_version() == _capi.ABI_VERSION == 6


def test_struct_layouts_match_header():
    # InvPrefTables: 4 x int64 + 7 pointers ; InvPrefCoefs: 6 floats ; InvPrefAdamSchedule: 2 ptr + int32 (padded)
    assert C.sizeof(_capi.Tables) == 4 * 8 + 7 * 8
    assert C.sizeof(_capi.Coefs) == 24
    assert C.sizeof(_capi.AdamSchedule) == 24
    # InvPrefRowPlan: 6 int32, 5 ptr, 2 int32, 1 ptr, 2 int32, int32[8][8], 2 ptr
    assert C.sizeof(planlib.RowPlanStruct) == 24 + 5 * 8 + 8 + 8 + 8 + 64 * 4 + 16
    fields = re.search(r'typedef struct InvPrefRowPlan \{(.*?)\} InvPrefRowPlan;', HEADER, re.S).group(1)
    fields = re.sub(r'/\*.*?\*/', '', fields, flags=re.S)
    fields = re.sub(r'\[\d+\]', '', fields)
    names = re.findall(r'[\*\s,](\w+)\s*(?=[,;])', fields)
    assert names == [f[0] for f in planlib.RowPlanStruct._fields_]
    # InvPrefAltPlan: 9 int32 (+ 4 padding), 4 ptr, 2 int32, 1 ptr, int32, int32[8][4], int32 (+ padding to 8)
    assert C.sizeof(planlib.AltPlanStruct) == 9 * 4 + 4 + 4 * 8 + 8 + 8 + 4 + 32 * 4 + 4
    fields = re.search(r'typedef struct InvPrefAltPlan \{(.*?)\} InvPrefAltPlan;', HEADER, re.S).group(1)
    fields = re.sub(r'/\*.*?\*/', '', fields, flags=re.S)
    fields = re.sub(r'\[\d+\]', '', fields)
    names = re.findall(r'[\*\s,](\w+)\s*(?=[,;])', fields)
    assert names == [f[0] for f in planlib.AltPlanStruct._fields_]


def test_argument_validation_without_a_device(lib):
    lib.invpref_adam_hip.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int64] + [C.c_double] * 4 + [C.c_int, C.c_void_p]
    assert lib.invpref_adam_hip(None, None, None, None, 16, 1, 0.01, 0.9, 0.999, 1e-8, 1, None) == -1   # EINVAL
    t = _capi.Tables(10, 10, 2, 300, 1, 1, 1, 1, 1, 1, 1)   # factor_num 300 > INVPREF_MAX_FACTORS
    lib.invpref_forward_hip.argtypes = [C.POINTER(_capi.Tables)] + [C.c_void_p] * 3 + [C.c_int64, C.c_uint32] + [C.c_void_p] * 4
    assert lib.invpref_forward_hip(C.byref(t), None, None, None, 0, 0, None, None, None, None) == -2    # EUNSUPPORTED
    lib.invpref_adam_schedule_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_double] * 4
    buf = (C.c_float * 16)()   # two rows of 8 floats: six Adam scalars, alpha (NaN = the call's), unused
    assert lib.invpref_adam_schedule_fill(buf, 1, 2, 0.01, 0.9, 0.999, 1e-8) == 0
    assert abs(buf[0] - 0.01 / (1 - 0.9)) < 1e-6 and abs(buf[8] - 0.01 / (1 - 0.81)) < 1e-6
    assert buf[6] != buf[6] and buf[14] != buf[14] and buf[7] == 0.0
    # the alternating form: bad arguments come back as codes before anything touches a device
    L = _capi.lib()
    t = _capi.Tables(10, 10, 4, 64, 1, 1, 1, 1, 1, 1, 1)
    cf = _capi.Coefs(1, 1, 1, 0, 0, 0)
    assert L.invpref_mstep_alt_hip(C.byref(t), C.byref(t), C.byref(t), None, None, None, 8, 8, C.byref(cf), 1, None, 1, 0.01,
                                   0.9, 0.999, 1e-8, None, None, 0, 8, 8, 0, None) == -1
    ap = planlib.AltPlanStruct(side=0, has_prev=0, has_cur=1, n=4, n_prev=0, lanes_per_group=16, slots_per_round=16, n_rounds=0,
                               rounds_per_task=2)
    assert L.invpref_mstep_alt_hip(C.byref(t), C.byref(t), C.byref(t), C.byref(ap), 1, None, 8, 8, C.byref(cf), 1, None, 1, 0.01,
                                   0.9, 0.999, 1e-8, None, 1, 1 << 20, 8, 8, 0, None) == -1     # rounds_per_task must be 1
    wide = _capi.Tables(10, 10, 8, 128, 1, 1, 1, 1, 1, 1, 1)
    assert L.invpref_alt_supported(C.byref(wide)) == 0 and L.invpref_alt_supported(C.byref(t)) == 1
    assert L.invpref_mstep_alt_hip(C.byref(wide), C.byref(wide), C.byref(wide), C.byref(ap), 1, None, 8, 8, C.byref(cf), 1, None,
                                   1, 0.01, 0.9, 0.999, 1e-8, None, 1, 1 << 20, 8, 8, 0, None) == -2   # EUNSUPPORTED
    assert L.invpref_alt_workspace_bytes(C.byref(t), 8, 8) > L.invpref_alt_error_offset(C.byref(t), 8, 8) > 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, '_lib', None)
    monkeypatch.setattr(_capi, 'LIB_PATH', '/nonexistent/libinvpref_hip.so')
    with pytest.raises(_capi.InvPrefError):
        _capi.lib()
