"""The same C ABI, two back ends (SURVEY section 8(b) level 4): ONE ctypes driver calls `invpref_<entry>_<suffix>` with the
InvPrefTables / InvPrefCoefs structs of include/invpref_hip.h -- suffix `cpu` = oracle/libinvpref_cpu_abi.so over host
memory (test infrastructure built from the oracle; the product has no CPU path), suffix `hip` = the product library over
device memory.  CPU: the twins against the reference's golden vectors.  GPU: the HIP library against the twins, same driver."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

from invpref_kdd_2022_amd import _capi
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G1 = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'g1_*.npz')))
TWINS = os.path.join(ROOT, 'oracle', 'libinvpref_cpu_abi.so')
ENTRIES = ('forward', 'mstep_grad', 'adam', 'estep', 'stat_envs')


def bind(path, suffix):
    """the prototypes of the five core-path entry points -- one table for both libraries"""
    L = C.CDLL(path)
    vp, i64, u32, f64, T, K = C.c_void_p, C.c_int64, C.c_uint32, C.c_double, C.POINTER(_capi.Tables), C.POINTER(_capi.Coefs)
    protos = {'forward': [T, vp, vp, vp, i64, u32, vp, vp, vp, vp],
              'mstep_grad': [T, T, vp, vp, vp, vp, vp, i64, i64, K, u32, vp, vp, C.c_size_t, vp],
              'adam': [vp, vp, vp, vp, i64, i64, f64, f64, f64, f64, C.c_int, vp],
              'estep': [T, vp, vp, vp, i64, u32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp],
              'stat_envs': [vp, i64, i64, vp, vp, vp, vp, C.c_size_t, vp]}
    fns = {}
    for name, args in protos.items():
        f = getattr(L, f'invpref_{name}_{suffix}')
        f.argtypes, f.restype = args, C.c_int
        fns[name] = f
    for name in ('mstep_workspace_bytes', 'estep_workspace_bytes'):   # (the HIP library's size queries carry no suffix)
        f = getattr(L, f'invpref_{name}' + ('' if suffix == 'hip' else '_' + suffix))
        f.argtypes, f.restype = [T, i64], C.c_size_t
        fns[name] = f
    return fns


def run_all(fns, device, z, implicit, flags_bits):
    """forward, M-step gradient, three Adam steps, E-step + stat_envs through `fns` on tensors of `device`"""
    dev = torch.device(device)
    gpu = dev.type == 'cuda'
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if gpu else None

    def T(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)

    def p(t):
        return C.c_void_p(t.data_ptr())
    U, I, E, D, B = [int(x) for x in z['meta'][:5]]
    params = [T(z['p_' + k], np.float32) for k in O.PARAM_NAMES]
    grads = [torch.zeros_like(x) for x in params]

    def tables(ts):
        return _capi.Tables(U, I, E, D, *[x.data_ptr() for x in ts])
    tab, gtab = tables(params), tables(grads)
    u, v, e = T(z['u'], np.int64), T(z['v'], np.int64), T(z['e'], np.int64)
    y, w = T(z['y'], np.float32), T(z['w'], np.float32)
    need = max(fns['mstep_workspace_bytes'](C.byref(tab), B), fns['estep_workspace_bytes'](C.byref(tab), B), 16)
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    out = {}
    inv, env, eo = torch.zeros(B, device=dev), torch.zeros(B, device=dev), torch.zeros(B, E, device=dev)
    flags = (1 if implicit else 0) | flags_bits
    assert fns['forward'](C.byref(tab), p(u), p(v), p(e), B, flags & 1, p(inv), p(env), p(eo), stream) == 0
    losses = torch.zeros(6, device=dev)
    c = z['coefs']
    coefs = _capi.Coefs(*[float(x) for x in c[:6]])
    assert fns['mstep_grad'](C.byref(tab), C.byref(gtab), p(u), p(v), p(e), p(y), p(w), B, B, C.byref(coefs),
                             flags | _capi.DENSE_REG, p(losses), p(ws), ws.numel(), stream) == 0
    if gpu:
        torch.cuda.synchronize()
    out.update(inv=inv.cpu().numpy(), env=env.cpu().numpy(), envout=eo.cpu().numpy(), losses=losses.cpu().numpy(),
               grads=[g.cpu().numpy().copy() for g in grads])
    # three Adam steps on the first table with its gradient held fixed (zero_grad = 0)
    n = params[0].numel() // 4 * 4
    pp, gg = params[0].reshape(-1)[:n].clone(), grads[0].reshape(-1)[:n].clone()
    m, vv = torch.zeros_like(pp), torch.zeros_like(pp)
    for step in (1, 2, 3):
        assert fns['adam'](p(pp), p(gg), p(m), p(vv), n, step, float(c[6]), 0.9, 0.999, 1e-8, 0, stream) == 0
        if step == 1:
            if gpu:
                torch.cuda.synchronize()
            out['adam1_p'] = pp.cpu().numpy().copy()
    new, counts, diff = torch.zeros(B, dtype=torch.int64, device=dev), torch.zeros(E, dtype=torch.int64, device=dev), \
        torch.zeros(1, dtype=torch.int64, device=dev)
    cw, sw = torch.zeros(E, device=dev), torch.zeros(B, device=dev)
    assert fns['estep'](C.byref(tab), p(u), p(v), p(y), B, flags & 1, None, p(e), p(new), p(counts), p(diff), p(cw), p(sw),
                        p(ws), ws.numel(), stream) == 0
    counts2, cw2, sw2 = torch.zeros_like(counts), torch.zeros_like(cw), torch.zeros_like(sw)
    assert fns['stat_envs'](p(new), B, E, p(counts2), p(cw2), p(sw2), p(ws), ws.numel(), stream) == 0
    if gpu:
        torch.cuda.synchronize()
    out.update(adam_p=pp.cpu().numpy(), adam_m=m.cpu().numpy(), new=new.cpu().numpy(), counts=counts.cpu().numpy(),
               diff=int(diff.item()), cw=cw.cpu().numpy(), sw=sw.cpu().numpy(), counts2=counts2.cpu().numpy(),
               cw2=cw2.cpu().numpy(), sw2=sw2.cpu().numpy())
    return out


def load(path):
    z = np.load(path)
    roe, ree, cls_w, rec_w = [int(x) for x in z['meta'][5:9]]
    bits = (_capi.REWEIGHT_REC if rec_w else 0) | (_capi.REWEIGHT_CLS if cls_w else 0) | \
        (_capi.REG_ONLY_EMBED if roe else 0) | (_capi.REG_ENV_EMBED if ree else 0)
    return z, '_implicit_' in path, bits


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_twin_library_exports_the_hip_argument_lists():
    O.build()
    assert os.path.exists(TWINS), 'build it: make -C oracle (or __graft_entry__.build())'
    bind(TWINS, 'cpu')


@pytest.mark.parametrize('path', G1[::3], ids=[os.path.basename(p)[3:-4] for p in G1[::3]])
def test_cpu_twins_reproduce_the_reference_goldens(path):
    z, implicit, bits = load(path)
    o = run_all(bind(TWINS, 'cpu'), 'cpu', z, implicit, bits)
    assert rel(o['inv'], z['inv_f32']) < 2e-6 and rel(o['envout'], z['envout_f32']) < 2e-6
    np.testing.assert_allclose(o['losses'], z['losses_f32'], rtol=1e-5)
    for k, g in zip(O.PARAM_NAMES, o['grads']):
        assert rel(g, z['g_f32_' + k]) < 2e-5, k
    k0 = O.PARAM_NAMES[0]
    n = o['adam_p'].size
    assert rel(o['adam1_p'], z['adam1_f32_' + k0].reshape(-1)[:n]) < 1e-5     # the reference's first optimiser step
    assert (o['new'] != z['newenv_f32']).mean() <= 0.02                        # (rounding-level ties only)
    np.testing.assert_array_equal(o['counts'], np.bincount(o['new'], minlength=len(o['counts'])))
    np.testing.assert_array_equal(o['counts'], o['counts2'])
    np.testing.assert_array_equal(o['cw'], o['cw2'])
    np.testing.assert_array_equal(o['sw'], o['sw2'])
    assert o['diff'] == int((o['new'] != z['e']).sum())


@pytest.mark.gpu
@pytest.mark.parametrize('path', G1, ids=[os.path.basename(p)[3:-4] for p in G1])
def test_hip_library_against_its_cpu_twins(path):
    z, implicit, bits = load(path)
    cpu = run_all(bind(TWINS, 'cpu'), 'cpu', z, implicit, bits)
    hip = run_all(bind(_capi.LIB_PATH, 'hip'), 'cuda:0', z, implicit, bits)
    for k in ('inv', 'env', 'envout', 'new', 'counts', 'cw', 'sw', 'counts2', 'cw2', 'sw2'):   # canonical arithmetic: bit for bit
        np.testing.assert_array_equal(hip[k], cpu[k], err_msg=k)
    assert hip['diff'] == cpu['diff']
    np.testing.assert_allclose(hip['losses'], cpu['losses'], rtol=1e-5)
    for k, a, b in zip(O.PARAM_NAMES, hip['grads'], cpu['grads']):
        assert rel(a, b) < 2e-5, k
    assert rel(hip['adam_p'], cpu['adam_p']) < 1e-5 and rel(hip['adam_m'], cpu['adam_m']) < 1e-5
    assert rel(hip['adam1_p'], cpu['adam1_p']) < 1e-5
