"""Builds libinvpref_hip.so (hand-written gfx950 kernels + the C ABI of include/invpref_hip.h)
in-tree with hipcc.  hipcc cross-compiles without a GPU, so this also runs in CI containers."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libinvpref_hip.so')
OBJDIR = os.path.join(PKG, 'build')
INGEST_LIB = os.path.join(PKG, 'libinvpref_ingest.so')   # host-only data ingest (include/invpref_ingest.h)
INGEST_SRC = os.path.join(CSRC, 'invpref_ingest.cpp')
PLAN_SRC = os.path.join(CSRC, 'invpref_plan.cpp')        # host-only row-plan builder (include/invpref_plan.h), same library
SOURCES = ['invpref_kernels.hip', 'invpref_step.hip', 'invpref_eval.hip']
HEADERS = ['canon_math.hpp', 'kernel_common.hpp', 'step_wide.hpp', os.path.join('..', '..', 'include', 'invpref_hip.h')]
# -ffp-contract=off: every fma of the canonical arithmetic is written explicitly
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-function']


def _hipcc() -> str:
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: the InvPref HIP library cannot be built')


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    # (every header next to the sources counts, listed or not: a stale library is worse than a spare rebuild)
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h'))]
    return any(os.path.getmtime(d) > t for d in deps)


def build_ingest(force: bool = False, verbose: bool = False) -> str:
    deps = [INGEST_SRC, PLAN_SRC, os.path.join(PKG, '..', 'include', 'invpref_ingest.h'),
            os.path.join(PKG, '..', 'include', 'invpref_plan.h')]
    if force or not os.path.exists(INGEST_LIB) or os.path.getmtime(INGEST_LIB) < max(os.path.getmtime(d) for d in deps):
        cxx = shutil.which('g++') or _hipcc()
        cmd = [cxx, '-O2', '-std=c++17', '-shared', '-fPIC', '-pthread', '-Wall', INGEST_SRC, PLAN_SRC, '-o', INGEST_LIB]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return INGEST_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    build_ingest(force, verbose)
    if force or needs_build():
        os.makedirs(OBJDIR, exist_ok=True)
        extra = os.environ.get('INVPREF_HIPCC_EXTRA', '').split()
        procs, objs = [], []
        for src in SOURCES:  # one hipcc per translation unit, in parallel
            obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + '.o')
            cmd = [_hipcc()] + FLAGS + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_variant(name: str, extra_flags, sources=('invpref_step.hip',), force: bool = False) -> str:
    """An A/B or test build of the HIP library under variants/<name>.so (git-ignored; shipped by gpurun): the listed
    translation units recompiled with `extra_flags`, the others taken from the regular build's objects.  Select it with
    INVPREF_LIB=<path> (read by _capi at import).  tests/test_alt_gpu.py builds its forced-time-out library this way."""
    build()
    vdir = os.path.join(PKG, 'variants')
    out = os.path.join(vdir, name + '.so')
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in deps):
        return out
    odir = os.path.join(OBJDIR, 'variant_' + name)
    os.makedirs(odir, exist_ok=True)
    os.makedirs(vdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        if src in sources:
            obj = os.path.join(odir, os.path.splitext(src)[0] + '.o')
            cmd = [_hipcc()] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, src), '-o', obj]
            procs.append((cmd, subprocess.Popen(cmd)))
        else:
            obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + '.o')
            if not os.path.exists(obj):   # (a library that came prebuilt without its objects)
                cmd = [_hipcc()] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
                procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    return out


if __name__ == '__main__':
    print(build(force=True, verbose=True))
