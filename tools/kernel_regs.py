#!/usr/bin/env python3
"""tools/kernel_regs.py [file.s] -- registers / scratch / LDS per kernel of a device assembly listing
(hipcc --cuda-device-only -S): every instance of the step kernels must stay free of scratch memory."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing(path=None):
    if path:
        return open(path).read()
    out = '/tmp/invpref_step_dev.s'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-ffp-contract=off',
                           '-fno-fast-math', '-Wno-unused-function', '--cuda-device-only', '-S',
                           os.path.join(ROOT, 'invpref_kdd_2022_amd', 'csrc', 'invpref_step.hip'), '-o', out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _deep_scratch_ops(code: str, min_depth: int = 2) -> int:
    """scratch loads / stores in basic blocks at loop depth >= min_depth (the assembler's block comments: `in Loop: Header=BBx_y
    Depth=d`).  Depth 1 of a task function is its loop over ROUNDS -- a spill there (e.g. an address of the once-per-task table
    staging) runs once per round; depth 2 and deeper is the lock-step interaction loop, where a reload drains the gathers."""
    depth, n = 0, 0
    for line in code.split('\n'):
        m = re.match(r'\s*;\s*%bb\.\d+:|\.LBB\d+_\d+:', line)
        if m:
            d = re.search(r'Depth=(\d+)', line)
            depth = int(d.group(1)) if d else 0
        if re.search(r'scratch_(load|store)', line) and depth >= min_depth:
            n += 1
    return n


def kernels(s):
    res = []
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: int(re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body).group(1))  # noqa: E731
        i = s.index(name + ':')
        j = s.index('.Lfunc_end', i)
        code = s[i:j]
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r'\(anonymous namespace\)::|void |\(.*', '', dem)
        res.append(dict(name=short, vgpr=g('next_free_vgpr'), accum=g('accum_offset'), sgpr=g('next_free_sgpr'),
                        scratch=g('private_segment_fixed_size'), scratch_ops=len(re.findall(r'scratch_(load|store)', code)),
                        scratch_ops_in_loops=_deep_scratch_ops(code),
                        mfma=code.count('v_mfma'), lines=code.count('\n')))
    return res


if __name__ == '__main__':
    for k in kernels(listing(sys.argv[1] if len(sys.argv) > 1 else None)):
        print('%-62s vgpr %3d (arch %3d) sgpr %3d scratch %4d B (%d ops) mfma %3d lines %5d' % (
            k['name'][:62], k['vgpr'], k['accum'], k['sgpr'], k['scratch'], k['scratch_ops'], k['mfma'], k['lines']))
