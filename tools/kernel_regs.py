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
                        mfma=code.count('v_mfma'), lines=code.count('\n')))
    return res


if __name__ == '__main__':
    for k in kernels(listing(sys.argv[1] if len(sys.argv) > 1 else None)):
        print('%-62s vgpr %3d (arch %3d) sgpr %3d scratch %4d B (%d ops) mfma %3d lines %5d' % (
            k['name'][:62], k['vgpr'], k['accum'], k['sgpr'], k['scratch'], k['scratch_ops'], k['mfma'], k['lines']))
