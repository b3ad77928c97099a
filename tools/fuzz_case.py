#!/usr/bin/env python3
"""tools/fuzz_case.py SEED... -- one case of tests/test_edge_cases_gpu.py::test_random_plan_parameters_and_shapes per child process
(a GPU memory fault aborts the process: the parent reports it and goes on)"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); import tests.test_edge_cases_gpu as t; "
        "t.test_random_plan_parameters_and_shapes(int(sys.argv[1])); print('ok')" % root)
for seed in sys.argv[1:]:
    r = subprocess.run([sys.executable, '-c', code, seed], capture_output=True, text=True, cwd=root,
                       env=dict(os.environ, INVPREF_FUZZ='100000'))
    tail = [l for l in (r.stdout + r.stderr).splitlines() if 'amdgpu.ids' not in l][-2:]
    print('seed', seed, 'rc', r.returncode, ' | '.join(tail)[:300])
