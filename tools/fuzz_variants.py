import os, subprocess, sys
root='/root/repo'
code = """
import sys, ast, os
sys.path.insert(0, %r)
import tests.test_edge_cases_gpu as t
from invpref_kdd_2022_amd import plan as planlib
ov = ast.literal_eval(sys.argv[2])
orig = planlib.build_row_plan
def patched(*a, **kw):
    kw.update(ov)
    return orig(*a, **kw)
planlib.build_row_plan = patched
t.planlib.build_row_plan = patched
t.test_random_plan_parameters_and_shapes(int(sys.argv[1])); print('ok')
""" % root
for ov in ({}, {'stream_split': 0.0}, {'stream_split': 1.0}, {'rounds_per_task': 1}, {'n_classes': 8}, {'n_classes': 1}, {'push': True}, {'per_slice': 2}, {'item_per_slice': 1}, {'native': False}):
    r = subprocess.run([sys.executable, '-c', code, '223', repr(ov)], capture_output=True, text=True, cwd=root, env=dict(os.environ, INVPREF_FUZZ='100000'))
    tail = [l for l in (r.stdout + r.stderr).splitlines() if 'amdgpu.ids' not in l][-1:]
    print(ov, 'rc', r.returncode, ' | '.join(tail)[:160])
