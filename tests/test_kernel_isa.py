"""CPU: what the compiler made of the step kernels (device assembly of csrc/invpref_step.hip, cross-compiled for gfx950 --
no GPU needed).  The planned M-step's instances must stay free of scratch MEMORY traffic in their loops (a dispatch that
needs scratch costs microseconds, a spill inside the interaction loop a multiple of that), the wide-row instances must keep
their outer products on the matrix cores, and rows on 32 lanes their cross-row exchanges on v_permlane16_swap (inline
assembly, csrc/invpref_step.hip: permlane16_swap -- hipcc's builtin mis-compiles)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def kernels():
    import kernel_regs
    if not os.path.exists('/opt/rocm/bin/hipcc'):
        pytest.skip('hipcc not available')
    text = kernel_regs.listing()
    return {k['name']: k for k in kernel_regs.kernels(text)}, text


def test_full_row_instances_have_no_scratch_traffic(kernels):
    ks, _ = kernels
    assert len(ks) >= 25
    # the instances the bench configurations run: Yahoo (16 lanes x 1 float4, E <= 4, full rows), MovieLens (16 x 2, E = 8),
    # and every 16-lane wide instance on full rows
    # (eval instances: <..., BYENV> -- the class-weight form of INVPREF_WEIGHTS_BY_ENV the managers' epochs run, and the
    #  per-interaction-weight form of train_a_batch on caller tensors)
    # (the Yahoo two-launch instance keeps ONE address pair of its once-per-task table staging in scratch memory -- loop depth
    #  1, the rounds; nothing inside the interaction loop)
    k0 = ks['mstep_eval_kernel<16, true, 4, true>']
    assert k0['scratch_ops_in_loops'] == 0 and k0['scratch_ops'] <= 2, k0
    for name in ('mstep_apply_kernel<16, true, 4, true>',
                 'mstep_eval_wide_kernel<16, 2, true, 8, false, true>', 'mstep_eval_wide_kernel<16, 2, true, 8, false, false>',
                 'mstep_apply_wide_kernel<16, 2, true, 8, false>',
                 'mstep_eval_wide_kernel<16, 1, true, 8, false, true>', 'mstep_eval_wide_kernel<16, 1, true, 16, false, true>',
                 'mstep_eval_wide_kernel<16, 1, true, 8, false, false>', 'mstep_eval_wide_kernel<16, 1, true, 16, false, false>',
                 'mstep_apply_wide_kernel<32, 2, true, 16, true>'):
        assert ks[name]['scratch_ops'] == 0, (name, ks[name])
    # ... and the MovieLens instance without a private segment at all (at 100 scalar registers it is one kernel-argument
    # layout away from spilling nine of them: csrc/invpref_step.hip, StepArgs.reserved_)
    assert ks['mstep_eval_wide_kernel<16, 2, true, 8, false, true>']['scratch'] == 0
    assert ks['mstep_eval_wide_kernel<16, 2, true, 8, false, false>']['scratch'] == 0
    # NO instance of the wide kernels touches scratch memory inside its lock-step interaction loop (loop depth >= 2: a reload
    # there waits for every gather in flight); rows on 32 lanes (MIND: 256 accumulator + row registers) and the E = 16 / D = 128
    # instance may keep a handful of once-per-round values there (addresses of the table staging)
    for name, k in ks.items():
        if name.startswith(('mstep_eval_wide_kernel', 'mstep_eval_mm_kernel', 'mstep_apply_wide_kernel')):
            assert k['scratch_ops_in_loops'] == 0, (name, k)
    for name in ('mstep_eval_wide_kernel<32, 2, true, 16, true, true>', 'mstep_eval_wide_kernel<32, 2, true, 8, true, true>',
                 'mstep_eval_wide_kernel<32, 2, true, 16, true, false>', 'mstep_eval_wide_kernel<32, 2, true, 8, true, false>',
                 'mstep_eval_wide_kernel<16, 2, true, 16, false, true>', 'mstep_eval_wide_kernel<16, 2, true, 16, false, false>'):
        assert ks[name]['scratch_ops'] <= 24, (name, ks[name])
    # the MFMA-classifier form of launch 1 (csrc/step_wide_mm.hpp): the default instance (rows on 32 lanes) free of scratch
    # traffic -- a reload from scratch memory waits for every outstanding vector-memory operation of the wave, the gathers in
    # flight included (profiles/r05/EXPERIMENTS.md) -- and its three products on the matrix cores
    for name in ('mstep_eval_mm_kernel<32, 2, 16, true, true>', 'mstep_eval_mm_kernel<32, 2, 8, true, true>',
                 'mstep_eval_mm_kernel<32, 2, 16, true, false>', 'mstep_eval_mm_kernel<32, 2, 8, true, false>'):
        assert ks[name]['scratch_ops'] <= 4 and ks[name]['mfma'] >= 40, (name, ks[name])
    # the latency-tuned Yahoo instance keeps three workgroups per CU: at most 168 registers
    assert ks['mstep_eval_kernel<16, true, 4, true>']['vgpr'] <= 168
    # the alternating one-launch-per-step kernels (csrc/step_alt.hpp): every instance free of scratch, three per CU
    alt = {name: k for name, k in ks.items() if name.startswith('mstep_alt_kernel')}
    assert len(alt) == 18     # {vector + full rows, vector, element-wise} x {first, steady, flush} x {256, 512 threads}
    for name, k in alt.items():
        # (256-thread workgroups run three per CU: 168 registers; the 512-thread ones -- a hot side's rounds of 32 slots -- one
        #  per CU, two waves per SIMD: the whole file)
        cap = 168 if name.endswith(', 256>') else 256
        assert k['scratch_ops'] == 0 and k['scratch'] == 0 and k['mfma'] == 0 and k['vgpr'] <= cap, (name, k)


def test_wide_instances_run_their_outer_products_on_mfma(kernels):
    ks, text = kernels
    for name, k in ks.items():
        if name.startswith('mstep_eval_wide_kernel'):
            assert k['mfma'] >= 16, (name, k['mfma'])       # both unrolled slots: >= 2 x 8 tiles
        if name.startswith(('mstep_eval_kernel', 'mstep_apply_kernel')):
            assert k['mfma'] == 0, name
    assert 'v_mfma_f32_16x16x4_f32' in text
    # rows on 32 lanes reduce across the wave's 16-lane rows with v_permlane16_swap (VALU, no LDS round trip)
    mangled = [m for m in re.findall(r'^(_ZN\S*mstep_eval_wide_kernelILi32\S*):', text, re.M)]
    assert mangled
    for m in mangled:
        i = text.index(m + ':')
        body = text[i:text.index('.Lfunc_end', i)]
        assert 'v_permlane16_swap_b32' in body, m
