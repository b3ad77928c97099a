#!/usr/bin/env python3
"""bench.py -- training interactions/sec of the InvPref hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): Yahoo!R3-implicit-shaped synthetic data,
U=15 400, I=1 000, 250 154 interactions PER GPU, E=4, D=64, minibatch 8 192 rows PER GPU (weak
scaling: the global minibatch is 8 192*N rows; every rank takes the interactions of the users it owns and
one RCCL all-reduce per step carries the shared -- item-side -- part of the flat gradient buffer, DESIGN.md
§6), reference Yahoo hyper-parameters (Yahoo_InvPref_Implicit.py:17-41).

A "step" is one optimiser step of the M-step (fused gradient kernel + dense Adam) on one minibatch;
every 155 steps (= cluster_interval 5 epochs x 31 minibatches) the E-step (+ stat_envs) over all
interactions runs inside the timed region, as in the reference loop.  Inputs are resident in HBM
before the timed region.  value = interactions processed by all ranks / max-over-ranks time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

U, I, E, D, N_PER_GPU, B_PER_GPU = 15400, 1000, 4, 64, 250154, 8192
SEED = 17373331
YAHOO = dict(lr=0.005, invariant_coe=3.351991776096847, env_aware_coe=9.988658447411407,
             env_coe=9.06447753571379, L2_coe=3.1351402017943117, L1_coe=0.4935216278026648,
             alpha=1.9053711444718746)
ESTEP_EVERY = 155  # cluster_interval (5 epochs) x 31 minibatches
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


class StubEvaluator:
    def evaluate(self):
        return {}


def pmc_traffic_bytes(kernel_name: str):
    """L2<->fabric bytes per launch of `kernel_name` from the committed rocprofv3 PMC summaries (separate
    --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, tools/profile.sh).  Counters are in KiB;
    FETCH_SIZE is doubled (gfx950 counts 128-byte read requests as 64 bytes, MI355X_MICROARCH.md §HBM)."""
    import csv
    import glob
    tot = {}
    for cname, scale in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'pmc_{cname.split("_")[0].lower()}_summary.csv')))
        if not files:
            return None
        for row in csv.DictReader(open(files[-1])):
            if kernel_name in row['kernel'] and row['counter'] == cname:
                tot[cname] = float(row['mean_value']) * scale
    return tot['FETCH_SIZE'] + tot['WRITE_SIZE'] if len(tot) == 2 else None


def rocprof_avg_ms(kernel_name: str):
    """average duration of `kernel_name` in the committed `rocprofv3 --kernel-trace --stats` summary of this command"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'rocprofv3_kernel_stats_bench_*.csv')))
    if not files:
        return None
    for row in csv.DictReader(open(files[-1])):
        if kernel_name in row['Name']:
            return float(row['AverageNs']) * 1e-6
    return None


def cpu_baseline(seconds_budget: float = 12.0):
    """The oracle (C port, one thread) on the same Yahoo-shaped workload, bounded sample."""
    from invpref_kdd_2022_amd import synth
    from oracle import oracle as O
    data = synth.interactions(SEED, U, I, N_PER_GPU, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    env0 = np.random.RandomState(SEED).randint(0, E, N_PER_GPU)
    cf = [YAHOO[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    tr = O.Trainer(tabs, data, env0, implicit=True, batch_size=B_PER_GPU, coefs=cf, lr=YAHOO['lr'],
                   reweight_rec=False, reweight_cls=True, reg_only_embed=True, reg_env_embed=False)
    tr.stat_envs()
    nb = (N_PER_GPU + B_PER_GPU - 1) // B_PER_GPU
    done, t0 = 0, time.perf_counter()
    k = 0
    while True:
        lo = (k % nb) * B_PER_GPU
        hi = min(lo + B_PER_GPU, N_PER_GPU)
        tr.train_a_batch(lo, hi)
        done += hi - lo
        k += 1
        if time.perf_counter() - t0 > seconds_budget and k % nb == 0:
            break
    t_m = time.perf_counter() - t0
    t1 = time.perf_counter()
    tr.cluster()
    t_e = time.perf_counter() - t1
    return {'value': done / t_m, 'unit': 'interactions/s', 'cores': 1, 'kind': 'port',
            'sample': f'{k} M-step minibatches ({done} interactions, {t_m:.1f} s) of the same Yahoo-shaped workload '
                      f'through oracle/invpref_oracle.c, single thread; E-step alone {N_PER_GPU / t_e:.0f} interactions/s',
            'estep_value': N_PER_GPU / t_e}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=620)
    ap.add_argument('--warmup', type=int, default=62)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from invpref_kdd_2022_amd import parallel, synth
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    from invpref_kdd_2022_amd.train import ImplicitTrainManager

    rank, local, world = parallel.init_from_env('nccl')
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    data = synth.interactions(SEED, U, I, N_PER_GPU * world, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=True, reg_env_embed=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tabs.items()})
    np.random.seed(SEED)
    mgr = ImplicitTrainManager(
        model=model, evaluator=StubEvaluator(), device=dev, training_data=torch.from_numpy(data).to(dev),
        batch_size=B_PER_GPU * world, epochs=10 ** 9, cluster_interval=5, evaluate_interval=10 ** 9,
        use_class_re_weight=True, use_recommend_re_weight=False, cluster_use_random_sort=False,
        rank=rank, world_size=world, **YAHOO)
    mgr.stat_envs()
    nb = mgr.batch_num
    # one-time setup outside the timed region: an eager epoch, then capture of the epoch graphs the loop
    # below replays (runs of 1..5 epochs between two E-steps, both parameter buffers)
    mgr.train_epochs(1)
    if mgr._graph_warm and mgr.use_plan and world == 1 and mgr.use_graph:
        mgr.prepare_graphs(range(1, ESTEP_EVERY // nb + 1))
    state = {'pos': 0, 'done': 0}
    pending = []  # device-side results (epoch losses, diff_num, env counts): read back after the timed region

    def run(n_steps):
        """n_steps optimiser steps of the training loop: whole epochs go through train_epochs() (on a single
        GPU one HIP graph launch per run of epochs), a partial epoch through the same per-step calls issued
        eagerly; the E-step + stat_envs run every ESTEP_EVERY steps as in the reference loop.  Nothing is
        read back to the host inside the loop: the results stay on the device and are fetched (and
        checked) after the timed region."""
        left = n_steps
        while left > 0:
            if state['pos'] == 0 and left >= nb:
                # whole epochs up to the next E-step: enqueued back to back, losses read back once
                n_ep = max(1, min(left, ESTEP_EVERY - state['done'] % ESTEP_EVERY) // nb)
                pending.append(mgr.train_epochs(n_ep, sync=False))
                left -= nb * n_ep
                state['done'] += nb * n_ep
            else:
                if getattr(mgr, '_raw_ptrs', None) is None:
                    mgr._raw_setup()
                mgr._raw_step(state['pos'], mgr.alpha, torch.cuda.current_stream().cuda_stream)
                state['pos'] = (state['pos'] + 1) % nb
                left -= 1
                state['done'] += 1
            if state['done'] % ESTEP_EVERY == 0 and state['pos'] == 0:
                pending.append(mgr.cluster(sync=False))
                pending.append(mgr.stat_envs(sync=False))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    run(args.warmup)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    for p in pending:  # every epoch's losses, every E-step's diff_num and env counts were really produced
        assert bool(torch.isfinite(p.double()).all()), 'non-finite training result'
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # per-launch device time of the step's kernels: HIP events on the launch stream around the same
    # calls issued eagerly (events cannot be read back from inside a replayed graph), one epoch's worth
    if getattr(mgr, '_raw_ptrs', None) is None:
        mgr._raw_setup()
    from invpref_kdd_2022_amd import _capi
    ev = {'m0': [], 'mk': [], 'm1': [], 'a1': []}
    stream = torch.cuda.current_stream().cuda_stream
    fused = mgr.use_plan and world == 1
    # keep the stream backlogged while the instrumented steps are enqueued, so that an event's time stamp is
    # taken right before the next kernel starts and not while the GPU waits for the host
    torch.cuda._sleep(int(4e6))
    for k in range(nb):
        e0, ek, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        if fused:  # the library records ek between mstep_rows_kernel and rows_finish_kernel
            ek.record()  # (creates the underlying hipEvent_t)
            _capi.lib().invpref_set_profile_event(ek.cuda_event)
        e0.record()
        mgr._raw_step(k, mgr.alpha, stream, mid_event=e1)
        e2.record()
        ev['m0'].append(e0); ev['mk'].append(ek); ev['m1'].append(e1); ev['a1'].append(e2)
    _capi.lib().invpref_set_profile_event(None)
    torch.cuda.synchronize()

    # M-step-only and E-step-only rates (report them separately from the blended figure)
    a0, a1, a2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a0.record()
    mgr.train_epochs(3, sync=False)
    a1.record()
    mgr.cluster(sync=False); mgr.stat_envs(sync=False)
    a2.record()
    torch.cuda.synchronize()
    detail = {'mstep_interactions_per_s_per_gpu': 3 * mgr.users_tensor.shape[0] / (a0.elapsed_time(a1) * 1e-3),
              'estep_interactions_per_s_per_gpu': mgr.users_tensor.shape[0] / (a1.elapsed_time(a2) * 1e-3),
              'estep_ms': a1.elapsed_time(a2)}

    inter = args.steps * B_PER_GPU * world
    value = inter / dt
    # per-op device time from HIP events recorded on the launch stream inside the timed region
    ms_m = float(np.median([a.elapsed_time(b) for a, b in zip(ev['m0'], ev['m1'])]))
    ms_a = float(np.median([a.elapsed_time(b) for a, b in zip(ev['m1'], ev['a1'])]))
    P = mgr.state.n
    # Algorithmic bytes.  SURVEY §8(d) prices the un-fused pair: M-step B*(32+32D) (ids/labels,
    # 4 row reads, 4 gradient-row adds) + Adam 32P (28 B/param + 4 B zeroing).  The fused owner pass never
    # stores the gradient, so it is priced at what it must move: B*(32+16D) + 24P (p,m,v read; p',m',v' written).
    bytes_m_survey, bytes_a_survey = B_PER_GPU * (32 + 32 * D), 32 * P
    if fused:
        # dominant kernel alone: mstep_rows_kernel (M-step + Adam of the four big tables); the three small
        # tables (E*D + E*D + E parameters) are finished by rows_finish_kernel and are not counted here
        ms_k = float(np.median([a.elapsed_time(b) for a, b in zip(ev['m0'], ev['mk'])]))
        nbytes = B_PER_GPU * (32 + 16 * D) + 24 * (P - 2 * E * D - 64)
        roof = {'kernel': 'mstep_rows_kernel (M-step with fused Adam)', 'bytes': nbytes, 'ms': ms_k}
        other = {'rows_plus_finish_ms_events': ms_m, 'rows_kernel_ms_events': ms_k,
                 'GBs_of_pair_at_survey_unfused_pricing': (bytes_m_survey + bytes_a_survey) / (ms_m * 1e-3) / 1e9}
    else:
        bytes_m = bytes_m_survey
        if ms_a >= ms_m:
            roof = {'kernel': 'adam_kernel', 'bytes': bytes_a_survey, 'ms': ms_a}
        else:
            roof = {'kernel': 'mstep kernel (+finish)', 'bytes': bytes_m, 'ms': ms_m}
        other = {'mstep_ms': ms_m, 'mstep_GBs': bytes_m / (ms_m * 1e-3) / 1e9, 'adam_ms': ms_a,
                 'adam_GBs': bytes_a_survey / (ms_a * 1e-3) / 1e9}
    achieved = roof['bytes'] / (roof['ms'] * 1e-3) / 1e9
    traffic = pmc_traffic_bytes(roof['kernel'].split()[0])
    roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': 'bytes/launch (rocprofv3 PMC, profiles/)',
                'kernel': roof['kernel'],
                'avg_launch_ms': roof['ms'], 'algorithmic_bytes_per_launch': roof['bytes'],
                'rocprofv3_avg_launch_ms': rocprof_avg_ms(roof['kernel'].split()[0]), 'other': other}
    out = {
        'metric': 'training interactions/sec, Yahoo-implicit InvPref', 'value': value, 'unit': 'interactions/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'yahoo_r3_implicit_shaped', 'users': U, 'items': I, 'envs': E, 'factor_num': D,
                   'interactions_per_gpu': N_PER_GPU, 'batch_per_gpu': B_PER_GPU, 'global_batch': B_PER_GPU * world,
                   'estep_every_steps': ESTEP_EVERY, 'parallelism': (f'{mgr.shard_mode}-sharded x{world}, 1 all-reduce/step ({4 * (mgr.state.n - mgr._ar_lo)} B)' if world > 1 else 'single GPU'),
                   'hip_graph_epochs': bool(mgr._graphs)},
        'roofline': roofline, 'detail': detail,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
