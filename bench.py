#!/usr/bin/env python3
"""bench.py -- training interactions/sec of the InvPref hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: spawns its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] at N = 1, configs[3] at N > 1): Yahoo!R3-implicit-shaped
synthetic data, U=15 400, I=1 000, 250 154 interactions, E=4, D=64, minibatch 8 192 rows, reference Yahoo
hyper-parameters (Yahoo_InvPref_Implicit.py:17-41).  --scaling strong (the default): the SAME problem at every N, as
BASELINE.json configs[3] / SURVEY 8(d)-4 word it -- 250 154 interactions in total, every global minibatch of 8 192 rows
(the reference's own, Yahoo_InvPref_Implicit.py:25; contiguous unshuffled slices, utils.py:12-19) cut N ways (1 024 rows
per rank at N = 8), interactions ROW-sharded, one RCCL all-reduce of the flat gradient buffer per optimiser step; the
B = N variant (one step per epoch), the other exchanges, the user-sharded layout and the weak-scaling
figures of rounds 1-5 (250 154 interactions and 8 192 rows PER GPU: --scaling weak) are in `detail`.

A "step" is one optimiser step of the M-step (fused gradient kernel + dense Adam) on one minibatch; every 155 steps
(= cluster_interval 5 epochs x 31 minibatches) the E-step (+ stat_envs) over all interactions runs inside the
timed region, as in the reference loop.  The timed region is always WHOLE cluster intervals (so the replayed HIP
graphs and the E-step are inside it) and at least MIN_TIMED_S long: --steps / --warmup are rounded up accordingly
and the line reports both (`steps_requested`, `steps` = timed).  Inputs are resident in HBM before the timed region;
nothing is read back inside it.  value = interactions processed by all ranks / max-over-ranks time (EXACT: whole epochs of
n interactions each -- the ragged last minibatch counts its 4 394 rows, not 8 192).

Parity gate: before the line is printed, one epoch + one E-step from a seeded state run through the timed
path (alternating launches, fused E-step with the reference's default tie-break) and through the CPU oracle; the line carries
`parity_gate` and NO `value` when the six loss terms differ by more than 1e-5 relative or a single environment assignment
differs on the same tables.  An N-rank run gates ITS path the same way (sharded_parity_gate: the row-sharded epoch and E-step of
all ranks against the oracle on the whole problem, plus bit-equal parameter replicas).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

U, I, E, D, N_PER_GPU, B_PER_GPU = 15400, 1000, 4, 64, 250154, 8192
SEED = 17373331
YAHOO = dict(lr=0.005, invariant_coe=3.351991776096847, env_aware_coe=9.988658447411407,
             env_coe=9.06447753571379, L2_coe=3.1351402017943117, L1_coe=0.4935216278026648,
             alpha=1.9053711444718746)
CLUSTER_INTERVAL = 5          # epochs (Yahoo_InvPref_Implicit.py:27)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
MIN_TIMED_S = 0.5
MAX_TIMED_S = 30.0


class StubEvaluator:
    def evaluate(self):
        return {}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes
    (torch.distributed.run) before this process has touched the GPU, and return their exit code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def pmc_traffic_bytes(kernel_names):
    """L2<->fabric bytes per step -- summed over the step's kernels -- from the committed rocprofv3 PMC summaries
    (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, tools/profile_r03.sh).  Counters are in KiB;
    FETCH_SIZE is doubled (gfx950 counts 128-byte read requests as 64 bytes, MI355X_MICROARCH.md §HBM).  Returns
    (bytes, stamp): the stamp names the profile directory and the commit the profile was taken at."""
    import csv
    import glob
    if isinstance(kernel_names, str):
        kernel_names = [kernel_names]
    tot, src = 0.0, None
    for cname, scale in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', f'pmc_{cname.split("_")[0].lower()}_summary.csv')))
        if not files:
            return None, None
        src = os.path.dirname(files[-1])
        found = set()
        for row in csv.DictReader(open(files[-1])):
            for k in kernel_names:
                if k in row['kernel'] and row['counter'] == cname and k not in found:
                    tot += float(row['mean_value']) * scale
                    found.add(k)
        if len(found) != len(kernel_names):
            return None, None
    stamp = {'profile': os.path.relpath(src, ROOT)}
    try:
        stamp.update(json.load(open(os.path.join(src, 'STAMP.json'))))
    except (OSError, ValueError):
        pass
    return tot, stamp


def rocprof_avg_ms(kernel_names):
    """Sum of the average durations of the step's kernels in the committed `rocprofv3 --kernel-trace --stats` summary
    of this command (profiles/rNN/rocprofv3_kernel_stats_bench_graph.csv: the graph-replayed steps)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'rocprofv3_kernel_stats_bench_graph*.csv'))) or \
        sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', 'rocprofv3_kernel_stats_bench_*.csv')))
    if not files:
        return None
    tot, seen = 0.0, set()
    for row in csv.DictReader(open(files[-1])):
        for k in kernel_names:
            if k in row['Name'] and k not in seen:
                tot += float(row['AverageNs']) * 1e-6
                seen.add(k)
    return tot if len(seen) == len(kernel_names) else None


def alt_profile_figures():
    """The alternating form in the committed profile (profiles/rNN): every steady-state launch of mstep_alt_kernel is ONE
    optimiser step, so the per-step figures are launch-weighted means over its instances (256- and 512-thread workgroups =
    user-side and item-side launches).  Returns (avg launch ms from `rocprofv3 --kernel-trace --stats`, L2<->fabric bytes
    per launch from the separate --pmc FETCH_SIZE / WRITE_SIZE passes, stamp)."""
    import csv
    import glob
    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, 'profiles', 'r*')) if os.path.exists(os.path.join(d, 'alt_pmc_fetch_summary.csv')))
    if not dirs:
        return None, None, None
    src = dirs[-1]
    steady = lambda name: 'mstep_alt_kernel' in name and ', 3,' in name     # noqa: E731  (MODE 3: previous + current step)
    ms = None
    f = os.path.join(src, 'rocprofv3_kernel_stats_bench_graph.csv')
    if os.path.exists(f):
        rows = [r for r in csv.DictReader(open(f)) if steady(r['Name'])]
        calls = sum(int(r['Calls']) for r in rows)
        if calls:
            ms = sum(float(r['TotalDurationNs']) for r in rows) / calls * 1e-6
    tot = 0.0
    for cname, scale in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):   # KiB; gfx950 counts 128-byte reads as 64
        rows = [r for r in csv.DictReader(open(os.path.join(src, f'alt_pmc_{cname.split("_")[0].lower()}_summary.csv')))
                if steady(r['kernel']) and r['counter'] == cname]
        n = sum(int(r['launches']) for r in rows)
        if not n:
            return ms, None, None
        tot += sum(float(r['mean_value']) * int(r['launches']) for r in rows) / n * scale
    stamp = {'profile': os.path.relpath(src, ROOT)}
    try:
        stamp.update(json.load(open(os.path.join(src, 'STAMP.json'))))
    except (OSError, ValueError):
        pass
    return ms, tot, stamp


def large_profile_figures():
    """roofline_large's traffic: L2<->fabric bytes and the rocprofv3 duration of the fused step's launches at 2^24
    interactions, D = 64 (profiles/rNN/large24_*), or (None, None, None)."""
    import csv
    import glob
    dirs = sorted(d for d in glob.glob(os.path.join(ROOT, 'profiles', 'r*')) if os.path.exists(os.path.join(d, 'large24_pmc_summary.csv')))
    if not dirs:
        return None, None, None
    tot, dur = 0.0, None
    rows = list(csv.DictReader(open(os.path.join(dirs[-1], 'large24_pmc_summary.csv'))))
    for cname, scale in (('FETCH_SIZE', 2048.0), ('WRITE_SIZE', 1024.0)):
        sel = [r for r in rows if r['counter'] == cname and 'D64' in r['kernel'] and 'mstep_' in r['kernel']]
        if not sel:
            return None, None, None
        tot += sum(float(r['mean_value']) for r in sel) * scale        # (both launches of the step)
    f = os.path.join(dirs[-1], 'large24_kernel_durations.csv')
    if os.path.exists(f):
        sel = [r for r in csv.DictReader(open(f)) if 'D64' in r['shape'] and 'mstep_' in r['kernel']]
        dur = sum(float(r['mean_us']) for r in sel) * 1e-3 if sel else None
    return tot, dur, os.path.relpath(dirs[-1], ROOT)


GATE_LOSS_TOL = 1e-5     # north_star: "within 1e-5 relative on fp32 loss"


def parity_gate(dev):
    """BASELINE.md §3.2 -- "parity gate before timing counts".  From ONE seeded state (the bench's own tables, interactions and
    initial environments) the timed path -- stat_envs, one epoch of alternating launches (train.py:204-233, :94-157), then the
    fused E-step with the reference's default tie-break (train.py:235-259, :192-196, :268-280) -- and the CPU oracle
    (oracle/invpref_oracle.c, pinned by the goldens recorded from the reference) run the same thing:
      * the epoch's six loss terms (train.py:159-166; the mean over its 31 minibatches) within GATE_LOSS_TOL relative;
      * the E-step on the SAME tables (the device's, after that epoch): every one of the 250 154 assignments, the counts, diff_num
        and the class weights equal, bit for bit (north_star: "bit-exact on integer env assignments").
    The oracle is the checker here, never what is timed.  Returns the `parity_gate` object of the line."""
    import math
    import numpy as np
    from invpref_kdd_2022_amd import synth
    from invpref_kdd_2022_amd.train import LOSS_KEYS, _unrank_permutations
    from oracle import oracle as O
    mgr = build_manager(dev, 0, 1, random_sort=True, scaling='strong')     # (seeds numpy, draws env0, runs stat_envs)
    data = synth.interactions(SEED, U, I, N_PER_GPU, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    env0 = mgr.envs.cpu().numpy().copy()
    cf = [YAHOO[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    got = np.array([[d[k] for k in LOSS_KEYS] for d in mgr.train_epochs(1)], np.float64)
    one_launch = getattr(mgr, '_alt', None) is not None
    th = max(1, min(O.omp_max_threads(), len(os.sched_getaffinity(0)), 16))
    tr = O.ParallelTrainer(tabs, data, env0, implicit=True, batch_size=B_PER_GPU, coefs=cf, lr=YAHOO['lr'], reweight_rec=False,
                           reweight_cls=True, reg_only_embed=True, reg_env_embed=False, threads=th)
    tr.stat_envs()
    want = np.array([tr.train_a_epoch()], np.float64)
    loss_err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))
    # the E-step: same tables on both sides, the same numpy draws of the permutation rows
    tab = O.Tables({k: p.detach().cpu().numpy() for k, p in zip(O.PARAM_NAMES, mgr.state.p_views)})
    st = np.random.get_state()
    diff, cnt = mgr.cluster_and_stat_envs()
    np.random.set_state(st)
    idx = np.concatenate([np.random.randint(0, math.factorial(E), mgr.shard.global_batch_len(k)) for k in range(mgr.batch_num)])
    rows = _unrank_permutations(idx, mgr._eps_base)
    on, oc, od, _ = O.estep(tab, data[:, 0], data[:, 1], data[:, 2], True, old_envs=env0, eps_rows=rows)
    _, ocw, _ = O.stat_envs(on, E)
    mism = int((mgr.envs.cpu().numpy() != on).sum())
    counts_ok = [cnt[e] for e in range(E)] == [int(c) for c in oc] and int(diff) == int(od) \
        and bool((mgr.class_weights.cpu().numpy() == ocw).all())
    ok = bool(loss_err <= GATE_LOSS_TOL and mism == 0 and counts_ok and mgr.alt_error() == 0 and np.isfinite(got).all())
    return {'pass': ok, 'loss_rel_err': loss_err, 'loss_tol': GATE_LOSS_TOL, 'envs_mismatch': mism,
            'counts_diff_class_weights_equal': bool(counts_ok), 'interactions': int(len(on)), 'epoch_steps': int(mgr.batch_num),
            'losses_device': got[0].tolist(), 'losses_oracle': want[0].tolist(), 'loss_keys': list(LOSS_KEYS),
            'path': ('one alternating launch per optimiser step (mstep_alt_kernel)' if one_launch else 'two-launch planned step')
                    + ' + fused E-step (estep_assign_kernel with the stat_envs epilogue), default tie-break',
            'oracle': f'oracle/invpref_oracle.c, {th} threads (M-step epoch), serial E-step with the same permutation rows'}


def sharded_parity_gate(dev, rank, world):
    """The gate of an N-rank run: the SHARDED path the line times -- interactions row-sharded, one all-reduce of the flat gradient
    per optimiser step, dense Adam on every rank, the E-step on each rank's rows with its counts all-reduced -- against the same
    oracle run on the WHOLE problem (strong scaling: the 1-GPU problem; weak: N times the interactions):
      * the epoch's six loss terms (already summed over the ranks by the step's all-reduce) within GATE_LOSS_TOL relative;
      * the parameter replicas bit for bit equal on every rank after that epoch;
      * the E-step on those tables: every assignment of every rank's rows, the all-reduced counts, diff_num and the class weights
        equal to the oracle's, bit for bit.
    Every rank runs it (collectives inside); rank 0 holds the oracle and the verdict."""
    import math
    import numpy as np
    import torch
    import torch.distributed as dist
    from invpref_kdd_2022_amd import synth
    from invpref_kdd_2022_amd.train import LOSS_KEYS, _unrank_permutations
    mgr = build_manager(dev, rank, world, 'rows', random_sort=True)     # (seeds numpy: the same draws on every rank)
    n_total, gbatch = problem_size(world)
    rows = mgr.shard.local_rows().to(dev)

    def whole(local):            # a per-interaction array of the whole problem from the ranks' shards (disjoint rows: a sum)
        full = torch.zeros(n_total, dtype=torch.int64, device=dev)
        full[rows] = local.to(torch.int64)
        dist.all_reduce(full, group=mgr.process_group)
        return full.cpu().numpy()
    env0 = whole(mgr.envs)
    out = mgr.train_epochs(1, sync=False)
    sync_or_die(world)                     # (a collective that never completes ends the run with an error, not a hang)
    got = np.array([[d[k] for k in LOSS_KEYS] for d in mgr.loss_dicts(out)], np.float64)
    mgr.sync_parameters()
    # replicas: the extremes over the ranks of every parameter's bit pattern must coincide
    bits = torch.cat([p.detach().reshape(-1).view(torch.int32) for p in mgr.state.p_views]).to(torch.int64)
    hi, lo = bits.clone(), bits.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=mgr.process_group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=mgr.process_group)
    replicas_differ = int((hi != lo).sum().item())
    st = np.random.get_state()
    diff, cnt = mgr.cluster_and_stat_envs(sync=False)
    sync_or_die(world)
    diff, cnt = int(diff.reshape(-1)[0].item()), [int(c) for c in cnt.reshape(-1).tolist()]
    new_envs = whole(mgr.envs)
    if rank != 0:
        return None
    from oracle import oracle as O
    data = synth.interactions(SEED, U, I, n_total, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    cf = [YAHOO[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    th = max(1, min(O.omp_max_threads(), len(os.sched_getaffinity(0)) // max(1, world), 16))
    tr = O.ParallelTrainer(tabs, data, env0, implicit=True, batch_size=gbatch, coefs=cf, lr=YAHOO['lr'], reweight_rec=False,
                           reweight_cls=True, reg_only_embed=True, reg_env_embed=False, threads=th)
    tr.stat_envs()
    want = np.array([tr.train_a_epoch()], np.float64)
    loss_err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-30)))
    tab = O.Tables({k: p.detach().cpu().numpy() for k, p in zip(O.PARAM_NAMES, mgr.state.p_views)})
    np.random.set_state(st)
    idx = np.concatenate([np.random.randint(0, math.factorial(E), mgr.shard.global_batch_len(k)) for k in range(mgr.batch_num)])
    perm_rows = _unrank_permutations(idx, mgr._eps_base)
    on, oc, od, _ = O.estep(tab, data[:, 0], data[:, 1], data[:, 2], True, old_envs=env0, eps_rows=perm_rows)
    _, ocw, _ = O.stat_envs(on, E)
    mism = int((new_envs != on).sum())
    counts_ok = [int(cnt[e]) for e in range(E)] == [int(c) for c in oc] and int(diff) == int(od) \
        and bool((mgr.class_weights.cpu().numpy() == ocw).all())
    ok = bool(loss_err <= GATE_LOSS_TOL and mism == 0 and counts_ok and replicas_differ == 0 and np.isfinite(got).all())
    return {'pass': ok, 'ranks': world, 'loss_rel_err': loss_err, 'loss_tol': GATE_LOSS_TOL, 'envs_mismatch': mism,
            'counts_diff_class_weights_equal': bool(counts_ok), 'replica_words_that_differ': replicas_differ,
            'interactions': int(n_total), 'epoch_steps': int(mgr.batch_num), 'global_batch': int(gbatch),
            'losses_device': got[0].tolist(), 'losses_oracle': want[0].tolist(), 'loss_keys': list(LOSS_KEYS),
            'path': f'row-sharded x{world}: planned gradient pass + one all-reduce + dense Adam per step; E-step per shard, counts '
                    'all-reduced; default tie-break',
            'oracle': f'oracle/invpref_oracle.c, {th} threads, the whole problem on rank 0'}


def cpu_baseline():
    """The oracle's all-core (OpenMP) loops on the same Yahoo-shaped workload on this box's host cores: M-step
    (gradient + dense Adam) epochs and the E-step, all cores and one core, min of N (SURVEY.md §8(d);
    the reference's CPU path is train.py:204-259).  `kind: port` -- the reference is Python and cannot travel."""
    import numpy as np
    from invpref_kdd_2022_amd import synth
    from oracle import oracle as O
    data = synth.interactions(SEED, U, I, N_PER_GPU, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    env0 = np.random.RandomState(SEED).randint(0, E, N_PER_GPU)
    cf = [YAHOO[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    # threads: what the OpenMP runtime offers, capped by the CPUs this process may run on (affinity mask, cgroup
    # quota -- a container can show 128 CPUs and grant a few); then a quick scan keeps the fastest count
    avail = min(O.omp_max_threads(), len(os.sched_getaffinity(0)))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            avail = max(1, min(avail, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass

    def trainer(th):
        tr = O.ParallelTrainer(tabs, data, env0, implicit=True, batch_size=B_PER_GPU, coefs=cf, lr=YAHOO['lr'],
                               reweight_rec=False, reweight_cls=True, reg_only_embed=True, reg_env_embed=False,
                               threads=th)
        tr.stat_envs()
        return tr
    scan, th = {}, avail
    while th >= 2:
        tr = trainer(th)
        tr.train_a_epoch()              # thread pool / page warm-up
        t0 = time.perf_counter()
        tr.train_a_epoch()
        scan[th] = time.perf_counter() - t0
        th //= 2
    cores = min(scan, key=scan.get) if scan else 1
    res = {}
    # a bounded sample of ~10 s of CPU work: at least (5 epochs, 3 E-steps) and ~7 s on all cores, (3, 1) and ~2.5 s on one
    for th, min_ep, min_es, budget_s in ((cores, 5, 3, 7.0), (1, 3, 1, 2.5)):
        tr = trainer(th)
        if th > 1:
            tr.train_a_epoch()          # thread pool / page warm-up, not timed
        tm, te = [], []
        t_start = time.perf_counter()
        while len(tm) < min_ep or (time.perf_counter() - t_start < 0.85 * budget_s and len(tm) < 400):
            t0 = time.perf_counter()
            tr.train_a_epoch()
            tm.append(time.perf_counter() - t0)
        while len(te) < min_es or (time.perf_counter() - t_start < budget_s and len(te) < 100):
            t0 = time.perf_counter()
            tr.cluster()
            te.append(time.perf_counter() - t0)
            tr.stat_envs()
        res[th] = (min(tm), min(te), len(tm), len(te))
    tm, te, n_ep, n_es = res[cores]
    blended = CLUSTER_INTERVAL * N_PER_GPU / (CLUSTER_INTERVAL * tm + te)
    tm1, te1 = res[1][0], res[1][1]
    return {'value': blended, 'unit': 'interactions/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/invpref_oracle.c OpenMP loops, {cores} threads (the fastest of a scan over {avail}, {avail}/2, ... offered): min of {n_ep} M-step epochs '
                      f'({N_PER_GPU} interactions, 31 minibatches, gradient + dense Adam) and min of {n_es} E-steps of '
                      f'the same Yahoo-shaped workload, blended at the reference cadence (5 epochs : 1 E-step); '
                      f'one-thread figures from {res[1][2]} epochs / {res[1][3]} E-step',
            'threads_offered': avail, 'epoch_s_by_threads': {str(k): v for k, v in scan.items()},
            'mstep_all_cores': N_PER_GPU / tm, 'estep_all_cores': N_PER_GPU / te,
            'mstep_one_core': N_PER_GPU / tm1, 'estep_one_core': N_PER_GPU / te1,
            'value_one_core': CLUSTER_INTERVAL * N_PER_GPU / (CLUSTER_INTERVAL * tm1 + te1)}


SCALING = 'strong'           # set from --scaling in main(): 'strong' = the same 250 154 x 8 192 problem at every N


def problem_size(world, scaling=None, batch=None):
    """(total interactions, global minibatch) of a run on `world` ranks"""
    weak = (scaling or SCALING) == 'weak'
    return N_PER_GPU * (world if weak else 1), (batch if batch else B_PER_GPU * (world if weak else 1))


def build_manager(dev, rank, world, shard_mode=None, random_sort=True, scaling=None, batch=None):
    """random_sort: the reference's DEFAULT E-step path (cluster_use_random_sort=True, train.py:24, :192-196): every E-step
    draws a permutation index per interaction on the host (the reference's own numpy stream) and the device unranks it."""
    import numpy as np
    import torch
    from invpref_kdd_2022_amd import synth
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    from invpref_kdd_2022_amd.train import ImplicitTrainManager
    if shard_mode is not None:
        os.environ['INVPREF_SHARD'] = shard_mode
    n_total, gbatch = problem_size(world, scaling, batch)
    data = synth.interactions(SEED, U, I, n_total, implicit=True)
    tabs = synth.tables(SEED + 7, U, I, E, D)
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=True, reg_env_embed=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tabs.items()})
    np.random.seed(SEED)
    mgr = ImplicitTrainManager(
        model=model, evaluator=StubEvaluator(), device=dev, training_data=torch.from_numpy(data).to(dev),
        batch_size=gbatch, epochs=10 ** 9, cluster_interval=CLUSTER_INTERVAL, evaluate_interval=10 ** 9,
        use_class_re_weight=True, use_recommend_re_weight=False, cluster_use_random_sort=random_sort,
        rank=rank, world_size=world, **YAHOO)
    mgr.stat_envs()
    return mgr


def sync_or_die(world, seconds=120.0):
    """torch.cuda.synchronize() that gives up: in a multi-rank run a collective that never completes (e.g. a captured
    all-reduce the library cannot replay) must end the bench with an error, not hang the node."""
    import torch
    if world == 1:
        torch.cuda.synchronize()
        return
    ev = torch.cuda.Event()
    ev.record()
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > seconds:
            print(json.dumps({'error': f'GPU work did not complete within {seconds:.0f} s (rank {os.environ.get("RANK")}): '
                                       'set INVPREF_NO_COLLECTIVE_GRAPH=1 to keep the sharded loop eager'}), flush=True)
            os._exit(3)
        time.sleep(0.0005)


def timed_run(mgr, world, steps_req, warmup_req):
    """Warm-up + the timed region, both in whole cluster intervals (5 epochs + E-step + stat_envs).
    Returns (seconds max-over-ranks, steps timed, steps of warm-up, pending device results)."""
    import torch
    nb = mgr.batch_num
    per = CLUSTER_INTERVAL * nb
    pending = []

    def interval():
        left = CLUSTER_INTERVAL
        while left > 0:
            out = mgr.train_epochs(left, sync=False)
            pending.append(out)
            left -= out.shape[0]
        # train.py:329-330: cluster() and the stat_envs() that follows it -- ONE launch on one GPU (the kernel's epilogue folds
        # counts, diff_num and class weights; results are views of the E-step ring, nothing is cloned behind the replay)
        pending.extend(mgr.cluster_and_stat_envs(sync=False))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # one-time set-up outside everything: an eager epoch, then capture of the epoch graphs the loop replays
    mgr.train_epochs(1)
    graphs = mgr._graph_warm and mgr.graphs_enabled()
    if graphs:
        try:
            mgr.prepare_graphs(range(1, CLUSTER_INTERVAL + 1))
        except Exception as exc:   # (e.g. a collective that cannot be captured on this node): eager launches
            print(f'bench: graph capture failed ({exc!r}); timing the eager loop', file=sys.stderr)
            mgr.use_graph, graphs = False, False
    n_warm = max(1, -(-warmup_req // per))
    barrier(); sync_or_die(world)
    t0 = time.perf_counter()
    for _ in range(n_warm):
        interval()
    sync_or_die(world); barrier()
    est = (time.perf_counter() - t0) / n_warm                     # seconds per interval (first replays included)
    if world > 1:
        t = torch.tensor([est], dtype=torch.float64, device=mgr.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        est = float(t.item())
    n_int = max(-(-steps_req // per), int(MIN_TIMED_S / est) + 1)
    n_int = max(1, min(n_int, max(-(-steps_req // per), int(MAX_TIMED_S / est))))
    pending.clear()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_int):
        interval()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if dt < MIN_TIMED_S and world == 1:     # the estimate included first-replay costs: top the region up
        extra = int((MIN_TIMED_S - dt) / (dt / n_int) * 1.1) + 1
        for _ in range(extra):
            interval()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_int += extra
    for p in pending:  # every epoch's losses, every E-step's diff_num and env counts were really produced
        assert bool(torch.isfinite(p.double()).all()), 'non-finite training result'
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=mgr.device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, n_int * per, n_warm * per, bool(graphs and mgr._graphs)


def device_step_times(mgr, world):
    """Device time of the M-step and the E-step alone, HIP events on the launch stream (torch's current stream
    is the stream every kernel of this package is enqueued on): whole epochs -- one graph replay per run of epochs
    on a single GPU -- so the figure is the step as the training loop really runs it, kernel boundaries included."""
    import torch
    nb, reps = mgr.batch_num, 6
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    mgr.train_epochs(CLUSTER_INTERVAL, sync=False)
    torch.cuda.synchronize()
    ev[0].record()
    for r in range(reps):
        mgr.train_epochs(CLUSTER_INTERVAL, sync=False)
        ev[r + 1].record()
    torch.cuda.synchronize()
    ms_step = min(a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:])) / (CLUSTER_INTERVAL * nb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    mgr.cluster_and_stat_envs(sync=False)
    e1.record()
    torch.cuda.synchronize()
    ms_eager = e0.elapsed_time(e1)     # (an eagerly issued cluster(): the host's permutation draws sit inside)
    # the E-step as the timed loop runs it: one replay of the captured graph (device time, HIP events around each replay)
    ms_replay = None
    if world == 1 and mgr.graphs_enabled() and not mgr._pure:
        fused = mgr._fused_estep_ok()
        g, _, _ = mgr._estep_graph(mgr.cluster_use_random_sort, fused=fused, combined=True)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for i, (a, b) in enumerate(evs):
            a.record(); g.replay(); b.record()
        torch.cuda.synchronize()
        ms_replay = sorted(a.elapsed_time(b) for a, b in evs[2:])[5]
    return ms_step, ms_eager, ms_replay


def eager_kernel_times(mgr):
    """Per-launch time of the step's kernels from HIP events around EAGER launches (one epoch): the interval from
    the end of the previous kernel to the end of this one, i.e. kernel + launch gap.  The stream is kept backlogged
    so that the stamps are taken by the GPU, not while it waits for the host."""
    import numpy as np
    import torch
    from invpref_kdd_2022_amd import _capi
    if getattr(mgr, '_raw_ptrs', None) is None:
        mgr._raw_setup()
    nb = mgr.batch_num
    ev = {'m0': [], 'm1': [], 'a1': []}
    fused = mgr.use_plan and mgr.world_size == 1 and not mgr._unfused
    torch.cuda._sleep(int(4e6))
    for k in range(nb):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        mgr._raw_step(k, mgr.alpha, mid_event=e1)   # fused: the library records e1 between the step's two launches
        e2.record()
        ev['m0'].append(e0); ev['m1'].append(e1); ev['a1'].append(e2)
    torch.cuda.synchronize()
    med = lambda a, b: float(np.median([x.elapsed_time(y) for x, y in zip(ev[a], ev[b])]))  # noqa: E731
    out = {'step_ms_eager_events': med('m0', 'a1')}
    if fused:
        out['launch1_ms_eager_events'], out['launch2_ms_eager_events'] = med('m0', 'm1'), med('m1', 'a1')
    else:
        out['mstep_ms_eager_events'], out['adam_ms_eager_events'] = med('m0', 'm1'), med('m1', 'a1')
    return out


def step_bench(dev, Us, Is, Es, Ds, Bs, n_steps, zipf=False, seed=5, note=''):
    """The planned fused step (M-step + dense Adam, the headline's kernels) at another shape: `n_steps` DIFFERENT
    minibatches of Bs synthetic interactions, ping-pong parameter buffers, all steps captured in one HIP graph and timed
    with HIP events on the launch stream -- the training loop's conditions without the manager around it.  Returns the
    whole-step figures (both launches, kernel boundaries included) against the 8 TB/s yardstick."""
    import numpy as np
    import torch
    from invpref_kdd_2022_amd import ops, plan as planlib, synth
    data = synth.interactions(seed, Us, Is, n_steps * Bs, implicit=True, zipf=zipf)
    rs = np.random.RandomState(seed + 1)
    tabs = synth.tables(seed + 2, Us, Is, Es, Ds)
    P = [torch.from_numpy(tabs[k]).to(dev) for k in ops.PARAM_NAMES]
    P2 = [torch.zeros_like(p) for p in P]
    M = [torch.zeros_like(p) for p in P]
    V = [torch.zeros_like(p) for p in P]
    # (the interaction columns as the manager holds them: one contiguous array each)
    cu, ci = np.ascontiguousarray(data[:, 0]), np.ascontiguousarray(data[:, 1])
    cy = np.ascontiguousarray(data[:, 2].astype(np.float32))
    t0 = time.perf_counter()
    host_plans = [planlib.build_row_plan(cu[k * Bs:(k + 1) * Bs], ci[k * Bs:(k + 1) * Bs], cy[k * Bs:(k + 1) * Bs], Us, Is,
                                         factor_num=Ds, env_num=Es) for k in range(n_steps)]
    plan_s = (time.perf_counter() - t0) / n_steps
    t0 = time.perf_counter()
    plans = [planlib.upload(hp, dev) for hp in host_plans]
    torch.cuda.synchronize()
    upload_s = (time.perf_counter() - t0) / n_steps
    del host_plans
    e = torch.from_numpy(rs.randint(0, Es, n_steps * Bs).astype(np.int64)).to(dev)
    yt = torch.from_numpy(data[:, 2].astype(np.float32)).to(dev)
    # sample weights as the managers' epochs take them since round 6: class_weights[env] formed in the kernel from the E class
    # weights (train.py:274-278; INVPREF_WEIGHTS_BY_ENV) -- no N-length array read per interaction
    from invpref_kdd_2022_amd import _capi
    _, cw, _ = ops.stat_envs(e, Es, ops.Workspace(dev), want_sample_weights=False)
    ws = ops.Workspace(dev)
    losses = torch.zeros(6, device=dev)
    cf = [YAHOO[k] for k in ('invariant_coe', 'env_aware_coe', 'env_coe', 'L2_coe', 'L1_coe', 'alpha')]
    flags = ops.flags_of(True, False, True, True, False) | _capi.WEIGHTS_BY_ENV

    def run():
        a, b = P, P2
        for k in range(n_steps):
            sl = slice(k * Bs, (k + 1) * Bs)
            ops.mstep_rows_adam(a, b, M, V, plans[k], e[sl], yt[sl], cw, Bs, cf, flags, losses, k + 1, 0.005, ws)
            a, b = b, a
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
        g.replay()
        torch.cuda.synchronize()
        best, reps = 1e30, max(1, int(2000 // max(1, n_steps)) if Bs <= 70000 else 2)
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (reps * n_steps))
    assert bool(torch.isfinite(losses).all())
    Pn = sum(p.numel() for p in P)
    nbytes = Bs * (32 + 16 * Ds) + 24 * Pn
    res = {'ms_per_step': best, 'interactions_per_s': Bs / (best * 1e-3), 'bound': 'hbm',
           'achieved': nbytes / (best * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': nbytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_step': nbytes,
           'cache_resident': 16 * Pn < 200e6,
           'shape': {'users': Us, 'items': Is, 'envs': Es, 'factor_num': Ds, 'minibatch': Bs, 'minibatches': n_steps,
                     'ids': 'zipf' if zipf else 'uniform'},
           'flat_buffer_MB': 4 * Pn / 1e6, 'plan_build_s': plan_s, 'plan_upload_s': upload_s,
           'timing': f'HIP events around {reps} replays of a graph of {n_steps} fused steps (different minibatch plans, '
                     'ping-pong parameter buffers); whole step = both launches'}
    if note:
        res['note'] = note
    del plans, P, P2, M, V
    torch.cuda.empty_cache()
    return res


def other_configs(dev):
    """BASELINE.json configs 3 and 5 (and config 4's large-batch variant) on ONE GPU, driver-run: the planned fused step
    at the MovieLens shape (MovieLens_InvPref.py:17-26), at the MIND shape (MIND_InvPref.py:17-25: the full minibatch
    and the 1/8 row share one of config 5's eight ranks walks), and the Yahoo shape with one minibatch per epoch."""
    out = {}
    out['movielens_config3'] = step_bench(dev, 6040, 3706, 8, 128, 65536, 16, seed=31,
                                          note='BASELINE.json configs[2]: 2^20 uniform synthetic interactions, 16 minibatches')
    out['mind_config5_full_minibatch'] = step_bench(dev, 50000, 51283, 16, 256, 262144, 4, seed=41,
                                                    note='BASELINE.json configs[4] on one GPU: the whole 262 144-row minibatch')
    out['mind_config5_rank_share'] = step_bench(dev, 50000, 51283, 16, 256, 32768, 8, seed=43,
                                                note='the 1/8 row share of a MIND minibatch one of eight ranks walks, '
                                                     'fused form (a sharded rank runs gradient pass + all-reduce + Adam)')
    out['yahoo_config4_large_batch'] = step_bench(dev, U, I, E, D, N_PER_GPU, 2, zipf=True, seed=SEED,
                                                  note='SURVEY 8(d)-4: B = N = 250 154, one optimiser step per epoch')
    return out


def roofline_large(dev):
    """SURVEY.md 8(d) "Roofline launch": the same fused step on cache-exceeding launches -- uniformly random interactions
    over tables of 400 000 users x 100 000 items (256 MB .. 1 GB per flat buffer, four to five buffers: far beyond the
    256 MiB Infinity Cache), D in {64, 128, 256} with E = 4 / 8 / 16, 2^24 interactions per launch (SURVEY's size;
    INVPREF_BENCH_LARGE_LOG2N for another; the plan of such a launch is a few seconds of native host work)."""
    res, small = {}, {}
    lg = int(os.environ.get('INVPREF_BENCH_LARGE_LOG2N', '24'))   # one launch over 2^24 uniform-random rows
    keep = ('frac', 'achieved', 'ms_per_step', 'algorithmic_bytes_per_step', 'shape', 'plan_build_s', 'plan_upload_s')
    for Dl, El, lg_r03 in ((64, 4, 22), (128, 8, 21), (256, 16, 20)):
        r = step_bench(dev, 400000, 100000, El, Dl, 1 << lg, 1, seed=5 + Dl)
        r['kernel'] = 'the fused M-step + Adam step (same kernels as the headline)'
        res[f'D{Dl}_E{El}'] = r
        # (the sizes rounds 2 and 3 reported -- 2^22 / 2^21 / 2^20 interactions, two minibatches -- for continuity: with 10 to
        #  40 times fewer interactions per table row the Adam stream weighs more and the evaluation less)
        small[f'D{Dl}_E{El}'] = {kk: vv for kk, vv in step_bench(dev, 400000, 100000, El, Dl, 1 << lg_r03, 2, seed=5 + Dl).items()
                                 if kk in keep}
    head = dict(res['D64_E4'])
    head['sweep'] = {k: {kk: v[kk] for kk in keep} for k, v in res.items()}
    head['sweep_r03_sizes'] = small
    # L2 <-> fabric bytes of the D = 64 launch pair at THIS size, from the committed profile (separate PMC passes) with its
    # rocprofv3 duration beside it
    tr, dur, src = large_profile_figures()
    head['traffic'] = tr
    head['traffic_profile'] = src
    head['rocprofv3_ms_per_step'] = dur
    return head


def estep_random_sort_timing(dev, n=20):
    """The reference's DEFAULT E-step path (cluster_use_random_sort=True, train.py:24, :192-196) against the plain one at
    the Yahoo shape: device time of the captured E-step (+ the count / weight half of stat_envs) per replay -- HIP events
    around each replay, the permutation indices already in the pinned buffer the kernel reads -- and, apart, the host's
    share: the reference's own np.random.randint call per minibatch (same numpy stream), which in a training loop runs
    while the GPU is still busy with the epochs enqueued before it."""
    import numpy as np
    import torch
    out = {}
    for rs in (False, True):
        mgr = build_manager(dev, 0, 1, random_sort=rs)
        mgr.train_epochs(1)
        mgr.prepare_graphs([1])
        g, eps_buf, _ = mgr._estep_graph(rs, fused=mgr._fused_estep_ok(), combined=True)
        host_s = 0.0
        if rs:
            t0 = time.perf_counter()
            draws = [mgr._eps_index() for _ in range(4)]
            host_s = (time.perf_counter() - t0) / 4
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n + 3):
            if rs:
                torch.cuda.synchronize()                       # (the previous replay has read the buffer)
                if eps_buf.is_cuda:
                    eps_buf.copy_(torch.from_numpy(draws[i % 4]))
                else:
                    eps_buf.numpy()[:] = draws[i % 4]
            if i >= 3:
                ev[i - 3][0].record()
            g.replay()
            if i >= 3:
                ev[i - 3][1].record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        if rs:
            out.update(estep_random_sort_ms=ms, estep_random_sort_host_draw_ms=host_s * 1e3)
        else:
            out['estep_plain_ms'] = ms
        del mgr
        torch.cuda.empty_cache()
    out['ratio'] = out['estep_random_sort_ms'] / out['estep_plain_ms']
    out['note'] = ('device time of one replay of the captured E-step (ONE launch: estep_assign_kernel with the stat_envs fold as its epilogue), median of %d; '
                   'random sort = 1 byte of permutation index per interaction read from pinned host memory + the row looked up '
                   'in an LDS table; host draws = np.random.randint per minibatch, the reference\'s own numpy stream' % n)
    return out


def test_loader(n_users=U, n_items=I, n_test=5400, mask_max=32, n_truth=10, seed=11):
    """a data-loader stand-in of a test shape: n_test test users x n_items items, 1 .. mask_max train items to mask and n_truth
    ground-truth items per user (default: the Yahoo test shape, Yahoo_InvPref_Implicit.py:43-48 evaluates top-k 3/5/7 in test
    batches of 1 024)"""
    import numpy as np
    rs = np.random.RandomState(seed)
    users = sorted(rs.choice(n_users, n_test, replace=False).tolist())
    mask = {u: set(rs.choice(n_items, rs.randint(1, mask_max + 1), replace=False).tolist()) for u in users}
    truth = {u: set(rs.choice(n_items, n_truth, replace=False).tolist()) for u in users}

    class Loader:
        all_test_users_by_sorted_list = users
        get_sorted_all_test_users_ground_truth = [truth[u] for u in users]

        @staticmethod
        def user_mask_items(u):
            return mask[u]
    return Loader()


def yahoo_test_loader():
    return test_loader()


def end_to_end(dev, env_num, factor_num, epochs=200):
    """What a user of the reference's Yahoo driver waits for (Yahoo_InvPref_Implicit.py:24-41, main(): :56-159; the loop:
    train.py:282-342): manager construction, train(silent=True) for `epochs` epochs (the reference runs 1 000) with the real
    ImplicitTestManager every 10 epochs, the E-step every 5 with the reference's default random tie-break -- wall seconds
    from the constructor to train()'s return, split into where they go, and interactions per second over that wall."""
    import numpy as np
    import torch
    from invpref_kdd_2022_amd import synth
    from invpref_kdd_2022_amd.evaluate import ImplicitTestManager
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    from invpref_kdd_2022_amd.train import ImplicitTrainManager
    data = torch.from_numpy(synth.interactions(SEED, U, I, N_PER_GPU, implicit=True)).to(dev)
    tabs = synth.tables(SEED + 7, U, I, env_num, factor_num)
    loader = yahoo_test_loader()
    torch.cuda.synchronize()
    spent = {'evaluation': 0.0, 'graph_capture': 0.0}

    def timed(fn, key):
        def wrapped(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            spent[key] += time.perf_counter() - t
            return out
        return wrapped
    t0 = time.perf_counter()
    model = InvPrefImplicit(U, I, env_num, factor_num, reg_only_embed=True, reg_env_embed=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in tabs.items()})
    evaluator = ImplicitTestManager(model, loader, test_batch_size=1024, top_k_list=[3, 5, 7], use_item_pool=False)
    np.random.seed(SEED)
    mgr = ImplicitTrainManager(model=model, evaluator=evaluator, device=dev, training_data=data, batch_size=B_PER_GPU,
                               epochs=epochs, cluster_interval=CLUSTER_INTERVAL, evaluate_interval=10,
                               use_class_re_weight=True, use_recommend_re_weight=False, test_begin_epoch=0, **YAHOO)
    evaluator.evaluate = timed(evaluator.evaluate, 'evaluation')
    cap_graph, cap_estep = mgr._graph_for, mgr._estep_graph

    def graph_for(n):   # (time only the calls that capture)
        return cap_graph(n) if mgr._graph_key(n) in mgr._graphs else timed(cap_graph, 'graph_capture')(n)

    def estep_graph(with_eps, fused=False, combined=True):
        key = (mgr.state.p_views[0].data_ptr(), mgr.envs.data_ptr(), mgr.users_tensor.data_ptr(), with_eps, fused, combined)
        return cap_estep(with_eps, fused, combined) if key in mgr._estep_graphs \
            else timed(cap_estep, 'graph_capture')(with_eps, fused, combined)
    mgr._graph_for, mgr._estep_graph = graph_for, estep_graph
    t_ctor = time.perf_counter() - t0
    (losses, _), (tests, test_epochs), (diffs, _, cluster_epochs) = mgr.train(silent=True)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert len(losses) == epochs and len(tests) == epochs // 10 + 1 and len(diffs) == epochs // CLUSTER_INTERVAL
    assert all(np.isfinite(list(d.values())).all() for d in losses)
    plan_s = float(getattr(mgr, 'plan_build_s', 0.0))
    rest = wall - t_ctor - spent['evaluation'] - spent['graph_capture'] - plan_s
    return {'wall_s': wall, 'epochs': epochs, 'interactions_per_s_over_wall': epochs * N_PER_GPU / wall,
            'split_s': {'construction': t_ctor, 'plan_build': plan_s, 'graph_capture': spent['graph_capture'],
                        'evaluation': spent['evaluation'], 'training_and_readbacks': rest},
            'evaluations': len(tests), 'e_steps': len(diffs), 'env_num': env_num, 'factor_num': factor_num,
            'cluster_use_random_sort': True,
            'note': 'train(silent=True): epochs enqueued in runs up to the next evaluate / cluster event, losses / diff_num / '
                    'env counts read back once at the end; the reference runs 1 000 epochs of this (same cadence)'}


def eval_timing(dev):
    """ImplicitTestManager.evaluate() (models.py:393-407, evaluate.py:76-135) at the test shapes of the three
    implicit BASELINE configurations -- Yahoo (5 400 test users x 1 000 items, top-k 3/5/7, test batch 1 024:
    Yahoo_InvPref_Implicit.py:43-48; the reference's CPU path took 4.8 s per call in the survey container),
    MovieLens (6 040 x 3 706, D = 128, top-k 10/20/30, test batch 2 048: MovieLens_InvPref.py:45-46) and MIND (50 000 test users
    x 51 283 items, D = 256, top-k 5/10/20/40, test batch 256: MIND_InvPref.py:45-46) -- wall seconds per call, min of 3, and
    the device time of predict_kernel alone over all the call's batches (the score matrix is the call's bulk:
    n_test x items x (2 D flops, 4 bytes written))."""
    import torch
    from invpref_kdd_2022_amd.evaluate import ImplicitTestManager
    from invpref_kdd_2022_amd.models import InvPrefImplicit
    out = {}
    for name, (nu, ni, ne, nd, n_test, tb, topk, mask_max, n_truth) in dict(
            yahoo=(U, I, E, D, 5400, 1024, [3, 5, 7], 32, 10),
            movielens=(6040, 3706, 8, 128, 6040, 2048, [10, 20, 30], 300, 20),
            mind=(50000, 51283, 16, 256, 50000, 256, [5, 10, 20, 40], 60, 10)).items():
        loader = test_loader(nu, ni, n_test, mask_max, n_truth)
        model = InvPrefImplicit(nu, ni, ne, nd).to(dev)
        tm = ImplicitTestManager(model, loader, test_batch_size=tb, top_k_list=list(topk), use_item_pool=False)
        tm.evaluate()                       # builds the CSR arrays from the python sets once
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = tm.evaluate()
            ts.append(time.perf_counter() - t0)
        assert 0.0 <= res['ndcg'][topk[1]] <= 1.0
        # predict alone, device time: every test batch's score matrix
        users = tm._users
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for lo in range(0, n_test, tb):
            r = model.predict(users[lo:lo + tb].contiguous())
        e1.record()
        torch.cuda.synchronize()
        pred_ms = e0.elapsed_time(e1)
        flops, byts = 2.0 * n_test * ni * nd, 4.0 * n_test * ni
        out[name] = {'evaluate_s': min(ts), 'test_users': n_test, 'items': ni, 'factor_num': nd, 'top_k': topk, 'test_batch': tb,
                     'predict_ms': pred_ms, 'predict_tflops': flops / (pred_ms * 1e-3) / 1e12,
                     'predict_score_matrix_GBs_written': byts / (pred_ms * 1e-3) / 1e9}
        del model, tm, loader, r
        torch.cuda.empty_cache()
    out['yahoo']['reference_cpu_s_survey_container'] = 4.8
    # (kept flat for continuity with rounds 1-5: the Yahoo figure)
    out.update({k: out['yahoo'][k] for k in ('evaluate_s', 'test_users', 'items', 'top_k')})
    out['reference_cpu_s_survey_container'] = 4.8
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1550)
    ap.add_argument('--warmup', type=int, default=155)
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong',
                    help='strong (default): 250 154 interactions and the 8 192-row global minibatch at every N (BASELINE configs[3]); '
                         'weak: 250 154 interactions and 8 192 rows PER GPU (rounds 1-5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-gate', action='store_true', help='profiling runs only: the line then says so')
    ap.add_argument('--no-extras', action='store_true', help='skip roofline_large / eval timing (profiling runs)')
    ap.add_argument('--spawn', action='store_true', help='launch the ranks as child processes even for --gpus 1 (checks the launcher path)')
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))      # (nothing in this process has touched the GPU)

    import numpy as np  # noqa: F401
    import torch
    from invpref_kdd_2022_amd import parallel
    global SCALING
    SCALING = args.scaling

    # (INVPREF_BENCH_BACKEND=gloo INVPREF_BENCH_SAME_DEVICE=1: a rehearsal of the N-rank flow on a ONE-GPU box -- every
    #  rank on cuda:0, collectives through gloo, so no captured collectives; never what a measurement uses)
    rank, local, world = parallel.init_from_env(os.environ.get('INVPREF_BENCH_BACKEND', 'nccl'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if os.environ.get('INVPREF_BENCH_SAME_DEVICE'):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    # N > 1: BASELINE.json configs[3] as written -- interactions row-sharded, ONE all-reduce of the whole flat gradient
    # buffer per optimiser step; the reduce-scatter form and the user-sharded layout are timed after it
    rccl_ranks = None
    if world > 1:
        os.environ['INVPREF_EXCHANGE'] = 'allreduce'
        probe = torch.ones(1, device=dev)
        torch.distributed.all_reduce(probe)            # (through the backend the run uses: RCCL unless rehearsing on gloo)
        torch.cuda.synchronize()
        rccl_ranks = int(probe.item()) if torch.distributed.get_backend() == 'nccl' else 0
        assert int(probe.item()) == torch.distributed.get_world_size() == world
    def rate(m, steps_, dt_):
        """interactions per second, exactly: whole epochs of m.n_total interactions (the ragged last minibatch is shorter)"""
        return steps_ / m.batch_num * m.n_total / dt_

    # the parity gate first: a path that does not reproduce the oracle is not timed
    gate = None
    if not args.no_parity_gate:
        if world == 1:
            gate = parity_gate(dev)
            failed = not gate['pass']
        else:
            gate = sharded_parity_gate(dev, rank, world)
            verdict = torch.tensor([0 if (gate is None or gate['pass']) else 1], device=dev)
            torch.distributed.all_reduce(verdict)          # (every rank leaves together)
            failed = bool(verdict.item())
        if failed:
            if rank == 0:
                print(json.dumps({'metric': 'training interactions/sec, Yahoo-implicit InvPref', 'value': None,
                                  'unit': 'interactions/s', 'n_gpus': world, 'error': 'parity gate failed: no value is reported',
                                  'parity_gate': gate}))
            sys.exit(4)
        torch.cuda.empty_cache()
    mgr = build_manager(dev, rank, world, 'rows' if world > 1 else None)
    n_total, gbatch = problem_size(world)
    b_loc = gbatch // world                      # rows of a full minibatch on one rank (the byte models below are per rank)
    dt, steps, warm_steps, graphs = timed_run(mgr, world, args.steps, args.warmup)
    value = rate(mgr, steps, dt)
    ms_step_dev, ms_estep_eager, ms_estep_replay = device_step_times(mgr, world)
    n_local = mgr.users_tensor.shape[0]
    ms_estep = ms_estep_replay if ms_estep_replay else ms_estep_eager
    detail = {'mstep_interactions_per_s_per_gpu': b_loc / (ms_step_dev * 1e-3),
              # the E-step rate from ONE REPLAY of the captured E-step (what the timed loop runs); the eagerly issued cluster()
              # -- the host's numpy permutation draws inside the interval -- beside it
              'estep_interactions_per_s_per_gpu': n_local / (ms_estep * 1e-3), 'estep_ms': ms_estep,
              'estep_timing': 'graph replay' if ms_estep_replay else 'eager cluster()',
              'estep_eager_ms': ms_estep_eager, 'estep_eager_interactions_per_s_per_gpu': n_local / (ms_estep_eager * 1e-3),
              'timed_seconds': dt}
    if world > 1 and getattr(mgr, 'cluster_use_random_sort', False):
        # every rank draws the permutation indices of the GLOBAL minibatches (the reference's numpy stream is sequential:
        # a rank cannot skip to its share), so this host time grows with the rank count while the interval does not --
        # printed so that a SCALE record explains itself
        import time as _t
        st_rng = np.random.get_state()
        t0 = _t.perf_counter()
        mgr._eps_index()
        detail['host_draw_ms'] = (_t.perf_counter() - t0) * 1e3
        np.random.set_state(st_rng)
        detail['host_draw_note'] = ('np.random.randint over the global interaction count per E-step, on every rank; it overlaps the '
                                    'interval only while the epochs are launched asynchronously')
    detail.update(eager_kernel_times(mgr))

    P = mgr.state.n
    fused = mgr.use_plan and world == 1 and not mgr._unfused
    # Algorithmic bytes.  SURVEY §8(d) prices the un-fused pair: M-step B*(32+32D) (ids/labels,
    # 4 row reads, 4 gradient-row adds) + Adam 32P (28 B/param + 4 B zeroing).  The fused owner pass never
    # stores the gradient, so it is priced at what it must move: B*(32+16D) + 24P (p,m,v read; p',m',v' written).
    bytes_survey = b_loc * (32 + 32 * D) + 32 * P
    alt = fused and getattr(mgr, '_alt', None) is not None     # ONE launch per step, the evaluating side alternating
    if fused:
        nbytes = b_loc * (32 + 16 * D) + 24 * P
        kname, knames = 'mstep_eval_kernel', ['mstep_eval_kernel', 'mstep_apply_kernel']
        what = 'the whole optimiser step: mstep_eval_kernel (user jobs: evaluation, records, fused Adam) + mstep_apply_kernel (item jobs, fold, fused Adam)'
        if alt:
            kname, knames = 'mstep_alt_kernel', ['mstep_alt_kernel']
            what = ('the whole optimiser step = ONE launch of mstep_alt_kernel (csrc/step_alt.hpp): the evaluating side -- users '
                    'on even steps, items on odd ones -- applies the previous step\'s pending update to its rows, evaluates, '
                    'applies its own update and pushes contribution rows for the other side; fold blocks at the head of the '
                    'launch finish the small tables of the previous step')
    else:
        nbytes = bytes_survey
        kname = 'mstep_eval_kernel' if mgr.use_plan else 'mstep_atomic_kernel'
        knames = [kname]
        what = 'the whole optimiser step: gradient pass + all-reduce + stand-alone Adam'
    achieved = nbytes / (ms_step_dev * 1e-3) / 1e9
    # the second byte model: what the step would have to move if user rows the minibatch does not touch
    # were left alone (the exact deferred form was built in round 4 and measured SLOWER -- profiles/r04/deferred_adam_ab.txt,
    # 25.3 vs 18.9 us per step -- and was removed in round 6; the run is the dense form, so `frac` uses the dense bytes)
    untouched = 0.0
    if fused and getattr(mgr, '_raw_batches', None):
        untouched = sum(mgr.model.user_num - int(torch.unique(b[3]).numel()) for b in mgr._raw_batches) / len(mgr._raw_batches)
    nbytes_touched = nbytes - 24 * 2 * D * untouched
    traffic, traffic_stamp = pmc_traffic_bytes(knames) if world == 1 else (None, None)
    rocprof_ms = rocprof_avg_ms(knames) if world == 1 else None      # (the committed profiles are 1-GPU runs)
    if alt:
        rocprof_ms, traffic, traffic_stamp = alt_profile_figures()
    roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_profile': traffic_stamp,
                'traffic_unit': 'bytes/step (rocprofv3 PMC, FETCH_SIZE x 2 + WRITE_SIZE, separate passes)', 'kernel': what,
                'avg_launch_ms': ms_step_dev, 'algorithmic_bytes_per_launch': nbytes,
                'yardstick_note': 'algorithmic_bytes_per_launch is the UNCHANGED yardstick of rounds 3-4, B(32 + 16 D) + 24 P: '
                                  'ids / labels / weights + 4 row reads per interaction, p, m, v read and written once per step',
                'timing': 'HIP events on the launch stream around replays of whole-epoch graphs, per step '
                          '(kernel boundaries included)' if graphs else 'HIP events on the launch stream around '
                          'whole eagerly issued epochs, per step',
                'cache_resident': True,
                'cache_note': 'all flat buffers (34 MB) sit in the 256 MiB Infinity Cache at this size: the 8 TB/s '
                              'HBM peak is the yardstick north_star names, not the level that serves the bytes; '
                              'roofline_large is the cache-exceeding launch',
                'rocprofv3_avg_launch_ms': rocprof_ms,
                'rocprofv3_note': 'launch-weighted mean duration of the step\'s kernel(s) in the committed profile of this command',
                'GBs_at_survey_unfused_pricing': bytes_survey / (ms_step_dev * 1e-3) / 1e9,
                'touched_rows_model': {'bytes_per_launch': nbytes_touched, 'untouched_user_rows_per_step': untouched,
                                       'achieved': nbytes_touched / (ms_step_dev * 1e-3) / 1e9,
                                       'frac': nbytes_touched / (ms_step_dev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       'note': 'untouched user rows not counted (what a lazy Adam would move); the run '
                                               'itself is the dense form'}}
    if alt:
        # the second byte model: what the ALTERNATING form itself has to move per step -- per interaction ids /
        # labels / weights, 2 partner rows gathered, 2 contribution rows written and read back; the big tables' p, m, v read
        # and written once per TWO steps; the small tables every step -- so that the gain cannot hide in the denominator
        p_small = 2 * E * D + E
        alt_bytes = b_loc * (32 + 24 * D) + 12 * (P - p_small) + 24 * p_small
        roofline['alt_byte_model'] = {'bytes_per_launch': alt_bytes, 'achieved': alt_bytes / (ms_step_dev * 1e-3) / 1e9,
                                      'frac': alt_bytes / (ms_step_dev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      'note': 'B(32 + 24 D) + 12 P_big + 24 P_small: the alternating form streams a side\'s rows '
                                              'once per two steps and passes the other side\'s gradient through contribution rows'}
        roofline['alt'] = {'plans': len(mgr._alt['plans']), 'plan_build_s': mgr._alt['build_s'],
                           'slots_per_round_users_items': list(mgr._alt.get('slots', ())), 'fold_flag_timeouts': mgr.alt_error()}
    out = {
        'metric': 'training interactions/sec, Yahoo-implicit InvPref', 'value': value, 'unit': 'interactions/s',
        'n_gpus': world, 'steps': steps, 'steps_requested': args.steps, 'warmup': warm_steps,
        'warmup_requested': args.warmup, 'ms_per_step': dt / steps * 1e3,
        'higher_is_better': True, 'scaling': SCALING, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'yahoo_r3_implicit_shaped', 'users': U, 'items': I, 'envs': E, 'factor_num': D,
                   'scaling': SCALING, 'interactions_total': n_total, 'global_batch': gbatch,
                   'interactions_per_gpu': n_total // world, 'batch_per_gpu': b_loc,
                   'interactions_counted': 'exact: steps / minibatches-per-epoch x interactions_total (the last minibatch of an '
                                           'epoch has 4 394 rows, not 8 192: rounds 1-5 counted steps x 8 192, +1.5 %)',
                   'estep_every_steps': CLUSTER_INTERVAL * mgr.batch_num,
                   'parallelism': (f'{mgr.shard_mode}-sharded x{world}, 1 all-reduce/step '
                                   f'({4 * (mgr.state.n + 8 - mgr._ar_lo)} B)' if world > 1 else 'single GPU')},
        'timed_path': 'graph' if graphs else 'eager', 'plan_build_s': getattr(mgr, 'plan_build_s', None),
        'roofline': roofline, 'detail': detail,
    }
    out['rccl_ranks'] = rccl_ranks    # ranks that took part in an RCCL all-reduce in front of the run (None: single GPU)
    out['parity_gate'] = gate if gate is not None else {
        'pass': None, 'note': 'not run (--no-parity-gate)'}
    if world > 1:
        # the headline is the literal form BASELINE.json configs[3] names (rows + one all-reduce); beside it: the same
        # split with the exchange as reduce-scatter -> Adam on the rank's slice -> all-gather, and the xGMI-first
        # user-sharded layout.  UNMEASURED on hardware until a SCALE record exists.
        out['config']['parallelism'] = (f'rows-sharded x{world}, exchange {mgr.exchange}: 1 all-reduce of the flat gradient per '
                                        f'optimiser step ({4 * (mgr.state.n + 8 - mgr._ar_lo)} B)')
        del mgr
        torch.cuda.empty_cache()
        os.environ['INVPREF_EXCHANGE'] = 'scatter'
        mgr_a = build_manager(dev, rank, world, 'rows')
        dt_a, steps_a, _, _ = timed_run(mgr_a, world, args.steps, args.warmup)
        out['detail']['rows_scatter'] = {'value': rate(mgr_a, steps_a, dt_a), 'ms_per_step': dt_a / steps_a * 1e3,
                                         'steps': steps_a, 'exchange': 'reduce-scatter(grad) + all-gather(param)',
                                         'flat_buffer_bytes': 4 * mgr_a.state.cap}
        del mgr_a
        torch.cuda.empty_cache()
        # the all-reduce over the rows each GLOBAL minibatch touches + the small tables only (packed on the device)
        os.environ['INVPREF_EXCHANGE'] = 'packed'
        mgr_p = build_manager(dev, rank, world, 'rows')
        dt_p, steps_p, _, _ = timed_run(mgr_p, world, args.steps, args.warmup)
        out['detail']['rows_packed'] = {'value': rate(mgr_p, steps_p, dt_p), 'ms_per_step': dt_p / steps_p * 1e3,
                                        'steps': steps_p, 'exchange': 'pack touched rows -> 1 all-reduce -> unpack',
                                        'all_reduce_bytes_mean': 4 * sum(mgr_p.packed_floats) / len(mgr_p.packed_floats),
                                        'flat_gradient_bytes': 4 * mgr_p.state.n}
        del mgr_p
        torch.cuda.empty_cache()
        os.environ.pop('INVPREF_EXCHANGE', None)
        mgr_u = build_manager(dev, rank, world, 'users')
        dt_u, steps_u, _, _ = timed_run(mgr_u, world, args.steps, args.warmup)
        out['detail']['user_sharded'] = {'value': rate(mgr_u, steps_u, dt_u), 'ms_per_step': dt_u / steps_u * 1e3,
                                         'steps': steps_u, 'all_reduce_bytes': 4 * (mgr_u.state.n + 8 - mgr_u._ar_lo)}
        del mgr_u
        torch.cuda.empty_cache()
        # SURVEY 8(d)-4's large-batch variant: B = N, one optimiser step per epoch (the exchange amortised over 250 154 rows)
        os.environ['INVPREF_EXCHANGE'] = 'allreduce'
        mgr_b = build_manager(dev, rank, world, 'rows', batch=n_total)
        dt_b, steps_b, _, _ = timed_run(mgr_b, world, args.steps, args.warmup)
        out['detail']['large_batch'] = {'value': rate(mgr_b, steps_b, dt_b), 'ms_per_step': dt_b / steps_b * 1e3, 'steps': steps_b,
                                        'global_batch': n_total, 'exchange': 'allreduce'}
        del mgr_b
        torch.cuda.empty_cache()
        # ... and the other scaling mode, same exchange (strong run: the weak figures of rounds 1-5, and vice versa)
        other = 'weak' if SCALING == 'strong' else 'strong'
        mgr_w = build_manager(dev, rank, world, 'rows', scaling=other)
        dt_w, steps_w, _, _ = timed_run(mgr_w, world, args.steps, args.warmup)
        out['detail'][other] = {'value': rate(mgr_w, steps_w, dt_w), 'ms_per_step': dt_w / steps_w * 1e3, 'steps': steps_w,
                                'interactions_total': mgr_w.n_total, 'global_batch': mgr_w.batch_size, 'exchange': 'allreduce'}
        os.environ.pop('INVPREF_EXCHANGE', None)
    if rank == 0 and world == 1 and not args.no_extras:
        del mgr
        torch.cuda.empty_cache()
        out['roofline_large'] = roofline_large(dev)
        out['detail']['configs'] = other_configs(dev)
        # the other configurations INSIDE the record the driver keeps (parsed.roofline): whole-step fractions of 8 TB/s
        rl = out['roofline_large']
        out['roofline']['configs'] = {k: {'frac': v['frac'], 'ms_per_step': v['ms_per_step']} for k, v in out['detail']['configs'].items()}
        out['roofline']['configs'].update({f'large_2p24_{k}': {'frac': v['frac'], 'ms_per_step': v['ms_per_step']}
                                           for k, v in rl['sweep'].items()})
        out['roofline']['configs']['large_2p24_traffic_D64_E4'] = rl.get('traffic')
        out['detail']['evaluation'] = eval_timing(dev)
        est = estep_random_sort_timing(dev)
        out['detail']['estep_random_sort_ms'] = est['estep_random_sort_ms']
        out['detail']['estep_random_sort'] = est
        out['detail']['end_to_end'] = {
            'reference_yahoo_config_E2_D40': end_to_end(dev, 2, 40),
            'baseline_config_E4_D64': end_to_end(dev, E, D)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
