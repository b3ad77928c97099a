#!/usr/bin/env python3
"""Diagnostic (GPU box; not part of the product or tests): the alternating one-launch-per-step form (csrc/step_alt.hpp)
against the two-launch form on the same minibatches.
  * parity: NB steps from the same state through both forms (parameters, moments, the six loss terms per step);
  * per-step time of both forms captured in one HIP graph each (PROBE_STEPS steps, cache state of the training loop);
  * with PROBE_STAMPS=1 (needs a library built with -DALT_STAMPS): phase stamps of one launch of each side.
Shape: PROBE_SHAPE="U,I,E,D,B" (default: the Yahoo shape)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

dev = torch.device('cuda:0')
want_stamps = os.environ.get('PROBE_STAMPS') == '1'
if want_stamps:
    stamps = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
    os.environ['INVPREF_STAMPS'] = hex(stamps.data_ptr())
    if os.environ.get('PROBE_DRAIN') != '1':
        os.environ['INVPREF_STAMPS_NODRAIN'] = '1'
from invpref_kdd_2022_amd import ops, plan as planlib, synth  # noqa: E402

U, I, E, D, B = 15400, 1000, 4, 64, 8192
nb = int(os.environ.get('PROBE_STEPS', '31'))
if os.environ.get('PROBE_SHAPE'):
    U, I, E, D, B = [int(x) for x in os.environ['PROBE_SHAPE'].replace('x', ',').split(',')]
    data = synth.interactions(1, U, I, nb * B, implicit=True, zipf=os.environ.get('PROBE_ZIPF', '1') == '1')
else:
    data = synth.yahoo_like()[:nb * B]
nb = min(nb, len(data) // B)
tabs = synth.tables(2, U, I, E, D)
std = float(os.environ.get('PROBE_STD', '1'))
P0 = [torch.from_numpy(tabs[k] * std).to(dev) for k in ops.PARAM_NAMES]
N = nb * B
y = torch.from_numpy(data[:N, 2].astype(np.float32)).to(dev)
e = torch.from_numpy(np.random.RandomState(3).randint(0, E, N).astype(np.int64)).to(dev)
w = torch.from_numpy(np.random.RandomState(4).rand(N).astype(np.float32)).to(dev)
coefs = (3.35, 9.99, 9.06, 3.13, 0.49, float(os.environ.get('PROBE_ALPHA', '1.9')))
fl = [x == '1' for x in os.environ.get('PROBE_FLAGS', '1,1,1,0,1').split(',')]   # implicit, rw_rec, rw_cls, reg_only_embed, reg_env_embed
flags = ops.flags_of(*fl)
FIRST, LR = 5, 0.005
Pn = sum(p.numel() for p in P0)
nbytes = B * (32 + 16 * D) + 24 * Pn


def mb(k):
    sl = slice(k * B, (k + 1) * B)
    return data[sl, 0], data[sl, 1], data[sl, 2].astype(np.float32)


# ---- the two-launch form
pls = [planlib.build_row_plan(*mb(k), U, I, factor_num=D, env_num=E) for k in range(nb)]
plans = [planlib.upload(p, dev) for p in pls]
ws = ops.Workspace(dev)


def run_two(P, P2, M, V, losses):
    a, b = P, P2
    for k in range(nb):
        sl = slice(k * B, (k + 1) * B)
        ops.mstep_rows_adam(a, b, M, V, plans[k], e[sl], y[sl], w[sl], B, coefs, flags, losses[k], FIRST + k, LR, ws)
        a, b = b, a
    return a


# ---- the alternating form: launch c evaluates minibatch c from side c % 2; a flush ends the run
SLOTS = (planlib.alt_slots_for(mb(0)[0], U, 2), planlib.alt_slots_for(mb(0)[1], I, 2))
if os.environ.get('PROBE_SLOTS'):
    SLOTS = tuple(int(x) for x in os.environ['PROBE_SLOTS'].split(','))
apl = []
for c in range(nb):
    prev = None if c == 0 else mb(c - 1)[:2]
    apl.append(planlib.build_alt_plan(mb(c), prev, c % 2, U, I, factor_num=D, n_partials_prev=apl[-1]['n_tasks'] if c else 0,
                                      slots=SLOTS[c % 2]))
apl.append(planlib.build_alt_plan(None, mb(nb - 1)[:2], nb % 2, U, I, factor_num=D, n_partials_prev=apl[-1]['n_tasks'],
                                  slots=SLOTS[nb % 2]))
aplans = [planlib.upload_alt(p, dev) for p in apl]
aws = ops.AltWorkspace(P0, B, max(p['n_tasks'] for p in apl) + 1)


def run_alt(P, M, V, losses):
    for c in range(nb):
        sl = slice(c * B, (c + 1) * B)
        ops.mstep_alt(P, M, V, aplans[c], e[sl], w[sl], B, B, coefs, flags, losses[c - 1] if c else None, FIRST + c, LR, aws,
                      c & 1)
    ops.mstep_alt(P, M, V, aplans[nb], None, None, B, B, coefs, flags, losses[nb - 1], FIRST + nb - 1, LR, aws, nb & 1)


def fresh():
    return [p.clone() for p in P0], [torch.zeros_like(p) for p in P0], [torch.zeros_like(p) for p in P0]


print(f'shape U={U} I={I} E={E} D={D} B={B}, {nb} steps; slots per round (users, items) {SLOTS}; alt workgroups: ' +
      ' '.join(f'{"UI"[p["side"]]}{planlib.alt_workgroups(p) + 40}' for p in apl[:4]) + ' ... flush ' +
      f'{planlib.alt_workgroups(apl[-1]) + 40}; two-launch {planlib.launch_workgroups(pls[0], 0)} + {planlib.launch_workgroups(pls[0], 1)}')

if os.environ.get('PROBE_STEPWISE'):
    # every step of the two-launch trajectory redone from the SAME state by the alternating form (one evaluating launch +
    # the flush), from either side: separates a wrong update from the drift of two correct trajectories
    Pa, Ma, Va = fresh()
    P2 = [p.clone() for p in Pa]
    la = torch.zeros(nb, 6, device=dev)
    a_, b_ = Pa, P2
    worst = 0.0
    for k in range(nb):
        S = [x.clone() for x in a_], [x.clone() for x in Ma], [x.clone() for x in Va]
        sl = slice(k * B, (k + 1) * B)
        ops.mstep_rows_adam(a_, b_, Ma, Va, plans[k], e[sl], y[sl], w[sl], B, coefs, flags, la[k], FIRST + k, LR, ws)
        a_, b_ = b_, a_
        for side in (0, 1):
            p1 = planlib.build_alt_plan(mb(k), None, side, U, I, factor_num=D)
            p2 = planlib.build_alt_plan(None, mb(k)[:2], 1 - side, U, I, factor_num=D, n_partials_prev=p1['n_tasks'])
            d1, d2 = planlib.upload_alt(p1, dev), planlib.upload_alt(p2, dev)
            Pb, Mb, Vb = [x.clone() for x in S[0]], [x.clone() for x in S[1]], [x.clone() for x in S[2]]
            lb = torch.zeros(6, device=dev)
            ops.mstep_alt(Pb, Mb, Vb, d1, e[sl], w[sl], B, B, coefs, flags, None, FIRST + k, LR, aws, 0)
            ops.mstep_alt(Pb, Mb, Vb, d2, None, None, B, B, coefs, flags, lb, FIRST + k, LR, aws, 1)
            torch.cuda.synchronize()
            rel = [((xa - xb).abs().max() / xa.abs().max()).item() for xa, xb in zip(a_, Pb)]
            rl = ((la[k] - lb).abs() / la[k].abs().clamp_min(1e-12)).max().item()
            worst = max(worst, max(rel), rl)
            if k % 5 == 0 or max(rel) > 1e-5:
                print(f'step {k} side {"UI"[side]}: params ' + ' '.join(f'{x:.1e}' for x in rel) + f' | losses {rl:.1e}')
    print('STEPWISE worst', f'{worst:.2e}', 'OK' if worst < 1e-5 else 'FAILED', 'error word', aws.error())
    sys.exit(0)

if not want_stamps:
    # parity
    Pa, Ma, Va = fresh()
    Pb, Mb, Vb = fresh()
    P2 = [p.clone() for p in Pa]
    la = torch.zeros(nb, 6, device=dev)
    lb = torch.zeros(nb, 6, device=dev)
    Pa = run_two(Pa, P2, Ma, Va, la)
    run_alt(Pb, Mb, Vb, lb)
    torch.cuda.synchronize()
    print('fold-flag wait error word:', aws.error())
    worst = 0.0
    if os.environ.get('PROBE_SELF'):
        # yardstick: the two-launch form against ITSELF under another plan (pull form, other slice lengths): how far two
        # correct summation orders drift apart over these steps
        pls2 = [planlib.build_row_plan(*mb(k), U, I, factor_num=D, env_num=E, per_slice=3, item_per_slice=3, push=False) for k in range(nb)]
        plans_keep, plans[:] = list(plans), [planlib.upload(p, dev) for p in pls2]
        Pd, Md, Vd = fresh()
        Pd2 = [p.clone() for p in Pd]
        ld = torch.zeros(nb, 6, device=dev)
        Pd = run_two(Pd, Pd2, Md, Vd, ld)
        torch.cuda.synchronize()
        plans[:] = plans_keep
        print('  two-launch vs two-launch under another plan: ' + ' '.join(f'{((xa - xd).abs().max() / xa.abs().max()).item():.1e}' for xa, xd in zip(Pa, Pd)))
    if os.environ.get('PROBE_ROWS'):
        for nm, xa, xb in zip(ops.PARAM_NAMES[:4], Pa, Pb):
            dr = (xa - xb).abs().amax(dim=1)
            bad = torch.nonzero(dr > 1e-6 * xa.abs().max()).flatten()
            print(f'  {nm}: {len(bad)} rows differ; first {bad[:12].tolist()}')
            if len(bad):
                r = int(bad[0])
                side = 0 if 'user' in nm else 1
                for c in range(nb):
                    cntc = int((data[c * B:(c + 1) * B, side] == r).sum())
                    if cntc:
                        print(f'      row {r}: {cntc} interactions in minibatch {c}')
                print('      two-launch', xa[r, :4].tolist(), ' alt', xb[r, :4].tolist())
    for nm, xa, xb in zip(ops.PARAM_NAMES, Pa, Pb):
        da = (xa - xb).abs().max().item()
        sc = xa.abs().max().item()
        worst = max(worst, da / sc)
        print(f'  {nm:40s} max |diff| {da:.3e} (max |x| {sc:.3e})')
    for nm, A_, B_ in (('exp_avg', Ma, Mb), ('exp_avg_sq', Va, Vb)):
        print(f'  {nm}: ' + ' '.join(f'{((xa - xb).abs().max() / xa.abs().max().clamp_min(1e-30)).item():.1e}' for xa, xb in zip(A_, B_)))
    dl = ((la - lb).abs() / la.abs().clamp_min(1e-12)).max(dim=0).values
    print('  losses, worst relative difference per term:', ' '.join(f'{x:.1e}' for x in dl.tolist()))
    # bitwise reproducibility of the alternating form
    same = True
    for _ in range(int(os.environ.get('PROBE_REPS', '8'))):
        Pc, Mc, Vc = fresh()
        lc = torch.zeros(nb, 6, device=dev)
        run_alt(Pc, Mc, Vc, lc)
        torch.cuda.synchronize()
        same = same and all(torch.equal(x1, x2) for x1, x2 in zip(Pb + Mb + Vb + [lb], Pc + Mc + Vc + [lc]))
    print('  alternating form bitwise reproducible:', same)
    print('PARITY', 'OK' if worst < 2e-4 and dl.max().item() < 2e-5 and same and aws.error() == 0 else 'FAILED')

    def graph_time(fn, reps=20):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            g.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(reps):
                    g.replay()
                t1.record()
                torch.cuda.synchronize()
                best = min(best, t0.elapsed_time(t1) / (reps * nb) * 1e3)
        return best

    Pt, Mt, Vt = fresh()
    Pt2 = [p.clone() for p in Pt]
    lt = torch.zeros(nb, 6, device=dev)
    us2 = graph_time(lambda: run_two(Pt, Pt2, Mt, Vt, lt))
    usa = graph_time(lambda: run_alt(Pt, Mt, Vt, lt))
    print(f'two-launch form : {us2:6.2f} us per step = {nbytes / us2 / 1e3 / 8000:.3f} of 8 TB/s (59 MB yardstick)')
    print(f'alternating form: {usa:6.2f} us per step = {nbytes / usa / 1e3 / 8000:.3f} of 8 TB/s (same yardstick; flush launch included)')
    print('fold-flag wait error word:', aws.error())
    sys.exit(0)

# ---- stamps of launch ks (and ks + 1: the other side)
ks = int(os.environ.get('PROBE_STAMP_STEP', '10'))
Pt, Mt, Vt = fresh()
lt = torch.zeros(nb, 6, device=dev)
for rep in range(2):
    for c in range(nb):
        sl = slice(c * B, (c + 1) * B)
        if c in (ks, ks + 1):
            torch.cuda.synchronize()
            stamps.zero_()
        ops.mstep_alt(Pt, Mt, Vt, aplans[c], e[sl], w[sl], B, B, coefs, flags, lt[c - 1] if c else None, FIRST + c, LR, aws, c & 1)
        if c in (ks, ks + 1) and rep == 1:
            torch.cuda.synchronize()
            raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.int64)
            pl = apl[c]
            ncls, cls = pl['n_classes'], np.asarray(pl['cls'])
            wg = planlib.alt_workgroups(pl) + 40
            st = raw[:wg]
            kind = np.full(wg, 'pad', dtype=object)
            kind[:33] = "fold"
            for bk in range(40, wg):
                cc, j = (bk - 40) % ncls, (bk - 40) // ncls
                tj = int(cls[cc, 1])
                kind[bk] = 'job' if j < tj else ('stream' if (j - tj) * pl['rows_per_stream_task'] < cls[cc, 3] else 'pad')
            live = st[:, 0] > 0
            t0 = st[live, 0].min()
            end = st[:, 7]
            print(f'== launch {c} (side {"UI"[pl["side"]]}): grid {wg}; ' + ', '.join(f'{k} {int(((kind == k) & live).sum())}' for k in ('fold', 'job', 'stream'))
                  + f'; last end {(end[live].max() - t0) / 100:.2f} us')
            for nm in ('fold', 'job', 'stream'):
                sel = (kind == nm) & live
                if not sel.any():
                    continue
                s0, e0 = (st[sel, 0] - t0) / 100, (end[sel] - t0) / 100
                life = e0 - s0
                print(f'  {nm:6s} n={sel.sum():4d} start med {np.median(s0):5.2f} max {s0.max():5.2f} | life med {np.median(life):5.2f} p90 '
                      f'{np.quantile(life, .9):5.2f} max {life.max():5.2f} | end med {np.median(e0):5.2f} p90 {np.quantile(e0, .9):5.2f} max {e0.max():5.2f}')
            j = st[(kind == 'job') & live].astype(np.float64)
            if len(j):
                ph = np.diff(j, axis=1) / 100
                print('  job phases (us, median): zero red %.2f | descriptor %.2f | pending + Adam(prev) %.2f | flag wait + stage %.2f | '
                      'interactions %.2f | slice meet %.2f | Adam + store %.2f' % tuple(np.median(ph, axis=0)))
                print('  job phases (us, p90)   : ' + ' | '.join('%.2f' % x for x in np.quantile(ph, .9, axis=0)))
                idx = np.flatnonzero((kind == 'job') & live)
                order = idx[np.argsort(-(end[idx] - st[idx, 0]))][:int(os.environ.get('PROBE_TOP', '8'))]
                for bk in order:
                    cc, jj = (bk - 40) % ncls, (bk - 40) // ncls
                    rnd = int(cls[cc, 0]) + jj
                    meta = pl['desc'][rnd, :, 1]
                    act = pl['desc'][rnd, :, 0] >= 0
                    cnts = ((meta >> 9) & 0x3fffff)[act]
                    pend_n = (pl['pend'][rnd, :, 1] - pl['pend'][rnd, :, 0])[act]
                    print('    wg %4d start %.2f life %5.2f end %5.2f phases %s | round %d: slices %d, row counts %s, pending per slice max %d' % (
                        bk, (st[bk, 0] - t0) / 100, (end[bk] - st[bk, 0]) / 100, (end[bk] - t0) / 100,
                        ' '.join('%.2f' % x for x in np.diff(st[bk].astype(np.float64)) / 100), rnd, (meta[0] >> 1) & 31,
                        sorted(set(cnts.tolist()), reverse=True)[:4], int(pend_n.max(initial=0))))
