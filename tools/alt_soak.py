#!/usr/bin/env python3
"""Diagnostic (GPU box): the bench's loop -- runs of 5 epochs as graph replays, an E-step after each -- through the manager,
alternating form against the two-launch form, interval by interval: where do the loss traces part, is anything non-finite,
did a workgroup ever give up on the fold flags.  SOAK_INTERVALS (default 60), SOAK_GRAPH=0 for eager launches."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from invpref_kdd_2022_amd import synth
from invpref_kdd_2022_amd.models import InvPrefImplicit
from invpref_kdd_2022_amd.train import ImplicitTrainManager
from oracle import oracle as O
DEV = torch.device('cuda:0')
NI = int(os.environ.get('SOAK_INTERVALS', '60'))
U, I, E, D, B = 15400, 1000, 4, 64, 8192
data = synth.yahoo_like()
BENCH = os.environ.get('SOAK_BENCH') == '1'   # bench.py's model flags, seeds and graph preparation
tabs = synth.tables(17373331 + 7 if os.environ.get('SOAK_BENCH') else 6, U, I, E, D)
class Stub:
    def evaluate(self): return {}
res = {}
for alt in os.environ.get('SOAK_ALTS', '1,0').split(','):
    os.environ['INVPREF_ALT'] = alt
    if os.environ.get('SOAK_GRAPH') == '0':
        os.environ['INVPREF_NO_GRAPH'] = '1'
    model = InvPrefImplicit(U, I, E, D, reg_only_embed=BENCH, reg_env_embed=not BENCH)
    model.load_state_dict({k: torch.from_numpy(tabs[k]) for k in O.PARAM_NAMES})
    np.random.seed(17373331 if BENCH else 3)
    mgr = ImplicitTrainManager(model=model, evaluator=Stub(), device=DEV, training_data=torch.from_numpy(data).to(DEV),
                               batch_size=B, epochs=10 ** 9, cluster_interval=5, evaluate_interval=10 ** 9, lr=0.005,
                               invariant_coe=3.351991776096847, env_aware_coe=9.988658447411407, env_coe=9.06447753571379,
                               L2_coe=3.1351402017943117, L1_coe=0.4935216278026648, alpha=1.9053711444718746,
                               use_class_re_weight=True, use_recommend_re_weight=False,
                               cluster_use_random_sort=os.environ.get('SOAK_RANDOM_SORT', '1') == '1')
    mgr.stat_envs()
    if BENCH:
        mgr.train_epochs(1)
        mgr.prepare_graphs(range(1, 6))
    tr, diffs = [], []
    for it in range(NI):
        out = mgr.train_epochs(5, sync=False)
        d, _ = mgr.cluster_and_stat_envs(sync=False)
        tr.append(out.cpu().numpy())
        if os.environ.get('SOAK_FLAGS') and mgr._alt is not None:
            ws = mgr._alt['ws']
            fl = ws.buf[ws.err_off - 63 * 4: ws.err_off + 4].view(torch.int32).cpu().numpy()
            print('flags', fl.tolist(), 'step', mgr.state.step, 'sched', mgr._sched['state'].cpu().numpy().tolist())
        if os.environ.get('SOAK_TRACE'):
            mx = [float(p_.abs().max()) for p_ in mgr.state.p_views]
            am = [int(p_.abs().amax(dim=1).argmax()) if p_.dim() == 2 else 0 for p_ in mgr.state.p_views]
            print(f'alt={alt} interval {it}: losses {np.round(tr[-1][-1], 4)} err {mgr.alt_error()} max|.| ' + ' '.join(f'{x:.3g}' for x in mx) + f' argmax rows {am[:4]}')
        diffs.append(int(d.item()) if torch.is_tensor(d) else int(d))
        if not np.isfinite(tr[-1]).all():
            print(f'alt={alt}: NON-FINITE losses in interval {it} (steps {mgr.state.step - 155} .. {mgr.state.step}):')
            print(tr[-1])
            break
    print(f'alt={alt}: {len(tr)} intervals, steps {mgr.state.step}, alt error word {mgr.alt_error()}, last losses {tr[-1][-1]}, diffs {diffs[:6]} ...')
    res[alt] = np.concatenate(tr)
ks = list(res)
res = {'1': res[ks[0]], '0': res[ks[-1]]}
n = min(len(res['1']), len(res['0']))
d = np.abs(res['1'][:n] / res['0'][:n] - 1).max(axis=1)
print('max relative loss difference per 25 epochs:', np.array2string(np.array([d[i:i + 25].max() for i in range(0, n, 25)]), precision=2))
