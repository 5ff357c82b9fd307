#!/usr/bin/env python3
"""How far do GPU and CPU-oracle solves of the C2 acrobot batch drift apart, and why?  (development aid)
Splits problems by whether every line-search decision agreed and prints the cost differences of each class."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ilqr_amd
from ilqr_amd import _lib
from oracle import oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = bench.acrobot_c2(4096, 0)
ws = ilqr_amd.Workspace('acrobot', cfg['N'], B, cfg['dt'], want_trace=True, n_iterations=10)
ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
for f, a in ((_lib.F_INIT, cfg['init'][:B]), (_lib.F_TARGET, cfg['target']), (_lib.F_PARAMS, cfg['params']), (_lib.F_GUESS, cfg['guess'][:B])):
    ws.upload(f, a)
ws.solve(); ws.sync()
tg, cg = ws.download(_lib.F_TRACE), ws.download(_lib.F_COST)
r = O.solve_batched('acrobot', cfg['init'][:B], cfg['target'], cfg['guess'][:B], cfg['Qk'], cfg['Rk'], cfg['QT'], cfg['dt'], cfg['params'], 10, n_threads=16)
tc, cc = r['trace'], r['cost']
same_j = np.array([np.array_equal(tg[b, :, 4], tc[b, :, 4], equal_nan=True) for b in range(B)])
rel = np.abs(cg - cc) / np.abs(cc)
print('problems', B, 'identical line-search decisions:', same_j.sum(), 'differ:', (~same_j).sum())
for name, m in (('same decisions', same_j), ('different decisions', ~same_j)):
    if m.any():
        q = np.quantile(rel[m], [0.5, 0.9, 0.99, 1.0])
        print('  %-20s cost rel diff median %.2e  p90 %.2e  p99 %.2e  max %.2e   within 1e-8: %.4f' % (name, *q, np.mean(rel[m] <= 1e-8)))
# per-iteration growth of the cost difference among problems with the same decisions
for it in range(10):
    d = np.abs(tg[same_j, it, 0] - tc[same_j, it, 0]) / np.abs(tc[same_j, it, 0])
    d = d[np.isfinite(d)]
    print('  iteration %d: start-cost rel diff median %.1e max %.1e (n=%d)' % (it, np.median(d), d.max(), d.size))
# margins of the flipped decisions
for b in np.where(~same_j)[0][:8]:
    it = int(np.argmax(~(np.nan_to_num(tg[b, :, 4], nan=-9) == np.nan_to_num(tc[b, :, 4], nan=-9))))
    print('  problem %d first differs at iteration %d: gpu j=%s cpu j=%s, cost rel diff there %.1e' % (b, it, tg[b, it, 4], tc[b, it, 4], abs(tg[b, it, 0] - tc[b, it, 0]) / abs(tc[b, it, 0])))
