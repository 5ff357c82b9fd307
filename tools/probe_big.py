#!/usr/bin/env python3
"""Large-batch throughput of the pendulum / acrobot example problems on one GPU (the north star's 8-GPU target is
>= 1e6 pendulum/acrobot solves/s: one GPU's share is 125 k/s).  usage: python tools/probe_big.py system B [n_iterations [W]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
system, B = sys.argv[1], int(sys.argv[2])
n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 10
W = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = np.load(G + '/%s_solve.npz' % system)
N, n = g['guess'].shape[0], g['init'].shape[0]
rng = np.random.default_rng(5)
ws = ilqr_amd.Workspace(system, N, B, float(g['dt']), want_trace=True, n_iterations=n_it, n_alpha_parallel=W)
print(ws.info())
ws.set_weights(g['Qk'], g['Rk'], g['QT']); ws.upload(_lib.F_PARAMS, g['params']); ws.upload(_lib.F_TARGET, g['target'])
ws.upload(_lib.F_INIT, g['init'][None] + rng.uniform(-0.2, 0.2, (B, n)))
blk = min(B, 1024)
guess = g['guess'][None] * (1.0 + 0.1 * rng.standard_normal((blk, 1, 1))) + 0.01 * rng.standard_normal((blk, N, g['guess'].shape[1]))
ws.upload(_lib.F_GUESS, np.ascontiguousarray(np.broadcast_to(guess[None], (B // blk, blk) + guess.shape[1:]).reshape(B, N, -1)))
ws.sync(); ws.profile(True)
for _ in range(2):
    t0 = time.time(); ws.solve(); ws.sync(); dt = time.time() - t0
    print('%s B=%d N=%d %d iterations: %.1f ms -> %.0f solves/s' % (system, B, N, n_it, 1e3 * dt, B / dt))
print({k: (round(v[0] / max(v[1], 1), 2), v[1]) for k, v in ws.profile_read().items() if v[1]})
st = ws.download(_lib.F_STATUS); it = ws.download(_lib.F_ITERS)
info = ws.info(); rounds = ws.download(_lib.F_LS_ROUNDS); tr = ws.download(_lib.F_TRACE)
ran = np.isfinite(tr[:, :, 1]).sum()
need = np.where(tr[:, :, 4] >= 0, tr[:, :, 4] + 1, info['n_trials'])[np.isfinite(tr[:, :, 1])].sum()
print('rounds per search %.2f, trials needed by the sequential search %.1f' % (rounds.sum() / ran, need / ran))
print('status counts', np.bincount(st, minlength=4), 'mean iterations', it.mean())
# histogram of the trials each search needed (accepted j + 1; n_trials when it ran out): what a round schedule is sized on
needed = np.where(tr[:, :, 4] >= 0, tr[:, :, 4] + 1, info['n_trials'])[np.isfinite(tr[:, :, 1])].astype(int)
edges = [1, 2, 3, 5, 9, 17, 25, 33, 41, 49, 60]
print('trials needed: ' + '  '.join('%d-%d: %.3f' % (lo, hi - 1, np.mean((needed >= lo) & (needed < hi))) for lo, hi in zip(edges[:-1], edges[1:])))
