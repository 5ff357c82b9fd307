#!/usr/bin/env python3
"""Time the BASELINE configs C3/C4/C5 at (per-GPU) full size through the workspace API (development aid)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
which = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
t0 = time.time()
if which == 'c3':
    csv, g = np.load(G + '/acrobot_csv.npz'), np.load(G + '/acrobot_ltvlqr.npz')
    B, N = int(65536 * scale), 3000
    ws = ilqr_amd.Workspace('acrobot', N, B, 1e-3, zero_first_state=False)
    ws.set_weights(g['Qk'], g['Rk'], g['QT']); ws.upload(_lib.F_PARAMS, g['params'])
    rng = np.random.default_rng(1)
    st = np.broadcast_to(csv['states'], (B, N + 1, 4)) + 1e-2 * rng.standard_normal((B, N + 1, 4)); st[0] = csv['states']
    u = np.broadcast_to(csv['inputs'], (B, N, 1)) + 1e-1 * rng.standard_normal((B, N, 1)); u[0] = csv['inputs']
    print('gen %.1fs' % (time.time() - t0)); t0 = time.time()
    ws.upload(_lib.F_STATES, st); ws.upload(_lib.F_INPUTS, u); ws.sync()
    print('upload %.1fs' % (time.time() - t0))
    for _ in range(2):
        t0 = time.time(); ws.ltvlqr(); ws.sync(); print('ltvlqr %.1f ms' % (1e3 * (time.time() - t0)))
    # the consumer of the gain schedule -- closed-loop tracking from perturbed initial states, K stays in HBM
    init = np.array([-np.pi / 2, 0, 0, 0]) + rng.uniform(-np.pi / 8, np.pi / 8, (B, 4)) * np.array([1, 1, 0.2, 0.2])
    ws.upload(_lib.F_INIT, init); ws.upload(_lib.F_ALPHA, np.zeros(B)); ws.sync()
    for _ in range(2):
        t0 = time.time(); ws.forward(with_cost=False); ws.sync(); print('tracking rollout %.1f ms' % (1e3 * (time.time() - t0)))
    xs = ws.download(_lib.F_STATES_OUT)
    err = np.linalg.norm(xs[:, -1, :2] - st[:, -1, :2], axis=1)
    print('tracking: final angle error median %.3g, p99 %.3g rad; finite %s' % (np.median(err), np.quantile(err, 0.99), np.isfinite(xs).all()))
    K = ws.download(_lib.F_KFB)
    print('K0 rel err vs golden', np.max(np.abs(K[0] - g['K'])) / np.max(np.abs(g['K'])), 'finite', np.isfinite(K).all())
else:
    sysname = {'c4': 'cartpole', 'c5': 'quadcopter'}[which]
    g = np.load(G + '/%s_solve.npz' % sysname)
    B = int({'c4': 131072, 'c5': 32768}[which] * scale)
    N = g['guess'].shape[0]
    n = g['init'].shape[0]
    rng = np.random.default_rng(2 if which == 'c4' else 3)
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    ws = ilqr_amd.Workspace(sysname, N, B, float(g['dt']), want_trace=True, n_iterations=10, n_alpha_parallel=W,
                            target_batched=(which == 'c5'))
    print(ws.info())
    ws.set_weights(g['Qk'], g['Rk'], g['QT']); ws.upload(_lib.F_PARAMS, g['params'])
    if which == 'c4':
        init = rng.uniform(-0.1, 0.1, (B, 4)); target = g['target']
        guess = np.broadcast_to(g['guess'], (B, N, 1)).copy()
    else:
        init = np.zeros((B, 12)); target = np.tile(g['target'], (B, 1)); target[:, :3] += rng.uniform(-0.5, 0.5, (B, 3))
        guess = np.broadcast_to(g['guess'], (B, N, 4)).copy()
    ws.upload(_lib.F_INIT, init); ws.upload(_lib.F_TARGET, target); ws.upload(_lib.F_GUESS, guess); ws.sync()
    print('setup %.1fs' % (time.time() - t0))
    ws.profile(True)
    for _ in range(2):
        t0 = time.time(); ws.solve(); ws.sync(); dt = time.time() - t0
        print('solve %.1f ms  -> %.0f solves/s' % (1e3 * dt, B / dt))
    print({k: (round(v[0] / max(v[1], 1), 2), v[1]) for k, v in ws.profile_read().items() if v[1]})
    st = ws.download(_lib.F_STATUS); cost = ws.download(_lib.F_COST)
    print('status counts', np.bincount(st, minlength=4), 'cost finite', np.isfinite(cost).mean(), 'median cost', np.median(cost))
