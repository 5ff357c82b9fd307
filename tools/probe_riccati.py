#!/usr/bin/env python3
"""Time the backward pass alone (linearize + Riccati, split schedule) on the C5 quadcopter shard (development aid).
usage: python tools/probe_riccati.py [scale] [system] [riccati]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sysname = sys.argv[2] if len(sys.argv) > 2 else 'quadcopter'
g = np.load(G + '/%s_solve.npz' % sysname)
B = int(32768 * scale)
N, n = g['guess'].shape[0], g['init'].shape[0]
ric = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ws = ilqr_amd.Workspace(sysname, N, B, float(g['dt']), n_iterations=10, pipeline=2, riccati=ric, overlap=1)
ws.set_weights(g['Qk'], g['Rk'], g['QT']); ws.upload(_lib.F_PARAMS, g['params']); ws.upload(_lib.F_TARGET, g['target'])
ws.upload(_lib.F_INIT, np.tile(g['init'], (B, 1))); ws.upload(_lib.F_GUESS, np.broadcast_to(g['guess'], (B,) + g['guess'].shape).copy())
ws.rollout(); ws.sync()
ws.profile(True)
for _ in range(5):
    ws.backward(0)
ws.sync()
print({k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in ws.profile_read().items() if v[1]})
print('dJ', ws.download(_lib.F_DJ)[0], 'active status', np.bincount(ws.download(_lib.F_STATUS), minlength=4))
