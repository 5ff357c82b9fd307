#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: one C2 solve (bench workload) + calibration kernels with known byte counts.
k_cost reads U (B*N*m*8 bytes) and one row of X; k_rollout reads G and writes X and U."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ilqr_amd
from ilqr_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = bench.acrobot_c2(B, 0)
ws = ilqr_amd.Workspace('acrobot', cfg['N'], B, cfg['dt'], want_trace=True, n_iterations=10)
ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
for f, a in ((_lib.F_INIT, cfg['init']), (_lib.F_TARGET, cfg['target']), (_lib.F_PARAMS, cfg['params']), (_lib.F_GUESS, cfg['guess'])):
    ws.upload(f, a)
ws.solve(); ws.sync()
import json
info = ws.info()
print('CONFIG ' + json.dumps(dict(system='acrobot', batch=B, n_timesteps=cfg['N'], n_alpha_parallel=info['n_alpha_parallel'], pipeline=info['pipeline'])))
ws.cost(); ws.sync()
tr = ws.download(_lib.F_TRACE)
print('rounds', int(ws.download(_lib.F_LS_ROUNDS).sum()), 'accepts', int(ws.download(_lib.F_ACCEPTS).sum()), 'bwd', int(ws.download(_lib.F_ITERS).sum()))
