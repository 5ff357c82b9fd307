#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes: ONE step of a bench configuration (bench.workload / bench.make_workspace, the very
objects bench.py times) + for c2 a k_cost launch as a calibration kernel with a known byte count (it reads U: B*N*m*8 bytes).
usage: python tools/pmc_probe.py [config [B]]      config = c2 (default) | c3 | c4 | c4_w8 | c5 | pendulum | acrobot_big | ...
Prints a CONFIG line (what tools/summarize_pmc.py writes into the traffic file bench.py matches against) and the launch
order of the line-search rounds (widths), so the summariser can tell the rounds of one search apart."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import ilqr_amd
from ilqr_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH[name]
args = argparse.Namespace(alphas=0, streams=0, riccati=0, overlap=None, device_wrap=False, pipeline=0)
cfg = bench.workload(name, B, 0)
ws = bench.make_workspace(cfg, B, 0, args)
one = {'ltvlqr': ws.ltvlqr, 'mpc': lambda: (ws.mpc_shift(1), ws.mpc_solve(False))}.get(cfg['kind'], ws.solve)
one()
if cfg['kind'] != 'solve':      # a single launch per call: the first one pays the code load; average over a few
    one()
    one()
ws.sync()
info = ws.info()
print('CONFIG ' + json.dumps(dict(name=name, system=cfg['system'], batch=B, n_timesteps=cfg['N'], n_alpha_parallel=info['n_alpha_parallel'],
                                  pipeline=info['pipeline'], ls_widths=info['ls_widths'], kind=cfg['kind'])))
if cfg['kind'] == 'solve':
    if name == 'c2':
        ws.cost()
        ws.sync()
    print('rounds', int(ws.download(_lib.F_LS_ROUNDS).sum()), 'accepts', int(ws.download(_lib.F_ACCEPTS).sum()), 'bwd',
          int(ws.download(_lib.F_ITERS).sum()))
