#!/usr/bin/env python3
"""Wall time of the drop-in boundary itself (development aid): ilqr(...).solve() with host numpy in and out, on the pendulum
example (C1, one problem) and on the 4096-problem headline batch (C2).  usage: python tools/probe_api.py [reps]"""
import contextlib, io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ilqr_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = bench.golden('pendulum_solve')
fab = ilqr_amd.symbolic_dynamics_pendulum()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        s = ilqr_amd.ilqr(g['init'], g['target'], g['guess'], float(g['dt']), float(g['start']), float(g['end']), *fab, g['Qk'], g['Rk'], g['QT'],
                          g['params'], int(g['n_iterations']))
        out = s.solve()
    ts.append(time.perf_counter() - t0)
print('c1 pendulum example: median %.2f ms, min %.2f ms, cost %.16g (reference %.16g), %d iterations' % (1e3 * np.median(ts), 1e3 * min(ts), out[4], g['cost'], s.iterations_))
cfg = bench.acrobot_c2(4096, 0)
fab = ilqr_amd.symbolic_dynamics_acrobot()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    s = ilqr_amd.ilqr(cfg['init'], cfg['target'], cfg['guess'], cfg['dt'], 0, 3, *fab, cfg['Qk'], cfg['Rk'], cfg['QT'], cfg['params'], 10)
    out = s.solve()
    ts.append(time.perf_counter() - t0)
    print('  c2 through the class: %.1f ms' % (1e3 * ts[-1]), flush=True)
nbytes = sum(o.nbytes for o in out[:4]) + cfg['guess'].nbytes
print('c2 api: median %.1f ms, min %.1f ms; %.2f GB across the boundary' % (1e3 * np.median(ts), 1e3 * min(ts), nbytes / 1e9))
