#!/usr/bin/env python3
"""Print how far the GPU path is from the reference fixtures (the numbers the 1e-8 tests bound), per system.
Development aid: run on the GPU box after any change to the arithmetic."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd as amd
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))
for system in ('pendulum', 'acrobot', 'quadcopter'):
    g = np.load(os.path.join(G, system + '_solve.npz'))
    f, A, B = getattr(amd, 'symbolic_dynamics_' + system)()
    for pipe in (1, 2):
        s = amd.ilqr(g['init'], g['target'], g['guess'], float(g['dt']), float(g['start']), float(g['end']), f, A, B, g['Qk'], g['Rk'],
                     g['QT'], g['params'], int(g['n_iterations']), verbose=False, pipeline=pipe)
        st, u, k, K, cost = s.solve()
        print('%-10s pipeline %d: iters %d/%d  states %.1e  inputs %.1e  K %.1e  k %.1e  cost %.1e  same trials %s' % (
            system, pipe, s.iterations_, int(g['iters']), rel(st, g['states']), rel(u, g['inputs']), rel(K, g['K']),
            float(np.max(np.abs(k - g['k'])) / max(np.max(np.abs(g['k'])), np.max(np.abs(g['inputs'])))),
            abs(cost - float(g['cost'])) / abs(float(g['cost'])),
            np.array_equal(s.trace_[:int(g['iters']), 4], g['trace'][:, 4])))
for system in ('acrobot', 'pendulum'):
    t, l = np.load(os.path.join(G, system + '_tracking.npz')), np.load(os.path.join(G, system + '_ltvlqr.npz'))
    ref = np.load(os.path.join(G, 'acrobot_csv.npz')) if system == 'acrobot' else l
    f, A, B = getattr(amd, 'symbolic_dynamics_' + system)()
    K = amd.ltvlqr(ref['states'], ref['inputs'], A, B, l['Qk'], l['Rk'], l['QT'], float(l['dt']), l['params'])
    xs, us = amd.tracking_rollout(ref['states'], ref['inputs'], l['K'], f, t['init'], float(l['dt']), l['params'])
    print('%-10s ltvlqr K %.1e  tracking states %.1e inputs %.1e' % (system, rel(K, l['K']), rel(xs, t['states']), rel(us, t['inputs'])))
