import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ilqr_amd
from ilqr_amd import _lib
cfg = bench.acrobot_c2(4096, 0)
for rep in range(3):
    T = {}
    t0 = time.perf_counter()
    ws = ilqr_amd.Workspace('acrobot', cfg['N'], 4096, cfg['dt'], want_trace=True, n_iterations=10)
    T['create'] = time.perf_counter() - t0; t0 = time.perf_counter()
    ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
    for f, a in ((_lib.F_INIT, cfg['init']), (_lib.F_TARGET, cfg['target']), (_lib.F_PARAMS, cfg['params']), (_lib.F_GUESS, cfg['guess'])):
        ws.upload(f, a)
    T['upload'] = time.perf_counter() - t0; t0 = time.perf_counter()
    ws.solve(); ws.sync()
    T['solve'] = time.perf_counter() - t0; t0 = time.perf_counter()
    for f in (_lib.F_STATES, _lib.F_INPUTS, _lib.F_KFF, _lib.F_KFB):
        t1 = time.perf_counter()
        a = ws.download(f)
        T['dl_%d' % f] = time.perf_counter() - t1
    T['download'] = time.perf_counter() - t0; t0 = time.perf_counter()
    ws.close()
    T['close'] = time.perf_counter() - t0
    print({k: round(1e3 * v, 1) for k, v in T.items()})
t0 = time.perf_counter(); x = np.empty((4096, 3001, 4)); x[:] = 0; print('first touch of 393 MB: %.1f ms' % (1e3 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); x[:] = 1; print('second touch: %.1f ms' % (1e3 * (time.perf_counter() - t0)))
