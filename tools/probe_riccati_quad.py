#!/usr/bin/env python3
"""k_riccati vs k_riccati_quad on the headline config (development aid): per-kernel ms from the workspace profile."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, ilqr_amd
from ilqr_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = bench.acrobot_c2(B, 0)
for riccati, overlap in ((1, 1), (3, 1), (3, 0)):
    ws = ilqr_amd.Workspace('acrobot', cfg['N'], B, cfg['dt'], n_iterations=10, riccati=riccati, overlap=overlap)
    ws.set_weights(cfg['Qk'], cfg['Rk'], cfg['QT'])
    for f, a in ((_lib.F_INIT, cfg['init'][:B]), (_lib.F_TARGET, cfg['target']), (_lib.F_PARAMS, cfg['params']), (_lib.F_GUESS, cfg['guess'][:B])):
        ws.upload(f, a)
    import time
    ws.solve(); ws.sync()
    ws.profile(True)
    t = time.perf_counter(); ws.solve(); ws.sync(); dt = time.perf_counter() - t
    prof = ws.profile_read()
    ms, cnt = prof['backward']
    ams, acnt = prof['accept']
    print('overlap=%d riccati=%d: solve %.1f ms (profiled)  backward %.3f ms/launch (%d)  accept %.3f ms  cost[0]=%.12g' % (overlap, riccati, dt * 1e3, ms / max(cnt, 1), cnt, ams / max(acnt, 1), ws.download(_lib.F_COST)[0]))
    ws.profile(False)
    ws.close()
