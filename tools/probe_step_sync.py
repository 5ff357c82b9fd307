"""What a per-step stream synchronisation costs the headline step (development aid, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from ilqr_amd import _lib
B = 4096
cfg = bench.workload('c2', B, 0)
class A: alphas = 0; streams = 0; riccati = 0; pipeline = 0; overlap = None
ws = bench.make_workspace(cfg, B, 0, A)
for prof in (False, True):
    ws.profile(prof)
    for mode in ('queued', 'sync each step', 'sync + 3 downloads'):
        ws.solve(); ws.sync()
        K = 8
        t0 = time.perf_counter()
        for _ in range(K):
            ws.solve()
            if mode != 'queued':
                ws.sync()
            if mode == 'sync + 3 downloads':
                ws.download(_lib.F_COST); ws.download(_lib.F_ITERS); ws.download(_lib.F_STATUS)
        ws.sync()
        print('profile %-5s %-20s %.3f ms per step' % (prof, mode, 1e3 * (time.perf_counter() - t0) / K))
# host time to enqueue one solve
ws.sync(); t0 = time.perf_counter(); ws.solve(); t1 = time.perf_counter(); ws.sync()
print('host time to enqueue one solve: %.3f ms' % (1e3 * (t1 - t0)))
