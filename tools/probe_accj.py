"""Histogram of the accepted line-search trial per iteration for one bench configuration (development aid, GPU box).
usage: python tools/probe_accj.py c5 [batch]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ilqr_amd import _lib

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_BATCH[name]
cfg = bench.workload(name, B, 0)
class A: alphas = 0; streams = 0; riccati = 0; pipeline = 0; overlap = None
ws = bench.make_workspace(cfg, B, 0, A)
ws.solve(); ws.sync()
tr = ws.download(_lib.F_TRACE)
j = tr[:, :, 4]
for it in range(j.shape[1]):
    jj = j[:, it]; a = jj[~np.isnan(jj)]
    need = np.where(a < 0, 59, a + 1)
    h = np.histogram(need, bins=[0, 1.5, 2.5, 4.5, 8.5, 16.5, 32.5, 48.5, 58.5, 60])[0]
    print('it %d active %6d mean need %5.1f  hist(1,2,3-4,5-8,9-16,17-32,33-48,49-58,exhausted) %s' % (it, a.size, need.mean() if a.size else 0, h))
ws.close()
