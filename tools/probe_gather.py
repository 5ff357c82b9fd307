"""Where the per-step all-gather of {cost, iters, status} spends its time (development aid, GPU box, one rank over RCCL)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
import bench
from ilqr_amd import _lib
from ilqr_amd.sharding import all_gather_summary, pack_summary
B = 4096
cfg = bench.workload('c2', B, 0)
class A: alphas = 0; streams = 0; riccati = 0; pipeline = 0; overlap = None
ws = bench.make_workspace(cfg, B, 0, A)
ws.solve(); ws.sync()
def t(f, n=20):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return 1e3 * (time.perf_counter() - t0) / n
print('download cost   %.3f ms' % t(lambda: ws.download(_lib.F_COST)))
print('download iters  %.3f ms' % t(lambda: ws.download(_lib.F_ITERS)))
print('download status %.3f ms' % t(lambda: ws.download(_lib.F_STATUS)))
rec = pack_summary(ws.download(_lib.F_COST), ws.download(_lib.F_ITERS), ws.download(_lib.F_STATUS))
print('pack            %.3f ms' % t(lambda: pack_summary(ws.download(_lib.F_COST), ws.download(_lib.F_ITERS), ws.download(_lib.F_STATUS))))
print('all_gather      %.3f ms' % t(lambda: all_gather_summary(rec, B, device=torch.device('cuda', 0))))
dist.destroy_process_group()
