import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import ilqr_amd
from ilqr_amd import _lib
g=np.load('/root/repo/tests/golden/quadcopter_solve.npz')
B,N=32768,400
ws=ilqr_amd.Workspace('quadcopter',N,B,float(g['dt']))
ws.set_weights(0.1*np.eye(12),0.01*np.eye(4),100*np.eye(12)); ws.upload(_lib.F_PARAMS,g['params'])
rng=np.random.default_rng(0)
st=g['states'][None]+1e-2*rng.standard_normal((B,N+1,12)); u=g['inputs'][None]+1e-2*rng.standard_normal((B,N,4))
ws.upload(_lib.F_STATES,st); ws.upload(_lib.F_INPUTS,u); ws.sync()
for _ in range(3):
    t0=time.time(); ws.ltvlqr(); ws.sync(); print('quadcopter ltvlqr %d x %d: %.1f ms'%(B,N,1e3*(time.time()-t0)))
K=ws.download(_lib.F_KFB); print(np.isfinite(K).all())
