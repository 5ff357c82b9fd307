import time, numpy as np, sys
sys.path.insert(0,'/root/repo')
import ilqr_amd as amd
g=np.load('/root/repo/tests/golden/pendulum_solve.npz')
f,A,B=amd.symbolic_dynamics_pendulum()
for sysn in ('pendulum','acrobot'):
    g=np.load('/root/repo/tests/golden/%s_solve.npz'%sysn)
    f,A,B=getattr(amd,'symbolic_dynamics_'+sysn)()
    for rep in range(3):
        t0=time.time()
        s=amd.ilqr(g['init'],g['target'],g['guess'],float(g['dt']),float(g['start']),float(g['end']),f,A,B,g['Qk'],g['Rk'],g['QT'],g['params'],int(g['n_iterations']),verbose=False)
        out=s.solve()
        print(sysn,'solve() %.1f ms'%(1e3*(time.time()-t0)),'iters',s.iterations_,'cost',out[4])
