#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE python (build container only).

The reference (/root/reference/python/{ilqr,ltvlqr,symbolic_dynamics}.py) is imported read-only,
driven with the parameter blocks of its own example scripts, and only NUMBERS are written out.
No reference source travels: the fixtures hold inputs and expected outputs.  This tool refuses
to run without /root/reference (it is never run on the GPU box).

Fixtures (all float64, reference numpy layout):
  dyn_<sys>.npz            random-point f/A/B of the lambdified callables (acrobot: the reference's own
                           symbolic_dynamics_acrobot(); pendulum: the reference expression with the
                           sp.sp.sin typo of symbolic_dynamics.py:12 fixed; the other systems: the
                           build's sympy re-derivation of the MATLAB specs, lambdified; the floating base has no example
                           script in the reference, its configuration is synthetic -- see cfg_floating_base)
  mat_kat_<sys>.npz        one-step pairs x_i,u_i -> x_{i+1} from the reference's MATLAB .mat trajectories
  acrobot_csv.npz          the reference's committed acrobot_swingup_{states,inputs}.csv as arrays
  <sys>_solve.npz          full ilqr.solve() of the example config + per-iteration trace + Armijo ratios
  <sys>_passes.npz         teacher-forced single passes on the initial rollout (backward; forward at 3 alphas)
  <sys>_ltvlqr.npz         ltvlqr gains (acrobot: on the CSV trajectory; pendulum: on its solve() output)
  <sys>_tracking.npz       the closed-loop simulation the two *_ltvlqr_tracking.py examples run with those gains from a
                           perturbed initial state (their own perturbation + a few seeded ones), reference f
  <sys>_batch.npz          8 seeded problems, short horizon, n_iterations=3, solved one by one
  acrobot_solve10.npz      the acrobot example stopped after 10 iterations (the bench's fixed iteration count; problem 0 of
                           the headline batch): `python tools/make_golden.py solve10`
  {c2,c4,c5}_sample.npz    the first 32 / 16 / 8 problems of bench.py's random C2 / C4 / C5 batches (bench.workload, seed 0) solved
                           by the reference class at the bench's 10 iterations, in parallel worker processes; trace, cost, exit
                           status in full, states / inputs / gains subsampled in time: `python tools/make_golden.py samples`
"""
import contextlib
import io
import os
import sys
import time

sys.dont_write_bytecode = True
REF = '/root/reference/python'
if not os.path.isdir(REF):
    sys.exit('make_golden.py: /root/reference is absent -- fixtures can only be regenerated in the build container')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, 'codegen'))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import scipy.io  # noqa: E402

import models  # noqa: E402  (own sympy models)
from ilqr import ilqr as RefIlqr  # noqa: E402  (reference)
from ltvlqr import ltvlqr as ref_ltvlqr  # noqa: E402  (reference)
import symbolic_dynamics as ref_dyn  # noqa: E402  (reference)

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


class TracedIlqr(RefIlqr):
    """The unmodified reference class; the overrides only record what passes through."""

    def __init__(self, *a):
        super().__init__(*a)
        self.log = []

    def backwards_pass(self):
        r = super().backwards_pass()
        self.log.append(('b', float(self.expected_cost_reduction_), float(self.expected_cost_reduction_grad_),
                         float(self.expected_cost_reduction_hess_)))
        return r

    def forwards_pass(self, learning_rate):
        self.log.append(('f', float(learning_rate)))
        return super().forwards_pass(learning_rate)

    def compute_cost(self, states, inputs):
        c = super().compute_cost(states, inputs)
        self.log.append(('c', float(c)))
        return c


def run_solve(obj):
    """obj.solve() with stdout captured; returns outputs + trace (n_it,6) + flat Armijo ratios."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        states, inputs, k, K, cost = obj.solve()
    text = buf.getvalue()
    status = 1 if 'optimal trajectory' in text else (2 if 'low learning rate' in text else 0)
    log = obj.log
    assert log[0][0] == 'c'
    cur = log[0][1]
    rows, ratios, ratio_off = [], [], [0]
    i = 1
    while i < len(log):
        assert log[i][0] == 'b', log[i]
        _, dj, g, h = log[i]
        row = [cur, dj, g, h, -1.0, np.nan]
        i += 1
        j = 0
        while i < len(log) and log[i][0] == 'f':
            lr = log[i][1]
            new_cost = log[i + 1][1]
            ratio = (cur - new_cost) / (lr * g + lr * lr * h)
            ratios.append(ratio)
            if ratio > 0.1:
                row[4], row[5] = float(j), new_cost
                cur = new_cost
            j += 1
            i += 2
        ratio_off.append(len(ratios))
        rows.append(row)
    assert abs(cur - cost) == 0.0
    return dict(states=states, inputs=inputs, k=k, K=K, cost=np.float64(cost), status=np.int32(status),
                iters=np.int32(len(rows)), trace=np.array(rows), armijo_ratios=np.array(ratios),
                armijo_offsets=np.array(ratio_off, np.int64))


def lambdified(name):
    """(f, A, B) numpy callables per system; acrobot comes from the reference itself."""
    if name == 'acrobot':
        with contextlib.redirect_stdout(io.StringIO()):
            return ref_dyn.symbolic_dynamics_acrobot()
    return models.BUILDERS[name]().lambdify()


CONFIG = {}


def cfg_pendulum():   # example_pendulum_swingup.py:14-41
    dt, t0, t1 = 0.005, 0, 5
    N = np.arange(t0, t1, dt).shape[0]
    return dict(dt=dt, start=t0, end=t1, init=np.array([0, 0]), target=np.array([np.pi, 0]),
                guess=0.1 * np.ones((N, 1)), Qk=np.zeros((2, 2)), Rk=0.001 * np.eye(1), QT=100 * np.eye(2),
                params=np.array([1, 9.8, 1]), n_iterations=50)


def cfg_acrobot():    # example_acrobot_swingup.py:14-48
    dt, t0, t1 = 0.001, 0, 3
    N = np.arange(t0, t1, dt).shape[0]
    np.random.seed(1)
    guess = 0.5 * np.random.normal(loc=0, scale=1, size=(N, 1))
    QT = 10 * np.eye(4); QT[0, 0] = 1000; QT[1, 1] = 1000
    return dict(dt=dt, start=t0, end=t1, init=np.array([-np.pi / 2, 0, 0, 0]), target=np.array([np.pi / 2, 0, 0, 0]),
                guess=guess, Qk=np.zeros((4, 4)), Rk=0.01 * np.eye(1), QT=QT, params=np.array([1, 1, 2, 1, 9.8]),
                n_iterations=50)


def cfg_cartpole():   # example_cartpole.m:4-36, swing-up variant
    dt, t0, t1 = 0.005, 0, 5
    N = np.arange(t0, t1, dt).shape[0]
    rng = np.random.default_rng(2)
    QT = np.eye(4); QT[1, 1] = 100
    return dict(dt=dt, start=t0, end=t1, init=np.zeros(4), target=np.array([0, np.pi, 0, 0]),
                guess=0.05 * rng.standard_normal((N, 1)), Qk=np.zeros((4, 4)), Rk=0.0001 * np.eye(1), QT=QT,
                params=np.array([1, 1, 1, 9.8]), n_iterations=50)


def cfg_quadcopter():  # example_quadcopter.m:4-34
    dt, t0, t1 = 0.005, 0, 2
    N = np.arange(t0, t1, dt).shape[0]
    target = np.zeros(12); target[:3] = 1
    return dict(dt=dt, start=t0, end=t1, init=np.zeros(12), target=target,
                guess=0.8 * 9.8 / 4 / 4 * np.ones((N, 4)), Qk=np.zeros((12, 12)), Rk=0.1 * np.eye(4),
                QT=100 * np.eye(12), params=np.array([0.05, 0.05, 0.09, 0.8, 9.8, 0.33, 4, 0.05]), n_iterations=100)


def cfg_car():   # example_car_accelerate.m:4-35 (time span end-exclusive as in the python reference; parameters = [])
    dt, t0, t1 = 0.005, 0, 4
    N = np.arange(t0, t1, dt).shape[0]
    return dict(dt=dt, start=t0, end=t1, init=np.zeros(4), target=np.array([4.0, 0, 1.0, 0]), guess=np.zeros((N, 2)),
                Qk=np.zeros((4, 4)), Rk=0.0001 * np.eye(2), QT=100 * np.eye(4), params=np.zeros(1), n_iterations=50)


def cfg_double_integrator():   # example_double_integrator.m:5-33 (unconstrained: the python reference has no bounds)
    dt, t0, t1 = 0.01, 0, 3
    N = np.arange(t0, t1, dt).shape[0]
    return dict(dt=dt, start=t0, end=t1, init=np.array([-2.0, 0]), target=np.zeros(2), guess=np.ones((N, 1)),
                Qk=np.zeros((2, 2)), Rk=0.001 * np.eye(1), QT=100 * np.eye(2), params=np.array([1.0]), n_iterations=50)


def cfg_double_pendulum():   # example_double_pendulum.m:4-38 (horizon shortened from 4 s to 2 s to keep the fixture small)
    dt, t0, t1 = 0.001, 0, 2
    N = np.arange(t0, t1, dt).shape[0]
    rng = np.random.default_rng(1)
    QT = 10 * np.eye(4); QT[0, 0] = 1000; QT[1, 1] = 1000
    return dict(dt=dt, start=t0, end=t1, init=np.array([-np.pi / 2, 0, 0, 0]), target=np.array([np.pi / 2, 0, 0, 0]),
                guess=0.1 * rng.standard_normal((N, 2)), Qk=np.zeros((4, 4)), Rk=0.001 * np.eye(2), QT=QT,
                params=np.array([1, 1, 2, 1, 9.8]), n_iterations=50)


def cfg_floating_base():   # no example script in the reference (symbolic_dynamics_floating_base.m only): a synthetic quadruped
    # trunk -- 9 kg, feet on the ground under the corners, standing guess (every foot carries a quarter of the weight),
    # asked to shift, rise and yaw within half a second.  gravity enters as [0; 0; g] (:70), hence g < 0.  The trunk starts
    # at the origin (feet 0.3 below): row 0 of the reference's state trajectory is never written (ilqr.py:31-42), so with a
    # non-zero initial state the first step's feedback acts on a spurious deviation and the search runs out after 3 iterations.
    dt, t0, t1 = 0.005, 0, 0.5
    N = np.arange(t0, t1, dt).shape[0]
    mass, g = 9.0, -9.81
    feet = [0.2, 0.15, -0.3, 0.2, -0.15, -0.3, -0.2, 0.15, -0.3, -0.2, -0.15, -0.3]
    init = np.zeros(12)
    target = np.zeros(12); target[:3] = [0.1, -0.05, 0.05]; target[3:6] = [0.0, 0.05, 0.2]
    guess = np.zeros((N, 12)); guess[:, 2::3] = -mass * g / 4
    return dict(dt=dt, start=t0, end=t1, init=init, target=target, guess=guess, Qk=np.zeros((12, 12)), Rk=1e-4 * np.eye(12),
                QT=100 * np.eye(12), params=np.array([0.07, 0.26, 0.242, mass, g] + feet), n_iterations=30)


CONFIG = {'pendulum': cfg_pendulum, 'acrobot': cfg_acrobot, 'cartpole': cfg_cartpole, 'quadcopter': cfg_quadcopter,
          'car': cfg_car, 'double_integrator': cfg_double_integrator, 'double_pendulum': cfg_double_pendulum,
          'floating_base': cfg_floating_base}


def make(c, fab, cls=TracedIlqr, **over):
    c = dict(c, **over)
    f, A, B = fab
    return cls(c['init'], c['target'], c['guess'], c['dt'], c['start'], c['end'], f, A, B, c['Qk'], c['Rk'], c['QT'],
               c['params'], c['n_iterations'])


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez(path, **arrays)
    print('  %-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def cfg_arrays(c):
    return dict(dt=np.float64(c['dt']), start=np.float64(c['start']), end=np.float64(c['end']),
                init=np.asarray(c['init'], float), target=np.asarray(c['target'], float), guess=c['guess'],
                Qk=c['Qk'], Rk=c['Rk'], QT=c['QT'], params=np.asarray(c['params'], float),
                n_iterations=np.int32(c['n_iterations']))


def gen_dynamics(name, fab):
    mdl = models.BUILDERS[name]()
    n, m = mdl.n, mdl.m
    c = CONFIG[name]()
    rng = np.random.default_rng(100 + models.SYSTEM_IDS[name])
    cnt = 256
    x = rng.uniform(-3.0, 3.0, (cnt, n))
    if name in ('quadcopter', 'floating_base'):
        x[:, 4] = rng.uniform(-1.2, 1.2, cnt)      # pitch away from the tan/sec singularity
    u = rng.uniform(-2.0, 2.0, (cnt, m))
    x[0] = 0.0; u[0] = 0.0
    p = np.asarray(c['params'], float)
    f, A, B = fab
    fo = np.stack([np.asarray(f(x[i], u[i], c['dt'], p), float).reshape(n) for i in range(cnt)])
    Ao = np.stack([np.asarray(A(x[i], u[i], c['dt'], p), float).reshape(n, n) for i in range(cnt)])
    Bo = np.stack([np.asarray(B(x[i], u[i], c['dt'], p), float).reshape(n, m) for i in range(cnt)])
    save('dyn_' + name, x=x, u=u, dt=np.float64(c['dt']), params=p, f=fo, A=Ao, B=Bo)


def gen_mat_kats():
    mdir = '/root/reference/matlab/smooth-ilqr'
    for name, fn in (('pendulum', 'swingup-trajectory.mat'), ('acrobot', 'acrobot-swingup-trajectory.mat'),
                     ('cartpole', 'cartpole-swingup-trajectory.mat'), ('quadcopter', 'quadcopter-trajectory.mat'),
                     ('car', 'car-accelerate-trajectory.mat')):
        d = scipy.io.loadmat(os.path.join(mdir, fn))
        states = np.asarray(d['states'], float)
        inputs = np.asarray(d['inputs'], float)
        if inputs.ndim == 1:
            inputs = inputs[:, None]
        n = models.BUILDERS[name]().n
        if states.shape[1] != n:
            states = states.T
        if inputs.shape[0] < inputs.shape[1]:
            inputs = inputs.T
        T = min(states.shape[0] - 1, inputs.shape[0])
        idx = np.unique(np.linspace(0, T - 1, 64).astype(int))
        save('mat_kat_' + name, x=states[idx], u=inputs[idx], x_next=states[idx + 1], dt=np.float64(d['dt'].item()),
             params=np.asarray(d['parameters'], float).reshape(-1) if np.size(d['parameters']) else np.zeros(1))


def gen_passes(name, fab, alphas=(1.0, 0.95 ** 3, 0.95 ** 20)):
    c = CONFIG[name]()
    obj = make(c, fab, cls=RefIlqr)
    states0, inputs0 = obj.rollout()
    cost0 = obj.compute_cost(states0, inputs0)
    k, K, dj = obj.backwards_pass()
    out = dict(cfg_arrays(c), states0=states0.copy(), inputs0=inputs0.copy(), cost0=np.float64(cost0), k=k, K=K,
               dJ=np.array([obj.expected_cost_reduction_, obj.expected_cost_reduction_grad_,
                            obj.expected_cost_reduction_hess_], float), alphas=np.array(alphas))
    stride = max(1, states0.shape[0] // 150)
    out['fwd_stride'] = np.int32(stride)
    for i, a in enumerate(alphas):
        s, u = obj.forwards_pass(a)
        out['fwd%d_states' % i] = np.vstack([s[::stride], s[-1:]])
        out['fwd%d_inputs' % i] = np.vstack([u[::stride], u[-1:]])
        out['fwd%d_cost' % i] = np.float64(obj.compute_cost(s, u))
    save(name + '_passes', **out)


def gen_solve(name, fab):
    c = CONFIG[name]()
    t = time.time()
    obj = make(c, fab)
    r = run_solve(obj)
    print('  %s solve: %d iterations, status %d, cost %.16g  (%.1f s)' % (name, r['iters'], r['status'], r['cost'],
                                                                           time.time() - t))
    save(name + '_solve', **cfg_arrays(c), **r)
    return r


def gen_batch(name, fab, B=8):
    c = CONFIG[name]()
    n = len(c['init'])
    m = c['guess'].shape[1]
    short = {'pendulum': 1.0, 'acrobot': 0.4, 'cartpole': 1.5, 'quadcopter': 0.5, 'car': 1.0, 'double_integrator': 1.0,
             'double_pendulum': 0.4, 'floating_base': 0.25}[name]
    rng = np.random.default_rng(7 + models.SYSTEM_IDS[name])
    N = np.arange(c['start'], short, c['dt']).shape[0]
    # (floating base: larger deviations end some of the 8 searches early, and the fixture stacks equal iteration counts)
    inits = np.asarray(c['init'], float) + rng.uniform(-0.2, 0.2, (B, n)) * (0.25 if name == 'floating_base' else 1.0)
    scale = {'pendulum': 0.1, 'acrobot': 0.5, 'cartpole': 0.05, 'quadcopter': 0.02, 'car': 0.2, 'double_integrator': 0.2,
             'double_pendulum': 0.2, 'floating_base': 1.0}[name]
    guesses = c['guess'][:N][None] + scale * rng.standard_normal((B, N, m))
    res = []
    for b in range(B):
        obj = make(c, fab, init=inits[b], guess=guesses[b], end=short, n_iterations=3)
        res.append(run_solve(obj))
    out = dict(cfg_arrays(c))
    out.update(end=np.float64(short), n_iterations=np.int32(3), init=inits, guess=guesses)
    for key in ('states', 'inputs', 'k', 'K', 'cost', 'status', 'iters', 'trace'):
        out[key] = np.stack([r[key] for r in res])
    out['armijo_ratios'] = np.concatenate([r['armijo_ratios'] for r in res])
    save(name + '_batch', **out)


def track(f, states, inputs, K, x0, dt, p):
    """What example_acrobot_swingup_ltvlqr_tracking.py:47-67 / example_pendulum_swingup_ltvlqr_tracking.py:61-81 simulate:
    u = inputs[i] + K[i] @ (x - states[i]), x <- f(x, u); row 0 of the output is the perturbed initial state."""
    xs, us = states.copy(), inputs.copy()
    x = np.array(x0, dtype=float)
    xs[0, :] = x
    for i in range(inputs.shape[0]):
        err = (x - states[i, :]).flatten()
        u = inputs[i, :] + K[i, :, :] @ err
        x = f(x, u, dt, p).flatten()
        xs[i + 1, :] = x
        us[i, :] = u
    return xs, us


def gen_tracking(name, f, states, inputs, K, init, example_perturbation, dt, p):
    rng = np.random.default_rng(7)
    perts = [np.asarray(example_perturbation, float), np.zeros(len(init))]
    perts += [0.3 * example_perturbation[0] * rng.standard_normal(len(init)) for _ in range(4)]
    x0 = np.stack([np.asarray(init, float) + d for d in perts])
    outs = [track(f, states, inputs, K, x, dt, p) for x in x0]
    save(name + '_tracking', init=x0, states=np.stack([o[0] for o in outs]), inputs=np.stack([o[1] for o in outs]))


def gen_ltvlqr(fabs, pend_solve):
    # acrobot: example_acrobot_swingup_ltvlqr_tracking.py:26-45
    states = np.loadtxt(os.path.join(REF, 'acrobot_swingup_states.csv'), dtype=float, delimiter=',')
    inputs = np.loadtxt(os.path.join(REF, 'acrobot_swingup_inputs.csv'), dtype=float, delimiter=',')
    inputs = inputs.reshape((inputs.shape[0], 1))
    save('acrobot_csv', states=states, inputs=inputs)
    Qk, Rk, QT = 0.1 * np.eye(4), 0.01 * np.eye(1), 100 * np.eye(4)
    p = np.array([1, 1, 2, 1, 9.8])
    _, A, B = fabs['acrobot']
    K = ref_ltvlqr(states, inputs, A, B, Qk, Rk, QT, 0.001, p)
    save('acrobot_ltvlqr', Qk=Qk, Rk=Rk, QT=QT, params=p.astype(float), dt=np.float64(0.001), K=K)
    gen_tracking('acrobot', fabs['acrobot'][0], states, inputs, K, [-np.pi / 2, 0, 0, 0], np.array([-np.pi / 8, 0, 0, 0]), 0.001, p)
    # pendulum: example_pendulum_swingup_ltvlqr_tracking.py:53-59 on the solve() output
    Qk, Rk, QT = 0.1 * np.eye(2), 0.01 * np.eye(1), 100 * np.eye(2)
    p = np.array([1, 9.8, 1])
    _, A, B = fabs['pendulum']
    K = ref_ltvlqr(pend_solve['states'], pend_solve['inputs'], A, B, Qk, Rk, QT, 0.005, p)
    save('pendulum_ltvlqr', Qk=Qk, Rk=Rk, QT=QT, params=p.astype(float), dt=np.float64(0.005), K=K,
         states=pend_solve['states'], inputs=pend_solve['inputs'])
    gen_tracking('pendulum', fabs['pendulum'][0], pend_solve['states'], pend_solve['inputs'], K, [0, 0],
                 np.array([-np.pi / 4, 0]), 0.005, p)


SAMPLES = {'c2': ('acrobot', 32, 32), 'c4': ('cartpole', 16, 16), 'c5': ('quadcopter', 8, 8)}   # system, problems, time stride
_SAMPLE_FAB = {}


def _sample_worker(job):
    """one problem of a bench batch through the reference class (a worker process; the lambdified model is built once)"""
    system, idx, c = job
    if system not in _SAMPLE_FAB:
        _SAMPLE_FAB[system] = lambdified(system)
    t = time.time()
    r = run_solve(make(c, _SAMPLE_FAB[system]))
    return idx, r, time.time() - t


def gen_samples(which):
    """The random bench batches pinned to the reference itself (bench.workload(name, B, 0): problems 0 .. count-1)."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    import bench
    jobs, meta = [], {}
    for name in which:
        system, count, stride = SAMPLES[name]
        w = bench.workload(name, bench.DEFAULT_BATCH[name], 0)
        N = w['N']
        meta[name] = (w, count, stride)
        for b in range(count):
            target = w['target'][b] if np.ndim(w['target']) == 2 else w['target']
            c = dict(dt=w['dt'], start=0, end=N * w['dt'] - w['dt'] / 2, init=w['init'][b].copy(), target=np.array(target, float),
                     guess=w['guess'][b].copy(), Qk=w['Qk'], Rk=w['Rk'], QT=w['QT'], params=w['params'], n_iterations=w['n_iterations'])
            assert np.arange(c['start'], c['end'], c['dt']).shape[0] == N
            jobs.append((system, (name, b), c))
    res = {}
    with mp.Pool(min(7, len(jobs))) as pool:
        for (name, b), r, secs in pool.imap_unordered(_sample_worker, jobs):
            res[(name, b)] = r
            print('  %s problem %2d: %d iterations, status %d, cost %.16g (%.1f s)' % (name, b, r['iters'], r['status'], r['cost'], secs),
                  flush=True)
    for name in which:
        w, count, stride = meta[name]
        rs = [res[(name, b)] for b in range(count)]
        n_it = w['n_iterations']
        trace = np.full((count, n_it, 6), np.nan)
        for b, r in enumerate(rs):
            trace[b, :r['trace'].shape[0]] = r['trace']
        N = w['N']
        ts = np.unique(np.concatenate([np.arange(0, N, stride), [N - 1]]))        # rows of inputs / gains kept
        tx = np.unique(np.concatenate([np.arange(0, N + 1, stride), [N]]))        # rows of states kept
        save(name + '_sample', config=np.array(list(name.encode()), np.uint8), count=np.int32(count), n_iterations=np.int32(n_it),
             init=w['init'][:count], guess_checksum=np.array([float(np.sum(w['guess'][b] * np.arange(1, N + 1)[:, None])) for b in range(count)]),
             rows_inputs=ts, rows_states=tx, cost=np.array([r['cost'] for r in rs]), status=np.array([r['status'] for r in rs], np.int32),
             iters=np.array([r['iters'] for r in rs], np.int32), trace=trace,
             states=np.stack([r['states'][tx] for r in rs]), inputs=np.stack([r['inputs'][ts] for r in rs]),
             k=np.stack([r['k'][ts] for r in rs]), K=np.stack([r['K'][ts] for r in rs]))


def main():
    only = set(sys.argv[1:])
    if only and only <= {'samples', 'c2sample', 'c4sample', 'c5sample'}:
        gen_samples(['c2', 'c4', 'c5'] if 'samples' in only else sorted(o[:2] for o in only))
        return
    if only == {'solve10'}:           # the acrobot example with n_iterations = 10 (BASELINE configs[1] runs 10 iterations)
        c = dict(CONFIG['acrobot'](), n_iterations=10)
        r = run_solve(make(c, lambdified('acrobot')))
        print('  acrobot, 10 iterations: status %d, cost %.16g' % (r['status'], r['cost']))
        save('acrobot_solve10', **cfg_arrays(c), **r)
        return
    fabs = {}
    for name in models.BUILDERS:
        if not only or name in only or (name in ('acrobot', 'pendulum') and (only == {'tracking'} or 'pendulum' in only)):
            fabs[name] = lambdified(name)
    print('writing fixtures to', OUT)
    if only == {'tracking'}:          # only the tracking simulations, from the trajectories already in tests/golden
        pl = np.load(os.path.join(OUT, 'pendulum_ltvlqr.npz'))
        gen_ltvlqr(fabs, dict(states=pl['states'], inputs=pl['inputs']))
        return
    if not only:
        gen_mat_kats()
    solves = {}
    for name in models.BUILDERS:
        if only and name not in only:
            continue
        print(name)
        gen_dynamics(name, fabs[name])
        gen_passes(name, fabs[name])
        gen_batch(name, fabs[name])
        solves[name] = gen_solve(name, fabs[name])
    if not only or 'pendulum' in only:
        gen_ltvlqr(fabs, solves['pendulum'])


if __name__ == '__main__':
    main()
