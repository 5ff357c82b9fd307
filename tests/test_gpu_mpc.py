"""Receding-horizon tracking iLQR (matlab/smooth-ilqr/ilqr_mpc.m) on the GPU.

PARITY UNPINNED: the reference has this solver in MATLAB only and MATLAB cannot run here, so there are no reference outputs.
The checker is oracle/ilqr_oracle.c's line-by-line restatement of the .m file (oracle_mpc_*), itself tied to the pinned
python path in the one case where the two algorithms coincide (tests/test_cpu.py::test_mpc_restatement_reduces_to_ilqr_py);
and on the GPU the tracking kernels must reproduce the pinned ilqr.py kernels BIT FOR BIT in that case.
"""
import numpy as np
import pytest

from conftest import golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope='module')
def amd():
    import ilqr_amd
    ilqr_amd._lib.require_gpu()
    return ilqr_amd


def acrobot_mpc_problem(horizon):
    """example_acrobot_ilqr_mpc.m:5-28 with the python reference's own swing-up (acrobot_swingup_*.csv) as the trajectory to
    track: Q_k = 0.001 I, R_k = 0.01, Q_T = 1000 I, targets padded by `horizon` rows of the fixed point / zero torque"""
    csv, g = golden('acrobot_csv'), golden('acrobot_solve')
    states, inputs = csv['states'].copy(), csv['inputs'].copy()
    states[0] = g['init']                                      # the python reference leaves row 0 at zero; a tracker needs the real one
    states = np.vstack([states, np.tile(g['target'], (horizon, 1))])
    inputs = np.vstack([inputs, np.zeros((horizon, 1))])
    return dict(states=states, inputs=inputs, dt=float(g['dt']), params=g['params'], Qk=0.001 * np.eye(4), Rk=0.01 * np.eye(1),
                QT=1000.0 * np.eye(4), init=g['init'])


def run_oracle_mpc(oracle, system, pr, H, n_it, x0, idxs, f_step):
    """the closed loop of example_acrobot_ilqr_mpc.m:47-63 for a batch, every solve by the restated reference"""
    Bn = x0.shape[0]
    cur = x0.copy()
    sol = None
    outs, prev = [], None
    for idx in idxs:
        i0 = idx - 1
        xd = np.broadcast_to(pr['states'][i0:i0 + H + 1], (Bn, H + 1, pr['states'].shape[1])).copy()
        ud = np.broadcast_to(pr['inputs'][i0:i0 + H], (Bn, H, pr['inputs'].shape[1])).copy()
        if idx == 1:
            st, u = xd.copy(), ud.copy()
            k, K = np.zeros((Bn, H, ud.shape[2])), np.zeros((Bn, H, ud.shape[2], xd.shape[2]))
        else:
            st, u, K = (np.stack(v) for v in zip(*[oracle.mpc_shift(idx - prev, xd[b], ud[b], sol['states'][b], sol['inputs'][b], sol['K'][b])
                                                   for b in range(Bn)]))
            k = sol['k']
        sol = oracle.mpc_solve_batched(system, idx == 1, cur, xd, ud, st, u, k, K, pr['Qk'], pr['Rk'], pr['QT'], pr['dt'], pr['params'], n_it)
        outs.append((cur.copy(), sol))
        for _ in range(idx, idxs[idxs.index(idx) + 1] if idx != idxs[-1] else idx):     # simulate up to the next solve
            pass
        nxt = idxs[idxs.index(idx) + 1] if idx != idxs[-1] else idx
        for s in range(nxt - idx):                                                      # apply the first inputs of the plan (:57-66)
            cur = f_step(cur, sol['inputs'][:, s])
        prev = idx
    return outs


@pytest.mark.parametrize('idxs', [[1, 2, 3, 4], [1, 3, 6]])
def test_mpc_closed_loop_matches_the_restated_reference(amd, oracle, idxs):
    """acrobot, horizon 100, 10 iterations, a batch of perturbed initial states (the example's -pi/8 among them): every
    solve_ilqr call of the closed loop -- first call, warm starts moved by 1, 2 and 3 steps -- against the restatement:
    same accepted trials, exit status and iteration count, states / inputs / gains / cost within 1e-8"""
    H, n_it = 100, 10
    pr = acrobot_mpc_problem(H)
    f, A, B = amd.symbolic_dynamics_acrobot()
    rng = np.random.default_rng(11)
    Bn = 70
    x0 = pr['init'] + np.vstack([[-np.pi / 8, 0, 0, 0], 0.3 * rng.uniform(-1, 1, (Bn - 1, 4)) * [1, 1, 0.5, 0.5]])
    f_step = lambda x, u: f.batch(x, u, pr['dt'], pr['params'])
    ref = run_oracle_mpc(oracle, 'acrobot', pr, H, n_it, x0, idxs, f_step)
    mpc = amd.ilqr_mpc(pr['states'], pr['inputs'], pr['dt'], H, f, A, B, pr['Qk'], pr['Rk'], pr['QT'], pr['params'], n_it)
    for idx, (cur, r) in zip(idxs, ref):
        states, inputs, k, K, cost = mpc.solve_ilqr(idx, cur)              # the same current states as the restatement's loop
        assert np.array_equal(mpc.status_, r['status']) and np.array_equal(mpc.iterations_, r['iters']), idx
        assert np.array_equal(mpc.trace_[:, :, 4], r['trace'][:, :, 4], equal_nan=True), idx
        assert np.array_equal(states[:, 0], cur)                             # row 0 is the current state (ilqr_mpc.m:314)
        assert rel_err(states, r['states']) < TOL and rel_err(inputs, r['inputs']) < TOL, idx
        assert rel_err(K, r['K']) < TOL and np.max(np.abs(cost - r['cost']) / np.abs(r['cost'])) < TOL, idx
        assert np.max(np.abs(k - r['k'])) / max(np.max(np.abs(r['k'])), np.max(np.abs(r['inputs']))) < TOL, idx
    assert (ref[0][1]['iters'] > 1).any() and (ref[-1][1]['trace'][:, 0, 4] >= 0).any()     # the solves did real work
    # the tracker tracks: after the last solve the plan's cost is far below the first call's
    assert np.median(ref[-1][1]['cost']) < np.median(ref[0][1]['trace'][:, 0, 0])
    mpc.close()


def test_mpc_single_problem_and_quadcopter(amd, oracle):
    """unbatched call (the MATLAB signature), and a system with several inputs (quadcopter, n = 12, m = 4: hover targets)"""
    H, n_it = 100, 10
    pr = acrobot_mpc_problem(H)
    f, A, B = amd.symbolic_dynamics_acrobot()
    x0 = pr['init'] + np.array([-np.pi / 8, 0, 0, 0])
    mpc = amd.ilqr_mpc(pr['states'], pr['inputs'], pr['dt'], H, f, A, B, pr['Qk'], pr['Rk'], pr['QT'], pr['params'], n_it)
    states, inputs, k, K, cost = mpc.solve_ilqr(1, x0)
    assert states.shape == (H + 1, 4) and inputs.shape == (H, 1) and K.shape == (H, 1, 4) and np.ndim(cost) == 0
    r = run_oracle_mpc(oracle, 'acrobot', pr, H, n_it, x0[None], [1], None)[0][1]
    assert int(mpc.status_) == r['status'][0] and rel_err(states, r['states'][0]) < TOL and rel_err(K, r['K'][0]) < TOL
    with pytest.raises(ValueError):
        mpc.solve_ilqr(H + 5, x0)                               # further than one horizon after the last call
    with pytest.raises(ValueError):
        amd.ilqr_mpc(pr['states'], pr['inputs'], pr['dt'], H, f, A, B, pr['Qk'], pr['Rk'], pr['QT'], pr['params'], n_it).solve_ilqr(2, x0)
    mpc.close()
    g = golden('quadcopter_solve')
    fq, Aq, Bq = amd.symbolic_dynamics_quadcopter()
    Hq, T = 40, 60
    rng = np.random.default_rng(4)
    tgt_x = np.tile(g['target'], (T + Hq + 1, 1)) * np.linspace(0.0, 1.0, T + Hq + 1)[:, None]      # a ramp to the target position
    tgt_u = np.tile(g['guess'][0], (T + Hq, 1))
    pq = dict(states=tgt_x, inputs=tgt_u, dt=float(g['dt']), params=g['params'], Qk=0.1 * np.eye(12), Rk=g['Rk'], QT=g['QT'])
    x0q = 0.05 * rng.standard_normal((9, 12))
    fq_step = lambda x, u: fq.batch(x, u, pq['dt'], pq['params'])
    ref = run_oracle_mpc(oracle, 'quadcopter', pq, Hq, 5, x0q, [1, 2], fq_step)
    mq = amd.ilqr_mpc(tgt_x, tgt_u, pq['dt'], Hq, fq, Aq, Bq, pq['Qk'], pq['Rk'], pq['QT'], pq['params'], 5)
    for idx, (cur, r) in zip([1, 2], ref):
        states, inputs, k, K, cost = mq.solve_ilqr(idx, cur)
        assert np.array_equal(mq.status_, r['status']) and np.array_equal(mq.trace_[:, :, 4], r['trace'][:, :, 4], equal_nan=True)
        assert rel_err(states, r['states']) < TOL and rel_err(inputs, r['inputs']) < TOL and rel_err(K, r['K']) < TOL
        assert np.max(np.abs(cost - r['cost']) / np.abs(r['cost'])) < TOL
    mq.close()


@pytest.mark.parametrize('system', ['acrobot', 'car'])
def test_tracking_kernels_reduce_to_the_pinned_path_bitwise(amd, system):
    """x_des = target_state in every row, u_des = 0, Q_k = 0, zero gains as warm start: the tracking cost IS ilqr.py's cost
    (ilqr.py:49-62 vs ilqr_mpc.m:193-216) and the closed-loop rollout with learning rate 0 IS the open-loop rollout, so the
    tracking kernels must reproduce the pinned solve() bit for bit -- states, inputs, gains, costs, every trace entry"""
    from ilqr_amd import _lib
    g = golden(system + '_batch')
    Bn, N, n, m = g['guess'].shape[0], g['guess'].shape[1], g['init'].shape[1], g['guess'].shape[2]
    n_it = 3
    common = dict(want_trace=True, n_iterations=n_it, n_alpha_parallel=16, pipeline=1)
    a = amd.Workspace(system, N, Bn, float(g['dt']), **common)
    b = amd.Workspace(system, N, Bn, float(g['dt']), tracking=True, **common)
    Z = np.zeros((n, n))
    for ws in (a, b):
        ws.set_weights(Z, g['Rk'], g['QT'])
        ws.upload(_lib.F_PARAMS, g['params'])
        ws.upload(_lib.F_INIT, g['init'])
    a.upload(_lib.F_TARGET, g['target'])
    a.upload(_lib.F_GUESS, g['guess'])
    a.solve()
    b.upload(_lib.F_TARGET_STATES, np.broadcast_to(g['target'], (Bn, N + 1, n)).copy())
    b.upload(_lib.F_TARGET_INPUTS, np.zeros((Bn, N, m)))
    b.upload(_lib.F_STATES, np.zeros((Bn, N + 1, n)))
    b.upload(_lib.F_INPUTS, g['guess'])
    b.upload(_lib.F_KFB, np.zeros((Bn, N, m, n)))
    b.upload(_lib.F_KFF, np.zeros((Bn, N, m)))
    b.mpc_solve(False)
    a.sync(); b.sync()
    for fld in (_lib.F_STATES, _lib.F_INPUTS, _lib.F_KFB, _lib.F_KFF, _lib.F_COST, _lib.F_STATUS, _lib.F_ITERS, _lib.F_TRACE, _lib.F_DJ):
        assert np.array_equal(a.download(fld), b.download(fld), equal_nan=True), fld
    assert (a.download(_lib.F_TRACE)[:, :, 4] >= 0).any()
    # ... and compute_cost / forwards_pass of the tracking workspace give the same numbers as the plain one
    a.cost(); b.cost(); a.sync(); b.sync()
    assert np.array_equal(a.download(_lib.F_COST), b.download(_lib.F_COST))
    a.close(); b.close()


@pytest.mark.parametrize('system', ['quadcopter', 'acrobot', 'car'])
def test_tracking_backward_pass_and_cost_with_a_dense_Qk(amd, oracle, system):
    """One teacher-forced pass of the tracking kernels with a DENSE, non-zero Q_k and targets away from the trajectory, per
    kernel against the restatement: k_backward<TRACK> (gains k, K and the three expected reductions; l_x = Q_k (x - x_des),
    ilqr_mpc.m:260) and k_cost<TRACK> (stage cost with the full quadratic form in x, :207-209) -- every entry of Q_k in play.
    Conditioning decides what can be asked.  The reference's recursion (explicit inverse, V_xx never symmetrised) is UNSTABLE
    for the quadcopter once Q_k reaches ~0.1: a 1e-13 perturbation of the states grows ~100x per 50 steps and is O(1) after
    the example's 400 (measured on the CPU restatement against itself, see below) -- which is what round 2 took for a
    compiler bug in k_backward<quadcopter, TRACK> ("wrong gains at -O3, right at -O1": two roundings of one unstable
    recursion).  So: the large Q_k over a horizon short enough to stay conditioned (120 steps), a Q_k of the example's size
    (1e-3, example_acrobot_ilqr_mpc.m:13) over the whole horizon, and the instability itself asserted on the checker."""
    from ilqr_amd import _lib
    p = golden(system + '_passes')
    N, n, m = p['inputs0'].shape[0], p['states0'].shape[1], p['inputs0'].shape[1]
    rng = np.random.default_rng(11)
    M = rng.standard_normal((n, n))
    big = 0.01 * (M @ M.T) + 0.002 * np.eye(n)                      # dense, symmetric positive definite, entries up to ~0.15
    Bn = 3
    xs = np.stack([p['states0'] + 0.01 * b * rng.standard_normal(p['states0'].shape) for b in range(Bn)])
    us = np.stack([p['inputs0'] + 0.01 * b * rng.standard_normal(p['inputs0'].shape) for b in range(Bn)])
    xd = xs + 0.05 * rng.standard_normal(xs.shape)
    ud = us + 0.05 * rng.standard_normal(us.shape)
    for Qk, Nuse in ((big, min(N, 120)), (0.01 * big, N)):
        ws = amd.Workspace(system, Nuse, Bn, float(p['dt']), want_trace=False, n_iterations=1, tracking=True, zero_first_state=False,
                           pipeline=1, low_expected_reduction=0.0)
        ws.set_weights(Qk, p['Rk'], p['QT'])
        ws.upload(_lib.F_PARAMS, p['params'])
        sl = (xs[:, :Nuse + 1], us[:, :Nuse], xd[:, :Nuse + 1], ud[:, :Nuse])
        for f, a in zip((_lib.F_STATES, _lib.F_INPUTS, _lib.F_TARGET_STATES, _lib.F_TARGET_INPUTS), sl):
            ws.upload(f, np.ascontiguousarray(a))
        ws.cost()
        cost = ws.download(_lib.F_COST)
        ws.backward(0)
        k, K, dJ = ws.download(_lib.F_KFF), ws.download(_lib.F_KFB), ws.download(_lib.F_DJ)
        ws.close()
        for b in range(Bn):
            c_ref = oracle.mpc_cost(sl[0][b], sl[1][b], sl[2][b], sl[3][b], Qk, p['Rk'], p['QT'])
            assert abs(cost[b] - c_ref) <= 1e-12 * abs(c_ref), (b, cost[b], c_ref)
            k_ref, K_ref, dJ_ref = oracle.mpc_backward(system, sl[0][b], sl[1][b], sl[2][b], sl[3][b], Qk, p['Rk'], p['QT'], float(p['dt']), p['params'])
            tol = 1e-7 if system == 'quadcopter' else 1e-8   # (the restatement SOLVES with Q_uu, the kernels multiply by its inverse: 3e-9 on the quadcopter's K)
            errs = (rel_err(K[b], K_ref), rel_err(k[b], k_ref), rel_err(dJ[b], dJ_ref))
            assert max(errs) < tol, (system, Nuse, b, errs)
    if system == 'quadcopter':
        # the instability, on the checker alone: the same pass twice, the states 1e-13 apart
        args = (big, p['Rk'], p['QT'], float(p['dt']), p['params'])
        _, K1, _ = oracle.mpc_backward(system, xs[0], us[0], xd[0], ud[0], *args)
        _, K2, _ = oracle.mpc_backward(system, xs[0] + 1e-13 * rng.standard_normal(xs[0].shape), us[0], xd[0], ud[0], *args)
        growth = np.max(np.abs(K2 - K1), axis=(1, 2)) / np.max(np.abs(K1))
        assert growth[-1] < 1e-12 and growth[0] > 1e-3, (growth[-1], growth[0])       # last step: rounding; first step: O(1)
