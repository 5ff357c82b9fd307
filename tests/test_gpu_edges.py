"""Edge cases of the hot path on the GPU (through the python drop-ins, i.e. the C ABI): degenerate sizes, the slow trig
path, non-finite problems inside a batch, singular Q_uu, an exhausted line search.  Expected values come from the CPU
oracle (itself pinned to the reference by tests/test_cpu.py) or from what the reference does in the same situation."""
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


def test_horizon_of_one_step_and_batch_of_one(amd, oracle):
    """N = 1 (np.arange(0, dt, dt) has one element) and B = 1: the smallest legal problem, unbatched and as a (1, ...) batch"""
    g = golden('pendulum_solve')
    f, A, B = amd.symbolic_dynamics_pendulum()
    dt = float(g['dt'])
    guess = np.array([[0.3]])
    init = np.array([0.2, -0.1])
    s = amd.ilqr(init, g['target'], guess, dt, 0.0, dt, f, A, B, g['Qk'], g['Rk'], g['QT'], g['params'], 4)
    assert s.n_timesteps_ == 1
    states, inputs, k, K, cost = s.solve()
    r = oracle.solve('pendulum', init, g['target'], guess, g['Qk'], g['Rk'], g['QT'], dt, g['params'], 4)
    assert states.shape == (2, 2) and inputs.shape == (1, 1) and K.shape == (1, 1, 2)
    assert int(s.status_) == r['status'] and int(s.iterations_) == r['iters']
    assert rel_err(states, r['states']) < TOL and rel_err(inputs, r['inputs']) < TOL and abs(cost - r['cost']) <= TOL * abs(r['cost'])
    sb = amd.ilqr(init[None], g['target'], guess[None], dt, 0.0, dt, f, A, B, g['Qk'], g['Rk'], g['QT'], g['params'], 4)
    out = sb.solve()
    assert out[0].shape == (1, 2, 2) and np.array_equal(out[0][0], states) and np.array_equal(out[3][0], K)


@pytest.mark.parametrize('system', ['pendulum', 'acrobot', 'cartpole', 'quadcopter'])
def test_huge_angles_take_the_library_trig_path(amd, oracle, system):
    """|angle| >= 1e6 leaves the two-round Cody-Waite reduction for ocml's sin/cos (one branch per generated function);
    points with small and huge angles are mixed so both sides of the branch run in one wave"""
    d = golden('dyn_' + system)
    x, u = d['x'][:64].copy(), d['u'][:64].copy()
    rng = np.random.default_rng(9)
    big = rng.uniform(1e6, 1e9, x.shape) * rng.choice([-1.0, 1.0], x.shape)
    mask = rng.random(x.shape) < 0.3
    if system == 'quadcopter':
        mask[:, 4] = False                                  # keep the pitch away from the tan singularity
        mask[:, :3] = False
    x[mask] = big[mask]
    for which, fn in zip('fAB', [getattr(amd, 'symbolic_dynamics_' + system)()[i] for i in range(3)]):
        out = fn.batch(x, u, float(d['dt']), d['params'])
        ref = oracle.dynamics(system, which, x, u, float(d['dt']), d['params'])
        assert np.all(np.isfinite(out))
        # a huge angle is known to +-ulp(angle) only; the comparison is between two correctly reduced evaluations
        err = np.abs(out - ref) / np.maximum(np.abs(ref), 1.0)
        assert err.max() < 1e-9, (system, which, err.max())


def test_non_finite_problem_does_not_disturb_its_tile(amd, oracle):
    """a NaN initial state in the middle of a tile: that problem ends the way the reference would (every Armijo test is
    False -> 'low learning rate' after one iteration), its 69 neighbours are bit-identical to the batch without it"""
    g = golden('acrobot_batch')
    rng = np.random.default_rng(21)
    Bn, N = 70, 120
    init = g['init'][0] + rng.uniform(-0.2, 0.2, (Bn, 4))
    guess = np.resize(g['guess'][0], (N, 1))[None] + 0.05 * rng.standard_normal((Bn, N, 1))
    f, A, B = amd.symbolic_dynamics_acrobot()
    dt = float(g['dt'])
    args = (dt, 0.0, N * dt - dt / 2, f, A, B, g['Qk'], g['Rk'], g['QT'], g['params'], 3)
    clean = amd.ilqr(init, g['target'], guess, *args)
    c_out = clean.solve()
    bad_init = init.copy()
    bad_init[33, 1] = np.nan
    bad = amd.ilqr(bad_init, g['target'], guess, *args)
    b_out = bad.solve()
    keep = np.arange(Bn) != 33
    for a, b in zip(c_out, b_out):
        assert np.array_equal(a[keep], b[keep])
    assert np.array_equal(clean.status_[keep], bad.status_[keep])
    r = oracle.solve('acrobot', bad_init[33], g['target'], guess[33], g['Qk'], g['Rk'], g['QT'], dt, g['params'], 3)
    assert int(bad.status_[33]) == r['status'] == 2 and int(bad.iterations_[33]) == r['iters'] == 1
    assert np.isnan(b_out[4][33])


def test_singular_quu_is_a_status_not_a_crash(amd):
    """R_k = Q_T = Q_k = 0 makes Q_uu exactly zero at every step: numpy raises LinAlgError('Singular matrix') at
    ilqr.py:106; the single-problem drop-in raises the same, a batch reports status 3 for those problems only"""
    g = golden('pendulum_solve')
    f, A, B = amd.symbolic_dynamics_pendulum()
    dt, N = float(g['dt']), 50
    guess = 0.1 * np.ones((N, 1))
    Z2, Z1 = np.zeros((2, 2)), np.zeros((1, 1))
    s = amd.ilqr(g['init'], g['target'], guess, dt, 0.0, N * dt - dt / 2, f, A, B, Z2, Z1, Z2, g['params'], 3)
    s.rollout()
    with pytest.raises(np.linalg.LinAlgError):
        s.backwards_pass()
    with pytest.raises(np.linalg.LinAlgError):
        s.solve()
    sb = amd.ilqr(np.tile(g['init'], (3, 1)), g['target'], np.tile(guess, (3, 1, 1)), dt, 0.0, N * dt - dt / 2, f, A, B, Z2, Z1, Z2,
                  g['params'], 3)
    sb.solve()
    assert np.all(sb.status_ == 3) and np.all(sb.iterations_ == 1)


def test_line_search_can_run_out(amd, oracle):
    """all 59 trials rejected (the double integrator's second iteration: the first step already solved the LQ problem)
    -> 'low learning rate' exit, status 2, the trajectory of the last accepted step is returned (ilqr.py:204-211)"""
    g = golden('double_integrator_solve')
    assert int(g['status']) == 2
    f, A, B = amd.symbolic_dynamics_double_integrator()
    for W in (1, 8, 32, 64):                                  # 59, 8, 2 and 1 rounds of the compacted search
        s = amd.ilqr(g['init'], g['target'], g['guess'], float(g['dt']), float(g['start']), float(g['end']), f, A, B, g['Qk'],
                     g['Rk'], g['QT'], g['params'], int(g['n_iterations']), n_alpha_parallel=W)
        states, inputs, k, K, cost = s.solve()
        assert int(s.status_) == 2 and int(s.iterations_) == int(g['iters'])
        assert np.array_equal(s.trace_[:int(g['iters']), 4], g['trace'][:, 4])
        assert rel_err(states, g['states']) < TOL and rel_err(K, g['K']) < TOL and abs(cost - float(g['cost'])) <= TOL * abs(float(g['cost']))


def test_car_accepts_an_empty_parameter_vector(amd):
    """the MATLAB car model has `parameters = []` (example_car_accelerate.m:31): the drop-in takes [] as well"""
    g = golden('car_solve')
    f, A, B = amd.symbolic_dynamics_car()
    s = amd.ilqr(g['init'], g['target'], g['guess'], float(g['dt']), float(g['start']), float(g['end']), f, A, B, g['Qk'], g['Rk'],
                 g['QT'], [], int(g['n_iterations']))
    states, inputs, k, K, cost = s.solve()
    assert rel_err(states, g['states']) < TOL and abs(cost - float(g['cost'])) <= TOL * abs(float(g['cost']))
    assert f(np.zeros(4), np.zeros(2), 0.01, []).shape == (4, 1)


@pytest.mark.parametrize('system', ['car', 'double_pendulum', 'quadcopter', 'floating_base'])
def test_nan_problem_with_several_inputs_ends_like_the_reference(amd, oracle, system):
    """m > 1: Q_uu of a diverged (NaN) trajectory is not positive definite, so the Cholesky fast path hands over to the
    LU inverse with np.linalg.inv's rule (error on an exactly zero pivot only; 12 inputs: the factorisation across the lane
    group carries the NaN through).  The problem then ends the way it does in
    the reference and for m = 1: every Armijo test is False -> 'low learning rate' after one iteration, not 'numerical'."""
    g = golden(system + '_batch')
    init, guess = g['init'].copy(), g['guess']
    init[2, 1] = np.nan
    f, A, B = getattr(amd, 'symbolic_dynamics_' + system)()
    dt = float(g['dt'])
    s = amd.ilqr(init, g['target'], guess, dt, float(g['start']), float(g['end']), f, A, B, g['Qk'], g['Rk'], g['QT'], g['params'], 3)
    out = s.solve()
    r = oracle.solve_batched(system, init, g['target'], guess, g['Qk'], g['Rk'], g['QT'], dt, g['params'], 3)
    assert int(s.status_[2]) == int(r['status'][2]) == 2 and int(s.iterations_[2]) == int(r['iters'][2]) == 1
    keep = np.arange(init.shape[0]) != 2
    assert np.array_equal(s.status_[keep], r['status'][keep]) and np.array_equal(s.iterations_[keep], r['iters'][keep])
    assert rel_err(out[0][keep], r['states'][keep]) < TOL and rel_err(out[3][keep], r['K'][keep]) < TOL


@pytest.mark.parametrize('system', ['car', 'double_pendulum', 'floating_base'])
def test_indefinite_quu_is_inverted_like_numpy(amd, oracle, system):
    """an indefinite R_k makes Q_uu indefinite: np.linalg.inv (LU, partial pivoting) inverts it without complaint
    (ilqr.py:106), and so must the backward pass -- same gains as the oracle's LU inverse, status 0, no exception
    (12 inputs: L D L' across the lane group, D with a negative entry)"""
    g = golden(system + '_passes')
    f, A, B = getattr(amd, 'symbolic_dynamics_' + system)()
    dt = float(g['dt'])
    Bl = oracle.dynamics(system, 'B', g['states0'][-2:-1], g['inputs0'][-1:], dt, g['params'])[0]
    BQB = Bl.T @ g['QT'] @ Bl
    m = Bl.shape[1]
    Rk = 0.5 * np.eye(m)
    Rk[:2, :2] = [[-3.0, 0.4], [0.4, 0.5]]
    Rk *= max(float(np.abs(g['Rk']).max()), 10.0 * float(np.abs(BQB).max()))
    ev = np.linalg.eigvalsh(Rk + BQB)                                      # Q_uu of the first step of the recursion
    assert ev[0] < 0 < ev[-1]
    s = amd.ilqr(g['init'], g['target'], g['guess'], dt, float(g['start']), float(g['end']), f, A, B, g['Qk'], Rk, g['QT'], g['params'], 1)
    s.states_, s.inputs_ = g['states0'], g['inputs0']
    k, K, ecr = s.backwards_pass()
    ko, Ko, dJ = oracle.backward(system, g['states0'], g['inputs0'], g['target'], g['Qk'], Rk, g['QT'], dt, g['params'])
    assert np.isfinite(Ko).all()
    assert rel_err(K, Ko) < TOL and rel_err(k, ko) < TOL and abs(ecr - dJ[0]) <= TOL * abs(dJ[0])


def test_low_learning_rate_and_the_search_floor_are_two_things(amd, oracle):
    """ilqr.py:185 stops the search at the LITERAL 0.05, ilqr.py:204 ends the solve when learning_rate < low_learning_rate
    (:167).  With low_learning_rate above the literal a trial accepted at a small alpha is taken and ends the solve; with
    it below, an exhausted search does not end it.  The oracle keeps the two apart the same way."""
    from ilqr_amd import _lib
    g = golden('acrobot_batch')
    rng = np.random.default_rng(5)
    Bn, N, n_it = 64, 600, 8
    init = g['init'][0] + rng.uniform(-0.2, 0.2, (Bn, 4))
    guess = 0.5 * rng.standard_normal((Bn, N, 1))
    dt = float(g['dt'])
    seen = set()
    for low_lr in (0.9, 0.01):
        opts = (0.95, low_lr, 1e-3, 0.1)
        r = oracle.solve_batched('acrobot', init, g['target'], guess, g['Qk'], g['Rk'], g['QT'], dt, g['params'], n_it, opts=opts)
        for pipeline in (1, 2):
            ws = amd.Workspace('acrobot', N, Bn, dt, want_trace=True, n_iterations=n_it, low_learning_rate=low_lr, pipeline=pipeline)
            ws.set_weights(g['Qk'], g['Rk'], g['QT'])
            for fld, a in ((_lib.F_PARAMS, g['params']), (_lib.F_INIT, init), (_lib.F_TARGET, g['target']), (_lib.F_GUESS, guess)):
                ws.upload(fld, a)
            ws.solve(); ws.sync()
            status, iters, trace = ws.download(_lib.F_STATUS), ws.download(_lib.F_ITERS), ws.download(_lib.F_TRACE)
            assert np.array_equal(status, r['status']) and np.array_equal(iters, r['iters'])
            assert np.array_equal(trace[:, :, 4], r['trace'][:, :, 4], equal_nan=True)
            assert rel_err(ws.download(_lib.F_STATES), r['states']) < TOL and rel_err(ws.download(_lib.F_COST), r['cost']) < TOL
            ws.close()
        last = np.maximum(r['iters'] - 1, 0)
        accj = r['trace'][np.arange(Bn), last, 4]
        if low_lr > 0.05:        # accepted below low_learning_rate -> the step is taken and the solve ends (status 2)
            hit = (r['status'] == 2) & (accj >= 0)
            assert hit.any() and np.all(0.95 ** accj[hit] < low_lr)
            seen.add('accepted-and-ended')
        else:                    # exhausted, but 0.95^59 = 0.0485 is not < 0.01 -> the loop goes on to the next iteration
            assert not np.any(r['status'] == 2)
            if np.any((r['trace'][:, :, 4] == -1) & np.isfinite(r['trace'][:, :, 1]) & (np.arange(n_it)[None] < r['iters'][:, None] - 1)):
                seen.add('exhausted-and-continued')
    assert seen == {'accepted-and-ended', 'exhausted-and-continued'}


def test_ltvlqr_singular_gain_matrix(amd):
    """R_k = Q_T = 0: R + B'PB is exactly zero at the first step of the recursion; np.linalg.inv raises LinAlgError at
    ltvlqr.py:24 and so does the drop-in for a single problem; a batch can ask for the per-problem status instead"""
    g = golden('pendulum_ltvlqr')
    _, A, B = amd.symbolic_dynamics_pendulum()
    Z2, Z1 = np.zeros((2, 2)), np.zeros((1, 1))
    with pytest.raises(np.linalg.LinAlgError):
        amd.ltvlqr(g['states'], g['inputs'], A, B, g['Qk'], Z1, Z2, float(g['dt']), g['params'])
    K, status = amd.ltvlqr(np.tile(g['states'], (3, 1, 1)), np.tile(g['inputs'], (3, 1, 1)), A, B, g['Qk'], Z1, Z2, float(g['dt']),
                           g['params'], return_status=True)
    assert np.all(status == 3)
    K, status = amd.ltvlqr(g['states'], g['inputs'], A, B, g['Qk'], g['Rk'], g['QT'], float(g['dt']), g['params'], return_status=True)
    assert status == 0 and rel_err(K, g['K']) < TOL


@pytest.mark.parametrize('system,Bn', [('acrobot', 150), ('quadcopter', 37)])
def test_shards_equal_the_single_batch_bitwise(amd, system, Bn):
    """ilqr_desc.n_devices: the one-shot entry points cut the batch into contiguous shards by problem index, one host
    thread and one device each.  On this one-GPU box the shards share the device (device_wrap); every output of every
    entry point must equal the unsharded call bit for bit, for shard counts that do and do not divide the batch."""
    from ilqr_amd import _lib
    import ctypes as C
    g = golden(system + '_batch')
    rng = np.random.default_rng(31)
    n, m = g['init'].shape[1], g['guess'].shape[2]
    N = 90
    init = g['init'][0] + rng.uniform(-0.1, 0.1, (Bn, n))
    guess = np.resize(g['guess'][0], (N, m))[None] + 0.02 * rng.standard_normal((Bn, N, m))
    target = np.tile(g['target'], (Bn, 1)) + rng.uniform(-0.05, 0.05, (Bn, n))
    params = np.tile(g['params'], (Bn, 1)) * (1.0 + rng.uniform(-0.01, 0.01, (Bn, g['params'].size)))
    Qk, Rk, QT = (_lib.f64(g[k_]) for k_ in ('Qk', 'Rk', 'QT'))
    n_it, dt = 3, float(g['dt'])

    def run(G):
        d = _lib.make_desc(system, N, Bn, dt, n_iterations=n_it, target_batched=True, params_batched=True, n_devices=G, device_wrap=True)
        o = dict(states=np.empty((Bn, N + 1, n)), inputs=np.empty((Bn, N, m)), k=np.empty((Bn, N, m)), K=np.empty((Bn, N, m, n)),
                 cost=np.empty(Bn), iters=np.empty(Bn, np.int32), status=np.empty(Bn, np.int32), trace=np.empty((Bn, n_it, 6)))
        p = _lib.dptr
        _lib.check(_lib.lib().ilqr_solve_batched(C.byref(d), p(init), p(target), p(guess), p(Qk), p(Rk), p(QT), p(params), p(o['states']),
                                                 p(o['inputs']), p(o['k']), p(o['K']), p(o['cost']), _lib.iptr(o['iters']),
                                                 _lib.iptr(o['status']), p(o['trace'])))
        o['K_ltv'] = np.empty((Bn, N, m, n)); o['ltv_status'] = np.empty(Bn, np.int32)
        _lib.check(_lib.lib().ltvlqr_batched(C.byref(d), N + 1, p(o['states']), p(o['inputs']), p(Qk), p(Rk), p(QT), p(params),
                                             p(o['K_ltv']), _lib.iptr(o['ltv_status'])))
        o['k2'] = np.empty((Bn, N, m)); o['K2'] = np.empty((Bn, N, m, n)); o['dJ'] = np.empty((Bn, 3)); o['bst'] = np.empty(Bn, np.int32)
        _lib.check(_lib.lib().ilqr_backward_batched(C.byref(d), p(o['states']), p(o['inputs']), p(target), p(Qk), p(Rk), p(QT), p(params),
                                                    p(o['k2']), p(o['K2']), p(o['dJ']), _lib.iptr(o['bst'])))
        lr = np.linspace(1.0, 0.3, Bn)
        o['fs'] = np.empty((Bn, N + 1, n)); o['fu'] = np.empty((Bn, N, m)); o['fc'] = np.empty(Bn)
        _lib.check(_lib.lib().ilqr_forward_batched(C.byref(d), p(init), p(o['states']), p(o['inputs']), p(o['k2']), p(o['K2']), p(lr),
                                                   p(params), p(o['fs']), p(o['fu']), p(Rk), p(QT), p(target), p(o['fc'])))
        o['rs'] = np.empty((Bn, N + 1, n)); o['ru'] = np.empty((Bn, N, m)); o['rc'] = np.empty(Bn)
        _lib.check(_lib.lib().ilqr_rollout_batched(C.byref(d), p(init), p(guess), p(params), p(o['rs']), p(o['ru'])))
        _lib.check(_lib.lib().ilqr_cost_batched(C.byref(d), p(o['rs']), p(o['ru']), p(Rk), p(QT), p(target), p(o['rc'])))
        return o

    one = run(1)
    assert np.isfinite(one['cost']).all() and len(np.unique(one['cost'])) == Bn
    for G in (2, 3, 8):
        many = run(G)
        for key in one:
            assert np.array_equal(one[key], many[key], equal_nan=True), (G, key)
    # more shards than visible devices without device_wrap is an error, not an oversubscription
    d = _lib.make_desc(system, N, Bn, dt, n_devices=_lib.lib().ilqr_device_count() + 1)
    rs, ru = np.empty((Bn, N + 1, n)), np.empty((Bn, N, m))
    rc = _lib.lib().ilqr_rollout_batched(C.byref(d), _lib.dptr(init), _lib.dptr(guess), _lib.dptr(params), _lib.dptr(rs), _lib.dptr(ru))
    assert rc < 0 and b'visible' in _lib.lib().ilqr_last_error()
