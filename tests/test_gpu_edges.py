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
