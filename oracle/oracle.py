"""ctypes binding of oracle/libilqr_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module
(see the header of oracle/ilqr_oracle.c).  All arrays use the reference's numpy layout.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libilqr_oracle.so')
SYSTEM_IDS = {'pendulum': 0, 'acrobot': 1, 'cartpole': 2, 'quadcopter': 3, 'car': 4, 'double_integrator': 5,
              'double_pendulum': 6, 'floating_base': 7}
DEFAULT_OPTS = (0.95, 0.05, 1e-3, 0.1)   # ilqr.py:166-169

_SO_FMA = os.path.join(_HERE, 'libilqr_oracle_fma.so')
_lib = None
_lib_fma = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_cost.restype = C.c_double
    return _lib


def lib_fma():
    """the oracle rebuilt WITH fused multiply-adds (oracle/Makefile): last-bit rounding differs, nothing else -- used only
    to measure how far a problem amplifies rounding (the noise floor of a GPU-vs-oracle comparison)"""
    global _lib_fma
    if _lib_fma is None:
        if not os.path.exists(_SO_FMA):
            build()
        _lib_fma = C.CDLL(_SO_FMA)
    return _lib_fma


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def dims(system):
    n, m, npar = C.c_int(), C.c_int(), C.c_int()
    assert lib().oracle_system_dims(SYSTEM_IDS[system], C.byref(n), C.byref(m), C.byref(npar)) == 0
    return n.value, m.value, npar.value


def dynamics(system, which, x, u, dt, p):
    """which in 'f','A','B'; x (count,n), u (count,m) -> (count,n)/(count,n,n)/(count,n,m)."""
    n, m, _ = dims(system)
    x, u, p = _d(x).reshape(-1, n), _d(u).reshape(-1, m), _d(p)
    cnt = x.shape[0]
    shape = {'f': (cnt, n), 'A': (cnt, n, n), 'B': (cnt, n, m)}[which]
    out = np.empty(shape)
    rc = lib().oracle_dynamics(SYSTEM_IDS[system], 'fAB'.index(which), C.c_int64(cnt), _p(x), _p(u), C.c_double(dt),
                               _p(p), _p(out))
    assert rc == 0
    return out


def rollout(system, init_state, inputs, dt, p, zero_first=True):
    n, m, _ = dims(system)
    inputs = _d(inputs)
    N = inputs.shape[0]
    states = np.empty((N + 1, n))
    out_u = np.empty((N, m))
    lib().oracle_rollout(SYSTEM_IDS[system], N, _p(_d(init_state)), _p(inputs), C.c_double(dt), _p(_d(p)),
                         int(zero_first), _p(states), _p(out_u))
    return states, out_u


def cost(states, inputs, Rk, QT, target):
    states, inputs = _d(states), _d(inputs)
    N, m = inputs.shape
    n = states.shape[1]
    return lib().oracle_cost(n, m, N, _p(states), _p(inputs), _p(_d(Rk)), _p(_d(QT)), _p(_d(target)))


def backward(system, states, inputs, target, Qk, Rk, QT, dt, p):
    n, m, _ = dims(system)
    states, inputs = _d(states), _d(inputs)
    N = inputs.shape[0]
    k = np.empty((N, m))
    K = np.empty((N, m, n))
    dJ = np.empty(3)
    rc = lib().oracle_backward(SYSTEM_IDS[system], N, _p(states), _p(inputs), _p(_d(target)), _p(_d(Qk)), _p(_d(Rk)),
                               _p(_d(QT)), C.c_double(dt), _p(_d(p)), _p(k), _p(K), _p(dJ))
    if rc != 0:
        raise np.linalg.LinAlgError('Singular matrix')
    return k, K, dJ


def forward(system, init_state, ref_states, ref_inputs, k, K, learning_rate, dt, p, zero_first=True):
    n, m, _ = dims(system)
    ref_inputs = _d(ref_inputs)
    N = ref_inputs.shape[0]
    states = np.empty((N + 1, n))
    inputs = np.empty((N, m))
    lib().oracle_forward(SYSTEM_IDS[system], N, _p(_d(init_state)), _p(_d(ref_states)), _p(ref_inputs), _p(_d(k)),
                         _p(_d(K)), C.c_double(learning_rate), C.c_double(dt), _p(_d(p)), int(zero_first),
                         _p(states), _p(inputs))
    return states, inputs


def solve_batched(system, init_state, target, guess, Qk, Rk, QT, dt, params, n_iterations, opts=DEFAULT_OPTS,
                  zero_first=True, n_threads=0, want_trace=True, fma=False):
    """init_state (B,n), guess (B,N,m), target (n,) or (B,n), params (P,) or (B,P).  fma: the FMA-contracted build."""
    n, m, npar = dims(system)
    init_state, guess = _d(init_state).reshape(-1, n), _d(guess)
    B, N = guess.shape[0], guess.shape[1]
    target, params = _d(target), _d(params)
    ts = n if target.ndim == 2 else 0
    ps = npar if params.ndim == 2 else 0
    states = np.empty((B, N + 1, n)); inputs = np.empty((B, N, m))
    k = np.zeros((B, N, m)); K = np.zeros((B, N, m, n))
    cost_ = np.empty(B); iters = np.empty(B, np.int32); status = np.empty(B, np.int32)
    trace = np.empty((B, n_iterations, 6)) if want_trace else None
    o = _d(opts)
    rc = (lib_fma() if fma else lib()).oracle_solve_batched(SYSTEM_IDS[system], C.c_int64(B), N, _p(init_state), _p(target), C.c_int64(ts),
                                    _p(guess), _p(_d(Qk)), _p(_d(Rk)), _p(_d(QT)), C.c_double(dt), _p(params),
                                    C.c_int64(ps), int(n_iterations), _p(o), int(zero_first), _p(states), _p(inputs),
                                    _p(k), _p(K), _p(cost_), iters.ctypes.data_as(_ip), status.ctypes.data_as(_ip),
                                    _p(trace) if want_trace else None, int(n_threads))
    assert rc == 0
    return dict(states=states, inputs=inputs, k=k, K=K, cost=cost_, iters=iters, status=status, trace=trace)


def solve(system, init_state, target, guess, Qk, Rk, QT, dt, params, n_iterations, **kw):
    r = solve_batched(system, np.asarray(init_state, float)[None], target, np.asarray(guess, float)[None], Qk, Rk, QT,
                      dt, params, n_iterations, **kw)
    return {key: (v[0] if v is not None else None) for key, v in r.items()}


def ltvlqr_batched(system, states, inputs, Qk, Rk, QT, dt, params, n_threads=0):
    """states (B,rows>=N,n), inputs (B,N,m) -> K (B,N,m,n)."""
    n, m, npar = dims(system)
    states, inputs, params = _d(states), _d(inputs), _d(params)
    B, N = inputs.shape[0], inputs.shape[1]
    rows = states.shape[1]
    ps = npar if params.ndim == 2 else 0
    K = np.empty((B, N, m, n))
    rc = lib().oracle_ltvlqr_batched(SYSTEM_IDS[system], C.c_int64(B), N, rows, _p(states), _p(inputs), _p(_d(Qk)),
                                     _p(_d(Rk)), _p(_d(QT)), C.c_double(dt), _p(params), C.c_int64(ps), _p(K),
                                     int(n_threads))
    if rc != 0:
        raise np.linalg.LinAlgError('Singular matrix')
    return K


def ltvlqr(system, states, inputs, Qk, Rk, QT, dt, params):
    return ltvlqr_batched(system, np.asarray(states, float)[None], np.asarray(inputs, float)[None], Qk, Rk, QT, dt,
                          params)[0]


# ---- receding-horizon tracking iLQR (matlab/smooth-ilqr/ilqr_mpc.m; restated, not pinned: MATLAB cannot run here) ----
def mpc_cost(states, inputs, x_des, u_des, Qk, Rk, QT):
    states, inputs = _d(states), _d(inputs)
    N, m = inputs.shape
    n = states.shape[1]
    lib().oracle_mpc_cost.restype = C.c_double
    return lib().oracle_mpc_cost(n, m, N, _p(states), _p(inputs), _p(_d(x_des)), _p(_d(u_des)), _p(_d(Qk)), _p(_d(Rk)), _p(_d(QT)))


def mpc_backward(system, states, inputs, x_des, u_des, Qk, Rk, QT, dt, p):
    n, m, _ = dims(system)
    states, inputs = _d(states), _d(inputs)
    N = inputs.shape[0]
    k, K, dJ = np.empty((N, m)), np.empty((N, m, n)), np.empty(3)
    rc = lib().oracle_mpc_backward(SYSTEM_IDS[system], N, _p(states), _p(inputs), _p(_d(x_des)), _p(_d(u_des)), _p(_d(Qk)),
                                   _p(_d(Rk)), _p(_d(QT)), C.c_double(dt), _p(_d(p)), _p(k), _p(K), _p(dJ))
    if rc != 0:
        raise np.linalg.LinAlgError('Singular matrix')
    return k, K, dJ


def mpc_shift(span, x_des, u_des, states, inputs, K):
    """in place on copies: returns the shifted (states, inputs, K) of ilqr_mpc.m:100-121"""
    states, inputs, K = _d(states).copy(), _d(inputs).copy(), _d(K).copy()
    N, m = inputs.shape
    n = states.shape[1]
    lib().oracle_mpc_shift(n, m, N, int(span), _p(_d(x_des)), _p(_d(u_des)), _p(states), _p(inputs), _p(K))
    return states, inputs, K


def mpc_solve_batched(system, first_call, current_state, x_des, u_des, states, inputs, k, K, Qk, Rk, QT, dt, params, n_iterations,
                      opts=DEFAULT_OPTS, n_threads=0):
    """one solve_ilqr call for B problems; states/inputs/k/K (B, ...) are the warm start, returned updated (copies)"""
    n, m, _ = dims(system)
    cur, x_des, u_des = _d(current_state).reshape(-1, n), _d(x_des), _d(u_des)
    states, inputs, k, K = _d(states).copy(), _d(inputs).copy(), _d(k).copy(), _d(K).copy()
    B, N = inputs.shape[0], inputs.shape[1]
    cost_ = np.empty(B); iters = np.empty(B, np.int32); status = np.empty(B, np.int32)
    trace = np.empty((B, n_iterations, 6))
    rc = lib().oracle_mpc_solve_batched(SYSTEM_IDS[system], C.c_int64(B), N, int(first_call), _p(cur), _p(x_des), _p(u_des), _p(_d(Qk)),
                                        _p(_d(Rk)), _p(_d(QT)), C.c_double(dt), _p(_d(params)), int(n_iterations), _p(_d(opts)),
                                        _p(states), _p(inputs), _p(k), _p(K), _p(cost_), iters.ctypes.data_as(_ip),
                                        status.ctypes.data_as(_ip), _p(trace), int(n_threads))
    assert rc == 0
    return dict(states=states, inputs=inputs, k=k, K=K, cost=cost_, iters=iters, status=status, trace=trace)


def num_threads():
    return lib().oracle_num_threads()
