"""ltvlqr_device_noalloc (include/ilqr_hip.h): the allocation-free variant of ltvlqr_batched, through the C ABI.

It has no counterpart in the reference; the check is against the existing entry point it is a variant of
(ltvlqr_batched, same kernels): bit-identical gains and status flags, on two calls in a row on the same GPU, the second
one smaller in batch and horizon (the static scratch array keeps the first call's contents), for the one-lane kernel
(acrobot) and the 16-lane group path with its LIN buffer (quadcopter); a batch beyond the array is refused with a clear
error; the device's free memory does not move across the calls.

Device memory for the test comes from the HIP runtime the library itself is linked against, through ctypes (hipMalloc /
hipMemcpy here are the TEST's allocations, made before the calls under test) -- not from torch: a process that loads
libilqr_hip.so first and imports torch afterwards ends up with two HIP runtimes, and the second one sees no GPU
("one HIP runtime per process").
"""
import ctypes as C

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


class Hip:
    """the few HIP runtime calls the test needs, on the runtime libilqr_hip.so has already loaded"""

    def __init__(self):
        from ilqr_amd import _lib
        _lib.require_gpu()
        paths = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})
        assert len(paths) == 1, paths
        self.rt = C.CDLL(paths[0])
        self.live = []

    def chk(self, rc):
        assert rc == 0, 'HIP error %d' % rc

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self.chk(self.rt.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 8))))
        self.chk(self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1))      # hipMemcpyHostToDevice
        self.live.append(p)
        return p

    def filled(self, shape, dtype, value):
        return self.to_device(np.full(shape, value, dtype=dtype))

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self.chk(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), 2))   # hipMemcpyDeviceToHost
        return out

    def sync(self):
        self.chk(self.rt.hipDeviceSynchronize())

    def free_bytes(self):
        free, total = C.c_size_t(), C.c_size_t()
        self.chk(self.rt.hipMemGetInfo(C.byref(free), C.byref(total)))
        return free.value

    def release(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


@pytest.fixture()
def hip():
    h = Hip()
    yield h
    h.release()


def _problem(system, Bn, N, seed):
    g = golden(system + '_batch')
    rng = np.random.default_rng(seed)
    n, m = g['init'].shape[1], g['guess'].shape[2]
    base_x = np.resize(g['states'][0], (N + 1, n))
    base_u = np.resize(g['inputs'][0], (N, m))
    states = base_x[None] + 1e-2 * rng.standard_normal((Bn, N + 1, n))
    inputs = base_u[None] + 1e-2 * rng.standard_normal((Bn, N, m))
    params = np.tile(g['params'], (Bn, 1)) * (1.0 + rng.uniform(-0.01, 0.01, (Bn, g['params'].size)))
    return g, states, inputs, params


def _host(lib, _lib, system, g, states, inputs, params, N, rows, Qk, Rk, QT):
    Bn, n, m = states.shape[0], states.shape[2], inputs.shape[2]
    d = _lib.make_desc(system, N, Bn, float(g['dt']), params_batched=True)
    K = np.empty((Bn, N, m, n))
    st = np.empty(Bn, np.int32)
    _lib.check(lib.ltvlqr_batched(C.byref(d), rows, _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(Qk), _lib.dptr(Rk), _lib.dptr(QT),
                                  _lib.dptr(params), _lib.dptr(K), _lib.iptr(st)))
    return K, st


@pytest.mark.parametrize('system,sizes', [('acrobot', ((200, 700), (70, 333))), ('quadcopter', ((100, 150), (37, 90)))])
def test_noalloc_ltvlqr_equals_ltvlqr_batched(hip, system, sizes):
    from ilqr_amd import _lib
    lib = _lib.lib()
    cases = []
    for i, (Bn, N) in enumerate(sizes):                       # the second call is smaller in both batch and horizon
        g, states, inputs, params = _problem(system, Bn, N, 40 + i)
        Qk, Rk, QT = (_lib.f64(g[k_]) for k_ in ('Qk', 'Rk', 'QT'))
        K_ref, st_ref = _host(lib, _lib, system, g, states, inputs, params, N, N + 1, Qk, Rk, QT)
        t = {k_: hip.to_device(v) for k_, v in dict(states=states, inputs=inputs, params=params, Qk=Qk, Rk=Rk, QT=QT).items()}
        t['K'] = hip.filled(K_ref.shape, np.float64, np.nan)
        t['status'] = hip.filled((Bn,), np.int32, -7)
        cases.append((g, Bn, N, K_ref, st_ref, t))
    hip.sync()
    free_before = hip.free_bytes()
    for g, Bn, N, K_ref, st_ref, t in cases:
        d = _lib.make_desc(system, N, Bn, float(g['dt']), params_batched=True)
        _lib.check(lib.ltvlqr_device_noalloc(C.byref(d), N + 1, t['states'], t['inputs'], t['Qk'], t['Rk'], t['QT'], t['params'], t['K'],
                                             t['status'], None))
    hip.sync()
    free_after = hip.free_bytes()
    for g, Bn, N, K_ref, st_ref, t in cases:
        K = hip.to_host(t['K'], K_ref.shape, np.float64)
        assert np.isfinite(K).all()
        assert np.array_equal(K, K_ref), (Bn, N)              # same kernels, same bits
        assert np.array_equal(hip.to_host(t['status'], (Bn,), np.int32), st_ref) and np.all(st_ref == 0)
    assert free_after == free_before, (free_before, free_after)


def test_noalloc_ltvlqr_reports_singular_problems_and_refuses_oversize(hip):
    from ilqr_amd import _lib
    lib = _lib.lib()
    g = golden('pendulum_ltvlqr')
    N = g['inputs'].shape[0]
    Bn = 5
    states, inputs = np.tile(g['states'], (Bn, 1, 1)), np.tile(g['inputs'], (Bn, 1, 1))
    rows = states.shape[1]
    tt = lambda a: hip.to_device(np.asarray(a, dtype=np.float64))
    ts, tu, tp, tQ = tt(states), tt(inputs), tt(g['params']), tt(g['Qk'])
    tR, tT, zR, zT = tt(g['Rk']), tt(g['QT']), tt(np.zeros((1, 1))), tt(np.zeros((2, 2)))
    K = hip.filled((Bn, N, 1, 2), np.float64, 0.0)
    st = hip.filled((Bn,), np.int32, 0)
    d = _lib.make_desc('pendulum', N, Bn, float(g['dt']))
    # R = Q_T = 0: R + B'PB is exactly zero at the first step (np.linalg.inv raises at ltvlqr.py:24) -> status 3 everywhere
    _lib.check(lib.ltvlqr_device_noalloc(C.byref(d), rows, ts, tu, tQ, zR, zT, tp, K, st, None))
    hip.sync()
    assert np.all(hip.to_host(st, (Bn,), np.int32) == 3)
    # the next call on the same array starts from clean flags, and matches the reference's own gains
    _lib.check(lib.ltvlqr_device_noalloc(C.byref(d), rows, ts, tu, tQ, tR, tT, tp, K, st, None))
    hip.sync()
    assert np.all(hip.to_host(st, (Bn,), np.int32) == 0)
    Kh = hip.to_host(K, (Bn, N, 1, 2), np.float64)
    assert np.max(np.abs(Kh[0] - g['K'])) / np.max(np.abs(g['K'])) < 1e-8 and np.array_equal(Kh[0], Kh[4])
    # beyond the static array: refused with a message that names the limit; nothing is launched
    big = _lib.make_desc('acrobot', 3000, 4096, 1e-3)
    rc = lib.ltvlqr_device_noalloc(C.byref(big), 3001, ts, tu, tQ, tQ, tQ, tp, K, st, None)
    assert rc == _lib.ERR_SCRATCH
    msg = lib.ilqr_last_error().decode()
    assert 'scratch' in msg and str(_lib.NOALLOC_SCRATCH_BYTES) in msg and '256 problems' in msg, msg
    two = _lib.make_desc('pendulum', N, Bn, float(g['dt']), n_devices=2)
    assert lib.ltvlqr_device_noalloc(C.byref(two), rows, ts, tu, tQ, tQ, tQ, tp, K, st, None) < 0
