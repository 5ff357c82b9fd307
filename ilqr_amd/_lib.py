"""ctypes binding of libilqr_hip.so (include/ilqr_hip.h).

The HIP library is the only compute path of this package: if it is missing or no GPU is visible the
functions here raise -- there is no CPU fallback (the CPU restatement under oracle/ is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ILQR_HIP_LIB: alternative build of the same library (A/B timing of kernel variants); never a different backend
LIB_PATH = os.environ.get('ILQR_HIP_LIB') or os.path.join(_HERE, 'libilqr_hip.so')

SYSTEM_IDS = {'pendulum': 0, 'acrobot': 1, 'cartpole': 2, 'quadcopter': 3, 'car': 4, 'double_integrator': 5,
              'double_pendulum': 6, 'floating_base': 7}
STATUS_NAMES = {0: 'max_iter', 1: 'optimal', 2: 'low_learning_rate', 3: 'numerical'}

# fields of the device-resident workspace (include/ilqr_hip.h)
F_INIT, F_TARGET, F_PARAMS, F_STATES, F_INPUTS, F_KFF, F_KFB, F_COST, F_DJ, F_ITERS, F_STATUS, F_TRACE, \
    F_LS_ROUNDS, F_ACCEPTS, F_ALPHA, F_STATES_OUT, F_INPUTS_OUT, F_NEW_COST, F_GUESS, F_TARGET_STATES, F_TARGET_INPUTS = range(21)
KERNEL_NAMES = ('rollout', 'backward', 'linesearch', 'accept', 'ltvlqr', 'forward', 'cost', 'linearize')


ERR_SINGULAR = -5      # ILQR_ERR_SINGULAR: np.linalg.inv would have raised LinAlgError
ERR_SCRATCH = -6       # ILQR_ERR_SCRATCH: ltvlqr_device_noalloc, batch larger than the static scratch array
NOALLOC_SCRATCH_BYTES = 64 * 1024 * 1024


class IlqrHipError(RuntimeError):
    pass


class Desc(C.Structure):
    _fields_ = [('system_id', C.c_int32), ('n_timesteps', C.c_int32), ('batch', C.c_int64), ('dt', C.c_double),
                ('n_iterations', C.c_int32), ('n_alpha_parallel', C.c_int32), ('zero_first_state', C.c_int32),
                ('device', C.c_int32), ('learning_speed', C.c_double), ('low_learning_rate', C.c_double),
                ('low_expected_reduction', C.c_double), ('armijo_threshold', C.c_double), ('quu_reg', C.c_double),
                ('target_batched', C.c_int32), ('params_batched', C.c_int32), ('pipeline', C.c_int32), ('n_streams', C.c_int32), ('riccati', C.c_int32), ('overlap', C.c_int32),
                ('search_floor', C.c_double), ('n_devices', C.c_int32), ('device_wrap', C.c_int32), ('tracking', C.c_int32)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lib = None

# every exported symbol of include/ilqr_hip.h: (name, restype, argtypes)
_DESCP = C.POINTER(Desc)
_WS = C.c_void_p
SIGNATURES = {
    'ilqr_desc_default': (None, [_DESCP]),
    'ilqr_abi_version': (C.c_int, []),
    'ilqr_device_count': (C.c_int, []),
    'ilqr_last_error': (C.c_char_p, []),
    'ilqr_system_lookup': (C.c_int, [C.c_char_p, _ip, _ip, _ip]),
    'ilqr_system_dims': (C.c_int, [C.c_int32, _ip, _ip, _ip]),
    'ilqr_dynamics_eval': (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _dp, _dp, C.c_double, _dp, _dp, C.c_int32]),
    'ilqr_rollout_batched': (C.c_int, [_DESCP, _dp, _dp, _dp, _dp, _dp]),
    'ilqr_cost_batched': (C.c_int, [_DESCP, _dp, _dp, _dp, _dp, _dp, _dp]),
    'ilqr_backward_batched': (C.c_int, [_DESCP, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    'ilqr_forward_batched': (C.c_int, [_DESCP] + [_dp] * 13),
    'ilqr_solve_batched': (C.c_int, [_DESCP] + [_dp] * 12 + [_ip, _ip, _dp]),
    'ltvlqr_batched': (C.c_int, [_DESCP, C.c_int32] + [_dp] * 7 + [_ip]),
    'ltvlqr_device_noalloc': (C.c_int, [_DESCP, C.c_int32] + [C.c_void_p] * 7 + [C.c_void_p, C.c_void_p]),
    'ilqr_ws_create': (C.c_int, [_DESCP, C.c_int32, C.POINTER(_WS)]),
    'ilqr_ws_destroy': (C.c_int, [_WS]),
    'ilqr_ws_set_weights': (C.c_int, [_WS, _dp, _dp, _dp]),
    'ilqr_ws_upload': (C.c_int, [_WS, C.c_int32, C.c_void_p, C.c_int64]),
    'ilqr_ws_upload_device': (C.c_int, [_WS, C.c_int32, C.c_void_p, C.c_int64]),
    'ilqr_ws_download': (C.c_int, [_WS, C.c_int32, C.c_void_p]),
    'ilqr_ws_download_device': (C.c_int, [_WS, C.c_int32, C.c_void_p]),
    'ilqr_ws_reset': (C.c_int, [_WS]),
    'ilqr_ws_rollout': (C.c_int, [_WS]),
    'ilqr_ws_cost': (C.c_int, [_WS]),
    'ilqr_ws_backward': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_linesearch': (C.c_int, [_WS]),
    'ilqr_ws_accept': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_forward': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_solve': (C.c_int, [_WS]),
    'ilqr_ws_ltvlqr': (C.c_int, [_WS]),
    'ilqr_ws_mpc_shift': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_mpc_solve': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_sync': (C.c_int, [_WS]),
    'ilqr_ws_query': (C.c_int, [_WS]),
    'ilqr_ws_profile': (C.c_int, [_WS, C.c_int32]),
    'ilqr_ws_profile_read': (C.c_int, [_WS, _dp, C.POINTER(C.c_int64)]),
    'ilqr_ws_info': (C.c_int, [_WS, _ip, _ip, C.POINTER(C.c_int64)]),
    'ilqr_ws_ls_schedule': (C.c_int, [_WS, _ip, _ip]),
    'ilqr_ws_pipeline': (C.c_int, [_WS]),
}


def lib():
    """Load libilqr_hip.so (built in-tree by ``make -C ilqr_amd/csrc`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IlqrHipError('%s not found: build it with `make -C ilqr_amd/csrc` (hipcc, gfx950). '
                               'There is no CPU fallback.' % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc < 0:
        raise IlqrHipError('libilqr_hip: %s (code %d)' % (lib().ilqr_last_error().decode(), rc))
    return rc


def require_gpu():
    if lib().ilqr_device_count() < 1:
        raise IlqrHipError('no HIP device visible: ilqr_amd computes on an AMD GPU only (no CPU fallback)')


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


NO_PARAMETER_SYSTEMS = ('car',)     # MATLAB models with `parameters = []`: compiled with one unused slot


def params_of(system, parameters):
    """float64 parameters; an empty vector for a system without physical parameters fills its unused slot."""
    p = f64(parameters)
    if p.size == 0 and system in NO_PARAMETER_SYSTEMS:
        p = np.zeros(1)
    return p


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def system_dims(system):
    sid = SYSTEM_IDS[system] if isinstance(system, str) else int(system)
    n, m, npar = C.c_int32(), C.c_int32(), C.c_int32()
    check(lib().ilqr_system_dims(sid, C.byref(n), C.byref(m), C.byref(npar)))
    return n.value, m.value, npar.value


def make_desc(system, N, B, dt, n_iterations=0, n_alpha_parallel=0, zero_first_state=True, device=0,
              target_batched=False, params_batched=False, learning_speed=0.95, low_learning_rate=0.05,
              low_expected_reduction=1e-3, armijo_threshold=0.1, quu_reg=0.0, pipeline=0, n_streams=0, riccati=0, overlap=0,
              search_floor=0.05, n_devices=0, device_wrap=False, tracking=False):
    d = Desc()
    lib().ilqr_desc_default(C.byref(d))
    d.system_id = SYSTEM_IDS[system] if isinstance(system, str) else int(system)
    d.n_timesteps = int(N)
    d.batch = int(B)
    d.dt = float(dt)
    d.n_iterations = int(n_iterations)
    d.n_alpha_parallel = int(n_alpha_parallel)
    d.zero_first_state = 1 if zero_first_state else 0
    d.device = int(device)
    d.target_batched = 1 if target_batched else 0
    d.params_batched = 1 if params_batched else 0
    d.learning_speed = float(learning_speed)
    d.low_learning_rate = float(low_learning_rate)
    d.low_expected_reduction = float(low_expected_reduction)
    d.armijo_threshold = float(armijo_threshold)
    d.quu_reg = float(quu_reg)
    d.pipeline = int(pipeline)
    d.n_streams = int(n_streams)
    d.riccati = int(riccati)
    d.overlap = int(overlap)
    d.search_floor = float(search_floor)
    d.n_devices = int(n_devices)
    d.device_wrap = 1 if device_wrap else 0
    d.tracking = 1 if tracking else 0
    return d


class Workspace:
    """Device-resident batch (``ilqr_ws`` of include/ilqr_hip.h): upload once, run passes, download results."""

    _SHAPES = None

    def __init__(self, system, N, B, dt, want_trace=False, **desc_kw):
        require_gpu()
        self.system = system
        self.n, self.m, self.np = system_dims(system)
        self.N, self.B = int(N), int(B)
        self.desc = make_desc(system, N, B, dt, **desc_kw)
        self.n_iterations = self.desc.n_iterations
        self._h = _WS()
        check(lib().ilqr_ws_create(C.byref(self.desc), 1 if want_trace else 0, C.byref(self._h)))
        self.want_trace = bool(want_trace) and self.n_iterations > 0

    # -- lifetime
    def close(self):
        if self._h:
            lib().ilqr_ws_destroy(self._h)
            self._h = _WS()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data movement
    def _shape(self, field):
        B, N, n, m, P = self.B, self.N, self.n, self.m, self.np
        return {F_INIT: (B, n), F_TARGET: (B, n) if self.desc.target_batched else (n,),
                F_PARAMS: (B, P) if self.desc.params_batched else (P,), F_STATES: (B, N + 1, n),
                F_INPUTS: (B, N, m), F_KFF: (B, N, m), F_KFB: (B, N, m, n), F_COST: (B,), F_DJ: (B, 3), F_ITERS: (B,),
                F_STATUS: (B,), F_TRACE: (B, self.n_iterations, 6), F_LS_ROUNDS: (B,), F_ACCEPTS: (B,), F_ALPHA: (B,),
                F_STATES_OUT: (B, N + 1, n), F_INPUTS_OUT: (B, N, m), F_NEW_COST: (B,), F_GUESS: (B, N, m),
                F_TARGET_STATES: (B, N + 1, n), F_TARGET_INPUTS: (B, N, m)}[field]

    def set_weights(self, Qk, Rk, QT):
        n, m = self.n, self.m
        Qk = f64(np.zeros((n, n)) if Qk is None else Qk).reshape(n, n)
        Rk = f64(np.zeros((m, m)) if Rk is None else Rk).reshape(m, m)
        QT = f64(np.zeros((n, n)) if QT is None else QT).reshape(n, n)
        check(lib().ilqr_ws_set_weights(self._h, dptr(Qk), dptr(Rk), dptr(QT)))

    def upload(self, field, array, rows=0):
        a = f64(array)
        if field == F_STATES and rows:
            expect = (self.B, rows, self.n)
        else:
            expect = self._shape(field)
        if a.size != int(np.prod(expect)):
            raise ValueError('field %d expects shape %s, got %s' % (field, expect, a.shape))
        check(lib().ilqr_ws_upload(self._h, field, a.ctypes.data_as(C.c_void_p), int(rows)))

    def upload_device(self, field, dev_ptr, rows=0):
        check(lib().ilqr_ws_upload_device(self._h, field, C.c_void_p(int(dev_ptr)), int(rows)))

    def download(self, field):
        shape = self._shape(field)
        dtype = np.int32 if field in (F_ITERS, F_STATUS, F_LS_ROUNDS, F_ACCEPTS) else np.float64
        out = np.empty(shape, dtype=dtype)
        check(lib().ilqr_ws_download(self._h, field, out.ctypes.data_as(C.c_void_p)))
        return out

    def download_device(self, field, dev_ptr):
        check(lib().ilqr_ws_download_device(self._h, field, C.c_void_p(int(dev_ptr))))

    # -- passes (asynchronous on the workspace stream)
    def reset(self): check(lib().ilqr_ws_reset(self._h))
    def rollout(self): check(lib().ilqr_ws_rollout(self._h))
    def cost(self): check(lib().ilqr_ws_cost(self._h))
    def backward(self, iteration=0): check(lib().ilqr_ws_backward(self._h, int(iteration)))
    def linesearch(self): check(lib().ilqr_ws_linesearch(self._h))
    def accept(self, iteration=0): check(lib().ilqr_ws_accept(self._h, int(iteration)))
    def forward(self, with_cost=True): check(lib().ilqr_ws_forward(self._h, 1 if with_cost else 0))
    def solve(self): check(lib().ilqr_ws_solve(self._h))
    def ltvlqr(self): check(lib().ilqr_ws_ltvlqr(self._h))
    def mpc_shift(self, span): check(lib().ilqr_ws_mpc_shift(self._h, int(span)))
    def mpc_solve(self, first_call): check(lib().ilqr_ws_mpc_solve(self._h, 1 if first_call else 0))
    def sync(self): check(lib().ilqr_ws_sync(self._h))
    def query(self): return check(lib().ilqr_ws_query(self._h))     # 0 idle, 1 work pending (non-blocking)

    def profile(self, enable=True, only=None):
        """HIP-event timing of the passes: every kernel, or only the one named `only` (KERNEL_NAMES)"""
        check(lib().ilqr_ws_profile(self._h, (2 + KERNEL_NAMES.index(only)) if (enable and only) else (1 if enable else 0)))

    def profile_read(self):
        ms = np.zeros(8)
        cnt = np.zeros(8, dtype=np.int64)
        check(lib().ilqr_ws_profile_read(self._h, dptr(ms), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return {KERNEL_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(8)}

    def info(self):
        w, t, b = C.c_int32(), C.c_int32(), C.c_int64()
        check(lib().ilqr_ws_info(self._h, C.byref(w), C.byref(t), C.byref(b)))
        nr = C.c_int32()
        widths = (C.c_int32 * 64)()
        check(lib().ilqr_ws_ls_schedule(self._h, C.byref(nr), widths))
        return dict(n_alpha_parallel=w.value, n_trials=t.value, hbm_bytes=b.value, ls_widths=[int(widths[i]) for i in range(nr.value)],
                    pipeline={1: 'fused', 2: 'split'}[check(lib().ilqr_ws_pipeline(self._h))])
