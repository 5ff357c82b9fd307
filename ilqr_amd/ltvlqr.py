"""Drop-in for the reference's ``ltvlqr`` (python/ltvlqr.py:3-28), computed by the HIP kernel ``k_ltvlqr``."""
import numpy as np

from . import _lib
from .symbolic_dynamics import system_of


def ltvlqr(states, inputs, A, B, Q_k, R_k, Q_T, dt, parameters, device=0):
    """Time-varying discrete Riccati gain schedule along a trajectory.

    Reference signature and return value (ltvlqr.py:3, :28): ``states (rows>=N, n)``, ``inputs (N, m)`` ->
    ``K_feedback (N, m, n)``.  Additive batch extension: ``states (B, rows, n)``, ``inputs (B, N, m)`` (and optionally
    ``parameters (B, P)``) -> ``(B, N, m, n)``.
    """
    system = system_of(A, B)
    n, m, npar = _lib.system_dims(system)
    _lib.require_gpu()
    states = _lib.f64(states)
    inputs = _lib.f64(inputs)
    batched = inputs.ndim == 3
    if not batched:
        states, inputs = states[None], inputs[None]
    Bn, N = inputs.shape[0], inputs.shape[1]                      # ltvlqr.py:8
    rows = states.shape[1]
    if states.shape[2] != n or inputs.shape[2] != m:
        raise ValueError('%s expects %d states and %d inputs' % (system, n, m))
    params = _lib.params_of(system, parameters)
    pb = params.ndim == 2
    desc = _lib.make_desc(system, N, Bn, dt, device=device, params_batched=pb)
    K = np.empty((Bn, N, m, n))                                   # ltvlqr.py:12
    Qk, Rk, QT = (_lib.f64(Q_k).reshape(n, n), _lib.f64(R_k).reshape(m, m), _lib.f64(Q_T).reshape(n, n))
    _lib.check(_lib.lib().ltvlqr_batched(desc, rows, _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(Qk), _lib.dptr(Rk),
                                         _lib.dptr(QT), _lib.dptr(params), _lib.dptr(K)))
    return K if batched else K[0]
