"""Drop-in for the reference's ``ltvlqr`` (python/ltvlqr.py:3-28), computed by the HIP kernel ``k_ltvlqr``."""
import numpy as np

from . import _lib
from .symbolic_dynamics import system_of


def ltvlqr(states, inputs, A, B, Q_k, R_k, Q_T, dt, parameters, device=0, n_devices=1, return_status=False):
    """Time-varying discrete Riccati gain schedule along a trajectory.

    Reference signature and return value (ltvlqr.py:3, :28): ``states (rows>=N, n)``, ``inputs (N, m)`` ->
    ``K_feedback (N, m, n)``.  Additive batch extension: ``states (B, rows, n)``, ``inputs (B, N, m)`` (and optionally
    ``parameters (B, P)``) -> ``(B, N, m, n)``; ``n_devices=G`` splits the batch over G GPUs by problem index
    (``0`` = every visible GPU).

    Like ``np.linalg.inv`` at ltvlqr.py:24 a single problem whose ``R + B'PB`` is exactly singular raises
    ``np.linalg.LinAlgError``.  A batch does not stop for one bad problem: it raises only with ``return_status=False``
    (default); with ``return_status=True`` it returns ``(K, status)``, status 3 marking the singular problems.
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
    if states.ndim != 3 or states.shape[2] != n or inputs.shape[2] != m:
        raise ValueError('%s expects %d states and %d inputs' % (system, n, m))
    if states.shape[0] != Bn:
        raise ValueError('states has %d problems, inputs %d' % (states.shape[0], Bn))
    if rows < N:
        raise ValueError('states needs at least N=%d rows, got %d' % (N, rows))
    params = _lib.params_of(system, parameters)
    pb = params.ndim == 2
    if params.shape != ((Bn, npar) if pb else (npar,)):
        raise ValueError('%s takes parameters of shape (%d,) or (%d, %d), got %s' % (system, npar, Bn, npar, params.shape))
    if n_devices == 0:
        n_devices = _lib.lib().ilqr_device_count()
    desc = _lib.make_desc(system, N, Bn, dt, device=device, params_batched=pb, n_devices=n_devices)
    K = np.empty((Bn, N, m, n))                                   # ltvlqr.py:12
    status = np.zeros(Bn, dtype=np.int32)
    Qk, Rk, QT = (_lib.f64(Q_k).reshape(n, n), _lib.f64(R_k).reshape(m, m), _lib.f64(Q_T).reshape(n, n))
    _lib.check(_lib.lib().ltvlqr_batched(desc, rows, _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(Qk), _lib.dptr(Rk),
                                         _lib.dptr(QT), _lib.dptr(params), _lib.dptr(K), _lib.iptr(status)))
    if return_status:
        return (K, status) if batched else (K[0], int(status[0]))
    if np.any(status == 3):
        raise np.linalg.LinAlgError('Singular matrix')             # what np.linalg.inv raises at ltvlqr.py:24
    return K if batched else K[0]
