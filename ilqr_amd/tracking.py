"""Closed-loop tracking of a trajectory with a time-varying gain schedule.

The reference has no function for this: both ``*_ltvlqr_tracking.py`` examples write the loop out
(python/example_acrobot_swingup_ltvlqr_tracking.py:47-67, python/example_pendulum_swingup_ltvlqr_tracking.py:61-81)::

    current_error = current_state - states[ii]          u = inputs[ii] + K[ii] @ current_error
    next_state    = f(current_state, u, dt, parameters)

from a perturbed initial state, storing that state in row 0.  That is ``ilqr.forwards_pass`` (ilqr.py:137-158) with a
zero feed-forward term and row 0 written, so it runs on the same HIP kernel (``k_forward``) through the same C entry
point, ``ilqr_forward_batched``, with ``zero_first_state = 0``.
"""
import numpy as np

from . import _lib
from .symbolic_dynamics import system_of


def tracking_rollout(states, inputs, K_feedback, f, init_state, dt, parameters, device=0):
    """Simulate ``x <- f(x, inputs[i] + K_feedback[i] @ (x - states[i]))`` for i = 0..N-1.

    ``states (rows>=N, n)``, ``inputs (N, m)``, ``K_feedback (N, m, n)`` as returned by ``ltvlqr``; ``init_state (n,)``
    -> ``(perturbed_states (N+1, n), perturbed_inputs (N, m))``, the two arrays the reference examples fill.
    Batch extension: ``init_state (B, n)`` simulates B perturbations; the trajectory and gains are shared when given
    without a batch axis, per problem when given with one (``states (B, rows, n)``, ``inputs (B, N, m)``,
    ``K_feedback (B, N, m, n)``); ``parameters`` may be ``(B, P)``.  Outputs then carry the leading B axis.
    """
    system = system_of(f)
    n, m, npar = _lib.system_dims(system)
    _lib.require_gpu()
    init = _lib.f64(init_state)
    states, inputs, K = _lib.f64(states), _lib.f64(inputs), _lib.f64(K_feedback)
    params = _lib.params_of(system, parameters)
    batched = init.ndim == 2 or inputs.ndim == 3 or K.ndim == 4
    Bn = max(init.shape[0] if init.ndim == 2 else 1, inputs.shape[0] if inputs.ndim == 3 else 1,
             K.shape[0] if K.ndim == 4 else 1, states.shape[0] if states.ndim == 3 else 1)
    N = inputs.shape[-2]
    if inputs.shape[-1] != m or states.shape[-1] != n or K.shape[-2:] != (m, n) or init.shape[-1] != n:
        raise ValueError('%s expects %d states and %d inputs' % (system, n, m))
    if K.shape[-3] != N or states.shape[-2] < N:
        raise ValueError('K_feedback needs N=%d steps and states at least N rows' % N)

    def bcast(a, core_ndim):
        if a.ndim == core_ndim:
            a = np.broadcast_to(a, (Bn,) + a.shape)
        elif a.shape[0] != Bn:
            raise ValueError('batch axes disagree: %d vs %d' % (a.shape[0], Bn))
        return np.ascontiguousarray(a)

    # the kernel reads rows 0..N of the reference states buffer layout (N+1 rows); row N is never used by the loop
    ref_states = np.zeros((Bn, N + 1, n))
    src = states if states.ndim == 3 else states[None]
    rows = min(src.shape[1], N + 1)
    ref_states[:, :rows] = src[:, :rows]
    ref_inputs = bcast(inputs, 2)
    Kb = bcast(K, 3)
    initb = bcast(init, 1)
    pb = params.ndim == 2
    desc = _lib.make_desc(system, N, Bn, dt, device=device, params_batched=pb, zero_first_state=False)
    k0 = np.zeros((Bn, N, m))
    lr = np.zeros(Bn)
    out_states = np.empty((Bn, N + 1, n))
    out_inputs = np.empty((Bn, N, m))
    _lib.check(_lib.lib().ilqr_forward_batched(desc, _lib.dptr(initb), _lib.dptr(ref_states), _lib.dptr(ref_inputs),
                                               _lib.dptr(k0), _lib.dptr(Kb), _lib.dptr(lr), _lib.dptr(params),
                                               _lib.dptr(out_states), _lib.dptr(out_inputs), None, None, None, None))
    return (out_states, out_inputs) if batched else (out_states[0], out_inputs[0])
