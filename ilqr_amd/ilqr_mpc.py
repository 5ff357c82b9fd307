"""Receding-horizon trajectory-tracking iLQR: the reference's MATLAB class ``ilqr_mpc``
(matlab/smooth-ilqr/ilqr_mpc.m) computed on an AMD MI355X through libilqr_hip.so.

Same constructor arguments and the same method as the MATLAB class (ilqr_mpc.m:41, :63)::

    mpc = ilqr_mpc(target_states, initial_guess, dt, horizon, f_disc, A, B, Q_k, R_k, Q_T, parameters, n_iterations)
    states, inputs, k_feedforward, K_feedback, current_cost = mpc.solve_ilqr(current_idx, current_state)

``current_idx`` counts from 1 as in MATLAB (the example loops ``for ii=1:duration``, example_acrobot_ilqr_mpc.m:47): the
call tracks ``target_states[current_idx-1 : current_idx+horizon]`` / ``target_inputs[current_idx-1 : current_idx-1+horizon]``,
so the targets must be padded by ``horizon`` rows as the examples do (:27-28).  The first call (``current_idx == 1``) takes
its gain schedule from one backward pass along the reference (:68-78); later calls warm-start from the previous solution
moved ``current_idx - previous_idx`` steps (:100-121).  The MATLAB class has no python twin in the reference, so parity of
this path is against a line-by-line restatement of the .m file (oracle/ilqr_oracle.c), not against reference outputs.

Additive batch extension: ``current_state (B, n)`` solves B independent trackers at once (each with its own warm start);
``target_states`` / ``initial_guess`` may be shared ``(T, n)`` / ``(T, m)`` or per problem ``(B, T, n)`` / ``(B, T, m)``.
The batch lives in a device-resident workspace between calls (only the current states, the target window and the
solution cross the PCIe bus).
"""
import numpy as np

from . import _lib
from .symbolic_dynamics import system_of


class ilqr_mpc:
    def __init__(self, target_states, initial_guess, dt, horizon, f_disc, A, B, Q_k, R_k, Q_T, parameters, n_iterations,
                 device=0, n_alpha_parallel=0):
        self.target_states_ = _lib.f64(target_states)              # ilqr_mpc.m:42
        self.target_inputs_ = _lib.f64(initial_guess)              # :43
        self.n_states_ = self.target_states_.shape[-1]             # :44
        self.n_inputs_ = self.target_inputs_.shape[-1]             # :45
        self.dt_ = dt
        self.n_timesteps_ = int(horizon)                           # :49
        self.f_, self.A_, self.B_ = f_disc, A, B
        self.system_ = system_of(f_disc, A, B)
        self.Q_k_, self.R_k_, self.Q_T_ = Q_k, R_k, Q_T
        self.parameters_ = parameters
        self.n_iterations_ = int(n_iterations)                     # :61
        self.device_, self.n_alpha_parallel_ = device, n_alpha_parallel
        n, m, npar = _lib.system_dims(self.system_)
        if (self.n_states_, self.n_inputs_) != (n, m):
            raise ValueError('%s has %d states and %d inputs; the targets have %d and %d' % (self.system_, n, m, self.n_states_, self.n_inputs_))
        if self.target_states_.ndim not in (2, 3) or self.target_inputs_.ndim != self.target_states_.ndim:
            raise ValueError('target_states / initial_guess must be (T,n) / (T,m) or (B,T,n) / (B,T,m)')
        if self.n_timesteps_ < 1 or self.n_iterations_ < 1:
            raise ValueError('horizon and n_iterations must be >= 1')
        p = _lib.params_of(self.system_, parameters)
        if p.shape != (npar,):
            raise ValueError('%s takes %d parameters' % (self.system_, npar))
        for name, w, k_ in (('Q_k', Q_k, n), ('R_k', R_k, m), ('Q_T', Q_T, n)):
            if np.size(w) != k_ * k_:
                raise ValueError('%s must be (%d,%d)' % (name, k_, k_))
        self.current_idx_ = None
        self._ws = None

    def _window(self, arr, i0, rows):
        """rows of the target arrays for the horizon window starting at 0-based index i0, with a leading batch axis"""
        w = arr[..., i0:i0 + rows, :]
        if w.shape[-2] != rows:
            raise ValueError('the targets end before index %d + horizon: pad them by `horizon` rows (example_acrobot_ilqr_mpc.m:27-28)' % (i0 + 1))
        if w.ndim == 2:
            w = np.broadcast_to(w, (self.batch_,) + w.shape)
        elif w.shape[0] != self.batch_:
            raise ValueError('targets carry a batch of %d, current_state one of %d' % (w.shape[0], self.batch_))
        return np.ascontiguousarray(w)

    def solve_ilqr(self, current_idx, current_state):
        """ilqr_mpc.m:63-192; current_idx is 1-based."""
        _lib.require_gpu()
        cur = _lib.f64(current_state)
        n, m, N = self.n_states_, self.n_inputs_, self.n_timesteps_
        if cur.shape[-1] != n or cur.ndim not in (1, 2):
            raise ValueError('current_state must be (%d,) or (B,%d)' % (n, n))
        first = int(current_idx) == 1                                          # :68
        if self._ws is None:
            if not first:
                raise ValueError('the first call must be solve_ilqr(1, ...): later calls warm-start from it (ilqr_mpc.m:68-79)')
            self.batched_ = cur.ndim == 2
            self.batch_ = cur.shape[0] if self.batched_ else 1
            self._ws = _lib.Workspace(self.system_, N, self.batch_, self.dt_, want_trace=True, n_iterations=self.n_iterations_,
                                      n_alpha_parallel=self.n_alpha_parallel_, zero_first_state=False, device=self.device_,
                                      tracking=True)
            self._ws.set_weights(self.Q_k_, self.R_k_, self.Q_T_)
            self._ws.upload(_lib.F_PARAMS, _lib.params_of(self.system_, self.parameters_))
        ws = self._ws
        cur = cur.reshape(self.batch_, n) if cur.size == self.batch_ * n else None
        if cur is None:
            raise ValueError('current_state does not match the batch of the first call (%d)' % self.batch_)
        i0 = int(current_idx) - 1
        xd, ud = self._window(self.target_states_, i0, N + 1), self._window(self.target_inputs_, i0, N)
        ws.upload(_lib.F_TARGET_STATES, xd)
        ws.upload(_lib.F_TARGET_INPUTS, ud)
        ws.upload(_lib.F_INIT, cur)
        if first:
            ws.upload(_lib.F_STATES, xd)                                       # :73-74
            ws.upload(_lib.F_INPUTS, ud)
        else:
            span = int(current_idx) - self.current_idx_                        # :82
            if span < 0 or span > N:
                raise ValueError('solve_ilqr was last called at index %d; %d is not within one horizon after it' % (self.current_idx_, current_idx))
            ws.mpc_shift(span)                                                 # :100-121
        self.current_idx_ = int(current_idx)
        ws.mpc_solve(first)                                                    # :75-78, :127-184
        ws.sync()
        out = [ws.download(f) for f in (_lib.F_STATES, _lib.F_INPUTS, _lib.F_KFF, _lib.F_KFB, _lib.F_COST)]
        self.iterations_, self.status_, self.trace_ = (ws.download(f) for f in (_lib.F_ITERS, _lib.F_STATUS, _lib.F_TRACE))
        if not self.batched_:
            if self.status_[0] == 3:
                raise np.linalg.LinAlgError('Singular matrix')
            out = [o[0] for o in out]
            self.iterations_, self.status_, self.trace_ = self.iterations_[0], self.status_[0], self.trace_[0]
        self.states_, self.inputs_, self.k_feedforward_, self.K_feedback_ = out[:4]
        return tuple(out)

    def close(self):
        if self._ws is not None:
            self._ws.close()
            self._ws = None
