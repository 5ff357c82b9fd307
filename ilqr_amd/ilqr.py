"""Drop-in for the reference's ``ilqr`` class (python/ilqr.py), computing on an AMD MI355X through libilqr_hip.so.

Same constructor (14 positional arguments, ilqr.py:5), same methods and return values, same public
attributes (trailing underscore), same progress prints for a single problem.  Additive extension: a leading
batch axis -- ``init_state (B,n)``, ``initial_guess (B,N,m)`` and optionally ``target_state (B,n)``,
``parameters (B,P)`` -- solves B independent problems at once; every output gains the leading ``B`` axis,
``current_cost`` becomes ``(B,)`` and ``iterations_`` / ``status_`` report each problem's exit.

All numerics happen in the HIP kernels (ilqr_amd/csrc/ilqr_hip.hip); there is no CPU path here.
"""
import numpy as np

from . import _lib
from .symbolic_dynamics import system_of


class ilqr:
    def __init__(self, init_state, target_state, initial_guess, dt, start_time, end_time, f_disc, A, B, Q_k, R_k, Q_T,
                 parameters, n_iterations, device=0, n_alpha_parallel=0, verbose=None, zero_first_state=True, pipeline=0, n_streams=0,
                 n_devices=1, overlap=0):
        self.init_state_ = init_state
        self.target_state_ = target_state
        self.inputs_ = initial_guess
        self.batched_ = np.ndim(initial_guess) == 3
        self.n_states_ = np.shape(init_state)[-1]                 # ilqr.py:9
        self.n_inputs_ = np.shape(initial_guess)[-1]              # ilqr.py:10
        self.batch_ = np.shape(initial_guess)[0] if self.batched_ else 1

        self.dt_ = dt
        self.start_time_ = start_time
        self.end_time_ = end_time
        self.time_span_ = np.arange(start_time, end_time, dt).flatten()   # ilqr.py:15
        self.n_timesteps_ = np.shape(self.time_span_)[0]                  # ilqr.py:16
        # Dynamics
        self.f_ = f_disc
        self.A_ = A
        self.B_ = B
        self.system_ = system_of(f_disc, A, B)
        # Weighting
        self.Q_k_ = Q_k
        self.R_k_ = R_k
        self.Q_T_ = Q_T
        self.parameters_ = parameters
        # Max iterations
        self.n_iterations_ = n_iterations

        # device-side options (not in the reference)
        self.device_ = device
        self.n_alpha_parallel_ = n_alpha_parallel
        self.zero_first_state_ = zero_first_state
        self.pipeline_ = pipeline              # 0 auto, 1 fused, 2 split kernel schedule (same results)
        self.n_streams_ = n_streams            # independent sub-batches of solve(), one HIP stream each
        # GPUs the batch is split over by problem index (device, device + 1, ...; 0 = every visible GPU): one host
        # thread per GPU inside the library, no exchange between them -- the problems are independent
        self.n_devices_ = n_devices
        # 0 auto | 1 off | 2 on: accept step of iteration i alongside the recursion of iteration i + 1 on two HIP streams (small
        # batches of 4-state systems).  Pass 1 when OTHER PROCESSES use the same GPU.  Same results.
        self.overlap_ = overlap
        self.verbose_ = (not self.batched_) if verbose is None else verbose

        n, m, npar = _lib.system_dims(self.system_)
        if self.n_states_ != n or self.n_inputs_ != m:
            raise ValueError('%s has %d states and %d inputs; got init_state with %d and initial_guess with %d'
                             % (self.system_, n, m, self.n_states_, self.n_inputs_))
        if np.shape(initial_guess)[-2] != self.n_timesteps_:
            raise ValueError('initial_guess has %d rows but np.arange(start_time, end_time, dt) has %d steps'
                             % (np.shape(initial_guess)[-2], self.n_timesteps_))
        # Every array is handed to C as a raw pointer: shapes are settled here.  With a batch, init_state is (B,n) -- or
        # (n,), one initial state shared by the B guesses --, target_state (n,) or (B,n), parameters (P,) or (B,P).
        Bn = self.batch_
        init_shape, tgt_shape = np.shape(init_state), np.shape(target_state)
        par_shape = np.shape(_lib.params_of(self.system_, parameters))
        if np.ndim(initial_guess) not in (2, 3):
            raise ValueError('initial_guess must be (N,m) or (B,N,m), got %s' % (np.shape(initial_guess),))
        if init_shape not in (((n,), (Bn, n)) if self.batched_ else ((n,),)):
            raise ValueError('init_state must be (%d,)%s, got %s' % (n, ' or (%d,%d)' % (Bn, n) if self.batched_ else '', init_shape))
        if tgt_shape not in (((n,), (Bn, n)) if self.batched_ else ((n,),)):
            raise ValueError('target_state must be (%d,)%s, got %s' % (n, ' or (%d,%d)' % (Bn, n) if self.batched_ else '', tgt_shape))
        if par_shape not in (((npar,), (Bn, npar)) if self.batched_ else ((npar,),)):
            raise ValueError('%s takes parameters of shape (%d,)%s, got %s'
                             % (self.system_, npar, ' or (%d,%d)' % (Bn, npar) if self.batched_ else '', par_shape))
        for name, w, k_ in (('Q_k', Q_k, n), ('R_k', R_k, m), ('Q_T', Q_T, n)):
            if np.size(w) != k_ * k_:
                raise ValueError('%s must be (%d,%d), got %s' % (name, k_, k_, np.shape(w)))

    # ------------------------------------------------------------------ helpers
    def _b(self, a, tail_shape, name='array'):
        """float64 contiguous, shape (B,) + tail_shape: the C side reads exactly that many doubles."""
        a = _lib.f64(a)
        if a.shape == tuple(tail_shape) and (not self.batched_ or len(tail_shape) == 1):
            # a single problem, or one vector shared by the whole batch (init_state (n,) with B guesses)
            a = np.ascontiguousarray(np.broadcast_to(a, (self.batch_,) + tuple(tail_shape)))
        if a.shape != (self.batch_,) + tuple(tail_shape):
            raise ValueError('%s must have shape %s%s, got %s' % (name, (self.batch_,) + tuple(tail_shape) if self.batched_ else tuple(tail_shape),
                                                                   '' if self.batched_ else ' (single problem)', a.shape))
        return a

    def _unb(self, a):
        return a if self.batched_ else a[0]

    def _weights(self):
        n, m = self.n_states_, self.n_inputs_
        return (_lib.f64(self.Q_k_).reshape(n, n), _lib.f64(self.R_k_).reshape(m, m), _lib.f64(self.Q_T_).reshape(n, n))

    def _target(self):
        t = _lib.f64(self.target_state_)
        return (t, True) if t.ndim == 2 else (t.reshape(-1), False)

    def _params(self):
        p = _lib.params_of(self.system_, self.parameters_)
        return (p, True) if p.ndim == 2 else (p.reshape(-1), False)

    def _desc(self, n_iterations=0):
        _lib.require_gpu()
        _, tb = self._target()
        _, pb = self._params()
        n_dev = self.n_devices_ if self.n_devices_ else _lib.lib().ilqr_device_count()
        return _lib.make_desc(self.system_, self.n_timesteps_, self.batch_, self.dt_, n_iterations=n_iterations,
                              n_alpha_parallel=self.n_alpha_parallel_, zero_first_state=self.zero_first_state_,
                              device=self.device_, target_batched=tb, params_batched=pb, pipeline=self.pipeline_, n_streams=self.n_streams_,
                              n_devices=n_dev, overlap=self.overlap_)

    # ------------------------------------------------------------------ ilqr.py:30-47
    def rollout(self):
        Bn, N, n, m = self.batch_, self.n_timesteps_, self.n_states_, self.n_inputs_
        init = self._b(self.init_state_, (n,), 'init_state')
        u_in = self._b(self.inputs_, (N, m), 'inputs_')
        params, _ = self._params()
        states = np.empty((Bn, N + 1, n))
        inputs = np.empty((Bn, N, m))
        _lib.check(_lib.lib().ilqr_rollout_batched(self._desc(), _lib.dptr(init), _lib.dptr(u_in), _lib.dptr(params),
                                                   _lib.dptr(states), _lib.dptr(inputs)))
        self.states_ = self._unb(states)
        self.inputs_ = self._unb(inputs)
        return self.states_, self.inputs_

    # ------------------------------------------------------------------ ilqr.py:49-62
    def compute_cost(self, states, inputs):
        N, n, m = self.n_timesteps_, self.n_states_, self.n_inputs_
        states = self._b(states, (N + 1, n), 'states')
        inputs = self._b(inputs, (N, m), 'inputs')
        _, Rk, QT = self._weights()
        target, _ = self._target()
        cost = np.empty(self.batch_)
        _lib.check(_lib.lib().ilqr_cost_batched(self._desc(), _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(Rk),
                                                _lib.dptr(QT), _lib.dptr(target), _lib.dptr(cost)))
        return cost if self.batched_ else float(cost[0])

    # ------------------------------------------------------------------ ilqr.py:64-135
    def backwards_pass(self):
        Bn, N, n, m = self.batch_, self.n_timesteps_, self.n_states_, self.n_inputs_
        states = self._b(self.states_, (N + 1, n), 'states_')
        inputs = self._b(self.inputs_, (N, m), 'inputs_')
        Qk, Rk, QT = self._weights()
        target, _ = self._target()
        params, _ = self._params()
        k = np.empty((Bn, N, m))
        K = np.empty((Bn, N, m, n))
        dJ = np.empty((Bn, 3))
        status = np.empty(Bn, dtype=np.int32)
        _lib.check(_lib.lib().ilqr_backward_batched(self._desc(), _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(target),
                                                    _lib.dptr(Qk), _lib.dptr(Rk), _lib.dptr(QT), _lib.dptr(params),
                                                    _lib.dptr(k), _lib.dptr(K), _lib.dptr(dJ), _lib.iptr(status)))
        if not self.batched_ and status[0] == 3:
            raise np.linalg.LinAlgError('Singular matrix')       # what np.linalg.inv raises at ilqr.py:106
        self.expected_cost_reduction_grad_ = self._unb(dJ[:, 1])
        self.expected_cost_reduction_hess_ = self._unb(dJ[:, 2])
        self.expected_cost_reduction_ = self._unb(dJ[:, 0])
        self.k_feedforward_ = self._unb(k)
        self.K_feedback_ = self._unb(K)
        return (self.k_feedforward_, self.K_feedback_, self.expected_cost_reduction_)

    # ------------------------------------------------------------------ ilqr.py:137-158
    def forwards_pass(self, learning_rate):
        Bn, N, n, m = self.batch_, self.n_timesteps_, self.n_states_, self.n_inputs_
        init = self._b(self.init_state_, (n,), 'init_state')
        ref_states = self._b(self.states_, (N + 1, n), 'states_')
        ref_inputs = self._b(self.inputs_, (N, m), 'inputs_')
        k = self._b(self.k_feedforward_, (N, m), 'k_feedforward_')
        K = self._b(self.K_feedback_, (N, m, n), 'K_feedback_')
        params, _ = self._params()
        lr = np.ascontiguousarray(np.broadcast_to(np.asarray(learning_rate, dtype=np.float64), (Bn,)))
        states = np.empty((Bn, N + 1, n))
        inputs = np.empty((Bn, N, m))
        _lib.check(_lib.lib().ilqr_forward_batched(self._desc(), _lib.dptr(init), _lib.dptr(ref_states),
                                                   _lib.dptr(ref_inputs), _lib.dptr(k), _lib.dptr(K), _lib.dptr(lr),
                                                   _lib.dptr(params), _lib.dptr(states), _lib.dptr(inputs), None, None,
                                                   None, None))
        return (self._unb(states), self._unb(inputs))

    # ------------------------------------------------------------------ ilqr.py:159-211
    def solve(self):
        if self.n_iterations_ < 1:
            # the reference falls off the loop with k_feedforward unbound (ilqr.py:211)
            raise UnboundLocalError("local variable 'k_feedforward' referenced before assignment")
        Bn, N, n, m = self.batch_, self.n_timesteps_, self.n_states_, self.n_inputs_
        init = self._b(self.init_state_, (n,), 'init_state')
        guess = self._b(self.inputs_, (N, m), 'inputs_')
        Qk, Rk, QT = self._weights()
        target, _ = self._target()
        params, _ = self._params()
        states = np.empty((Bn, N + 1, n))
        inputs = np.empty((Bn, N, m))
        k = np.empty((Bn, N, m))
        K = np.empty((Bn, N, m, n))
        cost = np.empty(Bn)
        iters = np.empty(Bn, dtype=np.int32)
        status = np.empty(Bn, dtype=np.int32)
        trace = np.empty((Bn, self.n_iterations_, 6))
        _lib.check(_lib.lib().ilqr_solve_batched(self._desc(self.n_iterations_), _lib.dptr(init), _lib.dptr(target),
                                                 _lib.dptr(guess), _lib.dptr(Qk), _lib.dptr(Rk), _lib.dptr(QT),
                                                 _lib.dptr(params), _lib.dptr(states), _lib.dptr(inputs), _lib.dptr(k),
                                                 _lib.dptr(K), _lib.dptr(cost), _lib.iptr(iters), _lib.iptr(status),
                                                 _lib.dptr(trace)))
        if not self.batched_ and status[0] == 3:
            raise np.linalg.LinAlgError('Singular matrix')
        self.states_ = self._unb(states)
        self.inputs_ = self._unb(inputs)
        self.k_feedforward_ = self._unb(k)
        self.K_feedback_ = self._unb(K)
        self.iterations_ = self._unb(iters)
        self.status_ = self._unb(status)
        self.trace_ = self._unb(trace)
        last = np.maximum(iters - 1, 0)
        rows = trace[np.arange(Bn), last]
        self.expected_cost_reduction_ = self._unb(rows[:, 1])
        self.expected_cost_reduction_grad_ = self._unb(rows[:, 2])
        self.expected_cost_reduction_hess_ = self._unb(rows[:, 3])
        if self.verbose_:
            self._print_progress(trace, iters, status)
        current_cost = cost if self.batched_ else cost[0]
        return self.states_, self.inputs_, self.k_feedforward_, self.K_feedback_, current_cost

    def _print_progress(self, trace, iters, status):
        """The reference's prints (ilqr.py:171,174,178,206), replayed from the device-side trace of problem 0."""
        for ii in range(int(iters[0])):
            print('Starting iteration: ', ii, ', Current cost: ', trace[0, ii, 0])
            print('Expected cost reduction: ', trace[0, ii, 1])
        if status[0] == 1:
            print("Stopping optimization, optimal trajectory")
        elif status[0] == 2:
            print("Stopping optimization, low learning rate")
