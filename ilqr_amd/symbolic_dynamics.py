"""Drop-in for the reference's ``symbolic_dynamics`` module (python/symbolic_dynamics.py).

``symbolic_dynamics_<system>()`` returns ``(f_disc_func, A_disc_func, B_disc_func)`` like the reference
(symbolic_dynamics.py:26, :92).  Instead of sympy-lambdified numpy functions the callables are handles to
the sympy-generated HIP ``__device__`` functions compiled into libilqr_hip.so (codegen/gen_dynamics.py ->
ilqr_amd/csrc/dynamics/<system>.h).  They still evaluate one point when called -- on the GPU, through
``ilqr_dynamics_eval`` -- with the reference's return shapes (n,1) / (n,n) / (n,m), and they carry the
``system`` name so ``ilqr`` / ``ltvlqr`` can run the whole horizon on the device.
"""
import numpy as np

from . import _lib


class CompiledDynamics:
    """One of f_disc / A_disc / B_disc of a compiled system; call as ``fn(states, inputs, dt, parameters)``."""

    def __init__(self, system, which):
        self.system = system
        self.which = which                       # 'f', 'A' or 'B'
        self.n, self.m, self.n_params = _lib.system_dims(system)

    def __call__(self, states, inputs, dt, parameters, device=0):
        x = _lib.f64(states).reshape(-1)
        u = _lib.f64(inputs).reshape(-1)
        p = _lib.params_of(self.system, parameters).reshape(-1)
        if x.size != self.n or u.size != self.m or p.size != self.n_params:
            raise ValueError('%s %s expects %d states, %d inputs, %d parameters'
                             % (self.system, self.which, self.n, self.m, self.n_params))
        out = self.batch(x[None], u[None], dt, p, device=device)[0]
        return out.reshape(self.n, 1) if self.which == 'f' else out

    def batch(self, states, inputs, dt, parameters, device=0):
        """Evaluate at many points: states (count,n), inputs (count,m) -> (count,n) / (count,n,n) / (count,n,m)."""
        _lib.require_gpu()
        x = _lib.f64(states).reshape(-1, self.n)
        u = _lib.f64(inputs).reshape(-1, self.m)
        p = _lib.params_of(self.system, parameters).reshape(-1)
        cnt = x.shape[0]
        shape = {'f': (cnt, self.n), 'A': (cnt, self.n, self.n), 'B': (cnt, self.n, self.m)}[self.which]
        out = np.empty(shape)
        _lib.check(_lib.lib().ilqr_dynamics_eval(_lib.SYSTEM_IDS[self.system], 'fAB'.index(self.which), cnt,
                                                 _lib.dptr(x), _lib.dptr(u), float(dt), _lib.dptr(p), _lib.dptr(out),
                                                 int(device)))
        return out

    def __repr__(self):
        return '<compiled %s_disc of %s (n=%d, m=%d)>' % (self.which, self.system, self.n, self.m)


def _triple(system):
    return tuple(CompiledDynamics(system, w) for w in 'fAB')


def symbolic_dynamics_pendulum():
    """symbolic_dynamics.py:5-26 (parameters [m, g, L])."""
    return _triple('pendulum')


def symbolic_dynamics_acrobot():
    """symbolic_dynamics.py:28-92 (parameters [m1, L1, m2, L2, g])."""
    return _triple('acrobot')


def symbolic_dynamics_cartpole():
    """matlab/smooth-ilqr/symbolic_dynamics_cartpole.m (states [x, theta, x_dot, theta_dot]; parameters [mc, mp, ll, g])."""
    return _triple('cartpole')


def symbolic_dynamics_quadcopter():
    """matlab/smooth-ilqr/symbolic_dynamics_quadcopter.m (12 states, 4 inputs; parameters [J_x J_y J_z m g ell k_1 k_2])."""
    return _triple('quadcopter')


def symbolic_dynamics_car():
    """matlab/smooth-ilqr/symbolic_dynamics_car.m (states [px, py, v, theta]; inputs [v_dot, theta_dot]; the MATLAB model has
    no parameters: pass ``[]`` -- or anything of length <= 1 -- as ``parameters``)."""
    return _triple('car')


def symbolic_dynamics_double_integrator():
    """matlab/smooth-ilqr/symbolic_dynamics_double_integrator.m (states [x, x_dot]; input force; parameters [m])."""
    return _triple('double_integrator')


def symbolic_dynamics_double_pendulum():
    """matlab/smooth-ilqr/symbolic_dynamics_double_pendulum.m (the acrobot with both joints actuated; parameters
    [m1, L1, m2, L2, g])."""
    return _triple('double_pendulum')


def symbolic_dynamics_floating_base():
    """matlab/smooth-ilqr/symbolic_dynamics_floating_base.m (a rigid trunk driven by four feet's ground-reaction forces:
    12 states [position, roll-pitch-yaw, linear velocity, world angular velocity], 12 inputs [F_1; F_2; F_3; F_4],
    parameters [I_xx, I_yy, I_zz, m, g, foot_pos_1 (3), ..., foot_pos_4 (3)])."""
    return _triple('floating_base')


def system_of(*callables):
    """The common compiled system of the given dynamics callables; TypeError for anything else."""
    names = set()
    for fn in callables:
        if not isinstance(fn, CompiledDynamics):
            raise TypeError('dynamics must come from ilqr_amd.symbolic_dynamics.symbolic_dynamics_<system>(): an arbitrary '
                            'python callable cannot run on the GPU and this package has no CPU path (got %r)' % (fn,))
        names.add(fn.system)
    if len(names) != 1:
        raise TypeError('f_disc, A and B belong to different systems: %s' % sorted(names))
    return names.pop()
