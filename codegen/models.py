"""Symbolic models for the systems of the hot path (own sympy code).

Each builder returns a ``Model`` whose ``f_disc``/``A_disc``/``B_disc`` are the
forward-Euler discretised dynamics and their Jacobians, exactly the triple the
reference lambdifies:

* pendulum   -- /root/reference/python/symbolic_dynamics.py:5-26  (the reference
  text has the typo ``sp.sp.sin`` at :12, so it cannot run; the expression
  here is the same one written with ``sp.sin``)
* acrobot    -- /root/reference/python/symbolic_dynamics.py:28-92  (same
  construction sequence -- simplify(M), Christoffel loop with python-float
  ``1/2``, ``M.inv()`` -- so sympy produces the same expression tree and the
  same 15-digit literals when printed)
* cartpole   -- spec: matlab/smooth-ilqr/symbolic_dynamics_cartpole.m:2-22
* quadcopter -- spec: matlab/smooth-ilqr/symbolic_dynamics_quadcopter.m:3-51
* car, double_integrator, double_pendulum, floating_base -- specs: matlab/smooth-ilqr/symbolic_dynamics_{car,
  double_integrator,double_pendulum,floating_base}.m

The MATLAB files are specifications only (read as text; nothing here executes
them).  This module needs sympy and is used by ``codegen/gen_dynamics.py`` and
``tools/make_golden.py``; nothing on the GPU box imports it.
"""
from dataclasses import dataclass

import sympy as sp
from sympy.matrices import Matrix


@dataclass
class Model:
    name: str
    states: Matrix
    inputs: Matrix
    dt: sp.Symbol
    parameters: Matrix
    f_disc: Matrix
    A_disc: Matrix
    B_disc: Matrix

    @property
    def n(self):
        return self.states.shape[0]

    @property
    def m(self):
        return self.inputs.shape[0]

    @property
    def n_params(self):
        return self.parameters.shape[0]

    def lambdify(self):
        """The (f, A, B) numpy callables the reference would build (symbolic_dynamics.py:23-25)."""
        args = (self.states, self.inputs, self.dt, self.parameters)
        return (sp.lambdify(args, self.f_disc),
                sp.lambdify(args, self.A_disc),
                sp.lambdify(args, self.B_disc))


def _finish(name, states, inputs, dt, parameters, f):
    f_disc = states + f * dt
    A_disc = f_disc.jacobian(states)
    B_disc = f_disc.jacobian(inputs)
    return Model(name, states, inputs, dt, parameters, f_disc, A_disc, B_disc)


def pendulum():
    m, g, L, theta, theta_dot, u, dt = sp.symbols('m g L theta theta_dot u dt')
    inputs = Matrix([u])
    states = Matrix([theta, theta_dot])
    f = Matrix([theta_dot, (u - m * g * L * sp.sin(theta)) / (m * L * L)])
    return _finish('pendulum', states, inputs, dt, Matrix([m, g, L]), f)


def _two_link_arm(name, both_joints_actuated):
    """The planar two-link arm of symbolic_dynamics.py:31-77 (manipulator equation M(q) q_ddot + C q_dot + N = Y through body
    Jacobians and Christoffel symbols), with the torque on the second joint only (the reference's acrobot) or on both
    joints (matlab/smooth-ilqr/symbolic_dynamics_double_pendulum.m:3-58).  The reference returns only lambdified callables,
    so reproducing its expression tree -- which the generated code must match literal for literal -- means building the
    expressions in the same sequence."""
    q1, q2, q1_dot, q2_dot = sp.symbols('q1 q2 q1_dot q2_dot')
    tau_1, tau_2 = sp.symbols('tau_1 tau_2')
    m1, L1, m2, L2, g, dt = sp.symbols('m1 L1 m2 L2 g dt ')
    q = Matrix([[q1], [q2]])
    q_dot = Matrix([[q1_dot], [q2_dot]])
    tau = Matrix([[tau_1 if both_joints_actuated else 0], [tau_2]])
    states = Matrix([[q], [q_dot]])
    inputs = Matrix([tau_1, tau_2]) if both_joints_actuated else Matrix([tau_2])
    # generalised inertia of each link (python float 1/12 on purpose: it becomes a sympy Float)
    M1 = Matrix([[m1, 0, 0], [0, m1, 0], [0, 0, 1 / 12 * m1 * L1 * L1]])
    M2 = Matrix([[m2, 0, 0], [0, m2, 0], [0, 0, 1 / 12 * m2 * L2 * L2]])
    # body Jacobians
    Jb_sL1 = Matrix([[0, 0], [L1 / 2, 0], [1, 0]])
    Jb_sL2 = Matrix([[L1 * sp.sin(q2), 0], [L1 * sp.cos(q2) + L2 / 2, L2 / 2], [1, 1]])
    M = sp.simplify(Jb_sL1.T @ M1 @ Jb_sL1 + Jb_sL2.T @ M2 @ Jb_sL2)
    nq = 2
    C = sp.zeros(nq, nq)
    for ii in range(nq):
        for jj in range(nq):
            for kk in range(nq):
                C[ii, jj] = C[ii, jj] + 1 / 2 * (sp.diff(M[ii, jj], q[kk]) + sp.diff(M[ii, kk], q[jj])
                                                 - sp.diff(M[jj, kk], q[ii])) * q_dot[kk]
    V = Matrix([(m1 * g * L1 / 2 * sp.sin(q1) + m2 * g * (L1 * sp.sin(q1) + L2 / 2 * sp.sin(q1 + q2)))])
    N = V.jacobian(q).T
    Y = tau
    f = Matrix([q_dot, M.inv() @ (Y - N - C @ q_dot)])
    return _finish(name, states, inputs, dt, Matrix([m1, L1, m2, L2, g]), f)


def acrobot():
    return _two_link_arm('acrobot', both_joints_actuated=False)


def cartpole():
    mc, mp, ll, g, x, x_dot, theta, theta_dot, u, dt = sp.symbols('mc mp ll g x x_dot theta theta_dot u dt', real=True)
    inputs = Matrix([u])
    states = Matrix([x, theta, x_dot, theta_dot])
    s, c = sp.sin(theta), sp.cos(theta)
    f = Matrix([
        x_dot,
        theta_dot,
        1 / (mc + mp * s**2) * (u + mp * s * (ll * theta_dot**2 + g * c)),
        1 / (ll * (mc + mp * s**2)) * (-u * c - mp * ll * theta_dot**2 * c * s - (mc + mp) * g * s),
    ])
    return _finish('cartpole', states, inputs, dt, Matrix([mc, mp, ll, g]), f)


def quadcopter():
    px, py, pz, u, v, w, phi, theta, psi, p, q, r, dt = sp.symbols('px py pz u v w phi theta psi p q r dt', real=True)
    delta_l, delta_r, delta_b, delta_f = sp.symbols('delta_l delta_r delta_b delta_f', real=True)
    J_x, J_y, J_z, m, g, ell, k_1, k_2 = sp.symbols('J_x J_y J_z m g ell k_1 k_2', real=True)
    sin, cos, tan = sp.sin, sp.cos, sp.tan
    F_l, F_r, F_b, F_f = k_1 * delta_l, k_1 * delta_r, k_1 * delta_b, k_1 * delta_f
    tau_l, tau_r, tau_b, tau_f = k_2 * delta_l, k_2 * delta_r, k_2 * delta_b, k_2 * delta_f
    F = F_f + F_r + F_b + F_l
    tau_phi = ell * (F_l - F_r)
    tau_theta = ell * (F_f - F_b)
    tau_psi = -tau_f + tau_r - tau_b + tau_l
    R = Matrix([
        [cos(theta) * cos(psi), sin(phi) * sin(theta) * cos(psi) - cos(phi) * sin(psi),
         cos(phi) * sin(theta) * cos(psi) + sin(phi) * sin(psi)],
        [cos(theta) * sin(psi), sin(phi) * sin(theta) * sin(psi) + cos(phi) * cos(psi),
         cos(phi) * sin(theta) * sin(psi) - sin(phi) * cos(psi)],
        [sin(theta), -sin(phi) * cos(theta), -cos(phi) * cos(theta)]])
    trans_vel = R * Matrix([u, v, w])
    trans_vel[2] = -trans_vel[2]
    trans_accel = (Matrix([r * v - q * w, p * w - r * u, q * u - p * v])
                   + Matrix([-g * sin(theta), g * cos(theta) * sin(phi), g * cos(theta) * cos(phi)])
                   + (1 / m) * Matrix([0, 0, -F]))
    angular_vel = Matrix([[1, sin(phi) * tan(theta), cos(phi) * tan(theta)],
                          [0, cos(phi), -sin(phi)],
                          [0, sin(phi) / cos(theta), cos(phi) / cos(theta)]]) * Matrix([p, q, r])
    angular_accel = (Matrix([(J_y - J_z) / J_x * q * r, (J_z - J_x) / J_y * p * r, (J_x - J_y) / J_z * p * q])
                     + Matrix([1 / J_x * tau_phi, 1 / J_y * tau_theta, 1 / J_z * tau_psi]))
    states = Matrix([px, py, pz, phi, theta, psi, u, v, w, p, q, r])
    inputs = Matrix([delta_f, delta_r, delta_b, delta_l])
    f = Matrix([trans_vel, angular_vel, trans_accel, angular_accel])
    return _finish('quadcopter', states, inputs, dt, Matrix([J_x, J_y, J_z, m, g, ell, k_1, k_2]), f)


def car():
    """spec: matlab/smooth-ilqr/symbolic_dynamics_car.m:2-20 (unicycle: position, speed, heading; no physical parameters).
    The MATLAB model has an EMPTY parameter vector; the compiled systems keep one unused slot (`unused`) so that every
    array has a size -- the python drop-in pads an empty `parameters` argument."""
    px, py, v, theta, v_dot, theta_dot, dt, unused = sp.symbols('px py v theta v_dot theta_dot dt unused')
    states = Matrix([px, py, v, theta])
    inputs = Matrix([v_dot, theta_dot])
    f = Matrix([v * sp.cos(theta), v * sp.sin(theta), v_dot, theta_dot])
    return _finish('car', states, inputs, dt, Matrix([unused]), f)


def double_integrator():
    """spec: matlab/smooth-ilqr/symbolic_dynamics_double_integrator.m:3-19"""
    x, x_dot, force, m, dt = sp.symbols('x x_dot force m dt', real=True)
    states = Matrix([x, x_dot])
    inputs = Matrix([force])
    f = Matrix([x_dot, force / m])
    return _finish('double_integrator', states, inputs, dt, Matrix([m]), f)


def double_pendulum():
    """spec: matlab/smooth-ilqr/symbolic_dynamics_double_pendulum.m:3-58 -- the acrobot's two links with BOTH joints
    actuated (tau = [tau_1; tau_2])"""
    return _two_link_arm('double_pendulum', both_joints_actuated=True)


def floating_base():
    """spec: matlab/smooth-ilqr/symbolic_dynamics_floating_base.m:2-92 -- a single rigid body (quadruped trunk) driven by the
    ground-reaction forces of four feet: 12 states [position, roll-pitch-yaw, linear velocity, world angular velocity],
    12 inputs [F_1; F_2; F_3; F_4], 17 parameters [I_xx I_yy I_zz m g, foot positions 1..4] (:81-85).  The same sequence as
    the .m file: world inertia rot' I rot and its simplified inverse (:55-57), body rates from the simplified inverse of the
    rate map (:60-64), angular acceleration from the feet's wrenches (:67, skew.m), gravity added as [0;0;g] (:70-72)."""
    x, y, z, x_dot, y_dot, z_dot, phi, theta, psi = sp.symbols('x y z x_dot y_dot z_dot phi theta psi', real=True)
    phi_dot_world, theta_dot_world, psi_dot_world, dt = sp.symbols('phi_dot_world theta_dot_world psi_dot_world dt', real=True)
    feet = [Matrix(sp.symbols('foot_pos_%dx foot_pos_%dy foot_pos_%dz' % (i, i, i), real=True)) for i in (1, 2, 3, 4)]
    F = [Matrix(sp.symbols('F_%dx F_%dy F_%dz' % (i, i, i), real=True)) for i in (1, 2, 3, 4)]
    I_xx, I_yy, I_zz, m, g = sp.symbols('I_xx I_yy I_zz m g', real=True)
    robot_pos = Matrix([x, y, z])
    robot_orientation = Matrix([phi, theta, psi])
    robot_lin_vel = Matrix([x_dot, y_dot, z_dot])
    robot_ang_vel_world = Matrix([phi_dot_world, theta_dot_world, psi_dot_world])
    states = Matrix([robot_pos, robot_orientation, robot_lin_vel, robot_ang_vel_world])
    r = [robot_pos - fp for fp in feet]
    inputs = Matrix([F[0], F[1], F[2], F[3]])
    sin, cos = sp.sin, sp.cos
    rot_roll = Matrix([[1, 0, 0], [0, cos(phi), -sin(phi)], [0, sin(phi), cos(phi)]])
    rot_pitch = Matrix([[cos(theta), 0, sin(theta)], [0, 1, 0], [-sin(theta), 0, cos(theta)]])
    rot_yaw = Matrix([[cos(psi), -sin(psi), 0], [sin(psi), cos(psi), 0], [0, 0, 1]])
    rot_mat = rot_yaw * rot_pitch * rot_roll
    I_body = sp.diag(I_xx, I_yy, I_zz)
    # I_hat = rot' I rot and I_hat_inv = simplify(inv(I_hat)) in the .m file (:55-57).  MATLAB's simplifier is not available and
    # sympy's does not finish on the symbolic 3 x 3 inverse within half an hour; rot is orthogonal, so the inverse is written
    # down directly -- the same function, as an expression neither simplifier would need to touch.
    I_hat_inv = rot_mat.T * sp.diag(1 / I_xx, 1 / I_yy, 1 / I_zz) * rot_mat
    rot_mat_vel = Matrix([[cos(theta) * cos(psi), -sin(psi), 0], [cos(theta) * sin(psi), cos(psi), 0], [0, 0, 1]])
    ang_vel_body = sp.simplify(rot_mat_vel.inv() * robot_ang_vel_world)

    def skew(w):
        return Matrix([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])

    ang_accel_world = I_hat_inv * (skew(r[0]) * F[0] + skew(r[1]) * F[1] + skew(r[2]) * F[2] + skew(r[3]) * F[3])   # (:67, unsimplified)
    gravity_vec = Matrix([0, 0, g])
    linear_accel = 1 / m * (F[0] + F[1] + F[2] + F[3]) + gravity_vec
    f = Matrix([robot_lin_vel, ang_vel_body, linear_accel, ang_accel_world])
    parameters = Matrix([I_xx, I_yy, I_zz, m, g] + [c for fp in feet for c in fp])
    return _finish('floating_base', states, inputs, dt, parameters, f)


BUILDERS = {'pendulum': pendulum, 'acrobot': acrobot, 'cartpole': cartpole, 'quadcopter': quadcopter, 'car': car,
            'double_integrator': double_integrator, 'double_pendulum': double_pendulum, 'floating_base': floating_base}
# system ids are part of the C-ABI (include/ilqr_hip.h)
SYSTEM_IDS = {'pendulum': 0, 'acrobot': 1, 'cartpole': 2, 'quadcopter': 3, 'car': 4, 'double_integrator': 5,
              'double_pendulum': 6, 'floating_base': 7}
