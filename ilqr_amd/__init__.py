"""ilqr_amd -- batched iLQR / LTV-LQR for AMD MI355X behind the python API of Kongx231/ilqr.

    from ilqr_amd import ilqr, ltvlqr, symbolic_dynamics_acrobot

``ilqr`` and ``ltvlqr`` keep the reference's signatures (python/ilqr.py:5, python/ltvlqr.py:3); the dynamics
triples come from ``symbolic_dynamics_<system>()`` (python/symbolic_dynamics.py:5,28); ``ilqr_mpc`` is the MATLAB tree's
receding-horizon tracking solver (matlab/smooth-ilqr/ilqr_mpc.m).  Everything is computed by
hand-written HIP kernels for gfx950 in ``libilqr_hip.so`` (C ABI: include/ilqr_hip.h); importing this package
does not need a GPU, calling it does.
"""
from . import _lib
from ._lib import IlqrHipError, Workspace, lib
from .ilqr import ilqr
from .ilqr_mpc import ilqr_mpc
from .ltvlqr import ltvlqr
from .tracking import tracking_rollout
from .trajectory_io import load_trajectory_csv, load_trajectory_mat, save_trajectory_csv, save_trajectory_mat
from .symbolic_dynamics import (CompiledDynamics, symbolic_dynamics_acrobot, symbolic_dynamics_car,
                                symbolic_dynamics_cartpole, symbolic_dynamics_double_integrator,
                                symbolic_dynamics_double_pendulum, symbolic_dynamics_floating_base,
                                symbolic_dynamics_pendulum,
                                symbolic_dynamics_quadcopter)

__all__ = ['ilqr', 'ilqr_mpc', 'ltvlqr', 'tracking_rollout', 'save_trajectory_csv', 'load_trajectory_csv',
           'save_trajectory_mat', 'load_trajectory_mat', 'symbolic_dynamics_pendulum', 'symbolic_dynamics_acrobot', 'symbolic_dynamics_cartpole',
           'symbolic_dynamics_quadcopter', 'symbolic_dynamics_car', 'symbolic_dynamics_double_integrator',
           'symbolic_dynamics_double_pendulum', 'symbolic_dynamics_floating_base', 'CompiledDynamics', 'Workspace', 'IlqrHipError', 'lib']
