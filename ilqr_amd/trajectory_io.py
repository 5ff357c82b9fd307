"""Trajectory files in the reference's own formats, so batched results can feed the
reference's tracking / animation scripts and its saved trajectories can be loaded here.

* CSV pair, as written by python/example_acrobot_swingup.py:56-57 (``np.savetxt(name, array, delimiter=',')``, i.e.
  ``%.18e`` rows: exact float64 round trip) and read back by python/example_acrobot_swingup_ltvlqr_tracking.py:26-28
  (``np.loadtxt(..., delimiter=',')``, inputs reshaped to ``(N, 1)``): ``<stem>_states.csv`` ``(N+1, n)``,
  ``<stem>_inputs.csv`` ``(N, m)``.
* MATLAB ``.mat`` with the field set of matlab/smooth-ilqr/example_pendulum.m:39, example_acrobot.m:44,
  example_cartpole.m:43: ``states (N+1,n)``, ``inputs (N,m)``, ``dt (1,1)``, ``parameters (1,P)``,
  ``K_feedback (N,m,n)``, optional ``k_feedforward (N,m)``, ``target_state (n,1)`` (shapes as found in the
  reference's committed ``*-trajectory.mat`` files).

Host-side numpy / scipy only; nothing here touches the GPU.
"""
import numpy as np


def _stem(stem, index):
    return stem if index is None else '%s_%05d' % (stem, index)


def save_trajectory_csv(stem, states, inputs):
    """Write ``<stem>_states.csv`` and ``<stem>_inputs.csv``; a batch ``(B, ...)`` writes ``<stem>_00000_...`` etc.
    Returns the list of (states_path, inputs_path)."""
    states, inputs = np.asarray(states, dtype=np.float64), np.asarray(inputs, dtype=np.float64)
    if states.ndim == 2:
        states, inputs, idx = states[None], inputs[None], [None]
    else:
        idx = list(range(states.shape[0]))
    if inputs.shape[0] != states.shape[0]:
        raise ValueError('states and inputs disagree on the batch size')
    paths = []
    for b, i in enumerate(idx):
        ps, pi = _stem(stem, i) + '_states.csv', _stem(stem, i) + '_inputs.csv'
        np.savetxt(ps, states[b], delimiter=',')
        np.savetxt(pi, inputs[b], delimiter=',')
        paths.append((ps, pi))
    return paths


def load_trajectory_csv(stem, index=None):
    """-> ``(states (N+1, n), inputs (N, m))`` from ``<stem>[_<index>]_states.csv`` / ``_inputs.csv``."""
    states = np.loadtxt(_stem(stem, index) + '_states.csv', dtype=float, delimiter=',')
    inputs = np.loadtxt(_stem(stem, index) + '_inputs.csv', dtype=float, delimiter=',')
    states = np.atleast_2d(states)
    if inputs.ndim < 2:                                   # one input: loadtxt returns (N,), the reference reshapes
        inputs = inputs.reshape((inputs.shape[0] if inputs.ndim else 1, 1))
    return states, inputs


def save_trajectory_mat(path, states, inputs, dt, parameters, K_feedback=None, k_feedforward=None, target_state=None):
    """One problem per file, the reference's field names and shapes."""
    import scipy.io
    states, inputs = np.asarray(states, dtype=np.float64), np.asarray(inputs, dtype=np.float64)
    if states.ndim != 2 or inputs.ndim != 2:
        raise ValueError('save_trajectory_mat writes one problem per file: states (N+1,n), inputs (N,m)')
    out = {'states': states, 'inputs': inputs, 'dt': np.array([[float(dt)]]),
           'parameters': np.asarray(parameters, dtype=np.float64).reshape(1, -1)}
    if K_feedback is not None:
        K = np.asarray(K_feedback, dtype=np.float64)
        if K.shape != (inputs.shape[0], inputs.shape[1], states.shape[1]):
            raise ValueError('K_feedback must be (N, m, n)')
        out['K_feedback'] = K
    if k_feedforward is not None:
        out['k_feedforward'] = np.asarray(k_feedforward, dtype=np.float64).reshape(inputs.shape)
    if target_state is not None:
        out['target_state'] = np.asarray(target_state, dtype=np.float64).reshape(-1, 1)
    scipy.io.savemat(path, out)
    return path


def load_trajectory_mat(path):
    """-> dict with python-side shapes: ``dt`` float, ``parameters (P,)``, ``target_state (n,)``, arrays as stored.
    Works on the reference's own ``*-trajectory.mat`` files (extra fields such as ``modes`` are passed through)."""
    import scipy.io
    d = scipy.io.loadmat(path)
    out = {}
    for k, v in d.items():
        if k.startswith('__'):
            continue
        if k == 'dt':
            out[k] = float(np.asarray(v).reshape(-1)[0])
        elif k in ('parameters', 'target_state'):
            out[k] = np.asarray(v, dtype=np.float64).reshape(-1)
        elif k in ('states', 'inputs', 'K_feedback', 'k_feedforward'):
            out[k] = np.ascontiguousarray(v, dtype=np.float64)
        else:
            out[k] = v
    if 'inputs' in out and out['inputs'].ndim == 1:
        out['inputs'] = out['inputs'].reshape(-1, 1)
    return out
