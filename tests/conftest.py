import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs an AMD GPU (run on the MI355X box with -m gpu)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def rel_err(a, b):
    """max |a-b| / max |b|  (the 1e-8 relative bar of the north star is applied to this)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / den


def elem_err(a, b, floor_frac=1e-3):
    """max over ELEMENTS of |a-b| / max(|b|, floor_frac * max|b|): every entry is judged on its own scale, down to an
    absolute floor (entries smaller than floor_frac of the largest one are rounding residue of sums of larger terms and
    are judged against that floor).  rel_err above is norm-wise and hides errors in the small entries of an array whose
    entries span orders of magnitude (gains K: 1e-1 ... 1e2); this one does not."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), max(floor_frac * float(np.max(np.abs(b))), 1e-300))
    return float(np.max(np.abs(a - b) / den))


# element-wise bar for the gains of one teacher-forced backward pass.  n <= 4: the 1e-8 of the north star.  Quadcopter:
# the CPU oracle -- bit-identical dynamics, LU inverse like numpy -- is itself 4.6e-8 from numpy on this measure (numpy's
# 12 x 12 products go through OpenBLAS dgemm, another summation order), the GPU (Cholesky-based inverse, fused products)
# 1.7e-7; in both cases the difference sits in entries at the floor (|K| ~ 1e-10 where the hover symmetry makes the exact
# value 0: absolute error 1e-9 on gains of size 7, 1.7e-10 norm-wise).  Floating base (12 x 12 Q_uu = 1e-4 I + a rank-6
# term, 100 steps): the same picture -- oracle vs numpy 2.6e-7, in entries that are exactly 0 by the stance's symmetry and
# come out as +-3e-8 on gains of size 110 (3e-10 norm-wise), with opposite signs in the two.
def elem_tol(system):
    return {'quadcopter': 5e-7, 'floating_base': 2e-6}.get(system, 1e-8)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def sample_batch(name):
    """The first problems of one of bench.py's random batches (BASELINE configs[1], [3], [4]) together with what the REFERENCE
    class returned for them (tests/golden/<name>_sample.npz, written by tools/make_golden.py samples): inputs, fixture."""
    import bench
    g = golden(name + '_sample')
    cnt = int(g['count'])
    w = bench.workload(name, bench.DEFAULT_BATCH[name], 0)
    N = w['N']
    assert np.array_equal(w['init'][:cnt], g['init'])                       # the fixture belongs to this workload
    chk = np.array([float(np.sum(w['guess'][b] * np.arange(1, N + 1)[:, None])) for b in range(cnt)])
    assert np.array_equal(chk, g['guess_checksum'])
    return w, g, cnt


def problem_dist(a, b):
    """per problem (leading axis), norm-wise as rel_err"""
    ax = tuple(range(1, np.ndim(a)))
    return np.max(np.abs(a - b), axis=ax) / np.max(np.abs(b), axis=ax) if ax else np.abs(a - b) / np.abs(b)
