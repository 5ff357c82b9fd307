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


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
