import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLD = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def colnorm_err(A, Aref):
    """SURVEY 8c gate L2: per-column max|dA| / max|Aref| (columns span 21 decades)."""
    A = np.asarray(A).reshape(-1, A.shape[-1])
    Aref = np.asarray(Aref).reshape(-1, Aref.shape[-1])
    scale = np.max(np.abs(Aref), axis=0)
    scale[scale == 0] = 1.0
    return np.max(np.abs(A - Aref), axis=0) / scale


def rel(x, y):
    x, y = np.ravel(x), np.ravel(y)
    return float(np.linalg.norm(x - y) / np.linalg.norm(y))
