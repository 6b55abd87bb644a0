import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'drone-sim-python_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gold():
    def load(name):
        return dict(np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False))
    return load
