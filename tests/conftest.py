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


# Two-rank rehearsal of the multi-GPU path with the REAL HIP plan (tests/test_gpu_dist.py): bench.py --gpus 2 with the gloo
# backend, both ranks on cuda:0.  The launcher re-executes python, which a process that has initialised the GPU must not do on
# this pool -- so the child is started here, at session start, before any test (or fixture) of this process touches the GPU,
# and only when GPU tests are selected on a box that has a GPU (torch.cuda.device_count() does not initialise it).
REHEARSAL = {}


def pytest_sessionstart(session):
    expr = session.config.getoption('markexpr', '') or ''
    if 'gpu' not in expr or 'not gpu' in expr:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import subprocess
    env = dict(os.environ, D2D_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config3-batch', '0', '--no-sim',
           '--no-nlp', '--no-groups', '--no-cpu-baseline', '--no-extra-modes', '--dump-costs', os.path.join(ROOT, 'gpurun_out', 'rehearsal')]
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        REHEARSAL.update(rc=r.returncode, out=r.stdout, err=r.stderr[-4000:])
    except Exception as e:       # noqa: BLE001
        REHEARSAL.update(rc=-1, out='', err=repr(e))


@pytest.fixture(scope='session')
def rehearsal():
    return REHEARSAL


@pytest.fixture(scope='session')
def gold():
    def load(name):
        return dict(np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False))
    return load
