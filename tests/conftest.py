import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gold_dir():
    return GOLD


def assert_same_trajectory(a, b, lr=1e-4, steps=2, tag=''):
    """Two engines that took the same `steps` Adam steps on the same data: their parameter arenas agree -- up to what the optimiser does with
    gradient elements inside the run-to-run noise of the fp32 split-K atomics.  Adam moves a weight by lr * g / (|g| + 1e-8 ...) in its first
    steps, so an element whose gradient is ~1e-8 (the one-hot conv's columns of rarely hit classes) gets anything between -lr and +lr
    depending on the last bits of g: measured, up to ~200 of 19.4 M elements differ by up to lr between two identical runs on fresh engines
    (5 pairs of 25 at 16 x 128).  What a WRONG trajectory looks like is different in kind: a repeated or skipped step, a wrong step counter or
    1/world factor moves EVERY weight by ~lr.  So: at most 1e-4 of the elements beyond 2e-6, none beyond 2.1 * lr * steps."""
    import torch
    d = (torch.as_tensor(a).detach().double() - torch.as_tensor(b).detach().double()).abs()
    frac = float((d > 2e-6).double().mean())
    assert float(d.max()) <= 2.1 * lr * steps, (tag, float(d.max()))
    assert frac <= 1e-4, (tag, frac, float(d.max()))
