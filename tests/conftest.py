import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def oracle64():
    from oracle import Oracle
    import synth
    o = Oracle("f64")
    o.set_weights(synth.synth_state_dict(1234))
    return o


@pytest.fixture(scope="session")
def oracle32():
    from oracle import Oracle
    import synth
    o = Oracle("f32")
    o.set_weights(synth.synth_state_dict(1234))
    return o
