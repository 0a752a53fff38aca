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


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` call the HIP library on a device: on a box without one they are skipped with the reason spelled out, so a
    plain `pytest` there is green instead of failing in fixtures.  On a GPU box nothing is skipped: a missing library must fail
    loudly (fastdiff_amd._capi.load raises), never pass by omission."""
    reason = None
    try:
        import torch
        if not torch.cuda.is_available():
            reason = "no HIP device visible (torch.cuda.is_available() is False)"
    except Exception as e:      # noqa: BLE001
        reason = f"torch not importable: {e!r}"
    if reason is None:
        return
    skip = pytest.mark.skip(reason="gpu test: " + reason)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
