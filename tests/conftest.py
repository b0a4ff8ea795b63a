import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Compile the oracle (and the product library when hipcc is available) once per session."""
    import __graft_entry__ as g
    from liorf_amd import s2m
    from oracle import oracle as O
    if not os.path.exists(s2m.LIB_PATH) or not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        g.build()
    O.lib()


@pytest.fixture(scope="session")
def cfg_tiny():
    from liorf_amd import synth
    return synth.make_config("tiny")


@pytest.fixture(scope="session")
def cfg_small():
    from liorf_amd import synth
    return synth.make_config("small")


@pytest.fixture(scope="session")
def cfg_kitti64():
    from liorf_amd import synth
    return synth.make_config("kitti64")


def transform_points(T, p):
    """pointAssociateToMap (reference :302-308) in fp32 with the reference's association order."""
    T = np.asarray(T, np.float32).reshape(3, 4)
    p = np.asarray(p, np.float32)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
    return out
