import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with g++)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pkg():
    import orb_slam3_rust_amd as P
    return P


@pytest.fixture(scope="session")
def gpu_handle(pkg):
    """A handle on cuda:0 through the C ABI.  GPU tests only; fails loudly without the library."""
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 2000, device=0, max_w=1920, max_h=1080,
                   max_batch=8)
    yield h
    h.close()


def records_equal(a, b):
    """bit-exact comparison of two structured arrays (NaN-safe: compares raw bytes)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()
