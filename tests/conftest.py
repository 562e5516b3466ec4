import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The extractor describes small calls (a pair, a few pairs) with the per-keypoint kernel and large batches with the tile kernel (bench.py's
# workload).  The suite's calls are small, so the tile form is forced here: every extractor parity test then checks the kernel the headline is
# measured on against the oracle; the per-keypoint form is compared with it bit for bit by test_describe_tile_form_equals_per_keypoint_form
# (child processes with ORBX_DESC_TILE=0 / 1) and runs in smoke() and bench.py's small legs.  Read by the library when it first prepares a geometry.
os.environ.setdefault("ORBX_DESC_TILE", "1")
# Likewise the stereo matcher: calls of fewer pairs than CUs run stereo_bucket / stereo_match / stereo_compact_kernel, large batches the one-launch
# stereo_match_lds_kernel (the pair's right image in LDS).  The suite forces the LDS form (it steps aside by itself where a pair's 52 bytes per
# keypoint + 32 KB exceed 160 KB of LDS); test_stereo_match_lds_form_equals_global_form compares the forms bit for bit in child processes.  Read at
# the first stereo-match call.
os.environ.setdefault("ORBX_SM_LDS", "1")


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


def pose_errors(a, b):
    """Per-keyframe errors between two pose arrays [K,7] (qw,qx,qy,qz,tx,ty,tz): (largest rotation angle of
    q_a * conj(q_b) in radians, largest |dt| / |t_b|).  This is the north star's "poses within 1e-6 relative error"
    taken pose by pose — not a max-abs over a mixed quaternion/translation array, which is an absolute bound on the
    quaternion entries and a window-scale-relative one on the translations."""
    a = np.asarray(a, np.float64).reshape(-1, 7); b = np.asarray(b, np.float64).reshape(-1, 7)
    assert a.shape == b.shape
    if len(a) == 0:
        return 0.0, 0.0
    qa, qb = a[:, :4], b[:, :4] * np.array([1.0, -1.0, -1.0, -1.0])
    w = qa[:, 0] * qb[:, 0] - (qa[:, 1:] * qb[:, 1:]).sum(1)
    v = qa[:, :1] * qb[:, 1:] + qb[:, :1] * qa[:, 1:] + np.cross(qa[:, 1:], qb[:, 1:])
    ang = 2.0 * np.arctan2(np.linalg.norm(v, axis=1), np.abs(w))
    dt = np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1) / np.maximum(np.linalg.norm(b[:, 4:], axis=1), 1e-12)
    return float(ang.max()), float(dt.max())


def point_errors(a, b):
    """Largest per-point |dX| / |X_b| between two point arrays [M,3]."""
    a = np.asarray(a, np.float64).reshape(-1, 3); b = np.asarray(b, np.float64).reshape(-1, 3)
    if len(a) == 0:
        return 0.0
    return float((np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-12)).max())


def assert_ba_close(got, want, tol=1e-6):
    """optimised poses and points of two solves: every keyframe's rotation within `tol` rad and translation within `tol`
    relative, every point within `tol` relative."""
    ang, dt = pose_errors(got["poses_wc"], want["poses_wc"])
    dx = point_errors(got["points"], want["points"])
    assert ang < tol and dt < tol and dx < tol, "rotation %.3e rad, translation %.3e rel, points %.3e rel (tol %.1e)" % (ang, dt, dx, tol)
