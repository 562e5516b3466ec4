"""The extractor / stereo-matcher SPECIFICATION, frozen (VERDICT r1 "Nothing freezes the spec").

tests/golden/extract_frozen.json holds sha256 digests and counts of everything the per-frame path produces
(pyramid levels, blurred levels, FAST candidate sets, keypoints, descriptors, matches, triangulated points) for
five fixed synthetic stereo pairs, written once by scripts/gen_extract_golden.py.  The CPU test pins the oracle to
them, the GPU test pins the HIP path (through the C ABI) to them — neither compares the two implementations with
each other, so they cannot drift together.  Parity with OpenCV itself stays unpinned: the extractor
is not in the reference's repository and the reference holds no fixture for it.
"""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_extract_golden", os.path.join(ROOT, "scripts", "gen_extract_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GEN = _gen()
FROZEN = json.load(open(GEN.PATH))
CASE_NAMES = [c[0] for c in GEN.CASES]


def _inputs(pkg_synth, name):
    c = FROZEN["cases"][name]
    if name.startswith("small_quota"):      # this case's images are committed as data: no dependence on numpy's generator
        z = np.load(os.path.join(ROOT, "tests", "golden", name + "_images.npz"))
        return c, z["left"], z["right"]
    L, R = pkg_synth.stereo_pair(c["seed"], c["frame"], c["width"], c["height"])
    return c, L, R


def _compare(c, d):
    # inputs first: a generator drift must not read as a specification drift
    assert d["image_sha256"] == c["image_sha256"], "synthetic input generator changed (numpy version?), not the extractor"
    for k in d:
        assert d[k] == c[k], "%s differs from the frozen specification: %r vs %r" % (k, d[k], c[k])


def test_frozen_file_covers_the_required_sizes():
    sizes = {(c["width"], c["height"], c["n_features"]) for c in FROZEN["cases"].values()}
    assert {(752, 480, 1200), (752, 480, 2000), (1920, 1080, 4000)} <= sizes and len(sizes) >= 4
    assert any(c["width"] % 4 and c["height"] % 2 for c in FROZEN["cases"].values())      # an odd size


def test_committed_images_equal_the_generator(pkg):
    c, L, R = _inputs(pkg.synth, CASE_NAMES[-1])
    L2, R2 = pkg.synth.stereo_pair(c["seed"], c["frame"], c["width"], c["height"])
    assert np.array_equal(L, L2) and np.array_equal(R, R2)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_reproduces_frozen_spec(oracle, pkg, name):
    c, L, R = _inputs(pkg.synth, name)
    cam = oracle.Camera(**FROZEN["camera"])
    _compare(c, GEN.digest(L, R, GEN.oracle_results(oracle, cam, L, R, c["n_features"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASE_NAMES)
def test_hip_path_reproduces_frozen_spec(pkg, name):
    """orbx_process_stereo + the stage read-backs against the committed digests; the oracle is not involved."""
    c, L, R = _inputs(pkg.synth, name)
    n = c["n_features"]
    h = pkg.Handle(pkg.CameraModel(**FROZEN["camera"]), n, device=0, max_w=c["width"], max_h=c["height"], max_batch=1)
    try:
        kl, dl, kr, dr, m, pts, has = h.process_stereo(L, R, cap_kp=2 * n + 4096)
        r = dict(kl=kl, dl=dl, kr=kr, dr=dr, m=m, pts=pts, has=has,
                 levels=[L if l == 0 else h.debug_level(0, l) for l in range(8)],
                 blurs=[h.debug_level(0, l, blurred=True) for l in range(8)],
                 cands=[h.debug_candidates(0, l) for l in range(8)])
        _compare(c, GEN.digest(L, R, r))
    finally:
        h.close()


@pytest.mark.gpu
def test_hip_batch_path_reproduces_frozen_spec(pkg):
    """The throughput form (orbx_process_stereo_batch_device) on a batch holding the two 752x480 cases: same digests."""
    import torch
    names = [n for n in CASE_NAMES if FROZEN["cases"][n]["width"] == 752]
    feats = {FROZEN["cases"][n]["n_features"] for n in names}
    for nf in sorted(feats):
        sel = [n for n in names if FROZEN["cases"][n]["n_features"] == nf]
        pairs = [_inputs(pkg.synth, n) for n in sel]
        imgs = torch.from_numpy(np.stack([np.stack([L, R]) for _, L, R in pairs])).cuda()
        h = pkg.Handle(pkg.CameraModel(**FROZEN["camera"]), nf, device=0, max_w=752, max_h=480, max_batch=len(sel))
        try:
            out = h.alloc_batch_outputs(len(sel), nf + 1024)
            h.process_stereo_batch_device(imgs, out)
            h.check_status()
            for b, (c, L, R) in enumerate(pairs):
                fl, fr, m, pts, has = h.unpack_batch_outputs(out, b)
                kl, dl, kr, dr = fl.keypoints, fl.descriptors, fr.keypoints, fr.descriptors
                d = GEN.digest(L, R, dict(kl=kl, dl=dl, kr=kr, dr=dr, m=m, pts=pts, has=has, levels=[], blurs=[], cands=[]))
                for k in ("n_keypoints", "keypoints_sha256", "descriptors_sha256", "n_matches", "matches_sha256", "n_points",
                          "has_point_sha256", "points_sha256"):
                    assert d[k] == c[k], (sel[b], k)
        finally:
            h.close()
