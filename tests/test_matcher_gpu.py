"""GPU parity of the Hamming matchers, through the C ABI, bit-exact against the oracle
(stereo.rs:80-216, tracker.rs:1001-1010)."""
import numpy as np
import pytest

from conftest import records_equal

pytestmark = pytest.mark.gpu


def test_hamming_known_answers_gpu(gpu_handle, golden):
    for g in golden["hamming"]:      # vocabulary/mod.rs:429-441, corrector.rs:625-634
        a = np.array(g["a"], np.uint8); b = np.array(g["b"], np.uint8)
        assert gpu_handle.hamming_batch(a, b)[0] == g["expect"]


def test_hamming_batch_parity(gpu_handle, oracle):
    rng = np.random.default_rng(1)
    for n in (1, 63, 64, 65, 1000, 100003):
        a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        assert np.array_equal(gpu_handle.hamming_batch(a, b), oracle.hamming_batch(a, b))
    assert len(gpu_handle.hamming_batch(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8))) == 0


@pytest.mark.parametrize("nL,nR,seed", [(2000, 2000, 0), (1200, 1180, 1), (4000, 4100, 2), (1, 1, 3),
                                        (17, 3000, 4), (2500, 5, 5), (64, 64, 6), (257, 2049, 7)])
def test_stereo_match_parity(gpu_handle, oracle, pkg, nL, nR, seed):
    kpL, dL, kpR, dR = pkg.synth.matcher_features(seed, nL, nR, pkg.KEYPOINT)
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    m0, p0, h0 = oracle.stereo_match(cam, kpL, dL, kpR, dR)
    m1, p1, h1 = gpu_handle.stereo_match(kpL, dL, kpR, dR)
    assert records_equal(m0, m1)
    assert np.array_equal(h0, h1)
    assert np.array_equal(p0[h0 == 1], p1[h1 == 1])        # f64, bit-exact
    if nL >= 1000 and nR >= 1000:
        assert len(m0) > 0.2 * min(nL, nR)


def test_stereo_match_ties_and_quirks(gpu_handle, oracle, pkg):
    # equal best distances -> no match; duplicates of the same descriptor everywhere
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    rng = np.random.default_rng(9)
    n = 300
    kpL = np.zeros(n, pkg.KEYPOINT); kpR = np.zeros(n, pkg.KEYPOINT)
    kpL["x"] = rng.uniform(200, 700, n).astype(np.float32); kpL["y"] = rng.integers(40, 60, n).astype(np.float32)
    kpR["x"] = rng.uniform(31, 500, n).astype(np.float32); kpR["y"] = rng.integers(40, 60, n).astype(np.float32)
    base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    dL = base[rng.integers(0, 4, n)]; dR = base[rng.integers(0, 4, n)]
    dR = dR.copy(); dR[:, 0] ^= rng.integers(0, 4, n).astype(np.uint8)   # distances 0..2, many ties
    m0, p0, h0 = oracle.stereo_match(cam, kpL, dL, kpR, dR)
    m1, p1, h1 = gpu_handle.stereo_match(kpL, dL, kpR, dR)
    assert records_equal(m0, m1) and np.array_equal(h0, h1) and np.array_equal(p0[h0 == 1], p1[h1 == 1])
    # |disparity| < 0.5 -> match kept, point None (stereo.rs:205-207)
    kl = np.zeros(2, pkg.KEYPOINT); kr = np.zeros(2, pkg.KEYPOINT)
    kl["x"] = [400.0, 90.0]; kl["y"] = [100.0, 300.0]
    kr["x"] = [398.6, 5.0]; kr["y"] = [100.0, 470.0]
    d = np.zeros((2, 32), np.uint8)
    m0, p0, h0 = oracle.stereo_match(cam, kl, d, kr, d)
    m1, p1, h1 = gpu_handle.stereo_match(kl, d, kr, d)
    assert records_equal(m0, m1) and np.array_equal(h0, h1)


def test_stereo_match_empty(gpu_handle, pkg):
    e = np.zeros(0, pkg.KEYPOINT); de = np.zeros((0, 32), np.uint8)
    m, p, h = gpu_handle.stereo_match(e, de, e, de)
    assert len(m) == 0 and len(h) == 0
    k = np.zeros(3, pkg.KEYPOINT); k["x"] = 100; k["y"] = 100
    m, p, h = gpu_handle.stereo_match(k, np.zeros((3, 32), np.uint8), e, de)
    assert len(m) == 0 and h.sum() == 0


@pytest.mark.parametrize("nq,nt,seed", [(2000, 2000, 0), (1, 1, 1), (15, 17, 2), (16, 256, 3), (1999, 2333, 4), (4000, 3900, 5)])
def test_crosscheck_parity(gpu_handle, oracle, pkg, nq, nt, seed):
    _, q, _, t = pkg.synth.matcher_features(100 + seed, nq, nt, pkg.KEYPOINT)
    m0 = oracle.crosscheck_match(q, t)
    m1 = gpu_handle.hamming_match_crosscheck(q, t)
    assert records_equal(m0, m1)


def test_crosscheck_ties(gpu_handle, oracle):
    rng = np.random.default_rng(4)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
    q = base[rng.integers(0, 8, 500)]; t = base[rng.integers(0, 8, 700)]
    assert records_equal(oracle.crosscheck_match(q, t), gpu_handle.hamming_match_crosscheck(q, t))
    assert len(gpu_handle.hamming_match_crosscheck(np.zeros((0, 32), np.uint8), t)) == 0


def test_batch_device_matches_host_path(gpu_handle, oracle, pkg):
    """The device-resident batch form = the single-pair host form on each pair."""
    import torch
    B, cap = 5, 2304
    out = gpu_handle.alloc_batch_outputs(B, cap)
    sets = []
    kp_h = np.zeros((B, 2, cap), pkg.KEYPOINT); desc_h = np.zeros((B, 2, cap, 32), np.uint8)
    nkp_h = np.zeros((B, 2), np.int32)
    for b in range(B):
        nL, nR = 1500 + 100 * b, 2200 - 150 * b
        s = pkg.synth.matcher_features(50 + b, nL, nR, pkg.KEYPOINT)
        sets.append(s)
        kp_h[b, 0, :nL] = s[0]; desc_h[b, 0, :nL] = s[1]; kp_h[b, 1, :nR] = s[2]; desc_h[b, 1, :nR] = s[3]
        nkp_h[b] = (nL, nR)
    out["kp"].copy_(torch.from_numpy(kp_h.view(np.float32).reshape(B, 2, cap, 7)))
    out["desc"].copy_(torch.from_numpy(desc_h)); out["nkp"].copy_(torch.from_numpy(nkp_h))
    torch.cuda.synchronize()
    gpu_handle.stereo_match_batch_device(out)
    gpu_handle.check_status()
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    for b in range(B):
        _, _, m, pts, has = gpu_handle.unpack_batch_outputs(out, b)
        m0, p0, h0 = oracle.stereo_match(cam, *sets[b])
        assert records_equal(m0, m) and np.array_equal(h0, has) and np.array_equal(p0[h0 == 1], pts[has == 1])


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,nq,radius,seed", [(2000, 1500, 15.0, 0), (1200, 3000, 15.0, 1), (300, 64, 40.0, 2), (5, 9, 15.0, 3), (4000, 100, 7.5, 4)])
def test_guided_match_parity(gpu_handle, oracle, pkg, mode, n, nq, radius, seed):
    """FeatureGrid + descriptor search (tracking_frame.rs:52-128, tracker.rs:880-923 / :1126-1157), bit-exact"""
    kp, d, kq, dq = pkg.synth.matcher_features(200 + seed, n, max(nq, 1), pkg.KEYPOINT)
    rng = np.random.default_rng(seed)
    uv = np.stack([kq["x"].astype(np.float64) + rng.uniform(-20, 140, len(kq)), kq["y"].astype(np.float64) + rng.uniform(-3, 3, len(kq))], 1)[:nq]
    uv[::17] = rng.uniform(-200, 1000, (len(uv[::17]), 2))        # queries outside the image: the wrap-around cell quirk
    i0, d0 = oracle.guided_match(kp, d, 752.0, 480.0, uv, dq[:nq], radius, mode)
    i1, d1 = gpu_handle.guided_match(kp, d, 752.0, 480.0, uv, dq[:nq], radius, mode)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    if n >= 1000 and nq >= 1000:
        assert (i0 >= 0).sum() > 50


def test_guided_match_ties_and_empty(gpu_handle, oracle, pkg):
    rng = np.random.default_rng(5)
    n = 800
    kp = np.zeros(n, pkg.KEYPOINT)
    kp["x"] = rng.uniform(0, 752, n).astype(np.float32); kp["y"] = rng.uniform(0, 480, n).astype(np.float32)
    base = rng.integers(0, 256, (3, 32), dtype=np.uint8)
    d = base[rng.integers(0, 3, n)]                          # only three distinct descriptors: ties everywhere
    uv = np.stack([rng.uniform(0, 752, 500), rng.uniform(0, 480, 500)], 1)
    dq = base[rng.integers(0, 3, 500)].copy(); dq[:, 0] ^= rng.integers(0, 2, 500).astype(np.uint8)
    for mode in (0, 1):
        i0, d0 = oracle.guided_match(kp, d, 752.0, 480.0, uv, dq, 15.0, mode)
        i1, d1 = gpu_handle.guided_match(kp, d, 752.0, 480.0, uv, dq, 15.0, mode)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    i1, d1 = gpu_handle.guided_match(kp[:0], d[:0], 752.0, 480.0, uv[:4], dq[:4], 15.0, 1)
    assert np.all(i1 == -1)
    i1, d1 = gpu_handle.guided_match(kp, d, 752.0, 480.0, uv[:0], dq[:0], 15.0, 0)
    assert len(i1) == 0
