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


def test_stereo_match_crowded_rows(gpu_handle, oracle, pkg):
    """More right keypoints in a left keypoint's rows than one pass of the LDS matcher's candidate mask holds (32 per lane, four lanes per left
    keypoint): 1500 right keypoints in three image rows, so every left keypoint walks several chunks of 128 candidates; and row coordinates a hair's
    breadth inside / outside the vertical gate (|vl - vr| <= 2 in f32), which the matcher's row range of vl -+ 2.01 must still reach."""
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    rng = np.random.default_rng(21)
    nL, nR = 400, 1500
    kpL = np.zeros(nL, pkg.KEYPOINT); kpR = np.zeros(nR, pkg.KEYPOINT)
    kpL["x"] = rng.uniform(300, 740, nL).astype(np.float32); kpL["y"] = rng.uniform(99.0, 103.0, nL).astype(np.float32)
    kpR["x"] = rng.uniform(31, 700, nR).astype(np.float32); kpR["y"] = rng.uniform(100.0, 102.999, nR).astype(np.float32)
    dL = rng.integers(0, 256, (nL, 32), dtype=np.uint8)
    dR = dL[rng.integers(0, nL, nR)].copy(); dR[:, :4] ^= rng.integers(0, 256, (nR, 4), dtype=np.uint8)   # distances of 0..32 bits to some left descriptor
    m0, p0, h0 = oracle.stereo_match(cam, kpL, dL, kpR, dR)
    m1, p1, h1 = gpu_handle.stereo_match(kpL, dL, kpR, dR)
    assert records_equal(m0, m1) and np.array_equal(h0, h1) and np.array_equal(p0[h0 == 1], p1[h1 == 1]) and len(m0) > 50
    # the vertical gate at its edge: vr = vl + 2 exactly, the next float above and below, and the same on the other side, across a row boundary
    vl = np.float32(200.75)
    edge = np.array([vl + np.float32(2.0), np.nextafter(vl + np.float32(2.0), np.float32(1e9)), np.nextafter(vl + np.float32(2.0), np.float32(0)),
                     vl - np.float32(2.0), np.nextafter(vl - np.float32(2.0), np.float32(0)), np.nextafter(vl - np.float32(2.0), np.float32(1e9)),
                     np.float32(203.0), np.float32(198.0), np.float32(202.9999), np.float32(198.5)], np.float32)
    kl = np.zeros(len(edge), pkg.KEYPOINT); kr = np.zeros(len(edge), pkg.KEYPOINT)
    kl["x"] = 400.0 + 10.0 * np.arange(len(edge)); kl["y"] = vl
    kr["x"] = kl["x"] - 20.0; kr["y"] = edge
    d = np.zeros((len(edge), 32), np.uint8); d[:, 0] = np.arange(len(edge))          # every pair its own descriptor: who matches whom is the gates' doing
    m0, p0, h0 = oracle.stereo_match(cam, kl, d, kr, d)
    m1, p1, h1 = gpu_handle.stereo_match(kl, d, kr, d)
    assert records_equal(m0, m1) and np.array_equal(h0, h1) and np.array_equal(p0[h0 == 1], p1[h1 == 1])


def test_stereo_match_lds_form_equals_global_form(pkg, tmp_path):
    """stereo_match_lds_kernel (the pair's right image in LDS — bucket sort, matcher, compaction and triangulation in one workgroup per pair: what large
    batches run and this suite forces) against the three-launch form with the LDS matcher in the middle (ORBX_SM_LDS=2) and against stereo_match_kernel
    (what calls of fewer pairs than CUs run; ORBX_SM_LDS=0): same candidate sets, same order-independent top-2 — the same bytes.  Three child
    processes: the switch is read at the first stereo-match call."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import torch
        import orb_slam3_rust_amd as P
        cam = P.CameraModel(**P.synth.EUROC_CAMERA)
        out = {}
        h = P.Handle(cam, 2000, device=0, max_w=752, max_h=480, max_batch=6)
        for i, (nL, nR) in enumerate([(2000, 2000), (1200, 1180), (1, 1), (17, 3000), (2500, 5), (257, 2049)]):
            kpL, dL, kpR, dR = P.synth.matcher_features(30 + i, nL, nR, P.KEYPOINT)
            m, pts, has = h.stereo_match(kpL, dL, kpR, dR)
            out["m%%d" %% i] = np.frombuffer(m.tobytes(), np.uint8); out["p%%d" %% i] = pts; out["h%%d" %% i] = has
        pairs = np.stack([np.stack(P.synth.stereo_pair(5, f)) for f in range(6)])
        o = h.alloc_batch_outputs(6, 2000 + 64)
        h.process_stereo_batch_device(torch.from_numpy(pairs).cuda(), o)
        h.synchronize()
        out["bm"] = o["matches"].cpu().numpy().view(np.uint8).reshape(-1); out["bn"] = o["nmatches"].cpu().numpy()
        out["bp"] = o["points"].cpu().numpy(); out["bh"] = o["has_point"].cpu().numpy()
        h.close()
        np.savez(sys.argv[1], **out)
    """ % root)
    res = {}
    for mode in ("1", "2", "0"):
        path = str(tmp_path / ("sm%s.npz" % mode))
        env = dict(os.environ, ORBX_SM_LDS=mode)
        subprocess.run([sys.executable, "-c", script, path], check=True, env=env, timeout=600)
        res[mode] = np.load(path)
    assert sorted(res["1"].files) == sorted(res["0"].files) == sorted(res["2"].files) and len(res["1"].files) == 22
    for k in res["1"].files:
        assert np.array_equal(res["1"][k], res["0"][k]) and np.array_equal(res["1"][k], res["2"][k]), k
    assert int(res["1"]["bn"].min()) > 300
