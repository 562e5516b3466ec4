"""GPU parity of the ORB extractor and of the whole per-frame path (StereoProcessor::process,
stereo.rs:52-66), through the C ABI, bit-exact against the CPU specification (oracle/orb_ref.cpp,
SURVEY.md Appendix A; parity with OpenCV itself is unpinned)."""
import numpy as np
import pytest

from conftest import records_equal

pytestmark = pytest.mark.gpu


def _first_diff(a, b):
    for name in a.dtype.names:
        bad = np.nonzero(a[name].view(np.uint32) != b[name].view(np.uint32))[0]
        if len(bad):
            return "%s differs at %d rows, first %d: %r vs %r" % (name, len(bad), bad[0], a[name][bad[0]], b[name][bad[0]])
    return "equal"


@pytest.fixture(scope="module")
def h2000(pkg):
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 2000, device=0, max_w=1920, max_h=1080, max_batch=8)
    yield h
    h.close()


@pytest.fixture(scope="module")
def frame0(pkg, h2000):
    L, R = pkg.synth.stereo_pair(11, 0)
    res = h2000.process_stereo(L, R)
    return L, R, res


def test_pyramid_levels_bit_exact(h2000, oracle, frame0):
    L, R, _ = frame0
    p = oracle.orb_params(2000)
    for img_i, img in enumerate((L, R)):
        for l in range(1, 8):
            want = oracle.orb_pyramid_level(img, p, l)
            got = h2000.debug_level(img_i, l)
            assert got.shape == want.shape
            assert np.array_equal(got, want), "level %d of image %d: %d pixels differ" % (l, img_i, (got != want).sum())


def test_blur_levels_bit_exact(h2000, oracle, frame0):
    L, R, _ = frame0
    p = oracle.orb_params(2000)
    for l in range(8):
        want = oracle.orb_blur_level(L, p, l)
        got = h2000.debug_level(0, l, blurred=True)
        assert np.array_equal(got, want), "blur level %d: %d pixels differ" % (l, (got != want).sum())


@pytest.mark.parametrize("w,h", [(101, 77), (249, 131), (250, 300), (333, 258), (501, 97), (641, 481), (753, 261), (997, 64), (1241, 376),
                                 (756, 140), (1004, 90), (260, 200)])   # rows 4- but not 16-byte aligned: the resize staging's clamped last column
def test_pyramid_and_blur_levels_odd_sizes(pkg, oracle, w, h):
    """Whole levels (not only the patches keypoints sample) at widths whose last dword holds 1, 2, 3 or 4 pixels and whose
    remainder after the 248-px blur strips exercises the full, half and quarter strip modes; heights below, at and above
    one 256-row block: BORDER_REFLECT_101 at all four edges, resize windows clamped at the right edge."""
    rng = np.random.default_rng(w * 1000 + h)
    L = rng.integers(0, 256, (h, w), dtype=np.uint8)
    R = np.ascontiguousarray(L[:, ::-1])
    hd = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 500, device=0, max_w=w, max_h=h, max_batch=1)
    try:
        hd.process_stereo(L, R)
        p = oracle.orb_params(500)
        for img_i, img in enumerate((L, R)):
            for l in range(8):
                if l > 0:
                    want = oracle.orb_pyramid_level(img, p, l)
                    got = hd.debug_level(img_i, l)
                    assert np.array_equal(got, want), "%dx%d level %d image %d: %d pixels differ" % (w, h, l, img_i, (got != want).sum())
                want = oracle.orb_blur_level(img, p, l)
                got = hd.debug_level(img_i, l, blurred=True)
                bad = np.argwhere(got != want)
                assert len(bad) == 0, "%dx%d blur level %d image %d: %d pixels differ, first (y,x)=%s" % (w, h, l, img_i, len(bad), bad[:1])
    finally:
        hd.close()


def test_fast_candidates_bit_exact(h2000, oracle, frame0):
    L, R, _ = frame0
    p = oracle.orb_params(2000)
    for img_i, img in enumerate((L, R)):
        for l in range(8):
            want = np.sort(oracle.orb_fast_level(img, p, l))
            got = np.sort(h2000.debug_candidates(img_i, l))
            assert np.array_equal(got, want), "FAST level %d: %d vs %d candidates" % (l, len(got), len(want))


@pytest.mark.parametrize("thr", [1, 20, 64, 126, 127])
def test_fast_pretest_byte_edges(oracle, pkg, thr):
    """the byte-parallel compass pre-test at the ends of its arithmetic: pixels at 0 / 255 (v + t beyond 255, v - t below 0), ring values
    exactly at v + t, v + t + 1, v - t, v - t - 1, the sign bit of a byte on either side of every comparison — candidates of every level
    against the oracle (oracle/orb_ref.cpp A.5)"""
    rng = np.random.default_rng(1000 + thr)
    h_px, w_px = 240, 320
    base = rng.choice(np.array([0, 1, thr, thr + 1, 127, 128, 129, 254 - thr, 255 - thr, 254, 255], np.int32), size=(h_px // 8, w_px // 8))
    img = np.kron(base, np.ones((8, 8), np.int32))
    # isolated pixels and short runs at the exact thresholds around each cell's value
    delta = rng.choice(np.array([-thr - 1, -thr, thr, thr + 1, 0, 0, 0, 0], np.int32), size=(h_px, w_px))
    img = np.clip(img + np.where(rng.random((h_px, w_px)) < 0.35, delta, 0), 0, 255).astype(np.uint8)
    img[::7, ::5] = 255
    img[3::11, 2::9] = 0
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 600, device=0, max_w=w_px, max_h=h_px, max_batch=1, orb_params=dict(fast_threshold=thr, n_levels=4))
    p = oracle.orb_params(600)
    p.fast_threshold = thr
    p.n_levels = 4
    h.process_stereo(img, img[:, ::-1].copy(), cap_kp=20000)
    total = 0
    for img_i, im in enumerate((img, img[:, ::-1].copy())):
        for l in range(4):
            want = np.sort(oracle.orb_fast_level(im, p, l))
            got = np.sort(h.debug_candidates(img_i, l))
            assert np.array_equal(got, want), "threshold %d, FAST level %d: %d vs %d candidates" % (thr, l, len(got), len(want))
            total += len(want)
    assert total > (200 if thr <= 64 else 0)
    h.close()


def test_process_stereo_bit_exact(h2000, oracle, pkg, frame0):
    L, R, (kpL, dL, kpR, dR, m, pts, has) = frame0
    p = oracle.orb_params(2000)
    okL, odL = oracle.orb_extract(L, p)
    okR, odR = oracle.orb_extract(R, p)
    assert len(kpL) == len(okL) and len(kpR) == len(okR)
    assert records_equal(kpL, okL), _first_diff(kpL, okL)
    assert records_equal(kpR, okR), _first_diff(kpR, okR)
    assert np.array_equal(dL, odL), "%d descriptor rows differ" % (dL != odL).any(1).sum()
    assert np.array_equal(dR, odR)
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    m0, p0, h0 = oracle.stereo_match(cam, okL, odL, okR, odR)
    assert records_equal(m0, m) and np.array_equal(h0, has) and np.array_equal(p0[h0 == 1], pts[has == 1])
    assert len(m) > 200


@pytest.mark.parametrize("n_features,seed", [(1200, 1), (2000, 2), (4000, 3), (500, 4)])
def test_extract_other_quotas(oracle, pkg, n_features, seed):
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), n_features, device=0, max_w=752, max_h=480, max_batch=1)
    L, R = pkg.synth.stereo_pair(seed, 5)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(L, R, cap_kp=n_features + 1024)
    p = oracle.orb_params(n_features)
    for (k, d, img) in ((kpL, dL, L), (kpR, dR, R)):
        ok, od = oracle.orb_extract(img, p)
        assert records_equal(k, ok), _first_diff(k, ok) if len(k) == len(ok) else "%d vs %d" % (len(k), len(ok))
        assert np.array_equal(d, od)
    h.close()


@pytest.mark.parametrize("w,hh", [(1920, 1080), (640, 480), (333, 257), (752, 480)])
def test_extract_other_sizes_and_strides(oracle, pkg, w, hh):
    n = 4000 if w == 1920 else 1500
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), n, device=0, max_w=1920, max_h=1080, max_batch=1)
    L, R = pkg.synth.stereo_pair(21, 0, w, hh)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(L, R, cap_kp=n + 2048)
    p = oracle.orb_params(n)
    ok, od = oracle.orb_extract(L, p)
    assert records_equal(kpL, ok), _first_diff(kpL, ok) if len(kpL) == len(ok) else "%d vs %d" % (len(kpL), len(ok))
    assert np.array_equal(dL, od)
    ok, od = oracle.orb_extract(R, p)
    assert records_equal(kpR, ok) and np.array_equal(dR, od)
    h.close()


def test_degenerate_images(oracle, pkg):
    """flat image -> no keypoints; binary dot grid -> all FAST scores tie, every candidate is kept by
    retainBest's tie rule and the Harris responses tie too (exercises the big rank-sort path)."""
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 300, device=0, max_w=752, max_h=480, max_batch=1)
    flat = np.full((480, 752), 90, np.uint8)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(flat, flat)
    assert len(kpL) == 0 and len(kpR) == 0 and len(m) == 0
    dots = np.full((480, 752), 20, np.uint8)
    dots[8::12, 8::12] = 240
    p = oracle.orb_params(300)
    ok, od = oracle.orb_extract(dots, p)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(dots, dots, cap_kp=len(ok) + 64)
    assert len(ok) > 1500    # far more than n_features: ties are all kept (keypoint.cpp retainBest)
    assert records_equal(kpL, ok), _first_diff(kpL, ok) if len(kpL) == len(ok) else "%d vs %d" % (len(kpL), len(ok))
    assert np.array_equal(dL, od)
    # capacity too small -> error, never truncation
    with pytest.raises(pkg.OrbxError) as e:
        h.process_stereo(dots, dots, cap_kp=1000)
    assert e.value.code == -4
    h.close()


def test_batch_device_equals_single(h2000, oracle, pkg):
    """[batch,2,h,w] device-resident form: every pair equals the oracle run on that pair."""
    import torch
    B = 6
    imgs = pkg.synth.stereo_batch(7, 100, B)
    out = h2000.alloc_batch_outputs(B, 2304)
    d_imgs = torch.from_numpy(imgs).cuda()
    h2000.process_stereo_batch_device(d_imgs, out)
    h2000.check_status()
    p = oracle.orb_params(2000)
    cam = oracle.Camera(**pkg.synth.EUROC_CAMERA)
    for b in range(B):
        fl, fr, m, pts, has = h2000.unpack_batch_outputs(out, b)
        okL, odL = oracle.orb_extract(imgs[b, 0], p)
        okR, odR = oracle.orb_extract(imgs[b, 1], p)
        assert records_equal(fl.keypoints, okL) and np.array_equal(fl.descriptors, odL)
        assert records_equal(fr.keypoints, okR) and np.array_equal(fr.descriptors, odR)
        m0, p0, h0 = oracle.stereo_match(cam, okL, odL, okR, odR)
        assert records_equal(m0, m) and np.array_equal(h0, has) and np.array_equal(p0[h0 == 1], pts[has == 1])


def test_unaligned_rows(oracle, pkg):
    """caller rows that are not 4-byte aligned go through the level-0 copy"""
    import torch
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 800, device=0, max_w=752, max_h=480, max_batch=1)
    L, R = pkg.synth.stereo_pair(31, 2, 333, 257)
    out = h.alloc_batch_outputs(1, 2048)
    d = torch.from_numpy(np.stack([L, R])[None]).cuda()     # stride 333: unaligned
    h.process_stereo_batch_device(d, out)
    h.check_status()
    fl, fr, m, pts, has = h.unpack_batch_outputs(out, 0)
    ok, od = oracle.orb_extract(L, oracle.orb_params(800))
    assert records_equal(fl.keypoints, ok) and np.array_equal(fl.descriptors, od)
    h.close()


def test_stereo_processor_mirror(pkg, oracle):
    """the host mirror with the reference's names: StereoProcessor::new / process"""
    cam = pkg.CameraModel(**pkg.synth.EUROC_CAMERA)
    sp = pkg.StereoProcessor.new(cam, 1200)          # main.rs:53 uses 1200
    L, R = pkg.synth.stereo_pair(1, 1)
    f = sp.process(L, R, 1403636579763555584)
    assert f.timestamp_ns == 1403636579763555584
    assert len(f.left_features.keypoints) == len(f.left_features.descriptors) <= 1200 + 50
    assert len(f.points_cam_options()) == len(f.left_features.keypoints)
    assert sum(p is not None for p in f.points_cam_options()) == int(f.has_point.sum())
    ok, od = oracle.orb_extract(L, oracle.orb_params(1200))
    assert records_equal(f.left_features.keypoints, ok)


def test_pipelined_host_batch_equals_device_batch(pkg):
    """orbx_process_stereo_batch (three-stream pipelined host form, chunks of max_batch) == the device form"""
    import torch
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 1000, device=0, max_w=752, max_h=480, max_batch=3)
    B, cap = 8, 1304                         # 3 chunks: 3 + 3 + 2 pairs, both staging buffers reused
    imgs = pkg.synth.stereo_batch(41, 0, B)
    host_in = torch.from_numpy(imgs).pin_memory()
    hout = h.alloc_host_outputs(B, cap)
    h.process_stereo_batch_host(host_in, hout)
    h2 = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 1000, device=0, max_w=752, max_h=480, max_batch=B)
    dout = h2.alloc_batch_outputs(B, cap)
    h2.process_stereo_batch_device(torch.from_numpy(imgs).cuda(), dout)
    h2.check_status()
    nk = dout["nkp"].cpu()
    assert torch.equal(hout["nkp"], nk) and torch.equal(hout["nmatches"], dout["nmatches"].cpu())
    for b in range(B):
        for s in range(2):
            n = int(nk[b, s])
            assert torch.equal(hout["kp"][b, s, :n].view(torch.int32), dout["kp"][b, s, :n].cpu().view(torch.int32))
            assert torch.equal(hout["desc"][b, s, :n], dout["desc"][b, s, :n].cpu())
        nm = int(hout["nmatches"][b])
        assert torch.equal(hout["matches"][b, :nm], dout["matches"][b, :nm].cpu())
        nl = int(nk[b, 0])
        hp = dout["has_point"][b, :nl].cpu()
        assert torch.equal(hout["has_point"][b, :nl], hp)
        assert torch.equal(hout["points"][b, :nl][hp.bool()], dout["points"][b, :nl].cpu()[hp.bool()])
    # pageable host memory works too (no overlap, same results)
    hout2 = h.alloc_host_outputs(B, cap, pin=False)
    h.process_stereo_batch_host(torch.from_numpy(imgs), hout2)
    assert torch.equal(hout2["nkp"], nk) and torch.equal(hout2["desc"], hout["desc"])
    h.close(); h2.close()


@pytest.mark.parametrize("over", [dict(n_levels=4), dict(fast_threshold=10), dict(fast_threshold=45, n_levels=6), dict(fast_threshold=127), dict(fast_threshold=128),
                                  dict(fast_threshold=150, n_levels=3), dict(fast_threshold=1, n_levels=2),
                                  dict(scale_factor=1.3), dict(scale_factor=1.5, n_levels=5), dict(n_levels=1)])
def test_orb_parameter_variations(oracle, pkg, over):
    """the ORB parameters the ABI lets vary (levels, scale factor, FAST threshold) — still bit-exact"""
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 900, device=0, max_w=752, max_h=480, max_batch=1, orb_params=over)
    p = oracle.orb_params(900)
    for k, v in over.items():
        setattr(p, k, v)
    L, R = pkg.synth.stereo_pair(61, 3)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(L, R)
    ok, od = oracle.orb_extract(L, p)
    assert records_equal(kpL, ok), _first_diff(kpL, ok) if len(kpL) == len(ok) else "%d vs %d" % (len(kpL), len(ok))
    assert np.array_equal(dL, od) and len(ok) > (300 if p.fast_threshold < 100 else 0)
    assert len(ok) == 0 or int(kpL["octave"].max()) <= p.n_levels - 1
    h.close()


def test_graph_replay_survives_workspace_growth(oracle, pkg):
    """the captured single-pair graph must be dropped when a larger batch call re-allocates the workspaces"""
    import torch
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 700, device=0, max_w=752, max_h=480, max_batch=6)
    L, R = pkg.synth.stereo_pair(71, 0)
    first = [h.process_stereo(L, R) for _ in range(4)]            # eager, eager, capture, replay
    out = h.alloc_batch_outputs(6, 1004)
    h.process_stereo_batch_device(torch.from_numpy(pkg.synth.stereo_batch(72, 0, 6)).cuda(), out)   # grows every workspace
    h.check_status()
    again = [h.process_stereo(L, R) for _ in range(4)]
    for r in first[1:] + again:
        assert all(a.tobytes() == b.tobytes() for a, b in zip(first[0], r))
    ok, od = oracle.orb_extract(L, oracle.orb_params(700))
    assert records_equal(first[0][0], ok) and np.array_equal(first[0][1], od)
    h.close()


def test_graph_replay_survives_geometry_change(oracle, pkg):
    """a call at ANOTHER image size between two replays rewrites the resize / tile tables in place (same
    buffer, no re-allocation) — the captured graph has the old geometry baked in and must be dropped, and the table
    upload must not race kernels still in flight.  Alternates two sizes around the captured graph, both bit-exact."""
    import torch
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 600, device=0, max_w=752, max_h=480, max_batch=2)
    L, R = pkg.synth.stereo_pair(81, 0)
    l2, r2 = pkg.synth.stereo_pair(82, 0, 416, 304)
    p = oracle.orb_params(600)
    okL, odL = oracle.orb_extract(L, p)
    ok2, od2 = oracle.orb_extract(l2, p)
    small = torch.from_numpy(np.stack([l2, r2])[None]).cuda()
    out = h.alloc_batch_outputs(1, 1200)
    big0 = [h.process_stereo(L, R) for _ in range(5)]              # eager calls, capture, replays
    for rnd in range(3):
        h.process_stereo_batch_device(small, out)                  # asynchronous, other geometry: tables rewritten
        again = [h.process_stereo(L, R) for _ in range(4 if rnd == 1 else 1)]   # must not replay the stale graph
        h.check_status()
        fl, fr, m, pts, has = h.unpack_batch_outputs(out, 0)
        assert records_equal(fl.keypoints, ok2) and np.array_equal(fl.descriptors, od2), "round %d small" % rnd
        for r in again:
            assert all(a.tobytes() == b.tobytes() for a, b in zip(big0[0], r)), "round %d" % rnd
        k2 = h.process_stereo(l2, r2)                              # the host-buffer form at the small size as well
        assert records_equal(k2[0], ok2) and np.array_equal(k2[1], od2)
    assert records_equal(big0[0][0], okL) and np.array_equal(big0[0][1], odL)
    for r in big0[1:]:
        assert all(a.tobytes() == b.tobytes() for a, b in zip(big0[0], r))
    h.close()


def test_profiling_of_one_kernel_only(h2000, frame0):
    """orbx_set_profiling_only: HIP events around the launches of one kernel — the per-kernel table then holds that kernel alone, with
    as many launches as the calls made; orbx_set_profiling(on) brackets every launch again; results do not depend on either."""
    L, R, a = frame0
    h2000.set_profiling(True, only="fast_kernel")
    b = [h2000.process_stereo(L, R) for _ in range(3)]
    kt = h2000.kernel_times()
    assert list(kt) == ["fast_kernel"] and kt["fast_kernel"][1] == 3 and kt["fast_kernel"][0] > 0.0
    only_ms = kt["fast_kernel"][0] / 3
    h2000.set_profiling(True)
    h2000.process_stereo(L, R)
    kt = h2000.kernel_times()
    assert "fast_kernel" in kt and ("describe_tile_kernel" in kt or "describe_fused_kernel" in kt) and len(kt) >= 5
    assert only_ms < 3.0 * kt["fast_kernel"][0] + 0.02                            # the kernel's own duration, not the span between two of its launches
    h2000.set_profiling(False)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a, b[-1]))


def test_describe_tile_form_equals_per_keypoint_form(pkg, tmp_path):
    """describe_tile_kernel (one workgroup per describe tile: the tile's rectangle blurred once into LDS, the keypoints ordered by tile by
    rank_select_kernel) against describe_fused_kernel (a 48 x 48 window blurred per keypoint; ORBX_DESC_TILE=0): the same integers, so
    keypoints and descriptors must be the same bits.  Sizes: the bench's, 400 x 226 whose level 7 is 63 rows — keypoints but no room for a 64 x 64
    window: the tile form must step aside by itself (no describe tiles) —, odd widths that pull the last window of a row / column back inside the level, a row
    pitch that is not a multiple of 4 (level 0 copied), 1920 x 1080, and a batch whose images differ.  Two child processes: the switch is
    read when the geometry is prepared."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import torch
        import orb_slam3_rust_amd as P
        cam = P.CameraModel(**P.synth.EUROC_CAMERA)
        out = {}
        rng = np.random.default_rng(77)
        def scene(w, h, seed):
            L, R = P.synth.stereo_pair(seed, seed, w=w, h=h)
            return L, R
        for i, (w, hh, n) in enumerate([(752, 480, 2000), (333, 259, 500), (400, 226, 500), (641, 479, 1200), (190, 150, 300), (1241, 376, 1500), (1920, 1080, 4000)]):
            h = P.Handle(cam, n, device=0, max_w=w, max_h=hh, max_batch=1)
            L, R = scene(w, hh, 40 + i)
            kpL, dL, kpR, dR, m, pts, has = h.process_stereo(L, R, cap_kp=n + 2048)
            out["kL%%d" %% i] = np.frombuffer(kpL.tobytes(), np.uint8); out["dL%%d" %% i] = dL
            out["kR%%d" %% i] = np.frombuffer(kpR.tobytes(), np.uint8); out["dR%%d" %% i] = dR
            out["m%%d" %% i] = np.frombuffer(m.tobytes(), np.uint8)
            h.close()
        # binary dot grid: every FAST score and every Harris response ties, all candidates are kept — more than 2048 on level 0, so the
        # rank-sort path (which orders by tile too) runs
        h = P.Handle(cam, 300, device=0, max_w=752, max_h=480, max_batch=1)
        dots = np.full((480, 752), 20, np.uint8)
        dots[8::12, 8::12] = 240
        kpL, dL, kpR, dR, m, pts, has = h.process_stereo(dots, dots, cap_kp=20000)
        out["kn"] = np.frombuffer(kpL.tobytes(), np.uint8); out["dn"] = dL
        h.close()
        # a device batch of different images
        h = P.Handle(cam, 2000, device=0, max_w=752, max_h=480, max_batch=6)
        pairs = np.stack([np.stack(P.synth.stereo_pair(3, f)) for f in range(6)])
        o = h.alloc_batch_outputs(6, 2000 + 1024)
        h.process_stereo_batch_device(torch.from_numpy(pairs).cuda(), o)
        h.synchronize()
        out["bk"] = o["kp"].cpu().numpy().view(np.uint8).reshape(-1); out["bd"] = o["desc"].cpu().numpy().reshape(-1); out["bn"] = o["nkp"].cpu().numpy()
        h.close()
        np.savez(sys.argv[1], **out)
    """ % root)
    res = {}
    for mode in ("1", "0"):
        path = str(tmp_path / ("tile%s.npz" % mode))
        env = dict(os.environ, ORBX_DESC_TILE=mode)
        subprocess.run([sys.executable, "-c", script, path], check=True, env=env, timeout=600)
        res[mode] = np.load(path)
    assert sorted(res["1"].files) == sorted(res["0"].files) and len(res["1"].files) == 40
    for k in res["1"].files:
        assert np.array_equal(res["1"][k], res["0"][k]), k
    assert len(res["1"]["dn"]) > 1500 and int(res["1"]["bn"].min()) > 1500
