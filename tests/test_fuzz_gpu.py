"""Randomised parity sweep: odd image sizes, strides, feature counts and image statistics, GPU (through the C ABI)
against the oracle — bit-exact keypoints / descriptors / matches.  Seeds are fixed; every case is reproducible."""
import numpy as np
import pytest

from conftest import records_equal

pytestmark = pytest.mark.gpu


def _image(rng, w, h, kind):
    if kind == 0:      # white noise: a corner almost everywhere, huge candidate lists, many score ties
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 1:      # smooth gradient + sparse salt: few, isolated corners
        y, x = np.mgrid[0:h, 0:w]
        img = ((x * 3 + y * 2) % 256).astype(np.uint8)
        pts = rng.integers(0, [h, w], (max(8, w * h // 500), 2))
        img[pts[:, 0], pts[:, 1]] = 255 - img[pts[:, 0], pts[:, 1]]
        return img
    if kind == 2:      # checkerboard with random cell size: periodic structure, exact Harris ties
        c = int(rng.integers(5, 17))
        y, x = np.mgrid[0:h, 0:w]
        return (((x // c + y // c) % 2) * int(rng.integers(120, 256))).astype(np.uint8)
    # blocks + noise (the bench generator's statistics at a random density)
    img = np.full((h, w), 128, np.int16)
    for _ in range(int(rng.integers(20, 200))):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        img[y0:y0 + int(rng.integers(4, 50)), x0:x0 + int(rng.integers(4, 50))] = int(rng.integers(0, 256))
    img += rng.integers(-6, 7, (h, w), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("seed", range(12))
def test_random_images_sizes_quotas(oracle, pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    w = int(rng.integers(100, 900)); h = int(rng.integers(100, 620))
    n_features = int(rng.choice([50, 300, 1000, 2500]))
    kind = seed % 4
    L = _image(rng, w, h, kind)
    R = np.roll(L, -int(rng.integers(2, 30)), axis=1)          # a disparity shift so that matches exist
    hd = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), n_features, device=0, max_w=1024, max_h=1024, max_batch=1)
    p = oracle.orb_params(n_features)
    okL, odL = oracle.orb_extract(L, p)
    okR, odR = oracle.orb_extract(R, p)
    cap = max(len(okL), len(okR)) + 8
    kpL, dL, kpR, dR, m, pts, has = hd.process_stereo(L, R, cap_kp=cap)
    assert len(kpL) == len(okL) and len(kpR) == len(okR), (w, h, n_features, kind, len(kpL), len(okL))
    assert records_equal(kpL, okL) and records_equal(kpR, okR), (w, h, n_features, kind)
    assert np.array_equal(dL, odL) and np.array_equal(dR, odR)
    m0, p0, h0 = oracle.stereo_match(oracle.Camera(**pkg.synth.EUROC_CAMERA), okL, odL, okR, odR)
    assert records_equal(m0, m) and np.array_equal(h0, has) and np.array_equal(p0[h0 == 1], pts[has == 1])
    # the cross-check matcher and the guided matcher on the same features
    if len(okL) and len(okR):
        assert records_equal(oracle.crosscheck_match(odL, odR), hd.hamming_match_crosscheck(dL, dR))
        uv = np.stack([okR["x"].astype(np.float64) + 5.0, okR["y"].astype(np.float64)], 1)
        for mode in (0, 1):
            i0, d0 = oracle.guided_match(okL, odL, float(w), float(h), uv, odR, 15.0, mode)
            i1, d1 = hd.guided_match(kpL, dL, float(w), float(h), uv, dR, 15.0, mode)
            assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    hd.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_ba_windows(gpu_handle, oracle, pkg, seed):
    rng = np.random.default_rng(2000 + seed)
    # at least two fixed keyframes: with only the anchor fixed the monocular scale of the window is a free gauge
    # direction held by the LM damping alone, and even the oracle's own dense-LU and Schur forms then drift apart
    # by ~1e-3 once outliers are present (compare on well-conditioned windows)
    K = int(rng.integers(3, 16)); M = int(rng.integers(20, 400)); extra = int(rng.integers(1, min(4, K - 1)))
    w = pkg.synth.ba_window(300 + seed, K, M, pkg.BA_OBS, n_fixed_extra=extra, noise_px=float(rng.uniform(0.2, 3.0)))
    # a few gross outliers (Huber region) and a point behind a camera (the (100,100) rule)
    w["obs"]["u"][::37] += 40.0
    w["points"][0, 2] = -1.0
    cam = pkg.CameraModel(**w["camera"])
    g = gpu_handle.ba_solve_visual(cam, pkg.LocalBAConfigLM(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
    ocam = oracle.Camera(**w["camera"])
    o = oracle.ba_solve_dense(ocam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
    o2 = oracle.ba_solve_schur(ocam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
    rel = lambda a, b: np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))
    # 1e-6 relative (north_star) on well-conditioned windows; a window with weakly triangulated points cannot be
    # pinned tighter than the spread between the oracle's own two exact-arithmetic-equivalent formulations
    spread = max(rel(o2["poses_wc"], o["poses_wc"]), rel(o2["points"], o["points"]))
    tol = max(1e-6, 50.0 * spread)
    assert g["iterations"] == o["iterations"], (K, M, extra)
    assert rel(g["poses_wc"], o["poses_wc"]) < tol and rel(g["points"], o["points"]) < tol, (K, M, extra, spread)
    assert abs(g["final_error"] - o["final_error"]) < 1e-7 * max(o["final_error"], 1e-9)
