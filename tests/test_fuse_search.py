"""The search part of fuse_points_into_keyframes (reference src/local_mapping/search_in_neighbors.rs:273-343,
KeyFrame::get_features_in_area src/atlas/map/keyframe.rs:408-443).
CPU: oracle vs an independent numpy restatement.  GPU: HIP path vs oracle, bit-exact."""
import numpy as np
import pytest

import orb_slam3_rust_amd as P
from oracle import oracle as O

RADIUS_SCALE = 3.0 * (1.2 * (1.2 * 1.2) * ((1.2 * 1.2) * (1.2 * 1.2)))   # radius_factor * 1.2.powi(7), :303 (compiler-rt __powidf2 order)


def _scene(seed, n_points, n_kfs, n_feat, **kw):
    return P.synth.fuse_scene(seed, n_points, n_kfs, n_feat, O.KEYPOINT, **kw)


def _oracle(s, thr=50, scale=RADIUS_SCALE):
    return O.fuse_search(O.Camera(**s["camera"]), s["positions"], s["mp_desc"], s["kf_poses_wc"], s["kf_feat_offset"], s["kps"],
                         s["descs"], scale, thr)


def _numpy(s, thr=50, scale=RADIUS_SCALE):
    cam = s["camera"]
    Pn, T = len(s["positions"]), len(s["kf_poses_wc"])
    idx = np.full((Pn, T), -1, np.int32); dist = np.zeros((Pn, T), np.uint32)
    for t in range(T):
        q = s["kf_poses_wc"][t]
        w, x, y, z = q[:4]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        pc = (s["positions"] - q[4:]) @ R
        a, b = s["kf_feat_offset"][t], s["kf_feat_offset"][t + 1]
        kx, ky = s["kps"]["x"][a:b].astype(np.float64), s["kps"]["y"][a:b].astype(np.float64)
        for p in range(Pn):
            if pc[p, 2] <= 0:
                continue
            u = cam["fx"] * pc[p, 0] / pc[p, 2] + cam["cx"]; v = cam["fy"] * pc[p, 1] / pc[p, 2] + cam["cy"]
            if u < 0 or u >= 2 * cam["cx"] or v < 0 or v >= 2 * cam["cy"]:
                continue
            r = max(min(scale * pc[p, 2] / cam["fx"], 50.0), 10.0)
            c = np.nonzero((kx - u) ** 2 + (ky - v) ** 2 <= r * r)[0]
            if len(c) == 0:
                continue
            d = np.unpackbits(s["descs"][a:b][c] ^ s["mp_desc"][p], axis=1).sum(1)
            k = int(np.argmin(d))
            if d[k] < thr:
                idx[p, t] = c[k]; dist[p, t] = d[k]
    return idx, dist


def test_oracle_matches_numpy_restatement():
    s = _scene(1, 400, 5, 500)
    i0, d0 = _oracle(s)
    i1, d1 = _numpy(s)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert (i0 >= 0).sum() > 300
    far = np.abs(s["positions"][:, 2]) > 300
    assert (i0[far] >= 0).any()                       # the radius leaves its lower clamp for these


def test_oracle_semantics():
    s = _scene(2, 300, 3, 400)
    idx, dist = _oracle(s)
    assert (idx[s["positions"][:, 2] < -1.0] == -1).all()        # behind the camera (:286)
    assert dist[idx >= 0].max() < 50 and (dist[idx < 0] == 0).all()
    assert (_oracle(s, thr=0)[0] == -1).all()
    # ties go to the lowest feature index: duplicate every feature of keyframe 0 behind the originals
    n0 = s["kf_feat_offset"][1]
    s2 = dict(s)
    s2["kps"] = np.concatenate([s["kps"][:n0], s["kps"][:n0], s["kps"][n0:]])
    s2["descs"] = np.concatenate([s["descs"][:n0], s["descs"][:n0], s["descs"][n0:]])
    s2["kf_feat_offset"] = np.concatenate([[0], s["kf_feat_offset"][1:] + n0]).astype(np.int32)
    idx2, _ = _oracle(s2)
    assert np.array_equal(idx2, idx)


# ---- GPU parity ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def handle():
    h = P.Handle(P.CameraModel(**P.synth.EUROC_CAMERA), 1200)
    yield h
    h.close()


def _gpu(handle, s, thr=50, scale=RADIUS_SCALE):
    return handle.fuse_search(P.CameraModel(**s["camera"]), s["positions"], s["mp_desc"], s["kf_poses_wc"], s["kf_feat_offset"],
                              s["kps"], s["descs"], scale, thr)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_points,n_kfs,n_feat", [(1, 400, 5, 500), (2, 3000, 20, 1200), (3, 257, 1, 2049), (4, 5000, 30, 4500)])
def test_gpu_matches_oracle(handle, seed, n_points, n_kfs, n_feat):
    s = _scene(seed, n_points, n_kfs, n_feat)
    i0, d0 = _oracle(s)
    i1, d1 = _gpu(handle, s)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert (i0 >= 0).sum() > 100


@pytest.mark.gpu
def test_gpu_edge_cases(handle):
    s = _scene(5, 600, 4, 700, far_fraction=0.5)
    for thr, scale in ((0, RADIUS_SCALE), (256, RADIUS_SCALE), (50, 0.0), (50, 1e6)):
        i0, d0 = _oracle(s, thr, scale)
        i1, d1 = _gpu(handle, s, thr, scale)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    # ragged: keyframes without features, no map points, no keyframes
    r = dict(s)
    off = s["kf_feat_offset"].copy(); off[2:] = off[2]
    r["kf_feat_offset"] = off
    r["kps"], r["descs"] = s["kps"][:off[-1]], s["descs"][:off[-1]]
    i0, d0 = _oracle(r)
    i1, d1 = _gpu(handle, r)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1) and (i1[:, 2:] == -1).all()
    e = dict(s); e["positions"] = s["positions"][:0]; e["mp_desc"] = s["mp_desc"][:0]
    assert _gpu(handle, e)[0].shape == (0, 4)
    e = dict(s); e["kf_poses_wc"] = s["kf_poses_wc"][:0]; e["kf_feat_offset"] = np.zeros(1, np.int32); e["kps"] = s["kps"][:0]; e["descs"] = s["descs"][:0]
    assert _gpu(handle, e)[0].shape == (600, 0)
    bad = dict(s); bad["kf_feat_offset"] = s["kf_feat_offset"][::-1].copy()
    with pytest.raises(P.OrbxError):
        _gpu(handle, bad)
