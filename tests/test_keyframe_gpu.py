"""Device-resident keyframe hand-off (NewKeyFrameMsg, system/messages.rs:19-51): the features a frame
was just given by the extractor stay on the GPU inside an orbx_keyframe, and the searches local mapping runs on them —
guided match, triangulation search, fuse search — give bit-for-bit what the host-buffer entry points give on the
downloaded copies."""
import numpy as np
import pytest

from conftest import records_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chain(pkg):
    import torch
    cam = pkg.CameraModel(**pkg.synth.EUROC_CAMERA)
    h = pkg.Handle(cam, 1000, device=0, max_w=752, max_h=480, max_batch=3)
    imgs = pkg.synth.stereo_batch(91, 0, 3)
    out = h.alloc_batch_outputs(3, 1304)
    h.process_stereo_batch_device(torch.from_numpy(imgs).cuda(), out)
    h.check_status()
    poses = [np.array([1.0, 0, 0, 0, 0.1 * i, 0.01 * i, 0.0]) for i in range(3)]
    kfs = [pkg.KeyFrame.from_batch_outputs(h, out, b, keyframe_id=40 + b, timestamp_ns=1000 + b, pose_wc=poses[b]) for b in range(3)]
    host = [h.unpack_batch_outputs(out, b) for b in range(3)]
    yield h, cam, kfs, host, poses
    for k in kfs:
        k.close()
    h.close()


def test_keyframe_holds_the_extractor_output(chain, pkg):
    h, cam, kfs, host, poses = chain
    for b, kf in enumerate(kfs):
        fl, fr, m, pts, has = host[b]
        kp, desc, p2, h2 = kf.download()
        assert records_equal(kp, fl.keypoints) and np.array_equal(desc, fl.descriptors)
        assert np.array_equal(h2, has) and np.array_equal(p2[has == 1], pts[has == 1])
        info = kf.info()
        assert info["n"] == len(fl.keypoints) and info["keyframe_id"] == 40 + b and info["timestamp_ns"] == 1000 + b
        assert np.array_equal(info["pose_wc"], poses[b]) and kf.map_points() == [None] * kf.n


def test_guided_match_chained_on_device(chain, pkg):
    h, cam, kfs, host, _ = chain
    fl = host[0][0]
    rng = np.random.default_rng(3)
    sel = rng.choice(len(fl.keypoints), 400, replace=False)
    q_uv = np.stack([fl.keypoints["x"][sel], fl.keypoints["y"][sel]], 1).astype(np.float64) + rng.normal(0, 3, (400, 2))
    q_desc = fl.descriptors[sel].copy()
    q_desc[:, 0] ^= rng.integers(0, 256, 400, dtype=np.uint8)
    for mode, radius in ((0, 15.0), (1, 40.0)):
        want = h.guided_match(fl.keypoints, fl.descriptors, 752.0, 480.0, q_uv, q_desc, radius, mode)
        got = kfs[0].guided_match(752.0, 480.0, q_uv, q_desc, radius, mode)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (got[0] >= 0).sum() > 100


def test_triangulation_search_chained_on_device(chain, pkg):
    """new keyframe = the left image of pair 0, its neighbour = the RIGHT image of the same pair one baseline to the side (the
    two views of one scene this synthetic stream has); the right features come straight from the extractor's device output too"""
    h, cam, kfs, host, poses = chain
    out_b0 = host[0]
    fl, fr = out_b0[0], out_b0[1]
    pose_r = np.array([1.0, 0, 0, 0, poses[0][4] + cam.baseline, poses[0][5], poses[0][6]])
    import torch
    kr = torch.from_numpy(np.ascontiguousarray(fr.keypoints).view(np.float32).reshape(-1, 7).copy()).cuda()
    dr = torch.from_numpy(fr.descriptors.copy()).cuda()
    kf_r = pkg.KeyFrame(h, kr, dr, len(fr.keypoints), None, None, keyframe_id=99, pose_wc=pose_r)
    try:
        rng = np.random.default_rng(4)
        mp = []
        for kf in (kfs[0], kf_r):
            ids = [int(7000 + i) if rng.random() < 0.3 else None for i in range(kf.n)]
            kf.set_map_points(ids); mp.append(np.array([i is not None for i in ids], np.uint8))
            assert kf.map_points() == ids
        args = (fl.keypoints, fl.descriptors, mp[0], out_b0[4], fr.keypoints, fr.descriptors, mp[1])
        want = h.search_for_triangulation(cam, *args, poses[0], pose_r, 50)
        got = kfs[0].search_for_triangulation(cam, kf_r, 50)
        assert np.array_equal(got, want) and len(got) > 20
        moved = [1.0, 0, 0, 0, pose_r[4] + 0.05, 0.01, 0.02]               # a refined pose moves the epipolar geometry
        kf_r.set_pose(moved)
        want2 = h.search_for_triangulation(cam, *args, poses[0], moved, 50)
        got2 = kfs[0].search_for_triangulation(cam, kf_r, 50)
        assert np.array_equal(got2, want2) and not np.array_equal(got2, got)
    finally:
        kfs[0].set_map_points([None] * kfs[0].n)
        kf_r.close()


def test_fuse_search_chained_on_device(chain, pkg):
    h, cam, kfs, host, poses = chain
    # map points = the stereo points of frame 0 in world coordinates (pose 0 has identity rotation) with their descriptors
    fl, _, _, pts, has = host[0]
    keep = np.nonzero(has == 1)[0][:300]
    positions = pts[keep] + poses[0][4:]
    mp_desc = fl.descriptors[keep]
    rs = 3.0 * 1.2 ** 7
    off = np.concatenate([[0], np.cumsum([len(host[b][0].keypoints) for b in range(3)])]).astype(np.int32)
    kps = np.concatenate([host[b][0].keypoints for b in range(3)]); descs = np.concatenate([host[b][0].descriptors for b in range(3)])
    want = h.fuse_search(cam, positions, mp_desc, np.array(poses), off, kps, descs, rs, 50)
    got = pkg.KeyFrame.fuse_search(h, cam, positions, mp_desc, kfs, rs, 50)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and (got[0][:, 0] >= 0).sum() > 200
