"""Seeded synthetic inputs of the shapes BASELINE.json names.

There is no EuRoC data in the build or on the GPU box, so every test and bench input is
generated here: EuRoC-shaped stereo pairs (752x480 u8, rectified, cam0 intrinsics), matcher
descriptor sets, and local-BA windows.  Pure numpy; identical bytes feed the HIP path and the
CPU oracle.
"""
import numpy as np

# Public EuRoC cam0 values; the reference reads them from
# mav0/cam0/sensor.yaml (src/io/euroc.rs:329-359).
EUROC_CAMERA = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, baseline=0.11007)


def stereo_pair(seed, frame, w=752, h=480, n_rect=None, n_disc=None):
    """One rectified stereo pair (left, right) of u8 [h, w] images.

    A mid-grey canvas with random axis-aligned rectangles and discs, each at its own disparity
    (2..60 px at 752 wide, scaled with width), painted far-to-near so that nearer objects occlude;
    rows are identical in both views (rectified), per-pixel noise U[-4,4] is independent.
    """
    rng = np.random.default_rng([0x5EED, seed, frame])
    area = (w * h) / (752.0 * 480.0)
    n_rect = int(600 * area) if n_rect is None else n_rect
    n_disc = int(300 * area) if n_disc is None else n_disc
    n = n_rect + n_disc
    sx = w / 752.0
    disp = rng.uniform(2.0, 60.0, n) * sx
    order = np.argsort(disp, kind="stable")          # far first
    cx = rng.uniform(0, w, n)
    cy = rng.uniform(0, h, n)
    sz_w = rng.integers(6, 61, n)
    sz_h = rng.integers(6, 61, n)
    gray = rng.integers(0, 256, n)
    left = np.full((h, w), 128, np.int16)
    right = np.full((h, w), 128, np.int16)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in order:
        x0 = int(cx[i]) - int(sz_w[i]) // 2
        y0 = int(cy[i]) - int(sz_h[i]) // 2
        d = int(round(disp[i]))
        if i < n_rect:
            ww, hh = int(sz_w[i]), int(sz_h[i])
            for img, xs in ((left, x0), (right, x0 - d)):
                xa, xb = max(xs, 0), min(xs + ww, w)
                ya, yb = max(y0, 0), min(y0 + hh, h)
                if xa < xb and ya < yb:
                    img[ya:yb, xa:xb] = gray[i]
        else:
            r = int(sz_w[i]) // 2
            m = (yy[:2 * r + 1, :2 * r + 1] - r) ** 2 + (xx[:2 * r + 1, :2 * r + 1] - r) ** 2 <= r * r
            for img, xs in ((left, x0), (right, x0 - d)):
                xa, xb = max(xs, 0), min(xs + 2 * r + 1, w)
                ya, yb = max(y0, 0), min(y0 + 2 * r + 1, h)
                if xa < xb and ya < yb:
                    sub = m[ya - y0:yb - y0, xa - xs:xb - xs]
                    img[ya:yb, xa:xb][sub] = gray[i]
    left += rng.integers(-4, 5, (h, w), dtype=np.int16)
    right += rng.integers(-4, 5, (h, w), dtype=np.int16)
    return (np.clip(left, 0, 255).astype(np.uint8), np.clip(right, 0, 255).astype(np.uint8))


def stereo_batch(seed, first_frame, batch, w=752, h=480):
    """[batch, 2, h, w] u8: the layout orbx_process_stereo_batch_device takes."""
    out = np.empty((batch, 2, h, w), np.uint8)
    for b in range(batch):
        out[b, 0], out[b, 1] = stereo_pair(seed, first_frame + b, w, h)
    return out


def matcher_features(seed, n_left, n_right, keypoint_dtype, w=752, h=480):
    """Feature sets for the matcher microbench (SURVEY.md §8d 'Value distributions'):
    256 iid Bernoulli(1/2) bits; 70 % of right descriptors are a left one with Binomial(256,0.06)
    bits flipped, yR = yL + U{-1,0,1}, xR = xL - U[1.3,120]; the rest independent."""
    rng = np.random.default_rng([0xFEA7, seed])
    kpL = np.zeros(n_left, keypoint_dtype)
    kpR = np.zeros(n_right, keypoint_dtype)
    kpL["x"] = rng.uniform(31, w - 31, n_left).astype(np.float32)
    kpL["y"] = rng.uniform(31, h - 31, n_left).astype(np.float32)
    descL = rng.integers(0, 256, (n_left, 32), dtype=np.uint8)
    descR = rng.integers(0, 256, (n_right, 32), dtype=np.uint8)
    kpR["x"] = rng.uniform(31, w - 31, n_right).astype(np.float32)
    kpR["y"] = rng.uniform(31, h - 31, n_right).astype(np.float32)
    n_corr = min(int(0.7 * n_right), n_left)
    src = rng.permutation(n_left)[:n_corr]
    dst = rng.permutation(n_right)[:n_corr]
    flips = rng.random((n_corr, 256)) < 0.06
    bits = np.unpackbits(descL[src], axis=1, bitorder="little") ^ flips.astype(np.uint8)
    descR[dst] = np.packbits(bits, axis=1, bitorder="little")
    kpR["y"][dst] = kpL["y"][src] + rng.integers(-1, 2, n_corr).astype(np.float32)
    kpR["x"][dst] = kpL["x"][src] - rng.uniform(1.3, 120.0, n_corr).astype(np.float32)
    for kp in (kpL, kpR):
        kp["size"] = 31.0
        kp["angle"] = rng.uniform(0, 360, len(kp)).astype(np.float32)
        kp["response"] = rng.uniform(0, 1e-3, len(kp)).astype(np.float32)
        kp["octave"] = 0
        kp["class_id"] = -1
    return kpL, descL, kpR, descR


def _quat_from_axis_angle(axis, ang):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], axis * np.sin(ang / 2)])


def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _quat_rot(q, v):
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return v @ R.T


def ba_window(seed, K, M, obs_dtype, n_fixed_extra=0, w=752, h=480, camera=None, noise_px=1.0,
              perturb=True):
    """A local-BA window: keyframe k at (0.15k, 0.02 sin k, 0) m with yaw
    0.01k rad; points U[-6,6]xU[-3,3]xU[3,15] m; an observation wherever the projection lands
    inside the image; pixel noise N(0, noise_px); initial poses perturbed (rot N(0,0.5 deg), trans
    N(0,2 cm)), points N(0,3 cm).  Keyframe 0 is the anchor (fixed, fixed_idx 0); the next
    `n_fixed_extra` keyframes are fixed observers too; the remaining K-1-n_fixed_extra are optimised.

    Returns dict(poses_cw [Kopt,7], fixed_cw [F,7], points [M,3], obs, gt_poses_cw, gt_points).
    """
    cam = dict(EUROC_CAMERA if camera is None else camera)
    rng = np.random.default_rng([0xBA, seed])
    pts = np.stack([rng.uniform(-6, 6, M), rng.uniform(-3, 3, M), rng.uniform(3, 15, M)], 1)
    q_wc, t_wc = [], []
    for k in range(K):
        q_wc.append(_quat_from_axis_angle([0, 1, 0], 0.01 * k))
        t_wc.append(np.array([0.15 * k, 0.02 * np.sin(k), 0.0]))
    poses_cw = []
    for k in range(K):
        qi = q_wc[k] * np.array([1, -1, -1, -1.0])
        ti = -_quat_rot(qi, t_wc[k])
        poses_cw.append(np.concatenate([qi, ti]))
    poses_cw = np.array(poses_cw)
    F = 1 + n_fixed_extra
    obs = []
    for k in range(K):
        pc = _quat_rot(poses_cw[k, :4], pts) + poses_cw[k, 4:]
        u = cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"]
        v = cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]
        vis = (pc[:, 2] > 0.1) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        nz = rng.normal(0, noise_px, (M, 2)) if noise_px > 0 else np.zeros((M, 2))
        for j in np.nonzero(vis)[0]:
            obs.append((k - F if k >= F else -1, k if k < F else -1, j, 0, u[j] + nz[j, 0], v[j] + nz[j, 1]))
    obs = np.array(obs, dtype=obs_dtype)
    init = poses_cw[F:].copy()
    init_pts = pts.copy()
    if perturb:
        for i in range(len(init)):
            ax = rng.normal(0, 1, 3)
            dq = _quat_from_axis_angle(ax, np.deg2rad(rng.normal(0, 0.5)))
            init[i, :4] = _quat_mul(dq, init[i, :4])
            init[i, 4:] += rng.normal(0, 0.02, 3)
        init_pts += rng.normal(0, 0.03, init_pts.shape)
    return dict(poses_cw=init, fixed_cw=poses_cw[:F].copy(), points=init_pts, obs=obs,
                gt_poses_cw=poses_cw[F:].copy(), gt_points=pts, camera=cam)


def two_view_features(seed, n_points, keypoint_dtype, n_distractors=300, camera=None, dup=0.0):
    """Two keyframes looking at the same 3D points (for search_for_triangulation): returns
    dict(kp1, desc1, mp1, stereo1, kp2, desc2, mp2, pose1_wc, pose2_wc, camera).  Shared points get descriptors
    that differ in a few bits; `dup` > 0 makes that fraction of the descriptors identical copies (ties and
    competition for the same partner); flags are random."""
    cam = dict(EUROC_CAMERA if camera is None else camera)
    rng = np.random.default_rng([0x7F1, seed])
    w, h = 2.0 * cam["cx"], 2.0 * cam["cy"]
    X = np.stack([rng.uniform(-5, 5, n_points), rng.uniform(-3, 3, n_points), rng.uniform(4, 14, n_points)], 1)
    pose1 = np.concatenate([_quat_from_axis_angle([0, 1, 0], 0.02), [0.0, 0.0, 0.0]])
    pose2 = np.concatenate([_quat_from_axis_angle([0.1, 1, 0.05], -0.03), [0.45, 0.03, 0.05]])

    def project(pose_wc):
        qi = pose_wc[:4] * np.array([1, -1, -1, -1.0])
        pc = _quat_rot(qi, X - pose_wc[4:])
        return np.stack([cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"], cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]], 1), pc[:, 2]

    uv1, z1 = project(pose1)
    uv2, z2 = project(pose2)
    ok = (z1 > 0.5) & (z2 > 0.5) & (uv1[:, 0] > 0) & (uv1[:, 0] < w) & (uv1[:, 1] > 0) & (uv1[:, 1] < h) & \
         (uv2[:, 0] > 0) & (uv2[:, 0] < w) & (uv2[:, 1] > 0) & (uv2[:, 1] < h)
    uv1, uv2 = uv1[ok], uv2[ok]
    m = len(uv1)
    d = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    if dup > 0:
        k = int(dup * m)
        d[rng.permutation(m)[:k]] = d[rng.integers(0, max(m // 20, 1), k)]
    flips = rng.random((m, 256)) < 0.03
    d2 = np.packbits(np.unpackbits(d, axis=1, bitorder="little") ^ flips.astype(np.uint8), axis=1, bitorder="little")
    out = {}
    for name, uv, dd in (("1", uv1, d), ("2", uv2, d2)):
        nd = n_distractors
        kp = np.zeros(m + nd, keypoint_dtype)
        kp["x"] = np.concatenate([uv[:, 0] + rng.normal(0, 0.4, m), rng.uniform(0, w, nd)]).astype(np.float32)
        kp["y"] = np.concatenate([uv[:, 1] + rng.normal(0, 0.4, m), rng.uniform(0, h, nd)]).astype(np.float32)
        desc = np.concatenate([dd, rng.integers(0, 256, (nd, 32), dtype=np.uint8)])
        perm = rng.permutation(m + nd)
        out["kp" + name] = kp[perm]; out["desc" + name] = desc[perm]
        out["mp" + name] = (rng.random(m + nd) < 0.3).astype(np.uint8)
    out["stereo1"] = (rng.random(len(out["kp1"])) < 0.5).astype(np.uint8)
    out["pose1_wc"] = pose1; out["pose2_wc"] = pose2; out["camera"] = cam
    return out


def fuse_scene(seed, n_points, n_kfs, n_feat, keypoint_dtype, camera=None, far_fraction=0.1):
    """Map points + target keyframes for the fuse search (search_in_neighbors.rs:273-343): keyframe t sits at
    (0.2t, 0.03 sin t, 0) m with a small yaw; every keyframe sees ~60 % of the points as features (pixel noise
    N(0,1.5), ~4 % descriptor bits flipped), the rest of its n_feat features are random.  Some points lie behind
    the cameras or project outside the image; `far_fraction` of them sit 400-4000 m away so that the depth-scaled
    radius leaves its lower clamp.  Returns dict(positions, mp_desc, kf_poses_wc, kf_feat_offset, kps, descs, camera)."""
    cam = dict(EUROC_CAMERA if camera is None else camera)
    rng = np.random.default_rng([0xF05E, seed])
    w, h = 2.0 * cam["cx"], 2.0 * cam["cy"]
    z = rng.uniform(2, 20, n_points)
    far = rng.random(n_points) < far_fraction
    z[far] = rng.uniform(400, 4000, far.sum())
    z[rng.random(n_points) < 0.05] *= -1.0
    X = np.stack([rng.uniform(-0.9, 1.1, n_points) * z * cam["cx"] / cam["fx"], rng.uniform(-0.9, 1.1, n_points) * z * cam["cy"] / cam["fy"], z], 1)
    mp_desc = rng.integers(0, 256, (n_points, 32), dtype=np.uint8)
    poses, kps, descs, off = [], [], [], [0]
    for t in range(n_kfs):
        pose = np.concatenate([_quat_from_axis_angle([0.05, 1, 0.02], 0.015 * t), [0.2 * t, 0.03 * np.sin(t), 0.0]])
        poses.append(pose)
        qi = pose[:4] * np.array([1, -1, -1, -1.0])
        pc = _quat_rot(qi, X - pose[4:])
        with np.errstate(divide="ignore", invalid="ignore"):
            u = cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"]; v = cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]
        vis = np.nonzero((pc[:, 2] > 0) & (u >= 0) & (u < w) & (v >= 0) & (v < h) & (rng.random(n_points) < 0.6))[0]
        vis = vis[:n_feat]
        m = len(vis)
        kp = np.zeros(n_feat, keypoint_dtype)
        kp["x"] = np.concatenate([u[vis] + rng.normal(0, 1.5, m), rng.uniform(0, w, n_feat - m)]).astype(np.float32)
        kp["y"] = np.concatenate([v[vis] + rng.normal(0, 1.5, m), rng.uniform(0, h, n_feat - m)]).astype(np.float32)
        kp["octave"] = rng.integers(0, 8, n_feat)
        flips = rng.random((m, 256)) < 0.04
        d = np.packbits(np.unpackbits(mp_desc[vis], axis=1, bitorder="little") ^ flips.astype(np.uint8), axis=1, bitorder="little")
        d = np.concatenate([d, rng.integers(0, 256, (n_feat - m, 32), dtype=np.uint8)])
        perm = rng.permutation(n_feat)
        kps.append(kp[perm]); descs.append(d[perm]); off.append(off[-1] + n_feat)
    return dict(positions=X, mp_desc=mp_desc, kf_poses_wc=np.array(poses), kf_feat_offset=np.array(off, np.int32),
                kps=np.concatenate(kps) if kps else np.zeros(0, keypoint_dtype),
                descs=np.concatenate(descs) if descs else np.zeros((0, 32), np.uint8), camera=cam)


def vocabulary(seed, k=10, depth=3, ragged=False):
    """A synthetic DBoW2-style vocabulary tree (the real ORBvoc.txt is not in the build): node 0 is the root; every
    inner node has k children whose descriptors are the parent's with ~12 % of the bits flipped (so that descents are
    meaningful), leaves carry an IDF-like weight.  Nodes are numbered in the order a DBoW2 text file lists them
    (depth-first, children contiguous).  `ragged`: child counts vary from 1 to 2k (more than the 16 lanes a
    descriptor gets), some branches stop early, a few nodes name a parent that comes later (never linked).
    Returns (parent u32 [n], is_leaf u8 [n], desc u8 [n,32], weight f64 [n])."""
    rng = np.random.default_rng([0xB0, seed])
    parent, leaf, desc, weight = [0], [0], [np.zeros(32, np.uint8)], [0.0]

    def grow(pid, pdesc, level):
        nc = int(rng.integers(1, 2 * k + 1)) if ragged else k
        for _ in range(nc):
            flips = rng.random(256) < 0.12
            d = np.packbits(np.unpackbits(pdesc, bitorder="little") ^ flips.astype(np.uint8), bitorder="little")
            nid = len(parent)
            stop = level + 1 >= depth or (ragged and level >= 1 and rng.random() < 0.2)
            parent.append(pid); leaf.append(1 if stop else 0); desc.append(d)
            weight.append(float(rng.uniform(0.5, 12.0)) if stop else 0.0)
            if not stop:
                grow(nid, d, level + 1)

    grow(0, rng.integers(0, 256, 32, dtype=np.uint8), 0)
    parent = np.array(parent, np.uint32); leaf = np.array(leaf, np.uint8)
    if ragged:   # forward references: load_from_text never links these (mod.rs:196-198)
        for nid in rng.permutation(np.arange(1, len(parent) - 5))[:5]:
            parent[nid] = nid + 3
    return parent, leaf, np.array(desc, np.uint8), np.array(weight, np.float64)


def write_vocabulary_text(path, parent, is_leaf, desc, weight, k=10, depth=3, junk_lines=True):
    """DBoW2 text format as load_from_text reads it (mod.rs:101-116): header 'k L scoring weighting', then
    'parent_id is_leaf d0..d31 weight' per node (root excluded).  junk_lines adds short lines that the loader skips."""
    with open(path, "w") as f:
        f.write("%d %d 0 0\n" % (k, depth))
        for i in range(1, len(parent)):
            if junk_lines and i % 97 == 0:
                f.write("# %d short line\n\n" % i)
            f.write("%d %d %s %r \n" % (parent[i], is_leaf[i], " ".join(str(int(b)) for b in desc[i]), float(weight[i])))


def png_encode(img, filters="cycle", bit_depth=8, alpha=False, level=6):
    """A PNG file (bytes) of a greyscale image, written with zlib only: `filters` = 'cycle' (row y uses filter type
    y % 5, exercising None/Sub/Up/Average/Paeth), 'none', or an int 0..4; bit_depth 8 or 16 (16: value*257, so the
    high byte is the 8-bit value); alpha adds an opaque alpha channel (colour type 4)."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    px = img.astype(np.uint16)
    chans = [px * 257 if bit_depth == 16 else px]
    if alpha:
        chans.append(np.full_like(px, 65535 if bit_depth == 16 else 255))
    inter = np.stack(chans, -1)                                            # [h, w, c]
    raw = inter.astype(">u2").view(np.uint8).reshape(h, -1) if bit_depth == 16 else inter.astype(np.uint8).reshape(h, -1)
    bpp = len(chans) * (bit_depth // 8)
    rows = []
    prev = np.zeros(raw.shape[1], np.int32)
    for y in range(h):
        cur = raw[y].astype(np.int32)
        ft = y % 5 if filters == "cycle" else (0 if filters == "none" else int(filters))
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - a
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((a + prev) >> 1)
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            f = cur - pred
        rows.append(bytes([ft]) + (f & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, bit_depth, 4 if alpha else 0, 0, 0, 0)
    z = zlib.compress(b"".join(rows), level)
    half = len(z) // 2                                                     # two IDAT chunks: the stream may be split anywhere
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"tEXt", b"Comment\x00synthetic") + chunk(b"IDAT", z[:half]) + \
        chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


def write_euroc_mav0(root, n_frames, seed=0, w=752, h=480, camera=None):
    """A synthetic EuRoC `mav0` directory (cam0/cam1 data.csv + data/*.png + sensor.yaml in the dataset's own layout)
    holding stereo_pair(seed, i) for i < n_frames.  Returns the list of (left, right) arrays written."""
    import os
    cam = dict(EUROC_CAMERA if camera is None else camera)
    pairs = []
    t0 = 1403636579763555584
    # T_BS of the two cameras: cam1 displaced by the baseline along cam0's x axis, both slightly rotated vs. the body
    R = _quat_rot(_quat_from_axis_angle([0.2, -0.1, 1.0], 1.57), np.eye(3)).T
    t_b_c0 = np.array([-0.0216, -0.0647, 0.0098])
    t_b_c1 = t_b_c0 + R @ np.array([cam["baseline"], 0.0, 0.0])
    for c, t in ((0, t_b_c0), (1, t_b_c1)):
        d = os.path.join(root, "cam%d" % c, "data")
        os.makedirs(d, exist_ok=True)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        with open(os.path.join(root, "cam%d" % c, "sensor.yaml"), "w") as f:
            f.write("# General sensor definitions.\nsensor_type: camera\ncomment: VI-Sensor cam%d (MT9M034)\n\n" % c)
            f.write("# Sensor extrinsics wrt. the body-frame.\nT_BS:\n  cols: 4\n  rows: 4\n  data: [")
            f.write(",\n         ".join(", ".join(repr(float(v)) for v in T[r]) for r in range(4)))
            f.write("]\n\n# Camera specific definitions.\nrate_hz: 20\nresolution: [%d, %d]\ncamera_model: pinhole\n" % (w, h))
            f.write("intrinsics: [%r, %r, %r, %r] #fu, fv, cu, cv\n" % (cam["fx"], cam["fy"], cam["cx"], cam["cy"]))
            f.write("distortion_model: radial-tangential\ndistortion_coefficients: [0.0, 0.0, 0.0, 0.0]\n")
    rows = []
    for i in range(n_frames):
        L, Rr = stereo_pair(seed, i, w, h)
        pairs.append((L, Rr))
        ts = t0 + 50000000 * i
        name = "%d.png" % ts
        rows.append("%d,%s" % (ts, name))
        for c, img in ((0, L), (1, Rr)):
            with open(os.path.join(root, "cam%d" % c, "data", name), "wb") as f:
                f.write(png_encode(img, filters="cycle" if i % 2 == 0 else (i % 5)))
    for c in (0, 1):
        with open(os.path.join(root, "cam%d" % c, "data.csv"), "w") as f:
            f.write("#timestamp [ns],filename\n" + "\r\n".join(rows) + "\n")
    return pairs


def inertial_window(seed, K, M, obs_dtype, n_fixed=2, dt=0.25, w=752, h=480, camera=None, noise_px=1.0, stereo_fraction=0.5):
    """A local inertial-BA window (local_inertial_ba.rs): K consecutive keyframes dt apart on a smooth accelerating
    trajectory (T_wc poses, world velocities, per-keyframe biases), `n_fixed` older keyframes that only observe (T_cw),
    M points.  The preintegrated deltas of edge (k, k+1) are the exact ones of the ground-truth states
    (dR = Ri^T Rj, dv = Ri^T (vj - vi - g dt), dp = Ri^T (pj - pi - vi dt - g dt^2 / 2), imu_factors.rs:66-103) plus a
    little noise; initial states are perturbed.  obs[i]["_pad"] bit 0 marks stereo observations.
    Returns dict(poses_wc, velocities, biases, fixed_cw, points, obs, edge_kf, preint, camera, gt_*)."""
    cam = dict(EUROC_CAMERA if camera is None else camera)
    rng = np.random.default_rng([0x1BA, seed])
    g = np.array([0.0, 0.0, -9.81])
    tt = (np.arange(-n_fixed, K)) * dt

    def pos(t):
        return np.array([0.6 * t + 0.05 * t * t, 0.08 * np.sin(1.3 * t), 0.03 * np.cos(0.9 * t)])

    def velo(t):
        return np.array([0.6 + 0.1 * t, 0.08 * 1.3 * np.cos(1.3 * t), -0.03 * 0.9 * np.sin(0.9 * t)])

    q_wc = [_quat_mul(_quat_from_axis_angle([0, 1, 0], 0.04 * t), _quat_from_axis_angle([1, 0, 0], 0.02 * np.sin(t))) for t in tt]
    p_wc = [pos(t) for t in tt]
    v_w = [velo(t) for t in tt]
    pts = np.stack([rng.uniform(-5, 7, M), rng.uniform(-3, 3, M), rng.uniform(3, 14, M)], 1)
    obs = []
    for idx in range(len(tt)):
        qi = q_wc[idx] * np.array([1, -1, -1, -1.0])
        pc = _quat_rot(qi, pts - p_wc[idx])
        u = cam["fx"] * pc[:, 0] / pc[:, 2] + cam["cx"]; v = cam["fy"] * pc[:, 1] / pc[:, 2] + cam["cy"]
        vis = (pc[:, 2] > 0.1) & (u >= 0) & (u < w) & (v >= 0) & (v < h)
        nz = rng.normal(0, noise_px, (M, 2)) if noise_px > 0 else np.zeros((M, 2))
        st = rng.random(M) < stereo_fraction
        for j in np.nonzero(vis)[0]:
            k = idx - n_fixed
            obs.append((k if k >= 0 else -1, idx if k < 0 else -1, j, int(st[j]), u[j] + nz[j, 0], v[j] + nz[j, 1]))
    obs = np.array(obs, dtype=obs_dtype)
    fixed_cw = []
    for idx in range(n_fixed):
        qi = q_wc[idx] * np.array([1, -1, -1, -1.0])
        fixed_cw.append(np.concatenate([qi, -_quat_rot(qi, p_wc[idx])]))
    gt_poses = np.array([np.concatenate([q_wc[n_fixed + k], p_wc[n_fixed + k]]) for k in range(K)])
    gt_vel = np.array(v_w[n_fixed:])
    edge_kf, preint = [], []
    for k in range(K - 1):
        qi, qj = gt_poses[k, :4], gt_poses[k + 1, :4]
        qic = qi * np.array([1, -1, -1, -1.0])
        dR = _quat_mul(qic, qj)
        dv = _quat_rot(qic, gt_vel[k + 1] - gt_vel[k] - g * dt)
        dp = _quat_rot(qic, gt_poses[k + 1, 4:] - gt_poses[k, 4:] - gt_vel[k] * dt - 0.5 * g * dt * dt)
        dR = _quat_mul(dR, _quat_from_axis_angle(rng.normal(0, 1, 3), rng.normal(0, 2e-3)))
        edge_kf.append((k, k + 1))
        preint.append(np.concatenate([dR / np.linalg.norm(dR), dv + rng.normal(0, 5e-3, 3), dp + rng.normal(0, 2e-3, 3), [dt]]))
    poses = gt_poses.copy()
    for k in range(K):
        dq = _quat_from_axis_angle(rng.normal(0, 1, 3), np.deg2rad(rng.normal(0, 0.4)))
        poses[k, :4] = _quat_mul(dq, poses[k, :4])
        poses[k, 4:] += rng.normal(0, 0.02, 3)
    vel = gt_vel + rng.normal(0, 0.05, (K, 3))
    bias = np.concatenate([rng.normal(0, 2e-3, (K, 3)), rng.normal(0, 2e-2, (K, 3))], 1)
    return dict(poses_wc=poses, velocities=vel, biases=bias, fixed_cw=np.array(fixed_cw).reshape(-1, 7), points=pts + rng.normal(0, 0.03, (M, 3)),
                obs=obs, edge_kf=np.array(edge_kf, np.int32).reshape(-1, 2), preint=np.array(preint).reshape(-1, 11), camera=cam,
                gt_poses_wc=gt_poses, gt_velocities=gt_vel, gt_points=pts)


def keypoint_precision(window):
    """The window with its pixel coordinates rounded to f32 and widened again — what the reference's observations are: kp.pt() is a
    cv::Point2f, widened to f64 when the problem is collected (local_ba_lm.rs:870-872)."""
    w = dict(window)
    o = np.array(window["obs"], copy=True)
    o["u"] = o["u"].astype(np.float32).astype(np.float64); o["v"] = o["v"].astype(np.float32).astype(np.float64)
    w["obs"] = o
    return w


def write_ba_batch_file(path, windows, obs_dtype):
    """`windows` (dicts of ba_window) in the layout tests/cpp/ba_batch_driver.cpp reads: int32 W, then per window int32 K, F, M, N |
    poses_cw [K][7] | fixed_cw [F][7] | points [M][3] | obs [N] (orbx_ba_obs)."""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(windows)))
        for w in windows:
            poses = np.ascontiguousarray(w["poses_cw"], np.float64).reshape(-1, 7)
            fixed = np.ascontiguousarray(w["fixed_cw"], np.float64).reshape(-1, 7)
            pts = np.ascontiguousarray(w["points"], np.float64).reshape(-1, 3)
            obs = np.ascontiguousarray(w["obs"], obs_dtype)
            f.write(struct.pack("<iiii", len(poses), len(fixed), len(pts), len(obs)))
            for a in (poses, fixed, pts, obs):
                f.write(a.tobytes())


def read_ba_batch_results(path, windows):
    """out.bin of tests/cpp/ba_batch_driver.cpp -> list of dicts (status, iterations, initial_error, final_error, poses_wc, points)."""
    import struct
    out = []
    with open(path, "rb") as f:
        for w in windows:
            K = len(np.asarray(w["poses_cw"]).reshape(-1, 7)); M = len(np.asarray(w["points"]).reshape(-1, 3))
            status, iters = struct.unpack("<ii", f.read(8))
            e0, e1 = struct.unpack("<dd", f.read(16))
            poses = np.frombuffer(f.read(56 * K), np.float64).reshape(K, 7)
            pts = np.frombuffer(f.read(24 * M), np.float64).reshape(M, 3)
            out.append(dict(status=status, iterations=iters, initial_error=e0, final_error=e1, poses_wc=poses, points=pts))
    return out
