"""Size-independent properties at the full bench size (128 stereo pairs of 752x480, 2000 ORB): determinism,
independence of a pair from its batch neighbours, orderings, and consistency between entry points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_batch(pkg):
    import torch
    B, cap = 128, 2304
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 2000, device=0, max_w=752, max_h=480, max_batch=B)
    base = torch.from_numpy(pkg.synth.stereo_batch(77, 0, 16)).cuda()
    imgs = torch.cat([base, torch.flip(base, dims=[2]), (base.to(torch.int16) + 5).clamp_(0, 255).to(torch.uint8),
                      torch.flip(base, dims=[2, 3])[:, [1, 0]]] * 2)[:B].contiguous()
    out = h.alloc_batch_outputs(B, cap)
    h.process_stereo_batch_device(imgs, out)
    h.check_status()
    snap = {k: v.clone() for k, v in out.items() if hasattr(v, "shape")}
    yield h, imgs, out, snap, B, cap
    h.close()


def test_rerun_is_bitwise_identical(full_batch):
    import torch
    h, imgs, out, snap, B, cap = full_batch
    for _ in range(2):
        h.process_stereo_batch_device(imgs, out)
        h.check_status()
        for k, v in snap.items():
            if k in ("kp", "desc", "matches", "points", "has_point"):
                continue   # compared below up to the valid counts (slots beyond the counts are scratch)
            assert torch.equal(out[k], v), k
        nk = snap["nkp"].cpu().numpy(); nm = snap["nmatches"].cpu().numpy()
        for b in (0, 17, 64, 127):
            for s in range(2):
                assert torch.equal(out["kp"][b, s, :nk[b, s]].view(torch.int32), snap["kp"][b, s, :nk[b, s]].view(torch.int32))
                assert torch.equal(out["desc"][b, s, :nk[b, s]], snap["desc"][b, s, :nk[b, s]])
            assert torch.equal(out["matches"][b, :nm[b]], snap["matches"][b, :nm[b]])


def test_large_batch_launch_shapes_equal_small_batch_ones(full_batch, pkg, oracle):
    """A 128-pair batch runs FAST with three tiles per block and the resize with six rows per thread; a single pair runs one tile per
    block and two rows (launch_orb_extract picks by batch size).  Same pair, both ways, bit for bit — and one of them against the oracle."""
    import torch
    h, imgs, out, snap, B, cap = full_batch
    h1 = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 2000, device=0, max_w=752, max_h=480, max_batch=1)
    try:
        o1 = h1.alloc_batch_outputs(1, cap)
        for b in (0, 41, 127):
            h1.process_stereo_batch_device(imgs[b:b + 1].contiguous(), o1)
            h1.check_status()
            big = h.unpack_batch_outputs(snap, b)
            one = h1.unpack_batch_outputs(o1, 0)
            for x, y in zip(big[:2], one[:2]):
                assert np.array_equal(x.keypoints.view(np.uint8), y.keypoints.view(np.uint8)) and np.array_equal(x.descriptors, y.descriptors)
            assert np.array_equal(big[2].view(np.uint8), one[2].view(np.uint8))
            assert np.array_equal(big[4], one[4]) and np.array_equal(big[3][big[4] == 1], one[3][one[4] == 1])
        p = oracle.orb_params(2000)
        img = imgs[41, 0].cpu().numpy()
        ok, od = oracle.orb_extract(img, p)
        fl = h.unpack_batch_outputs(snap, 41)[0]
        assert np.array_equal(fl.descriptors, od) and len(fl.keypoints) == len(ok)
    finally:
        h1.close()


@pytest.mark.parametrize("w,hh,B", [(640, 480, 32), (1241, 376, 24), (333, 257, 16), (1920, 1080, 8)])
def test_large_batch_equals_single_pair_other_sizes(pkg, w, hh, B):
    """the batch-size dependent launch shapes (FAST chains, resize rows per thread) at other image sizes: pairs of a batch equal the
    same pairs processed alone, bit for bit"""
    import torch
    rng = np.random.default_rng(w * 7 + hh)
    base = pkg.synth.stereo_batch(91, 0, 4)                                   # 4 pairs of 752x480 scenes, resampled by cropping / tiling
    reps_y, reps_x = (hh + 479) // 480, (w + 751) // 752
    big = np.tile(base, (1, 1, reps_y, reps_x))[:, :, :hh, :w]
    imgs_np = np.concatenate([big] * ((B + 3) // 4))[:B].copy()
    imgs_np[1::2] = imgs_np[1::2, :, ::-1, :]                                 # vary the content a little across the batch
    imgs_np = np.clip(imgs_np.astype(np.int16) + rng.integers(-3, 4, (B, 1, 1, 1)), 0, 255).astype(np.uint8)
    cam = pkg.CameraModel(**pkg.synth.EUROC_CAMERA)
    hb = pkg.Handle(cam, 1500, device=0, max_w=w, max_h=hh, max_batch=B)
    h1 = pkg.Handle(cam, 1500, device=0, max_w=w, max_h=hh, max_batch=1)
    try:
        imgs = torch.from_numpy(imgs_np).cuda()
        ob = hb.alloc_batch_outputs(B, 2048)
        hb.process_stereo_batch_device(imgs, ob); hb.check_status()
        o1 = h1.alloc_batch_outputs(1, 2048)
        for b in (0, B // 2 + 1, B - 1):
            h1.process_stereo_batch_device(imgs[b:b + 1].contiguous(), o1); h1.check_status()
            big_r = hb.unpack_batch_outputs(ob, b); one = h1.unpack_batch_outputs(o1, 0)
            assert len(big_r[0].keypoints) > 100
            for x, y in zip(big_r[:2], one[:2]):
                assert np.array_equal(x.keypoints.view(np.uint8), y.keypoints.view(np.uint8)) and np.array_equal(x.descriptors, y.descriptors)
            assert np.array_equal(big_r[2].view(np.uint8), one[2].view(np.uint8)) and np.array_equal(big_r[4], one[4])
    finally:
        hb.close(); h1.close()


def test_optional_blur_fork_gives_identical_results(full_batch):
    """ORBX_FORK_BLUR=1 runs the blur on a second stream beside the FAST chain (read at launch time): same bytes out."""
    import os
    import torch
    h, imgs, out, snap, B, cap = full_batch
    os.environ["ORBX_FORK_BLUR"] = "1"
    try:
        for _ in range(3):
            h.process_stereo_batch_device(imgs, out)
        h.check_status()
    finally:
        del os.environ["ORBX_FORK_BLUR"]
    nk = snap["nkp"].cpu().numpy(); nm = snap["nmatches"].cpu().numpy()
    assert torch.equal(out["nkp"], snap["nkp"]) and torch.equal(out["nmatches"], snap["nmatches"])
    for b in range(0, B, 9):
        for s in range(2):
            assert torch.equal(out["kp"][b, s, :nk[b, s]].view(torch.int32), snap["kp"][b, s, :nk[b, s]].view(torch.int32))
            assert torch.equal(out["desc"][b, s, :nk[b, s]], snap["desc"][b, s, :nk[b, s]])
        assert torch.equal(out["matches"][b, :nm[b]], snap["matches"][b, :nm[b]])


def test_two_stream_ranges_give_identical_results(full_batch):
    """ORBX_STAGGER=<n> (read at call time): the batch as n ranges of pairs on two streams, each range started behind the previous one's
    pyramid launches (a measured negative for throughput, kept opt-in): same bytes out as the one-stream call."""
    import os
    import torch
    h, imgs, out, snap, B, cap = full_batch
    nk = snap["nkp"].cpu().numpy(); nm = snap["nmatches"].cpu().numpy()
    for n in ("2", "3"):
        os.environ["ORBX_STAGGER"] = n
        try:
            for _ in range(2):
                h.process_stereo_batch_device(imgs, out)
            h.check_status()
        finally:
            del os.environ["ORBX_STAGGER"]
        assert torch.equal(out["nkp"], snap["nkp"]) and torch.equal(out["nmatches"], snap["nmatches"])
        for b in range(0, B, 7):
            for s in range(2):
                assert torch.equal(out["kp"][b, s, :nk[b, s]].view(torch.int32), snap["kp"][b, s, :nk[b, s]].view(torch.int32))
                assert torch.equal(out["desc"][b, s, :nk[b, s]], snap["desc"][b, s, :nk[b, s]])
            assert torch.equal(out["matches"][b, :nm[b]], snap["matches"][b, :nm[b]])
            assert torch.equal(out["has_point"][b, :nk[b, 0]], snap["has_point"][b, :nk[b, 0]])


def test_pair_result_independent_of_batch_position(full_batch, pkg):
    """reversing the batch order permutes the results and nothing else (no cross-talk through shared workspaces,
    the XCD-aware block mapping or the atomically appended candidate lists)"""
    import torch
    h, imgs, out, snap, B, cap = full_batch
    out2 = h.alloc_batch_outputs(B, cap)
    h.process_stereo_batch_device(torch.flip(imgs, dims=[0]).contiguous(), out2)
    h.check_status()
    assert torch.equal(torch.flip(out2["nkp"], dims=[0]), snap["nkp"])
    assert torch.equal(torch.flip(out2["nmatches"], dims=[0]), snap["nmatches"])
    nk = snap["nkp"].cpu().numpy(); nm = snap["nmatches"].cpu().numpy()
    for b in range(0, B, 7):
        r = B - 1 - b
        for s in range(2):
            n = nk[b, s]
            assert torch.equal(out2["kp"][r, s, :n].view(torch.int32), snap["kp"][b, s, :n].view(torch.int32))
            assert torch.equal(out2["desc"][r, s, :n], snap["desc"][b, s, :n])
        assert torch.equal(out2["matches"][r, :nm[b]], snap["matches"][b, :nm[b]])
        hp = snap["has_point"][b, :nk[b, 0]].bool()
        assert torch.equal(out2["points"][r, :nk[b, 0]][hp], snap["points"][b, :nk[b, 0]][hp])


def test_output_orderings_and_ranges(full_batch, pkg):
    h, imgs, out, snap, B, cap = full_batch
    for b in (0, 31, 100):
        fl, fr, m, pts, has = h.unpack_batch_outputs(out, b)
        for f in (fl, fr):
            k = f.keypoints
            assert np.all(np.diff(k["octave"]) >= 0)                         # levels concatenated 0..7
            for l in range(8):
                r = k["response"][k["octave"] == l]
                assert np.all(r[:-1] >= r[1:])                               # canonical order inside a level
            assert np.all((k["angle"] >= 0) & (k["angle"] <= 360.0)) and np.all(k["class_id"] == -1)
            assert k["x"].min() >= 31 and k["x"].max() < 752 - 31 + 1 and k["y"].min() >= 31 and k["y"].max() < 480 - 31 + 1
        assert np.all(np.diff(m["query_idx"]) > 0)                           # ascending, one match per left keypoint
        assert np.all((m["distance"] >= 0) & (m["distance"] < 100)) and np.all(m["img_idx"] == 0)
        kl, kr = fl.keypoints, fr.keypoints
        assert np.all(np.abs(kl["y"][m["query_idx"]] - kr["y"][m["train_idx"]]) <= 2.0)      # stereo.rs:117
        assert np.all(kl["x"][m["query_idx"]] > kr["x"][m["train_idx"]])                      # stereo.rs:127
        # matched distance = Hamming distance of the two rows (recomputed on the GPU through another entry point)
        d = h.hamming_batch(fl.descriptors[m["query_idx"]], fr.descriptors[m["train_idx"]])
        assert np.array_equal(d.astype(np.float32), m["distance"])
        # every triangulated point re-projects onto its left keypoint (stereo.rs:204-211 inverted)
        cam = pkg.synth.EUROC_CAMERA
        q = has == 1
        u = cam["fx"] * pts[q, 0] / pts[q, 2] + cam["cx"]; v = cam["fy"] * pts[q, 1] / pts[q, 2] + cam["cy"]
        assert np.allclose(u, kl["x"][q], atol=1e-6) and np.allclose(v, kl["y"][q], atol=1e-6) and np.all(pts[q, 2] > 0)
        assert q.sum() <= len(m)


def test_hamming_metric_properties(gpu_handle):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (4096, 32), dtype=np.uint8); b = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    c = rng.integers(0, 256, (4096, 32), dtype=np.uint8)
    dab, dba, dac, dcb = (gpu_handle.hamming_batch(x, y) for x, y in ((a, b), (b, a), (a, c), (c, b)))
    assert np.array_equal(dab, dba) and np.all(gpu_handle.hamming_batch(a, a) == 0)
    assert np.all(dab <= dac + dcb) and np.all(gpu_handle.hamming_batch(a, 255 - a) == 256)


def test_large_odd_image(oracle, pkg):
    """a size near the handle bound with pitches that are not multiples of anything convenient"""
    w, hh, n = 2047, 1153, 3000
    h = pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), n, device=0, max_w=2048, max_h=1200, max_batch=1)
    L, R = pkg.synth.stereo_pair(91, 0, w, hh)
    kpL, dL, kpR, dR, m, pts, has = h.process_stereo(L, R, cap_kp=n + 1024)
    ok, od = oracle.orb_extract(L, oracle.orb_params(n))
    assert kpL.tobytes() == ok.tobytes() and np.array_equal(dL, od)
    h.close()


def test_bench_two_rank_rehearsal_line(tmp_path):
    """bench.py as the driver launches it for N > 1 (python -m torch.distributed.run, one rank per GPU), rehearsed on this one-GPU box with
    N = 2 (ORBX_DIST_REHEARSE=1: the ranks share the card, gloo carries the all-reduce hook — RCCL refuses two ranks on one device):
    ONE JSON line from rank 0 with the whole-job value, weak scaling, and BOTH shardings of local BA — one window's map points
    partitioned over the ranks with the normal equations all-reduced (lm_iters_per_s + transport), and one window per rank
    (independent_windows).  Small sizes: this checks the N > 1 path end to end, not its speed."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ORBX_DIST_REHEARSE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                        "--no-cpu-baseline", "--no-files", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["batch_pairs_per_gpu"] == 8
    ba = d["local_ba"]
    assert ba["iterations"] == 10 and ba["lm_iters_per_s"] > 0 and "points partitioned over 2 ranks" in ba["workload"]
    assert "REHEARSAL" in ba["transport"] and "REHEARSAL" in d["config"]["parallelism"] and "all-reduced" in d["config"]["parallelism"]
    assert ba["independent_windows"]["windows"] == 2 and ba["independent_windows"]["lm_iters_per_s"] > 0
