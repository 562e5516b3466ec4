"""The input side (reference src/io/euroc.rs:64-132, :189-211, :325-360): EuRoC mav0 reader +
PNG decode, host code of the library (no GPU needed — these tests run everywhere), and the end-to-end hand-off to
the per-frame path on the GPU.  No EuRoC data in the build: the directory is synthetic, in the dataset's layout."""
import os

import numpy as np
import pytest

import orb_slam3_rust_amd as P


@pytest.mark.parametrize("kw", [dict(filters="cycle"), dict(filters="none"), dict(filters=4, level=1), dict(bit_depth=16),
                                dict(alpha=True), dict(bit_depth=16, alpha=True, filters=3)])
def test_png_decode_round_trip(kw):
    rng = np.random.default_rng(1)
    img = (np.add.outer(np.arange(97), np.arange(131)) % 256).astype(np.uint8) ^ rng.integers(0, 32, (97, 131), dtype=np.uint8)
    assert np.array_equal(P.png_decode_gray8(P.synth.png_encode(img, **kw)), img)


def test_png_decode_rejects_what_it_does_not_support():
    img = np.zeros((8, 8), np.uint8)
    good = P.synth.png_encode(img)
    with pytest.raises(P.OrbxError):
        P.png_decode_gray8(b"not a png at all, just bytes" * 4)
    with pytest.raises(P.OrbxError):
        P.png_decode_gray8(good[:60])                              # truncated
    rgb = bytearray(good); rgb[25] = 2                              # colour type RGB
    with pytest.raises(P.OrbxError):
        P.png_decode_gray8(bytes(rgb))
    lace = bytearray(good); lace[28] = 1                            # interlaced
    with pytest.raises(P.OrbxError):
        P.png_decode_gray8(bytes(lace))
    bad = bytearray(good); bad[-30] ^= 0xFF                         # damaged deflate stream
    with pytest.raises(P.OrbxError):
        P.png_decode_gray8(bytes(bad))


def test_euroc_dataset_reader(tmp_path):
    root = tmp_path / "mav0"
    pairs = P.synth.write_euroc_mav0(str(root), 5, seed=3, w=160, h=96)
    ds = P.EurocDataset.new(root)
    assert len(ds) == 5 and (ds.width, ds.height) == (160, 96)
    assert ds.frame_timestamp(0) == 1403636579763555584 and ds.frame_timestamp(4) == 1403636579763555584 + 4 * 50000000
    assert ds.frame_timestamp(5) is None
    cam = P.synth.EUROC_CAMERA
    assert (ds.camera.fx, ds.camera.fy, ds.camera.cx, ds.camera.cy) == (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    assert abs(ds.camera.baseline - cam["baseline"]) < 1e-12 and ds.k_right == (cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    for i, (L, R) in enumerate(pairs):
        l, r, ts = ds.stereo_pair(i)
        assert np.array_equal(l, L) and np.array_equal(r, R) and ts == ds.frame_timestamp(i)
    batch = ds.read_pairs(1, 3, threads=4)
    assert batch.shape == (3, 2, 96, 160) and np.array_equal(batch[2, 1], pairs[3][1])
    with pytest.raises(P.OrbxError):
        ds.read_pairs(3, 3)
    os.remove(root / "cam1" / "data" / ("%d.png" % ds.frame_timestamp(2)))
    with pytest.raises(P.OrbxError, match="right image of frame 2"):
        ds.read_pairs(0, 5)


def test_euroc_open_errors(tmp_path):
    root = tmp_path / "mav0"
    P.synth.write_euroc_mav0(str(root), 2, w=64, h=48)
    with pytest.raises(P.OrbxError, match="Failed to open"):
        P.EurocDataset(tmp_path / "nowhere")
    csv1 = root / "cam1" / "data.csv"
    keep = csv1.read_text()
    csv1.write_text(keep.splitlines()[0] + "\n" + keep.splitlines()[1] + "\n")          # one frame fewer (:69-71)
    with pytest.raises(P.OrbxError, match="different number of frames"):
        P.EurocDataset(root)
    csv1.write_text(keep.replace("1403636579763555584,", "14036x,", 1))                 # timestamp does not parse (:202)
    with pytest.raises(P.OrbxError, match="invalid digit"):
        P.EurocDataset(root)
    csv1.write_text(keep + "1,2,3\n")                                                    # csv crate: unequal record lengths
    with pytest.raises(P.OrbxError, match="different number of fields"):
        P.EurocDataset(root)
    csv1.write_text(keep)
    y = root / "cam0" / "sensor.yaml"
    y.write_text(y.read_text().replace("intrinsics: [", "intrinsics: [1.0, "))          # five intrinsics (:364-369)
    with pytest.raises(P.OrbxError, match="4 intrinsics"):
        P.EurocDataset(root)


@pytest.mark.gpu
def test_euroc_files_to_features_end_to_end(gpu_handle, oracle, tmp_path):
    """PNG files -> pinned staging -> pipelined host batch: the features equal the oracle's on the decoded images."""
    from conftest import records_equal
    root = tmp_path / "mav0"
    pairs = P.synth.write_euroc_mav0(str(root), 6, seed=9)
    ds = P.EurocDataset(root)
    import torch
    cap = gpu_handle.orb_params.n_features + 2048
    imgs = torch.zeros((6, 2, ds.height, ds.width), dtype=torch.uint8).pin_memory()      # pinned staging
    ds.read_pairs(0, 6, out=imgs.numpy(), threads=8)
    out = P.Handle.alloc_host_outputs(6, cap)
    gpu_handle.process_stereo_batch_host(imgs, out)
    p = oracle.orb_params(gpu_handle.orb_params.n_features)
    for i in (0, 5):
        assert np.array_equal(imgs[i, 0].numpy(), pairs[i][0])
        okl, odl = oracle.orb_extract(pairs[i][0], p)
        n = int(out["nkp"][i, 0])
        kp = out["kp"][i, 0, :n].numpy().view(P.KEYPOINT).reshape(-1)
        assert records_equal(kp, okl) and np.array_equal(out["desc"][i, 0, :n].numpy(), odl)
