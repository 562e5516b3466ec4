"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol of
include/orbx.h, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "orbx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "liborbx_hip.so does not export %s" % s
    assert sorted(pkg.ABI_SYMBOLS) == syms
    assert b"gfx950" in L.orbx_version()


def test_struct_layouts_match_header(pkg):
    assert pkg.KEYPOINT.itemsize == 28 and pkg.DMATCH.itemsize == 16 and pkg.BA_OBS.itemsize == 32
    assert pkg.KEYPOINT.names == ("x", "y", "size", "angle", "response", "octave", "class_id")
    assert pkg.DMATCH.names == ("query_idx", "train_idx", "img_idx", "distance")


def test_defaults_mirror_reference(pkg):
    import ctypes as C
    from orb_slam3_rust_amd.api import _BaConfig, _OrbParams
    L = pkg.load_library()
    p = _OrbParams(); L.orbx_default_orb_params(C.c_int(1200), C.byref(p))
    # stereo.rs:38-48
    assert (p.n_features, p.n_levels, p.edge_threshold, p.first_level, p.wta_k, p.score_type,
            p.patch_size, p.fast_threshold) == (1200, 8, 31, 0, 2, 0, 31, 20)
    assert abs(p.scale_factor - 1.2) < 1e-6
    c = _BaConfig(); L.orbx_default_ba_config(C.byref(c))
    # local_ba_lm.rs:109-119
    assert (c.max_iterations, c.param_tolerance, c.gradient_tolerance, c.max_covisible_keyframes) == (10, 1e-8, 1e-8, 20)
    assert c.huber_threshold == np.sqrt(5.991)
    d = pkg.LocalBAConfigLM()
    assert d.huber_threshold == c.huber_threshold and d.max_iterations == 10


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.OrbxError) as e:
        pkg.Handle(pkg.CameraModel(**pkg.synth.EUROC_CAMERA), 2000)
    assert e.value.code == -2   # ORBX_ERR_NO_DEVICE


def test_product_does_not_touch_oracle():
    """The product tree must not import, link or include anything under oracle/."""
    pdir = os.path.join(ROOT, "orb-slam3-rust_amd")
    for dp, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".inc")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle/" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
