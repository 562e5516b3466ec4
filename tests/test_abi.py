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


def test_ba_window_and_obs32_layouts_match_header(pkg, tmp_path):
    """orbx_ba_window / orbx_ba_obs32 as a C compiler lays them out (gcc on include/orbx.h) against the ctypes / numpy mirrors, and the
    orbx_ba_obs -> orbx_ba_obs32 mapping (fixed observer f as kf_idx = -1 - f, the identity pose as -1 - F; coordinates that are not f32 refused)."""
    import ctypes as C
    import subprocess
    from orb_slam3_rust_amd.api import _BaWindow
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "orbx.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(orbx_ba_window), '
                   'offsetof(orbx_ba_window, obs), offsetof(orbx_ba_window, obs32), sizeof(orbx_ba_obs32), offsetof(orbx_ba_obs32, u), sizeof(orbx_ba_obs)); return 0; }\n')
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(_BaWindow), _BaWindow.obs.offset, _BaWindow.obs32.offset, pkg.BA_OBS32.itemsize, pkg.BA_OBS32.fields["u"][1], pkg.BA_OBS.itemsize]
    assert pkg.BA_OBS32.itemsize == 16
    o = np.zeros(4, pkg.BA_OBS)
    o["kf_idx"] = [3, -1, -1, 0]; o["fixed_idx"] = [-1, 1, -1, -1]; o["mp_idx"] = [7, 8, 9, 10]
    o["u"] = [1.5, 2.25, 640.125, 0.0]; o["v"] = [3.0, 4.5, 100.0625, 479.5]
    c = pkg.ba_obs_to_obs32(o, 2)
    assert list(c["kf_idx"]) == [3, -2, -3, 0] and list(c["mp_idx"]) == [7, 8, 9, 10]          # fixed 1 -> -2; identity (fixed_idx -1) -> -1 - F = -3
    assert np.array_equal(c["u"].astype(np.float64), o["u"]) and np.array_equal(c["v"].astype(np.float64), o["v"])
    o["u"][2] = 640.1                                                                           # not an f32
    assert pkg.ba_obs_to_obs32(o, 2) is None


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
