"""Compile-time envelope of the FAST kernel's tile macros (a -DORBX_FT_H / -DORBX_FT_THREADS sweep once ran with the
static_asserts removed and one build ended in a GPU memory fault).  An unsupported combination must fail to COMPILE; the shapes
the round-2 sweep used must still compile.  hipcc -fsyntax-only on the device side: no GPU, under a second per case."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "orb-slam3-rust_amd", "csrc", "orb_kernels.hip")
HIPCC = "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) and not shutil.which("hipcc"), reason="hipcc not installed")


def _syntax(*defs):
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-fsyntax-only",
           "-Wno-everything"] + list(defs) + [SRC]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stderr


@pytest.mark.parametrize("defs", [(), ("-DORBX_FT_H=46",), ("-DORBX_FT_H=62", "-DORBX_FT_THREADS=512"), ("-DORBX_FT_H=94", "-DORBX_FT_THREADS=512"),
                                  ("-DORBX_FT_H=78",), ("-DORBX_FP_PITCH=80",), ("-DORBX_FAST_CHAIN=1",)])
def test_supported_fast_tile_shapes_compile(defs):
    rc, err = _syntax(*defs)
    assert rc == 0, err[-2000:]


@pytest.mark.parametrize("defs,needle", [
    (("-DORBX_FT_H=63",), "tile height"),                                  # odd: the 2x2 NMS bound of s_list
    (("-DORBX_FT_H=128",), "tile height"),                                 # score rows no longer fit the 16-bit position list
    (("-DORBX_FT_THREADS=96",), "whole waves"),
    (("-DORBX_FT_THREADS=2048",), "whole waves"),
    (("-DORBX_FT_H=126", "-DORBX_FT_THREADS=128"), "staging passes"),      # 14 rows per pass x 4 registers < 134 pixel rows
    (("-DORBX_FP_PITCH=64",), "pixel row holds"),
    (("-DORBX_FP_PITCH=76",), "pixel row holds"),                          # 8-byte staging stores need an 8-byte multiple
    (("-DORBX_FAST_CHAIN=0",), "tiles per block"),
])
def test_unsupported_fast_tile_shapes_do_not_compile(defs, needle):
    rc, err = _syntax(*defs)
    assert rc != 0 and "static assertion failed" in err and needle in err, err[-2000:]
