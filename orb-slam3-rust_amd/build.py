"""Build recipe for the HIP library (hipcc, gfx950 only).  `build()` is what
__graft_entry__.build() calls; it cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORBX_LIBRARY points at another build of the same library (A/B timing of two builds on one box)
LIB_PATH = os.environ.get("ORBX_LIBRARY") or os.path.join(_HERE, "liborbx_hip.so")


def build(jobs=4, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(jobs)]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH
