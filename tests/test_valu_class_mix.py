"""profiles/valu_class_mix.json — what bench.py prices roofline.valu_issue with — must be what scripts/valu_class_mix.py derives from
the library that ships: the script is re-run here (no GPU: llvm-objdump of the gfx950 code objects in liborbx_hip.so) and compared
with the committed file.  A kernel edit that changes FAST's loop structure makes the script refuse the stated loop weights
(profiles/valu_loop_weights.json), so stale weights cannot survive a change of the code they describe."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "orb-slam3-rust_amd", "liborbx_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")


def test_class_mix_file_matches_the_shipped_code_object(tmp_path):
    out = tmp_path / "mix.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "valu_class_mix.py"), "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    new = json.load(open(out))["kernels"]
    old = json.load(open(os.path.join(ROOT, "profiles", "valu_class_mix.json")))["kernels"]
    assert set(new) == set(old)
    f_new, f_old = new["fast_kernel"], old["fast_kernel"]
    assert f_new["weights"].startswith("stated") and len(f_new["loops"]) == len(f_old["loops"])
    for k in new:
        assert abs(new[k]["share_2cycle"] - old[k]["share_2cycle"]) < 0.01, (k, new[k]["share_2cycle"], old[k]["share_2cycle"])
    # every opcode of the dominant kernel is classified from a measured probe row (none by family rule alone)
    assert f_new["valu_static"]["unmeasured_opcodes"] <= 0.06 * sum(f_new["valu_static"][c] for c in ("2cycle", "4cycle", "8cycle"))
    assert 0.45 < f_new["share_2cycle"] < 0.60 and f_new["share_2cycle_bounds"][0] <= f_new["share_2cycle"] <= f_new["share_2cycle_bounds"][1]


def test_bench_reads_the_class_mix_file():
    sys.path.insert(0, ROOT)
    import bench
    share, bounds, _ = bench.two_cycle_share("fast_kernel")
    want = json.load(open(os.path.join(ROOT, "profiles", "valu_class_mix.json")))["kernels"]["fast_kernel"]["share_2cycle"]
    assert share == want and bounds[0] <= share <= bounds[1]
    assert bench.two_cycle_share("no_such_kernel") == (None, None, None)
