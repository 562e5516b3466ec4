"""The handle's persistent host workers (orbx_internal.hpp OrbxWorkPool: the batch BA call's per-window preprocessing runs on them):
every index once, fewer workers than the pool on request, exceptions reported by value.  Host-only C++, built with g++."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workpool(tmp_path):
    exe = str(tmp_path / "workpool_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "workpool_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "workpool ok" in r.stdout, r.stdout + r.stderr
