#!/usr/bin/env python3
"""Per-kernel HIP-event times of one local-BA solve (config 3) + wall time split."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_rust_amd as P
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
w = P.synth.ba_window(43 if K == 50 else 42, K, M, P.BA_OBS)       # configs[4] = synth_ba(seed=43, K=50, M=8000)
cfg = P.LocalBAConfigLM()
for _ in range(3):
    r = h.ba_solve_visual(cam, cfg, w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
t0 = time.perf_counter()
for _ in range(10):
    r = h.ba_solve_visual(cam, cfg, w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
print("wall ms/solve (no profiling): %.3f, iterations %d, obs %d" % ((time.perf_counter() - t0) * 100, r["iterations"], len(w["obs"])))
h.set_profiling(True)
r = h.ba_solve_visual(cam, cfg, w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
kt = h.kernel_times()
tot = 0
for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    print("  %-22s %8.3f ms  %3d scopes  %7.1f us each" % (k, ms, n, ms / n * 1e3)); tot += ms
print("sum of kernel scopes: %.3f ms" % tot)

if len(sys.argv) > 3 and sys.argv[3] == "visual-only":     # profiler passes: the visual solve alone
    h.close()
    sys.exit(0)
# local inertial BA (local_inertial_ba.rs:1074-1275): window of 10 keyframes, 2000 points
from oracle import oracle as O
iw = P.synth.inertial_window(42, 10, 2000, P.BA_OBS)
icfg = P.LocalInertialBAConfig()
args = (cam, icfg, iw["poses_wc"], iw["velocities"], iw["biases"], iw["fixed_cw"], iw["points"], iw["obs"], iw["edge_kf"], iw["preint"])
for _ in range(2):
    r = h.ba_solve_inertial(*args)
t0 = time.perf_counter()
for _ in range(5):
    r = h.ba_solve_inertial(*args)
gpu_ms = (time.perf_counter() - t0) * 200
print("inertial BA K=10 M=2000 obs=%d: %.3f ms/solve (%d iterations)" % (len(iw["obs"]), gpu_ms, r["iterations"]))
# the literal dense formulation (LU of 15K+3M unknowns per iteration) is O(n^3): timed on a small window only
sw = P.synth.inertial_window(42, 5, 300, P.BA_OBS)
t0 = time.perf_counter()
o = O.inertial_ba_solve(O.Camera(**sw["camera"]), O.inertial_ba_config(), sw["poses_wc"], sw["velocities"], sw["biases"], sw["fixed_cw"],
                        sw["points"], sw["obs"], sw["edge_kf"], sw["preint"])
cpu_ms = (time.perf_counter() - t0) * 1e3
sargs = (cam, icfg, sw["poses_wc"], sw["velocities"], sw["biases"], sw["fixed_cw"], sw["points"], sw["obs"], sw["edge_kf"], sw["preint"])
h.ba_solve_inertial(*sargs)
t0 = time.perf_counter()
r2 = h.ba_solve_inertial(*sargs)
print("inertial BA K=5 M=300: GPU %.3f ms, CPU oracle (dense, %d unknowns) %.0f ms, iterations %d / %d" %
      ((time.perf_counter() - t0) * 1e3, 15 * 5 + 900, cpu_ms, r2["iterations"], o["iterations"]))
