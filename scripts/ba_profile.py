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
w = P.synth.ba_window(42, K, M, P.BA_OBS)
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
