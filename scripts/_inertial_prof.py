import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_rust_amd as P
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
iw = P.synth.inertial_window(42, 10, 2000, P.BA_OBS)
icfg = P.LocalInertialBAConfig()
args = (cam, icfg, iw["poses_wc"], iw["velocities"], iw["biases"], iw["fixed_cw"], iw["points"], iw["obs"], iw["edge_kf"], iw["preint"])
for _ in range(2): r = h.ba_solve_inertial(*args)
t0 = time.perf_counter()
for _ in range(5): r = h.ba_solve_inertial(*args)
print("inertial K=10 M=2000 obs=%d: %.3f ms/solve (%d iterations)" % (len(iw["obs"]), (time.perf_counter() - t0) * 200, r["iterations"]))
h.set_profiling(True)
r = h.ba_solve_inertial(*args)
kt = h.kernel_times()
for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    print("  %-26s %8.3f ms  %3d scopes  %7.1f us each" % (k, ms, n, ms / n * 1e3))
