#!/usr/bin/env python3
"""Phase split of ba_solve_inertial_tiled_kernel (the 15-d system of a 10-keyframe inertial window, n = 150) from a -DORBX_SOLVE_STAMPS
build of the library (scripts/build_stamps.sh -> build_ab/bast.so): ORBX_LIBRARY=$PWD/build_ab/bast.so python scripts/ba_inertial_stamps.py [K]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
L = P.load_library()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
iw = P.synth.inertial_window(42, K, 2000, P.BA_OBS)
icfg = P.LocalInertialBAConfig()
args = (cam, icfg, iw["poses_wc"], iw["velocities"], iw["biases"], iw["fixed_cw"], iw["points"], iw["obs"], iw["edge_kf"], iw["preint"])
h.ba_solve_inertial(*args)
buf = (C.c_ulonglong * 16)()
L.orbx_debug_solve_stamps(buf, 1)
for _ in range(3):
    h.ba_solve_inertial(*args)
L.orbx_debug_solve_stamps(buf, 0)
n = buf[15]
names = ["assemble S, b, |g|", "-", "wave 0 at the panel's barrier (the rows below still being solved)", "-",
         "exit", "backward substitution + norms", "-", "wave 0: its tile of the previous panel's update", "wave 0: diagonal block from LDS",
         "wave 0: the 16 pivot steps (columns published as they finish)", "wave 0: diagonal block to LDS"]
tot = sum(buf[i] for i in range(11))
print("solves %d, ticks per solve %.0f" % (n, tot / max(n, 1)))
for i in range(11):
    print("  %-52s %8.0f  %5.1f %%" % (names[i], buf[i] / max(n, 1), 100.0 * buf[i] / max(tot, 1)))
