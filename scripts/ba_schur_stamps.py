#!/usr/bin/env python3
"""Phase split of ba_schur_diag_body (wave 0 of every block) from a -DORBX_SCHUR_STAMPS build of the library:
scripts/build_variant.sh schst ba_kernels.hip -DORBX_SCHUR_STAMPS && ORBX_LIBRARY=$PWD/build_ab/schst.so python scripts/ba_schur_stamps.py [windows]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
L = P.load_library()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(W)]
cfg = P.LocalBAConfigLM()
h.ba_solve_visual_batch(cam, cfg, wins)
buf = (C.c_ulonglong * 8)()
L.orbx_debug_schur_stamps(buf, 1)
for _ in range(2):
    h.ba_solve_visual_batch(cam, cfg, wins)
L.orbx_debug_schur_stamps(buf, 0)
n = buf[7]
names = ["prologue: zero fill of the LDS tiles, first fetches", "wait at the tile's first barrier (previous tile consumed)", "fill: W, Y from the stored numbers -> LDS",
         "next tile's loads issued + wait at the second barrier", "MFMA phase", "partials out"]
tot = sum(buf[i] for i in range(6))
print("%d windows: blocks %d, ticks per block %.0f" % (W, n, tot / max(n, 1)))
for i in range(6):
    print("  %-62s %9.0f  %5.1f %%" % (names[i], buf[i] / max(n, 1), 100.0 * buf[i] / max(tot, 1)))
