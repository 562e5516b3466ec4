#!/usr/bin/env python3
"""Phase split of the consumer wave 0 of every ba_schur_diag_ws_body workgroup from a -DORBX_SCHUR_STAMPS build of the library:
scripts/build_variant.sh schst ba_kernels.hip -DORBX_SCHUR_STAMPS && ORBX_LIBRARY=$PWD/build_ab/schst.so python scripts/ba_schur_stamps.py [windows]"""
import ctypes as C, os, sys
os.environ.setdefault("ORBX_BA_NO_SPLIT", "1")     # one launch per kernel of the loop (a large batch otherwise runs as two halves on two streams)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
L = P.load_library()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(W)]
cfg = P.LocalBAConfigLM()
h.ba_solve_visual_batch(cam, cfg, wins)
buf = (C.c_ulonglong * 16)()
L.orbx_debug_schur_stamps(buf, 1)
for _ in range(2):
    h.ba_solve_visual_batch(cam, cfg, wins)
L.orbx_debug_schur_stamps(buf, 0)
n = buf[7]
names = ["from the start of the consumer wave to the first tile (zero fill, tile 0 built by the producers)", "wait at the tile's barrier (producers still building the next tile)",
         "-", "-", "the tile's 54 MFMAs (operands from LDS)", "a k-split's nine partial tiles out"]
tot = sum(buf[i] for i in (0, 1, 4, 5))
print("%d windows: workgroups %d, cycles (s_memtime) per workgroup %.0f; shader clock over the consumer wave's life %.3f GHz" %
      (W, n, tot / max(n, 1), buf[12] / max(buf[6], 1) * 0.1))
for i in (0, 1, 4, 5):
    print("  %-100s %9.0f  %5.1f %%" % (names[i], buf[i] / max(n, 1), 100.0 * buf[i] / max(tot, 1)))
ptot = buf[8] + buf[9] + buf[10] + buf[11]
print("  producer wave 4: prologue %.0f | building tiles (arithmetic + LDS writes) %.0f | next tile's loads issued %.0f | at the barrier %.0f   (cycles per workgroup, total %.0f)" %
      (buf[11] / max(n, 1), buf[8] / max(n, 1), buf[9] / max(n, 1), buf[10] / max(n, 1), ptot / max(n, 1)))
# one launch on its own: from the first workgroup's start to the last one's end (100 MHz ticks) against one workgroup's life
c1 = P.LocalBAConfigLM(); c1.max_iterations = 1
L.orbx_debug_schur_stamps(buf, 1)
h.ba_solve_visual_batch(cam, c1, wins)
L.orbx_debug_schur_stamps(buf, 0)
print("one launch: %d workgroups, first start to last end %.1f us, a workgroup's consumer wave lives %.1f us on average" %
      (buf[7], (buf[3] - buf[2]) * 0.01, buf[6] / max(buf[7], 1) * 0.01))
