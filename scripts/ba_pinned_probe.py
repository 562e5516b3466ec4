#!/usr/bin/env python3
"""One window through orbx_ba_solve_visual with its observations in pageable memory and in page-locked memory (ORBX_BA_TIMING=1 for the phase split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
cam = P.CameraModel(**P.synth.EUROC_CAMERA); cfg = P.LocalBAConfigLM()
h = P.Handle(cam, 100)
w = P.synth.ba_window(43 if K == 50 else 42, K, M, P.BA_OBS)
pw = P.Handle.pack_ba_windows([w])[0]
for name, x in (("pageable", w), ("pinned", pw), ("pageable", w), ("pinned", pw)):
    for _ in range(2):
        h.ba_solve_visual(cam, cfg, x["poses_cw"], x["fixed_cw"], x["points"], x["obs"])
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = h.ba_solve_visual(cam, cfg, x["poses_cw"], x["fixed_cw"], x["points"], x["obs"]); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-9s K=%d M=%d obs=%d: median %.3f ms  min %.3f ms  (%d iterations)" % (name, K, M, len(w["obs"]), ts[len(ts) // 2] * 1e3, ts[0] * 1e3, r["iterations"]), flush=True)
