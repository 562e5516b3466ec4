#!/usr/bin/env python3
"""The bench's batched local-BA leg alone (32 windows of 20 keyframes / 2000 points through orbx_ba_solve_visual_batch), for
rocprofv3 passes:  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA
GRBM_GUI_ACTIVE SQ_WAVES -- python3 scripts/ba_batch_profile.py [windows [K M]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
pageable = [P.synth.ba_window(200 + i, K, M, P.BA_OBS) for i in range(W)]
wins = P.Handle.pack_ba_windows(pageable)          # the observations of all windows in one page-locked buffer (orbx.h)
cfg = P.LocalBAConfigLM()
reps = int(os.environ.get("ORBX_PROFILE_REPS", "3"))
for name, ws in (("pinned, one buffer", wins), ("pageable", pageable)):
    h.ba_solve_visual_batch(cam, cfg, ws)
    t0 = time.perf_counter()
    its = 0
    for _ in range(reps):
        its += sum(r["iterations"] for r in h.ba_solve_visual_batch(cam, cfg, ws))
    dt = time.perf_counter() - t0
    print("%d windows (K=%d, M=%d, %d observations each; observations %s): %.3f ms per call, %.0f LM iterations/s" % (W, K, M, len(ws[0]["obs"]), name, dt / reps * 1e3, its / dt))
if len(sys.argv) > 4 and sys.argv[4] == "kernels":     # per-kernel HIP-event times of one more call (one stream)
    h.set_profiling(True)
    h.ba_solve_visual_batch(cam, cfg, wins)
    kt = h.kernel_times()
    h.set_profiling(False)
    tot = sum(v[0] for k, v in kt.items() if k.startswith("ba_"))
    for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0]):
        if k.startswith("ba_"):
            print("  %-22s %8.3f ms  %3d scopes  %7.1f us each" % (k, ms, n, ms / n * 1e3))
    print("device ms per call %.3f -> %.0f LM iterations/s device-only" % (tot, its / reps / (tot * 1e-3)))
