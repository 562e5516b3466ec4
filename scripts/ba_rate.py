#!/usr/bin/env python3
"""Local-BA rates alone (the bench's local_ba leg without the extractor): single window and 32-window batch, LM iterations/s.
usage: [ORBX_LIBRARY=...] python scripts/ba_rate.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_rust_amd as P
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 100)
cfg = P.LocalBAConfigLM()
w = P.synth.ba_window(42, 20, 2000, P.BA_OBS)
h.ba_solve_visual_batch(cam, cfg, [w])
t0 = time.perf_counter(); its = 0
for _ in range(reps):
    its += h.ba_solve_visual_batch(cam, cfg, [w])[0]["iterations"]
dt = time.perf_counter() - t0
print("single window: %.0f LM iterations/s (%.3f ms per solve)" % (its / dt, dt / reps * 1e3))
wins = [P.synth.ba_window(200 + i, 20, 2000, P.BA_OBS) for i in range(32)]
h.ba_solve_visual_batch(cam, cfg, wins)
t0 = time.perf_counter(); its = 0
for _ in range(max(reps // 4, 2)):
    its += sum(r["iterations"] for r in h.ba_solve_visual_batch(cam, cfg, wins))
dt = time.perf_counter() - t0
print("32-window batch: %.0f LM iterations/s" % (its / dt))
