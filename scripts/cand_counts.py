"""Candidates (FAST NMS survivors) per level of a few synthetic bench-like images, beside 2 x quota: how many Harris responses a form that
computes them for EVERY survivor (inside the FAST tile) would take."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_rust_amd as P

cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 2000, device=0, max_w=752, max_h=480, max_batch=1)
tot = np.zeros(8); n = 0
for seed in range(4):
    L, R = P.synth.stereo_pair(seed, seed)
    h.process_stereo(L, R)
    for img in range(2):
        c = [len(h.debug_candidates(img, l)) for l in range(8)]
        tot += c; n += 1
print("mean candidates per level", (tot / n).round(0), "sum", tot.sum() / n)
h.close()
