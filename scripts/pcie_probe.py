#!/usr/bin/env python3
"""orbx_process_stereo_batch (pipelined host-buffer path) alone: 256 pairs in pinned host memory, timed per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_rust_amd as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 2000, device=0, max_w=752, max_h=480, max_batch=B)
pairs = [P.synth.stereo_pair(7, i) for i in range(8)]
img = torch.from_numpy(np.stack([np.stack(p) for p in pairs])).repeat((B + 7) // 8, 1, 1, 1)[:B].contiguous().pin_memory()
hout = P.Handle.alloc_host_outputs(B, 2304)
for _ in range(2):
    h.process_stereo_batch_host(img, hout)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); h.process_stereo_batch_host(img, hout); ts.append(time.perf_counter() - t0)
print("orbx_process_stereo_batch %d pairs: %s ms per call -> %.0f frames/s" % (B, " ".join("%.2f" % (t * 1e3) for t in ts), B / min(ts)))
dev = img.cuda(); out = h.alloc_batch_outputs(B, 2304)
for _ in range(2):
    h.process_stereo_batch_device(dev, out)
h.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    h.process_stereo_batch_device(dev, out)
h.synchronize(); print("device-resident: %.2f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3))
