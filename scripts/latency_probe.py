#!/usr/bin/env python3
"""Single stereo pair through the host-buffer drop-in entry point (PCIe-inclusive latency) and
throughput as a function of batch size (device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import orb_slam3_rust_amd as P
cam = P.CameraModel(**P.synth.EUROC_CAMERA)
h = P.Handle(cam, 2000, max_batch=256)
L, R = P.synth.stereo_pair(1, 0)
for _ in range(5):
    h.process_stereo(L, R)
t0 = time.perf_counter()
for _ in range(50):
    h.process_stereo(L, R)
print("orbx_process_stereo (host buffers, 1 pair, PCIe + sync inclusive): %.3f ms/frame" % ((time.perf_counter() - t0) / 50 * 1e3))
imgs = torch.from_numpy(P.synth.stereo_batch(2, 0, 32)).cuda()
for b in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    x = imgs.repeat((b + 31) // 32, 1, 1, 1)[:b].contiguous()
    out = h.alloc_batch_outputs(b, 2304)
    for _ in range(3):
        h.process_stereo_batch_device(x, out)
    h.synchronize()
    t0 = time.perf_counter()
    n = max(5, 200 // b)
    for _ in range(n):
        h.process_stereo_batch_device(x, out)
    h.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("batch %4d: %8.3f ms/step  %9.1f frames/s" % (b, dt * 1e3, b / dt))

# PCIe-inclusive batched form: pinned host images -> HBM -> process -> all results back to pinned host memory
for b in (8, 64):
    host_in = torch.from_numpy(P.synth.stereo_batch(3, 0, min(b, 16))).repeat((b + 15) // 16, 1, 1, 1)[:b].contiguous().pin_memory()
    out = h.alloc_batch_outputs(b, 2304)
    host_out = {k: torch.empty_like(v, device="cpu").pin_memory() for k, v in out.items() if hasattr(v, "shape")}
    dev_in = torch.empty_like(host_in, device="cuda")
    def step():      # serial: upload, kernels, download, each fenced (no overlap)
        dev_in.copy_(host_in, non_blocking=True)
        torch.cuda.synchronize()
        h.process_stereo_batch_device(dev_in, out)
        h.synchronize()
        for k, v in host_out.items():
            v.copy_(out[k], non_blocking=True)
        torch.cuda.synchronize()
    for _ in range(3):
        step()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step()
    dt = (time.perf_counter() - t0) / n
    print("PCIe-inclusive batch %3d (H2D images + process + D2H of every output buffer at full capacity, serial): %7.3f ms/step  %9.1f frames/s" % (b, dt * 1e3, b / dt))

# pipelined host form (orbx_process_stereo_batch): three streams, chunks of 32 pairs
hp = P.Handle(cam, 2000, max_batch=32)
for b in (64, 256):
    host_in = torch.from_numpy(P.synth.stereo_batch(3, 0, 16)).repeat((b + 15) // 16, 1, 1, 1)[:b].contiguous().pin_memory()
    hout = hp.alloc_host_outputs(b, 2304)
    for _ in range(2):
        hp.process_stereo_batch_host(host_in, hout)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        hp.process_stereo_batch_host(host_in, hout)
    dt = (time.perf_counter() - t0) / n
    print("PCIe-inclusive pipelined batch %3d (orbx_process_stereo_batch, pinned host buffers): %7.3f ms  %9.1f frames/s" % (b, dt * 1e3, b / dt))
