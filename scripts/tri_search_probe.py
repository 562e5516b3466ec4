"""Kernel times of orbx_search_for_triangulation (propose / resolve) on synthetic two-view scenes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_rust_amd as P

h = P.Handle(P.CameraModel(**P.synth.EUROC_CAMERA), 1200)
for n, dup in ((1200, 0.0), (2500, 0.3), (6000, 0.9)):
    s = P.synth.two_view_features(1, n, P.KEYPOINT, dup=dup)
    cam = P.CameraModel(**s["camera"])
    args = (cam, s["kp1"], s["desc1"], s["mp1"], s["stereo1"], s["kp2"], s["desc2"], s["mp2"], s["pose1_wc"], s["pose2_wc"], 50)
    for _ in range(3):
        m = h.search_for_triangulation(*args)
    h.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(20):
        m = h.search_for_triangulation(*args)
    dt = (time.perf_counter() - t0) / 20
    kt = h.kernel_times()
    h.set_profiling(False)
    print(f"n1={len(s['kp1'])} dup={dup} matches={len(m)} host_call={dt*1e3:.3f} ms  " +
          "  ".join(f"{k}={v[0]/max(v[1],1)*1e3:.1f}us" for k, v in kt.items()))
