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

# fuse search (search_in_neighbors.rs:273-343): P map points x T keyframes x N features
import ctypes
from oracle import oracle as O
for Pn, T, N in ((2000, 20, 1200), (8000, 30, 2000)):
    s = P.synth.fuse_scene(1, Pn, T, N, P.KEYPOINT)
    cam = P.CameraModel(**s["camera"])
    args = (cam, s["positions"], s["mp_desc"], s["kf_poses_wc"], s["kf_feat_offset"], s["kps"], s["descs"], 10.75, 50)
    for _ in range(3):
        idx, _d = h.fuse_search(*args)
    h.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(10):
        idx, _d = h.fuse_search(*args)
    dt = (time.perf_counter() - t0) / 10
    kt = h.kernel_times()
    h.set_profiling(False)
    t0 = time.perf_counter()
    O.fuse_search(O.Camera(**s["camera"]), *args[1:])
    tc = time.perf_counter() - t0
    print(f"fuse P={Pn} T={T} N={N} found={(idx >= 0).sum()} host_call={dt*1e3:.3f} ms  cpu_oracle={tc*1e3:.1f} ms  " +
          "  ".join(f"{k}={v[0]/max(v[1],1)*1e3:.1f}us" for k, v in kt.items()))

# BoW transform (vocabulary/mod.rs:296-325): k=10, depth 5 synthetic tree (111 111 nodes, the size class of ORBvoc.txt's upper 5 levels)
voc = P.synth.vocabulary(1, k=10, depth=5)
gv = P.OrbVocabulary.from_nodes(*voc, 10, 5, handle=h)
ov = O.Vocabulary.from_arrays(*voc, 10, 5)
rng = np.random.default_rng(0)
for n in (2000, 256000):
    q = voc[2][rng.integers(1, len(voc[2]), n)] ^ rng.integers(0, 2, (n, 32), dtype=np.uint8)
    for _ in range(2):
        gv.transform_arrays(q, 4)
    h.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(5):
        g = gv.transform_arrays(q, 4)
    dt = (time.perf_counter() - t0) / 5
    kt = h.kernel_times()
    h.set_profiling(False)
    t0 = time.perf_counter()
    o = ov.transform(q, 4)
    tc = time.perf_counter() - t0
    print(f"bow n={n} nodes={gv.num_nodes()} equal={all(np.array_equal(a, b) for a, b in zip(g, o))} host_call={dt*1e3:.3f} ms cpu_oracle={tc*1e3:.1f} ms  " +
          "  ".join(f"{k}={v[0]/max(v[1],1)*1e3:.1f}us" for k, v in kt.items()))
