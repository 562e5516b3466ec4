"""The CPU oracle under AddressSanitizer + UBSan (sanitizers run on the CPU build only).
Runs in a subprocess because the sanitizer runtime has to be preloaded into the interpreter."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ROOT)
from oracle import oracle as O
import orb_slam3_rust_amd.synth as synth
L = C.CDLL(os.path.join(ROOT, "oracle", "liborbx_oracle_asan.so"))
O._lib = L
L.oracle_hamming256.restype = C.c_uint32
L.oracle_fast_atan2.restype = C.c_float
L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
left, right = synth.stereo_pair(5, 0, 333, 257)
p = O.orb_params(500)
kl, dl = O.orb_extract(left, p)
kr, dr = O.orb_extract(right, p)
cam = O.Camera(**synth.EUROC_CAMERA)
m, pts, has = O.stereo_match(cam, kl, dl, kr, dr)
O.crosscheck_match(dl[:200], dr[:300])
O.guided_match(kl, dl, 333.0, 257.0, np.array([[-50.0, 10.0], [100.0, 100.0], [1e6, 5.0]]), dr[:3], 15.0, 1)
w = synth.ba_window(1, 5, 60, O.BA_OBS, n_fixed_extra=1)
a = O.ba_solve_dense(O.Camera(**w["camera"]), O.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
b = O.ba_solve_schur(O.Camera(**w["camera"]), O.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
assert a["iterations"] == b["iterations"]
# the §8f restatements: triangulation searches (grid / FeatureVector), fuse search, vocabulary, global + inertial BA
L.oracle_vocab_load_text.restype = C.c_void_p
L.oracle_vocab_from_arrays.restype = C.c_void_p
tv = synth.two_view_features(3, 600, O.KEYPOINT, dup=0.4)
O.search_for_triangulation(O.Camera(**tv["camera"]), tv["kp1"], tv["desc1"], tv["mp1"], tv["stereo1"], tv["kp2"], tv["desc2"], tv["mp2"],
                           tv["pose1_wc"], tv["pose2_wc"])
voc = synth.vocabulary(2, k=4, depth=3, ragged=True)
v = O.Vocabulary.from_arrays(*voc, 4, 3)
n1 = v.transform(tv["desc1"], 1)[2]; n2 = v.transform(tv["desc2"], 9)[2]
O.search_for_triangulation_bow(O.Camera(**tv["camera"]), tv["kp1"], tv["desc1"], tv["mp1"], tv["stereo1"], n1, tv["kp2"], tv["desc2"],
                               tv["mp2"], n2, tv["pose1_wc"], tv["pose2_wc"])
import tempfile
with tempfile.TemporaryDirectory() as td:
    synth.write_vocabulary_text(os.path.join(td, "v.txt"), *voc, 4, 3)
    assert O.Vocabulary.load_from_text(os.path.join(td, "v.txt")).n_nodes == v.n_nodes
fs = synth.fuse_scene(1, 200, 3, 150, O.KEYPOINT)
O.fuse_search(O.Camera(**fs["camera"]), fs["positions"], fs["mp_desc"], fs["kf_poses_wc"], fs["kf_feat_offset"], fs["kps"], fs["descs"], 10.75)
gw = synth.ba_window(2, 4, 50, O.BA_OBS)
O.global_ba_solve_dense(O.Camera(**gw["camera"]), O.ba_config(), gw["poses_cw"], gw["fixed_cw"], gw["points"], gw["obs"])
iw = synth.inertial_window(1, 3, 40, O.BA_OBS)
ir = O.inertial_ba_solve(O.Camera(**iw["camera"]), O.inertial_ba_config(), iw["poses_wc"], iw["velocities"], iw["biases"], iw["fixed_cw"],
                         iw["points"], iw["obs"], iw["edge_kf"], iw["preint"])
print("SANITIZED_OK", len(kl), len(m), a["iterations"], ir["iterations"])
'''


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborbx_oracle_asan.so"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
