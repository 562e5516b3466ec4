"""ctypes loader for the CPU oracle (oracle/liborbx_oracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"),
                   ("distance", "<f4")])
BA_OBS = np.dtype([("kf_idx", "<i4"), ("fixed_idx", "<i4"), ("mp_idx", "<i4"), ("_pad", "<i4"),
                   ("u", "<f8"), ("v", "<f8")])
assert KEYPOINT.itemsize == 28 and DMATCH.itemsize == 16 and BA_OBS.itemsize == 32


class Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "baseline")]


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int), ("scale_factor", C.c_float), ("n_levels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


class BaConfig(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("param_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("huber_threshold", C.c_double),
                ("max_covisible_keyframes", C.c_int)]


class InertialBaConfig(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("window_size", C.c_int), ("huber_threshold_mono", C.c_double),
                ("huber_threshold_stereo", C.c_double), ("initial_lambda", C.c_double), ("gyro_rw_info", C.c_double),
                ("accel_rw_info", C.c_double)]


def inertial_ba_config():
    """LocalInertialBAConfig::default, local_inertial_ba.rs:126-141"""
    return InertialBaConfig(10, 10, float(np.sqrt(5.991)), float(np.sqrt(7.815)), 1e-2, 1e6, 1e4)


class OrbLevels(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("w", C.c_int * 8), ("h", C.c_int * 8),
                ("scale", C.c_float * 8), ("quota", C.c_int * 8)]


def orb_params(n_features):
    """stereo.rs:38-48"""
    return OrbParams(n_features, 1.2, 8, 31, 0, 2, 0, 31, 20)


def ba_config():
    """LocalBAConfigLM::default, local_ba_lm.rs:109-119"""
    return BaConfig(10, 1e-8, 1e-8, float(np.sqrt(5.991)), 20)


def use_native_build():
    """Rebuild the oracle with -O3 -march=native on THIS machine and use it from now on (bench.py's cpu_baseline;
    the portable -O2 build travels between machines, a -march=native one must not).  Same results, faster."""
    global _lib
    import tempfile
    out = os.path.join(tempfile.gettempdir(), "liborbx_oracle_native_%d.so" % os.getuid())
    srcs = [os.path.join(_HERE, f) for f in ("match_ref.cpp", "orb_ref.cpp", "ba_ref.cpp")]
    subprocess.run(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-pthread", "-shared", "-o", out] + srcs, check=True)
    L = C.CDLL(out)
    L.oracle_hamming256.restype = C.c_uint32
    L.oracle_fast_atan2.restype = C.c_float
    L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.oracle_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    _lib = L
    return out


def build(asan=False):
    target = "liborbx_oracle_asan.so" if asan else "liborbx_oracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liborbx_oracle.so")
        srcs = [os.path.join(_HERE, f) for f in ("match_ref.cpp", "orb_ref.cpp", "ba_ref.cpp", "oracle.h")]
        if (not os.path.exists(path)) or any(
                os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            build()
        L = C.CDLL(path)
        L.oracle_hamming256.restype = C.c_uint32
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_sincos_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hamming_batch(a, b):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    out = np.zeros(len(a), np.uint32)
    lib().oracle_hamming_batch(_p(a), _p(b), C.c_int(len(a)), _p(out))
    return out


def stereo_match(cam, kpL, descL, kpR, descR):
    kpL = np.ascontiguousarray(kpL, KEYPOINT); kpR = np.ascontiguousarray(kpR, KEYPOINT)
    descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
    nL, nR = len(kpL), len(kpR)
    matches = np.zeros(max(nL, 1), DMATCH)
    pts = np.zeros((max(nL, 1), 3), np.float64)
    has = np.zeros(max(nL, 1), np.uint8)
    n = lib().oracle_stereo_match(C.byref(cam), _p(kpL), _p(descL), C.c_int(nL), _p(kpR), _p(descR),
                                  C.c_int(nR), _p(matches), _p(pts), _p(has))
    return matches[:n].copy(), pts[:nL].copy(), has[:nL].copy()


def crosscheck_match(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros(max(len(q), 1), DMATCH)
    n = lib().oracle_crosscheck_match(_p(q), C.c_int(len(q)), _p(t), C.c_int(len(t)), _p(out))
    return out[:n].copy()


def guided_match(kp, desc, img_w, img_h, q_uv, q_desc, radius, mode):
    kp = np.ascontiguousarray(kp, KEYPOINT); desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    q_uv = np.ascontiguousarray(q_uv, np.float64).reshape(-1, 2)
    q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
    nq = len(q_uv)
    idx = np.zeros(max(nq, 1), np.int32); dist = np.zeros(max(nq, 1), np.uint32)
    lib().oracle_guided_match(_p(kp), _p(desc), C.c_int(len(kp)), C.c_double(img_w), C.c_double(img_h), _p(q_uv),
                              _p(q_desc), C.c_int(nq), C.c_double(radius), C.c_int(mode), _p(idx), _p(dist))
    return idx[:nq].copy(), dist[:nq].copy()


def search_for_triangulation(cam, kp1, desc1, mp1, stereo1, kp2, desc2, mp2, pose1_wc, pose2_wc, max_dist=50):
    kp1 = np.ascontiguousarray(kp1, KEYPOINT); kp2 = np.ascontiguousarray(kp2, KEYPOINT)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8); stereo1 = np.ascontiguousarray(stereo1, np.uint8)
    p1 = np.ascontiguousarray(pose1_wc, np.float64); p2 = np.ascontiguousarray(pose2_wc, np.float64)
    out = np.zeros((max(len(kp1), 1), 2), np.int32)
    n = lib().oracle_search_for_triangulation(C.byref(cam), _p(kp1), _p(desc1), _p(mp1), _p(stereo1), C.c_int(len(kp1)), _p(kp2),
                                              _p(desc2), _p(mp2), C.c_int(len(kp2)), _p(p1), _p(p2), C.c_uint(max_dist), _p(out))
    return out[:n].copy()


def fuse_search(cam, positions, mp_desc, kf_poses_wc, kf_feat_offset, kps, descs, radius_scale, desc_threshold=50):
    positions = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
    poses = np.ascontiguousarray(kf_poses_wc, np.float64).reshape(-1, 7)
    off = np.ascontiguousarray(kf_feat_offset, np.int32)
    kps = np.ascontiguousarray(kps, KEYPOINT); descs = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
    P, T = len(positions), len(poses)
    idx = np.full((P, T), -1, np.int32); dist = np.zeros((P, T), np.uint32)
    lib().oracle_fuse_search(C.byref(cam), _p(positions), _p(mp_desc), C.c_int(P), _p(poses), _p(off), _p(kps), _p(descs),
                             C.c_int(T), C.c_double(radius_scale), C.c_uint(desc_threshold), _p(idx), _p(dist))
    return idx, dist


class Vocabulary:
    """OrbVocabulary (src/vocabulary/mod.rs): load_from_text / from arrays, transform."""

    def __init__(self, ptr):
        if not ptr:
            raise ValueError("vocabulary could not be loaded")
        self._p = C.c_void_p(ptr)
        k, l, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().oracle_vocab_info(self._p, C.byref(k), C.byref(l), C.byref(nn), C.byref(nw))
        self.k, self.l, self.n_nodes, self.n_words = k.value, l.value, nn.value, nw.value

    @classmethod
    def load_from_text(cls, path):
        lib().oracle_vocab_load_text.restype = C.c_void_p
        return cls(lib().oracle_vocab_load_text(str(path).encode()))

    @classmethod
    def from_arrays(cls, parent, is_leaf, desc, weight, k=10, l=6):
        parent = np.ascontiguousarray(parent, np.uint32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); weight = np.ascontiguousarray(weight, np.float64)
        lib().oracle_vocab_from_arrays.restype = C.c_void_p
        return cls(lib().oracle_vocab_from_arrays(C.c_int(len(parent)), _p(parent), _p(is_leaf), _p(desc), _p(weight), C.c_int(k), C.c_int(l)))

    def arrays(self):
        parent = np.zeros(self.n_nodes, np.uint32); leaf = np.zeros(self.n_nodes, np.uint8)
        desc = np.zeros((self.n_nodes, 32), np.uint8); weight = np.zeros(self.n_nodes, np.float64)
        lib().oracle_vocab_arrays(self._p, _p(parent), _p(leaf), _p(desc), _p(weight))
        return parent, leaf, desc, weight

    def transform(self, desc, levels_up=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        word = np.zeros(n, np.uint32); leaf = np.zeros(n, np.uint32); node = np.zeros(n, np.uint32); weight = np.zeros(n, np.float64)
        lib().oracle_bow_transform(self._p, _p(desc), C.c_int(n), C.c_int(levels_up), _p(word), _p(leaf), _p(node), _p(weight))
        return word, leaf, node, weight

    def __del__(self):
        try:
            lib().oracle_vocab_free(self._p)
        except Exception:
            pass


def search_for_triangulation_bow(cam, kp1, desc1, mp1, stereo1, node1, kp2, desc2, mp2, node2, pose1_wc, pose2_wc, max_dist=50):
    kp1 = np.ascontiguousarray(kp1, KEYPOINT); kp2 = np.ascontiguousarray(kp2, KEYPOINT)
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8); stereo1 = np.ascontiguousarray(stereo1, np.uint8)
    node1 = np.ascontiguousarray(node1, np.uint32); node2 = np.ascontiguousarray(node2, np.uint32)
    p1 = np.ascontiguousarray(pose1_wc, np.float64); p2 = np.ascontiguousarray(pose2_wc, np.float64)
    out = np.zeros((max(len(kp1), 1), 2), np.int32)
    n = lib().oracle_search_for_triangulation_bow(C.byref(cam), _p(kp1), _p(desc1), _p(mp1), _p(stereo1), _p(node1), C.c_int(len(kp1)),
                                                  _p(kp2), _p(desc2), _p(mp2), _p(node2), C.c_int(len(kp2)), _p(p1), _p(p2),
                                                  C.c_uint(max_dist), _p(out))
    return out[:n].copy()


def orb_level_table(w, h, params):
    T = OrbLevels()
    rc = lib().oracle_orb_level_table(C.c_int(w), C.c_int(h), C.byref(params), C.byref(T))
    assert rc == 0
    return T


def orb_pyramid_level(img, params, level):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    T = orb_level_table(w, h, params)
    out = np.zeros((T.h[level], T.w[level]), np.uint8)
    rc = lib().oracle_orb_pyramid_level(_p(img), C.c_size_t(img.strides[0]), C.c_int(w), C.c_int(h),
                                        C.byref(params), C.c_int(level), _p(out))
    assert rc == 0
    return out


def orb_blur_level(img, params, level):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    T = orb_level_table(w, h, params)
    out = np.zeros((T.h[level], T.w[level]), np.uint8)
    rc = lib().oracle_orb_blur_level(_p(img), C.c_size_t(img.strides[0]), C.c_int(w), C.c_int(h),
                                     C.byref(params), C.c_int(level), _p(out))
    assert rc == 0
    return out


def orb_fast_level(img, params, level):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = (w * h) // 4 + 16
    out = np.zeros(cap, np.uint32)
    n = lib().oracle_orb_fast_level(_p(img), C.c_size_t(img.strides[0]), C.c_int(w), C.c_int(h),
                                    C.byref(params), C.c_int(level), _p(out), C.c_int(cap))
    assert n >= 0
    return out[:n].copy()


def orb_extract(img, params, cap=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = cap or (params.n_features * 2 + 4096)
    kp = np.zeros(cap, KEYPOINT)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().oracle_orb_extract(_p(img), C.c_size_t(img.strides[0]), C.c_int(w), C.c_int(h),
                                 C.byref(params), _p(kp), _p(desc), C.c_int(cap))
    if n < 0:
        if n <= -1000000000:
            raise ValueError("unsupported ORB parameters")
        return orb_extract(img, params, cap=-n)
    return kp[:n].copy(), desc[:n].copy()


def fast_atan2(y, x):
    return float(lib().oracle_fast_atan2(C.c_float(y), C.c_float(x)))


def sincos_deg(a):
    c = C.c_float(); s = C.c_float()
    lib().oracle_sincos_deg(C.c_float(a), C.byref(c), C.byref(s))
    return c.value, s.value


def se3_to_params(pose7):
    pose7 = np.ascontiguousarray(pose7, np.float64); out = np.zeros(6)
    lib().oracle_se3_to_params(_p(pose7), _p(out)); return out


def se3_from_params(p6):
    p6 = np.ascontiguousarray(p6, np.float64); out = np.zeros(7)
    lib().oracle_se3_from_params(_p(p6), _p(out)); return out


def se3_inverse(pose7):
    pose7 = np.ascontiguousarray(pose7, np.float64); out = np.zeros(7)
    lib().oracle_se3_inverse(_p(pose7), _p(out)); return out


def ba_obs_terms(cam, huber, pose7, X, u, v):
    pose7 = np.ascontiguousarray(pose7, np.float64); X = np.ascontiguousarray(X, np.float64)
    e = np.zeros(2); r = np.zeros(2); A = np.zeros((2, 6)); B = np.zeros((2, 3)); sw = C.c_double()
    lib().oracle_ba_obs_terms(C.byref(cam), C.c_double(huber), _p(pose7), _p(X), C.c_double(u),
                              C.c_double(v), _p(e), C.byref(sw), _p(r), _p(A), _p(B))
    return e, sw.value, r, A, B


def _ba_solve(fn, cam, cfg, poses_cw, fixed_cw, points, obs, stop_after=-1):
    poses_cw = np.ascontiguousarray(poses_cw, np.float64).reshape(-1, 7)
    fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
    pts = np.array(points, np.float64, copy=True).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, BA_OBS)
    K, F, M, N = len(poses_cw), len(fixed_cw), len(pts), len(obs)
    out_wc = np.zeros((max(K, 1), 7))
    it = C.c_int(); e0 = C.c_double(); e1 = C.c_double()
    trace = np.zeros((max(cfg.max_iterations, 1), 4))
    rc = fn(C.byref(cam), C.byref(cfg), C.c_int(K), _p(poses_cw), C.c_int(F), _p(fixed_cw),
            C.c_int(M), _p(pts), C.c_int(N), _p(obs), C.c_int(stop_after), _p(out_wc),
            C.byref(it), C.byref(e0), C.byref(e1), _p(trace))
    if rc != 0:
        return None
    return dict(poses_wc=out_wc[:K], points=pts, iterations=it.value, initial_error=e0.value,
                final_error=e1.value, trace=trace[:it.value])


def ba_solve_dense(cam, cfg, poses_cw, fixed_cw, points, obs, stop_after=-1):
    return _ba_solve(lib().oracle_ba_solve_dense, cam, cfg, poses_cw, fixed_cw, points, obs, stop_after)


def ba_solve_schur(cam, cfg, poses_cw, fixed_cw, points, obs, stop_after=-1):
    return _ba_solve(lib().oracle_ba_solve_schur, cam, cfg, poses_cw, fixed_cw, points, obs, stop_after)


def global_ba_solve_dense(cam, cfg, poses_cw, fixed_cw, points, obs, stop_after=-1):
    return _ba_solve(lib().oracle_global_ba_solve_dense, cam, cfg, poses_cw, fixed_cw, points, obs, stop_after)


def global_ba_solve_schur(cam, cfg, poses_cw, fixed_cw, points, obs, stop_after=-1):
    return _ba_solve(lib().oracle_global_ba_solve_schur, cam, cfg, poses_cw, fixed_cw, points, obs, stop_after)


def inertial_imu_residual(state_i9, state_j9, preint11):
    a = np.ascontiguousarray(state_i9, np.float64); b = np.ascontiguousarray(state_j9, np.float64)
    pr = np.ascontiguousarray(preint11, np.float64); r = np.zeros(9)
    lib().oracle_inertial_imu_residual(_p(a), _p(b), _p(pr), _p(r))
    return r


def inertial_ba_solve(cam, cfg, poses_wc, velocities, biases, fixed_cw, points, obs, edge_kf, preint, stop_after=-1):
    poses_wc = np.ascontiguousarray(poses_wc, np.float64).reshape(-1, 7)
    vel = np.ascontiguousarray(velocities, np.float64).reshape(-1, 3); bias = np.ascontiguousarray(biases, np.float64).reshape(-1, 6)
    fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
    pts = np.array(points, np.float64, copy=True).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, BA_OBS)
    ek = np.ascontiguousarray(edge_kf, np.int32).reshape(-1, 2); pre = np.ascontiguousarray(preint, np.float64).reshape(-1, 11)
    K, F, M, N, E = len(poses_wc), len(fixed_cw), len(pts), len(obs), len(ek)
    out_p = np.zeros((max(K, 1), 7)); out_v = np.zeros((max(K, 1), 3)); out_b = np.zeros((max(K, 1), 6))
    it = C.c_int(); e0 = C.c_double(); e1 = C.c_double()
    trace = np.zeros((max(cfg.max_iterations, 1), 4))
    rc = lib().oracle_inertial_ba_solve(C.byref(cam), C.byref(cfg), C.c_int(K), _p(poses_wc), _p(vel), _p(bias), C.c_int(F), _p(fixed_cw),
                                        C.c_int(M), _p(pts), C.c_int(N), _p(obs), C.c_int(E), _p(ek), _p(pre), C.c_int(stop_after),
                                        _p(out_p), _p(out_v), _p(out_b), C.byref(it), C.byref(e0), C.byref(e1), _p(trace))
    if rc != 0:
        return None
    return dict(poses_wc=out_p[:K], velocities=out_v[:K], biases=out_b[:K], points=pts, iterations=it.value,
                initial_error=e0.value, final_error=e1.value, trace=trace[:it.value])


def ba_reduced_system(cam, cfg, lam, params_pose, fixed_cw, points, obs):
    params_pose = np.ascontiguousarray(params_pose, np.float64).reshape(-1)
    K = len(params_pose) // 6
    fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
    points = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, BA_OBS)
    U = np.zeros((K, 6, 6)); gp = np.zeros(6 * K); S = np.zeros((6 * K, 6 * K)); b = np.zeros(6 * K)
    chi2 = C.c_double()
    rc = lib().oracle_ba_reduced_system(C.byref(cam), C.byref(cfg), C.c_double(lam), C.c_int(K),
                                        _p(params_pose), C.c_int(len(fixed_cw)), _p(fixed_cw),
                                        C.c_int(len(points)), _p(points), C.c_int(len(obs)), _p(obs),
                                        _p(U), _p(gp), _p(S), _p(b), C.byref(chi2))
    assert rc == 0
    return U, gp, S, b, chi2.value
