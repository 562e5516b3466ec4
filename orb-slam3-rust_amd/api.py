"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

Rust is not in the toolchain here, so the thin host layer a Rust shim would be is
written in Python with the reference's own names and argument meaning:

    reference (Rust)                                     here
    ---------------------------------------------------  -----------------------------------------
    CameraModel                  camera.rs:3-10          CameraModel
    StereoProcessor::new/process stereo.rs:37,52         StereoProcessor(camera, n_features).process
    FeatureSet / StereoFrame     stereo.rs:15-29         FeatureSet / StereoFrame
    descriptor_distance          stereo.rs:166           descriptor_distance
    BFMatcher(HAMMING,true)      tracker.rs:1001-1010    bf_match_crosscheck
    LocalBAConfigLM              local_ba_lm.rs:96-119   LocalBAConfigLM
    VisualBAProblemData/Result   local_ba_lm.rs:48-93    VisualBAProblemData / VisualBAResultData
    solve_visual_ba              local_ba_lm.rs:912      solve_visual_ba

Everything computes on the GPU through liborbx_hip.so; there is no CPU fallback, and a missing
library or device raises.
"""
import atexit
import ctypes as C
import math
import os
import weakref
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from .build import LIB_PATH

KEYPOINT = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH = np.dtype([("query_idx", "<i4"), ("train_idx", "<i4"), ("img_idx", "<i4"),
                   ("distance", "<f4")])
BA_OBS = np.dtype([("kf_idx", "<i4"), ("fixed_idx", "<i4"), ("mp_idx", "<i4"), ("_pad", "<i4"),
                   ("u", "<f8"), ("v", "<f8")])
# orbx_ba_obs32 (include/orbx.h): the same observation in 16 bytes — kf_idx >= 0 optimised keyframe, < 0 fixed observer -1 - kf_idx (F = identity); u, v f32
BA_OBS32 = np.dtype([("kf_idx", "<i4"), ("mp_idx", "<i4"), ("u", "<f4"), ("v", "<f4")])


def ba_obs_to_obs32(obs, n_fixed):
    """orbx_ba_obs -> orbx_ba_obs32, or None if a coordinate is not exactly an f32 (the reference's always are: kp.pt() widened, local_ba_lm.rs:870-872)."""
    obs = np.ascontiguousarray(obs, BA_OBS)
    u32, v32 = obs["u"].astype(np.float32), obs["v"].astype(np.float32)
    if not (np.array_equal(u32.astype(np.float64), obs["u"]) and np.array_equal(v32.astype(np.float64), obs["v"])):
        return None
    out = np.zeros(len(obs), BA_OBS32)
    fixed = np.where(obs["fixed_idx"] >= 0, obs["fixed_idx"], n_fixed)
    out["kf_idx"] = np.where(obs["kf_idx"] >= 0, obs["kf_idx"], -1 - fixed)
    out["mp_idx"] = obs["mp_idx"]; out["u"] = u32; out["v"] = v32
    return out



ORBX_OK, ORBX_ERR_INVALID, ORBX_ERR_NO_DEVICE, ORBX_ERR_HIP = 0, -1, -2, -3
ORBX_ERR_CAPACITY, ORBX_ERR_NUMERIC, ORBX_ERR_EMPTY = -4, -5, -6

# ORB-SLAM3 matching thresholds, stereo.rs:10-12
TH_HIGH, TH_LOW, NN_RATIO = 100, 50, 0.75

# every symbol include/orbx.h declares
ABI_VERSION = 2      # = ORBX_ABI_VERSION of include/orbx.h, whose struct layouts the ctypes mirrors below restate
ABI_SYMBOLS = [
    "orbx_version", "orbx_abi_version", "orbx_last_error", "orbx_default_orb_params", "orbx_create", "orbx_destroy",
    "orbx_stream", "orbx_synchronize", "orbx_process_stereo", "orbx_process_stereo_batch_device",
    "orbx_process_stereo_batch", "orbx_host_alloc", "orbx_host_free",
    "orbx_extract_batch_device", "orbx_check_status", "orbx_stereo_match",
    "orbx_stereo_match_batch_device", "orbx_hamming_match_crosscheck",
    "orbx_hamming_match_crosscheck_device", "orbx_hamming_batch", "orbx_hamming_batch_device",
    "orbx_keyframe_create", "orbx_keyframe_destroy", "orbx_keyframe_info", "orbx_keyframe_set_pose", "orbx_keyframe_set_map_points",
    "orbx_keyframe_get_map_points", "orbx_keyframe_download", "orbx_keyframe_device_keypoints", "orbx_keyframe_device_descriptors",
    "orbx_keyframe_guided_match", "orbx_keyframe_search_for_triangulation", "orbx_keyframe_fuse_search",
    "orbx_default_ba_config", "orbx_ba_set_allreduce", "orbx_rccl_unique_id", "orbx_ba_init_rccl", "orbx_ba_set_rccl_comm", "orbx_ba_has_collective", "orbx_ba_rccl_world", "orbx_ba_solve_visual", "orbx_ba_solve_visual_obs32", "orbx_ba_solve_global_obs32", "orbx_ba_solve_visual_batch", "orbx_debug_ba_blocks", "orbx_debug_imu_residual", "orbx_ba_solve_global", "orbx_default_inertial_ba_config", "orbx_ba_solve_inertial",
    "orbx_guided_match", "orbx_guided_match_device", "orbx_search_for_triangulation", "orbx_search_for_triangulation_device", "orbx_search_for_triangulation_bow",
    "orbx_fuse_search", "orbx_fuse_search_device",
    "orbx_vocab_load_text", "orbx_vocab_create", "orbx_vocab_destroy", "orbx_vocab_info", "orbx_vocab_nodes",
    "orbx_bow_transform", "orbx_bow_transform_device", "orbx_bow_vectors", "orbx_bow_vectors_device", "orbx_bow_score",
    "orbx_png_decode_gray8", "orbx_euroc_open", "orbx_euroc_close", "orbx_euroc_len", "orbx_euroc_last_error",
    "orbx_euroc_frame_timestamp", "orbx_euroc_calibration", "orbx_euroc_read_pairs",
    "orbx_set_profiling", "orbx_set_profiling_only", "orbx_get_kernel_times", "orbx_debug_read_level",
    "orbx_debug_read_candidates",
]


BOW_VECTORS_MAX = 8192      # descriptors per orbx_bow_vectors call (bow_kernels.hip: BOWV_MAX)


def _seq_sum(a):
    """left-to-right f64 sum (the order the device kernel adds a run in)"""
    t = 0.0
    for x in a:
        t += float(x)
    return t


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx error %d: %s" % (code, msg))
        self.code = code


class _Camera(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "baseline")]


class _OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int), ("scale_factor", C.c_float), ("n_levels", C.c_int),
                ("edge_threshold", C.c_int), ("first_level", C.c_int), ("wta_k", C.c_int),
                ("score_type", C.c_int), ("patch_size", C.c_int), ("fast_threshold", C.c_int)]


class _BaConfig(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("param_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("huber_threshold", C.c_double),
                ("max_covisible_keyframes", C.c_int)]


class _InertialBaConfig(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("window_size", C.c_int), ("huber_threshold_mono", C.c_double),
                ("huber_threshold_stereo", C.c_double), ("initial_lambda", C.c_double), ("gyro_rw_info", C.c_double),
                ("accel_rw_info", C.c_double)]


class _KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_int)]


class _BaWindow(C.Structure):
    """orbx_ba_window (include/orbx.h)"""
    _fields_ = [("K", C.c_int), ("poses_cw", C.c_void_p), ("F", C.c_int), ("fixed_poses_cw", C.c_void_p), ("M", C.c_int),
                ("points", C.c_void_p), ("N", C.c_int), ("obs", C.c_void_p), ("poses_wc_out", C.c_void_p), ("status", C.c_int),
                ("iterations", C.c_int), ("initial_error", C.c_double), ("final_error", C.c_double), ("obs32", C.c_void_p)]


SHOULD_STOP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None
_live_handles = weakref.WeakSet()


@atexit.register
def _close_live_handles():
    # destroy handles while the HIP runtime is still loaded (interpreter teardown order is arbitrary)
    for h in list(_live_handles):
        try:
            h.close()
        except Exception:
            pass


def load_library():
    """dlopen liborbx_hip.so.  Raises if it has not been built: the product has no other path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liborbx_hip.so is missing (%s): build it with __graft_entry__.build(); "
                "this package has no CPU or PyTorch fallback" % LIB_PATH)
        if not os.environ.get("ORBX_NO_TORCH_PRELOAD"):
            # PyTorch-ROCm wheels carry their own libamdhip64.so.7; two HIP runtimes in one process
            # do not coexist ("No HIP GPUs are available").  Loading torch first makes the dynamic
            # linker bind this library's libamdhip64.so.7 dependency to the copy torch already
            # mapped (same soname), so device memory and streams are shared with torch.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.orbx_version.restype = C.c_char_p
        if L.orbx_abi_version() != ABI_VERSION:
            raise RuntimeError("liborbx_hip.so was built from an orbx.h of ABI version %d, this mirror restates version %d: rebuild with "
                               "__graft_entry__.build()" % (L.orbx_abi_version(), ABI_VERSION))
        L.orbx_last_error.restype = C.c_char_p
        L.orbx_last_error.argtypes = [C.c_void_p]
        L.orbx_stream.restype = C.c_void_p
        L.orbx_stream.argtypes = [C.c_void_p]
        L.orbx_destroy.argtypes = [C.c_void_p]
        L.orbx_euroc_close.argtypes = [C.c_void_p]
        L.orbx_euroc_close.restype = None
        L.orbx_euroc_len.argtypes = [C.c_void_p]
        L.orbx_euroc_last_error.argtypes = [C.c_void_p]
        L.orbx_euroc_last_error.restype = C.c_char_p
        L.orbx_euroc_open.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.orbx_euroc_frame_timestamp.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orbx_euroc_calibration.argtypes = [C.c_void_p] * 5
        L.orbx_euroc_read_pairs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orbx_png_decode_gray8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orbx_vocab_destroy.argtypes = [C.c_void_p]
        L.orbx_vocab_destroy.restype = None
        L.orbx_vocab_info.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orbx_vocab_nodes.argtypes = [C.c_void_p] * 5
        L.orbx_vocab_load_text.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        L.orbx_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L.orbx_destroy.restype = None
        for name in ABI_SYMBOLS:
            getattr(L, name)
        _lib = L
    return _lib


def _vp(x):
    """void* from a numpy array, an int device pointer, a torch tensor, or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(C.c_void_p)
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


@dataclass
class CameraModel:
    """camera.rs:3-10"""
    fx: float
    fy: float
    cx: float
    cy: float
    baseline: float

    def _c(self):
        return _Camera(self.fx, self.fy, self.cx, self.cy, self.baseline)


@dataclass
class LocalBAConfigLM:
    """local_ba_lm.rs:96-119 (Default impl :109-119)"""
    max_iterations: int = 10
    param_tolerance: float = 1e-8
    gradient_tolerance: float = 1e-8
    huber_threshold: float = math.sqrt(5.991)
    max_covisible_keyframes: int = 20

    def _c(self):
        return _BaConfig(self.max_iterations, self.param_tolerance, self.gradient_tolerance,
                         self.huber_threshold, self.max_covisible_keyframes)


@dataclass
class GlobalBAConfig:
    """global_ba.rs:21-46 (Default impl :36-45)"""
    max_iterations: int = 10
    param_tolerance: float = 1e-6
    gradient_tolerance: float = 1e-6
    huber_threshold: float = math.sqrt(5.991)

    def _c(self):
        return _BaConfig(self.max_iterations, self.param_tolerance, self.gradient_tolerance, self.huber_threshold, 0)


@dataclass
class LocalInertialBAConfig:
    """local_inertial_ba.rs:109-141 (Default impl :126-141)"""
    max_iterations: int = 10
    window_size: int = 10
    huber_threshold_mono: float = math.sqrt(5.991)
    huber_threshold_stereo: float = math.sqrt(7.815)
    initial_lambda: float = 1e-2
    gyro_rw_info: float = 1e6
    accel_rw_info: float = 1e4

    def _c(self):
        return _InertialBaConfig(self.max_iterations, self.window_size, self.huber_threshold_mono, self.huber_threshold_stereo,
                                 self.initial_lambda, self.gyro_rw_info, self.accel_rw_info)


@dataclass
class FeatureSet:
    """stereo.rs:15-19: keypoints (KEYPOINT records) + descriptors [N,32] u8"""
    keypoints: np.ndarray
    descriptors: np.ndarray


@dataclass
class StereoFrame:
    """stereo.rs:21-29.  points_cam is [nL,3] f64 with `has_point` as the Option mask."""
    left_features: FeatureSet
    right_features: FeatureSet
    matches_lr: np.ndarray
    points_cam: np.ndarray
    has_point: np.ndarray
    timestamp_ns: int

    def points_cam_options(self) -> List[Optional[np.ndarray]]:
        return [self.points_cam[i] if self.has_point[i] else None for i in range(len(self.has_point))]


class Handle:
    """Owns one orbx_handle (one HIP stream on one device).  Not thread-safe, like `&mut self`."""

    def __init__(self, camera: CameraModel, n_features: int, device: int = 0, max_w: int = 752,
                 max_h: int = 480, max_batch: int = 1, orb_params=None):
        L = load_library()
        self._L = L
        p = _OrbParams()
        L.orbx_default_orb_params(C.c_int(n_features), C.byref(p))
        if orb_params:
            for k, v in orb_params.items():
                setattr(p, k, v)
        self.orb_params = p
        self.camera = camera
        self._h = C.c_void_p()
        cam = camera._c()
        rc = L.orbx_create(C.byref(cam), C.byref(p), C.c_int(device), C.c_int(max_w), C.c_int(max_h),
                           C.c_int(max_batch), C.byref(self._h))
        if rc != 0:
            raise OrbxError(rc, L.orbx_last_error(None).decode())
        self.device = device
        self._keep = []
        self._ext_stream = None
        _live_handles.add(self)

    def _after_torch(self, *tensors):
        """Order the library's (non-blocking) stream after whatever torch has queued on its current stream — the
        caller's tensors may still be being written there — and keep the inputs alive until the next synchronise."""
        import torch
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        if self._ext_stream is None:
            self._ext_stream = torch.cuda.ExternalStream(self.stream, device=torch.device("cuda", self.device))
        self._ext_stream.wait_event(ev)
        self._keep.extend(tensors)

    def close(self):
        self._ext_stream = None
        self._keep = []
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise OrbxError(rc, self._L.orbx_last_error(self._h).decode())

    @property
    def stream(self):
        return self._L.orbx_stream(self._h)

    def synchronize(self):
        self._check(self._L.orbx_synchronize(self._h))
        self._keep = []

    def check_status(self):
        rc = self._L.orbx_check_status(self._h)
        self._keep = []
        self._check(rc)

    def set_profiling(self, on=True, only=None):
        """HIP events around every launch (on), or around the launches of the one kernel `only` names."""
        if on and only:
            self._check(self._L.orbx_set_profiling_only(self._h, C.c_char_p(only.encode())))
        else:
            self._check(self._L.orbx_set_profiling(self._h, C.c_int(1 if on else 0)))

    def kernel_times(self):
        arr = (_KernelTime * 64)()
        n = self._L.orbx_get_kernel_times(self._h, arr, C.c_int(64))
        return {arr[i].name.decode(): (arr[i].ms, arr[i].launches) for i in range(min(n, 64))}

    # ---- host-buffer entry points -------------------------------------------------------------
    def process_stereo(self, left, right, cap_kp=None):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        if left.ndim != 2 or left.shape != right.shape:
            raise ValueError("left/right must be 2-d u8 images of equal size")
        hh, ww = left.shape
        cap = cap_kp or (self.orb_params.n_features + 2048)
        kpL = np.zeros(cap, KEYPOINT); kpR = np.zeros(cap, KEYPOINT)
        dL = np.zeros((cap, 32), np.uint8); dR = np.zeros((cap, 32), np.uint8)
        m = np.zeros(cap, DMATCH); pts = np.zeros((cap, 3), np.float64); has = np.zeros(cap, np.uint8)
        nL = C.c_int(); nR = C.c_int(); nm = C.c_int()
        self._check(self._L.orbx_process_stereo(
            self._h, _vp(left), C.c_size_t(left.strides[0]), _vp(right), C.c_size_t(right.strides[0]),
            C.c_int(ww), C.c_int(hh), _vp(kpL), _vp(dL), C.byref(nL), _vp(kpR), _vp(dR), C.byref(nR),
            C.c_int(cap), _vp(m), C.byref(nm), _vp(pts), _vp(has)))
        return (kpL[:nL.value].copy(), dL[:nL.value].copy(), kpR[:nR.value].copy(), dR[:nR.value].copy(),
                m[:nm.value].copy(), pts[:nL.value].copy(), has[:nL.value].copy())

    def stereo_match(self, kpL, descL, kpR, descR):
        kpL = np.ascontiguousarray(kpL, KEYPOINT); kpR = np.ascontiguousarray(kpR, KEYPOINT)
        descL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
        descR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
        nL, nR = len(kpL), len(kpR)
        m = np.zeros(max(nL, 1), DMATCH); pts = np.zeros((max(nL, 1), 3)); has = np.zeros(max(nL, 1), np.uint8)
        nm = C.c_int()
        self._check(self._L.orbx_stereo_match(self._h, _vp(kpL), _vp(descL), C.c_int(nL), _vp(kpR),
                                              _vp(descR), C.c_int(nR), _vp(m), C.byref(nm), _vp(pts), _vp(has)))
        return m[:nm.value].copy(), pts[:nL].copy(), has[:nL].copy()

    def hamming_match_crosscheck(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(q), 1), DMATCH)
        n = C.c_int()
        self._check(self._L.orbx_hamming_match_crosscheck(self._h, _vp(q), C.c_int(len(q)), _vp(t),
                                                          C.c_int(len(t)), _vp(out), C.byref(n)))
        return out[:n.value].copy()

    def guided_match(self, kp, desc, img_w, img_h, q_uv, q_desc, radius, mode):
        """FeatureGrid + descriptor search (tracking_frame.rs:52-128; tracker.rs:880-923 mode 1, :1126-1157 mode 0)."""
        kp = np.ascontiguousarray(kp, KEYPOINT); desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        q_uv = np.ascontiguousarray(q_uv, np.float64).reshape(-1, 2)
        q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(q_uv)
        idx = np.zeros(max(nq, 1), np.int32); dist = np.zeros(max(nq, 1), np.uint32)
        self._check(self._L.orbx_guided_match(self._h, _vp(kp), _vp(desc), C.c_int(len(kp)), C.c_double(img_w),
                                              C.c_double(img_h), _vp(q_uv), _vp(q_desc), C.c_int(nq), C.c_double(radius),
                                              C.c_int(mode), _vp(idx), _vp(dist)))
        return idx[:nq].copy(), dist[:nq].copy()

    def search_for_triangulation(self, camera, kp1, desc1, mp1, stereo1, kp2, desc2, mp2, pose1_wc, pose2_wc, max_dist=50):
        """triangulation.rs:401-527.  Returns [(idx1, idx2)] as an int32 [n,2] array, ascending idx1."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT); kp2 = np.ascontiguousarray(kp2, KEYPOINT)
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8)
        stereo1 = np.ascontiguousarray(stereo1, np.uint8)
        p1 = np.ascontiguousarray(pose1_wc, np.float64); p2 = np.ascontiguousarray(pose2_wc, np.float64)
        out = np.zeros((max(len(kp1), 1), 2), np.int32)
        n = C.c_int()
        cam = camera._c()
        self._check(self._L.orbx_search_for_triangulation(
            self._h, C.byref(cam), _vp(kp1), _vp(desc1), _vp(mp1), _vp(stereo1), C.c_int(len(kp1)), _vp(kp2), _vp(desc2),
            _vp(mp2), C.c_int(len(kp2)), _vp(p1), _vp(p2), C.c_uint(max_dist), _vp(out), C.byref(n)))
        return out[:n.value].copy()

    def search_for_triangulation_device(self, camera, kp1, desc1, mp1, stereo1, kp2, desc2, mp2, pose1_wc, pose2_wc, max_dist=50):
        """Device-resident form: torch CUDA tensors (kp [n,7] f32 as the extractor writes them, desc [n,32] u8, flags [n] u8).
        Returns (pairs [n1,2] int32 tensor, count [1] int32 tensor); asynchronous."""
        import torch
        n1, n2 = kp1.shape[0], kp2.shape[0]
        pairs = torch.empty((max(n1, 1), 2), dtype=torch.int32, device=kp1.device)
        cnt = torch.zeros(1, dtype=torch.int32, device=kp1.device)
        p1 = np.ascontiguousarray(pose1_wc, np.float64); p2 = np.ascontiguousarray(pose2_wc, np.float64)
        cam = camera._c()
        self._after_torch(kp1, desc1, mp1, stereo1, kp2, desc2, mp2, pairs, cnt)
        self._check(self._L.orbx_search_for_triangulation_device(
            self._h, C.byref(cam), _vp(kp1), _vp(desc1), _vp(mp1), _vp(stereo1), C.c_int(n1), _vp(kp2), _vp(desc2), _vp(mp2), C.c_int(n2),
            _vp(p1), _vp(p2), C.c_uint(max_dist), _vp(pairs), _vp(cnt)))
        return pairs, cnt

    def fuse_search_device(self, camera, positions, mp_desc, kf_poses_wc, kf_feat_offset, kps, descs, radius_scale, desc_threshold=TH_LOW):
        """Device-resident form of fuse_search: torch CUDA tensors except the poses.  Returns (idx [P,T] int32, dist [P,T] int32 view of u32)."""
        import torch
        P_, T = positions.shape[0], len(kf_poses_wc)
        idx = torch.empty((P_, T), dtype=torch.int32, device=positions.device)
        dist = torch.empty((P_, T), dtype=torch.int32, device=positions.device)
        poses = np.ascontiguousarray(kf_poses_wc, np.float64).reshape(-1, 7)
        cam = camera._c()
        self._after_torch(positions, mp_desc, kf_feat_offset, kps, descs, idx, dist)
        self._check(self._L.orbx_fuse_search_device(self._h, C.byref(cam), _vp(positions), _vp(mp_desc), C.c_int(P_), _vp(poses),
                                                    _vp(kf_feat_offset), _vp(kps), _vp(descs), C.c_int(T), C.c_double(radius_scale),
                                                    C.c_uint(desc_threshold), _vp(idx), _vp(dist)))
        return idx, dist

    def search_for_triangulation_bow(self, camera, kp1, desc1, mp1, stereo1, node1, kp2, desc2, mp2, node2, pose1_wc, pose2_wc,
                                     max_dist=50):
        """triangulation.rs:541-658.  node1/node2: FeatureVector key per feature.  int32 [n,2], ascending idx1."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT); kp2 = np.ascontiguousarray(kp2, KEYPOINT)
        desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8)
        stereo1 = np.ascontiguousarray(stereo1, np.uint8)
        node1 = np.ascontiguousarray(node1, np.uint32); node2 = np.ascontiguousarray(node2, np.uint32)
        p1 = np.ascontiguousarray(pose1_wc, np.float64); p2 = np.ascontiguousarray(pose2_wc, np.float64)
        out = np.zeros((max(len(kp1), 1), 2), np.int32)
        n = C.c_int()
        cam = camera._c()
        self._check(self._L.orbx_search_for_triangulation_bow(
            self._h, C.byref(cam), _vp(kp1), _vp(desc1), _vp(mp1), _vp(stereo1), _vp(node1), C.c_int(len(kp1)), _vp(kp2), _vp(desc2),
            _vp(mp2), _vp(node2), C.c_int(len(kp2)), _vp(p1), _vp(p2), C.c_uint(max_dist), _vp(out), C.byref(n)))
        return out[:n.value].copy()

    def fuse_search(self, camera, positions, mp_desc, kf_poses_wc, kf_feat_offset, kps, descs, radius_scale, desc_threshold=TH_LOW):
        """search_in_neighbors.rs:273-343 for every (map point, keyframe) pair -> (idx [P,T] int32, dist [P,T] uint32)."""
        positions = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
        mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        poses = np.ascontiguousarray(kf_poses_wc, np.float64).reshape(-1, 7)
        off = np.ascontiguousarray(kf_feat_offset, np.int32)
        kps = np.ascontiguousarray(kps, KEYPOINT); descs = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
        P, T = len(positions), len(poses)
        if len(off) != T + 1 or len(mp_desc) != P or len(kps) != len(descs):
            raise ValueError("fuse_search: inconsistent array lengths")
        idx = np.full((P, T), -1, np.int32); dist = np.zeros((P, T), np.uint32)
        cam = camera._c()
        self._check(self._L.orbx_fuse_search(self._h, C.byref(cam), _vp(positions), _vp(mp_desc), C.c_int(P), _vp(poses), _vp(off),
                                             _vp(kps), _vp(descs), C.c_int(T), C.c_double(radius_scale), C.c_uint(desc_threshold),
                                             _vp(idx), _vp(dist)))
        return idx, dist

    def hamming_batch(self, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        if len(a) != len(b):
            raise ValueError("hamming_batch: row counts differ")
        out = np.zeros(len(a), np.uint32)
        self._check(self._L.orbx_hamming_batch(self._h, _vp(a), _vp(b), C.c_int(len(a)), _vp(out)))
        return out

    # ---- device-resident batch entry points (torch tensors on this handle's GPU) ----------------
    def alloc_batch_outputs(self, batch, cap_kp):
        import torch
        dev = torch.device("cuda", self.device)
        return dict(
            kp=torch.zeros((batch, 2, cap_kp, 7), dtype=torch.float32, device=dev),
            desc=torch.zeros((batch, 2, cap_kp, 32), dtype=torch.uint8, device=dev),
            nkp=torch.zeros((batch, 2), dtype=torch.int32, device=dev),
            matches=torch.zeros((batch, cap_kp, 4), dtype=torch.int32, device=dev),
            nmatches=torch.zeros((batch,), dtype=torch.int32, device=dev),
            points=torch.zeros((batch, cap_kp, 3), dtype=torch.float64, device=dev),
            has_point=torch.zeros((batch, cap_kp), dtype=torch.uint8, device=dev),
            cap_kp=cap_kp, batch=batch)

    def process_stereo_batch_device(self, images, out):
        """images: torch u8 [batch,2,h,w] on the GPU; out: alloc_batch_outputs().  Asynchronous."""
        b, two, hh, ww = images.shape
        assert two == 2 and images.is_contiguous() and b <= out["batch"]
        self._after_torch(images)
        self._check(self._L.orbx_process_stereo_batch_device(
            self._h, _vp(images), C.c_int(b), C.c_int(ww), C.c_int(hh), C.c_size_t(ww), _vp(out["kp"]),
            _vp(out["desc"]), _vp(out["nkp"]), C.c_int(out["cap_kp"]), _vp(out["matches"]),
            _vp(out["nmatches"]), _vp(out["points"]), _vp(out["has_point"])))

    def process_stereo_batch_host(self, images, out):
        """Pipelined host-buffer form: `images` and every tensor of `out` are (ideally pinned) CPU torch tensors with
        the layouts of alloc_batch_outputs.  Synchronous."""
        b, two, hh, ww = images.shape
        assert two == 2 and images.is_contiguous() and not images.is_cuda and b <= out["batch"]
        self._check(self._L.orbx_process_stereo_batch(
            self._h, _vp(images), C.c_int(b), C.c_int(ww), C.c_int(hh), C.c_size_t(ww), _vp(out["kp"]),
            _vp(out["desc"]), _vp(out["nkp"]), C.c_int(out["cap_kp"]), _vp(out["matches"]),
            _vp(out["nmatches"]), _vp(out["points"]), _vp(out["has_point"])))

    @staticmethod
    def alloc_host_outputs(batch, cap_kp, pin=True):
        import torch
        mk = lambda shape, dt: (torch.zeros(shape, dtype=dt).pin_memory() if pin else torch.zeros(shape, dtype=dt))
        return dict(kp=mk((batch, 2, cap_kp, 7), torch.float32), desc=mk((batch, 2, cap_kp, 32), torch.uint8),
                    nkp=mk((batch, 2), torch.int32), matches=mk((batch, cap_kp, 4), torch.int32),
                    nmatches=mk((batch,), torch.int32), points=mk((batch, cap_kp, 3), torch.float64),
                    has_point=mk((batch, cap_kp), torch.uint8), cap_kp=cap_kp, batch=batch)

    def extract_batch_device(self, images, out):
        """images: torch u8 [n,h,w]; writes out['kp'|'desc'|'nkp'] viewed as n slots."""
        n, hh, ww = images.shape
        assert images.is_contiguous() and n <= 2 * out["batch"]
        self._after_torch(images)
        self._check(self._L.orbx_extract_batch_device(
            self._h, _vp(images), C.c_int(n), C.c_int(ww), C.c_int(hh), C.c_size_t(ww), _vp(out["kp"]),
            _vp(out["desc"]), _vp(out["nkp"]), C.c_int(out["cap_kp"])))

    def stereo_match_batch_device(self, out, batch=None):
        b = batch or out["batch"]
        self._after_torch()
        self._check(self._L.orbx_stereo_match_batch_device(
            self._h, C.c_int(b), _vp(out["kp"]), _vp(out["desc"]), _vp(out["nkp"]), C.c_int(out["cap_kp"]),
            _vp(out["matches"]), _vp(out["nmatches"]), _vp(out["points"]), _vp(out["has_point"])))

    @staticmethod
    def unpack_batch_outputs(out, b):
        """Host copies of pair `b` in the reference's containers (synchronise first)."""
        nkp = out["nkp"][b].cpu().numpy()
        kp = out["kp"][b].cpu().numpy()
        desc = out["desc"][b].cpu().numpy()
        nm = int(out["nmatches"][b].item())
        res = []
        for s in range(2):
            k = np.ascontiguousarray(kp[s, :nkp[s]]).view(np.uint8).reshape(-1, 28).copy().view(KEYPOINT).reshape(-1)
            res.append(FeatureSet(k, desc[s, :nkp[s]].copy()))
        m = np.ascontiguousarray(out["matches"][b, :nm].cpu().numpy()).view(np.uint8).reshape(-1, 16).copy().view(DMATCH).reshape(-1)
        pts = out["points"][b, :nkp[0]].cpu().numpy()
        has = out["has_point"][b, :nkp[0]].cpu().numpy()
        return res[0], res[1], m, pts, has

    # ---- stage inspection -----------------------------------------------------------------------
    def debug_level(self, image_index, level, blurred=False):
        w = C.c_int(); hh = C.c_int()
        self._check(self._L.orbx_debug_read_level(self._h, C.c_int(image_index), C.c_int(level),
                                                  C.c_int(1 if blurred else 0), None, C.byref(w), C.byref(hh)))
        out = np.zeros((hh.value, w.value), np.uint8)
        self._check(self._L.orbx_debug_read_level(self._h, C.c_int(image_index), C.c_int(level),
                                                  C.c_int(1 if blurred else 0), _vp(out), C.byref(w), C.byref(hh)))
        return out

    def debug_candidates(self, image_index, level):
        n = C.c_int()
        self._check(self._L.orbx_debug_read_candidates(self._h, C.c_int(image_index), C.c_int(level), None,
                                                       C.c_int(0), C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        self._check(self._L.orbx_debug_read_candidates(self._h, C.c_int(image_index), C.c_int(level), _vp(out),
                                                       C.c_int(len(out)), C.byref(n)))
        return out[:n.value].copy()

    # ---- BA -----------------------------------------------------------------------------------------
    def set_allreduce(self, fn):
        """fn(dev_ptr:int, n_doubles:int, hip_stream:int) -> None, or None to clear."""
        if fn is None:
            self._ar = None
            self._check(self._L.orbx_ba_set_allreduce(self._h, None, None))
            return

        def tramp(user, ptr, n, stream):
            try:
                fn(int(ptr), int(n), int(stream) if stream else 0)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return -1
        self._ar = ALLREDUCE_FN(tramp)
        self._check(self._L.orbx_ba_set_allreduce(self._h, self._ar, None))

    # ---- native RCCL collective of the point-partitioned solve (include/orbx.h) -----------------------------------
    @staticmethod
    def rccl_unique_id():
        """ncclGetUniqueId through the library (rank 0); hand the bytes to every rank, then init_rccl."""
        buf = (C.c_uint8 * 256)()
        n = load_library().orbx_rccl_unique_id(buf, C.c_size_t(256))
        if n <= 0:
            raise OrbxError(n, "ncclGetUniqueId failed")
        return bytes(buf[:n])

    def init_rccl(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on this handle's device; collective over all ranks.  The library owns the communicator."""
        b = (C.c_uint8 * len(unique_id)).from_buffer_copy(unique_id)
        self._L.orbx_ba_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        self._check(self._L.orbx_ba_init_rccl(self._h, b, C.c_size_t(len(unique_id)), C.c_int(rank), C.c_int(world)))

    def set_rccl_comm(self, comm_ptr):
        """Use an existing ncclComm_t (integer address), e.g. torch's; None / 0 clears."""
        self._L.orbx_ba_set_rccl_comm.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self._L.orbx_ba_set_rccl_comm(self._h, C.c_void_p(comm_ptr or None)))

    def has_collective(self):
        """Bit 0: an RCCL communicator is installed on the handle; bit 1: the all-reduce hook is (orbx_ba_has_collective)."""
        self._L.orbx_ba_has_collective.argtypes = [C.c_void_p]
        self._L.orbx_ba_has_collective.restype = C.c_int
        return int(self._L.orbx_ba_has_collective(self._h))

    def rccl_world(self):
        """(ranks, this rank) of the handle's RCCL communicator, from ncclCommCount / ncclCommUserRank (orbx_ba_rccl_world)."""
        n, r = C.c_int(0), C.c_int(-1)
        self._L.orbx_ba_rccl_world.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._L.orbx_ba_rccl_world.restype = C.c_int
        self._check(self._L.orbx_ba_rccl_world(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def ba_solve_visual(self, camera, cfg, poses_cw, fixed_cw, points, obs, should_stop=None):
        poses_cw = np.ascontiguousarray(poses_cw, np.float64).reshape(-1, 7)
        fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
        pts = np.array(points, np.float64, copy=True).reshape(-1, 3)
        # observations of dtype BA_OBS32 (ba_obs_to_obs32: the 16-byte form, f32 pixel coordinates as the reference's keypoints are) go through
        # orbx_ba_solve_visual_obs32: the same result bit for bit, half the upload
        o32 = getattr(obs, "dtype", None) == BA_OBS32
        obs = np.ascontiguousarray(obs, BA_OBS32 if o32 else BA_OBS)
        K, F, M, N = len(poses_cw), len(fixed_cw), len(pts), len(obs)
        out_wc = np.zeros((max(K, 1), 7))
        it = C.c_int(); e0 = C.c_double(); e1 = C.c_double()
        cb = SHOULD_STOP_FN((lambda user: 1 if should_stop() else 0)) if should_stop else C.cast(None, SHOULD_STOP_FN)
        cam = camera._c(); c = cfg._c()
        fn = self._L.orbx_ba_solve_visual_obs32 if o32 else self._L.orbx_ba_solve_visual
        rc = fn(self._h, C.byref(cam), C.byref(c), C.c_int(K), _vp(poses_cw),
                C.c_int(F), _vp(fixed_cw), C.c_int(M), _vp(pts), C.c_int(N),
                _vp(obs), cb, None, _vp(out_wc), C.byref(it), C.byref(e0), C.byref(e1))
        if rc in (ORBX_ERR_EMPTY,):
            return None                      # reference returns None, local_ba_lm.rs:923-925
        self._check(rc)
        return dict(poses_wc=out_wc[:K], points=pts, iterations=it.value, initial_error=e0.value,
                    final_error=e1.value)

    @staticmethod
    def pack_ba_windows(windows, obs32=False):
        """The same windows with every `obs` array a consecutive slice of ONE page-locked host buffer (window order): what a caller that
        owns its observation storage would hand to orbx_ba_solve_visual_batch — the copy engine then reads the observations where they lie
        and each half of the batch travels as one copy (orbx.h).  The buffer is a pinned torch tensor kept alive by the returned dicts."""
        import torch
        dt = BA_OBS32 if obs32 else BA_OBS
        if obs32:       # (coordinates that are not f32 values cannot take the 16-byte form: ba_obs_to_obs32 answers None)
            arrs = [w["obs"] if getattr(w["obs"], "dtype", None) == BA_OBS32 else ba_obs_to_obs32(np.ascontiguousarray(w["obs"], BA_OBS), len(np.asarray(w["fixed_cw"]).reshape(-1, 7))) for w in windows]
            if any(a is None for a in arrs):
                raise ValueError("pack_ba_windows(obs32=True): an observation's pixel coordinates are not f32 values")
        else:
            arrs = [np.ascontiguousarray(w["obs"], BA_OBS) for w in windows]
        total = sum(len(a) for a in arrs)
        buf = torch.empty(max(total, 1) * dt.itemsize, dtype=torch.uint8).pin_memory()
        flat = buf.numpy().view(dt)
        out, o = [], 0
        for w, a in zip(windows, arrs):
            flat[o:o + len(a)] = a
            d = dict(w); d["obs"] = flat[o:o + len(a)]; d["_pinned_obs"] = buf
            out.append(d); o += len(a)
        return out

    def ba_solve_visual_batch(self, camera, cfg, windows, should_stop=None):
        """orbx_ba_solve_visual_batch: `windows` = list of dicts with poses_cw, fixed_cw, points, obs (as ba_solve_visual).
        Returns one result dict per window (None where the reference returns None)."""
        keep = []
        arr = (_BaWindow * max(len(windows), 1))()
        # the in/out points and the output poses of ALL windows live in two arrays allocated once per call (a fresh 48 KB copy per
        # window cost 19 us each in page faults: 0.6 ms of a 5.6 ms call at 32 windows); the per-window results are views of them
        in_pts = [np.asarray(w["points"], np.float64).reshape(-1, 3) for w in windows]
        in_poses = [np.ascontiguousarray(w["poses_cw"], np.float64).reshape(-1, 7) for w in windows]
        all_pts = np.concatenate(in_pts) if in_pts else np.zeros((0, 3))
        all_out = np.zeros((sum(max(len(p), 1) for p in in_poses), 7))
        p_pts, p_out, o_pts, o_out = all_pts.ctypes.data, all_out.ctypes.data, 0, 0
        for i, w in enumerate(windows):
            poses_cw = in_poses[i]
            fixed_cw = np.ascontiguousarray(w["fixed_cw"], np.float64).reshape(-1, 7)
            obs = np.ascontiguousarray(w["obs"], BA_OBS)
            M, Ko = len(in_pts[i]), max(len(poses_cw), 1)
            pts = all_pts[o_pts:o_pts + M]; out_wc = all_out[o_out:o_out + Ko]
            keep.append((poses_cw, fixed_cw, pts, obs, out_wc))
            a = arr[i]
            a.K, a.F, a.M, a.N = len(poses_cw), len(fixed_cw), M, len(obs)
            a.poses_cw = poses_cw.ctypes.data; a.fixed_poses_cw = fixed_cw.ctypes.data; a.points = p_pts + 24 * o_pts
            a.obs = obs.ctypes.data; a.poses_wc_out = p_out + 56 * o_out
            o_pts += M; o_out += Ko
        cb = SHOULD_STOP_FN((lambda user: 1 if should_stop() else 0)) if should_stop else C.cast(None, SHOULD_STOP_FN)
        cam = camera._c(); c = cfg._c()
        self._check(self._L.orbx_ba_solve_visual_batch(self._h, C.byref(cam), C.byref(c), C.c_int(len(windows)), arr, cb, None))
        res = []
        for i, (poses_cw, fixed_cw, pts, obs, out_wc) in enumerate(keep):
            a = arr[i]
            if a.status == ORBX_ERR_EMPTY:
                res.append(None)
                continue
            res.append(dict(poses_wc=out_wc[:len(poses_cw)], points=pts, iterations=a.iterations, initial_error=a.initial_error,
                            final_error=a.final_error))
        return res

    def prepare_ba_batch(self, windows, obs32=False):
        """A BaBatch: the windows laid out once the way orbx_ba_solve_visual_batch takes them (see BaBatch)."""
        return BaBatch(self, windows, obs32=obs32)

    def debug_ba_blocks(self, camera, cfg, poses_cw, fixed_cw, points, obs, global_mode=False):
        """orbx_debug_ba_blocks: (residual [N,2], A [N,2,6], B [N,2,3]) of every observation, from the solver's device functions."""
        poses_cw = np.ascontiguousarray(poses_cw, np.float64).reshape(-1, 7)
        fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, BA_OBS)
        out = np.zeros((max(len(obs), 1), 20))
        cam = camera._c(); c = cfg._c()
        self._check(self._L.orbx_debug_ba_blocks(self._h, C.byref(cam), C.byref(c), C.c_int(len(poses_cw)), _vp(poses_cw), C.c_int(len(fixed_cw)),
                                                 _vp(fixed_cw), C.c_int(len(pts)), _vp(pts), C.c_int(len(obs)), _vp(obs),
                                                 C.c_int(1 if global_mode else 0), _vp(out)))
        out = out[:len(obs)]
        return out[:, 0:2].copy(), out[:, 2:14].reshape(-1, 2, 6).copy(), out[:, 14:20].reshape(-1, 2, 3).copy()

    def debug_imu_residual(self, poses_wc, velocities, edge_kf, preint):
        """orbx_debug_imu_residual: compute_imu_residual (imu_factors.rs:66-103) of every edge, [E, 9], from the solver's device function."""
        poses_wc = np.ascontiguousarray(poses_wc, np.float64).reshape(-1, 7)
        vel = np.ascontiguousarray(velocities, np.float64).reshape(-1, 3)
        ek = np.ascontiguousarray(edge_kf, np.int32).reshape(-1, 2); pre = np.ascontiguousarray(preint, np.float64).reshape(-1, 11)
        if len(vel) != len(poses_wc) or len(pre) != len(ek):
            raise ValueError("debug_imu_residual: one velocity per pose and one preintegration per edge")
        out = np.zeros((max(len(ek), 1), 9))
        self._check(self._L.orbx_debug_imu_residual(self._h, C.c_int(len(poses_wc)), _vp(poses_wc), _vp(vel), C.c_int(len(ek)), _vp(ek), _vp(pre), _vp(out)))
        return out[:len(ek)]

    def ba_solve_inertial(self, camera, cfg, poses_wc, velocities, biases, fixed_cw, points, obs, edge_kf, preint, should_stop=None):
        """solve_inertial_ba (local_inertial_ba.rs:1074-1275) on flat arrays; every keyframe of the window is returned."""
        poses_wc = np.ascontiguousarray(poses_wc, np.float64).reshape(-1, 7)
        vel = np.ascontiguousarray(velocities, np.float64).reshape(-1, 3); bias = np.ascontiguousarray(biases, np.float64).reshape(-1, 6)
        fixed_cw = np.ascontiguousarray(fixed_cw, np.float64).reshape(-1, 7)
        pts = np.array(points, np.float64, copy=True).reshape(-1, 3)
        obs = np.ascontiguousarray(obs, BA_OBS)
        ek = np.ascontiguousarray(edge_kf, np.int32).reshape(-1, 2); pre = np.ascontiguousarray(preint, np.float64).reshape(-1, 11)
        K, F, M, N, E = len(poses_wc), len(fixed_cw), len(pts), len(obs), len(ek)
        if len(vel) != K or len(bias) != K or len(pre) != E:
            raise ValueError("ba_solve_inertial: inconsistent array lengths")
        out_p = np.zeros((max(K, 1), 7)); out_v = np.zeros((max(K, 1), 3)); out_b = np.zeros((max(K, 1), 6))
        it = C.c_int(); e0 = C.c_double(); e1 = C.c_double()
        cb = SHOULD_STOP_FN((lambda user: 1 if should_stop() else 0)) if should_stop else C.cast(None, SHOULD_STOP_FN)
        cam = camera._c(); c = cfg._c()
        rc = self._L.orbx_ba_solve_inertial(self._h, C.byref(cam), C.byref(c), C.c_int(K), _vp(poses_wc), _vp(vel), _vp(bias), C.c_int(F),
                                            _vp(fixed_cw), C.c_int(M), _vp(pts), C.c_int(N), _vp(obs), C.c_int(E), _vp(ek), _vp(pre), cb,
                                            None, _vp(out_p), _vp(out_v), _vp(out_b), C.byref(it), C.byref(e0), C.byref(e1))
        if rc in (ORBX_ERR_EMPTY,):
            return None                      # reference returns None, local_inertial_ba.rs:1080-1082
        self._check(rc)
        return dict(poses_wc=out_p[:K], velocities=out_v[:K], biases=out_b[:K], points=pts, iterations=it.value,
                    initial_error=e0.value, final_error=e1.value)

    def ba_solve_global(self, camera, cfg, poses_cw, fixed_pose_cw, points, obs, should_stop=None):
        """solve_global_ba (global_ba.rs:184-418): every keyframe but the first is optimised."""
        poses_cw = np.ascontiguousarray(poses_cw, np.float64).reshape(-1, 7)
        fixed = np.ascontiguousarray(fixed_pose_cw, np.float64).reshape(7)
        pts = np.array(points, np.float64, copy=True).reshape(-1, 3)
        o32 = getattr(obs, "dtype", None) == BA_OBS32       # (as ba_solve_visual)
        obs = np.ascontiguousarray(obs, BA_OBS32 if o32 else BA_OBS)
        K, M, N = len(poses_cw), len(pts), len(obs)
        out_wc = np.zeros((max(K, 1), 7))
        it = C.c_int(); e0 = C.c_double(); e1 = C.c_double()
        cb = SHOULD_STOP_FN((lambda user: 1 if should_stop() else 0)) if should_stop else C.cast(None, SHOULD_STOP_FN)
        cam = camera._c(); c = cfg._c()
        fn = self._L.orbx_ba_solve_global_obs32 if o32 else self._L.orbx_ba_solve_global
        rc = fn(self._h, C.byref(cam), C.byref(c), C.c_int(K), _vp(poses_cw), _vp(fixed), C.c_int(M),
                _vp(pts), C.c_int(N), _vp(obs), cb, None, _vp(out_wc), C.byref(it), C.byref(e0),
                C.byref(e1))
        if rc in (ORBX_ERR_EMPTY,):
            return None                      # reference returns None, global_ba.rs:194-196
        self._check(rc)
        return dict(poses_wc=out_wc[:K], points=pts, iterations=it.value, initial_error=e0.value, final_error=e1.value)


class StereoProcessor:
    """stereo.rs:31-66.  `new` = the constructor; `process` takes two grayscale u8 images."""

    def __init__(self, camera: CameraModel, n_features: int, device: int = 0, max_w: int = 1920,
                 max_h: int = 1080):
        self.camera = camera
        self.handle = Handle(camera, n_features, device=device, max_w=max_w, max_h=max_h, max_batch=1)

    @classmethod
    def new(cls, camera: CameraModel, n_features: int, **kw):
        return cls(camera, n_features, **kw)

    def process(self, left, right, timestamp_ns: int) -> StereoFrame:
        kpL, dL, kpR, dR, m, pts, has = self.handle.process_stereo(left, right)
        return StereoFrame(FeatureSet(kpL, dL), FeatureSet(kpR, dR), m, pts, has, int(timestamp_ns))


_default_handle = None


def _handle():
    global _default_handle
    if _default_handle is None:
        from .synth import EUROC_CAMERA
        _default_handle = Handle(CameraModel(**EUROC_CAMERA), 2000)
    return _default_handle


def descriptor_distance(desc1, desc2) -> int:
    """stereo.rs:166-175 for one pair of 32-byte rows (computed on the GPU)."""
    return int(_handle().hamming_batch(np.asarray(desc1, np.uint8).reshape(1, 32),
                                       np.asarray(desc2, np.uint8).reshape(1, 32))[0])


def bf_match_crosscheck(query_descriptors, train_descriptors):
    """tracker.rs:1001-1010: BFMatcher::new(NORM_HAMMING, true).train_match(query, train)."""
    return _handle().hamming_match_crosscheck(query_descriptors, train_descriptors)


@dataclass
class VisualObservation:
    """local_ba_lm.rs:68-78"""
    kf_id: int
    mp_id: int
    observed_uv: tuple
    is_kf_optimized: bool


@dataclass
class VisualBAProblemData:
    """local_ba_lm.rs:48-65.  Poses are 7-vectors (qw,qx,qy,qz,tx,ty,tz), T_cw."""
    local_kf_poses: Dict[int, np.ndarray]
    local_mp_positions: Dict[int, np.ndarray]
    fixed_kf_poses: Dict[int, np.ndarray]
    anchor_kf_id: int
    observations: List[VisualObservation]
    optimized_kf_ids: List[int]
    mp_ids: List[int]


@dataclass
class VisualBAResultData:
    """local_ba_lm.rs:81-93.  optimized_poses are T_wc."""
    optimized_poses: Dict[int, np.ndarray] = field(default_factory=dict)
    optimized_points: Dict[int, np.ndarray] = field(default_factory=dict)
    iterations: int = 0
    initial_error: float = 0.0
    final_error: float = 0.0


def flatten_ba_problem(problem: VisualBAProblemData):
    """The id -> index re-keying of local_ba_lm.rs:928-987 (what the Rust shim does before the FFI
    call): observations whose map point is unknown are dropped (:947), an optimised-flagged
    observation whose keyframe is not in optimized_kf_ids, or a fixed one whose id is not in
    fixed_kf_poses, falls back to the identity pose (:569)."""
    kf_idx = {k: i for i, k in enumerate(problem.optimized_kf_ids)}
    mp_idx = {m: i for i, m in enumerate(problem.mp_ids)}
    fixed_ids = list(problem.fixed_kf_poses.keys())
    fixed_idx = {k: i for i, k in enumerate(fixed_ids)}
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0])
    # :966-977 leaves the parameters of a keyframe without a pose at zero = identity
    poses = np.array([problem.local_kf_poses.get(k, ident) for k in problem.optimized_kf_ids],
                     np.float64).reshape(-1, 7)
    fixed = np.array([problem.fixed_kf_poses[k] for k in fixed_ids], np.float64).reshape(-1, 7)
    pts = np.array([problem.local_mp_positions.get(m, np.zeros(3)) for m in problem.mp_ids],
                   np.float64).reshape(-1, 3)
    rows = []
    for o in problem.observations:
        if o.mp_id not in mp_idx:
            continue
        if o.is_kf_optimized and o.kf_id in kf_idx:
            rows.append((kf_idx[o.kf_id], -1, mp_idx[o.mp_id], 0, o.observed_uv[0], o.observed_uv[1]))
        else:
            rows.append((-1, fixed_idx.get(o.kf_id, -1), mp_idx[o.mp_id], 0, o.observed_uv[0], o.observed_uv[1]))
    return poses, fixed, pts, np.array(rows, BA_OBS)


def solve_visual_ba(problem: VisualBAProblemData, camera: CameraModel, config: LocalBAConfigLM,
                    should_stop: Callable[[], bool], handle: Handle = None) -> Optional[VisualBAResultData]:
    """local_ba_lm.rs:912-1098."""
    h = handle or _handle()
    poses, fixed, pts, obs = flatten_ba_problem(problem)
    r = h.ba_solve_visual(camera, config, poses, fixed, pts, obs, should_stop)
    if r is None:
        return None
    return VisualBAResultData(
        {k: r["poses_wc"][i] for i, k in enumerate(problem.optimized_kf_ids)},
        {m: r["points"][i] for i, m in enumerate(problem.mp_ids)},
        r["iterations"], r["initial_error"], r["final_error"])


class KeyFrame:
    """orbx_keyframe: NewKeyFrameMsg (system/messages.rs:19-51) with keypoints, descriptors, stereo points and map-point flags
    resident on the GPU.  Built from the device outputs of the extractor; feeds guided_match / search_for_triangulation /
    fuse_search without moving the features over PCIe."""

    def __init__(self, handle: "Handle", d_kp, d_desc, n, d_points_cam=None, d_has_point=None, keyframe_id=0, timestamp_ns=0,
                 pose_wc=(1.0, 0, 0, 0, 0, 0, 0)):
        self._handle = handle
        self._L = handle._L
        L = self._L
        L.orbx_keyframe_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                           C.c_void_p, C.c_void_p]
        L.orbx_keyframe_destroy.argtypes = [C.c_void_p]; L.orbx_keyframe_destroy.restype = None
        for name in ("orbx_keyframe_info", "orbx_keyframe_download"):
            getattr(L, name).argtypes = [C.c_void_p] * 5
        for name in ("orbx_keyframe_set_pose", "orbx_keyframe_set_map_points", "orbx_keyframe_get_map_points"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.orbx_keyframe_guided_match.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                 C.c_int, C.c_void_p, C.c_void_p]
        L.orbx_keyframe_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.orbx_keyframe_fuse_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_uint,
                                                C.c_void_p, C.c_void_p]
        self._p = C.c_void_p()
        pose = np.ascontiguousarray(pose_wc, np.float64).reshape(7)
        handle._after_torch()
        handle._check(L.orbx_keyframe_create(handle._h, _vp(d_kp), _vp(d_desc), C.c_int(int(n)), _vp(d_points_cam), _vp(d_has_point),
                                             C.c_uint64(int(keyframe_id)), C.c_uint64(int(timestamp_ns)), _vp(pose), C.byref(self._p)))
        self.n = int(n)

    @classmethod
    def from_batch_outputs(cls, handle, out, pair, keyframe_id=0, timestamp_ns=0, pose_wc=(1.0, 0, 0, 0, 0, 0, 0)):
        """The LEFT feature set of stereo pair `pair` of alloc_batch_outputs() / process_stereo_batch_device (synchronise first:
        the keypoint count is read on the host; the features themselves never leave the device)."""
        n = int(out["nkp"][pair, 0].item())
        return cls(handle, out["kp"][pair, 0], out["desc"][pair, 0], n, out["points"][pair], out["has_point"][pair], keyframe_id, timestamp_ns, pose_wc)

    def set_pose(self, pose_wc):
        self._handle._check(self._L.orbx_keyframe_set_pose(self._p, _vp(np.ascontiguousarray(pose_wc, np.float64).reshape(7))))

    def set_map_points(self, mp_ids):
        """matched_map_points: iterable of ids or None."""
        a = np.array([-1 if m is None else int(m) for m in mp_ids], np.int64)
        assert len(a) == self.n
        self._handle._check(self._L.orbx_keyframe_set_map_points(self._p, _vp(a)))

    def map_points(self):
        a = np.zeros(max(self.n, 1), np.int64)
        self._handle._check(self._L.orbx_keyframe_get_map_points(self._p, _vp(a)))
        return [None if v < 0 else int(v) for v in a[:self.n]]

    def info(self):
        n = C.c_int(); kid = C.c_uint64(); ts = C.c_uint64(); pose = np.zeros(7)
        self._handle._check(self._L.orbx_keyframe_info(self._p, C.byref(n), C.byref(kid), C.byref(ts), _vp(pose)))
        return dict(n=n.value, keyframe_id=kid.value, timestamp_ns=ts.value, pose_wc=pose)

    def download(self):
        kp = np.zeros(max(self.n, 1), KEYPOINT); desc = np.zeros((max(self.n, 1), 32), np.uint8)
        pts = np.zeros((max(self.n, 1), 3)); has = np.zeros(max(self.n, 1), np.uint8)
        self._handle._check(self._L.orbx_keyframe_download(self._p, _vp(kp), _vp(desc), _vp(pts), _vp(has)))
        return kp[:self.n], desc[:self.n], pts[:self.n], has[:self.n]

    def guided_match(self, img_w, img_h, q_uv, q_desc, radius, mode):
        q_uv = np.ascontiguousarray(q_uv, np.float64).reshape(-1, 2); q_desc = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        nq = len(q_uv)
        idx = np.full(max(nq, 1), -1, np.int32); dist = np.zeros(max(nq, 1), np.uint32)
        self._handle._check(self._L.orbx_keyframe_guided_match(self._handle._h, self._p, float(img_w), float(img_h), _vp(q_uv), _vp(q_desc), nq,
                                                               float(radius), int(mode), _vp(idx), _vp(dist)))
        return idx[:nq], dist[:nq]

    def search_for_triangulation(self, camera, other: "KeyFrame", max_dist=50):
        pairs = np.zeros((max(self.n, 1), 2), np.int32); n = C.c_int()
        cam = camera._c()
        self._handle._check(self._L.orbx_keyframe_search_for_triangulation(self._handle._h, C.byref(cam), self._p, other._p, int(max_dist),
                                                                           _vp(pairs), C.byref(n)))
        return pairs[:n.value].copy()

    @staticmethod
    def fuse_search(handle, camera, positions, mp_desc, keyframes, radius_scale, desc_threshold=50):
        positions = np.ascontiguousarray(positions, np.float64).reshape(-1, 3); mp_desc = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        P, T = len(positions), len(keyframes)
        arr = (C.c_void_p * max(T, 1))(*[k._p for k in keyframes])
        idx = np.full((max(P, 1), max(T, 1)), -1, np.int32); dist = np.zeros((max(P, 1), max(T, 1)), np.uint32)
        cam = camera._c()
        handle._check(handle._L.orbx_keyframe_fuse_search(handle._h, C.byref(cam), _vp(positions), _vp(mp_desc), P, arr, T, float(radius_scale),
                                                          int(desc_threshold), _vp(idx), _vp(dist)))
        return idx[:P, :T], dist[:P, :T]

    def close(self):
        if self._p:
            self._L.orbx_keyframe_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            if self._handle._h:
                self.close()
        except Exception:
            pass


class BaBatch:
    """A batch of local-BA windows kept in the layout orbx_ba_solve_visual_batch takes — what a host that owns its window storage (the Rust
    crate behind INTEGRATION.md's shim) hands over without any marshalling: the orbx_ba_window array filled once, every window's
    observations consecutive slices of ONE page-locked buffer (the copy engine reads them where they lie, one copy per half of the batch),
    poses / fixed poses / points / output poses each one array.  solve() refreshes the in/out points from the initial ones and makes the
    call; Handle.ba_solve_visual_batch(list of dicts) is the ad-hoc form that builds all of this per call."""

    def __init__(self, handle, windows, obs32=False):
        """obs32: carry the observations in the 16-byte form orbx_ba_obs32 (half the upload; bit-identical results).  Needs every pixel
        coordinate to be exactly an f32, as the reference's are; raises ValueError otherwise."""
        import torch
        self._handle = handle
        dt = BA_OBS32 if obs32 else BA_OBS
        if obs32:
            obs = [ba_obs_to_obs32(w["obs"], len(np.asarray(w["fixed_cw"]).reshape(-1, 7))) for w in windows]
            if any(a is None for a in obs):
                raise ValueError("BaBatch(obs32=True): a pixel coordinate is not exactly representable as f32")
        else:
            obs = [np.ascontiguousarray(w["obs"], BA_OBS) for w in windows]
        self.obs32 = bool(obs32)
        self._obs_buf = torch.empty(max(sum(len(a) for a in obs), 1) * dt.itemsize, dtype=torch.uint8).pin_memory()
        self.obs = self._obs_buf.numpy().view(dt)
        self.poses = [np.ascontiguousarray(w["poses_cw"], np.float64).reshape(-1, 7) for w in windows]
        self.fixed = [np.ascontiguousarray(w["fixed_cw"], np.float64).reshape(-1, 7) for w in windows]
        pts = [np.asarray(w["points"], np.float64).reshape(-1, 3) for w in windows]
        self.points0 = np.concatenate(pts) if pts else np.zeros((0, 3))
        self.points = self.points0.copy()
        self.out = np.zeros((sum(max(len(p), 1) for p in self.poses), 7))
        self.arr = (_BaWindow * max(len(windows), 1))()
        self._views = []
        o_obs = o_pts = o_out = 0
        for i, a_obs in enumerate(obs):
            n, M, K = len(a_obs), len(pts[i]), len(self.poses[i])
            self.obs[o_obs:o_obs + n] = a_obs
            a = self.arr[i]
            a.K, a.F, a.M, a.N = K, len(self.fixed[i]), M, n
            a.poses_cw = self.poses[i].ctypes.data; a.fixed_poses_cw = self.fixed[i].ctypes.data
            a.points = self.points.ctypes.data + 24 * o_pts
            if obs32:
                a.obs = None; a.obs32 = self.obs.ctypes.data + BA_OBS32.itemsize * o_obs
            else:
                a.obs = self.obs.ctypes.data + BA_OBS.itemsize * o_obs; a.obs32 = None
            a.poses_wc_out = self.out.ctypes.data + 56 * o_out
            self._views.append((self.out[o_out:o_out + K], self.points[o_pts:o_pts + M]))
            o_obs += n; o_pts += M; o_out += max(K, 1)
        self.n = len(windows)

    def solve(self, camera, cfg, should_stop=None):
        """One orbx_ba_solve_visual_batch call.  Returns one dict per window (None where the reference returns None); `poses_wc` and `points`
        are views of the batch's own arrays, valid until the next solve()."""
        h = self._handle
        np.copyto(self.points, self.points0)
        cb = SHOULD_STOP_FN((lambda user: 1 if should_stop() else 0)) if should_stop else C.cast(None, SHOULD_STOP_FN)
        cam = camera._c(); c = cfg._c()
        h._check(h._L.orbx_ba_solve_visual_batch(h._h, C.byref(cam), C.byref(c), C.c_int(self.n), self.arr, cb, None))
        res = []
        for i in range(self.n):
            a = self.arr[i]
            if a.status == ORBX_ERR_EMPTY:
                res.append(None)
            else:
                res.append(dict(poses_wc=self._views[i][0], points=self._views[i][1], iterations=a.iterations, initial_error=a.initial_error,
                                final_error=a.final_error))
        return res


class MapSnapshot:
    """The map as flat arrays (CSR) — what the host side of local BA reads and writes; the same layout and the same stated
    orders as include/orbx_map.hpp (see there for why the reference's HashMap orders are replaced by stated ones).

      keyframes:  kf_ids u64[nkf], kf_bad u8[nkf], kf_pose_wc f64[nkf,7] (T_wc), kf_n_keypoints i32[nkf],
                  kf_feat_start i32[nkf+1], feat_mp_id i64[nfeat] (-1 = None), feat_uv f32[nfeat,2],
                  cov_start i32[nkf+1], cov_kf_id u64[ncov] (covisibility neighbours in the stated order)
      map points: mp_ids u64[nmp], mp_bad u8[nmp], mp_pos f64[nmp,3], mp_obs_start i32[nmp+1], mp_obs_kf_id u64[nobs]
      inertial branch (optional, default "no IMU"): kf_prev_id i64[nkf] (-1 = None), kf_velocity f64[nkf,3], kf_bias f64[nkf,6] (gyro, accel),
                  kf_has_preint u8[nkf], kf_preint f64[nkf,11] (delta_rot qw,qx,qy,qz | delta_vel | delta_pos | dt, from prev_kf),
                  feat_stereo u8[nfeat] (points_cam[i].is_some()), mp_obs_feat_idx i32[nobs], imu_initialized (Map::is_imu_initialized())
    """
    FIELDS = (("kf_ids", np.uint64), ("kf_bad", np.uint8), ("kf_pose_wc", np.float64), ("kf_n_keypoints", np.int32),
              ("kf_feat_start", np.int32), ("feat_mp_id", np.int64), ("feat_uv", np.float32), ("cov_start", np.int32),
              ("cov_kf_id", np.uint64), ("mp_ids", np.uint64), ("mp_bad", np.uint8), ("mp_pos", np.float64),
              ("mp_obs_start", np.int32), ("mp_obs_kf_id", np.uint64))
    INERTIAL_FIELDS = (("kf_prev_id", np.int64), ("kf_velocity", np.float64), ("kf_bias", np.float64), ("kf_has_preint", np.uint8),
                       ("kf_preint", np.float64), ("feat_stereo", np.uint8), ("mp_obs_feat_idx", np.int32))

    def __init__(self, imu_initialized=False, **arrays):
        for name, dt in self.FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays[name], dt))
        self.kf_pose_wc = self.kf_pose_wc.reshape(-1, 7)
        self.feat_uv = self.feat_uv.reshape(-1, 2)
        self.mp_pos = self.mp_pos.reshape(-1, 3)
        nkf, nfeat, nobs = len(self.kf_ids), len(self.feat_mp_id), len(self.mp_obs_kf_id)
        default = dict(kf_prev_id=np.full(nkf, -1, np.int64), kf_velocity=np.zeros((nkf, 3)), kf_bias=np.zeros((nkf, 6)), kf_has_preint=np.zeros(nkf, np.uint8),
                       kf_preint=np.zeros((nkf, 11)), feat_stereo=np.zeros(nfeat, np.uint8), mp_obs_feat_idx=np.full(nobs, -1, np.int32))
        for name, dt in self.INERTIAL_FIELDS:
            setattr(self, name, np.ascontiguousarray(arrays.get(name, default[name]), dt))
        self.kf_velocity = self.kf_velocity.reshape(-1, 3); self.kf_bias = self.kf_bias.reshape(-1, 6); self.kf_preint = self.kf_preint.reshape(-1, 11)
        self.imu_initialized = bool(imu_initialized)
        self.build_index()

    def build_index(self):
        self._kf = {int(k): i for i, k in enumerate(self.kf_ids)}
        self._mp = {int(k): i for i, k in enumerate(self.mp_ids)}

    def to_bytes(self):
        """[22 x u64: 21 element counts, imu_initialized] then the arrays in FIELDS + INERTIAL_FIELDS order (tests/cpp/local_mapper_driver.cpp reads this)."""
        arrs = [getattr(self, n).reshape(-1) for n, _ in self.FIELDS + self.INERTIAL_FIELDS]
        return np.array([len(a) for a in arrs] + [1 if self.imu_initialized else 0], np.uint64).tobytes() + b"".join(a.tobytes() for a in arrs)

    # ---- local_ba_lm.rs:665-726 --------------------------------------------------------------------------------
    def _local_keyframes(self, current, max_cov):
        local = [int(current)]
        k = self._kf.get(int(current), -1)
        if k >= 0:
            s, e = int(self.cov_start[k]), int(self.cov_start[k + 1])
            for nid in self.cov_kf_id[s:min(e, s + max_cov)]:        # take(max_covisible) before the filter (:675)
                nb = self._kf.get(int(nid), -1)
                if nb >= 0 and not self.kf_bad[nb]:
                    local.append(int(nid))
        return local

    def _local_map_points(self, local):
        seen = {}
        for kid in local:
            k = self._kf.get(kid, -1)
            if k < 0:
                continue
            for mp_id in self.feat_mp_id[int(self.kf_feat_start[k]):int(self.kf_feat_start[k + 1])]:
                if mp_id >= 0:
                    j = self._mp.get(int(mp_id), -1)
                    if j >= 0 and not self.mp_bad[j]:
                        seen.setdefault(int(mp_id), True)
        return list(seen)

    def _fixed_keyframes(self, local, mp_ids):
        loc = set(local)
        seen = {}
        for mid in mp_ids:
            j = self._mp.get(mid, -1)
            if j < 0:
                continue
            for kid in self.mp_obs_kf_id[int(self.mp_obs_start[j]):int(self.mp_obs_start[j + 1])]:
                if int(kid) not in loc:
                    seen.setdefault(int(kid), True)
        return list(seen)

    def collect_visual_ba_data(self, current_kf_id, config: "LocalBAConfigLM" = None) -> Optional["VisualBAProblemData"]:
        """PHASE 1 = collect_visual_ba_data, local_ba_lm.rs:800-897."""
        config = config or LocalBAConfigLM()
        local = self._local_keyframes(current_kf_id, config.max_covisible_keyframes)
        if not local:
            return None
        mp_ids = self._local_map_points(local)
        if not mp_ids:
            return None
        fixed = self._fixed_keyframes(local, mp_ids)
        anchor, optimized = local[0], local[1:]
        pose = lambda kid: se3_inverse(self.kf_pose_wc[self._kf[kid]])
        local_kf_poses = {k: pose(k) for k in optimized if k in self._kf}
        fixed_kf_poses = {}
        if anchor in self._kf:
            fixed_kf_poses[anchor] = pose(anchor)
        for k in fixed:
            if k in self._kf:
                fixed_kf_poses[k] = pose(k)
        local_mp_positions = {m: self.mp_pos[self._mp[m]].copy() for m in mp_ids if m in self._mp}
        opt_set, mp_set = set(optimized), set(mp_ids)
        obs = []
        for kid in local + fixed:
            k = self._kf.get(kid, -1)
            if k < 0:
                continue
            s, e = int(self.kf_feat_start[k]), int(self.kf_feat_start[k + 1])
            nkp = int(self.kf_n_keypoints[k])
            for f in range(s, e):
                mp_id = int(self.feat_mp_id[f])
                if mp_id >= 0 and mp_id in mp_set and f - s < nkp:
                    obs.append(VisualObservation(kid, mp_id, (float(self.feat_uv[f, 0]), float(self.feat_uv[f, 1])), kid in opt_set))
        if not obs:
            return None
        return VisualBAProblemData(local_kf_poses, local_mp_positions, fixed_kf_poses, anchor, obs, optimized, mp_ids)

    # ---- local_inertial_ba.rs:366-429, :933-1072, :1289-1330 ----------------------------------------------------
    def _temporal_keyframes(self, current, window_size):
        ids = [int(current)]
        prev = int(self.kf_prev_id[self._kf[int(current)]]) if int(current) in self._kf else -1
        while len(ids) < window_size and prev >= 0:
            ids.append(prev)
            prev = int(self.kf_prev_id[self._kf[prev]]) if prev in self._kf else -1
        ids.reverse()                                                 # oldest first: the anchor
        return ids

    def _fixed_keyframes_inertial(self, opt, mp_ids):
        o = set(opt)
        seen = {}
        for mid in mp_ids:
            j = self._mp.get(mid, -1)
            if j < 0:
                continue
            for kid in self.mp_obs_kf_id[int(self.mp_obs_start[j]):int(self.mp_obs_start[j + 1])]:
                kid = int(kid)
                if kid not in o:
                    k = self._kf.get(kid, -1)
                    if k >= 0 and not self.kf_bad[k]:                 # (:419-423: checked here, not in the visual collect)
                        seen.setdefault(kid, True)
        return list(seen)

    def collect_inertial_ba_data(self, current_kf_id, config: "LocalInertialBAConfig" = None) -> Optional["InertialBAProblemData"]:
        """PHASE 1 = collect_inertial_ba_data, local_inertial_ba.rs:933-1072."""
        config = config or LocalInertialBAConfig()
        opt = self._temporal_keyframes(current_kf_id, config.window_size)
        if len(opt) < 2:
            return None
        mp_ids = self._local_map_points(opt)
        if not mp_ids:
            return None
        fixed = self._fixed_keyframes_inertial(opt, mp_ids)
        kf_poses, kf_vel, kf_bias = {}, {}, {}
        for kid in opt:
            k = self._kf.get(kid, -1)
            if k >= 0:
                kf_poses[kid] = self.kf_pose_wc[k].copy(); kf_vel[kid] = self.kf_velocity[k].copy(); kf_bias[kid] = self.kf_bias[k].copy()
        fixed_poses = {kid: se3_inverse(self.kf_pose_wc[self._kf[kid]]) for kid in fixed if kid in self._kf}
        if opt[0] in self._kf:
            fixed_poses[opt[0]] = se3_inverse(self.kf_pose_wc[self._kf[opt[0]]])
        mp_positions = {m: self.mp_pos[self._mp[m]].copy() for m in mp_ids if m in self._mp}
        opt_set, all_kf = set(opt), set(opt) | set(fixed)
        obs = []
        for mid in mp_ids:
            j = self._mp.get(mid, -1)
            if j < 0:
                continue
            for o in range(int(self.mp_obs_start[j]), int(self.mp_obs_start[j + 1])):
                kid = int(self.mp_obs_kf_id[o])
                k = self._kf.get(kid, -1)
                if kid not in all_kf or k < 0:
                    continue
                fi = int(self.mp_obs_feat_idx[o])
                s, e = int(self.kf_feat_start[k]), int(self.kf_feat_start[k + 1])
                if fi < 0 or fi >= int(self.kf_n_keypoints[k]) or s + fi >= e:       # keypoints.get(feat_idx) is Err
                    continue
                f = s + fi
                obs.append(InertialVisualObs(kid, mid, (float(self.feat_uv[f, 0]), float(self.feat_uv[f, 1])), bool(self.feat_stereo[f]),
                                             kid in opt_set and kid != opt[0]))
        edges = []
        for i in range(len(opt) - 1):
            kj = self._kf.get(opt[i + 1], -1)
            if kj >= 0 and self.kf_has_preint[kj] and self.kf_preint[kj, 10] > 0.0:
                edges.append(ImuEdgeData(opt[i], opt[i + 1], self.kf_preint[kj].copy()))
        return InertialBAProblemData(kf_poses, kf_vel, kf_bias, mp_positions, fixed_poses, obs, edges, opt, mp_ids)

    def apply_inertial_ba_results(self, result: "InertialBAResultData") -> int:
        """PHASE 3 = apply_inertial_ba_results, local_inertial_ba.rs:1289-1330: poses and points count, velocities and biases do not."""
        updated = 0
        for kid, pose in result.optimized_poses.items():
            k = self._kf.get(int(kid), -1)
            if k >= 0 and not self.kf_bad[k]:
                self.kf_pose_wc[k] = pose
                updated += 1
        for kid, v in result.optimized_velocities.items():
            k = self._kf.get(int(kid), -1)
            if k >= 0 and not self.kf_bad[k]:
                self.kf_velocity[k] = v
        for kid, b in result.optimized_biases.items():
            k = self._kf.get(int(kid), -1)
            if k >= 0 and not self.kf_bad[k]:
                self.kf_bias[k] = b
        for mid, pos in result.optimized_points.items():
            j = self._mp.get(int(mid), -1)
            if j >= 0 and not self.mp_bad[j]:
                self.mp_pos[j] = pos
                updated += 1
        return updated

    # ---- global_ba.rs:100-181, :421-443 ----------------------------------------------------------------------------
    def collect_global_ba_data(self) -> Optional["GlobalBAProblemData"]:
        """PHASE 1 = collect_global_ba_data, global_ba.rs:100-181.  Keyframes: not bad, ids ascending (:121), the first one fixed; map points:
        not bad and seen by ANY collected keyframe (:137), in the snapshot's order (the reference's is its HashMap's); observations keyframe by
        keyframe in ascending id, features in order."""
        kf_ids = sorted(int(k) for k, bad in zip(self.kf_ids, self.kf_bad) if not bad)
        if not kf_ids:
            return None
        kf_poses = {k: se3_inverse(self.kf_pose_wc[self._kf[k]]) for k in kf_ids}
        kf_set = set(kf_ids)
        mp_ids, mp_positions = [], {}
        for j, mid in enumerate(self.mp_ids):
            if self.mp_bad[j]:
                continue
            s, e = int(self.mp_obs_start[j]), int(self.mp_obs_start[j + 1])
            if not any(int(k) in kf_set for k in self.mp_obs_kf_id[s:e]):
                continue
            mp_ids.append(int(mid))
            mp_positions[int(mid)] = self.mp_pos[j].copy()
        if not mp_ids:
            return None
        mp_set = set(mp_ids)
        obs = []
        for kid in kf_ids:
            k = self._kf[kid]
            s, e = int(self.kf_feat_start[k]), int(self.kf_feat_start[k + 1])
            nkp = int(self.kf_n_keypoints[k])
            for f in range(s, e):
                mp_id = int(self.feat_mp_id[f])
                if mp_id >= 0 and mp_id in mp_set and f - s < nkp:
                    obs.append(GlobalBAObservation(kid, mp_id, (float(self.feat_uv[f, 0]), float(self.feat_uv[f, 1]))))
        if not obs:
            return None
        return GlobalBAProblemData(kf_poses, mp_positions, obs, kf_ids, mp_ids, kf_ids[0])

    def apply_global_ba_results(self, result: "GlobalBAResult") -> int:
        """PHASE 3 = apply_global_ba_results, global_ba.rs:421-443: the same silent skips as the local one; the fixed keyframe's pose is in the
        result too and counts."""
        return self.apply_visual_ba_results(result)

    def apply_visual_ba_results(self, result: "VisualBAResultData") -> int:
        """PHASE 3 = apply_visual_ba_results, local_ba_lm.rs:1112-1138: gone or bad entities are skipped silently."""
        updated = 0
        for kid, pose in result.optimized_poses.items():
            k = self._kf.get(int(kid), -1)
            if k >= 0 and not self.kf_bad[k]:
                self.kf_pose_wc[k] = pose
                updated += 1
        for mid, pos in result.optimized_points.items():
            j = self._mp.get(int(mid), -1)
            if j >= 0 and not self.mp_bad[j]:
                self.mp_pos[j] = pos
                updated += 1
        return updated


def local_bundle_adjustment(snapshot: MapSnapshot, kf_id, camera: "CameraModel", should_stop: Callable[[], bool] = None,
                            handle: "Handle" = None):
    """LocalMapper::local_bundle_adjustment (local_mapper.rs:334-410): the inertial branch (:343-375) once the map's IMU is
    initialised, else the visual one (:378-408); each collect -> solve (GPU, no lock) -> apply only if the solve ran an iteration
    (:363, :396).  Returns (updated | None where the reference returns early, result)."""
    if snapshot.imu_initialized:
        iconfig = LocalInertialBAConfig()
        iproblem = snapshot.collect_inertial_ba_data(kf_id, iconfig)
        if iproblem is None:
            return None, None
        iresult = solve_inertial_ba(iproblem, camera, iconfig, should_stop or (lambda: False), handle=handle)
        if iresult is None:
            return None, None
        return (snapshot.apply_inertial_ba_results(iresult) if iresult.iterations > 0 else 0), iresult
    config = LocalBAConfigLM()
    problem = snapshot.collect_visual_ba_data(kf_id, config)
    if problem is None:
        return None, None
    result = solve_visual_ba(problem, camera, config, should_stop or (lambda: False), handle=handle)
    if result is None:
        return None, None
    return (snapshot.apply_visual_ba_results(result) if result.iterations > 0 else 0), result


@dataclass
class GlobalBAObservation:
    """global_ba.rs:70-80"""
    kf_id: int
    mp_id: int
    observed_uv: tuple


@dataclass
class GlobalBAProblemData:
    """global_ba.rs:49-67.  Poses are 7-vectors (qw,qx,qy,qz,tx,ty,tz), T_cw."""
    kf_poses: Dict[int, np.ndarray]
    mp_positions: Dict[int, np.ndarray]
    observations: List[GlobalBAObservation]
    kf_ids: List[int]
    mp_ids: List[int]
    fixed_kf_id: int


@dataclass
class GlobalBAResult:
    """global_ba.rs:83-98.  optimized_poses are T_wc and include the fixed keyframe."""
    optimized_poses: Dict[int, np.ndarray] = field(default_factory=dict)
    optimized_points: Dict[int, np.ndarray] = field(default_factory=dict)
    iterations: int = 0
    initial_error: float = 0.0
    final_error: float = 0.0


def se3_inverse(pose7):
    """se3.rs:56-63 with nalgebra's quaternion-vector product (v + w t + q x t, t = 2 q x v)."""
    p = np.asarray(pose7, np.float64)
    w, x, y, z = p[0], -p[1], -p[2], -p[3]
    v = p[4:]
    t = np.array([2.0 * (y * v[2] - z * v[1]), 2.0 * (z * v[0] - x * v[2]), 2.0 * (x * v[1] - y * v[0])])
    c = np.array([y * t[2] - z * t[1], z * t[0] - x * t[2], x * t[1] - y * t[0]])
    return np.concatenate([[w, x, y, z], -(t * w + c + v)])


def flatten_global_ba_problem(problem: GlobalBAProblemData):
    """The id -> index re-keying of global_ba.rs:198-261: the fixed keyframe is removed from the parameter order,
    a keyframe without a pose starts at the identity, an observation of an unknown keyframe uses the identity pose
    (:649).  Returns (opt_ids, poses_cw, fixed_pose_cw, points, obs) or None where the reference returns None."""
    if len(problem.kf_ids) < 2 or len(problem.mp_ids) == 0 or problem.fixed_kf_id not in problem.kf_ids:
        return None
    fixed_pos = problem.kf_ids.index(problem.fixed_kf_id)
    opt_ids = [k for i, k in enumerate(problem.kf_ids) if i != fixed_pos]
    kf_to_param = {k: i for i, k in enumerate(opt_ids)}
    mp_to_param = {m: i for i, m in enumerate(problem.mp_ids)}
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0])
    poses = np.array([problem.kf_poses.get(k, ident) for k in opt_ids], np.float64).reshape(-1, 7)
    fixed = np.asarray(problem.kf_poses.get(problem.fixed_kf_id, ident), np.float64)
    pts = np.array([problem.mp_positions.get(m, np.zeros(3)) for m in problem.mp_ids], np.float64).reshape(-1, 3)
    rows = []
    for o in problem.observations:
        if o.mp_id not in mp_to_param:
            raise ValueError("solve_global_ba: observation of a map point that is not in mp_ids (never produced by collect_global_ba_data)")
        if o.kf_id == problem.fixed_kf_id:
            rows.append((-1, 0, mp_to_param[o.mp_id], 0, o.observed_uv[0], o.observed_uv[1]))
        elif o.kf_id in kf_to_param:
            rows.append((kf_to_param[o.kf_id], -1, mp_to_param[o.mp_id], 0, o.observed_uv[0], o.observed_uv[1]))
        else:
            rows.append((-1, -1, mp_to_param[o.mp_id], 0, o.observed_uv[0], o.observed_uv[1]))
    return opt_ids, poses, fixed, pts, np.array(rows, BA_OBS)


def solve_global_ba(problem: GlobalBAProblemData, camera: CameraModel, config: GlobalBAConfig,
                    should_stop: Callable[[], bool], handle: Handle = None) -> Optional[GlobalBAResult]:
    """global_ba.rs:184-418."""
    flat = flatten_global_ba_problem(problem)
    if flat is None:
        return None
    opt_ids, poses, fixed, pts, obs = flat
    r = (handle or _handle()).ba_solve_global(camera, config, poses, fixed, pts, obs, should_stop)
    if r is None:
        return None
    out = {problem.fixed_kf_id: se3_inverse(fixed)}
    out.update({k: r["poses_wc"][i] for i, k in enumerate(opt_ids)})
    return GlobalBAResult(out, {m: r["points"][i] for i, m in enumerate(problem.mp_ids)}, r["iterations"], r["initial_error"],
                          r["final_error"])


def run_global_ba(snapshot: MapSnapshot, camera: CameraModel, config: GlobalBAConfig = None, running: Optional[list] = None,
                  handle: Handle = None) -> Optional[GlobalBAResult]:
    """run_global_ba (global_ba.rs:450-500): collect -> solve (should_stop = the running flag cleared by stop_global_ba, loop_closer.rs:285) ->
    apply, unconditionally (unlike local BA, which applies only when an iteration ran).  `running`: a one-element list standing in for the
    reference's AtomicBool — set on entry, cleared on every way out."""
    running = running if running is not None else [False]
    running[0] = True
    try:
        problem = snapshot.collect_global_ba_data()
        if problem is None:
            return None
        result = solve_global_ba(problem, camera, config or GlobalBAConfig(), lambda: not running[0], handle=handle)
        if result is None:
            return None
        snapshot.apply_global_ba_results(result)
        return result
    finally:
        running[0] = False


class OrbVocabulary:
    """vocabulary/mod.rs:83-94: the tree lives in device memory; transform runs on the GPU."""

    def __init__(self, handle: Handle, ptr):
        self._handle = handle
        self._v = ptr
        L = handle._L
        k, l, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.orbx_vocab_info(self._v, C.byref(k), C.byref(l), C.byref(nn), C.byref(nw))
        self.k, self.l, self._n_nodes, self._n_words = k.value, l.value, nn.value, nw.value

    @classmethod
    def load_from_text(cls, path, handle: Handle = None):
        """mod.rs:117-211.  Raises OrbxError where the reference returns VocabularyError."""
        h = handle or _handle()
        v = C.c_void_p()
        h._check(h._L.orbx_vocab_load_text(h._h, str(path).encode(), C.byref(v)))
        return cls(h, v)

    @classmethod
    def from_nodes(cls, parent, is_leaf, desc, weight, k=10, l=6, handle: Handle = None):
        h = handle or _handle()
        parent = np.ascontiguousarray(parent, np.uint32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); weight = np.ascontiguousarray(weight, np.float64)
        v = C.c_void_p()
        h._check(h._L.orbx_vocab_create(h._h, C.c_int(len(parent)), _vp(parent), _vp(is_leaf), _vp(desc), _vp(weight), C.c_int(k),
                                        C.c_int(l), C.byref(v)))
        return cls(h, v)

    def params(self):
        return self.k, self.l

    def num_words(self):
        return self._n_words

    def num_nodes(self):
        return self._n_nodes

    def nodes(self):
        parent = np.zeros(self._n_nodes, np.uint32); leaf = np.zeros(self._n_nodes, np.uint8)
        desc = np.zeros((self._n_nodes, 32), np.uint8); weight = np.zeros(self._n_nodes, np.float64)
        self._handle._L.orbx_vocab_nodes(self._v, _vp(parent), _vp(leaf), _vp(desc), _vp(weight))
        return parent, leaf, desc, weight

    def transform_arrays(self, descriptors, levels_up=4):
        """Per descriptor: (word id, leaf node, FeatureVector key, leaf weight)."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(n, np.uint32); leaf = np.zeros(n, np.uint32); node = np.zeros(n, np.uint32); w = np.zeros(n, np.float64)
        h = self._handle
        h._check(h._L.orbx_bow_transform(h._h, self._v, _vp(d), C.c_int(n), C.c_int(levels_up), _vp(word), _vp(leaf), _vp(node), _vp(w)))
        return word, leaf, node, w

    def vectors_arrays(self, descriptors, levels_up=4):
        """orbx_bow_vectors: the two maps of OrbVocabulary::transform accumulated ON THE DEVICE ->
        (bow_word [nb] ascending, bow_weight [nb] L1-normalised, fv_node [nf] ascending, fv_start [nf+1], fv_index [n]).
        Up to BOW_VECTORS_MAX = 8192 descriptors per call on the device; more: device descent + host accumulation, same result."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        if n > BOW_VECTORS_MAX:
            # orbx_bow_vectors sorts one call's descriptors in one workgroup's LDS (at most 8192 of them: four frames' worth).  Beyond
            # that the tree descent still runs on the GPU (orbx_bow_transform) and the two maps are accumulated here with the device
            # kernel's own orders: weights of a word added in feature order, the L1 norm in ascending word id.
            word, _leaf, node, w = self.transform_arrays(d, levels_up)
            order = np.argsort(word, kind="stable")
            bw, first = np.unique(word[order], return_index=True)
            bv = np.array([_seq_sum(w[order[a:b]]) for a, b in zip(first, list(first[1:]) + [n])], np.float64)
            norm = _seq_sum(bv)
            if norm > 0.0:
                bv = bv / norm
            forder = np.argsort(node, kind="stable")
            fn, ffirst = np.unique(node[forder], return_index=True)
            fs = np.concatenate([ffirst, [n]]).astype(np.int32)
            return bw.astype(np.uint32), bv, fn.astype(np.uint32), fs, forder.astype(np.int32)
        bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); fn = np.zeros(max(n, 1), np.uint32)
        fs = np.zeros(n + 1, np.int32); fi = np.zeros(max(n, 1), np.int32); nb = C.c_int(); nf = C.c_int()
        h = self._handle
        h._L.orbx_bow_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        h._check(h._L.orbx_bow_vectors(h._h, self._v, _vp(d), n, int(levels_up), _vp(bw), _vp(bv), C.byref(nb), _vp(fn), _vp(fs), _vp(fi), C.byref(nf)))
        return bw[:nb.value], bv[:nb.value], fn[:nf.value], fs[:nf.value + 1], fi[:n]

    def transform(self, descriptors, levels_up=4):
        """mod.rs:296-325 -> (BowVector dict word -> weight, FeatureVector dict node -> [feature indices]); descent, accumulation
        and L1 normalisation on the GPU (the norm is summed in ascending word id: the reference sums in HashMap order)."""
        bw, bv, fn, fs, fi = self.vectors_arrays(descriptors, levels_up)
        bow = {int(k): float(v) for k, v in zip(bw, bv)}
        feat = {int(fn[i]): [int(x) for x in fi[fs[i]:fs[i + 1]]] for i in range(len(fn))}
        return bow, feat

    def transform_bow_only(self, descriptors):
        """mod.rs:330-356"""
        return self.transform(descriptors, 0)[0]

    @staticmethod
    def score(v1, v2):
        """mod.rs:357-374: 1 - 0.5 * |v1 - v2|_1 (orbx_bow_score; terms added in ascending word id)"""
        k1 = np.array(sorted(v1), np.uint32); k2 = np.array(sorted(v2), np.uint32)
        w1 = np.array([v1[int(k)] for k in k1], np.float64); w2 = np.array([v2[int(k)] for k in k2], np.float64)
        out = C.c_double()
        L = load_library()
        L.orbx_bow_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        rc = L.orbx_bow_score(_vp(k1), _vp(w1), len(k1), _vp(k2), _vp(w2), len(k2), C.byref(out))
        if rc != 0:
            raise OrbxError(rc, "orbx_bow_score: bad argument")
        return out.value

    def close(self):
        if self._v:
            self._handle._L.orbx_vocab_destroy(self._v)
            self._v = None

    def __del__(self):
        try:
            if self._handle._h:
                self.close()
        except Exception:
            pass


def png_decode_gray8(data: bytes):
    """cv::imread(IMREAD_GRAYSCALE) for the PNG subset EuRoC uses (host code in the library, no GPU)."""
    L = load_library()
    buf = np.frombuffer(data, np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = L.orbx_png_decode_gray8(_vp(buf), C.c_size_t(len(buf)), None, C.c_size_t(0), C.byref(w), C.byref(h))
    if rc != 0:
        raise OrbxError(rc, "not a supported greyscale PNG")
    out = np.empty((h.value, w.value), np.uint8)
    rc = L.orbx_png_decode_gray8(_vp(buf), C.c_size_t(len(buf)), _vp(out), C.c_size_t(w.value), None, None)
    if rc != 0:
        raise OrbxError(rc, "damaged PNG")
    return out


class EurocDataset:
    """io/euroc.rs:54-132, the image side: cam0/cam1 lists, calibration, stereo pairs (host code, no GPU)."""

    def __init__(self, root):
        L = load_library()
        self._L = L
        self._d = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.orbx_euroc_open(str(root).encode(), C.byref(self._d), err, C.c_size_t(512))
        if rc != 0:
            self._d = None
            raise OrbxError(rc, err.value.decode())
        cam = _Camera(); kr = (C.c_double * 4)(); w, h = C.c_int(), C.c_int()
        L.orbx_euroc_calibration(self._d, C.byref(cam), kr, C.byref(w), C.byref(h))
        self.camera = CameraModel(cam.fx, cam.fy, cam.cx, cam.cy, cam.baseline)
        self.k_right = tuple(kr)
        self.width, self.height = w.value, h.value

    @classmethod
    def new(cls, root):
        return cls(root)

    def __len__(self):
        return self._L.orbx_euroc_len(self._d)

    def len(self):
        return len(self)

    def frame_timestamp(self, idx):
        ts = C.c_uint64()
        return ts.value if self._L.orbx_euroc_frame_timestamp(self._d, C.c_int(idx), C.byref(ts)) == 0 else None

    def read_pairs(self, first, count, out=None, threads=8):
        """[count, 2, h, w] u8, decoded by `threads` host threads; `out` may be a pinned buffer of that shape."""
        if out is None:
            out = np.empty((count, 2, self.height, self.width), np.uint8)
        rc = self._L.orbx_euroc_read_pairs(self._d, C.c_int(first), C.c_int(count), _vp(out), C.c_int(threads))
        if rc != 0:
            raise OrbxError(rc, self._L.orbx_euroc_last_error(self._d).decode())
        return out

    def stereo_pair(self, idx):
        """euroc.rs:100-132 -> (left, right, timestamp_ns)"""
        p = self.read_pairs(idx, 1, threads=2)
        return p[0, 0], p[0, 1], self.frame_timestamp(idx)

    def close(self):
        if self._d:
            self._L.orbx_euroc_close(self._d)
            self._d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class InertialVisualObs:
    """local_inertial_ba.rs:64-77"""
    kf_id: int
    mp_id: int
    observed_uv: tuple
    is_stereo: bool
    is_kf_in_window: bool


@dataclass
class ImuEdgeData:
    """local_inertial_ba.rs:79-88; preint = (delta_rot qw,qx,qy,qz, delta_vel, delta_pos, dt) of PreintegratedState"""
    kf_i_id: int
    kf_j_id: int
    preint: np.ndarray


@dataclass
class InertialBAProblemData:
    """local_inertial_ba.rs:40-62.  kf_poses are T_wc, fixed_kf_poses T_cw (7-vectors); biases 6-vectors (gyro, accel)."""
    kf_poses: Dict[int, np.ndarray]
    kf_velocities: Dict[int, np.ndarray]
    kf_biases: Dict[int, np.ndarray]
    mp_positions: Dict[int, np.ndarray]
    fixed_kf_poses: Dict[int, np.ndarray]
    visual_observations: List[InertialVisualObs]
    imu_edges: List[ImuEdgeData]
    opt_kf_ids: List[int]
    mp_ids: List[int]


@dataclass
class InertialBAResultData:
    """local_inertial_ba.rs:90-106: the first keyframe of the window is not reported (:1250)."""
    optimized_poses: Dict[int, np.ndarray] = field(default_factory=dict)
    optimized_velocities: Dict[int, np.ndarray] = field(default_factory=dict)
    optimized_biases: Dict[int, np.ndarray] = field(default_factory=dict)
    optimized_points: Dict[int, np.ndarray] = field(default_factory=dict)
    iterations: int = 0
    initial_error: float = 0.0
    final_error: float = 0.0


def flatten_inertial_ba_problem(problem: InertialBAProblemData):
    """The id -> index re-keying of local_inertial_ba.rs:1084-1185: observations of unknown map points and IMU edges
    with a keyframe outside the window are dropped (:1107, :1129-1130); an in-window flag whose keyframe is not in
    opt_kf_ids, or a fixed keyframe without a pose, falls back to the identity pose (:637); missing states start at 0."""
    kf_idx = {k: i for i, k in enumerate(problem.opt_kf_ids)}
    mp_idx = {m: i for i, m in enumerate(problem.mp_ids)}
    fixed_ids = list(problem.fixed_kf_poses.keys())
    fixed_idx = {k: i for i, k in enumerate(fixed_ids)}
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0])
    poses = np.array([problem.kf_poses.get(k, ident) for k in problem.opt_kf_ids], np.float64).reshape(-1, 7)
    vel = np.array([problem.kf_velocities.get(k, np.zeros(3)) for k in problem.opt_kf_ids], np.float64).reshape(-1, 3)
    bias = np.array([problem.kf_biases.get(k, np.zeros(6)) for k in problem.opt_kf_ids], np.float64).reshape(-1, 6)
    fixed = np.array([problem.fixed_kf_poses[k] for k in fixed_ids], np.float64).reshape(-1, 7)
    pts = np.array([problem.mp_positions.get(m, np.zeros(3)) for m in problem.mp_ids], np.float64).reshape(-1, 3)
    rows = []
    for o in problem.visual_observations:
        if o.mp_id not in mp_idx:
            continue
        if o.is_kf_in_window and o.kf_id in kf_idx:
            rows.append((kf_idx[o.kf_id], -1, mp_idx[o.mp_id], int(o.is_stereo), o.observed_uv[0], o.observed_uv[1]))
        else:
            rows.append((-1, fixed_idx.get(o.kf_id, -1), mp_idx[o.mp_id], int(o.is_stereo), o.observed_uv[0], o.observed_uv[1]))
    edges = [(kf_idx[e.kf_i_id], kf_idx[e.kf_j_id], e.preint) for e in problem.imu_edges if e.kf_i_id in kf_idx and e.kf_j_id in kf_idx]
    ek = np.array([(a, b) for a, b, _ in edges], np.int32).reshape(-1, 2)
    pre = np.array([p for _, _, p in edges], np.float64).reshape(-1, 11)
    return poses, vel, bias, fixed, pts, np.array(rows, BA_OBS), ek, pre


def solve_inertial_ba(problem: InertialBAProblemData, camera: CameraModel, config: LocalInertialBAConfig,
                      should_stop: Callable[[], bool], handle: Handle = None) -> Optional[InertialBAResultData]:
    """local_inertial_ba.rs:1074-1275."""
    if len(problem.opt_kf_ids) < 2:
        return None
    poses, vel, bias, fixed, pts, obs, ek, pre = flatten_inertial_ba_problem(problem)
    r = (handle or _handle()).ba_solve_inertial(camera, config, poses, vel, bias, fixed, pts, obs, ek, pre, should_stop)
    if r is None:
        return None
    ids = problem.opt_kf_ids
    return InertialBAResultData({k: r["poses_wc"][i] for i, k in enumerate(ids) if i > 0}, {k: r["velocities"][i] for i, k in enumerate(ids) if i > 0},
                                {k: r["biases"][i] for i, k in enumerate(ids) if i > 0}, {m: r["points"][i] for i, m in enumerate(problem.mp_ids)},
                                r["iterations"], r["initial_error"], r["final_error"])
