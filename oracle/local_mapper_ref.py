"""CPU restatement (test infrastructure — see oracle.h) of the host side of local BA:

    collect_local_keyframes / collect_local_map_points / collect_fixed_keyframes   local_ba_lm.rs:665-726
    collect_visual_ba_data                                                         local_ba_lm.rs:800-897
    apply_visual_ba_results                                                        local_ba_lm.rs:1112-1138
    LocalMapper::local_bundle_adjustment (visual and inertial branches)            local_mapper.rs:334-410
    collect_temporal_keyframes / collect_map_points / collect_fixed_keyframes      local_inertial_ba.rs:366-429
    collect_inertial_ba_data                                                       local_inertial_ba.rs:933-1072
    apply_inertial_ba_results                                                      local_inertial_ba.rs:1289-1330
    collect_global_ba_data / apply_global_ba_results / run_global_ba               global_ba.rs:100-181, :421-443, :450-500

over a Map made of Python dicts, written line by line after the Rust.  The reference iterates `HashMap`s and `HashSet`s
(covisibility weights :675, the map-point set :704, the fixed-keyframe set :725, observations.keys() :718), whose order is
unspecified and differs from run to run; Python dicts keep insertion order, so here "HashMap order" is
the order in which the test inserted the entries and "HashSet order" is first-insertion order — the deterministic order
the product states for itself (include/orbx_map.hpp).  Only tests import this module.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclass
class KeyFrame:                      # atlas/map/keyframe.rs — the fields the path reads
    pose: np.ndarray                 # T_wc as (qw,qx,qy,qz,tx,ty,tz)
    keypoints: List[Tuple[float, float]]                 # kp.pt() as f32 values
    map_point_ids: List[Optional[int]]
    covisibility_weights: Dict[int, int] = field(default_factory=dict)
    is_bad: bool = False
    # what the inertial branch reads besides (keyframe.rs: prev_kf, velocity, imu_bias, imu_preintegrated, points_cam)
    prev_kf: Optional[int] = None
    velocity: np.ndarray = field(default_factory=lambda: np.zeros(3))
    imu_bias: np.ndarray = field(default_factory=lambda: np.zeros(6))            # gyro, accel
    imu_preintegrated: Optional[np.ndarray] = None                                # (delta_rot qw,qx,qy,qz | delta_vel | delta_pos | dt) from prev_kf
    points_cam: List[bool] = field(default_factory=list)                          # points_cam[i].is_some()


@dataclass
class MapPoint:                      # atlas/map/map_point.rs
    position: np.ndarray
    observations: Dict[int, int] = field(default_factory=dict)   # kf_id -> feature index
    is_bad: bool = False


@dataclass
class Map:
    keyframes: Dict[int, KeyFrame] = field(default_factory=dict)
    map_points: Dict[int, MapPoint] = field(default_factory=dict)
    imu_initialized: bool = False                                                 # Map::is_imu_initialized()


def se3_inverse(p):                   # geometry/se3.rs:56-63 (nalgebra quaternion * vector)
    q = np.array([p[0], -p[1], -p[2], -p[3]])
    v = np.asarray(p[4:7], np.float64)
    t = 2.0 * np.array([q[2] * v[2] - q[3] * v[1], q[3] * v[0] - q[1] * v[2], q[1] * v[1] - q[2] * v[0]])
    c = np.array([q[2] * t[2] - q[3] * t[1], q[3] * t[0] - q[1] * t[2], q[1] * t[1] - q[2] * t[0]])
    r = t * q[0] + c + v
    return np.concatenate([q, -r])


def collect_local_keyframes(m: Map, current_kf_id, max_covisible):        # :665-683
    local_kfs = [current_kf_id]
    kf = m.keyframes.get(current_kf_id)
    if kf is not None:
        for neighbor_id in list(kf.covisibility_weights.keys())[:max_covisible]:     # .iter().take(max_covisible)
            nb = m.keyframes.get(neighbor_id)
            if nb is not None and not nb.is_bad:
                local_kfs.append(neighbor_id)
    return local_kfs


def collect_local_map_points(m: Map, local_kf_ids):                        # :686-704
    mp_set = {}
    for kf_id in local_kf_ids:
        kf = m.keyframes.get(kf_id)
        if kf is None:
            continue
        for mp_id in kf.map_point_ids:
            if mp_id is not None:
                mp = m.map_points.get(mp_id)
                if mp is not None and not mp.is_bad:
                    mp_set.setdefault(mp_id, True)
    return list(mp_set.keys())


def collect_fixed_keyframes(m: Map, local_kf_ids, local_mp_ids):           # :707-726
    local = set(local_kf_ids)
    fixed = {}
    for mp_id in local_mp_ids:
        mp = m.map_points.get(mp_id)
        if mp is not None:
            for kf_id in mp.observations.keys():
                if kf_id not in local:
                    fixed.setdefault(kf_id, True)
    return list(fixed.keys())


def collect_visual_ba_data(m: Map, current_kf_id, max_covisible_keyframes=20):   # :800-897
    local_kf_ids = collect_local_keyframes(m, current_kf_id, max_covisible_keyframes)
    if not local_kf_ids:
        return None
    mp_ids = collect_local_map_points(m, local_kf_ids)
    if not mp_ids:
        return None
    fixed_kf_ids = collect_fixed_keyframes(m, local_kf_ids, mp_ids)
    anchor_kf_id = local_kf_ids[0]                                         # :821
    optimized_kf_ids = local_kf_ids[1:]
    local_kf_poses = {}
    for kf_id in optimized_kf_ids:                                         # :825-830
        kf = m.keyframes.get(kf_id)
        if kf is not None:
            local_kf_poses[kf_id] = se3_inverse(kf.pose)
    fixed_kf_poses = {}
    kf = m.keyframes.get(anchor_kf_id)                                     # :833-836
    if kf is not None:
        fixed_kf_poses[anchor_kf_id] = se3_inverse(kf.pose)
    for kf_id in fixed_kf_ids:                                             # :837-841
        kf = m.keyframes.get(kf_id)
        if kf is not None:
            fixed_kf_poses[kf_id] = se3_inverse(kf.pose)
    local_mp_positions = {}
    for mp_id in mp_ids:                                                   # :844-849
        mp = m.map_points.get(mp_id)
        if mp is not None:
            local_mp_positions[mp_id] = np.array(mp.position, np.float64)
    local_kf_set = set(optimized_kf_ids)
    local_mp_set = set(mp_ids)
    observations = []
    for kf_id in local_kf_ids + fixed_kf_ids:                              # :856-882
        kf = m.keyframes.get(kf_id)
        if kf is None:
            continue
        for feat_idx, mp_id in enumerate(kf.map_point_ids):
            if mp_id is not None and mp_id in local_mp_set:
                if feat_idx < len(kf.keypoints):                           # keypoints.get(feat_idx) is Ok
                    x, y = kf.keypoints[feat_idx]
                    observations.append(dict(kf_id=kf_id, mp_id=mp_id, uv=(float(np.float32(x)), float(np.float32(y))),
                                             is_kf_optimized=kf_id in local_kf_set))
    if not observations:
        return None
    return dict(local_kf_poses=local_kf_poses, local_mp_positions=local_mp_positions, fixed_kf_poses=fixed_kf_poses,
                anchor_kf_id=anchor_kf_id, observations=observations, optimized_kf_ids=optimized_kf_ids, mp_ids=mp_ids)


def apply_visual_ba_results(m: Map, optimized_poses, optimized_points):    # :1112-1138
    updated = 0
    for kf_id, pose in optimized_poses.items():
        kf = m.keyframes.get(kf_id)
        if kf is not None and not kf.is_bad:
            kf.pose = np.array(pose, np.float64)
            updated += 1
    for mp_id, pos in optimized_points.items():
        mp = m.map_points.get(mp_id)
        if mp is not None and not mp.is_bad:
            mp.position = np.array(pos, np.float64)
            updated += 1
    return updated


# ---- the inertial branch ------------------------------------------------------------------------------------------------------
def collect_temporal_keyframes(m: Map, current_kf_id, window_size):          # local_inertial_ba.rs:366-384
    kf_ids = [current_kf_id]
    kf = m.keyframes.get(current_kf_id)
    prev_id = kf.prev_kf if kf is not None else None
    while len(kf_ids) < window_size and prev_id is not None:
        kf_ids.append(prev_id)
        kf = m.keyframes.get(prev_id)
        prev_id = kf.prev_kf if kf is not None else None
    kf_ids.reverse()                                                        # oldest first: the anchor
    return kf_ids


def collect_map_points_inertial(m: Map, kf_ids):                             # :387-403
    return collect_local_map_points(m, kf_ids)                              # (the same walk; HashSet order = first seen)


def collect_fixed_keyframes_inertial(m: Map, opt_kf_ids, mp_ids):            # :406-429: unlike the visual one, existence and is_bad are checked
    opt = set(opt_kf_ids)
    fixed = {}
    for mp_id in mp_ids:
        mp = m.map_points.get(mp_id)
        if mp is not None:
            for obs_kf_id in mp.observations.keys():
                if obs_kf_id not in opt:
                    kf = m.keyframes.get(obs_kf_id)
                    if kf is not None and not kf.is_bad:
                        fixed.setdefault(obs_kf_id, True)
    return list(fixed.keys())


def collect_inertial_ba_data(m: Map, current_kf_id, window_size=10):         # :933-1072
    opt_kf_ids = collect_temporal_keyframes(m, current_kf_id, window_size)
    if len(opt_kf_ids) < 2:                                                 # :940-942
        return None
    mp_ids = collect_map_points_inertial(m, opt_kf_ids)
    if not mp_ids:                                                          # :946-948
        return None
    fixed_kf_ids = collect_fixed_keyframes_inertial(m, opt_kf_ids, mp_ids)
    opt_kf_set = set(opt_kf_ids)
    kf_poses, kf_velocities, kf_biases = {}, {}, {}
    for kf_id in opt_kf_ids:                                                # :955-964
        kf = m.keyframes.get(kf_id)
        if kf is not None:
            kf_poses[kf_id] = np.array(kf.pose, np.float64)
            kf_velocities[kf_id] = np.array(kf.velocity, np.float64)
            kf_biases[kf_id] = np.array(kf.imu_bias, np.float64)
    fixed_kf_poses = {}
    for kf_id in fixed_kf_ids:                                              # :967-972
        kf = m.keyframes.get(kf_id)
        if kf is not None:
            fixed_kf_poses[kf_id] = se3_inverse(kf.pose)
    kf = m.keyframes.get(opt_kf_ids[0])                                     # :974-978 the anchor's pose too
    if kf is not None:
        fixed_kf_poses[opt_kf_ids[0]] = se3_inverse(kf.pose)
    mp_positions = {}
    for mp_id in mp_ids:                                                    # :981-986
        mp = m.map_points.get(mp_id)
        if mp is not None:
            mp_positions[mp_id] = np.array(mp.position, np.float64)
    all_kf_ids = set(opt_kf_ids) | set(fixed_kf_ids)
    visual_observations = []
    for mp_id in mp_ids:                                                    # :996-1029: by map point, its observers in map order
        mp = m.map_points.get(mp_id)
        if mp is None:
            continue
        for obs_kf_id, feat_idx in mp.observations.items():
            if obs_kf_id not in all_kf_ids:
                continue
            kf = m.keyframes.get(obs_kf_id)
            if kf is None or not (0 <= feat_idx < len(kf.keypoints)):       # keypoints.get(feat_idx) is Err
                continue
            x, y = kf.keypoints[feat_idx]
            is_stereo = bool(kf.points_cam[feat_idx]) if feat_idx < len(kf.points_cam) else False    # points_cam.get(i).map_or(false, ..)
            is_kf_in_window = obs_kf_id in opt_kf_set and opt_kf_ids[0] != obs_kf_id
            visual_observations.append(dict(kf_id=obs_kf_id, mp_id=mp_id, uv=(float(np.float32(x)), float(np.float32(y))),
                                            is_stereo=is_stereo, is_kf_in_window=is_kf_in_window))
    imu_edges = []
    for i in range(max(len(opt_kf_ids) - 1, 0)):                            # :1032-1049
        kf_j = m.keyframes.get(opt_kf_ids[i + 1])
        if kf_j is not None and kf_j.imu_preintegrated is not None and kf_j.imu_preintegrated[10] > 0.0:
            imu_edges.append(dict(kf_i_id=opt_kf_ids[i], kf_j_id=opt_kf_ids[i + 1], preint=np.array(kf_j.imu_preintegrated, np.float64)))
    return dict(kf_poses=kf_poses, kf_velocities=kf_velocities, kf_biases=kf_biases, mp_positions=mp_positions, fixed_kf_poses=fixed_kf_poses,
                visual_observations=visual_observations, imu_edges=imu_edges, opt_kf_ids=opt_kf_ids, mp_ids=mp_ids)


def apply_inertial_ba_results(m: Map, optimized_poses, optimized_velocities, optimized_biases, optimized_points):   # :1289-1330
    updated = 0
    for kf_id, pose in optimized_poses.items():
        kf = m.keyframes.get(kf_id)
        if kf is not None and not kf.is_bad:
            kf.pose = np.array(pose, np.float64)
            updated += 1
    for kf_id, vel in optimized_velocities.items():                         # (velocities and biases are written, not counted)
        kf = m.keyframes.get(kf_id)
        if kf is not None and not kf.is_bad:
            kf.velocity = np.array(vel, np.float64)
    for kf_id, bias in optimized_biases.items():
        kf = m.keyframes.get(kf_id)
        if kf is not None and not kf.is_bad:
            kf.imu_bias = np.array(bias, np.float64)
    for mp_id, pos in optimized_points.items():
        mp = m.map_points.get(mp_id)
        if mp is not None and not mp.is_bad:
            mp.position = np.array(pos, np.float64)
            updated += 1
    return updated


def local_bundle_adjustment(m: Map, kf_id, solve, max_covisible_keyframes=20, solve_inertial=None, window_size=10):   # local_mapper.rs:334-410
    """solve(problem) -> None | dict(optimized_poses, optimized_points, iterations, ...); solve_inertial(problem) likewise with
    optimized_velocities / optimized_biases.  Returns #updated or None.  The branch is the map's is_imu_initialized() (:338-341)."""
    if m.imu_initialized:                                                   # :343-375
        problem = collect_inertial_ba_data(m, kf_id, window_size)
        if problem is None:
            return None
        result = solve_inertial(problem)
        if result is None:
            return None
        if result["iterations"] > 0:                                        # :361
            return apply_inertial_ba_results(m, result["optimized_poses"], result["optimized_velocities"], result["optimized_biases"],
                                             result["optimized_points"])
        return 0
    problem = collect_visual_ba_data(m, kf_id, max_covisible_keyframes)
    if problem is None:
        return None
    result = solve(problem)
    if result is None:
        return None
    if result["iterations"] > 0:                                           # :396
        return apply_visual_ba_results(m, result["optimized_poses"], result["optimized_points"])
    return 0


# ---- global BA: the host phases around solve_global_ba (src/optimizer/global_ba.rs) ---------------------------------------------
def collect_global_ba_data(m: Map):                                             # :100-181
    kf_ids = []
    kf_poses = {}
    for kf_id, kf in m.keyframes.items():                                       # :108-114
        if kf.is_bad:
            continue
        kf_ids.append(kf_id)
        kf_poses[kf_id] = se3_inverse(kf.pose)                                  # T_cw
    if not kf_ids:
        return None
    kf_ids.sort()                                                               # :121
    fixed_kf_id = kf_ids[0]                                                     # :124
    kf_set = set(kf_ids)
    mp_ids = []
    mp_positions = {}
    for mp_id, mp in m.map_points.items():                                      # :129-144 (HashMap order = insertion order here)
        if mp.is_bad:
            continue
        if not any(k in kf_set for k in mp.observations.keys()):                # :137
            continue
        mp_ids.append(mp_id)
        mp_positions[mp_id] = mp.position
    if not mp_ids:
        return None
    mp_set = set(mp_ids)
    observations = []
    for kf_id in kf_ids:                                                        # :153-170
        kf = m.keyframes.get(kf_id)
        if kf is None:
            continue
        for feat_idx, mp_id in enumerate(kf.map_point_ids):
            if mp_id is not None and mp_id in mp_set and feat_idx < len(kf.keypoints):
                observations.append(dict(kf_id=kf_id, mp_id=mp_id, uv=(float(kf.keypoints[feat_idx][0]), float(kf.keypoints[feat_idx][1]))))
    if not observations:
        return None
    return dict(kf_poses=kf_poses, mp_positions=mp_positions, observations=observations, kf_ids=kf_ids, mp_ids=mp_ids, fixed_kf_id=fixed_kf_id)


def apply_global_ba_results(m: Map, optimized_poses, optimized_points):          # :421-443
    updated = 0
    for kf_id, pose in optimized_poses.items():
        kf = m.keyframes.get(kf_id)
        if kf is not None and not kf.is_bad:
            kf.pose = np.asarray(pose, np.float64).copy()
            updated += 1
    for mp_id, pos in optimized_points.items():
        mp = m.map_points.get(mp_id)
        if mp is not None and not mp.is_bad:
            mp.position = np.asarray(pos, np.float64).copy()
            updated += 1
    return updated


def run_global_ba(m: Map, solve, running):                                       # :450-500; `running` = one-element list (the AtomicBool)
    running[0] = True
    problem = collect_global_ba_data(m)
    if problem is None:
        running[0] = False
        return None
    result = solve(problem, lambda: not running[0])                              # should_stop = the flag cleared
    if result is None:
        running[0] = False
        return None
    apply_global_ba_results(m, result["optimized_poses"], result["optimized_points"])   # unconditionally (:486-489)
    running[0] = False
    return result
