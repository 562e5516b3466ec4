"""Local inertial bundle adjustment (reference src/optimizer/local_inertial_ba.rs:1074-1275, imu_factors.rs:66-103;
SURVEY.md §8f row 2).  CPU: the oracle against closed-form values and its own invariants.  GPU: orbx_ba_solve_inertial vs
the oracle within 1e-6 relative (f64 throughout, the tolerance of the visual solver)."""
import numpy as np
import pytest

import orb_slam3_rust_amd as P
from oracle import oracle as O

TOL = 1e-6


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


def _oracle(w, cfg=None, **kw):
    return O.inertial_ba_solve(O.Camera(**w["camera"]), cfg or O.inertial_ba_config(), w["poses_wc"], w["velocities"], w["biases"],
                               w["fixed_cw"], w["points"], w["obs"], w["edge_kf"], w["preint"], **kw)


def _scaled_axis(q):
    v = q[1:] * (1 if q[0] >= 0 else -1)
    n = np.linalg.norm(v)
    return v / n * 2 * np.arctan2(n, abs(q[0])) if n > 0 else np.zeros(3)


def test_imu_residual_is_zero_on_consistent_states_and_matches_closed_form():
    w = P.synth.inertial_window(1, 4, 30, P.BA_OBS)
    g = np.array([0, 0, -9.81])
    for e, (i, j) in enumerate(w["edge_kf"]):
        si = np.concatenate([_scaled_axis(w["gt_poses_wc"][i, :4]), w["gt_poses_wc"][i, 4:], w["gt_velocities"][i]])
        sj = np.concatenate([_scaled_axis(w["gt_poses_wc"][j, :4]), w["gt_poses_wc"][j, 4:], w["gt_velocities"][j]])
        r = O.inertial_imu_residual(si, sj, w["preint"][e])
        # the synthetic deltas are the exact ones plus noise of (2e-3 rad, 5e-3 m/s, 2e-3 m)
        assert np.abs(r[:3]).max() < 2e-2 and np.abs(r[3:6]).max() < 3e-2 and np.abs(r[6:]).max() < 1.5e-2
        exact = w["preint"][e].copy()
        qi = w["gt_poses_wc"][i, :4] * np.array([1, -1, -1, -1.0])
        dt = exact[10]
        exact[4:7] = P.synth._quat_rot(qi, w["gt_velocities"][j] - w["gt_velocities"][i] - g * dt)
        exact[7:10] = P.synth._quat_rot(qi, w["gt_poses_wc"][j, 4:] - w["gt_poses_wc"][i, 4:] - w["gt_velocities"][i] * dt - 0.5 * g * dt * dt)
        r2 = O.inertial_imu_residual(si, sj, exact)
        assert np.abs(r2[3:]).max() < 1e-12                      # imu_factors.rs:89-99 in closed form


def test_oracle_converges_and_respects_the_loop_rules():
    w = P.synth.inertial_window(2, 5, 150, P.BA_OBS)
    r = _oracle(w)
    assert r["final_error"] < 0.6 * r["initial_error"] and 1 <= r["iterations"] <= 10
    # (no claim about the distance to the ground truth: the reference pairs T_wc parameters with the T_cw form of the
    # pose Jacobian, :774-803, so its steps are not Gauss-Newton steps of this cost; the restatement keeps that)
    # accepted steps never increase |r|^2; lambda bookkeeping is the reference's (:1233-1238)
    tr = r["trace"]
    assert all(tr[i + 1, 0] <= tr[i, 0] for i in range(len(tr) - 1))
    assert _oracle(w, stop_after=3)["iterations"] == 3
    one = dict(w); one["poses_wc"] = w["poses_wc"][:1]; one["velocities"] = w["velocities"][:1]; one["biases"] = w["biases"][:1]
    one["edge_kf"] = w["edge_kf"][:0]; one["preint"] = w["preint"][:0]; one["obs"] = w["obs"][w["obs"]["kf_idx"] <= 0]
    assert _oracle(one) is None                                   # fewer than two keyframes (:1080-1082)
    # the stereo flag selects the Huber threshold (:648): forcing every observation to mono changes the result
    m = dict(w); m["obs"] = w["obs"].copy(); m["obs"]["_pad"] = 0
    assert abs(_oracle(m)["final_error"] - r["final_error"]) > 1e-9


# ---- GPU ------------------------------------------------------------------------------------------------------------
def _gpu(gpu_handle, w, cfg=None, **kw):
    return gpu_handle.ba_solve_inertial(P.CameraModel(**w["camera"]), cfg or P.LocalInertialBAConfig(), w["poses_wc"], w["velocities"],
                                        w["biases"], w["fixed_cw"], w["points"], w["obs"], w["edge_kf"], w["preint"], **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,K,M,fixed", [(1, 3, 60, 1), (2, 5, 150, 2), (3, 10, 400, 2), (4, 8, 300, 0), (6, 11, 250, 1)])
def test_gpu_inertial_ba_matches_oracle(gpu_handle, seed, K, M, fixed):
    """(K = 11: 15 K = 165 unknowns, the largest system of the tiled LDS solve — 66 tiles = 135 168 B of dynamic LDS beside the kernel's static arrays)"""
    w = P.synth.inertial_window(seed, K, M, P.BA_OBS, n_fixed=fixed)
    o = _oracle(w)
    g = _gpu(gpu_handle, w)
    assert g["iterations"] == o["iterations"]
    assert abs(g["initial_error"] - o["initial_error"]) < 1e-10 * o["initial_error"]
    assert abs(g["final_error"] - o["final_error"]) < 1e-7 * o["final_error"]
    for key in ("poses_wc", "velocities", "biases", "points"):
        assert _rel(g[key], o[key]) < TOL, key
    g2 = _gpu(gpu_handle, w)                                      # fixed-order reductions: bitwise repeatable
    assert all(np.array_equal(g[k], g2[k]) for k in ("poses_wc", "velocities", "biases", "points"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,K,M", [(11, 13, 300), (12, 22, 400), (13, 21, 300)])
def test_gpu_inertial_ba_beyond_the_lds_tiles(gpu_handle, seed, K, M):
    """15 K = 195 unknowns: the one-launch factorisation in global memory; 15 K = 315: the same with an ODD row stride, two rows per
    thread below the first panels and a short (11-column) last panel; 15 K = 330: one launch per panel (inertial windows of the
    reference hold 10 keyframes; these sizes are the same code paths a 50-keyframe visual window takes)."""
    w = P.synth.inertial_window(seed, K, M, P.BA_OBS, n_fixed=1)
    o = _oracle(w)
    g = _gpu(gpu_handle, w)
    assert g["iterations"] == o["iterations"]
    assert abs(g["final_error"] - o["final_error"]) < 1e-7 * o["final_error"]
    for key in ("poses_wc", "velocities", "biases", "points"):
        assert _rel(g[key], o[key]) < TOL, key


@pytest.mark.gpu
def test_gpu_inertial_ba_config_abort_and_none(gpu_handle):
    w = P.synth.inertial_window(5, 5, 120, P.BA_OBS)
    cfg = P.LocalInertialBAConfig(max_iterations=6, initial_lambda=1e-1, gyro_rw_info=1e5, accel_rw_info=1e3, huber_threshold_mono=1.5)
    ocfg = O.InertialBaConfig(6, 10, 1.5, float(np.sqrt(7.815)), 1e-1, 1e5, 1e3)
    o = _oracle(w, ocfg); g = _gpu(gpu_handle, w, cfg)
    assert g["iterations"] == o["iterations"] and _rel(g["poses_wc"], o["poses_wc"]) < TOL and _rel(g["points"], o["points"]) < TOL
    calls = []
    r = _gpu(gpu_handle, w, should_stop=lambda: (calls.append(1) or len(calls) > 2))
    o2 = _oracle(w, stop_after=2)
    assert r["iterations"] == 2 and _rel(r["poses_wc"], o2["poses_wc"]) < TOL and _rel(r["velocities"], o2["velocities"]) < TOL
    one = dict(w); one["poses_wc"] = w["poses_wc"][:1]; one["velocities"] = w["velocities"][:1]; one["biases"] = w["biases"][:1]
    one["edge_kf"] = w["edge_kf"][:0]; one["preint"] = w["preint"][:0]; one["obs"] = w["obs"][w["obs"]["kf_idx"] <= 0]
    assert _gpu(gpu_handle, one) is None
    bad = dict(w); bad["edge_kf"] = w["edge_kf"].copy(); bad["edge_kf"][0, 1] = 99
    with pytest.raises(P.OrbxError):
        _gpu(gpu_handle, bad)
    # points behind a camera keep the 100-px penalty and get no Jacobian (:656-659, :735)
    b = dict(w); b["points"] = w["points"].copy(); b["points"][:5, 2] = -b["points"][:5, 2] - 4.0
    ob = _oracle(b); gb = _gpu(gpu_handle, b)
    assert gb["iterations"] == ob["iterations"] and _rel(gb["points"], ob["points"]) < TOL and _rel(gb["poses_wc"], ob["poses_wc"]) < TOL
    assert np.array_equal(gb["points"][:5], b["points"][:5])


@pytest.mark.gpu
def test_gpu_inertial_ba_failed_call_leaves_no_copy_of_caller_memory_in_flight(gpu_handle):
    """page-locked observations are read by the copy engine where they lie, so a call that FAILS must not return while that
    read is pending.  An inertial call with a bad IMU edge index (validated on the host before anything is enqueued) and one with a bad
    observation index (found on the device; the call drains before it answers): the buffer is scribbled over the moment each call is back,
    restored, and the next valid solve gives the bits of the first."""
    w = P.Handle.pack_ba_windows([P.synth.inertial_window(8, 6, 4000, P.BA_OBS, n_fixed=1)])[0]
    ref = _gpu(gpu_handle, w)
    keep = w["obs"].copy()
    bad = dict(w); bad["edge_kf"] = w["edge_kf"].copy(); bad["edge_kf"][2, 0] = -1
    with pytest.raises(P.OrbxError) as e:
        _gpu(gpu_handle, bad)
    assert "IMU edge 2" in str(e.value)
    w["obs"]["u"][:] = -1.0e9; w["obs"]["mp_idx"][:] = 0            # reuse at once
    w["obs"][:] = keep
    w["obs"]["mp_idx"][5] = len(w["points"]) + 1
    with pytest.raises(P.OrbxError) as e:
        _gpu(gpu_handle, w)
    assert "observation 5" in str(e.value)
    w["obs"]["u"][:] = -1.0e9
    w["obs"][:] = keep
    again = _gpu(gpu_handle, w)
    assert again["iterations"] == ref["iterations"] and all(np.array_equal(again[k], ref[k]) for k in ("poses_wc", "velocities", "biases", "points"))


@pytest.mark.gpu
def test_gpu_solve_inertial_ba_keyed_by_ids(gpu_handle):
    """The reference-shaped entry: InertialBAProblemData keyed by ids; the first keyframe of the window is not reported."""
    w = P.synth.inertial_window(6, 4, 90, P.BA_OBS, n_fixed=2)
    kf_ids = [500 + 11 * k for k in range(4)]
    fixed_ids = [7, 3]
    mp_ids = [9000 + 2 * j for j in range(len(w["points"]))]
    obs = [P.InertialVisualObs(kf_ids[o["kf_idx"]] if o["kf_idx"] >= 0 else fixed_ids[o["fixed_idx"]], mp_ids[o["mp_idx"]], (o["u"], o["v"]),
                               bool(o["_pad"] & 1), o["kf_idx"] >= 0) for o in w["obs"]]
    obs.append(P.InertialVisualObs(kf_ids[0], 123456, (1.0, 2.0), False, True))            # unknown map point: dropped (:1107)
    edges = [P.ImuEdgeData(kf_ids[i], kf_ids[j], w["preint"][e]) for e, (i, j) in enumerate(w["edge_kf"])]
    edges.append(P.ImuEdgeData(kf_ids[3], 999, w["preint"][0]))                              # keyframe outside the window: dropped (:1129)
    prob = P.InertialBAProblemData({k: w["poses_wc"][i] for i, k in enumerate(kf_ids)}, {k: w["velocities"][i] for i, k in enumerate(kf_ids)},
                                   {k: w["biases"][i] for i, k in enumerate(kf_ids)}, {m: w["points"][j] for j, m in enumerate(mp_ids)},
                                   {fixed_ids[f]: w["fixed_cw"][f] for f in range(2)}, obs, edges, kf_ids, mp_ids)
    r = P.solve_inertial_ba(prob, P.CameraModel(**w["camera"]), P.LocalInertialBAConfig(), lambda: False, handle=gpu_handle)
    o = _oracle(w)
    assert r.iterations == o["iterations"] and set(r.optimized_poses) == set(kf_ids[1:]) == set(r.optimized_velocities) == set(r.optimized_biases)
    assert _rel(np.array([r.optimized_poses[k] for k in kf_ids[1:]]), o["poses_wc"][1:]) < TOL
    assert _rel(np.array([r.optimized_biases[k] for k in kf_ids[1:]]), o["biases"][1:]) < TOL
    assert _rel(np.array([r.optimized_points[m] for m in mp_ids]), o["points"]) < TOL
    prob.opt_kf_ids = kf_ids[:1]
    assert P.solve_inertial_ba(prob, P.CameraModel(**w["camera"]), P.LocalInertialBAConfig(), lambda: False, handle=gpu_handle) is None


def _imu_case_states(case):
    poses = np.array(case["poses_wc"], np.float64); vel = np.array(case["velocities"], np.float64)
    st = [np.concatenate([_scaled_axis(poses[k, :4]), poses[k, 4:], vel[k]]) for k in (0, 1)]
    return poses, vel, st


def test_oracle_imu_residual_reference_known_answers(golden):
    """The reference's own known answers for compute_imu_residual (imu_factors.rs:264-321: test_imu_residual_zero_motion,
    test_imu_residual_pure_gravity) on the oracle."""
    for case in golden["imu_residual"]:
        _, _, st = _imu_case_states(case)
        r = O.inertial_imu_residual(st[0], st[1], np.array(case["preint"], np.float64))
        assert np.abs(r - np.array(case["expect"])).max() < case["tol"], case["name"]


@pytest.mark.gpu
def test_gpu_imu_residual_reference_known_answers(gpu_handle, golden):
    """The same known answers read back from the device function ba_imu_kernel calls (orbx_debug_imu_residual), and that function
    against the oracle's on the edges of a synthetic window (ground-truth and perturbed states)."""
    for case in golden["imu_residual"]:
        poses, vel, _ = _imu_case_states(case)
        r = gpu_handle.debug_imu_residual(poses, vel, [[0, 1]], [case["preint"]])[0]
        assert np.abs(r - np.array(case["expect"])).max() < case["tol"], case["name"]
        if case["name"] == "zero_motion":                        # imu_factors.rs:273-275: the three norms
            assert np.linalg.norm(r[:3]) < 1e-10 and np.linalg.norm(r[3:6]) < 1e-10 and np.linalg.norm(r[6:]) < 1e-10
    w = P.synth.inertial_window(3, 6, 40, P.BA_OBS)
    for poses, vel in ((w["gt_poses_wc"], w["gt_velocities"]), (w["poses_wc"], w["velocities"])):
        got = gpu_handle.debug_imu_residual(poses, vel, w["edge_kf"], w["preint"])
        for e, (i, j) in enumerate(w["edge_kf"]):
            si = np.concatenate([_scaled_axis(poses[i, :4]), poses[i, 4:], vel[i]])
            sj = np.concatenate([_scaled_axis(poses[j, :4]), poses[j, 4:], vel[j]])
            want = O.inertial_imu_residual(si, sj, w["preint"][e])
            assert np.abs(got[e] - want).max() < 1e-12 * max(1.0, np.abs(want).max()), e
    with pytest.raises(P.OrbxError):
        gpu_handle.debug_imu_residual(w["poses_wc"], w["velocities"], [[0, 99]], w["preint"][:1])
