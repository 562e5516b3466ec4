"""The BA oracle against the reference's own known answers (local_ba_lm.rs:1144-1243, SURVEY.md
Appendix D4-D6, D12) and its two formulations against each other.  CPU only."""
import numpy as np


def _cam(oracle, g):
    return oracle.Camera(**g)


def test_jacobian_known_answer(oracle, golden):
    g = golden["ba_jacobian_identity"]            # local_ba_lm.rs:1166-1185
    cam = _cam(oracle, g["camera"])
    for key in ("huber_default", "huber_test"):
        hub = g[key]
        e, sw, r, A, B = oracle.ba_obs_terms(cam, hub["threshold"], g["pose_cw"], g["point"], *g["observed"])
        assert np.allclose(e, g["error"], rtol=1e-14)
        assert abs(sw - hub["sqrt_w"]) < 1e-15
        assert np.allclose(r, hub["residual"], rtol=1e-14)
        assert np.allclose(A / sw, g["J_pose"], rtol=1e-13, atol=1e-12)
        assert np.allclose(B / sw, g["J_point"], rtol=1e-13, atol=1e-12)
    # the reference's own test bound: translation columns of the analytic Jacobian agree with a
    # numerical derivative of its additive parameterisation (:1219-1233)
    eps = 1e-5
    J = np.array(g["J_pose"])
    num = np.zeros((2, 6))
    for i in range(6):
        for sgn in (+1, -1):
            p6 = np.zeros(6); p6[i] = sgn * eps
            pose = oracle.se3_from_params(p6)
            e, *_ = oracle.ba_obs_terms(cam, 1e9, pose, g["point"], *g["observed"])
            num[:, i] += sgn * e / (2 * eps)
    assert np.linalg.norm(J[:, 3:] - num[:, 3:]) < 1.0
    assert np.linalg.norm(J - num) < 50.0


def test_behind_camera_rule(oracle, golden):
    cam = _cam(oracle, golden["ba_jacobian_identity"]["camera"])
    e, sw, r, A, B = oracle.ba_obs_terms(cam, 2.5, [1, 0, 0, 0, 0, 0, 0], [0.1, 0.1, 0.0005], 320, 240)
    assert np.array_equal(e, [100.0, 100.0])          # local_ba_lm.rs:201-204
    assert np.any(A != 0)                             # Jacobian still evaluated for |z| >= 1e-6 (:227)
    e, sw, r, A, B = oracle.ba_obs_terms(cam, 2.5, [1, 0, 0, 0, 0, 0, 0], [0.1, 0.1, 1e-7], 320, 240)
    assert np.all(A == 0) and np.all(B == 0)


def test_se3_roundtrip(oracle, golden):
    g = golden["se3_roundtrip"]                       # local_ba_lm.rs:1145-1161
    p6 = oracle.se3_to_params(g["pose"])
    assert np.allclose(p6[:3], g["axis_angle"], atol=1e-14) and np.allclose(p6[3:], [1, 2, 3])
    back = oracle.se3_from_params(p6)
    assert np.allclose(back, g["pose"], atol=g["tol"])
    inv = oracle.se3_inverse(oracle.se3_inverse(g["pose"]))
    assert np.allclose(inv, g["pose"], atol=1e-14)
    assert np.allclose(oracle.se3_from_params(np.zeros(6)), [1, 0, 0, 0, 0, 0, 0])


def test_noise_free_problem_converges_immediately(oracle, pkg):
    # gradient test fires at iteration 1, nothing moves (local_ba_lm.rs:1027-1029)
    w = pkg.synth.ba_window(1, 5, 60, oracle.BA_OBS, noise_px=0.0, perturb=False)
    cam = oracle.Camera(**w["camera"])
    for solve in (oracle.ba_solve_dense, oracle.ba_solve_schur):
        r = solve(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
        assert r["iterations"] == 1 and r["final_error"] < 1e-9
        assert np.allclose(r["points"], w["points"], atol=1e-12)


def test_dense_and_schur_agree(oracle, pkg):
    w = pkg.synth.ba_window(2, 6, 150, oracle.BA_OBS, n_fixed_extra=1)
    cam = oracle.Camera(**w["camera"])
    a = oracle.ba_solve_dense(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
    b = oracle.ba_solve_schur(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"])
    assert a["iterations"] == b["iterations"] >= 3
    assert a["final_error"] < 0.5 * a["initial_error"]
    assert abs(a["final_error"] - b["final_error"]) < 1e-9 * a["final_error"]
    assert np.allclose(a["trace"], b["trace"], rtol=1e-7)
    assert np.allclose(a["poses_wc"], b["poses_wc"], rtol=0, atol=1e-9)
    assert np.allclose(a["points"], b["points"], rtol=0, atol=1e-8)


def test_abort_and_empty(oracle, pkg):
    w = pkg.synth.ba_window(3, 4, 40, oracle.BA_OBS)
    cam = oracle.Camera(**w["camera"])
    r = oracle.ba_solve_schur(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"], stop_after=0)
    assert r["iterations"] == 0 and r["final_error"] == r["initial_error"]     # :1013
    r2 = oracle.ba_solve_schur(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"], stop_after=2)
    assert r2["iterations"] == 2
    assert oracle.ba_solve_schur(cam, oracle.ba_config(), w["poses_cw"], w["fixed_cw"], w["points"], w["obs"][:0]) is None  # :923-925


def test_reduced_system_partition_sums(oracle, pkg):
    """point partition: the partial reduced systems of the two halves add up to the whole"""
    w = pkg.synth.ba_window(4, 5, 80, oracle.BA_OBS)
    cam = oracle.Camera(**w["camera"])
    cfg = oracle.ba_config()
    pp = np.concatenate([oracle.se3_to_params(p) for p in w["poses_cw"]])
    full = oracle.ba_reduced_system(cam, cfg, 1e-3, pp, w["fixed_cw"], w["points"], w["obs"])
    parts = [oracle.ba_reduced_system(cam, cfg, 1e-3, pp, w["fixed_cw"], w["points"], w["obs"][w["obs"]["mp_idx"] % 2 == r])
             for r in range(2)]
    for i in range(4):
        assert np.allclose(parts[0][i] + parts[1][i], full[i], rtol=1e-12, atol=1e-9)
    assert abs(parts[0][4] + parts[1][4] - full[4]) < 1e-9 * full[4]
