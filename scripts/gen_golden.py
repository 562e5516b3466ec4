#!/usr/bin/env python3
"""Writes tests/golden/reference_known_answers.json: the known-answer values the reference's own
unit tests hold for the hot path, plus the values derived (with numpy, NOT with the oracle) from
the reference's formulas at the inputs of those tests — SURVEY.md Appendix D.

Nothing from the reference is executed or copied: each entry cites the reference file:line that
fixes the inputs / the expected value.
"""
import json, math, os
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = {}

# D1-D3: Hamming known answers, vocabulary/mod.rs:429-441 and loop_closing/corrector.rs:625-634
a = [0] * 32
c = [0xFF] + [0] * 31
c2 = [0xFF, 0x0F] + [0] * 30
G["hamming"] = [
    dict(a=a, b=a, expect=0, cite="vocabulary/mod.rs:431-433; corrector.rs:627-629"),
    dict(a=a, b=c, expect=8, cite="vocabulary/mod.rs:435-437; corrector.rs:631-633"),
    dict(a=a, b=c2, expect=12, cite="vocabulary/mod.rs:439-440"),
]

# D4/D5: pose/point Jacobian and residual at the inputs of test_jacobian_pose_numerical,
# local_ba_lm.rs:1166-1185,1211 (identity pose, X=(0.5,0.3,3), fx=fy=400, cx=320, cy=240,
# observed (320,240)), evaluated from the formulas at :207-211, :239-254, :281-287, :291-297
fx = fy = 400.0; cx, cy = 320.0, 240.0
x, y, z = 0.5, 0.3, 3.0
iz = 1.0 / z; iz2 = iz * iz
Jp = [[x * y * iz2 * fx, -(1 + x * x * iz2) * fx, y * iz * fx, -iz * fx, 0.0, x * iz2 * fx],
      [(1 + y * y * iz2) * fy, -x * y * iz2 * fy, -x * iz * fy, 0.0, -iz * fy, y * iz2 * fy]]
tmp = np.array([[fx, 0, -fx * x * iz], [0, fy, -fy * y * iz]])
Jx = (-iz * tmp @ np.eye(3)).tolist()
u, v = fx * x / z + cx, fy * y / z + cy
e = [320.0 - u, 240.0 - v]
en = math.hypot(*e)
G["ba_jacobian_identity"] = dict(
    camera=dict(fx=fx, fy=fy, cx=cx, cy=cy, baseline=0.1), pose_cw=[1, 0, 0, 0, 0, 0, 0],
    point=[x, y, z], observed=[320.0, 240.0], J_pose=Jp, J_point=Jx, projection=[u, v], error=e,
    cite="local_ba_lm.rs:1166-1185,1211 inputs; formulas :207-211,:239-254,:281-287")
for name, th in (("huber_default", math.sqrt(5.991)), ("huber_test", 2.5)):
    w = 1.0 if en <= th else th / en
    G["ba_jacobian_identity"][name] = dict(threshold=th, weight=w, sqrt_w=math.sqrt(w),
                                           residual=[e[0] * math.sqrt(w), e[1] * math.sqrt(w)])

# D6: axis-angle round trip of euler(0.1,0.2,0.3), t=(1,2,3), local_ba_lm.rs:1145-1161
# nalgebra from_euler_angles(roll,pitch,yaw) = Rz(yaw) Ry(pitch) Rx(roll)
def q_axis(ax, ang):
    s = math.sin(ang / 2); return np.array([math.cos(ang / 2), ax[0] * s, ax[1] * s, ax[2] * s])
def qmul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1*w2-x1*x2-y1*y2-z1*z2, w1*x2+x1*w2+y1*z2-z1*y2, w1*y2-x1*z2+y1*w2+z1*x2, w1*z2+x1*y2-y1*x2+z1*w2])
q = qmul(q_axis([0, 0, 1], 0.3), qmul(q_axis([0, 1, 0], 0.2), q_axis([1, 0, 0], 0.1)))
ang = 2 * math.atan2(np.linalg.norm(q[1:]), abs(q[0]))
aa = (q[1:] / np.linalg.norm(q[1:]) * ang).tolist()
G["se3_roundtrip"] = dict(pose=q.tolist() + [1.0, 2.0, 3.0], axis_angle=aa, tol=1e-10,
                          cite="local_ba_lm.rs:1145-1161")

# D7/D8: disparity bounds and one triangulation with the EuRoC cam0 numbers, stereo.rs:89-90,204-211
fxE, fyE, cxE, cyE, b = 458.654, 457.296, 367.215, 248.375, 0.11007
G["disparity_bounds"] = dict(camera=dict(fx=fxE, fy=fyE, cx=cxE, cy=cyE, baseline=b),
                             max_disp=float(np.float32(fxE * b / 0.1)), min_disp=float(np.float32(fxE * b / 40.0)),
                             cite="stereo.rs:89-90")
d = 400.0 - 380.0
zz = fxE * b / d
G["triangulate"] = dict(xl=400.0, yl=200.0, xr=380.0, point=[(400.0 - cxE) * zz / fxE, (200.0 - cyE) * zz / fyE, zz],
                        cite="stereo.rs:204-211")

# D9/D10: per-level feature quotas and the IC-disc half widths
G["quota"] = {"1200": [261, 217, 181, 151, 126, 105, 87, 72], "2000": [434, 362, 302, 251, 209, 175, 145, 122],
              "4000": [869, 724, 603, 503, 419, 349, 291, 242]}
G["level_sizes"] = {"752x480": [[752, 480], [627, 400], [522, 333], [435, 278], [363, 231], [302, 193], [252, 161], [210, 134]],
                    "1920x1080": [[1920, 1080], [1600, 900], [1333, 750], [1111, 625], [926, 521], [772, 434], [643, 362], [536, 301]]}
G["umax"] = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]

# D13/D14: the reference's known answers for the preintegration factor, imu_factors.rs:264-321 (compute_imu_residual :68-103,
# GRAVITY = (0, 0, -9.81) imu/sample.rs:6, PreintegratedState::identity imu/preintegration.rs:101-111).
#   zero motion (:264-276): identity pose, zero velocity, both keyframes the same, identity preintegration (dt = 0) -> every
#   component < 1e-10;
#   free fall (:278-321): identity poses, v_i = 0, v_j = g * dt, dt = 0.1; with delta_vel = 0 the velocity residual is
#   R_i^T (v_j - v_i - g dt) - 0 = 0 (< 0.01 in the reference's assert) and, from the formulas at :85-99, the rotation residual
#   is 0 and the position residual R_i^T (p_j - p_i - v_i dt - g dt^2 / 2) - 0 = (0, 0, +0.04905); with delta_vel = g * dt
#   (the first half of that test) the velocity residual is -g * dt = (0, 0, +0.981).
g = [0.0, 0.0, -9.81]; dt = 0.1
ident7 = [1.0, 0, 0, 0, 0, 0, 0]
G["imu_residual"] = [
    dict(name="zero_motion", poses_wc=[ident7, ident7], velocities=[[0, 0, 0], [0, 0, 0]], preint=[1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.0],
         expect=[0.0] * 9, tol=1e-10, cite="imu_factors.rs:264-276"),
    dict(name="free_fall_consistent", poses_wc=[ident7, ident7], velocities=[[0, 0, 0], [g[0] * dt, g[1] * dt, g[2] * dt]],
         preint=[1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, dt], expect=[0, 0, 0, 0, 0, 0, 0, 0, -0.5 * g[2] * dt * dt], tol=1e-12,
         cite="imu_factors.rs:303-320 (velocity residual < 0.01); position component from :93-96"),
    dict(name="free_fall_delta_vel_g_dt", poses_wc=[ident7, ident7], velocities=[[0, 0, 0], [g[0] * dt, g[1] * dt, g[2] * dt]],
         preint=[1.0, 0, 0, 0, g[0] * dt, g[1] * dt, g[2] * dt, 0, 0, 0, dt], expect=[0, 0, 0, 0, 0, -g[2] * dt, 0, 0, -0.5 * g[2] * dt * dt],
         tol=1e-12, cite="imu_factors.rs:290-302 (the residual the comment derives: -g * dt)"),
]

with open(os.path.join(root, "tests", "golden", "reference_known_answers.json"), "w") as f:
    json.dump(G, f, indent=1)
print("written")
