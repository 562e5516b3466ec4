// inertial_ba_ref.cpp — CPU restatement of the reference's local inertial bundle adjustment.
// TEST INFRASTRUCTURE ONLY (see oracle.h).  All arithmetic f64.
//
// Follows (paths relative to the reference crate root):
//   src/optimizer/local_inertial_ba.rs:584-613   extract_pose / velocity / bias / point
//   src/optimizer/local_inertial_ba.rs:616-702   compute_all_residuals
//   src/optimizer/local_inertial_ba.rs:708-885   compute_all_jacobians (visual analytic, IMU forward differences)
//   src/optimizer/local_inertial_ba.rs:1074-1275 solve_inertial_ba (LM loop :1198-1243)
//   src/optimizer/imu_factors.rs:66-103          compute_imu_residual
//   src/imu/sample.rs:6                          GRAVITY = (0, 0, -9.81)
// nalgebra restated: UnitQuaternion::from_scaled_axis (= exp of the pure quaternion v/2, identity when |v/2|^2 <= eps^2),
// scaled_axis, quaternion product, `UnitQuaternion * Vector3`, to_rotation_matrix, DMatrix::lu().solve().
//
// The normal equations are accumulated block by block (the same sums as the reference's dense J^T J, in another
// order) and solved densely with partial-pivot LU, as the reference does (:1222).
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

struct Q { double w, x, y, z; };
inline Q q_mul(const Q& a, const Q& b) {
  return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
           a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline Q q_conj(const Q& q) { return Q{q.w, -q.x, -q.y, -q.z}; }
inline void q_rot(const Q& q, const double* v, double* o) {
  const double t[3] = {2.0 * (q.y * v[2] - q.z * v[1]), 2.0 * (q.z * v[0] - q.x * v[2]), 2.0 * (q.x * v[1] - q.y * v[0])};
  const double c[3] = {q.y * t[2] - q.z * t[1], q.z * t[0] - q.x * t[2], q.x * t[1] - q.y * t[0]};
  for (int i = 0; i < 3; ++i) o[i] = t[i] * q.w + c[i] + v[i];
}
inline Q q_from_scaled_axis(const double* r) {               // nalgebra from_scaled_axis -> Quaternion::exp
  const double v[3] = {r[0] / 2.0, r[1] / 2.0, r[2] / 2.0};
  const double nn = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double eps = 2.220446049250313e-16;
  if (nn <= eps * eps) return Q{1, 0, 0, 0};
  const double n = std::sqrt(nn), s = 1.0 * std::sin(n) / n;
  return Q{1.0 * std::cos(n), v[0] * s, v[1] * s, v[2] * s};
}
inline void q_scaled_axis(const Q& q, double* o) {           // nalgebra scaled_axis
  double v[3] = {q.x, q.y, q.z};
  if (!(q.w >= 0.0)) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n > 0.0) {
    const double ang = std::atan2(n, std::fabs(q.w)) * 2.0;
    o[0] = v[0] / n * ang; o[1] = v[1] / n * ang; o[2] = v[2] / n * ang;
  } else o[0] = o[1] = o[2] = 0.0;
}
inline void q_to_R(const Q& q, double* R) {
  const double w = q.w, i = q.x, j = q.y, k = q.z;
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k, ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0,
               jk = j * k * 2.0, wi = w * i * 2.0;
  R[0] = ww + ii - jj - kk; R[1] = ij - wk; R[2] = wj + ik; R[3] = wk + ij; R[4] = ww - ii + jj - kk; R[5] = jk - wi;
  R[6] = ik - wj; R[7] = wi + jk; R[8] = ww - ii - jj + kk;
}

struct Prob {
  orbx_camera cam;
  orbx_inertial_ba_config cfg;
  int K, F, M, N, E;
  const double* fixed_cw;
  const orbx_ba_obs* obs;
  const int* edge_kf;          // [E][2]
  const double* preint;        // [E][11]: qw qx qy qz | dv | dp | dt
  size_t np() const { return 15 * (size_t)K + 3 * (size_t)M; }
};

// imu_factors.rs:66-103 with the poses taken from the parameter vector (:584-598)
void imu_residual(const Prob& P, const std::vector<double>& p, int e, double* r9) {
  const int ki = P.edge_kf[2 * e], kj = P.edge_kf[2 * e + 1];
  const double* pi = &p[15 * (size_t)ki];
  const double* pj = &p[15 * (size_t)kj];
  const double* pre = P.preint + 11 * (size_t)e;
  const double dt = pre[10];
  const Q ri = q_from_scaled_axis(pi), rj = q_from_scaled_axis(pj), dr{pre[0], pre[1], pre[2], pre[3]};
  const Q err = q_mul(q_mul(q_conj(dr), q_conj(ri)), rj);                 // :85
  q_scaled_axis(err, r9);
  const double g[3] = {0.0, 0.0, -9.81};
  double a[3], b[3];
  for (int i = 0; i < 3; ++i) a[i] = pj[6 + i] - pi[6 + i] - g[i] * dt;   // :89
  q_rot(q_conj(ri), a, b);
  for (int i = 0; i < 3; ++i) r9[3 + i] = b[i] - pre[4 + i];
  for (int i = 0; i < 3; ++i) a[i] = pj[3 + i] - pi[3 + i] - pi[6 + i] * dt - 0.5 * g[i] * dt * dt;   // :93-94
  q_rot(q_conj(ri), a, b);
  for (int i = 0; i < 3; ++i) r9[6 + i] = b[i] - pre[7 + i];
}

struct VisTerms { double r[2], A[12], B[6]; bool front; };
// :633-659 (residual), :735-804 (Jacobian rows)
void vis_terms(const Prob& P, const std::vector<double>& p, const orbx_ba_obs& o, VisTerms& t) {
  Q qcw; double tcw[3];
  if (o.kf_idx >= 0) {
    const double* pp = &p[15 * (size_t)o.kf_idx];
    const Q qwc = q_from_scaled_axis(pp);
    qcw = q_conj(qwc);                                                     // SE3::inverse, se3.rs:56-63
    double rt[3];
    q_rot(qcw, pp + 3, rt);
    tcw[0] = -rt[0]; tcw[1] = -rt[1]; tcw[2] = -rt[2];
  } else if (o.fixed_idx >= 0 && o.fixed_idx < P.F) {
    const double* f = P.fixed_cw + 7 * (size_t)o.fixed_idx;
    qcw = Q{f[0], f[1], f[2], f[3]}; tcw[0] = f[4]; tcw[1] = f[5]; tcw[2] = f[6];
  } else { qcw = Q{1, 0, 0, 0}; tcw[0] = tcw[1] = tcw[2] = 0.0; }           // :637 unwrap_or_else(SE3::identity)
  const double* X = &p[15 * (size_t)P.K + 3 * (size_t)o.mp_idx];
  double pc[3];
  q_rot(qcw, X, pc);
  pc[0] += tcw[0]; pc[1] += tcw[1]; pc[2] += tcw[2];
  memset(t.A, 0, sizeof(t.A)); memset(t.B, 0, sizeof(t.B));
  t.front = pc[2] > 0.001;
  if (!t.front) { t.r[0] = 100.0; t.r[1] = 100.0; return; }               // :656-659, no Jacobian (:735)
  const double x = pc[0], y = pc[1], z = pc[2], fx = P.cam.fx, fy = P.cam.fy;
  const double thr = (o._pad & 1) ? P.cfg.huber_threshold_stereo : P.cfg.huber_threshold_mono;
  {
    const double u = fx * x / z + P.cam.cx, v = fy * y / z + P.cam.cy;   // :644-646
    const double e0 = o.u - u, e1 = o.v - v;
    const double en = std::sqrt(e0 * e0 + e1 * e1);
    const double w = en <= thr ? 1.0 : thr / en;
    t.r[0] = e0 * std::sqrt(w); t.r[1] = e1 * std::sqrt(w);
  }
  const double zi = 1.0 / z, zi2 = zi * zi;
  const double u = fx * x * zi + P.cam.cx, v = fy * y * zi + P.cam.cy;    // :743-745
  const double e0 = o.u - u, e1 = o.v - v;
  const double en = std::sqrt(e0 * e0 + e1 * e1);
  const double sw = en <= thr ? 1.0 : std::sqrt(thr / en);
  const double du[3] = {fx * zi, 0.0, -fx * x * zi2}, dv[3] = {0.0, fy * zi, -fy * y * zi2};
  double R[9];
  q_to_R(qcw, R);
  for (int c = 0; c < 3; ++c) {                                            // R_cw^T * d(u,v)/dp_cam, negated (:760-771)
    const double a = R[0 * 3 + c] * du[0] + R[1 * 3 + c] * du[1] + R[2 * 3 + c] * du[2];
    const double b = R[0 * 3 + c] * dv[0] + R[1 * 3 + c] * dv[1] + R[2 * 3 + c] * dv[2];
    t.B[c] = -a * sw; t.B[3 + c] = -b * sw;
  }
  if (o.kf_idx >= 0) {                                                     // :774-803
    const double xy = x * y, xs = x * x, ys = y * y;
    t.A[0] = -fx * xy * zi2 * sw; t.A[1] = fx * (1.0 + xs * zi2) * sw; t.A[2] = -fx * y * zi * sw;
    t.A[3] = fx * zi * sw; t.A[4] = 0.0; t.A[5] = -fx * x * zi2 * sw;
    t.A[6] = -fy * (1.0 + ys * zi2) * sw; t.A[7] = fy * xy * zi2 * sw; t.A[8] = fy * x * zi * sw;
    t.A[9] = 0.0; t.A[10] = fy * zi * sw; t.A[11] = -fy * y * zi2 * sw;
  }
}

double total_sq(const Prob& P, const std::vector<double>& p) {            // |compute_all_residuals|^2
  double s = 0.0;
  VisTerms t;
  for (int i = 0; i < P.N; ++i) { vis_terms(P, p, P.obs[i], t); s += t.r[0] * t.r[0]; s += t.r[1] * t.r[1]; }
  const double gw = std::sqrt(P.cfg.gyro_rw_info), aw = std::sqrt(P.cfg.accel_rw_info);
  for (int e = 0; e < P.E; ++e) {
    double r[9];
    imu_residual(P, p, e, r);
    for (int k = 0; k < 9; ++k) s += r[k] * r[k];
  }
  for (int e = 0; e < P.E; ++e) {
    const double* bi = &p[15 * (size_t)P.edge_kf[2 * e] + 9];
    const double* bj = &p[15 * (size_t)P.edge_kf[2 * e + 1] + 9];
    for (int k = 0; k < 3; ++k) { const double d = (bj[k] - bi[k]) * gw; s += d * d; }
    for (int k = 0; k < 3; ++k) { const double d = (bj[3 + k] - bi[3 + k]) * aw; s += d * d; }
  }
  return s;
}

bool lu_solve_dense(std::vector<double>& A, int n, std::vector<double>& b) {   // partial pivoting, as nalgebra's LU
  std::vector<int> piv(n);
  for (int c = 0; c < n; ++c) {
    int p = c; double best = std::fabs(A[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r) { const double v = std::fabs(A[(size_t)r * n + c]); if (v > best) { best = v; p = r; } }
    if (best == 0.0) return false;
    if (p != c) { for (int k = 0; k < n; ++k) std::swap(A[(size_t)p * n + k], A[(size_t)c * n + k]); std::swap(b[p], b[c]); }
    const double d = A[(size_t)c * n + c];
    for (int r = c + 1; r < n; ++r) {
      const double f = A[(size_t)r * n + c] / d;
      if (f == 0.0) continue;
      A[(size_t)r * n + c] = f;
      for (int k = c + 1; k < n; ++k) A[(size_t)r * n + k] -= f * A[(size_t)c * n + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[(size_t)r * n + k] * b[k];
    b[r] = s / A[(size_t)r * n + r];
  }
  return true;
}

}  // namespace

extern "C" {

void oracle_inertial_imu_residual(const double* state_i9, const double* state_j9, const double* preint11, double* r9) {
  Prob P{};
  P.K = 2; P.M = 0; P.E = 1;
  const int ek[2] = {0, 1};
  P.edge_kf = ek; P.preint = preint11;
  std::vector<double> p(30, 0.0);
  memcpy(&p[0], state_i9, 9 * sizeof(double)); memcpy(&p[15], state_j9, 9 * sizeof(double));
  imu_residual(P, p, 0, r9);
}

// solve_inertial_ba (:1074-1275) on flat arrays: states [K] as T_wc pose (7) + velocity (3) + bias (gyro 3, accel 3).
// obs[i]._pad bit 0 = is_stereo.  trace [max_iterations][4]: current |r|^2, |gradient|, |delta|, trial |r|^2.
int oracle_inertial_ba_solve(const orbx_camera* cam, const orbx_inertial_ba_config* cfg, int K, const double* poses_wc,
                             const double* velocities, const double* biases, int F, const double* fixed_poses_cw, int M,
                             double* points, int N, const orbx_ba_obs* obs, int E, const int* edge_kf, const double* preint,
                             int stop_after, double* poses_wc_out, double* vel_out, double* bias_out, int* iterations,
                             double* initial_error, double* final_error, double* trace) {
  if (K < 2) return ORBX_ERR_EMPTY;                                        // :1080-1082
  Prob P{};
  P.cam = *cam; P.cfg = *cfg; P.K = K; P.F = F; P.M = M; P.N = N; P.E = E;
  P.fixed_cw = fixed_poses_cw; P.obs = obs; P.edge_kf = edge_kf; P.preint = preint;
  const int np = (int)P.np();
  std::vector<double> p((size_t)np, 0.0);
  for (int k = 0; k < K; ++k) {                                            // :1140-1174
    const double* s = poses_wc + 7 * (size_t)k;
    q_scaled_axis(Q{s[0], s[1], s[2], s[3]}, &p[15 * (size_t)k]);
    for (int i = 0; i < 3; ++i) p[15 * (size_t)k + 3 + i] = s[4 + i];
    for (int i = 0; i < 3; ++i) p[15 * (size_t)k + 6 + i] = velocities[3 * (size_t)k + i];
    for (int i = 0; i < 6; ++i) p[15 * (size_t)k + 9 + i] = biases[6 * (size_t)k + i];
  }
  for (int j = 0; j < 3 * M; ++j) p[15 * (size_t)K + j] = points[j];
  *initial_error = std::sqrt(total_sq(P, p));                              // :1188-1191 (norm, not RMS)
  double lambda = cfg->initial_lambda;
  int iters = 0;
  const double eps = 1e-6, gw = std::sqrt(cfg->gyro_rw_info), aw = std::sqrt(cfg->accel_rw_info);
  for (int iter = 0; iter < cfg->max_iterations; ++iter) {
    if (stop_after >= 0 && iter >= stop_after) break;                      // :1199-1201
    iters = iter + 1;
    std::vector<double> H((size_t)np * np, 0.0), g((size_t)np, 0.0);
    double cur_sq = 0.0;
    auto add_rows = [&](const int* cols, int nc, const double* Jrow, double r) {   // one residual row
      for (int a = 0; a < nc; ++a) {
        if (Jrow[a] == 0.0) continue;
        g[cols[a]] += Jrow[a] * r;
        for (int b = 0; b < nc; ++b) H[(size_t)cols[a] * np + cols[b]] += Jrow[a] * Jrow[b];
      }
    };
    VisTerms t;
    for (int i = 0; i < N; ++i) {
      const orbx_ba_obs& o = obs[i];
      vis_terms(P, p, o, t);
      cur_sq += t.r[0] * t.r[0]; cur_sq += t.r[1] * t.r[1];
      if (!t.front) continue;
      int cols[9]; double J[9];
      for (int row = 0; row < 2; ++row) {
        int nc = 0;
        if (o.kf_idx >= 0) for (int c = 0; c < 6; ++c) { cols[nc] = 15 * o.kf_idx + c; J[nc++] = t.A[row * 6 + c]; }
        for (int c = 0; c < 3; ++c) { cols[nc] = 15 * K + 3 * o.mp_idx + c; J[nc++] = t.B[row * 3 + c]; }
        add_rows(cols, nc, J, t.r[row]);
      }
    }
    for (int e = 0; e < E; ++e) {                                          // :806-861 forward differences, eps = 1e-6
      double base[9];
      imu_residual(P, p, e, base);
      for (int k = 0; k < 9; ++k) cur_sq += base[k] * base[k];
      const int kk[2] = {edge_kf[2 * e], edge_kf[2 * e + 1]};
      int cols[18]; double Jc[18][9];
      int nc = 0;
      for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 9; ++j) {                                      // pose 0..5, velocity 6..8
          const int col = 15 * kk[s] + j;
          // the same keyframe on both ends of an edge would be perturbed once per appearance (:822); edges link distinct ones
          std::vector<double> pp(p);
          pp[col] += eps;
          double plus[9];
          imu_residual(P, pp, e, plus);
          cols[nc] = col;
          for (int k = 0; k < 9; ++k) Jc[nc][k] = (plus[k] - base[k]) / eps;
          ++nc;
        }
      for (int k = 0; k < 9; ++k) {
        double Jrow[18];
        for (int a = 0; a < nc; ++a) Jrow[a] = Jc[a][k];
        add_rows(cols, nc, Jrow, base[k]);
      }
    }
    for (int e = 0; e < E; ++e) {                                          // :676-698 residual, :863-880 Jacobian
      const int ki = edge_kf[2 * e], kj = edge_kf[2 * e + 1];
      for (int k = 0; k < 6; ++k) {
        const double wgt = k < 3 ? gw : aw;
        const double r = (p[15 * (size_t)kj + 9 + k] - p[15 * (size_t)ki + 9 + k]) * wgt;
        cur_sq += r * r;
        const int cols[2] = {15 * ki + 9 + k, 15 * kj + 9 + k};
        const double J[2] = {-wgt, wgt};
        add_rows(cols, 2, J, r);
      }
    }
    double gn = 0.0;
    for (double v : g) gn += v * v;
    gn = std::sqrt(gn);
    if (trace) { trace[4 * iter] = cur_sq; trace[4 * iter + 1] = gn; trace[4 * iter + 2] = 0; trace[4 * iter + 3] = 0; }
    if (gn < 1e-8) break;                                                  // :1213-1215
    for (int i = 0; i < np; ++i) H[(size_t)i * np + i] += lambda * std::fmax(H[(size_t)i * np + i], 1e-6);   // :1217-1221
    std::vector<double> delta((size_t)np);
    for (int i = 0; i < np; ++i) delta[i] = -g[i];
    if (!lu_solve_dense(H, np, delta)) break;                              // :1222-1225
    double dn = 0.0;
    for (double v : delta) dn += v * v;
    if (trace) trace[4 * iter + 2] = std::sqrt(dn);
    std::vector<double> trial((size_t)np);
    for (int i = 0; i < np; ++i) trial[i] = p[i] + delta[i];
    const double trial_sq = total_sq(P, trial);
    if (trace) trace[4 * iter + 3] = trial_sq;
    if (trial_sq < cur_sq) { p.swap(trial); lambda = std::fmax(lambda * 0.1, 1e-10); }   // :1233-1238
    else lambda = std::fmin(lambda * 10.0, 1e10);
  }
  *iterations = iters;
  *final_error = std::sqrt(total_sq(P, p));                                // :1241-1244
  for (int k = 0; k < K; ++k) {                                            // :1250-1254 (every keyframe; the caller skips the first)
    const Q q = q_from_scaled_axis(&p[15 * (size_t)k]);
    double* o = poses_wc_out + 7 * (size_t)k;
    o[0] = q.w; o[1] = q.x; o[2] = q.y; o[3] = q.z;
    for (int i = 0; i < 3; ++i) o[4 + i] = p[15 * (size_t)k + 3 + i];
    for (int i = 0; i < 3; ++i) vel_out[3 * (size_t)k + i] = p[15 * (size_t)k + 6 + i];
    for (int i = 0; i < 6; ++i) bias_out[6 * (size_t)k + i] = p[15 * (size_t)k + 9 + i];
  }
  for (int j = 0; j < 3 * M; ++j) points[j] = p[15 * (size_t)K + j];
  return ORBX_OK;
}

}  // extern "C"
