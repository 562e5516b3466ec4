// ba_ref.cpp — CPU restatement of the reference's visual local bundle adjustment.
// TEST INFRASTRUCTURE ONLY (see oracle.h).  All arithmetic f64.
//
// Follows (paths relative to the reference crate root):
//   src/optimizer/local_ba_lm.rs:192-212   compute_error
//   src/optimizer/local_ba_lm.rs:216-288   jacobian_pose / jacobian_point (g2o EdgeSE3ProjectXYZ)
//   src/optimizer/local_ba_lm.rs:291-297   huber_weight
//   src/optimizer/local_ba_lm.rs:557-639   compute_residuals / compute_jacobian
//   src/optimizer/local_ba_lm.rs:642-662   se3_to_params / se3_from_params
//   src/optimizer/local_ba_lm.rs:912-1098  solve_visual_ba (LM loop :1004-1056)
//   src/geometry/se3.rs:56-76              SE3::inverse / transform_point
// nalgebra 0.34.1 (Cargo.lock:4879-4880) supplies quaternion ops, dense GEMM and partial-pivot
// LU; those standard algorithms are restated here (UnitQuaternion::scaled_axis, from_axis_angle,
// `UnitQuaternion * Vector3`, to_rotation_matrix, DMatrix::lu().solve()).
//
// Two solvers: oracle_ba_solve_dense is the literal formulation (dense 2N x P Jacobian, dense
// JtJ, LU) for problems small enough to hold it; oracle_ba_solve_schur is the block-structured
// equivalent (last bullet) the GPU path is organised like.  tests/ cross-check
// the two and pin both with the reference's own known-answer values (Appendix D4-D6, D12).
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

struct Pose { double q[4]; double t[3]; };  // q = (w, x, y, z), T_cw

inline Pose pose_from7(const double* p) { Pose r; memcpy(r.q, p, 4 * sizeof(double)); memcpy(r.t, p + 4, 3 * sizeof(double)); return r; }
inline void pose_to7(const Pose& p, double* o) { memcpy(o, p.q, 4 * sizeof(double)); memcpy(o + 4, p.t, 3 * sizeof(double)); }
inline Pose pose_identity() { Pose r = {{1, 0, 0, 0}, {0, 0, 0}}; return r; }

inline void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// nalgebra `UnitQuaternion * Vector3`: t = 2 (imag x v); v + w t + imag x t
inline void quat_rotate(const double* q, const double* v, double* o) {
  double t[3], c[3];
  cross3(q + 1, v, t);
  t[0] *= 2.0; t[1] *= 2.0; t[2] *= 2.0;
  cross3(q + 1, t, c);
  for (int i = 0; i < 3; ++i) o[i] = t[i] * q[0] + c[i] + v[i];
}
// se3.rs:74-76
inline void transform_point(const Pose& T, const double* X, double* o) {
  quat_rotate(T.q, X, o);
  o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2];
}
// nalgebra UnitQuaternion::to_rotation_matrix (row-major 3x3)
inline void quat_to_R(const double* q, double* R) {
  const double w = q[0], i = q[1], j = q[2], k = q[3];
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  const double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0;
  const double ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
  R[0] = ww + ii - jj - kk; R[1] = ij - wk;           R[2] = wj + ik;
  R[3] = wk + ij;           R[4] = ww - ii + jj - kk; R[5] = jk - wi;
  R[6] = ik - wj;           R[7] = wi + jk;           R[8] = ww - ii - jj + kk;
}
// local_ba_lm.rs:642-645 + nalgebra scaled_axis(): axis (sign-fixed imag, normalised) * angle,
// angle = 2 atan2(|imag|, |w|); zero vector when imag == 0.
inline void se3_to_params(const Pose& T, double* p6) {
  const double w = T.q[0];
  double v[3] = {T.q[1], T.q[2], T.q[3]};
  if (!(w >= 0.0)) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n > 0.0) {
    const double ang = std::atan2(n, std::fabs(w)) * 2.0;
    p6[0] = v[0] / n * ang; p6[1] = v[1] / n * ang; p6[2] = v[2] / n * ang;
  } else {
    p6[0] = p6[1] = p6[2] = 0.0;
  }
  p6[3] = T.t[0]; p6[4] = T.t[1]; p6[5] = T.t[2];
}
// local_ba_lm.rs:648-662
inline Pose se3_from_params(const double* p6) {
  Pose T;
  const double angle = std::sqrt(p6[0] * p6[0] + p6[1] * p6[1] + p6[2] * p6[2]);
  if (angle > 1e-10) {
    double a[3] = {p6[0] / angle, p6[1] / angle, p6[2] / angle};
    const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);  // Unit::new_normalize
    a[0] /= n; a[1] /= n; a[2] /= n;
    const double s = std::sin(angle / 2.0), c = std::cos(angle / 2.0);
    T.q[0] = c; T.q[1] = a[0] * s; T.q[2] = a[1] * s; T.q[3] = a[2] * s;
  } else {
    T.q[0] = 1; T.q[1] = T.q[2] = T.q[3] = 0;
  }
  T.t[0] = p6[3]; T.t[1] = p6[4]; T.t[2] = p6[5];
  return T;
}
// se3.rs:56-63
inline Pose se3_inverse(const Pose& T) {
  Pose r;
  r.q[0] = T.q[0]; r.q[1] = -T.q[1]; r.q[2] = -T.q[2]; r.q[3] = -T.q[3];
  double rt[3];
  quat_rotate(r.q, T.t, rt);
  r.t[0] = -rt[0]; r.t[1] = -rt[1]; r.t[2] = -rt[2];
  return r;
}

struct ObsTerms { double e[2], sw, r[2], A[12], B[6]; };

// one observation: error (:192-212), Huber (:291-297, :578-584), Jacobian blocks (:216-288) * sqrt(w)
// zero_behind: global_ba.rs:561-563 leaves both Jacobian rows of an observation zero when z_c <= 0.001 (the
// residual keeps its 100-px penalty); the local solver only zeroes them for |z_c| < 1e-6 (:227, :270).
inline void obs_terms(const orbx_camera& cam, double huber, const Pose& T, const double* X, double u,
                      double v, ObsTerms& o, bool zero_behind = false) {
  double pc[3];
  transform_point(T, X, pc);
  const double x = pc[0], y = pc[1], z = pc[2];
  if (z <= 0.001) { o.e[0] = 100.0; o.e[1] = 100.0; }
  else {
    const double pu = cam.fx * x / z + cam.cx;
    const double pv = cam.fy * y / z + cam.cy;
    o.e[0] = u - pu; o.e[1] = v - pv;
  }
  const double en = std::sqrt(o.e[0] * o.e[0] + o.e[1] * o.e[1]);
  const double w = (en <= huber) ? 1.0 : huber / en;
  o.sw = std::sqrt(w);
  o.r[0] = o.e[0] * o.sw; o.r[1] = o.e[1] * o.sw;
  if (std::fabs(z) < 1e-6 || (zero_behind && z <= 0.001)) {
    memset(o.A, 0, sizeof(o.A));
    memset(o.B, 0, sizeof(o.B));
    return;
  }
  const double invz = 1.0 / z, invz2 = invz * invz, fx = cam.fx, fy = cam.fy;
  o.A[0] = x * y * invz2 * fx;          o.A[1] = -(1.0 + x * x * invz2) * fx; o.A[2] = y * invz * fx;
  o.A[3] = -invz * fx;                  o.A[4] = 0.0;                         o.A[5] = x * invz2 * fx;
  o.A[6] = (1.0 + y * y * invz2) * fy;  o.A[7] = -x * y * invz2 * fy;         o.A[8] = -x * invz * fy;
  o.A[9] = 0.0;                         o.A[10] = -invz * fy;                 o.A[11] = y * invz2 * fy;
  double R[9];
  quat_to_R(T.q, R);
  const double tmp[6] = {fx, 0.0, -fx * x * invz, 0.0, fy, -fy * y * invz};
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += (-invz * tmp[r * 3 + k]) * R[k * 3 + c];
      o.B[r * 3 + c] = s;
    }
  for (int i = 0; i < 12; ++i) o.A[i] *= o.sw;
  for (int i = 0; i < 6; ++i) o.B[i] *= o.sw;
}

struct Problem {
  orbx_camera cam;
  orbx_ba_config cfg;
  int K, F, M, N;
  std::vector<Pose> fixed;
  const orbx_ba_obs* obs;
  bool global_mode = false;   // solve_global_ba (global_ba.rs:184-418) instead of solve_visual_ba
};

inline Pose obs_pose(const Problem& P, const std::vector<double>& params, const orbx_ba_obs& o) {
  if (o.kf_idx >= 0) return se3_from_params(&params[6 * (size_t)o.kf_idx]);
  if (o.fixed_idx >= 0 && o.fixed_idx < P.F) return P.fixed[o.fixed_idx];
  return pose_identity();  // local_ba_lm.rs:569 unwrap_or_else(SE3::identity)
}

double residual_sq(const Problem& P, const std::vector<double>& params) {
  double s = 0.0;
  ObsTerms t;
  for (int i = 0; i < P.N; ++i) {
    const orbx_ba_obs& o = P.obs[i];
    obs_terms(P.cam, P.cfg.huber_threshold, obs_pose(P, params, o),
              &params[6 * (size_t)P.K + 3 * (size_t)o.mp_idx], o.u, o.v, t);
    s += t.r[0] * t.r[0];
    s += t.r[1] * t.r[1];
  }
  return s;
}

double vec_norm(const std::vector<double>& v) {
  double s = 0.0;
  for (double x : v) s += x * x;
  return std::sqrt(s);
}

// DMatrix::lu().solve(): partial (row) pivoting; None on an exactly zero pivot.
bool lu_solve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int c = 0; c < n; ++c) {
    int piv = c;
    double best = std::fabs(A[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r) {
      const double a = std::fabs(A[(size_t)r * n + c]);
      if (a > best) { best = a; piv = r; }
    }
    if (best == 0.0) return false;
    if (piv != c) {
      for (int k = 0; k < n; ++k) std::swap(A[(size_t)c * n + k], A[(size_t)piv * n + k]);
      std::swap(b[c], b[piv]);
    }
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

bool chol_solve(std::vector<double>& A, int n, std::vector<double>& b) {
  for (int c = 0; c < n; ++c) {
    double d = A[(size_t)c * n + c];
    for (int k = 0; k < c; ++k) d -= A[(size_t)c * n + k] * A[(size_t)c * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[(size_t)c * n + c] = d;
    for (int r = c + 1; r < n; ++r) {
      double s = A[(size_t)r * n + c];
      for (int k = 0; k < c; ++k) s -= A[(size_t)r * n + k] * A[(size_t)c * n + k];
      A[(size_t)r * n + c] = s / d;
    }
  }
  for (int r = 0; r < n; ++r) {
    double s = b[r];
    for (int k = 0; k < r; ++k) s -= A[(size_t)r * n + k] * b[k];
    b[r] = s / A[(size_t)r * n + r];
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[(size_t)k * n + r] * b[k];
    b[r] = s / A[(size_t)r * n + r];
  }
  return true;
}

inline bool inv3_sym(const double* V, double* I) {
  const double a = V[0], b = V[1], c = V[2], d = V[4], e = V[5], f = V[8];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  if (det == 0.0) return false;
  const double id = 1.0 / det;
  I[0] = c00 * id; I[1] = c01 * id; I[2] = c02 * id;
  I[3] = I[1];     I[4] = (a * f - c * c) * id; I[5] = (b * c - a * e) * id;
  I[6] = I[2];     I[7] = I[5];     I[8] = (a * d - b * b) * id;
  return true;
}

// Block normal equations of one linearisation at `params` over the problem's observations.
// U [K*36], gp [6K], V [M*9], gl [3M], W per observation [N*18] (6x3 row-major, zero for fixed KF)
void build_blocks(const Problem& P, const std::vector<double>& params, std::vector<double>& U,
                  std::vector<double>& gp, std::vector<double>& V, std::vector<double>& gl,
                  std::vector<double>& W, double* chi2) {
  U.assign((size_t)P.K * 36, 0.0); gp.assign((size_t)P.K * 6, 0.0);
  V.assign((size_t)P.M * 9, 0.0); gl.assign((size_t)P.M * 3, 0.0);
  W.assign((size_t)P.N * 18, 0.0);
  double c2 = 0.0;
  ObsTerms t;
  for (int i = 0; i < P.N; ++i) {
    const orbx_ba_obs& o = P.obs[i];
    obs_terms(P.cam, P.cfg.huber_threshold, obs_pose(P, params, o),
              &params[6 * (size_t)P.K + 3 * (size_t)o.mp_idx], o.u, o.v, t, P.global_mode);
    c2 += t.r[0] * t.r[0];
    c2 += t.r[1] * t.r[1];
    double* Vj = &V[9 * (size_t)o.mp_idx];
    double* glj = &gl[3 * (size_t)o.mp_idx];
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) Vj[a * 3 + b] += t.B[a] * t.B[b] + t.B[3 + a] * t.B[3 + b];
      glj[a] += t.B[a] * t.r[0] + t.B[3 + a] * t.r[1];
    }
    if (o.kf_idx >= 0) {
      double* Uk = &U[36 * (size_t)o.kf_idx];
      double* gk = &gp[6 * (size_t)o.kf_idx];
      double* Wi = &W[18 * (size_t)i];
      for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) Uk[a * 6 + b] += t.A[a] * t.A[b] + t.A[6 + a] * t.A[6 + b];
        gk[a] += t.A[a] * t.r[0] + t.A[6 + a] * t.r[1];
        for (int b = 0; b < 3; ++b) Wi[a * 3 + b] = t.A[a] * t.B[b] + t.A[6 + a] * t.B[3 + b];
      }
    }
  }
  *chi2 = c2;
}

// Sred += sum_j W_j Vinv_j W_j^T ; bred += sum_j W_j Vinv_j gl_j ; Vinv out [M*9]
bool schur_terms(const Problem& P, double lambda, const std::vector<double>& V,
                 const std::vector<double>& gl, const std::vector<double>& W,
                 std::vector<double>& Vinv, std::vector<double>& Sred, std::vector<double>& bred) {
  const int n = 6 * P.K;
  Vinv.assign((size_t)P.M * 9, 0.0);
  Sred.assign((size_t)n * n, 0.0);
  bred.assign((size_t)n, 0.0);
  for (int j = 0; j < P.M; ++j) {
    double Vd[9];
    memcpy(Vd, &V[9 * (size_t)j], sizeof(Vd));
    for (int a = 0; a < 3; ++a) Vd[a * 3 + a] += lambda * std::fmax(Vd[a * 3 + a], 1e-6);
    if (!inv3_sym(Vd, &Vinv[9 * (size_t)j])) return false;
  }
  // group observations by point
  std::vector<std::vector<int>> by_pt(P.M);
  for (int i = 0; i < P.N; ++i)
    if (P.obs[i].kf_idx >= 0) by_pt[P.obs[i].mp_idx].push_back(i);
  for (int j = 0; j < P.M; ++j) {
    const double* Vi = &Vinv[9 * (size_t)j];
    const double* g = &gl[3 * (size_t)j];
    for (int ia : by_pt[j]) {
      const int ka = P.obs[ia].kf_idx;
      const double* Wa = &W[18 * (size_t)ia];
      double Y[18];  // W_a Vinv (6x3)
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 3; ++b)
          Y[a * 3 + b] = Wa[a * 3 + 0] * Vi[0 * 3 + b] + Wa[a * 3 + 1] * Vi[1 * 3 + b] + Wa[a * 3 + 2] * Vi[2 * 3 + b];
      for (int a = 0; a < 6; ++a)
        bred[6 * (size_t)ka + a] += Y[a * 3 + 0] * g[0] + Y[a * 3 + 1] * g[1] + Y[a * 3 + 2] * g[2];
      for (int ib : by_pt[j]) {
        const int kb = P.obs[ib].kf_idx;
        const double* Wb = &W[18 * (size_t)ib];
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b)
            Sred[(size_t)(6 * ka + a) * n + (6 * kb + b)] +=
                Y[a * 3 + 0] * Wb[b * 3 + 0] + Y[a * 3 + 1] * Wb[b * 3 + 1] + Y[a * 3 + 2] * Wb[b * 3 + 2];
      }
    }
  }
  return true;
}

void init_problem(Problem& P, const orbx_camera* cam, const orbx_ba_config* cfg, int K, int F,
                  const double* fixed_poses_cw, int M, int N, const orbx_ba_obs* obs) {
  P.cam = *cam; P.cfg = *cfg; P.K = K; P.F = F; P.M = M; P.N = N; P.obs = obs;
  P.fixed.resize(F);
  for (int f = 0; f < F; ++f) P.fixed[f] = pose_from7(fixed_poses_cw + 7 * (size_t)f);
}

void init_params(const Problem& P, const double* poses_cw, const double* points, std::vector<double>& params) {
  params.assign(6 * (size_t)P.K + 3 * (size_t)P.M, 0.0);
  for (int k = 0; k < P.K; ++k) se3_to_params(pose_from7(poses_cw + 7 * (size_t)k), &params[6 * (size_t)k]);
  for (int j = 0; j < 3 * P.M; ++j) params[6 * (size_t)P.K + j] = points[j];
}

void write_results(const Problem& P, const std::vector<double>& params, double* points, double* poses_wc_out) {
  for (int k = 0; k < P.K; ++k) pose_to7(se3_inverse(se3_from_params(&params[6 * (size_t)k])), poses_wc_out + 7 * (size_t)k);
  for (int j = 0; j < 3 * P.M; ++j) points[j] = params[6 * (size_t)P.K + j];
}

template <class StepFn>
int lm_loop(const Problem& P, std::vector<double>& params, int stop_after, int* iterations,
            double* initial_error, double* final_error, double* trace, StepFn step) {
  const size_t np = params.size();
  const double nres = 2.0 * (double)P.N;
  if (np == 0 || P.N == 0) return ORBX_ERR_EMPTY;               // :923-925
  *initial_error = std::sqrt(residual_sq(P, params)) / std::sqrt(nres);   // :1000-1001
  int iters = 0;
  double lambda = 1e-3;                                          // :1006-1010
  for (int iter = 0; iter < P.cfg.max_iterations; ++iter) {      // :1012
    if (stop_after >= 0 && iter >= stop_after) break;            // should_stop(), :1013
    iters = iter + 1;                                            // :1017
    std::vector<double> delta;
    double cur_sq = 0.0, gnorm = 0.0;
    const int st = step(params, lambda, delta, &cur_sq, &gnorm);  // 0 ok, 1 grad-converged, 2 solve failed
    if (trace) { trace[4 * iter + 0] = cur_sq; trace[4 * iter + 1] = gnorm; trace[4 * iter + 2] = 0; trace[4 * iter + 3] = 0; }
    if (st != 0) break;                                          // :1027-1029, :1036-1039
    const double dn = vec_norm(delta);
    if (trace) trace[4 * iter + 2] = dn;
    if (dn < P.cfg.param_tolerance * (vec_norm(params) + P.cfg.param_tolerance)) break;  // :1041-1044
    std::vector<double> trial(np);
    for (size_t i = 0; i < np; ++i) trial[i] = params[i] + delta[i];   // :1046
    const double trial_sq = residual_sq(P, trial);                     // :1047-1048
    if (trace) trace[4 * iter + 3] = trial_sq;
    if (trial_sq < cur_sq) {                                     // :1050-1055
      params.swap(trial);
      lambda = std::fmax(lambda * 0.1, 1e-10);
    } else {
      lambda = std::fmin(lambda * 10.0, 1e10);
    }
  }
  *iterations = iters;
  *final_error = std::sqrt(residual_sq(P, params)) / std::sqrt(nres);   // :1059-1060
  return ORBX_OK;
}

}  // namespace

extern "C" {

void oracle_se3_to_params(const double* pose7, double* params6) { se3_to_params(pose_from7(pose7), params6); }
void oracle_se3_from_params(const double* params6, double* pose7) { pose_to7(se3_from_params(params6), pose7); }
void oracle_se3_inverse(const double* pose7, double* out7) { pose_to7(se3_inverse(pose_from7(pose7)), out7); }

void oracle_ba_obs_terms(const orbx_camera* cam, double huber, const double* pose7, const double* X,
                         double u, double v, double* e2, double* sqrt_w, double* r2,
                         double* Jpose12, double* Jpoint6) {
  ObsTerms t;
  obs_terms(*cam, huber, pose_from7(pose7), X, u, v, t);
  e2[0] = t.e[0]; e2[1] = t.e[1]; *sqrt_w = t.sw; r2[0] = t.r[0]; r2[1] = t.r[1];
  memcpy(Jpose12, t.A, sizeof(t.A));
  memcpy(Jpoint6, t.B, sizeof(t.B));
}

static int ba_solve_dense_impl(bool global_mode, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                          const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                          double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error,
                          double* final_error, double* trace) {
  Problem P;
  init_problem(P, cam, cfg, K, F, fixed_poses_cw, M, N, obs);
  P.global_mode = global_mode;
  std::vector<double> params;
  init_params(P, poses_cw, points, params);
  const int np = (int)params.size();
  auto step = [&](const std::vector<double>& p, double lambda, std::vector<double>& delta,
                  double* cur_sq, double* gnorm) -> int {
    // compute_residuals + compute_jacobian (:1019-1020): dense J (2N x P), row-major here
    const int nr = 2 * P.N;
    std::vector<double> r((size_t)nr), J((size_t)nr * np, 0.0);
    ObsTerms t;
    for (int i = 0; i < P.N; ++i) {
      const orbx_ba_obs& o = P.obs[i];
      obs_terms(P.cam, P.cfg.huber_threshold, obs_pose(P, p, o), &p[6 * (size_t)P.K + 3 * (size_t)o.mp_idx], o.u, o.v, t, P.global_mode);
      r[2 * i] = t.r[0]; r[2 * i + 1] = t.r[1];
      for (int row = 0; row < 2; ++row) {
        double* Jr = &J[(size_t)(2 * i + row) * np];
        if (o.kf_idx >= 0) for (int c = 0; c < 6; ++c) Jr[6 * o.kf_idx + c] = t.A[row * 6 + c];
        for (int c = 0; c < 3; ++c) Jr[6 * P.K + 3 * o.mp_idx + c] = t.B[row * 3 + c];
      }
    }
    double s = 0.0;
    for (double x : r) s += x * x;
    *cur_sq = s;                                                   // :1022
    std::vector<double> g((size_t)np, 0.0), H((size_t)np * np, 0.0);   // :1024-1025
    for (int i = 0; i < nr; ++i) {
      const double* Jr = &J[(size_t)i * np];
      // exploit row sparsity only to skip exact zeros of the dense product
      int nz[9], nnz = 0;
      for (int c = 0; c < np && nnz < 9; ++c) if (Jr[c] != 0.0) nz[nnz++] = c;
      bool overflow = false;
      if (nnz == 9) for (int c = nz[8] + 1; c < np; ++c) if (Jr[c] != 0.0) { overflow = true; break; }
      if (!overflow) {
        for (int a = 0; a < nnz; ++a) {
          g[nz[a]] += Jr[nz[a]] * r[i];
          for (int b = 0; b < nnz; ++b) H[(size_t)nz[a] * np + nz[b]] += Jr[nz[a]] * Jr[nz[b]];
        }
      } else {
        for (int a = 0; a < np; ++a) {
          if (Jr[a] == 0.0) continue;
          g[a] += Jr[a] * r[i];
          for (int b = 0; b < np; ++b) H[(size_t)a * np + b] += Jr[a] * Jr[b];
        }
      }
    }
    *gnorm = vec_norm(g);
    if (*gnorm < P.cfg.gradient_tolerance) return 1;               // :1027-1029
    for (int i = 0; i < np; ++i) H[(size_t)i * np + i] += lambda * std::fmax(H[(size_t)i * np + i], 1e-6);  // :1031-1034
    delta.resize(np);
    for (int i = 0; i < np; ++i) delta[i] = -g[i];
    if (!lu_solve(H, np, delta)) return 2;                         // :1036-1039
    return 0;
  };
  const int rc = lm_loop(P, params, stop_after, iterations, initial_error, final_error, trace, step);
  if (rc != ORBX_OK) return rc;
  write_results(P, params, points, poses_wc_out);
  return ORBX_OK;
}

static int ba_solve_schur_impl(bool global_mode, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                          const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                          double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error,
                          double* final_error, double* trace) {
  Problem P;
  init_problem(P, cam, cfg, K, F, fixed_poses_cw, M, N, obs);
  P.global_mode = global_mode;
  std::vector<double> params;
  init_params(P, poses_cw, points, params);
  const int n = 6 * K;
  auto step = [&](const std::vector<double>& p, double lambda, std::vector<double>& delta,
                  double* cur_sq, double* gnorm) -> int {
    std::vector<double> U, gp, V, gl, W, Vinv, Sred, bred;
    build_blocks(P, p, U, gp, V, gl, W, cur_sq);
    double gs = 0.0;
    for (double x : gp) gs += x * x;
    for (double x : gl) gs += x * x;
    *gnorm = std::sqrt(gs);
    if (*gnorm < P.cfg.gradient_tolerance) return 1;
    if (!schur_terms(P, lambda, V, gl, W, Vinv, Sred, bred)) return 2;
    delta.assign(p.size(), 0.0);
    std::vector<double> S((size_t)n * n, 0.0), b((size_t)n, 0.0);
    for (int k = 0; k < K; ++k)
      for (int a = 0; a < 6; ++a)
        for (int c = 0; c < 6; ++c) {
          double u = U[36 * (size_t)k + a * 6 + c];
          if (a == c) u += lambda * std::fmax(u, 1e-6);
          S[(size_t)(6 * k + a) * n + (6 * k + c)] = u;
        }
    for (size_t i = 0; i < S.size(); ++i) S[i] -= Sred[i];
    for (int i = 0; i < n; ++i) b[i] = -gp[i] + bred[i];
    if (n > 0 && !chol_solve(S, n, b)) return 2;
    for (int i = 0; i < n; ++i) delta[i] = b[i];
    // back-substitution: dl_j = Vinv_j (-gl_j - sum_k W_kj^T dp_k)
    std::vector<double> rhs((size_t)3 * M);
    for (int j = 0; j < 3 * M; ++j) rhs[j] = -gl[j];
    for (int i = 0; i < P.N; ++i) {
      const orbx_ba_obs& o = P.obs[i];
      if (o.kf_idx < 0) continue;
      const double* Wi = &W[18 * (size_t)i];
      const double* dp = &delta[6 * (size_t)o.kf_idx];
      for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        for (int a = 0; a < 6; ++a) s += Wi[a * 3 + c] * dp[a];
        rhs[3 * (size_t)o.mp_idx + c] -= s;
      }
    }
    for (int j = 0; j < M; ++j) {
      const double* Vi = &Vinv[9 * (size_t)j];
      for (int a = 0; a < 3; ++a)
        delta[(size_t)n + 3 * j + a] = Vi[a * 3 + 0] * rhs[3 * j] + Vi[a * 3 + 1] * rhs[3 * j + 1] + Vi[a * 3 + 2] * rhs[3 * j + 2];
    }
    return 0;
  };
  const int rc = lm_loop(P, params, stop_after, iterations, initial_error, final_error, trace, step);
  if (rc != ORBX_OK) return rc;
  write_results(P, params, points, poses_wc_out);
  return ORBX_OK;
}

int oracle_ba_reduced_system(const orbx_camera* cam, const orbx_ba_config* cfg, double lambda,
                             int K, const double* params_pose, int F, const double* fixed_poses_cw,
                             int M, const double* points, int N, const orbx_ba_obs* obs, double* U,
                             double* gp, double* Sred, double* bred, double* chi2) {
  Problem P;
  init_problem(P, cam, cfg, K, F, fixed_poses_cw, M, N, obs);
  std::vector<double> params(6 * (size_t)K + 3 * (size_t)M);
  memcpy(params.data(), params_pose, sizeof(double) * 6 * (size_t)K);
  memcpy(params.data() + 6 * (size_t)K, points, sizeof(double) * 3 * (size_t)M);
  std::vector<double> Uv, gpv, V, gl, W, Vinv, S, b;
  build_blocks(P, params, Uv, gpv, V, gl, W, chi2);
  if (!schur_terms(P, lambda, V, gl, W, Vinv, S, b)) return ORBX_ERR_NUMERIC;
  memcpy(U, Uv.data(), sizeof(double) * Uv.size());
  memcpy(gp, gpv.data(), sizeof(double) * gpv.size());
  memcpy(Sred, S.data(), sizeof(double) * S.size());
  memcpy(bred, b.data(), sizeof(double) * b.size());
  return ORBX_OK;
}

// solve_visual_ba (local_ba_lm.rs:912-1098) and solve_global_ba (global_ba.rs:184-418): same LM loop, the global one
// drops the Jacobian rows of observations behind the camera.
int oracle_ba_solve_dense(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                          const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error, double* final_error, double* trace) {
  return ba_solve_dense_impl(false, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, stop_after, poses_wc_out, iterations,
                           initial_error, final_error, trace);
}
int oracle_ba_solve_schur(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                          const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error, double* final_error, double* trace) {
  return ba_solve_schur_impl(false, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, stop_after, poses_wc_out, iterations,
                           initial_error, final_error, trace);
}
int oracle_global_ba_solve_dense(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                          const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error, double* final_error, double* trace) {
  return ba_solve_dense_impl(true, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, stop_after, poses_wc_out, iterations,
                           initial_error, final_error, trace);
}
int oracle_global_ba_solve_schur(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                          const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error, double* final_error, double* trace) {
  return ba_solve_schur_impl(true, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, stop_after, poses_wc_out, iterations,
                           initial_error, final_error, trace);
}

}  // extern "C"
