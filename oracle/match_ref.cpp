// match_ref.cpp — CPU restatement of the reference's descriptor matching.  TEST INFRASTRUCTURE
// ONLY (see oracle.h).  Build with -ffp-contract=off: every f32 expression below is meant to be
// evaluated one IEEE operation at a time, exactly as rustc emits it for the reference.
//
// Follows (paths relative to the reference crate root):
//   src/tracking/frame/stereo.rs:80-161   StereoProcessor::match_features
//   src/tracking/frame/stereo.rs:166-175  descriptor_distance
//   src/tracking/frame/stereo.rs:186-216  triangulate
//   src/tracking/tracker.rs:1001-1010     BFMatcher(NORM_HAMMING, crossCheck=true).train_match
//   src/tracking/tracking_frame.rs:52-128 FeatureGrid::new / get_features_in_area
//   src/tracking/tracker.rs:880-923       track_local_map descriptor search (ratio rule)
//   src/tracking/tracker.rs:1126-1157     track_with_motion_model descriptor search
//   src/local_mapping/triangulation.rs:339-527, 661-705   search_for_triangulation (grid, epipolar gate, greedy)
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

extern "C" {

// stereo.rs:166-175 — sum over the 32 bytes of popcount(a[j] ^ b[j]).
uint32_t oracle_hamming256(const uint8_t* a, const uint8_t* b) {
  uint32_t d = 0;
  for (int j = 0; j < 32; ++j) d += (uint32_t)__builtin_popcount((unsigned)(a[j] ^ b[j]));
  return d;
}

void oracle_hamming_batch(const uint8_t* a, const uint8_t* b, int n, uint32_t* out) {
  for (int i = 0; i < n; ++i) out[i] = oracle_hamming256(a + 32 * (size_t)i, b + 32 * (size_t)i);
}

// stereo.rs:84-90 — MIN_DEPTH 0.1, MAX_DEPTH 40.0; f64 product/quotient, then `as f32`.
void oracle_disparity_bounds(const orbx_camera* cam, float* max_disp, float* min_disp) {
  *max_disp = (float)(cam->fx * cam->baseline / 0.1);
  *min_disp = (float)(cam->fx * cam->baseline / 40.0);
}

// stereo.rs:80-161 then :186-216.  matches: capacity nL.  points_cam [nL*3], has_point [nL].
int oracle_stereo_match(const orbx_camera* cam, const orbx_keypoint* kpL, const uint8_t* descL,
                        int nL, const orbx_keypoint* kpR, const uint8_t* descR, int nR,
                        orbx_dmatch* matches, double* points_cam, uint8_t* has_point) {
  const uint32_t TH_HIGH = 100;          // stereo.rs:10
  const float VERTICAL_MARGIN = 2.0f;    // stereo.rs:86
  float max_disparity, min_disparity;
  oracle_disparity_bounds(cam, &max_disparity, &min_disparity);

  int n_matches = 0;
  for (int li = 0; li < nL; ++li) {  // :95
    const float ul = kpL[li].x, vl = kpL[li].y;
    const float min_u = fmaxf(ul - max_disparity, 0.0f);                       // :100
    const float lim = ((float)nR * ul) / (float)nL;                           // :102
    const float max_u = fminf(ul - min_disparity, lim);                       // :101-102
    uint32_t best = TH_HIGH, second = TH_HIGH;                                // :104-106
    int best_idx = -1;
    const uint8_t* dl = descL + 32 * (size_t)li;
    for (int ri = 0; ri < nR; ++ri) {  // :112
      const float ur = kpR[ri].x, vr = kpR[ri].y;
      if (fabsf(vl - vr) > VERTICAL_MARGIN) continue;                         // :117
      if (ur < min_u || ur > max_u) continue;                                 // :122
      if (ul <= ur) continue;                                                 // :127
      const uint32_t d = oracle_hamming256(dl, descR + 32 * (size_t)ri);      // :132-133
      if (d < best) {                                                         // :135-141
        second = best;
        best = d;
        best_idx = ri;
      } else if (d < second) {
        second = d;
      }
    }
    if (best_idx >= 0) {  // :145-156
      if ((float)best < 0.9f * (float)second || second == TH_HIGH) {
        orbx_dmatch m;
        m.query_idx = li;
        m.train_idx = best_idx;
        m.img_idx = 0;
        m.distance = (float)best;
        matches[n_matches++] = m;
      }
    }
  }

  // triangulate, stereo.rs:186-216 (all f64; keypoint coordinates widened from f32, :178-183)
  if (has_point) memset(has_point, 0, (size_t)nL);
  if (points_cam && has_point) {
    for (int k = 0; k < n_matches; ++k) {
      const orbx_dmatch& m = matches[k];
      const double lx = (double)kpL[m.query_idx].x, ly = (double)kpL[m.query_idx].y;
      const double rx = (double)kpR[m.train_idx].x;
      const double disparity = lx - rx;                 // :204
      if (fabs(disparity) < 0.5) continue;              // :205-207
      const double z = cam->fx * cam->baseline / disparity;   // :208
      const double x = (lx - cam->cx) * z / cam->fx;          // :209
      const double y = (ly - cam->cy) * z / cam->fy;          // :210
      points_cam[3 * (size_t)m.query_idx + 0] = x;
      points_cam[3 * (size_t)m.query_idx + 1] = y;
      points_cam[3 * (size_t)m.query_idx + 2] = z;
      has_point[m.query_idx] = 1;
    }
  }
  return n_matches;
}

// tracker.rs:1001-1010 — cv::BFMatcher(NORM_HAMMING, crossCheck=true)::match(query, train).
// OpenCV semantics restated (source absent): fwd[i] = argmin_j d(q_i,t_j),
// bwd[j] = argmin_i d(q_i,t_j), first minimum in index order (spec choice: lowest index wins
// ties — UNPINNED), emit (i, fwd[i], d) iff bwd[fwd[i]] == i, ascending i.  out capacity nq.
int oracle_crosscheck_match(const uint8_t* q, int nq, const uint8_t* t, int nt, orbx_dmatch* out) {
  if (nq <= 0 || nt <= 0) return 0;
  std::vector<int> fwd(nq, -1), bwd(nt, -1);
  std::vector<uint32_t> fd(nq, 0xffffffffu), bd(nt, 0xffffffffu);
  for (int i = 0; i < nq; ++i)
    for (int j = 0; j < nt; ++j) {
      const uint32_t d = oracle_hamming256(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d < fd[i]) { fd[i] = d; fwd[i] = j; }
      if (d < bd[j]) { bd[j] = d; bwd[j] = i; }
    }
  int n = 0;
  for (int i = 0; i < nq; ++i) {
    if (fwd[i] >= 0 && bwd[fwd[i]] == i) {
      out[n].query_idx = i;
      out[n].train_idx = fwd[i];
      out[n].img_idx = 0;
      out[n].distance = (float)fd[i];
      ++n;
    }
  }
  return n;
}

// ---- guided matching -------------------------------------------------------------------------------------
// Rust float -> int casts saturate (and NaN -> 0); `as usize` of a negative i32 wraps.
static inline long long sat_usize(double v) { return (v != v || v <= 0.0) ? 0 : (v >= 9.0e18 ? (long long)9.0e18 : (long long)v); }
static inline int sat_i32(double v) {
  if (v != v) return 0;
  if (v <= -2147483648.0) return INT32_MIN;
  if (v >= 2147483647.0) return INT32_MAX;
  return (int)v;
}

// tracking_frame.rs:52-128 + the two search loops.  mode 0 = track_with_motion_model (:1126-1157): smallest
// distance < TH_HIGH, first candidate wins ties.  mode 1 = track_local_map (:880-923): best/second over the
// candidates, reject best > TH_HIGH, and when there is more than one candidate reject best > 0.75*second.
// out_idx[q] = matched keypoint index or -1; out_dist[q] = its distance.
void oracle_guided_match(const orbx_keypoint* kp, const uint8_t* desc, int n, double img_w, double img_h,
                         const double* q_uv, const uint8_t* q_desc, int nq, double radius, int mode,
                         int* out_idx, uint32_t* out_dist) {
  const int GC = 64, GR = 48;                                   // tracking_frame.rs:43-44
  const double winv = (double)GC / (img_w - 0.0), hinv = (double)GR / (img_h - 0.0);   // :58-59
  std::vector<std::vector<int>> cells((size_t)GC * GR);
  for (int i = 0; i < n; ++i) {                                 // :65-78
    const double x = (double)kp[i].x, y = (double)kp[i].y;
    long long cx = sat_usize((x - 0.0) * winv), cy = sat_usize((y - 0.0) * hinv);
    if (cx > GC - 1) cx = GC - 1;
    if (cy > GR - 1) cy = GR - 1;
    cells[(size_t)cy * GC + cx].push_back(i);
  }
  for (int q = 0; q < nq; ++q) {
    const double x = q_uv[2 * q], y = q_uv[2 * q + 1];
    // :107-117 — note `(max as usize).min(cols-1)`: a negative max wraps and clamps to the LAST cell
    const int mnx = sat_i32(std::floor((x - 0.0 - radius) * winv)), mxx = sat_i32(std::ceil((x - 0.0 + radius) * winv));
    const int mny = sat_i32(std::floor((y - 0.0 - radius) * hinv)), mxy = sat_i32(std::ceil((y - 0.0 + radius) * hinv));
    const long long x0 = mnx > 0 ? mnx : 0, y0 = mny > 0 ? mny : 0;
    const long long x1 = (mxx < 0 || mxx > GC - 1) ? GC - 1 : mxx, y1 = (mxy < 0 || mxy > GR - 1) ? GR - 1 : mxy;
    std::vector<int> cand;
    for (long long cy = y0; cy <= y1; ++cy)
      for (long long cx = x0; cx <= x1; ++cx)
        for (int i : cells[(size_t)cy * GC + cx]) cand.push_back(i);
    out_idx[q] = -1; out_dist[q] = 0;
    uint32_t best = 0xffffffffu, second = 0xffffffffu;
    int bi = -1;
    if (mode == 0) {
      for (int i : cand) {
        const uint32_t d = oracle_hamming256(q_desc + 32 * (size_t)q, desc + 32 * (size_t)i);
        if (d < best && d < 100) { best = d; bi = i; }          // tracker.rs:1146-1149
      }
      if (bi >= 0) { out_idx[q] = bi; out_dist[q] = best; }
    } else {
      if (cand.empty()) continue;                               // :884-886
      for (int i : cand) {
        const uint32_t d = oracle_hamming256(q_desc + 32 * (size_t)q, desc + 32 * (size_t)i);
        if (d < best) { second = best; best = d; bi = i; }     // :898-904
        else if (d < second) second = d;
      }
      if (best > 100) continue;                                 // :907-909
      if (cand.size() > 1 && (float)best > 0.75f * (float)second) continue;   // :911-915
      out_idx[q] = bi; out_dist[q] = best;
    }
  }
}

// ---- search_for_triangulation (src/local_mapping/triangulation.rs:401-527) ------------------------------------
namespace {
struct Q { double w, x, y, z; };
inline void q_rot(const Q& q, const double* v, double* o) {            // nalgebra UnitQuaternion * Vector3
  const double t[3] = {2.0 * (q.y * v[2] - q.z * v[1]), 2.0 * (q.z * v[0] - q.x * v[2]), 2.0 * (q.x * v[1] - q.y * v[0])};
  const double c[3] = {q.y * t[2] - q.z * t[1], q.z * t[0] - q.x * t[2], q.x * t[1] - q.y * t[0]};
  for (int i = 0; i < 3; ++i) o[i] = t[i] * q.w + c[i] + v[i];
}
inline Q q_mul(const Q& a, const Q& b) {                                // nalgebra Quaternion * Quaternion
  return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
           a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline void q_to_R(const Q& q, double* R) {
  const double w = q.w, i = q.x, j = q.y, k = q.z;
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  const double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
  R[0] = ww + ii - jj - kk; R[1] = ij - wk; R[2] = wj + ik;
  R[3] = wk + ij; R[4] = ww - ii + jj - kk; R[5] = jk - wi;
  R[6] = ik - wj; R[7] = wi + jk; R[8] = ww - ii - jj + kk;
}
inline void mat3_mul(const double* A, const double* B, double* C) {     // row-major, k accumulated in order
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j] + A[i * 3 + 2] * B[2 * 3 + j];
}
inline long long f32_as_usize(float v) { return (v != v || v <= 0.0f) ? 0 : (v >= 9.0e18f ? (long long)9.0e18 : (long long)v); }
}  // namespace

// epipole (x,y) and fundamental matrix F (row-major) as the reference computes them (:418-431, :661-683)
void oracle_triangulation_geometry(const orbx_camera* cam, const double* pose1_wc, const double* pose2_wc, double* epipole2,
                                   double* F9) {
  const Q q1{pose1_wc[0], pose1_wc[1], pose1_wc[2], pose1_wc[3]}, q2{pose2_wc[0], pose2_wc[1], pose2_wc[2], pose2_wc[3]};
  const double* t1 = pose1_wc + 4;
  const double* t2 = pose2_wc + 4;
  const Q q2i{q2.w, -q2.x, -q2.y, -q2.z};                               // pose2.inverse() (se3.rs:56-63)
  double r[3];
  q_rot(q2i, t2, r);
  const double t2i[3] = {-r[0], -r[1], -r[2]};
  double c1[3];
  q_rot(q2i, t1, c1);                                                   // pose2_inv.transform_point(c1_world) (:420-421)
  c1[0] += t2i[0]; c1[1] += t2i[1]; c1[2] += t2i[2];
  epipole2[0] = cam->fx * c1[0] / c1[2] + cam->cx;                      // :422-426
  epipole2[1] = cam->fy * c1[1] / c1[2] + cam->cy;
  double rt[3];
  q_rot(q2i, t1, rt);
  const double t12[3] = {t2i[0] - rt[0], t2i[1] - rt[1], t2i[2] - rt[2]};   // :429
  const Q q1i{q1.w, -q1.x, -q1.y, -q1.z};
  const Q r12 = q_mul(q2i, q1i);                                        // :430
  const double tsk[9] = {0.0, -t12[2], t12[1], t12[2], 0.0, -t12[0], -t12[1], t12[0], 0.0};   // :697-703
  double R[9], E[9], KiT[9], T[9];
  q_to_R(r12, R);
  mat3_mul(tsk, R, E);                                                  // :670-672
  const double Ki[9] = {1.0 / cam->fx, 0.0, -cam->cx / cam->fx, 0.0, 1.0 / cam->fy, -cam->cy / cam->fy, 0.0, 0.0, 1.0};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) KiT[i * 3 + j] = Ki[j * 3 + i];
  mat3_mul(KiT, E, T);                                                  // :681 (left to right)
  mat3_mul(T, Ki, F9);
}

// out: pairs (idx1, idx2) in ascending idx1; returns the count.  mp1/mp2: 1 = already has a map point;
// stereo1: 1 = points_cam1[idx1] is Some.  Poses are T_wc (pose.translation = camera centre, :418).
int oracle_search_for_triangulation(const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1, const uint8_t* mp1,
                                    const uint8_t* stereo1, int n1, const orbx_keypoint* kp2, const uint8_t* desc2,
                                    const uint8_t* mp2, int n2, const double* pose1_wc, const double* pose2_wc,
                                    unsigned max_dist, int* out_pairs) {
  double ep[2], F[9];
  oracle_triangulation_geometry(cam, pose1_wc, pose2_wc, ep, F);
  const float CELL = 32.0f;                                              // :339
  auto f64_as_u32 = [](double v) -> unsigned { return v > 0 ? (v >= 4294967295.0 ? 4294967295u : (unsigned)v) : 0u; };
  const unsigned iw2 = f64_as_u32(cam->cx * 2.0), ih2 = f64_as_u32(cam->cy * 2.0);      // :434-435
  long long cols = f32_as_usize(std::ceil((float)iw2 / CELL)), rows = f32_as_usize(std::ceil((float)ih2 / CELL));   // :437-438
  if (cols > 64) cols = 64;
  if (rows > 64) rows = 64;
  if (cols < 1 || rows < 1) return 0;
  std::vector<std::vector<int>> grid((size_t)(cols * rows));
  for (int i = 0; i < n2; ++i) {                                         // :356-361
    long long c = f32_as_usize(kp2[i].x / CELL), r = f32_as_usize(kp2[i].y / CELL);
    if (c > cols - 1) c = cols - 1;
    if (r > rows - 1) r = rows - 1;
    grid[(size_t)(r * cols + c)].push_back(i);
  }
  std::vector<uint8_t> matched2((size_t)n2, 0);
  const float radius = 100.0f;                                           // :442
  int n_out = 0;
  for (int i1 = 0; i1 < n1; ++i1) {
    if (mp1[i1]) continue;                                               // :449-452
    unsigned best = max_dist;
    int best2 = -1;
    const float x = kp1[i1].x, y = kp1[i1].y;
    const long long c0 = f32_as_usize(std::fmax(std::floor((x - radius) / CELL), 0.0f));   // :376-379
    long long c1 = f32_as_usize(std::ceil((x + radius) / CELL));
    const long long r0 = f32_as_usize(std::fmax(std::floor((y - radius) / CELL), 0.0f));
    long long r1 = f32_as_usize(std::ceil((y + radius) / CELL));
    if (c1 > cols - 1) c1 = cols - 1;
    if (r1 > rows - 1) r1 = rows - 1;
    // epipolar line of kp1 in image 2: l2 = F p1 (:685-687)
    const double p1[3] = {(double)x, (double)y, 1.0};
    const double l2[3] = {F[0] * p1[0] + F[1] * p1[1] + F[2] * p1[2], F[3] * p1[0] + F[4] * p1[1] + F[5] * p1[2],
                          F[6] * p1[0] + F[7] * p1[1] + F[8] * p1[2]};
    const double den = std::sqrt(l2[0] * l2[0] + l2[1] * l2[1]);        // :692
    for (long long r = r0; r <= r1; ++r)
      for (long long c = c0; c <= c1; ++c)
        for (int i2 : grid[(size_t)(r * cols + c)]) {
          if (matched2[i2] || mp2[i2]) continue;                         // :479-481
          const double x2 = (double)kp2[i2].x, y2 = (double)kp2[i2].y;
          if (!stereo1[i1]) {                                            // :489-497
            const double dx = ep[0] - x2, dy = ep[1] - y2;
            if (dx * dx + dy * dy < 100.0 * 1.0) continue;
          }
          if (den < 1e-10) continue;                                     // :694-696
          const double num = std::fabs(l2[0] * x2 + l2[1] * y2 + l2[2] * 1.0);   // :691
          const double dist_l = num / den;
          if (!(dist_l * dist_l < 3.84 * 1.0)) continue;                 // :698-702
          const unsigned d = oracle_hamming256(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
          if (d < best && d <= max_dist) { best2 = i2; best = d; }       // :516-519
        }
    if (best2 >= 0) {                                                    // :522-525
      out_pairs[2 * n_out] = i1; out_pairs[2 * n_out + 1] = best2; ++n_out;
      matched2[best2] = 1;
    }
  }
  return n_out;
}

}  // extern "C"

// ---- fuse_points_into_keyframes: the projection + descriptor search of every (map point, target keyframe) pair
// (src/local_mapping/search_in_neighbors.rs:273-343; KeyFrame::get_features_in_area, src/atlas/map/keyframe.rs:408-443).
// The map mutation that follows (:345-383) consumes these results on the host and does not change positions,
// descriptors or features, so the search of all pairs is a pure function of its inputs.
//   radius_scale = config.radius_factor * scale_factor.powi(num_levels - 1)  (:303, left to the caller so that the
//   caller's own powi rounding is what is used).  out_idx[p*T + t] = best feature index or -1.
void oracle_fuse_search(const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                        const double* kf_poses_wc, const int* kf_feat_offset, const orbx_keypoint* kps,
                        const uint8_t* descs, int T, double radius_scale, unsigned desc_threshold, int* out_idx,
                        uint32_t* out_dist) {
  for (int t = 0; t < T; ++t) {
    const double* pw = kf_poses_wc + 7 * (size_t)t;
    const Q q{pw[0], pw[1], pw[2], pw[3]};
    const Q qi{q.w, -q.x, -q.y, -q.z};                                   // se3.rs:57
    double r[3];
    q_rot(qi, pw + 4, r);
    const double ti[3] = {-r[0], -r[1], -r[2]};                            // se3.rs:58
    const int f0 = kf_feat_offset[t], f1 = kf_feat_offset[t + 1];
    for (int p = 0; p < P; ++p) {
      int best_idx = -1;
      uint32_t best = 0xffffffffu;                                         // :320
      double pc[3];
      q_rot(qi, positions + 3 * (size_t)p, pc);                            // transform_point (se3.rs:74-76)
      pc[0] += ti[0]; pc[1] += ti[1]; pc[2] += ti[2];
      if (!(pc[2] <= 0.0)) {                                               // :286
        const double u = cam->fx * pc[0] / pc[2] + cam->cx;               // :291-292
        const double v = cam->fy * pc[1] / pc[2] + cam->cy;
        const double width = cam->cx * 2.0, height = cam->cy * 2.0;       // :295-296
        if (!(u < 0.0 || u >= width || v < 0.0 || v >= height)) {
          const double radius = radius_scale * pc[2] / cam->fx;           // :303
          const double sr = std::max(std::min(radius, 50.0), 10.0);       // :304  radius.min(50.0).max(10.0)
          const double r2 = sr * sr;                                       // keyframe.rs:417
          for (int i = f0; i < f1; ++i) {
            const double du = (double)kps[i].x - u, dv = (double)kps[i].y - v;   // keyframe.rs:435-437
            if (!(du * du + dv * dv <= r2)) continue;
            const uint32_t d = oracle_hamming256(mp_desc + 32 * (size_t)p, descs + 32 * (size_t)i);
            if (d < best && d < desc_threshold) { best = d; best_idx = i - f0; }   // :336
          }
        }
      }
      out_idx[(size_t)p * T + t] = best_idx;
      out_dist[(size_t)p * T + t] = best_idx >= 0 ? best : 0u;
    }
  }
}

// ---- ORB vocabulary: DBoW2 text format, tree descent, FeatureVector level (src/vocabulary/mod.rs) -----------------
#include <array>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
namespace {
struct OVocab {
  int k = 0, l = 0;
  std::vector<uint32_t> parent;                      // [n_nodes], root = 0xffffffff (:152)
  std::vector<std::vector<uint32_t>> children;       // in file order (:196-198)
  std::vector<std::array<uint8_t, 32>> desc;
  std::vector<double> weight;
  std::vector<long long> word_id;                    // -1 = None
  int n_words = 0;
};
void ovocab_push(OVocab& v, uint32_t parent_id, bool is_leaf, const uint8_t* d, double w) {   // :183-200
  const uint32_t id = (uint32_t)v.parent.size();
  v.parent.push_back(parent_id);
  v.children.emplace_back();
  std::array<uint8_t, 32> a;
  memcpy(a.data(), d, 32);
  v.desc.push_back(a);
  v.weight.push_back(w);
  v.word_id.push_back(is_leaf ? (long long)v.n_words++ : -1);
  if ((size_t)parent_id < v.parent.size() - 1) v.children[parent_id].push_back(id);
}
void ovocab_root(OVocab& v) {
  v.parent.push_back(0xffffffffu);
  v.children.emplace_back();
  v.desc.push_back(std::array<uint8_t, 32>{});
  v.weight.push_back(0.0);
  v.word_id.push_back(-1);
}
bool parse_u64(const std::string& s, unsigned long long max, unsigned long long* out) {   // Rust's str::parse::<uN>
  if (s.empty()) return false;
  size_t i = s[0] == '+' ? 1 : 0;
  if (i >= s.size()) return false;
  unsigned long long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (unsigned)(s[i] - '0');
    if (v > max) return false;
  }
  *out = v;
  return true;
}
}  // namespace

extern "C" {

// load_from_text (:117-211).  Returns null where the reference returns Err.
void* oracle_vocab_load_text(const char* path) {
  std::ifstream f(path);
  if (!f) return nullptr;
  std::string line;
  if (!std::getline(f, line)) return nullptr;                      // :124-127
  std::vector<std::string> parts;
  auto split = [&](const std::string& s) { parts.clear(); std::istringstream is(s); std::string t; while (is >> t) parts.push_back(t); };
  split(line);
  if (parts.size() < 2) return nullptr;                            // :130-134
  unsigned long long k, l;
  if (!parse_u64(parts[0], ~0ull, &k) || !parse_u64(parts[1], ~0ull, &l)) return nullptr;
  OVocab* v = new OVocab();
  v->k = (int)k; v->l = (int)l;
  ovocab_root(*v);
  while (std::getline(f, line)) {
    split(line);
    if (parts.size() < 35) continue;                               // :155-157
    unsigned long long pid;
    if (!parse_u64(parts[0], 0xffffffffull, &pid)) { delete v; return nullptr; }
    uint8_t d[32];
    for (int i = 0; i < 32; ++i) {
      unsigned long long b;
      if (!parse_u64(parts[2 + i], 255, &b)) { delete v; return nullptr; }
      d[i] = (uint8_t)b;
    }
    char* end = nullptr;
    const double w = strtod(parts[34].c_str(), &end);
    if (end == parts[34].c_str() || *end != 0) { delete v; return nullptr; }
    ovocab_push(*v, (uint32_t)pid, parts[1] == "1", d, w);
  }
  return v;
}

void* oracle_vocab_from_arrays(int n_nodes, const uint32_t* parent, const uint8_t* is_leaf, const uint8_t* desc, const double* weight,
                               int k, int l) {
  OVocab* v = new OVocab();
  v->k = k; v->l = l;
  ovocab_root(*v);
  for (int i = 1; i < n_nodes; ++i) ovocab_push(*v, parent[i], is_leaf[i] != 0, desc + 32 * (size_t)i, weight[i]);
  return v;
}

void oracle_vocab_free(void* p) { delete (OVocab*)p; }

void oracle_vocab_info(void* p, int* k, int* l, int* n_nodes, int* n_words) {
  const OVocab* v = (const OVocab*)p;
  *k = v->k; *l = v->l; *n_nodes = (int)v->parent.size(); *n_words = v->n_words;
}

void oracle_vocab_arrays(void* p, uint32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight) {
  const OVocab* v = (const OVocab*)p;
  for (size_t i = 0; i < v->parent.size(); ++i) {
    parent[i] = v->parent[i]; is_leaf[i] = v->word_id[i] >= 0; memcpy(desc + 32 * i, v->desc[i].data(), 32); weight[i] = v->weight[i];
  }
}

// transform (:296-325) without the HashMap accumulation: per descriptor the word id, the leaf node, the node
// `levels_up` above it (:262-275) and the leaf's weight.
void oracle_bow_transform(void* p, const uint8_t* desc, int n, int levels_up, uint32_t* word, uint32_t* leaf, uint32_t* node,
                          double* weight) {
  const OVocab* v = (const OVocab*)p;
  for (int i = 0; i < n; ++i) {
    const uint8_t* d = desc + 32 * (size_t)i;
    size_t id = 0;
    while (!v->children[id].empty()) {                               // :231-245
      uint32_t best = v->children[id][0];
      uint32_t bd = oracle_hamming256(d, v->desc[best].data());
      for (size_t c = 1; c < v->children[id].size(); ++c) {
        const uint32_t ch = v->children[id][c];
        const uint32_t dd = oracle_hamming256(d, v->desc[ch].data());
        if (dd < bd) { bd = dd; best = ch; }
      }
      id = best;
    }
    word[i] = v->word_id[id] >= 0 ? (uint32_t)v->word_id[id] : 0u;   // :247 unwrap_or(0)
    leaf[i] = (uint32_t)id;
    weight[i] = v->weight[id];
    uint32_t nd = (uint32_t)id;
    for (int s = 0; s < levels_up; ++s) {                            // :265-272
      const uint32_t pa = v->parent[nd];
      if (pa == 0xffffffffu) break;
      nd = pa;
    }
    node[i] = nd;
  }
}

// search_for_triangulation_bow (src/local_mapping/triangulation.rs:541-658).  node1/node2: the FeatureVector key of each
// feature (0xffffffff = in no list).  The reference walks feat_vec1 in HashMap order; features in different nodes
// never compete (a feature of keyframe 2 belongs to one node), so the SET of pairs is order-independent and only
// the output order is hash-random there.  Here: ascending idx1.
int oracle_search_for_triangulation_bow(const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1, const uint8_t* mp1,
                                        const uint8_t* stereo1, const uint32_t* node1, int n1, const orbx_keypoint* kp2,
                                        const uint8_t* desc2, const uint8_t* mp2, const uint32_t* node2, int n2,
                                        const double* pose1_wc, const double* pose2_wc, unsigned max_dist, int* out_pairs) {
  double ep[2], F[9];
  oracle_triangulation_geometry(cam, pose1_wc, pose2_wc, ep, F);
  std::vector<uint8_t> matched2((size_t)std::max(n2, 1), 0);
  int n = 0;
  for (int i1 = 0; i1 < n1; ++i1) {
    if (node1[i1] == 0xffffffffu || mp1[i1]) continue;               // :583-586
    const double p0 = (double)kp1[i1].x, p1 = (double)kp1[i1].y;
    const double l0 = F[0] * p0 + F[1] * p1 + F[2] * 1.0, l1 = F[3] * p0 + F[4] * p1 + F[5] * 1.0, l2 = F[6] * p0 + F[7] * p1 + F[8] * 1.0;
    const double den = std::sqrt(l0 * l0 + l1 * l1);
    unsigned best = max_dist;
    int bi = -1;
    for (int i2 = 0; i2 < n2; ++i2) {                                // indices2 of the node are ascending (:309 pushes in row order)
      if (node2[i2] != node1[i1]) continue;
      if (matched2[i2] || mp2[i2]) continue;                         // :606-608
      const double x2 = (double)kp2[i2].x, y2 = (double)kp2[i2].y;
      if (!stereo1[i1]) {                                            // :616-623
        const double dx = ep[0] - x2, dy = ep[1] - y2;
        if (dx * dx + dy * dy < 100.0) continue;
      }
      if (den < 1e-10) continue;                                     // check_epipolar_constraint (:661-705)
      const double dl = std::fabs(l0 * x2 + l1 * y2 + l2 * 1.0) / den;
      if (!(dl * dl < 3.84)) continue;
      const unsigned d = oracle_hamming256(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
      if (d < best && d <= max_dist) { best = d; bi = i2; }          // :644-647
    }
    if (bi >= 0) { out_pairs[2 * n] = i1; out_pairs[2 * n + 1] = bi; ++n; matched2[bi] = 1; }
  }
  return n;
}

}  // extern "C"
