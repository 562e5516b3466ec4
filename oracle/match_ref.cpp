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

}  // extern "C"
