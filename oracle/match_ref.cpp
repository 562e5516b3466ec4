// match_ref.cpp — CPU restatement of the reference's descriptor matching.  TEST INFRASTRUCTURE
// ONLY (see oracle.h).  Build with -ffp-contract=off: every f32 expression below is meant to be
// evaluated one IEEE operation at a time, exactly as rustc emits it for the reference.
//
// Follows (paths relative to the reference crate root):
//   src/tracking/frame/stereo.rs:80-161   StereoProcessor::match_features
//   src/tracking/frame/stereo.rs:166-175  descriptor_distance
//   src/tracking/frame/stereo.rs:186-216  triangulate
//   src/tracking/tracker.rs:1001-1010     BFMatcher(NORM_HAMMING, crossCheck=true).train_match
#include <cmath>
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

}  // extern "C"
