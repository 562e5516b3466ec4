// orb_ref.cpp — CPU specification of cv::ORB::detectAndCompute as configured by the reference
// (src/tracking/frame/stereo.rs:38-48, :68-78).  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// PARITY UNPINNED: the reference's extractor is OpenCV C++ reached through the `opencv` crate
// (Cargo.lock:5576-5577); its source is not under /root/reference, the OpenCV version is not
// pinned, and no reference test holds a vector for it.  This file restates the published
// algorithm of OpenCV 4.x features2d (orb.cpp, fast.cpp, fast_score.cpp, keypoint.cpp) and
// imgproc (resize INTER_LINEAR_EXACT, GaussianBlur fixed point) as listed in SURVEY.md
// Appendix A; where OpenCV's exact integer/rounding behaviour could not be verified offline the
// choice made here is marked "SPEC CHOICE".  The HIP kernels are held bit-exact to THIS file.
//
// Build with -ffp-contract=off: every float/double expression is one IEEE op at a time.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

// OpenCV bit_pattern_31_ (256 x {x0,y0,x1,y1}); data fixture tests/golden/orb_bit_pattern_31.txt
#include "orb_pattern_31.inc"

inline int cv_round_f(float v) { return (int)lrintf(v); }   // cvRound: round-half-to-even
inline int cv_round_d(double v) { return (int)lrint(v); }

// Appendix A.7 — half-widths of the 31-px intensity-centroid disc (749 pixels).
const int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

struct LevelImg {
  int w = 0, h = 0;
  float scale = 1.f;
  int quota = 0;
  std::vector<uint8_t> img;   // w*h, pitch = w
  std::vector<uint8_t> blur;  // w*h
};

// ---- A.2/A.3: level geometry and per-level feature quotas --------------------------------
int level_table(int W, int H, const orbx_orb_params& p, oracle_orb_levels& L) {
  if (p.n_levels < 1 || p.n_levels > 8) return -1;
  L.n_levels = p.n_levels;
  const double sf = (double)p.scale_factor;  // ORB_Impl stores the float argument in a double
  for (int l = 0; l < p.n_levels; ++l) {
    L.scale[l] = (float)std::pow(sf, (double)(l - p.first_level));
    L.w[l] = cv_round_f((float)W / L.scale[l]);
    L.h[l] = cv_round_f((float)H / L.scale[l]);
  }
  const float factor = (float)(1.0 / sf);
  float nper = (float)p.n_features * (1 - factor) /
               (1 - (float)std::pow((double)factor, (double)p.n_levels));
  int sum = 0;
  for (int l = 0; l < p.n_levels - 1; ++l) {
    L.quota[l] = cv_round_f(nper);
    sum += L.quota[l];
    nper *= factor;
  }
  L.quota[p.n_levels - 1] = std::max(p.n_features - sum, 0);
  return 0;
}

// ---- A.4: resize INTER_LINEAR_EXACT, u8, fixed point -----------------------------------------
// SPEC CHOICE (OpenCV resize.cpp bit-exact path as recalled): source coordinate
// f = (d + 0.5) * (src/dst) - 0.5 in double; i = floor(f); weights quantised to 8 fractional bits
// c1 = cvRound((f - i) * 256), c0 = 256 - c1; out-of-range taps replicate the edge pixel.
// Horizontal pass keeps 8.8 fixed point in 16 bits, vertical pass 16.16 in 32 bits, one final
// round-half-up: (v + 32768) >> 16.
struct ResizeTab {
  std::vector<int> ofs;
  std::vector<int> c1;
};
ResizeTab resize_tab(int src, int dst) {
  ResizeTab t;
  t.ofs.resize(dst);
  t.c1.resize(dst);
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; ++d) {
    const double f = scale * ((double)d + 0.5) - 0.5;
    int i = (int)std::floor(f);
    if (i < 0) { t.ofs[d] = 0; t.c1[d] = 0; }
    else if (i >= src - 1) { t.ofs[d] = src - 1; t.c1[d] = 0; }
    else { t.ofs[d] = i; t.c1[d] = cv_round_d((f - (double)i) * 256.0); }
  }
  return t;
}
void resize_linear_exact(const uint8_t* src, size_t sstride, int sw, int sh, uint8_t* dst, int dw,
                         int dh) {
  const ResizeTab tx = resize_tab(sw, dw), ty = resize_tab(sh, dh);
  for (int y = 0; y < dh; ++y) {
    const int y0 = ty.ofs[y], y1 = std::min(y0 + 1, sh - 1);
    const uint32_t cy1 = (uint32_t)ty.c1[y], cy0 = 256u - cy1;
    const uint8_t* r0 = src + sstride * (size_t)y0;
    const uint8_t* r1 = src + sstride * (size_t)y1;
    for (int x = 0; x < dw; ++x) {
      const int x0 = tx.ofs[x], x1 = std::min(x0 + 1, sw - 1);
      const uint32_t cx1 = (uint32_t)tx.c1[x], cx0 = 256u - cx1;
      const uint32_t h0 = cx0 * r0[x0] + cx1 * r0[x1];  // 8.8
      const uint32_t h1 = cx0 * r1[x0] + cx1 * r1[x1];
      const uint32_t v = cy0 * h0 + cy1 * h1;           // 16.16
      dst[(size_t)y * dw + x] = (uint8_t)((v + 32768u) >> 16);
    }
  }
}

// ---- A.8: GaussianBlur 7x7 sigma 2, u8 fixed point ---------------------------------------------
// SPEC CHOICE: taps exp(-k^2/8)/sum quantised to 8 fractional bits with error diffusion from the
// edge inwards, centre takes the remainder so the taps sum to 256: {18,34,48,56,48,34,18}.
// Horizontal pass keeps 8.8 in 16 bits, vertical pass 16.16, final (v + 32768) >> 16.
// Borders: BORDER_REFLECT_101 at the level-image edge (never sampled by a descriptor: keypoints
// sit >= 31 px inside, samples reach <= 18 px, the blur 3 px more).
const uint32_t kGauss[7] = {18, 34, 48, 56, 48, 34, 18};
inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
  return i;
}
void gaussian_blur_7x7(const uint8_t* src, int w, int h, uint8_t* dst) {
  std::vector<uint16_t> tmp((size_t)w * h);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      uint32_t s = 0;
      for (int k = -3; k <= 3; ++k) s += kGauss[k + 3] * src[(size_t)y * w + reflect101(x + k, w)];
      tmp[(size_t)y * w + x] = (uint16_t)s;  // <= 255*256
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      uint32_t s = 0;
      for (int k = -3; k <= 3; ++k) s += kGauss[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
      dst[(size_t)y * w + x] = (uint8_t)((s + 32768u) >> 16);
    }
}

// ---- A.5: FAST-9/16 score ------------------------------------------------------------------------
// Bresenham circle of radius 3 (x,y), the traversal order of OpenCV's fast_score.cpp.
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
// 0 if (x,y) is not a FAST-9 corner at threshold t; otherwise the corner score = the largest
// threshold at which it still is one (= max over the 16 arcs of 9 contiguous ring pixels of the
// minimum |centre - ring| with a common sign, minus 1).
inline int fast_score(const uint8_t* p, int pitch, int t) {
  const int v = p[0];
  int d[16];
  for (int k = 0; k < 16; ++k) d[k] = v - (int)p[kRing[k][1] * pitch + kRing[k][0]];
  int best = -256;
  for (int k = 0; k < 16; ++k) {
    int mn = 255, mx = -255;
    for (int i = 0; i < 9; ++i) {
      const int e = d[(k + i) & 15];
      mn = std::min(mn, e);
      mx = std::max(mx, e);
    }
    best = std::max(best, std::max(mn, -mx));  // all darker by >= mn, or all brighter by >= -mx
  }
  return best > t ? best - 1 : 0;
}

struct Cand { int x, y; float resp; };

// FAST + 3x3 NMS (strictly greater than all 8 neighbours) + runByImageBorder(edgeThreshold)
void fast_candidates(const LevelImg& L, const orbx_orb_params& p, std::vector<Cand>& out) {
  const int w = L.w, h = L.h, e = p.edge_threshold;
  out.clear();
  if (w < 7 || h < 7) return;
  std::vector<uint8_t> score((size_t)w * h, 0);
  for (int y = 3; y < h - 3; ++y)
    for (int x = 3; x < w - 3; ++x)
      score[(size_t)y * w + x] = (uint8_t)fast_score(&L.img[(size_t)y * w + x], w, p.fast_threshold);
  for (int y = e; y < h - e; ++y)
    for (int x = e; x < w - e; ++x) {
      const int s = score[(size_t)y * w + x];
      if (!s) continue;
      bool keep = true;
      for (int dy = -1; dy <= 1 && keep; ++dy)
        for (int dx = -1; dx <= 1; ++dx)
          if ((dx || dy) && s <= score[(size_t)(y + dy) * w + (x + dx)]) { keep = false; break; }
      if (keep) out.push_back({x, y, (float)s});
    }
}

// KeyPointsFilter::retainBest: keep the n best plus every point whose response equals the n-th.
// SPEC CHOICE (A.6): OpenCV's output order is whatever nth_element+partition leave; here the
// survivors are put in canonical order (response desc, y asc, x asc).
void retain_best(std::vector<Cand>& v, int n) {
  std::sort(v.begin(), v.end(), [](const Cand& a, const Cand& b) {
    if (a.resp != b.resp) return a.resp > b.resp;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
  });
  if (n < 0 || (int)v.size() <= n) return;
  if (n == 0) { v.clear(); return; }
  const float thr = v[n - 1].resp;
  size_t keep = n;
  while (keep < v.size() && v[keep].resp == thr) ++keep;
  v.resize(keep);
}

// ---- A.6: Harris response, blockSize 7, k 0.04 ---------------------------------------------------
float harris_response(const uint8_t* img, int pitch, int x0, int y0) {
  int a = 0, b = 0, c = 0;
  for (int i = -3; i <= 3; ++i)
    for (int j = -3; j <= 3; ++j) {
      const uint8_t* q = img + (y0 + i) * pitch + (x0 + j);
      const int Ix = (q[1] - q[-1]) * 2 + (q[-pitch + 1] - q[-pitch - 1]) + (q[pitch + 1] - q[pitch - 1]);
      const int Iy = (q[pitch] - q[-pitch]) * 2 + (q[pitch - 1] - q[-pitch - 1]) + (q[pitch + 1] - q[-pitch + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  const float scale = 1.f / (4 * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  const float t1 = fa * fb;
  const float t2 = fc * fc;
  const float s = fa + fb;
  const float t3 = 0.04f * s * s;       // (k*s)*s
  float r = (t1 - t2 - t3) * scale_sq_sq;
  if (r == 0.f) r = 0.f;                // canonical +0 so equality and ordering agree bitwise
  return r;
}

// ---- A.7: orientation ----------------------------------------------------------------------------
float ic_angle(const uint8_t* img, int pitch, int x, int y) {
  const uint8_t* c = img + y * pitch + x;
  int m01 = 0, m10 = 0;
  for (int u = -15; u <= 15; ++u) m10 += u * c[u];
  for (int v = 1; v <= 15; ++v) {
    int vsum = 0;
    const int d = kUmax[v];
    for (int u = -d; u <= d; ++u) {
      const int vp = c[u + v * pitch], vm = c[u - v * pitch];
      vsum += (vp - vm);
      m10 += u * (vp + vm);
    }
    m01 += v * vsum;
  }
  return oracle_fast_atan2((float)m01, (float)m10);
}

void build_pyramid(const uint8_t* img, size_t stride, int w, int h, const orbx_orb_params& p,
                   const oracle_orb_levels& T, std::vector<LevelImg>& pyr, int upto) {
  pyr.resize(T.n_levels);
  for (int l = 0; l < T.n_levels && l <= upto; ++l) {
    LevelImg& L = pyr[l];
    L.w = T.w[l]; L.h = T.h[l]; L.scale = T.scale[l]; L.quota = T.quota[l];
    L.img.resize((size_t)L.w * L.h);
    if (l == 0) {
      for (int y = 0; y < h; ++y) memcpy(&L.img[(size_t)y * w], img + stride * (size_t)y, (size_t)w);
    } else {
      // chain: level l from level l-1 (orb.cpp detectAndCompute pyramid loop)
      resize_linear_exact(pyr[l - 1].img.data(), (size_t)pyr[l - 1].w, pyr[l - 1].w, pyr[l - 1].h,
                          L.img.data(), L.w, L.h);
    }
  }
  (void)p;
}

bool params_supported(const orbx_orb_params& p) {
  return p.n_levels >= 1 && p.n_levels <= 8 && p.edge_threshold == 31 && p.first_level == 0 &&
         p.wta_k == 2 && p.score_type == 0 && p.patch_size == 31 && p.n_features >= 0 &&
         p.fast_threshold >= 1 && p.fast_threshold <= 254 && p.scale_factor > 1.0f;
}

}  // namespace

extern "C" {

int oracle_orb_level_table(int w, int h, const orbx_orb_params* p, oracle_orb_levels* out) {
  return level_table(w, h, *p, *out);
}

void oracle_orb_umax(int* umax16) { memcpy(umax16, kUmax, sizeof(kUmax)); }

// cv::fastAtan2 (degrees, [0,360)), the 7th-order polynomial of OpenCV core/mathfuncs_core.
// SPEC CHOICE: scalar form, float arithmetic, one IEEE op at a time.
float oracle_fast_atan2(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// cos/sin of a keypoint angle.  SPEC CHOICE (A.7/A.8): radians = angle * (float)(pi/180) in f32
// (orb.cpp computeOrbDescriptors), then cos/sin evaluated by ONE deterministic double-precision
// routine (quadrant reduction + Taylor polynomials, only IEEE mul/add) and rounded to f32, so
// host and device agree bit for bit; equals (float)cos(double) except for ~1e-9 of inputs.
void oracle_sincos_deg(float angle_deg, float* c_out, float* s_out) {
  const float rad_f = angle_deg * (float)(3.14159265358979323846 / 180.f);
  const double t = (double)rad_f;
  const int k = (int)(t * 0.63661977236758134308 + 0.5);       // t >= 0
  const double kd = (double)k;
  const double r = (t - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
  const double z = r * r;
  // sin(r) = r*(1 + z*S), cos(r) = 1 + z*C   (|r| <= pi/4 + eps)
  double S = 1.0 / 355687428096000.0;             // +1/17!
  S = S * z - 1.0 / 1307674368000.0;              // -1/15!
  S = S * z + 1.0 / 6227020800.0;                 // +1/13!
  S = S * z - 1.0 / 39916800.0;                   // -1/11!
  S = S * z + 1.0 / 362880.0;                     // +1/9!
  S = S * z - 1.0 / 5040.0;                       // -1/7!
  S = S * z + 1.0 / 120.0;                        // +1/5!
  S = S * z - 1.0 / 6.0;                          // -1/3!
  const double sr = r + r * (z * S);
  double C = 1.0 / 20922789888000.0;              // +1/16!
  C = C * z - 1.0 / 87178291200.0;                // -1/14!
  C = C * z + 1.0 / 479001600.0;                  // +1/12!
  C = C * z - 1.0 / 3628800.0;                    // -1/10!
  C = C * z + 1.0 / 40320.0;                      // +1/8!
  C = C * z - 1.0 / 720.0;                        // -1/6!
  C = C * z + 1.0 / 24.0;                         // +1/4!
  C = C * z - 0.5;                                // -1/2!
  const double cr = 1.0 + z * C;
  double cs, sn;
  switch (k & 3) {
    case 0: cs = cr; sn = sr; break;
    case 1: cs = -sr; sn = cr; break;
    case 2: cs = -cr; sn = -sr; break;
    default: cs = sr; sn = -cr; break;
  }
  *c_out = (float)cs;
  *s_out = (float)sn;
}

int oracle_orb_pyramid_level(const uint8_t* img, size_t stride, int w, int h,
                             const orbx_orb_params* p, int level, uint8_t* out) {
  if (!params_supported(*p)) return -1;
  oracle_orb_levels T;
  if (level_table(w, h, *p, T) || level < 0 || level >= T.n_levels) return -1;
  std::vector<LevelImg> pyr;
  build_pyramid(img, stride, w, h, *p, T, pyr, level);
  memcpy(out, pyr[level].img.data(), pyr[level].img.size());
  return 0;
}

int oracle_orb_blur_level(const uint8_t* img, size_t stride, int w, int h,
                          const orbx_orb_params* p, int level, uint8_t* out) {
  if (!params_supported(*p)) return -1;
  oracle_orb_levels T;
  if (level_table(w, h, *p, T) || level < 0 || level >= T.n_levels) return -1;
  std::vector<LevelImg> pyr;
  build_pyramid(img, stride, w, h, *p, T, pyr, level);
  gaussian_blur_7x7(pyr[level].img.data(), pyr[level].w, pyr[level].h, out);
  return 0;
}

int oracle_orb_fast_level(const uint8_t* img, size_t stride, int w, int h,
                          const orbx_orb_params* p, int level, uint32_t* out, int cap) {
  if (!params_supported(*p)) return -1;
  oracle_orb_levels T;
  if (level_table(w, h, *p, T) || level < 0 || level >= T.n_levels) return -1;
  std::vector<LevelImg> pyr;
  build_pyramid(img, stride, w, h, *p, T, pyr, level);
  std::vector<Cand> c;
  fast_candidates(pyr[level], *p, c);   // generated in (y,x) raster order
  if ((int)c.size() > cap) return -(int)c.size();
  for (size_t i = 0; i < c.size(); ++i)
    out[i] = ((uint32_t)c[i].resp << 24) | ((uint32_t)c[i].y << 12) | (uint32_t)c[i].x;
  return (int)c.size();
}

// detectAndCompute(image, noArray(), keypoints, descriptors, false): orb.cpp computeKeyPoints +
// computeOrbDescriptors, as configured at stereo.rs:38-48.
int oracle_orb_extract(const uint8_t* img, size_t stride, int w, int h, const orbx_orb_params* p,
                       orbx_keypoint* kp, uint8_t* desc, int cap) {
  if (!params_supported(*p)) return -1000000000;
  oracle_orb_levels T;
  if (level_table(w, h, *p, T)) return -1000000000;
  std::vector<LevelImg> pyr;
  build_pyramid(img, stride, w, h, *p, T, pyr, T.n_levels - 1);

  std::vector<orbx_keypoint> all;
  std::vector<std::pair<int, int>> lxy;  // level coords of each keypoint
  for (int l = 0; l < T.n_levels; ++l) {
    const LevelImg& L = pyr[l];
    std::vector<Cand> c;
    fast_candidates(L, *p, c);
    retain_best(c, 2 * L.quota);                       // HARRIS_SCORE keeps 2x by FAST score first
    for (Cand& k : c) k.resp = harris_response(L.img.data(), L.w, k.x, k.y);
    retain_best(c, L.quota);                           // then the quota by Harris response
    for (const Cand& k : c) {
      orbx_keypoint o;
      o.angle = ic_angle(L.img.data(), L.w, k.x, k.y);
      o.x = (float)k.x * L.scale;                      // pt *= scale (orb.cpp computeKeyPoints)
      o.y = (float)k.y * L.scale;
      o.size = (float)p->patch_size * L.scale;
      o.response = k.resp;
      o.octave = l;
      o.class_id = -1;
      all.push_back(o);
      lxy.push_back({k.x, k.y});
    }
  }
  const int n = (int)all.size();
  if (n > cap) return -n;

  for (int l = 0; l < T.n_levels; ++l) {
    pyr[l].blur.resize(pyr[l].img.size());
    gaussian_blur_7x7(pyr[l].img.data(), pyr[l].w, pyr[l].h, pyr[l].blur.data());
  }
  for (int i = 0; i < n; ++i) {
    kp[i] = all[i];
    const LevelImg& L = pyr[all[i].octave];
    float a, b;
    oracle_sincos_deg(all[i].angle, &a, &b);           // a = cos, b = sin
    const uint8_t* center = L.blur.data() + (size_t)lxy[i].second * L.w + lxy[i].first;
    uint8_t* d = desc + 32 * (size_t)i;
    for (int byte = 0; byte < 32; ++byte) {
      unsigned val = 0;
      for (int k = 0; k < 8; ++k) {
        const int* q = kPattern31[byte * 8 + k];
        const float x0 = (float)q[0] * a - (float)q[1] * b;
        const float y0 = (float)q[0] * b + (float)q[1] * a;
        const float x1 = (float)q[2] * a - (float)q[3] * b;
        const float y1 = (float)q[2] * b + (float)q[3] * a;
        const int t0 = center[cv_round_f(y0) * L.w + cv_round_f(x0)];
        const int t1 = center[cv_round_f(y1) * L.w + cv_round_f(x1)];
        val |= (unsigned)(t0 < t1) << k;
      }
      d[byte] = (uint8_t)val;
    }
  }
  return n;
}

}  // extern "C"
