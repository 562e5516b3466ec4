// orbx.hpp — C++17 host-side mirror of the reference crate's interface for the hot path, header-only, on top
// of the C ABI (orbx.h).  The reference is compiled Rust and Rust is not in this toolchain, so this is the
// compiled-language form of the thin shim INTEGRATION.md describes: same names, argument meaning and error
// behaviour as the reference (file:line relative to the reference crate root):
//
//   orbx::CameraModel                       src/tracking/frame/camera.rs:3-10
//   orbx::FeatureSet / StereoFrame          src/tracking/frame/stereo.rs:15-29
//   orbx::StereoProcessor::create/process   stereo.rs:37-66     (`new` is a keyword in C++)
//   orbx::descriptor_distance               stereo.rs:166-175
//   orbx::bf_match_crosscheck               src/tracking/tracker.rs:1001-1010
//   orbx::FeatureGrid-guided search         src/tracking/tracking_frame.rs:52-128, tracker.rs:880-923, :1126-1157
//   orbx::SE3                               src/geometry/se3.rs:5-8 (unit quaternion w,x,y,z + translation)
//   orbx::LocalBAConfigLM                   src/optimizer/local_ba_lm.rs:96-119
//   orbx::VisualObservation/ProblemData/ResultData   local_ba_lm.rs:48-93
//   orbx::solve_visual_ba                   local_ba_lm.rs:912-1098
//
// Errors: the reference propagates `anyhow::Error` with `?` — here orbx::Error is thrown; where the
// reference returns `None` (solve_visual_ba) std::nullopt is returned.  Everything computes on the GPU.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "orbx.h"

namespace orbx {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("orbx error " + std::to_string(c) + ": " + m), code(c) {}
};

// ORB-SLAM3 matching thresholds, stereo.rs:10-12
constexpr uint32_t TH_HIGH = 100, TH_LOW = 50;
constexpr float NN_RATIO = 0.75f;

struct CameraModel {   // camera.rs:3-10
  double fx, fy, cx, cy, baseline;
  orbx_camera c() const { return orbx_camera{fx, fy, cx, cy, baseline}; }
};

using KeyPoint = orbx_keypoint;
using DMatch = orbx_dmatch;

struct FeatureSet {   // stereo.rs:15-19
  std::vector<KeyPoint> keypoints;
  std::vector<uint8_t> descriptors;   // keypoints.size() rows of 32 bytes
};

struct StereoFrame {   // stereo.rs:21-29
  FeatureSet left_features, right_features;
  std::vector<DMatch> matches_lr;
  std::vector<std::optional<std::array<double, 3>>> points_cam;   // per left keypoint
  uint64_t timestamp_ns = 0;
};

// RAII owner of one orbx_handle (one HIP stream on one device).  Not thread-safe, like `&mut self`.
class Handle {
 public:
  Handle(const CameraModel& cam, int n_features, int device = 0, int max_w = 1920, int max_h = 1080, int max_batch = 1) {
    if (orbx_abi_version() != ORBX_ABI_VERSION)       // the library on the loader's path was built from another orbx.h: struct strides differ
      throw Error(ORBX_ERR_INVALID, "liborbx_hip.so has ABI version " + std::to_string(orbx_abi_version()) + ", this caller was compiled against " + std::to_string(ORBX_ABI_VERSION));
    orbx_orb_params p;
    orbx_default_orb_params(n_features, &p);
    const orbx_camera c = cam.c();
    const int rc = orbx_create(&c, &p, device, max_w, max_h, max_batch, &h_);
    if (rc != ORBX_OK) throw Error(rc, orbx_last_error(nullptr));
    n_features_ = n_features;
  }
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;
  Handle(Handle&& o) noexcept : h_(o.h_), n_features_(o.n_features_) { o.h_ = nullptr; }
  ~Handle() { if (h_) orbx_destroy(h_); }
  orbx_handle* get() const { return h_; }
  int n_features() const { return n_features_; }
  void check(int rc) const { if (rc != ORBX_OK) throw Error(rc, orbx_last_error(h_)); }

 private:
  orbx_handle* h_ = nullptr;
  int n_features_ = 0;
};

class StereoProcessor {   // stereo.rs:31-66
 public:
  // = StereoProcessor::new(camera, n_features) -> Result<Self>
  static StereoProcessor create(const CameraModel& camera, int n_features, int device = 0) {
    return StereoProcessor(camera, n_features, device);
  }
  // = process(&mut self, left: &Mat, right: &Mat, timestamp_ns) -> Result<StereoFrame>; images are CV_8UC1 rows
  StereoFrame process(const uint8_t* left, size_t lstride, const uint8_t* right, size_t rstride, int w, int h,
                      uint64_t timestamp_ns) {
    const size_t cap = (size_t)handle_.n_features() + 2048;
    StereoFrame f;
    f.left_features.keypoints.resize(cap); f.right_features.keypoints.resize(cap);
    f.left_features.descriptors.resize(cap * 32); f.right_features.descriptors.resize(cap * 32);
    f.matches_lr.resize(cap);
    std::vector<double> pts(cap * 3);
    std::vector<uint8_t> has(cap);
    int nl = 0, nr = 0, nm = 0;
    handle_.check(orbx_process_stereo(handle_.get(), left, lstride, right, rstride, w, h, f.left_features.keypoints.data(),
                                      f.left_features.descriptors.data(), &nl, f.right_features.keypoints.data(),
                                      f.right_features.descriptors.data(), &nr, (int)cap, f.matches_lr.data(), &nm,
                                      pts.data(), has.data()));
    f.left_features.keypoints.resize(nl); f.left_features.descriptors.resize((size_t)nl * 32);
    f.right_features.keypoints.resize(nr); f.right_features.descriptors.resize((size_t)nr * 32);
    f.matches_lr.resize(nm);
    f.points_cam.resize(nl);
    for (int i = 0; i < nl; ++i)
      if (has[i]) f.points_cam[i] = std::array<double, 3>{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    f.timestamp_ns = timestamp_ns;
    return f;
  }
  Handle& handle() { return handle_; }

 private:
  StereoProcessor(const CameraModel& camera, int n_features, int device) : handle_(camera, n_features, device) {}
  Handle handle_;
};

// stereo.rs:166-175
inline uint32_t descriptor_distance(Handle& h, const uint8_t* desc1, const uint8_t* desc2) {
  uint32_t d = 0;
  h.check(orbx_hamming_batch(h.get(), desc1, desc2, 1, &d));
  return d;
}

// tracker.rs:1001-1010: BFMatcher::new(NORM_HAMMING, true).train_match(query, train)
inline std::vector<DMatch> bf_match_crosscheck(Handle& h, const std::vector<uint8_t>& query, const std::vector<uint8_t>& train) {
  const int nq = (int)(query.size() / 32), nt = (int)(train.size() / 32);
  std::vector<DMatch> out((size_t)std::max(nq, 1));
  int n = 0;
  h.check(orbx_hamming_match_crosscheck(h.get(), query.data(), nq, train.data(), nt, out.data(), &n));
  out.resize(n);
  return out;
}

// FeatureGrid::get_features_in_area + descriptor search.  mode 0 = track_with_motion_model, 1 = track_local_map.
// Returns per query the matched keypoint index or -1.
inline std::vector<int> guided_match(Handle& h, const FeatureSet& frame, double img_w, double img_h,
                                     const std::vector<std::array<double, 2>>& uv, const std::vector<uint8_t>& q_desc,
                                     double radius, int mode) {
  const int nq = (int)uv.size();
  std::vector<int> idx((size_t)std::max(nq, 1));
  std::vector<uint32_t> dist((size_t)std::max(nq, 1));
  h.check(orbx_guided_match(h.get(), frame.keypoints.data(), frame.descriptors.data(), (int)frame.keypoints.size(), img_w, img_h,
                            nq ? &uv[0][0] : nullptr, q_desc.data(), nq, radius, mode, idx.data(), dist.data()));
  idx.resize(nq);
  return idx;
}

struct SE3 {   // se3.rs:5-8; rotation as unit quaternion (w, x, y, z)
  std::array<double, 4> rotation{1, 0, 0, 0};
  std::array<double, 3> translation{0, 0, 0};
};

// triangulation.rs:401-527.  has_map_point1/2: the `map_point_ids[i].is_some()` flags; has_point_cam1: the
// `points_cam1[i].is_some()` flags.  Returns (idx1, idx2) pairs in ascending idx1.
inline std::vector<std::pair<size_t, size_t>> search_for_triangulation(
    Handle& h, const FeatureSet& f1, const std::vector<uint8_t>& has_map_point1, const std::vector<uint8_t>& has_point_cam1,
    const FeatureSet& f2, const std::vector<uint8_t>& has_map_point2, const SE3& pose1, const SE3& pose2,
    const CameraModel& camera, uint32_t max_dist) {
  const int n1 = (int)f1.keypoints.size(), n2 = (int)f2.keypoints.size();
  const double p1[7] = {pose1.rotation[0], pose1.rotation[1], pose1.rotation[2], pose1.rotation[3], pose1.translation[0], pose1.translation[1], pose1.translation[2]};
  const double p2[7] = {pose2.rotation[0], pose2.rotation[1], pose2.rotation[2], pose2.rotation[3], pose2.translation[0], pose2.translation[1], pose2.translation[2]};
  std::vector<int> pairs((size_t)std::max(n1, 1) * 2);
  int n = 0;
  const orbx_camera cam = camera.c();
  h.check(orbx_search_for_triangulation(h.get(), &cam, f1.keypoints.data(), f1.descriptors.data(), has_map_point1.data(),
                                        has_point_cam1.data(), n1, f2.keypoints.data(), f2.descriptors.data(),
                                        has_map_point2.data(), n2, p1, p2, max_dist, pairs.data(), &n));
  std::vector<std::pair<size_t, size_t>> out((size_t)n);
  for (int i = 0; i < n; ++i) out[(size_t)i] = {(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]};
  return out;
}

// The search of fuse_points_into_keyframes (search_in_neighbors.rs:273-343) for every (map point, keyframe) pair.
// keyframes[t] = (pose, features); returns best feature index per pair, row-major [point][keyframe], -1 = none.
inline std::vector<int> fuse_search(Handle& h, const std::vector<std::array<double, 3>>& positions, const std::vector<uint8_t>& mp_descriptors,
                                    const std::vector<std::pair<SE3, const FeatureSet*>>& keyframes, const CameraModel& camera,
                                    double radius_factor = 3.0, uint32_t desc_threshold = 50) {
  const int P = (int)positions.size(), T = (int)keyframes.size();
  std::vector<double> poses(7 * (size_t)T);
  std::vector<int> off((size_t)T + 1, 0);
  std::vector<KeyPoint> kps;
  std::vector<uint8_t> descs;
  for (int t = 0; t < T; ++t) {
    const SE3& s = keyframes[(size_t)t].first;
    for (int i = 0; i < 4; ++i) poses[7 * (size_t)t + i] = s.rotation[i];
    for (int i = 0; i < 3; ++i) poses[7 * (size_t)t + 4 + i] = s.translation[i];
    const FeatureSet& f = *keyframes[(size_t)t].second;
    kps.insert(kps.end(), f.keypoints.begin(), f.keypoints.end());
    descs.insert(descs.end(), f.descriptors.begin(), f.descriptors.end());
    off[(size_t)t + 1] = (int)kps.size();
  }
  double s7 = 1.2;                       // scale_factor.powi(num_levels - 1) with num_levels = 8 (:248-249, :303)
  { const double a2 = 1.2 * 1.2, a4 = a2 * a2; s7 = (1.2 * a2) * a4; }
  std::vector<int> idx((size_t)P * T, -1);
  std::vector<uint32_t> dist((size_t)P * T, 0);
  const orbx_camera cam = camera.c();
  h.check(orbx_fuse_search(h.get(), &cam, P ? positions[0].data() : nullptr, mp_descriptors.data(), P, poses.data(), off.data(),
                           kps.data(), descs.data(), T, radius_factor * s7, desc_threshold, idx.data(), dist.data()));
  return idx;
}

struct LocalBAConfigLM {   // local_ba_lm.rs:96-119, Default :109-119
  int max_iterations = 10;
  double param_tolerance = 1e-8, gradient_tolerance = 1e-8, huber_threshold = std::sqrt(5.991);
  int max_covisible_keyframes = 20;
};

using KeyFrameId = uint64_t;   // atlas/map/types.rs:9
using MapPointId = uint64_t;   // atlas/map/types.rs:29

struct VisualObservation {   // local_ba_lm.rs:68-78
  KeyFrameId kf_id;
  MapPointId mp_id;
  std::array<double, 2> observed_uv;
  bool is_kf_optimized;
};

struct VisualBAProblemData {   // local_ba_lm.rs:48-65; poses are T_cw
  std::unordered_map<KeyFrameId, SE3> local_kf_poses;
  std::unordered_map<MapPointId, std::array<double, 3>> local_mp_positions;
  std::unordered_map<KeyFrameId, SE3> fixed_kf_poses;
  KeyFrameId anchor_kf_id = 0;
  std::vector<VisualObservation> observations;
  std::vector<KeyFrameId> optimized_kf_ids;
  std::vector<MapPointId> mp_ids;
};

struct VisualBAResultData {   // local_ba_lm.rs:81-93; optimized_poses are T_wc
  std::unordered_map<KeyFrameId, SE3> optimized_poses;
  std::unordered_map<MapPointId, std::array<double, 3>> optimized_points;
  size_t iterations = 0;
  double initial_error = 0, final_error = 0;
};

// local_ba_lm.rs:912-1098.  The id -> index re-keying is the reference's own (:928-987).
inline std::optional<VisualBAResultData> solve_visual_ba(Handle& h, const VisualBAProblemData& problem, const CameraModel& camera,
                                                         const LocalBAConfigLM& config, const std::function<bool()>& should_stop) {
  std::unordered_map<KeyFrameId, int> kf_idx, fixed_idx;
  std::unordered_map<MapPointId, int> mp_idx;
  for (size_t i = 0; i < problem.optimized_kf_ids.size(); ++i) kf_idx[problem.optimized_kf_ids[i]] = (int)i;   // :928-933
  for (size_t i = 0; i < problem.mp_ids.size(); ++i) mp_idx[problem.mp_ids[i]] = (int)i;                       // :935-940
  std::vector<double> poses, fixed, points;
  auto push7 = [](std::vector<double>& v, const SE3& p) {
    v.insert(v.end(), p.rotation.begin(), p.rotation.end());
    v.insert(v.end(), p.translation.begin(), p.translation.end());
  };
  for (KeyFrameId id : problem.optimized_kf_ids) {                       // :966-977 (missing pose -> zero params = identity)
    auto it = problem.local_kf_poses.find(id);
    push7(poses, it != problem.local_kf_poses.end() ? it->second : SE3{});
  }
  for (const auto& kv : problem.fixed_kf_poses) { fixed_idx[kv.first] = (int)fixed_idx.size(); push7(fixed, kv.second); }
  for (MapPointId id : problem.mp_ids) {                                 // :980-987
    auto it = problem.local_mp_positions.find(id);
    const std::array<double, 3> p = it != problem.local_mp_positions.end() ? it->second : std::array<double, 3>{0, 0, 0};
    points.insert(points.end(), p.begin(), p.end());
  }
  std::vector<orbx_ba_obs> obs;
  // the reference's observed_uv are keypoint coordinates widened from f32 (:870-872): when every one of them is an f32 value — always, for
  // a problem collect_visual_ba_data made — the observations travel in the 16-byte form (orbx_ba_obs32: same result, half the upload)
  std::vector<orbx_ba_obs32> obs32;
  bool all_f32 = true;
  const int F = (int)problem.fixed_kf_poses.size();
  for (const VisualObservation& o : problem.observations) {              // :943-961
    auto m = mp_idx.find(o.mp_id);
    if (m == mp_idx.end()) continue;                                     // :947
    orbx_ba_obs b{};
    b.mp_idx = m->second; b.u = o.observed_uv[0]; b.v = o.observed_uv[1];
    auto k = o.is_kf_optimized ? kf_idx.find(o.kf_id) : kf_idx.end();
    if (k != kf_idx.end()) { b.kf_idx = k->second; b.fixed_idx = -1; }
    else { auto f = fixed_idx.find(o.kf_id); b.kf_idx = -1; b.fixed_idx = f != fixed_idx.end() ? f->second : -1; }   // :569 identity
    obs.push_back(b);
    const float uf = (float)b.u, vf = (float)b.v;
    if ((double)uf != b.u || (double)vf != b.v) all_f32 = false;
    obs32.push_back(orbx_ba_obs32{b.kf_idx >= 0 ? b.kf_idx : -1 - (b.fixed_idx >= 0 ? b.fixed_idx : F), b.mp_idx, uf, vf});
  }
  const int K = (int)problem.optimized_kf_ids.size(), M = (int)problem.mp_ids.size();
  all_f32 = all_f32 && orbx_ba_has_collective(h.get()) == 0;             // (the point-partitioned solve takes orbx_ba_obs)
  std::vector<double> out((size_t)std::max(K, 1) * 7);
  int it = 0;
  double e0 = 0, e1 = 0;
  const orbx_camera c = camera.c();
  const orbx_ba_config cfg{config.max_iterations, config.param_tolerance, config.gradient_tolerance, config.huber_threshold,
                           config.max_covisible_keyframes};
  auto tramp = [](void* user) -> int { return (*static_cast<const std::function<bool()>*>(user))() ? 1 : 0; };
  void* user = const_cast<std::function<bool()>*>(&should_stop);
  const int rc = all_f32 ? orbx_ba_solve_visual_obs32(h.get(), &c, &cfg, K, poses.data(), F, fixed.data(), M, points.data(), (int)obs32.size(),
                                                      obs32.data(), should_stop ? +tramp : nullptr, user, out.data(), &it, &e0, &e1)
                         : orbx_ba_solve_visual(h.get(), &c, &cfg, K, poses.data(), F, fixed.data(), M, points.data(), (int)obs.size(),
                                                obs.data(), should_stop ? +tramp : nullptr, user, out.data(), &it, &e0, &e1);
  if (rc != ORBX_OK) return std::nullopt;                                // :923-925 and every failure -> None
  VisualBAResultData r;
  for (int i = 0; i < K; ++i) {
    SE3 p;
    for (int q = 0; q < 4; ++q) p.rotation[q] = out[7 * (size_t)i + q];
    for (int q = 0; q < 3; ++q) p.translation[q] = out[7 * (size_t)i + 4 + q];
    r.optimized_poses[problem.optimized_kf_ids[i]] = p;                  // T_wc, :1076
  }
  for (int j = 0; j < M; ++j) r.optimized_points[problem.mp_ids[j]] = {points[3 * (size_t)j], points[3 * (size_t)j + 1], points[3 * (size_t)j + 2]};
  r.iterations = (size_t)it; r.initial_error = e0; r.final_error = e1;
  return r;
}


// ---- input side (src/io/euroc.rs) -------------------------------------------------------------------------------
struct StereoImagePair {   // euroc.rs:21-26 (Mat -> row-major u8)
  std::vector<uint8_t> left, right;
  int width = 0, height = 0;
  uint64_t timestamp_ns = 0;
};

class EurocDataset {   // euroc.rs:54-132, the image side
 public:
  explicit EurocDataset(const std::string& mav0_dir) {   // EurocDataset::new (:64-90)
    char err[512] = {0};
    const int rc = orbx_euroc_open(mav0_dir.c_str(), &d_, err, sizeof(err));
    if (rc != ORBX_OK) throw Error(rc, err);
    orbx_camera c;
    orbx_euroc_calibration(d_, &c, nullptr, &w_, &h_);
    camera_ = CameraModel{c.fx, c.fy, c.cx, c.cy, c.baseline};
  }
  EurocDataset(const EurocDataset&) = delete;
  EurocDataset& operator=(const EurocDataset&) = delete;
  ~EurocDataset() { orbx_euroc_close(d_); }
  size_t len() const { return (size_t)orbx_euroc_len(d_); }
  std::optional<uint64_t> frame_timestamp(size_t idx) const {
    uint64_t ts;
    if (orbx_euroc_frame_timestamp(d_, (int)idx, &ts) != ORBX_OK) return std::nullopt;
    return ts;
  }
  const CameraModel& camera() const { return camera_; }
  int width() const { return w_; }
  int height() const { return h_; }
  StereoImagePair stereo_pair(size_t idx) const {        // :100-132
    StereoImagePair p;
    p.width = w_; p.height = h_;
    std::vector<uint8_t> both(2 * (size_t)w_ * h_);
    const int rc = orbx_euroc_read_pairs(d_, (int)idx, 1, both.data(), 2);
    if (rc != ORBX_OK) throw Error(rc, orbx_euroc_last_error(d_));
    p.left.assign(both.begin(), both.begin() + (size_t)w_ * h_);
    p.right.assign(both.begin() + (size_t)w_ * h_, both.end());
    p.timestamp_ns = *frame_timestamp(idx);
    return p;
  }
  // `count` pairs from `first` into out[pair][2][h][w] (e.g. orbx_host_alloc memory), decoded by `threads` host threads
  void read_pairs(size_t first, size_t count, uint8_t* out, int threads) const {
    const int rc = orbx_euroc_read_pairs(d_, (int)first, (int)count, out, threads);
    if (rc != ORBX_OK) throw Error(rc, orbx_euroc_last_error(d_));
  }

 private:
  orbx_euroc* d_ = nullptr;
  CameraModel camera_{};
  int w_ = 0, h_ = 0;
};

// ---- ORB vocabulary (src/vocabulary/mod.rs) ----------------------------------------------------------------------
using BowVector = std::unordered_map<uint32_t, double>;                  // mod.rs:31
using FeatureVector = std::unordered_map<uint32_t, std::vector<size_t>>; // mod.rs:37

class OrbVocabulary {   // mod.rs:83-94; the tree lives in device memory, transform runs on the GPU
 public:
  static OrbVocabulary load_from_text(Handle& h, const std::string& path) {   // mod.rs:117-211; throws where the reference returns Err
    orbx_vocabulary* v = nullptr;
    h.check(orbx_vocab_load_text(h.get(), path.c_str(), &v));
    return OrbVocabulary(h, v);
  }
  OrbVocabulary(const OrbVocabulary&) = delete;
  OrbVocabulary& operator=(const OrbVocabulary&) = delete;
  OrbVocabulary(OrbVocabulary&& o) noexcept : h_(o.h_), v_(o.v_) { o.v_ = nullptr; }
  ~OrbVocabulary() { if (v_) orbx_vocab_destroy(v_); }
  std::pair<size_t, size_t> params() const { int k, l; orbx_vocab_info(v_, &k, &l, nullptr, nullptr); return {(size_t)k, (size_t)l}; }
  size_t num_words() const { int n; orbx_vocab_info(v_, nullptr, nullptr, nullptr, &n); return (size_t)n; }
  size_t num_nodes() const { int n; orbx_vocab_info(v_, nullptr, nullptr, &n, nullptr); return (size_t)n; }
  const orbx_vocabulary* get() const { return v_; }

  // mod.rs:296-325: descent, accumulation of the two maps and the L1 normalisation on the GPU (orbx_bow_vectors; the norm is summed
  // in ascending word id, the reference sums in HashMap order).
  std::pair<BowVector, FeatureVector> transform(const std::vector<uint8_t>& descriptors, size_t levels_up) const {
    const int n = (int)(descriptors.size() / 32);
    std::vector<uint32_t> bw((size_t)std::max(n, 1)), fn((size_t)std::max(n, 1));
    std::vector<double> bv((size_t)std::max(n, 1));
    std::vector<int> fs((size_t)n + 1), fi((size_t)std::max(n, 1));
    int nb = 0, nf = 0;
    h_->check(orbx_bow_vectors(h_->get(), v_, descriptors.data(), n, (int)levels_up, bw.data(), bv.data(), &nb, fn.data(), fs.data(), fi.data(), &nf));
    BowVector bow;
    FeatureVector feat;
    for (int i = 0; i < nb; ++i) bow[bw[(size_t)i]] = bv[(size_t)i];
    for (int i = 0; i < nf; ++i) {
      std::vector<size_t>& l = feat[fn[(size_t)i]];
      for (int t = fs[(size_t)i]; t < fs[(size_t)i + 1]; ++t) l.push_back((size_t)fi[(size_t)t]);
    }
    return {std::move(bow), std::move(feat)};
  }
  BowVector transform_bow_only(const std::vector<uint8_t>& descriptors) const { return transform(descriptors, 0).first; }   // mod.rs:330-356

  static double score(const BowVector& v1, const BowVector& v2) {          // mod.rs:357-374 (orbx_bow_score: terms in ascending word id)
    auto flat = [](const BowVector& v, std::vector<uint32_t>& k, std::vector<double>& w) {
      for (const auto& kv : v) k.push_back(kv.first);
      std::sort(k.begin(), k.end());
      for (uint32_t x : k) w.push_back(v.at(x));
    };
    std::vector<uint32_t> k1, k2; std::vector<double> w1, w2;
    flat(v1, k1, w1); flat(v2, k2, w2);
    double s = 0.0;
    if (orbx_bow_score(k1.data(), w1.data(), (int)k1.size(), k2.data(), w2.data(), (int)k2.size(), &s) != ORBX_OK) throw Error(ORBX_ERR_INVALID, "orbx_bow_score");
    return s;
  }

 private:
  OrbVocabulary(Handle& h, orbx_vocabulary* v) : h_(&h), v_(v) {}
  Handle* h_;
  orbx_vocabulary* v_;
};

// triangulation.rs:541-658: candidates restricted to the same FeatureVector node.  Pairs in ascending idx1.
inline std::vector<std::pair<size_t, size_t>> search_for_triangulation_bow(
    Handle& h, const FeatureVector& feat_vec1, const FeatureVector& feat_vec2, const FeatureSet& f1, const std::vector<uint8_t>& has_map_point1,
    const std::vector<uint8_t>& has_point_cam1, const FeatureSet& f2, const std::vector<uint8_t>& has_map_point2, const SE3& pose1,
    const SE3& pose2, const CameraModel& camera, uint32_t max_dist) {
  const int n1 = (int)f1.keypoints.size(), n2 = (int)f2.keypoints.size();
  std::vector<uint32_t> node1((size_t)n1, 0xffffffffu), node2((size_t)n2, 0xffffffffu);
  for (const auto& kv : feat_vec1) for (size_t i : kv.second) if (i < (size_t)n1) node1[i] = kv.first;
  for (const auto& kv : feat_vec2) for (size_t i : kv.second) if (i < (size_t)n2) node2[i] = kv.first;
  const double p1[7] = {pose1.rotation[0], pose1.rotation[1], pose1.rotation[2], pose1.rotation[3], pose1.translation[0], pose1.translation[1], pose1.translation[2]};
  const double p2[7] = {pose2.rotation[0], pose2.rotation[1], pose2.rotation[2], pose2.rotation[3], pose2.translation[0], pose2.translation[1], pose2.translation[2]};
  std::vector<int> pairs((size_t)std::max(n1, 1) * 2);
  int n = 0;
  const orbx_camera cam = camera.c();
  h.check(orbx_search_for_triangulation_bow(h.get(), &cam, f1.keypoints.data(), f1.descriptors.data(), has_map_point1.data(),
                                            has_point_cam1.data(), node1.data(), n1, f2.keypoints.data(), f2.descriptors.data(),
                                            has_map_point2.data(), node2.data(), n2, p1, p2, max_dist, pairs.data(), &n));
  std::vector<std::pair<size_t, size_t>> out((size_t)n);
  for (int i = 0; i < n; ++i) out[(size_t)i] = {(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]};
  return out;
}

// ---- global bundle adjustment (src/optimizer/global_ba.rs) --------------------------------------------------------
struct GlobalBAConfig {   // global_ba.rs:21-46
  int max_iterations = 10;
  double param_tolerance = 1e-6, gradient_tolerance = 1e-6, huber_threshold = std::sqrt(5.991);
};

struct GlobalBAObservation {   // global_ba.rs:70-80
  KeyFrameId kf_id;
  MapPointId mp_id;
  std::array<double, 2> observed_uv;
};

struct GlobalBAProblemData {   // global_ba.rs:49-67; poses are T_cw
  std::unordered_map<KeyFrameId, SE3> kf_poses;
  std::unordered_map<MapPointId, std::array<double, 3>> mp_positions;
  std::vector<GlobalBAObservation> observations;
  std::vector<KeyFrameId> kf_ids;
  std::vector<MapPointId> mp_ids;
  KeyFrameId fixed_kf_id = 0;
};

struct GlobalBAResult {   // global_ba.rs:83-98; optimized_poses are T_wc and include the fixed keyframe
  std::unordered_map<KeyFrameId, SE3> optimized_poses;
  std::unordered_map<MapPointId, std::array<double, 3>> optimized_points;
  size_t iterations = 0;
  double initial_error = 0, final_error = 0;
};

inline SE3 se3_inverse(const SE3& p) {   // se3.rs:56-63 with nalgebra's quaternion-vector product
  const double w = p.rotation[0], x = -p.rotation[1], y = -p.rotation[2], z = -p.rotation[3];
  const double* v = p.translation.data();
  const double t[3] = {2.0 * (y * v[2] - z * v[1]), 2.0 * (z * v[0] - x * v[2]), 2.0 * (x * v[1] - y * v[0])};
  const double c[3] = {y * t[2] - z * t[1], z * t[0] - x * t[2], x * t[1] - y * t[0]};
  SE3 r;
  r.rotation = {w, x, y, z};
  for (int i = 0; i < 3; ++i) r.translation[i] = -(t[i] * w + c[i] + v[i]);
  return r;
}

// global_ba.rs:184-418.  The id -> index re-keying is the reference's own (:198-229).  Observations of a map point
// that is not in mp_ids are rejected (collect_global_ba_data never emits one, :160).
inline std::optional<GlobalBAResult> solve_global_ba(Handle& h, const GlobalBAProblemData& problem, const CameraModel& camera,
                                                     const GlobalBAConfig& config, const std::function<bool()>& should_stop) {
  const size_t n_kfs = problem.kf_ids.size(), n_mps = problem.mp_ids.size();
  if (n_kfs < 2 || n_mps == 0) return std::nullopt;                      // :194-196
  size_t fixed_pos = n_kfs;
  for (size_t i = 0; i < n_kfs; ++i) if (problem.kf_ids[i] == problem.fixed_kf_id) { fixed_pos = i; break; }
  if (fixed_pos == n_kfs) return std::nullopt;                           // :199-202
  std::unordered_map<KeyFrameId, int> kf_to_param;
  std::vector<KeyFrameId> opt_ids;
  for (size_t i = 0; i < n_kfs; ++i) if (i != fixed_pos) { kf_to_param[problem.kf_ids[i]] = (int)opt_ids.size(); opt_ids.push_back(problem.kf_ids[i]); }
  std::unordered_map<MapPointId, int> mp_to_param;
  for (size_t i = 0; i < n_mps; ++i) mp_to_param[problem.mp_ids[i]] = (int)i;
  auto push7 = [](std::vector<double>& v, const SE3& p) {
    v.insert(v.end(), p.rotation.begin(), p.rotation.end());
    v.insert(v.end(), p.translation.begin(), p.translation.end());
  };
  std::vector<double> poses, fixed, points;
  for (KeyFrameId id : opt_ids) { auto it = problem.kf_poses.find(id); push7(poses, it != problem.kf_poses.end() ? it->second : SE3{}); }   // :232-243
  auto fit = problem.kf_poses.find(problem.fixed_kf_id);
  const SE3 fixed_pose = fit != problem.kf_poses.end() ? fit->second : SE3{};                                                             // :256-261
  push7(fixed, fixed_pose);
  for (MapPointId id : problem.mp_ids) {
    auto it = problem.mp_positions.find(id);
    const std::array<double, 3> p = it != problem.mp_positions.end() ? it->second : std::array<double, 3>{0, 0, 0};
    points.insert(points.end(), p.begin(), p.end());
  }
  std::vector<orbx_ba_obs> obs;
  for (const GlobalBAObservation& o : problem.observations) {
    auto m = mp_to_param.find(o.mp_id);
    if (m == mp_to_param.end()) throw Error(ORBX_ERR_INVALID, "solve_global_ba: observation of a map point that is not in mp_ids");
    orbx_ba_obs b{};
    b.mp_idx = m->second; b.u = o.observed_uv[0]; b.v = o.observed_uv[1];
    auto k = kf_to_param.find(o.kf_id);
    if (o.kf_id == problem.fixed_kf_id) { b.kf_idx = -1; b.fixed_idx = 0; }            // :639-641
    else if (k != kf_to_param.end()) { b.kf_idx = k->second; b.fixed_idx = -1; }
    else { b.kf_idx = -1; b.fixed_idx = -1; }                                          // :649 identity
    obs.push_back(b);
  }
  const int K = (int)opt_ids.size(), M = (int)n_mps;
  std::vector<double> out((size_t)K * 7);
  int it = 0;
  double e0 = 0, e1 = 0;
  const orbx_camera c = camera.c();
  const orbx_ba_config cfg{config.max_iterations, config.param_tolerance, config.gradient_tolerance, config.huber_threshold, 0};
  auto tramp = [](void* user) -> int { return (*static_cast<const std::function<bool()>*>(user))() ? 1 : 0; };
  const int rc = orbx_ba_solve_global(h.get(), &c, &cfg, K, poses.data(), fixed.data(), M, points.data(), (int)obs.size(), obs.data(),
                                      should_stop ? +tramp : nullptr, const_cast<std::function<bool()>*>(&should_stop), out.data(), &it,
                                      &e0, &e1);
  if (rc != ORBX_OK) return std::nullopt;
  GlobalBAResult r;
  r.optimized_poses[problem.fixed_kf_id] = se3_inverse(fixed_pose);      // :386
  for (int i = 0; i < K; ++i) {
    SE3 p;
    for (int q = 0; q < 4; ++q) p.rotation[q] = out[7 * (size_t)i + q];
    for (int q = 0; q < 3; ++q) p.translation[q] = out[7 * (size_t)i + 4 + q];
    r.optimized_poses[opt_ids[(size_t)i]] = p;
  }
  for (int j = 0; j < M; ++j) r.optimized_points[problem.mp_ids[(size_t)j]] = {points[3 * (size_t)j], points[3 * (size_t)j + 1], points[3 * (size_t)j + 2]};
  r.iterations = (size_t)it; r.initial_error = e0; r.final_error = e1;
  return r;
}


// ---- local inertial bundle adjustment (src/optimizer/local_inertial_ba.rs) -----------------------------------------
struct ImuBias { std::array<double, 3> gyro{0, 0, 0}, accel{0, 0, 0}; };   // imu/types.rs

struct PreintegratedState {   // imu/preintegration.rs:85-98, the fields the residual reads (imu_factors.rs:66-103)
  std::array<double, 4> delta_rot{1, 0, 0, 0};
  std::array<double, 3> delta_vel{0, 0, 0}, delta_pos{0, 0, 0};
  double dt = 0.0;
};

struct LocalInertialBAConfig {   // local_inertial_ba.rs:109-141
  int max_iterations = 10, window_size = 10;
  double huber_threshold_mono = std::sqrt(5.991), huber_threshold_stereo = std::sqrt(7.815), initial_lambda = 1e-2;
  double gyro_rw_info = 1e6, accel_rw_info = 1e4;
};

struct InertialVisualObs {   // :64-77
  KeyFrameId kf_id;
  MapPointId mp_id;
  std::array<double, 2> observed_uv;
  bool is_stereo, is_kf_in_window;
};

struct ImuEdgeData {   // :79-88
  KeyFrameId kf_i_id, kf_j_id;
  PreintegratedState preint;
};

struct InertialBAProblemData {   // :40-62; kf_poses T_wc, fixed_kf_poses T_cw
  std::unordered_map<KeyFrameId, SE3> kf_poses;
  std::unordered_map<KeyFrameId, std::array<double, 3>> kf_velocities;
  std::unordered_map<KeyFrameId, ImuBias> kf_biases;
  std::unordered_map<MapPointId, std::array<double, 3>> mp_positions;
  std::unordered_map<KeyFrameId, SE3> fixed_kf_poses;
  std::vector<InertialVisualObs> visual_observations;
  std::vector<ImuEdgeData> imu_edges;
  std::vector<KeyFrameId> opt_kf_ids;
  std::vector<MapPointId> mp_ids;
};

struct InertialBAResultData {   // :90-106; the first keyframe of the window is not reported (:1250)
  std::unordered_map<KeyFrameId, SE3> optimized_poses;
  std::unordered_map<KeyFrameId, std::array<double, 3>> optimized_velocities;
  std::unordered_map<KeyFrameId, ImuBias> optimized_biases;
  std::unordered_map<MapPointId, std::array<double, 3>> optimized_points;
  size_t iterations = 0;
  double initial_error = 0, final_error = 0;
};

// local_inertial_ba.rs:1074-1275.  The id -> index re-keying is the reference's own (:1084-1185).
inline std::optional<InertialBAResultData> solve_inertial_ba(Handle& h, const InertialBAProblemData& problem, const CameraModel& camera,
                                                             const LocalInertialBAConfig& config, const std::function<bool()>& should_stop) {
  const int K = (int)problem.opt_kf_ids.size(), M = (int)problem.mp_ids.size();
  if (K < 2) return std::nullopt;                                         // :1080-1082
  std::unordered_map<KeyFrameId, int> kf_idx, fixed_idx;
  std::unordered_map<MapPointId, int> mp_idx;
  for (int i = 0; i < K; ++i) kf_idx[problem.opt_kf_ids[(size_t)i]] = i;
  for (int i = 0; i < M; ++i) mp_idx[problem.mp_ids[(size_t)i]] = i;
  std::vector<double> poses, vel, bias, fixed, points, preint;
  auto push7 = [](std::vector<double>& v, const SE3& p) {
    v.insert(v.end(), p.rotation.begin(), p.rotation.end());
    v.insert(v.end(), p.translation.begin(), p.translation.end());
  };
  for (KeyFrameId id : problem.opt_kf_ids) {                               // :1140-1174, missing entries stay zero
    auto p = problem.kf_poses.find(id);
    push7(poses, p != problem.kf_poses.end() ? p->second : SE3{});
    auto v = problem.kf_velocities.find(id);
    const std::array<double, 3> vv = v != problem.kf_velocities.end() ? v->second : std::array<double, 3>{0, 0, 0};
    vel.insert(vel.end(), vv.begin(), vv.end());
    auto b = problem.kf_biases.find(id);
    const ImuBias bb = b != problem.kf_biases.end() ? b->second : ImuBias{};
    bias.insert(bias.end(), bb.gyro.begin(), bb.gyro.end());
    bias.insert(bias.end(), bb.accel.begin(), bb.accel.end());
  }
  for (const auto& kv : problem.fixed_kf_poses) { fixed_idx[kv.first] = (int)fixed_idx.size(); push7(fixed, kv.second); }
  for (MapPointId id : problem.mp_ids) {
    auto it = problem.mp_positions.find(id);
    const std::array<double, 3> p = it != problem.mp_positions.end() ? it->second : std::array<double, 3>{0, 0, 0};
    points.insert(points.end(), p.begin(), p.end());
  }
  std::vector<orbx_ba_obs> obs;
  for (const InertialVisualObs& o : problem.visual_observations) {         // :1101-1120
    auto m = mp_idx.find(o.mp_id);
    if (m == mp_idx.end()) continue;
    orbx_ba_obs b{};
    b.mp_idx = m->second; b.u = o.observed_uv[0]; b.v = o.observed_uv[1]; b._pad = o.is_stereo ? 1 : 0;
    auto k = o.is_kf_in_window ? kf_idx.find(o.kf_id) : kf_idx.end();
    if (k != kf_idx.end()) { b.kf_idx = k->second; b.fixed_idx = -1; }
    else { auto f = fixed_idx.find(o.kf_id); b.kf_idx = -1; b.fixed_idx = f != fixed_idx.end() ? f->second : -1; }
    obs.push_back(b);
  }
  std::vector<int> edges;
  for (const ImuEdgeData& e : problem.imu_edges) {                         // :1123-1137
    auto i = kf_idx.find(e.kf_i_id), j = kf_idx.find(e.kf_j_id);
    if (i == kf_idx.end() || j == kf_idx.end()) continue;
    edges.push_back(i->second); edges.push_back(j->second);
    preint.insert(preint.end(), e.preint.delta_rot.begin(), e.preint.delta_rot.end());
    preint.insert(preint.end(), e.preint.delta_vel.begin(), e.preint.delta_vel.end());
    preint.insert(preint.end(), e.preint.delta_pos.begin(), e.preint.delta_pos.end());
    preint.push_back(e.preint.dt);
  }
  std::vector<double> po((size_t)K * 7), vo((size_t)K * 3), bo((size_t)K * 6);
  int it = 0;
  double e0 = 0, e1 = 0;
  const orbx_camera c = camera.c();
  const orbx_inertial_ba_config cfg{config.max_iterations, config.window_size, config.huber_threshold_mono, config.huber_threshold_stereo,
                                    config.initial_lambda, config.gyro_rw_info, config.accel_rw_info};
  auto tramp = [](void* user) -> int { return (*static_cast<const std::function<bool()>*>(user))() ? 1 : 0; };
  const int rc = orbx_ba_solve_inertial(h.get(), &c, &cfg, K, poses.data(), vel.data(), bias.data(), (int)fixed_idx.size(), fixed.data(), M,
                                        points.data(), (int)obs.size(), obs.data(), (int)(edges.size() / 2), edges.data(), preint.data(),
                                        should_stop ? +tramp : nullptr, const_cast<std::function<bool()>*>(&should_stop), po.data(), vo.data(),
                                        bo.data(), &it, &e0, &e1);
  if (rc != ORBX_OK) return std::nullopt;
  InertialBAResultData r;
  for (int i = 1; i < K; ++i) {                                            // skip(1), :1250
    const KeyFrameId id = problem.opt_kf_ids[(size_t)i];
    SE3 p;
    for (int q = 0; q < 4; ++q) p.rotation[q] = po[7 * (size_t)i + q];
    for (int q = 0; q < 3; ++q) p.translation[q] = po[7 * (size_t)i + 4 + q];
    r.optimized_poses[id] = p;
    r.optimized_velocities[id] = {vo[3 * (size_t)i], vo[3 * (size_t)i + 1], vo[3 * (size_t)i + 2]};
    ImuBias b;
    for (int q = 0; q < 3; ++q) { b.gyro[q] = bo[6 * (size_t)i + q]; b.accel[q] = bo[6 * (size_t)i + 3 + q]; }
    r.optimized_biases[id] = b;
  }
  for (int j = 0; j < M; ++j) r.optimized_points[problem.mp_ids[(size_t)j]] = {points[3 * (size_t)j], points[3 * (size_t)j + 1], points[3 * (size_t)j + 2]};
  r.iterations = (size_t)it; r.initial_error = e0; r.final_error = e1;
  return r;
}

}  // namespace orbx
