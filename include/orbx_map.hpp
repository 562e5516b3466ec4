// orbx_map.hpp — the host side of local BA over FLAT arrays: the three phases of
// LocalMapper::local_bundle_adjustment (src/local_mapping/local_mapper.rs:334-410), both branches:
//
//   visual (:378-408, before IMU initialisation)
//   collect_visual_ba_data    src/optimizer/local_ba_lm.rs:800-897 (with collect_local_keyframes :665-683,
//                             collect_local_map_points :686-704, collect_fixed_keyframes :707-726)
//   solve_visual_ba           orbx.hpp (GPU)
//   apply_visual_ba_results   local_ba_lm.rs:1112-1138
//
//   inertial (:343-375, once Map::is_imu_initialized())
//   collect_inertial_ba_data  src/optimizer/local_inertial_ba.rs:933-1072 (with collect_temporal_keyframes :366-384,
//                             collect_map_points :387-403, collect_fixed_keyframes :406-429)
//   solve_inertial_ba         orbx.hpp (GPU)
//   apply_inertial_ba_results local_inertial_ba.rs:1289-1330
//
// and the host phases of global BA (src/optimizer/global_ba.rs): collect_global_ba_data :100-181, apply_global_ba_results :421-443,
// run_global_ba :450-500 around solve_global_ba (orbx.hpp, GPU).
//
// The reference walks its HashMap-based `Map` under a read lock.  A GPU-side BA wants the same information as a handful
// of arrays, so the snapshot here is CSR: keyframes with their features (map-point id or none, keypoint position), their
// covisibility lists, map points with their observer lists.  A Rust shim fills it under the read lock (or keeps it in
// step with the map) — `INTEGRATION.md` shows the loop — and everything below is plain index arithmetic.
//
// ORDER.  The reference iterates HashMaps / HashSets at four places (covisibility weights :675, the map-point set :704,
// the fixed-keyframe set :725, mp.observations.keys() :718); their order is unspecified and changes from run to run
// (SURVEY.md F10), and it matters: `.take(max_covisible)` picks WHICH neighbours are local, and the order of the
// observations is the summation order of J^T J.  This file states a deterministic order instead:
//   - covisibility neighbours of a keyframe: the order of the snapshot's list (recommended fill: weight descending, then
//     id ascending — ORB-SLAM's ordered covisibility; filling it in the HashMap's iteration order reproduces one
//     particular run of the reference);
//   - map points: first seen, walking the local keyframes in order and their features in order;
//   - fixed keyframes: first seen, walking those map points in order and each one's observer list in order;
//   - observations: local keyframes (current first), then fixed keyframes, features in order — as :856-882.
// With the lists filled in the same order, the result equals the reference's line by line (oracle/local_mapper_ref.py is
// that restatement; tests/test_local_mapper_host.py compares them).
#ifndef ORBX_MAP_HPP
#define ORBX_MAP_HPP

#include <algorithm>
#include <atomic>
#include <functional>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "orbx.hpp"

namespace orbx {

struct MapSnapshot {
  // keyframes (atlas/map/keyframe.rs): pose = T_wc as the Map stores it
  std::vector<KeyFrameId> kf_ids;
  std::vector<uint8_t> kf_bad;                 // is_bad
  std::vector<double> kf_pose_wc;              // [nkf][7] qw,qx,qy,qz,tx,ty,tz
  std::vector<int> kf_n_keypoints;             // keypoints.len(): `keypoints.get(feat_idx)` fails beyond it (:866)
  std::vector<int> kf_feat_start;              // [nkf+1] CSR over map_point_ids
  std::vector<int64_t> feat_mp_id;             // Some(id) or -1 = None
  std::vector<float> feat_uv;                  // [nfeat][2] kp.pt()
  std::vector<int> cov_start;                  // [nkf+1] covisibility_weights(), in the stated order
  std::vector<KeyFrameId> cov_kf_id;
  // map points (atlas/map/map_point.rs)
  std::vector<MapPointId> mp_ids;
  std::vector<uint8_t> mp_bad;
  std::vector<double> mp_pos;                  // [nmp][3]
  std::vector<int> mp_obs_start;               // [nmp+1] observations.keys(), in the stated order
  std::vector<KeyFrameId> mp_obs_kf_id;
  // what the inertial branch reads besides (keyframe.rs: prev_kf, velocity, imu_bias, imu_preintegrated, points_cam; map_point.rs:
  // observations' feature index).  May stay empty for a map that never takes the inertial branch (imu_initialized = 0).
  std::vector<int64_t> kf_prev_id;             // [nkf] prev_kf, -1 = None
  std::vector<double> kf_velocity;             // [nkf][3]
  std::vector<double> kf_bias;                 // [nkf][6] gyro, accel
  std::vector<uint8_t> kf_has_preint;          // [nkf] imu_preintegrated.is_some()
  std::vector<double> kf_preint;               // [nkf][11] delta_rot (qw,qx,qy,qz) | delta_vel | delta_pos | dt — from prev_kf to this keyframe
  std::vector<uint8_t> feat_stereo;            // [nfeat] points_cam[i].is_some()
  std::vector<int> mp_obs_feat_idx;            // [nobs] the feature index of the observation (observations: HashMap<KeyFrameId, usize>)
  uint8_t imu_initialized = 0;                 // Map::is_imu_initialized()

  // id -> index (map.get_keyframe / map.get_map_point); call after filling or changing the id arrays
  void build_index() {
    kf_index_.clear(); mp_index_.clear();
    for (size_t i = 0; i < kf_ids.size(); ++i) kf_index_[kf_ids[i]] = (int)i;
    for (size_t i = 0; i < mp_ids.size(); ++i) mp_index_[mp_ids[i]] = (int)i;
  }
  int kf_index(KeyFrameId id) const { auto it = kf_index_.find(id); return it == kf_index_.end() ? -1 : it->second; }
  int mp_index(MapPointId id) const { auto it = mp_index_.find(id); return it == mp_index_.end() ? -1 : it->second; }
  SE3 kf_pose(int k) const {
    SE3 p;
    for (int i = 0; i < 4; ++i) p.rotation[i] = kf_pose_wc[7 * (size_t)k + i];
    for (int i = 0; i < 3; ++i) p.translation[i] = kf_pose_wc[7 * (size_t)k + 4 + i];
    return p;
  }

 private:
  std::unordered_map<KeyFrameId, int> kf_index_;
  std::unordered_map<MapPointId, int> mp_index_;
};

// local_ba_lm.rs:665-683
inline std::vector<KeyFrameId> collect_local_keyframes(const MapSnapshot& m, KeyFrameId current_kf_id, size_t max_covisible) {
  std::vector<KeyFrameId> local{current_kf_id};
  const int k = m.kf_index(current_kf_id);
  if (k >= 0) {
    const int s = m.cov_start[(size_t)k], e = m.cov_start[(size_t)k + 1];
    for (int i = s; i < e && (size_t)(i - s) < max_covisible; ++i) {     // .iter().take(max_covisible) comes BEFORE the filter
      const int nb = m.kf_index(m.cov_kf_id[(size_t)i]);
      if (nb >= 0 && !m.kf_bad[(size_t)nb]) local.push_back(m.cov_kf_id[(size_t)i]);
    }
  }
  return local;
}

// local_ba_lm.rs:686-704 (set in first-seen order)
inline std::vector<MapPointId> collect_local_map_points(const MapSnapshot& m, const std::vector<KeyFrameId>& local_kf_ids) {
  std::unordered_set<MapPointId> seen;
  std::vector<MapPointId> out;
  for (KeyFrameId id : local_kf_ids) {
    const int k = m.kf_index(id);
    if (k < 0) continue;
    for (int f = m.kf_feat_start[(size_t)k]; f < m.kf_feat_start[(size_t)k + 1]; ++f) {
      const int64_t mp_id = m.feat_mp_id[(size_t)f];
      if (mp_id < 0) continue;
      const int j = m.mp_index((MapPointId)mp_id);
      if (j >= 0 && !m.mp_bad[(size_t)j] && seen.insert((MapPointId)mp_id).second) out.push_back((MapPointId)mp_id);
    }
  }
  return out;
}

// local_ba_lm.rs:707-726 (set in first-seen order; neither existence nor is_bad is checked there)
inline std::vector<KeyFrameId> collect_fixed_keyframes(const MapSnapshot& m, const std::vector<KeyFrameId>& local_kf_ids,
                                                       const std::vector<MapPointId>& local_mp_ids) {
  const std::unordered_set<KeyFrameId> local(local_kf_ids.begin(), local_kf_ids.end());
  std::unordered_set<KeyFrameId> seen;
  std::vector<KeyFrameId> out;
  for (MapPointId id : local_mp_ids) {
    const int j = m.mp_index(id);
    if (j < 0) continue;
    for (int o = m.mp_obs_start[(size_t)j]; o < m.mp_obs_start[(size_t)j + 1]; ++o) {
      const KeyFrameId kf = m.mp_obs_kf_id[(size_t)o];
      if (!local.count(kf) && seen.insert(kf).second) out.push_back(kf);
    }
  }
  return out;
}

// PHASE 1, local_ba_lm.rs:800-897
inline std::optional<VisualBAProblemData> collect_visual_ba_data(const MapSnapshot& m, KeyFrameId current_kf_id,
                                                                 const LocalBAConfigLM& config) {
  const std::vector<KeyFrameId> local_kf_ids = collect_local_keyframes(m, current_kf_id, (size_t)config.max_covisible_keyframes);
  if (local_kf_ids.empty()) return std::nullopt;                                     // :807-809
  std::vector<MapPointId> mp_ids = collect_local_map_points(m, local_kf_ids);
  if (mp_ids.empty()) return std::nullopt;                                           // :813-815
  const std::vector<KeyFrameId> fixed_kf_ids = collect_fixed_keyframes(m, local_kf_ids, mp_ids);
  VisualBAProblemData p;
  p.anchor_kf_id = local_kf_ids.front();                                             // :821
  p.optimized_kf_ids.assign(local_kf_ids.begin() + 1, local_kf_ids.end());           // :822
  for (KeyFrameId id : p.optimized_kf_ids) {                                         // :825-830
    const int k = m.kf_index(id);
    if (k >= 0) p.local_kf_poses[id] = se3_inverse(m.kf_pose(k));                    // T_cw
  }
  {
    const int k = m.kf_index(p.anchor_kf_id);                                        // :833-836
    if (k >= 0) p.fixed_kf_poses[p.anchor_kf_id] = se3_inverse(m.kf_pose(k));
  }
  for (KeyFrameId id : fixed_kf_ids) {                                               // :837-841
    const int k = m.kf_index(id);
    if (k >= 0) p.fixed_kf_poses[id] = se3_inverse(m.kf_pose(k));
  }
  for (MapPointId id : mp_ids) {                                                     // :844-849
    const int j = m.mp_index(id);
    if (j >= 0) p.local_mp_positions[id] = {m.mp_pos[3 * (size_t)j], m.mp_pos[3 * (size_t)j + 1], m.mp_pos[3 * (size_t)j + 2]};
  }
  const std::unordered_set<KeyFrameId> local_kf_set(p.optimized_kf_ids.begin(), p.optimized_kf_ids.end());
  const std::unordered_set<MapPointId> local_mp_set(mp_ids.begin(), mp_ids.end());
  auto walk = [&](KeyFrameId id) {                                                   // :863-882
    const int k = m.kf_index(id);
    if (k < 0) return;
    const int s = m.kf_feat_start[(size_t)k], e = m.kf_feat_start[(size_t)k + 1];
    for (int f = s; f < e; ++f) {
      const int64_t mp_id = m.feat_mp_id[(size_t)f];
      if (mp_id < 0 || !local_mp_set.count((MapPointId)mp_id)) continue;
      if (f - s >= m.kf_n_keypoints[(size_t)k]) continue;                            // keypoints.get(feat_idx) is Err
      p.observations.push_back(VisualObservation{id, (MapPointId)mp_id,
                                                 {(double)m.feat_uv[2 * (size_t)f], (double)m.feat_uv[2 * (size_t)f + 1]},
                                                 local_kf_set.count(id) != 0});
    }
  };
  for (KeyFrameId id : local_kf_ids) walk(id);
  for (KeyFrameId id : fixed_kf_ids) walk(id);
  if (p.observations.empty()) return std::nullopt;                                   // :884-886
  p.mp_ids = std::move(mp_ids);
  return p;
}

// PHASE 3, local_ba_lm.rs:1112-1138: entities that are gone or bad by now are skipped silently
inline size_t apply_visual_ba_results(MapSnapshot& m, const VisualBAResultData& r) {
  size_t updated = 0;
  for (const auto& kv : r.optimized_poses) {
    const int k = m.kf_index(kv.first);
    if (k >= 0 && !m.kf_bad[(size_t)k]) {
      for (int i = 0; i < 4; ++i) m.kf_pose_wc[7 * (size_t)k + i] = kv.second.rotation[i];
      for (int i = 0; i < 3; ++i) m.kf_pose_wc[7 * (size_t)k + 4 + i] = kv.second.translation[i];
      ++updated;
    }
  }
  for (const auto& kv : r.optimized_points) {
    const int j = m.mp_index(kv.first);
    if (j >= 0 && !m.mp_bad[(size_t)j]) {
      for (int i = 0; i < 3; ++i) m.mp_pos[3 * (size_t)j + i] = kv.second[i];
      ++updated;
    }
  }
  return updated;
}

// ---- the inertial branch ---------------------------------------------------------------------------------------------------------
// local_inertial_ba.rs:366-384: the temporal chain through prev_kf, oldest first (the anchor)
inline std::vector<KeyFrameId> collect_temporal_keyframes(const MapSnapshot& m, KeyFrameId current_kf_id, size_t window_size) {
  std::vector<KeyFrameId> kf_ids{current_kf_id};
  auto prev_of = [&](KeyFrameId id) -> int64_t { const int k = m.kf_index(id); return k >= 0 && (size_t)k < m.kf_prev_id.size() ? m.kf_prev_id[(size_t)k] : -1; };
  int64_t prev = prev_of(current_kf_id);
  while (kf_ids.size() < window_size && prev >= 0) {
    kf_ids.push_back((KeyFrameId)prev);
    prev = prev_of((KeyFrameId)prev);
  }
  std::reverse(kf_ids.begin(), kf_ids.end());
  return kf_ids;
}

// local_inertial_ba.rs:406-429 (set in first-seen order; unlike the visual one, existence and is_bad ARE checked)
inline std::vector<KeyFrameId> collect_fixed_keyframes_inertial(const MapSnapshot& m, const std::vector<KeyFrameId>& opt_kf_ids,
                                                                const std::vector<MapPointId>& mp_ids) {
  const std::unordered_set<KeyFrameId> opt(opt_kf_ids.begin(), opt_kf_ids.end());
  std::unordered_set<KeyFrameId> seen;
  std::vector<KeyFrameId> out;
  for (MapPointId id : mp_ids) {
    const int j = m.mp_index(id);
    if (j < 0) continue;
    for (int o = m.mp_obs_start[(size_t)j]; o < m.mp_obs_start[(size_t)j + 1]; ++o) {
      const KeyFrameId kf = m.mp_obs_kf_id[(size_t)o];
      if (opt.count(kf)) continue;
      const int k = m.kf_index(kf);
      if (k >= 0 && !m.kf_bad[(size_t)k] && seen.insert(kf).second) out.push_back(kf);
    }
  }
  return out;
}

// PHASE 1, local_inertial_ba.rs:933-1072
inline std::optional<InertialBAProblemData> collect_inertial_ba_data(const MapSnapshot& m, KeyFrameId current_kf_id,
                                                                     const LocalInertialBAConfig& config) {
  InertialBAProblemData p;
  p.opt_kf_ids = collect_temporal_keyframes(m, current_kf_id, (size_t)config.window_size);
  if (p.opt_kf_ids.size() < 2) return std::nullopt;                                  // :940-942
  p.mp_ids = collect_local_map_points(m, p.opt_kf_ids);                               // :387-403: the same walk as the visual one
  if (p.mp_ids.empty()) return std::nullopt;                                         // :946-948
  const std::vector<KeyFrameId> fixed_kf_ids = collect_fixed_keyframes_inertial(m, p.opt_kf_ids, p.mp_ids);
  const std::unordered_set<KeyFrameId> opt_kf_set(p.opt_kf_ids.begin(), p.opt_kf_ids.end());
  for (KeyFrameId id : p.opt_kf_ids) {                                                // :955-964
    const int k = m.kf_index(id);
    if (k < 0) continue;
    p.kf_poses[id] = m.kf_pose(k);                                                    // T_wc
    std::array<double, 3> v{0, 0, 0};
    ImuBias b;
    if (3 * (size_t)k + 2 < m.kf_velocity.size()) for (int i = 0; i < 3; ++i) v[(size_t)i] = m.kf_velocity[3 * (size_t)k + i];
    if (6 * (size_t)k + 5 < m.kf_bias.size())
      for (int i = 0; i < 3; ++i) { b.gyro[(size_t)i] = m.kf_bias[6 * (size_t)k + i]; b.accel[(size_t)i] = m.kf_bias[6 * (size_t)k + 3 + i]; }
    p.kf_velocities[id] = v;
    p.kf_biases[id] = b;
  }
  for (KeyFrameId id : fixed_kf_ids) {                                                // :967-972
    const int k = m.kf_index(id);
    if (k >= 0) p.fixed_kf_poses[id] = se3_inverse(m.kf_pose(k));                     // T_cw
  }
  {
    const int k = m.kf_index(p.opt_kf_ids.front());                                   // :974-978: the anchor's pose too
    if (k >= 0) p.fixed_kf_poses[p.opt_kf_ids.front()] = se3_inverse(m.kf_pose(k));
  }
  for (MapPointId id : p.mp_ids) {                                                    // :981-986
    const int j = m.mp_index(id);
    if (j >= 0) p.mp_positions[id] = {m.mp_pos[3 * (size_t)j], m.mp_pos[3 * (size_t)j + 1], m.mp_pos[3 * (size_t)j + 2]};
  }
  std::unordered_set<KeyFrameId> all_kf(p.opt_kf_ids.begin(), p.opt_kf_ids.end());
  all_kf.insert(fixed_kf_ids.begin(), fixed_kf_ids.end());
  for (MapPointId mp_id : p.mp_ids) {                                                 // :996-1029: by map point, its observers in list order
    const int j = m.mp_index(mp_id);
    if (j < 0) continue;
    for (int o = m.mp_obs_start[(size_t)j]; o < m.mp_obs_start[(size_t)j + 1]; ++o) {
      const KeyFrameId kf_id = m.mp_obs_kf_id[(size_t)o];
      if (!all_kf.count(kf_id)) continue;
      const int k = m.kf_index(kf_id);
      if (k < 0) continue;
      const int fi = (size_t)o < m.mp_obs_feat_idx.size() ? m.mp_obs_feat_idx[(size_t)o] : -1;
      const int s = m.kf_feat_start[(size_t)k], e = m.kf_feat_start[(size_t)k + 1];
      if (fi < 0 || fi >= m.kf_n_keypoints[(size_t)k] || s + fi >= e) continue;       // keypoints.get(feat_idx) is Err
      const size_t f = (size_t)(s + fi);
      const bool is_stereo = f < m.feat_stereo.size() && m.feat_stereo[f] != 0;       // points_cam.get(i).map_or(false, |p| p.is_some())
      const bool in_window = opt_kf_set.count(kf_id) != 0 && p.opt_kf_ids.front() != kf_id;   // the anchor is treated as fixed
      p.visual_observations.push_back(InertialVisualObs{kf_id, mp_id, {(double)m.feat_uv[2 * f], (double)m.feat_uv[2 * f + 1]}, is_stereo, in_window});
    }
  }
  for (size_t i = 0; i + 1 < p.opt_kf_ids.size(); ++i) {                              // :1032-1049
    const int kj = m.kf_index(p.opt_kf_ids[i + 1]);
    if (kj < 0 || (size_t)kj >= m.kf_has_preint.size() || !m.kf_has_preint[(size_t)kj]) continue;
    const double* q = &m.kf_preint[11 * (size_t)kj];
    if (!(q[10] > 0.0)) continue;                                                     // preint.dt > 0.0
    ImuEdgeData e{p.opt_kf_ids[i], p.opt_kf_ids[i + 1], {}};
    for (int a = 0; a < 4; ++a) e.preint.delta_rot[(size_t)a] = q[a];
    for (int a = 0; a < 3; ++a) { e.preint.delta_vel[(size_t)a] = q[4 + a]; e.preint.delta_pos[(size_t)a] = q[7 + a]; }
    e.preint.dt = q[10];
    p.imu_edges.push_back(e);
  }
  return p;
}

// PHASE 3, local_inertial_ba.rs:1289-1330: gone or bad entities are skipped silently; poses and points count, velocities and biases do not
inline size_t apply_inertial_ba_results(MapSnapshot& m, const InertialBAResultData& r) {
  size_t updated = 0;
  for (const auto& kv : r.optimized_poses) {
    const int k = m.kf_index(kv.first);
    if (k >= 0 && !m.kf_bad[(size_t)k]) {
      for (int i = 0; i < 4; ++i) m.kf_pose_wc[7 * (size_t)k + i] = kv.second.rotation[(size_t)i];
      for (int i = 0; i < 3; ++i) m.kf_pose_wc[7 * (size_t)k + 4 + i] = kv.second.translation[(size_t)i];
      ++updated;
    }
  }
  for (const auto& kv : r.optimized_velocities) {
    const int k = m.kf_index(kv.first);
    if (k >= 0 && !m.kf_bad[(size_t)k] && 3 * (size_t)k + 2 < m.kf_velocity.size())
      for (int i = 0; i < 3; ++i) m.kf_velocity[3 * (size_t)k + i] = kv.second[(size_t)i];
  }
  for (const auto& kv : r.optimized_biases) {
    const int k = m.kf_index(kv.first);
    if (k >= 0 && !m.kf_bad[(size_t)k] && 6 * (size_t)k + 5 < m.kf_bias.size())
      for (int i = 0; i < 3; ++i) { m.kf_bias[6 * (size_t)k + i] = kv.second.gyro[(size_t)i]; m.kf_bias[6 * (size_t)k + 3 + i] = kv.second.accel[(size_t)i]; }
  }
  for (const auto& kv : r.optimized_points) {
    const int j = m.mp_index(kv.first);
    if (j >= 0 && !m.mp_bad[(size_t)j]) {
      for (int i = 0; i < 3; ++i) m.mp_pos[3 * (size_t)j + i] = kv.second[(size_t)i];
      ++updated;
    }
  }
  return updated;
}

// LocalMapper::local_bundle_adjustment (local_mapper.rs:334-410), both branches.  The reference takes the map's read
// lock around phase 1 and its write lock around phase 3 and holds none while solving; `lock_read` / `lock_write` wrap the
// two phases the same way (pass no-ops for a snapshot nobody else touches), plus the short read lock in which the reference asks
// is_imu_initialized().  Returns nullopt where the reference returns early (:351, :359, :384, :391), else the number of updated
// entities — 0 without touching the map when the solve ran no iteration (:363, :396).
inline std::optional<size_t> local_bundle_adjustment(Handle& h, MapSnapshot& map, KeyFrameId kf_id, const CameraModel& camera,
                                                     const std::function<bool()>& should_stop,
                                                     const std::function<void(const std::function<void()>&)>& lock_read = nullptr,
                                                     const std::function<void(const std::function<void()>&)>& lock_write = nullptr,
                                                     VisualBAResultData* result_out = nullptr,
                                                     InertialBAResultData* inertial_result_out = nullptr) {
  bool is_inertial = false;                                                          // :338-341: a quick read lock of its own
  auto phase0 = [&] { is_inertial = map.imu_initialized != 0; };
  if (lock_read) lock_read(phase0); else phase0();
  if (is_inertial) {                                                                 // :343-375
    const LocalInertialBAConfig config;                                              // :345
    std::optional<InertialBAProblemData> problem;
    auto phase1 = [&] { problem = collect_inertial_ba_data(map, kf_id, config); };   // :348-354
    if (lock_read) lock_read(phase1); else phase1();
    if (!problem) return std::nullopt;
    std::optional<InertialBAResultData> result = solve_inertial_ba(h, *problem, camera, config, should_stop);   // :357-360, no lock
    if (!result) return std::nullopt;
    if (inertial_result_out) *inertial_result_out = *result;
    size_t updated = 0;
    if (result->iterations > 0) {                                                    // :363
      auto phase3 = [&] { updated = apply_inertial_ba_results(map, *result); };
      if (lock_write) lock_write(phase3); else phase3();
    }
    return updated;
  }
  const LocalBAConfigLM config;                                                      // :380
  std::optional<VisualBAProblemData> problem;
  auto phase1 = [&] { problem = collect_visual_ba_data(map, kf_id, config); };       // :383-389
  if (lock_read) lock_read(phase1); else phase1();
  if (!problem) return std::nullopt;
  std::optional<VisualBAResultData> result = solve_visual_ba(h, *problem, camera, config, should_stop);   // :392-395, no lock
  if (!result) return std::nullopt;
  if (result_out) *result_out = *result;
  size_t updated = 0;
  if (result->iterations > 0) {                                                      // :396
    auto phase3 = [&] { updated = apply_visual_ba_results(map, *result); };
    if (lock_write) lock_write(phase3); else phase3();
  }
  return updated;
}

// ---- global BA (src/optimizer/global_ba.rs), the host phases around solve_global_ba ---------------------------------------------------
// PHASE 1, global_ba.rs:100-181.  ORDER: the reference walks map.keyframes() and map.map_points() — HashMap iterations — and sorts only the
// keyframe ids (:122); the map points keep the HashMap's order (:131-144), which is the parameter order of the solve.  Here: the snapshot's
// map-point order.  A map point is taken when it is not bad and ANY of its observers is a collected keyframe (:137); observations are walked
// keyframe by keyframe in ascending id, features in order (:154-170).
inline std::optional<GlobalBAProblemData> collect_global_ba_data(const MapSnapshot& m) {
  GlobalBAProblemData p;
  for (size_t k = 0; k < m.kf_ids.size(); ++k) {                                     // :108-114
    if (m.kf_bad[k]) continue;
    p.kf_ids.push_back(m.kf_ids[k]);
    p.kf_poses[m.kf_ids[k]] = se3_inverse(m.kf_pose((int)k));                        // T_cw
  }
  if (p.kf_ids.empty()) return std::nullopt;                                         // :116-118
  std::sort(p.kf_ids.begin(), p.kf_ids.end());                                       // :121
  p.fixed_kf_id = p.kf_ids.front();                                                  // :124
  const std::unordered_set<KeyFrameId> kf_set(p.kf_ids.begin(), p.kf_ids.end());
  for (size_t j = 0; j < m.mp_ids.size(); ++j) {                                     // :129-144
    if (m.mp_bad[j]) continue;
    bool has_valid_obs = false;
    for (int o = m.mp_obs_start[j]; o < m.mp_obs_start[j + 1] && !has_valid_obs; ++o) has_valid_obs = kf_set.count(m.mp_obs_kf_id[(size_t)o]) != 0;
    if (!has_valid_obs) continue;
    p.mp_ids.push_back(m.mp_ids[j]);
    p.mp_positions[m.mp_ids[j]] = {m.mp_pos[3 * j], m.mp_pos[3 * j + 1], m.mp_pos[3 * j + 2]};
  }
  if (p.mp_ids.empty()) return std::nullopt;                                         // :146-148
  const std::unordered_set<MapPointId> mp_set(p.mp_ids.begin(), p.mp_ids.end());
  for (KeyFrameId id : p.kf_ids) {                                                   // :153-170
    const int k = m.kf_index(id);
    if (k < 0) continue;
    const int s = m.kf_feat_start[(size_t)k], e = m.kf_feat_start[(size_t)k + 1];
    for (int f = s; f < e; ++f) {
      const int64_t mp_id = m.feat_mp_id[(size_t)f];
      if (mp_id < 0 || !mp_set.count((MapPointId)mp_id)) continue;
      if (f - s >= m.kf_n_keypoints[(size_t)k]) continue;                            // keypoints.get(feat_idx) is Err
      p.observations.push_back(GlobalBAObservation{id, (MapPointId)mp_id, {(double)m.feat_uv[2 * (size_t)f], (double)m.feat_uv[2 * (size_t)f + 1]}});
    }
  }
  if (p.observations.empty()) return std::nullopt;                                   // :172-174
  return p;
}

// PHASE 3, global_ba.rs:421-443: gone or bad entities are skipped silently (the fixed keyframe's unchanged pose counts as an update too)
inline size_t apply_global_ba_results(MapSnapshot& m, const GlobalBAResult& r) {
  size_t updated = 0;
  for (const auto& kv : r.optimized_poses) {
    const int k = m.kf_index(kv.first);
    if (k >= 0 && !m.kf_bad[(size_t)k]) {
      for (int i = 0; i < 4; ++i) m.kf_pose_wc[7 * (size_t)k + i] = kv.second.rotation[(size_t)i];
      for (int i = 0; i < 3; ++i) m.kf_pose_wc[7 * (size_t)k + 4 + i] = kv.second.translation[(size_t)i];
      ++updated;
    }
  }
  for (const auto& kv : r.optimized_points) {
    const int j = m.mp_index(kv.first);
    if (j >= 0 && !m.mp_bad[(size_t)j]) {
      for (int i = 0; i < 3; ++i) m.mp_pos[3 * (size_t)j + i] = kv.second[(size_t)i];
      ++updated;
    }
  }
  return updated;
}

// run_global_ba (global_ba.rs:450-500): collect under the read lock, solve with no lock (should_stop = the running flag cleared), apply under the
// write lock — unconditionally, unlike local BA (:486-489); `running` is set on entry and cleared on every way out (:456, :468, :479, :498).
inline std::optional<GlobalBAResult> run_global_ba(Handle& h, MapSnapshot& map, const CameraModel& camera, const GlobalBAConfig& config,
                                                   std::atomic<bool>& running,
                                                   const std::function<void(const std::function<void()>&)>& lock_read = nullptr,
                                                   const std::function<void(const std::function<void()>&)>& lock_write = nullptr) {
  running.store(true);
  std::optional<GlobalBAProblemData> problem;
  auto phase1 = [&] { problem = collect_global_ba_data(map); };
  if (lock_read) lock_read(phase1); else phase1();
  if (!problem) { running.store(false); return std::nullopt; }
  std::optional<GlobalBAResult> result = solve_global_ba(h, *problem, camera, config, [&] { return !running.load(); });
  if (!result) { running.store(false); return std::nullopt; }
  auto phase3 = [&] { apply_global_ba_results(map, *result); };
  if (lock_write) lock_write(phase3); else phase3();
  running.store(false);
  return result;
}

}  // namespace orbx
#endif  // ORBX_MAP_HPP
