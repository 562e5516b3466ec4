// Compiled-host test driver for include/orbx_map.hpp (the host side of local BA over flat arrays): reads a MapSnapshot
// written by api.MapSnapshot.to_bytes(), runs collect_visual_ba_data (phase 1) and — mode "apply" — apply_visual_ba_results
// (phase 3) on a result file, mode "lba": the whole three-phase local_bundle_adjustment on the GPU.
//   driver collect <snapshot.bin> <current_kf_id> <max_covisible> <out.bin>
//   driver apply   <snapshot.bin> <result.bin> <out.bin>
//   driver lba     <snapshot.bin> <current_kf_id> <stop_after_polls> <out.bin>     (the branch follows the snapshot's imu_initialized)
//   driver icollect <snapshot.bin> <current_kf_id> <window_size> <out.bin>         collect_inertial_ba_data (phase 1 of the inertial branch)
//   driver iapply   <snapshot.bin> <result.bin> <out.bin>                          apply_inertial_ba_results
//   driver gcollect <snapshot.bin> 0 0 <out.bin>                                   collect_global_ba_data
//   driver gapply   <snapshot.bin> <result.bin> <out.bin>                          apply_global_ba_results (result file as for `apply`)
//   driver gba      <snapshot.bin> <stop_after_polls> 0 <out.bin>                  run_global_ba on the GPU
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "orbx_map.hpp"

static std::vector<uint8_t> slurp(const char* p) {
  FILE* f = fopen(p, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", p); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}
template <class T> static void take(const uint8_t*& p, std::vector<T>& v, uint64_t n) { v.resize(n); if (n) memcpy(v.data(), p, n * sizeof(T)); p += n * sizeof(T); }
template <class T> static void put(FILE* f, const T* p, size_t n) { if (n) fwrite(p, sizeof(T), n, f); }

static orbx::MapSnapshot load(const char* path) {
  const std::vector<uint8_t> b = slurp(path);
  const uint64_t* c = (const uint64_t*)b.data();
  const uint8_t* p = b.data() + 22 * 8;                                    // 21 element counts + imu_initialized (api.MapSnapshot.to_bytes)
  orbx::MapSnapshot m;
  take(p, m.kf_ids, c[0]); take(p, m.kf_bad, c[1]); take(p, m.kf_pose_wc, c[2]); take(p, m.kf_n_keypoints, c[3]);
  take(p, m.kf_feat_start, c[4]); take(p, m.feat_mp_id, c[5]); take(p, m.feat_uv, c[6]); take(p, m.cov_start, c[7]);
  take(p, m.cov_kf_id, c[8]); take(p, m.mp_ids, c[9]); take(p, m.mp_bad, c[10]); take(p, m.mp_pos, c[11]);
  take(p, m.mp_obs_start, c[12]); take(p, m.mp_obs_kf_id, c[13]);
  take(p, m.kf_prev_id, c[14]); take(p, m.kf_velocity, c[15]); take(p, m.kf_bias, c[16]); take(p, m.kf_has_preint, c[17]);
  take(p, m.kf_preint, c[18]); take(p, m.feat_stereo, c[19]); take(p, m.mp_obs_feat_idx, c[20]);
  m.imu_initialized = c[21] != 0;
  m.build_index();
  return m;
}

// InertialBAProblemData: [some] then [n_opt, n_mp, n_obs, n_edges, n_fixed] | opt ids | mp ids | obs (kf, mp, stereo, in_window, u, v) |
// edges (kf_i, kf_j, preint[11]) | per opt id (present, pose7, vel3, bias6) | fixed poses sorted by id (id, pose7) | per mp id (present, xyz)
static void write_inertial_problem(FILE* f, const std::optional<orbx::InertialBAProblemData>& p) {
  const uint64_t some = p ? 1 : 0;
  put(f, &some, 1);
  if (!p) return;
  const uint64_t hdr[5] = {p->opt_kf_ids.size(), p->mp_ids.size(), p->visual_observations.size(), p->imu_edges.size(), p->fixed_kf_poses.size()};
  put(f, hdr, 5);
  put(f, p->opt_kf_ids.data(), p->opt_kf_ids.size());
  put(f, p->mp_ids.data(), p->mp_ids.size());
  for (const auto& o : p->visual_observations) {
    const uint64_t ids[4] = {o.kf_id, o.mp_id, o.is_stereo ? 1u : 0u, o.is_kf_in_window ? 1u : 0u};
    put(f, ids, 4); put(f, o.observed_uv.data(), 2);
  }
  for (const auto& e : p->imu_edges) {
    const uint64_t ids[2] = {e.kf_i_id, e.kf_j_id};
    put(f, ids, 2); put(f, e.preint.delta_rot.data(), 4); put(f, e.preint.delta_vel.data(), 3); put(f, e.preint.delta_pos.data(), 3); put(f, &e.preint.dt, 1);
  }
  for (uint64_t id : p->opt_kf_ids) {
    auto it = p->kf_poses.find(id);
    const uint64_t present = it != p->kf_poses.end();
    put(f, &present, 1);
    const orbx::SE3 z{};
    const orbx::SE3& s = present ? it->second : z;
    put(f, s.rotation.data(), 4); put(f, s.translation.data(), 3);
    const std::array<double, 3> z3{0, 0, 0};
    auto v = p->kf_velocities.find(id);
    put(f, v != p->kf_velocities.end() ? v->second.data() : z3.data(), 3);
    auto bb = p->kf_biases.find(id);
    put(f, bb != p->kf_biases.end() ? bb->second.gyro.data() : z3.data(), 3);
    put(f, bb != p->kf_biases.end() ? bb->second.accel.data() : z3.data(), 3);
  }
  std::vector<uint64_t> fid;
  for (const auto& kv : p->fixed_kf_poses) fid.push_back(kv.first);
  std::sort(fid.begin(), fid.end());
  for (uint64_t id : fid) { const orbx::SE3& s = p->fixed_kf_poses.at(id); put(f, &id, 1); put(f, s.rotation.data(), 4); put(f, s.translation.data(), 3); }
  for (uint64_t id : p->mp_ids) {
    auto it = p->mp_positions.find(id);
    const uint64_t present = it != p->mp_positions.end();
    const std::array<double, 3> z{0, 0, 0};
    put(f, &present, 1); put(f, present ? it->second.data() : z.data(), 3);
  }
}

static void write_problem(FILE* f, const std::optional<orbx::VisualBAProblemData>& p) {
  const uint64_t some = p ? 1 : 0;
  put(f, &some, 1);
  if (!p) return;
  const uint64_t hdr[5] = {p->anchor_kf_id, p->optimized_kf_ids.size(), p->mp_ids.size(), p->observations.size(), p->fixed_kf_poses.size()};
  put(f, hdr, 5);
  put(f, p->optimized_kf_ids.data(), p->optimized_kf_ids.size());
  put(f, p->mp_ids.data(), p->mp_ids.size());
  for (const auto& o : p->observations) {
    const uint64_t ids[3] = {o.kf_id, o.mp_id, o.is_kf_optimized ? 1u : 0u};
    put(f, ids, 3); put(f, o.observed_uv.data(), 2);
  }
  auto pose = [&](uint64_t id, const orbx::SE3& s, uint64_t present) {
    put(f, &id, 1); put(f, &present, 1); put(f, s.rotation.data(), 4); put(f, s.translation.data(), 3);
  };
  for (uint64_t id : p->optimized_kf_ids) {                              // local poses in optimized order (missing = absent)
    auto it = p->local_kf_poses.find(id);
    pose(id, it != p->local_kf_poses.end() ? it->second : orbx::SE3{}, it != p->local_kf_poses.end());
  }
  std::vector<uint64_t> fid;
  for (const auto& kv : p->fixed_kf_poses) fid.push_back(kv.first);
  std::sort(fid.begin(), fid.end());
  for (uint64_t id : fid) pose(id, p->fixed_kf_poses.at(id), 1);
  for (uint64_t id : p->mp_ids) {
    auto it = p->local_mp_positions.find(id);
    const uint64_t present = it != p->local_mp_positions.end();
    const std::array<double, 3> z{0, 0, 0};
    put(f, &id, 1); put(f, &present, 1); put(f, present ? it->second.data() : z.data(), 3);
  }
}

// GlobalBAProblemData: [some] then [fixed id, n_kf, n_mp, n_obs] | kf ids | mp ids | obs (kf, mp, u, v) | per kf id (id, present, pose7) | per mp id (id, present, xyz)
static void write_global_problem(FILE* f, const std::optional<orbx::GlobalBAProblemData>& p) {
  const uint64_t some = p ? 1 : 0;
  put(f, &some, 1);
  if (!p) return;
  const uint64_t hdr[4] = {p->fixed_kf_id, p->kf_ids.size(), p->mp_ids.size(), p->observations.size()};
  put(f, hdr, 4);
  put(f, p->kf_ids.data(), p->kf_ids.size());
  put(f, p->mp_ids.data(), p->mp_ids.size());
  for (const auto& o : p->observations) {
    const uint64_t ids[2] = {o.kf_id, o.mp_id};
    put(f, ids, 2); put(f, o.observed_uv.data(), 2);
  }
  for (uint64_t id : p->kf_ids) {
    auto it = p->kf_poses.find(id);
    const uint64_t present = it != p->kf_poses.end();
    const orbx::SE3 s = present ? it->second : orbx::SE3{};
    put(f, &id, 1); put(f, &present, 1); put(f, s.rotation.data(), 4); put(f, s.translation.data(), 3);
  }
  for (uint64_t id : p->mp_ids) {
    auto it = p->mp_positions.find(id);
    const uint64_t present = it != p->mp_positions.end();
    const std::array<double, 3> z{0, 0, 0};
    put(f, &id, 1); put(f, &present, 1); put(f, present ? it->second.data() : z.data(), 3);
  }
}

static void write_map_state(FILE* f, const orbx::MapSnapshot& m) {
  put(f, m.kf_pose_wc.data(), m.kf_pose_wc.size());
  put(f, m.mp_pos.data(), m.mp_pos.size());
  put(f, m.kf_velocity.data(), m.kf_velocity.size());
  put(f, m.kf_bias.data(), m.kf_bias.size());
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const std::string mode = argv[1];
  try {
    orbx::MapSnapshot m = load(argv[2]);
    if (mode == "collect") {
      orbx::LocalBAConfigLM cfg;
      cfg.max_covisible_keyframes = atoi(argv[4]);
      FILE* f = fopen(argv[5], "wb");
      write_problem(f, orbx::collect_visual_ba_data(m, strtoull(argv[3], nullptr, 10), cfg));
      fclose(f);
    } else if (mode == "apply") {
      const std::vector<uint8_t> b = slurp(argv[3]);                       // [nk, nm] then (id, pose7)*, (id, xyz)*
      const uint64_t* c = (const uint64_t*)b.data();
      const uint8_t* p = b.data() + 16;
      orbx::VisualBAResultData r;
      for (uint64_t i = 0; i < c[0]; ++i) {
        uint64_t id; double v[7];
        memcpy(&id, p, 8); memcpy(v, p + 8, 56); p += 64;
        orbx::SE3 s; for (int q = 0; q < 4; ++q) s.rotation[q] = v[q]; for (int q = 0; q < 3; ++q) s.translation[q] = v[4 + q];
        r.optimized_poses[id] = s;
      }
      for (uint64_t i = 0; i < c[1]; ++i) {
        uint64_t id; double v[3];
        memcpy(&id, p, 8); memcpy(v, p + 8, 24); p += 32;
        r.optimized_points[id] = {v[0], v[1], v[2]};
      }
      const uint64_t updated = orbx::apply_visual_ba_results(m, r);
      FILE* f = fopen(argv[4], "wb");
      put(f, &updated, 1);
      write_map_state(f, m);
      fclose(f);
    } else if (mode == "icollect") {
      orbx::LocalInertialBAConfig cfg;
      cfg.window_size = atoi(argv[4]);
      FILE* f = fopen(argv[5], "wb");
      write_inertial_problem(f, orbx::collect_inertial_ba_data(m, strtoull(argv[3], nullptr, 10), cfg));
      fclose(f);
    } else if (mode == "iapply") {
      const std::vector<uint8_t> b = slurp(argv[3]);                       // [nk, nm] then (id, pose7, vel3, bias6)*, (id, xyz)*
      const uint64_t* c = (const uint64_t*)b.data();
      const uint8_t* p = b.data() + 16;
      orbx::InertialBAResultData r;
      for (uint64_t i = 0; i < c[0]; ++i) {
        uint64_t id; double v[16];
        memcpy(&id, p, 8); memcpy(v, p + 8, 128); p += 136;
        orbx::SE3 s; for (int q = 0; q < 4; ++q) s.rotation[q] = v[q]; for (int q = 0; q < 3; ++q) s.translation[q] = v[4 + q];
        r.optimized_poses[id] = s;
        r.optimized_velocities[id] = {v[7], v[8], v[9]};
        orbx::ImuBias bb; for (int q = 0; q < 3; ++q) { bb.gyro[q] = v[10 + q]; bb.accel[q] = v[13 + q]; }
        r.optimized_biases[id] = bb;
      }
      for (uint64_t i = 0; i < c[1]; ++i) {
        uint64_t id; double v[3];
        memcpy(&id, p, 8); memcpy(v, p + 8, 24); p += 32;
        r.optimized_points[id] = {v[0], v[1], v[2]};
      }
      const uint64_t updated = orbx::apply_inertial_ba_results(m, r);
      FILE* f = fopen(argv[4], "wb");
      put(f, &updated, 1);
      write_map_state(f, m);
      fclose(f);
    } else if (mode == "gcollect") {
      FILE* f = fopen(argv[5], "wb");
      write_global_problem(f, orbx::collect_global_ba_data(m));
      fclose(f);
    } else if (mode == "gapply") {
      const std::vector<uint8_t> b = slurp(argv[3]);                       // [nk, nm] then (id, pose7)*, (id, xyz)*
      const uint64_t* c = (const uint64_t*)b.data();
      const uint8_t* p = b.data() + 16;
      orbx::GlobalBAResult r;
      for (uint64_t i = 0; i < c[0]; ++i) {
        uint64_t id; double v[7];
        memcpy(&id, p, 8); memcpy(v, p + 8, 56); p += 64;
        orbx::SE3 s; for (int q = 0; q < 4; ++q) s.rotation[q] = v[q]; for (int q = 0; q < 3; ++q) s.translation[q] = v[4 + q];
        r.optimized_poses[id] = s;
      }
      for (uint64_t i = 0; i < c[1]; ++i) {
        uint64_t id; double v[3];
        memcpy(&id, p, 8); memcpy(v, p + 8, 24); p += 32;
        r.optimized_points[id] = {v[0], v[1], v[2]};
      }
      const uint64_t updated = orbx::apply_global_ba_results(m, r);
      FILE* f = fopen(argv[4], "wb");
      put(f, &updated, 1);
      write_map_state(f, m);
      fclose(f);
    } else if (mode == "gba") {
      const orbx::CameraModel cam{458.654, 457.296, 367.215, 248.375, 0.11007};   // EuRoC cam0
      orbx::Handle h(cam, 100, 0, 752, 480, 1);
      std::atomic<bool> running{false};
      int locks[2] = {0, 0};
      const std::optional<orbx::GlobalBAResult> res = orbx::run_global_ba(
          h, m, cam, orbx::GlobalBAConfig{}, running,
          [&](const std::function<void()>& body) { ++locks[0]; body(); }, [&](const std::function<void()>& body) { ++locks[1]; body(); });
      FILE* f = fopen(argv[5], "wb");
      const int64_t hdr[5] = {res ? 1 : 0, res ? (int64_t)res->iterations : -1, locks[0], locks[1], running.load() ? 1 : 0};
      put(f, hdr, 5);
      const double err[2] = {res ? res->initial_error : 0.0, res ? res->final_error : 0.0};
      put(f, err, 2);
      write_map_state(f, m);
      fclose(f);
    } else if (mode == "lba") {
      const orbx::CameraModel cam{458.654, 457.296, 367.215, 248.375, 0.11007};   // EuRoC cam0
      orbx::Handle h(cam, 100, 0, 752, 480, 1);
      const int stop_after = atoi(argv[4]);                               // should_stop() turns true on this poll (0 = never)
      int polls = 0;
      orbx::VisualBAResultData res;
      orbx::InertialBAResultData ires;
      int locks[2] = {0, 0};
      const std::optional<size_t> up = orbx::local_bundle_adjustment(
          h, m, strtoull(argv[3], nullptr, 10), cam, [&] { ++polls; return stop_after > 0 && polls >= stop_after; },
          [&](const std::function<void()>& body) { ++locks[0]; body(); }, [&](const std::function<void()>& body) { ++locks[1]; body(); }, &res, &ires);
      FILE* f = fopen(argv[5], "wb");
      const bool inertial = m.imu_initialized != 0;
      const int64_t hdr[5] = {up ? (int64_t)*up : -1, (int64_t)(inertial ? ires.iterations : res.iterations), locks[0], locks[1], polls};
      put(f, hdr, 5);
      const double err[2] = {inertial ? ires.initial_error : res.initial_error, inertial ? ires.final_error : res.final_error};
      put(f, err, 2);
      write_map_state(f, m);
      fclose(f);
    } else return 2;
  } catch (const std::exception& e) {
    fprintf(stderr, "driver: %s\n", e.what());
    return 1;
  }
  return 0;
}
