// ba_batch_driver — orbx_ba_solve_visual_batch called the way a compiled host (the Rust crate behind INTEGRATION.md's shim) would call it:
// straight through the C ABI of include/orbx.h, no Python in the process.  Used by tests/test_ba_gpu.py (results equal the Python
// mirror's bit for bit) and by bench.py (`local_ba.batched.c_abi`: the call-level rate without the mirror's marshalling).
//
//   ba_batch_driver <batch.bin> <out.bin> <reps> <pinned|pageable|pinned32>
//
// batch.bin: int32 W, then per window int32 K, F, M, N | poses_cw [K][7] | fixed_poses_cw [F][7] | points [M][3] | obs [N] (orbx_ba_obs).
// out.bin:   per window int32 status, iterations | double initial_error, final_error | poses_wc [K][7] | points [M][3]   (last repetition)
// stdout:    one JSON line {"windows", "observations", "reps", "ms_per_call_median", "ms_per_call_min", "ms_per_call": [...], "obs_memory"}
//
// `pinned`: every window's observations are consecutive slices of ONE orbx_host_alloc buffer (the library's copy engine reads them where
// they lie, one copy per half of the batch); `pageable`: plain heap memory (the library stages them through its own pinned blob);
// `pinned32`: as pinned, in the 16-byte form orbx_ba_obs32 (the file's coordinates must be exactly representable as f32).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orbx.h"

static std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (n > 0 && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: ba_batch_driver batch.bin out.bin reps pinned|pageable\n"); return 2; }
  const std::vector<uint8_t> in = slurp(argv[1]);
  const int reps = std::max(1, atoi(argv[3]));
  const bool obs32 = std::string(argv[4]) == "pinned32";
  const bool pinned = std::string(argv[4]) == "pinned" || obs32;
  const uint8_t* p = in.data();
  int W;
  memcpy(&W, p, 4); p += 4;
  struct Win { int K, F, M, N; const double *poses, *fixed, *points; const orbx_ba_obs* obs; };
  std::vector<Win> src((size_t)W);
  size_t n_obs = 0, n_pts = 0, n_kf = 0;
  for (int w = 0; w < W; ++w) {
    Win& s = src[(size_t)w];
    int hd[4];
    memcpy(hd, p, 16); p += 16;
    s.K = hd[0]; s.F = hd[1]; s.M = hd[2]; s.N = hd[3];
    s.poses = (const double*)p; p += 56 * (size_t)s.K;
    s.fixed = (const double*)p; p += 56 * (size_t)s.F;
    s.points = (const double*)p; p += 24 * (size_t)s.M;
    s.obs = (const orbx_ba_obs*)p; p += sizeof(orbx_ba_obs) * (size_t)s.N;
    n_obs += (size_t)s.N; n_pts += (size_t)s.M; n_kf += (size_t)std::max(s.K, 1);
  }
  // the caller's own storage: observations (pinned or pageable), in/out points, output poses
  orbx_ba_obs* obs_all = pinned ? (orbx_ba_obs*)orbx_host_alloc(sizeof(orbx_ba_obs) * std::max<size_t>(n_obs, 1))
                                : (orbx_ba_obs*)malloc(sizeof(orbx_ba_obs) * std::max<size_t>(n_obs, 1));
  if (!obs_all) { fprintf(stderr, "allocation failed\n"); return 2; }
  orbx_ba_obs32* obs32_all = obs32 ? (orbx_ba_obs32*)orbx_host_alloc(sizeof(orbx_ba_obs32) * std::max<size_t>(n_obs, 1)) : nullptr;
  if (obs32 && !obs32_all) { fprintf(stderr, "allocation failed\n"); return 2; }
  std::vector<double> pts(3 * std::max<size_t>(n_pts, 1)), out_poses(7 * n_kf);
  std::vector<orbx_ba_window> wins((size_t)W);
  const orbx_camera cam{458.654, 457.296, 367.215, 248.375, 0.11007};        // EuRoC cam0
  orbx_ba_config cfg;
  orbx_default_ba_config(&cfg);
  orbx_orb_params orb;
  orbx_default_orb_params(100, &orb);
  orbx_handle* h = nullptr;
  if (orbx_create(&cam, &orb, 0, 752, 480, 1, &h) != ORBX_OK) { fprintf(stderr, "orbx_create failed\n"); return 3; }
  std::vector<double> ms;
  int rc = ORBX_OK;
  for (int r = 0; r < reps + 1 && rc == ORBX_OK; ++r) {                      // one untimed call first (workspaces, worker pool, second stream)
    size_t oo = 0, op = 0, ok = 0;
    for (int w = 0; w < W; ++w) {                                           // (the in/out points are the caller's to refresh: part of its own bookkeeping, not of the call)
      const Win& s = src[(size_t)w];
      if (r == 0) memcpy(obs_all + oo, s.obs, sizeof(orbx_ba_obs) * (size_t)s.N);
      if (r == 0 && obs32)
        for (int i = 0; i < s.N; ++i) {
          const orbx_ba_obs& o = s.obs[i];
          orbx_ba_obs32& q32 = obs32_all[oo + (size_t)i];
          q32.kf_idx = o.kf_idx >= 0 ? o.kf_idx : -1 - (o.fixed_idx >= 0 ? o.fixed_idx : s.F);
          q32.mp_idx = o.mp_idx; q32.u = (float)o.u; q32.v = (float)o.v;
          if ((double)q32.u != o.u || (double)q32.v != o.v) { fprintf(stderr, "pinned32: a coordinate is not an f32\n"); return 5; }
        }
      memcpy(&pts[3 * op], s.points, 24 * (size_t)s.M);
      orbx_ba_window& q = wins[(size_t)w];
      memset(&q, 0, sizeof(q));
      q.K = s.K; q.poses_cw = s.poses; q.F = s.F; q.fixed_poses_cw = s.fixed; q.M = s.M; q.points = &pts[3 * op];
      q.N = s.N; q.obs = obs32 ? nullptr : obs_all + oo; q.obs32 = obs32 ? obs32_all + oo : nullptr; q.poses_wc_out = &out_poses[7 * ok];
      oo += (size_t)s.N; op += (size_t)s.M; ok += (size_t)std::max(s.K, 1);
    }
    const auto t0 = std::chrono::steady_clock::now();
    rc = orbx_ba_solve_visual_batch(h, &cam, &cfg, W, wins.data(), nullptr, nullptr);
    const auto t1 = std::chrono::steady_clock::now();
    if (r > 0) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
  }
  if (rc != ORBX_OK) { fprintf(stderr, "orbx_ba_solve_visual_batch: %d %s\n", rc, orbx_last_error(h)); orbx_destroy(h); return 4; }
  FILE* fo = fopen(argv[2], "wb");
  if (!fo) { fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
  for (int w = 0; w < W; ++w) {
    const orbx_ba_window& q = wins[(size_t)w];
    const int hd[2] = {q.status, q.iterations};
    const double er[2] = {q.initial_error, q.final_error};
    fwrite(hd, 4, 2, fo); fwrite(er, 8, 2, fo);
    fwrite(q.poses_wc_out, 56, (size_t)q.K, fo);
    fwrite(q.points, 24, (size_t)q.M, fo);
  }
  fclose(fo);
  std::vector<double> sorted = ms;
  std::sort(sorted.begin(), sorted.end());
  printf("{\"windows\": %d, \"observations\": %zu, \"reps\": %d, \"ms_per_call_median\": %.4f, \"ms_per_call_min\": %.4f, \"ms_per_call\": [", W, n_obs, reps,
         sorted[sorted.size() / 2], sorted[0]);
  for (size_t i = 0; i < ms.size(); ++i) printf("%s%.4f", i ? ", " : "", ms[i]);
  printf("], \"obs_memory\": \"%s\"}\n", obs32 ? "pinned (orbx_host_alloc), one buffer, orbx_ba_obs32 (16 B per observation)" : pinned ? "pinned (orbx_host_alloc), one buffer" : "pageable (malloc)");
  orbx_destroy(h);
  if (pinned) orbx_host_free(obs_all); else free(obs_all);
  if (obs32_all) orbx_host_free(obs32_all);
  return 0;
}
