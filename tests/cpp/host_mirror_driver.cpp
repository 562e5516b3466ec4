// Compiled-host test driver for include/orbx.hpp: runs StereoProcessor::process and solve_visual_ba through the
// C++ mirror of the reference's interface on inputs written by tests/test_cpp_host_mirror.py and writes the
// results back for comparison with the oracle.  usage: driver <in_dir> <out_dir>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "orbx.hpp"

static std::vector<uint8_t> slurp(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> b((size_t)n);
  if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) exit(2);
  fclose(f);
  return b;
}
template <class T> static void put(FILE* f, const T* p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string in = argv[1], out = argv[2];
  const orbx::CameraModel cam{458.654, 457.296, 367.215, 248.375, 0.11007};   // EuRoC cam0
  try {
    // ---- StereoProcessor::new + process (stereo.rs:37-66) -------------------------------------------------
    std::vector<uint8_t> s = slurp(in + "/stereo.bin");
    const int w = ((int*)s.data())[0], h = ((int*)s.data())[1], nfeat = ((int*)s.data())[2];
    const uint8_t* left = s.data() + 12;
    const uint8_t* right = left + (size_t)w * h;
    orbx::StereoProcessor sp = orbx::StereoProcessor::create(cam, nfeat);
    orbx::StereoFrame f = sp.process(left, (size_t)w, right, (size_t)w, w, h, 1403636579763555584ull);
    FILE* fo = fopen((out + "/stereo_out.bin").c_str(), "wb");
    const int counts[3] = {(int)f.left_features.keypoints.size(), (int)f.right_features.keypoints.size(), (int)f.matches_lr.size()};
    put(fo, counts, 3);
    put(fo, f.left_features.keypoints.data(), f.left_features.keypoints.size());
    put(fo, f.left_features.descriptors.data(), f.left_features.descriptors.size());
    put(fo, f.right_features.keypoints.data(), f.right_features.keypoints.size());
    put(fo, f.right_features.descriptors.data(), f.right_features.descriptors.size());
    put(fo, f.matches_lr.data(), f.matches_lr.size());
    for (size_t i = 0; i < f.points_cam.size(); ++i) {
      const uint8_t has = f.points_cam[i].has_value();
      const double z[3] = {0, 0, 0};
      put(fo, &has, 1);
      put(fo, has ? f.points_cam[i]->data() : z, 3);
    }
    fclose(fo);
    // descriptor_distance + cross-check matcher on the extracted descriptors
    const uint32_t d01 = orbx::descriptor_distance(sp.handle(), f.left_features.descriptors.data(), f.left_features.descriptors.data() + 32);
    std::vector<orbx::DMatch> cc = orbx::bf_match_crosscheck(sp.handle(), f.left_features.descriptors, f.right_features.descriptors);
    fo = fopen((out + "/match_out.bin").c_str(), "wb");
    const int nc = (int)cc.size();
    put(fo, &d01, 1); put(fo, &nc, 1); put(fo, cc.data(), cc.size());
    fclose(fo);

    // search_for_triangulation (triangulation.rs:401-527): left/right feature sets as two keyframes
    {
      const size_t n1 = f.left_features.keypoints.size(), n2 = f.right_features.keypoints.size();
      std::vector<uint8_t> mp1(n1), st1(n1), mp2(n2);
      for (size_t i = 0; i < n1; ++i) { mp1[i] = i % 3 == 0; st1[i] = f.points_cam[i].has_value(); }
      for (size_t i = 0; i < n2; ++i) mp2[i] = i % 4 == 0;
      orbx::SE3 p1, p2;
      p2.rotation = {0.9998000066665778, 0.0, 0.01999866669333308, 0.0};
      p2.translation = {0.11007, 0.01, 0.02};
      auto pairs = orbx::search_for_triangulation(sp.handle(), f.left_features, mp1, st1, f.right_features, mp2, p1, p2, cam, 50);
      fo = fopen((out + "/tri_out.bin").c_str(), "wb");
      const int np = (int)pairs.size();
      put(fo, &np, 1);
      for (auto& pr : pairs) { const int v[2] = {(int)pr.first, (int)pr.second}; put(fo, v, 2); }
      fclose(fo);
    }

    // fuse_search (search_in_neighbors.rs:273-343): a few synthetic map points in front of two keyframes = the two feature sets
    {
      std::vector<std::array<double, 3>> pos;
      std::vector<uint8_t> mpd;
      for (int i = 0; i < 200; ++i) {
        const orbx::KeyPoint& k = f.left_features.keypoints[(size_t)i * 3 % f.left_features.keypoints.size()];
        const double z = 4.0 + 0.05 * i;
        pos.push_back({(k.x - cam.cx) * z / cam.fx, (k.y - cam.cy) * z / cam.fy, z});
        const uint8_t* d = f.left_features.descriptors.data() + 32 * ((size_t)i * 3 % f.left_features.keypoints.size());
        mpd.insert(mpd.end(), d, d + 32);
      }
      orbx::SE3 p0, p1;
      p1.translation = {0.11007, 0.0, 0.0};
      auto idx = orbx::fuse_search(sp.handle(), pos, mpd, {{p0, &f.left_features}, {p1, &f.right_features}}, cam);
      fo = fopen((out + "/fuse_out.bin").c_str(), "wb");
      const int np = (int)idx.size();
      put(fo, &np, 1);
      put(fo, idx.data(), idx.size());
      for (auto& q : pos) put(fo, q.data(), 3);
      fclose(fo);
    }

    // EurocDataset (io/euroc.rs): when the test wrote a mav0 directory, frame 1 read through the mirror must process to the
    // same StereoFrame sizes as the raw images of stereo.bin (the test stores the same pair as frame 1)
    {
      FILE* fc = fopen((in + "/mav0/cam0/data.csv").c_str(), "rb");
      if (fc) {
        fclose(fc);
        orbx::EurocDataset ds(in + "/mav0");
        orbx::StereoImagePair pr = ds.stereo_pair(1);
        if (pr.width != w || pr.height != h || memcmp(pr.left.data(), left, (size_t)w * h) != 0 || memcmp(pr.right.data(), right, (size_t)w * h) != 0) {
          fprintf(stderr, "EurocDataset::stereo_pair differs from the raw images\n");
          return 5;
        }
        fo = fopen((out + "/euroc_out.bin").c_str(), "wb");
        const uint64_t meta[3] = {(uint64_t)ds.len(), *ds.frame_timestamp(1), pr.timestamp_ns};
        put(fo, meta, 3);
        const double cal[5] = {ds.camera().fx, ds.camera().fy, ds.camera().cx, ds.camera().cy, ds.camera().baseline};
        put(fo, cal, 5);
        fclose(fo);
      }
    }

    // OrbVocabulary::load_from_text + transform + search_for_triangulation_bow (vocabulary/mod.rs, triangulation.rs:541-658)
    {
      FILE* ft = fopen((in + "/voc.txt").c_str(), "rb");
      if (ft) {
        fclose(ft);
        orbx::OrbVocabulary voc = orbx::OrbVocabulary::load_from_text(sp.handle(), in + "/voc.txt");
        auto t1 = voc.transform(f.left_features.descriptors, 1);
        auto t2 = voc.transform(f.right_features.descriptors, 1);
        const size_t n1 = f.left_features.keypoints.size(), n2 = f.right_features.keypoints.size();
        std::vector<uint8_t> mp1(n1), st1(n1), mp2(n2);
        for (size_t i = 0; i < n1; ++i) { mp1[i] = i % 3 == 0; st1[i] = f.points_cam[i].has_value(); }
        for (size_t i = 0; i < n2; ++i) mp2[i] = i % 4 == 0;
        orbx::SE3 p1, p2;
        p2.rotation = {0.9998000066665778, 0.0, 0.01999866669333308, 0.0};
        p2.translation = {0.11007, 0.01, 0.02};
        auto pairs = orbx::search_for_triangulation_bow(sp.handle(), t1.second, t2.second, f.left_features, mp1, st1, f.right_features, mp2, p1, p2, cam, 50);
        fo = fopen((out + "/bow_out.bin").c_str(), "wb");
        const int hdr[4] = {(int)voc.num_nodes(), (int)voc.num_words(), (int)t1.first.size(), (int)pairs.size()};
        put(fo, hdr, 4);
        double bsum = 0.0;
        for (const auto& kv : t1.first) bsum += kv.second;
        put(fo, &bsum, 1);
        for (auto& pr : pairs) { const int v[2] = {(int)pr.first, (int)pr.second}; put(fo, v, 2); }
        fclose(fo);
      }
    }

    // ---- solve_visual_ba (local_ba_lm.rs:912-1098) through VisualBAProblemData keyed by ids -----------------
    std::vector<uint8_t> b = slurp(in + "/ba.bin");
    const int* hd = (const int*)b.data();
    const int K = hd[0], F = hd[1], M = hd[2], N = hd[3], stop_after = hd[4];
    const double* dp = (const double*)(b.data() + 32);
    const double* poses = dp; const double* fixed = poses + 7 * (size_t)K; const double* pts = fixed + 7 * (size_t)F;
    const double* ob = pts + 3 * (size_t)M;   // N x (kf_idx, fixed_idx, mp_idx, u, v) as doubles
    orbx::VisualBAProblemData prob;
    auto se3 = [](const double* p) { orbx::SE3 s; for (int i = 0; i < 4; ++i) s.rotation[i] = p[i]; for (int i = 0; i < 3; ++i) s.translation[i] = p[4 + i]; return s; };
    for (int k = 0; k < K; ++k) { const orbx::KeyFrameId id = 1000 + 7 * (uint64_t)k; prob.optimized_kf_ids.push_back(id); prob.local_kf_poses[id] = se3(poses + 7 * (size_t)k); }
    for (int k = 0; k < F; ++k) prob.fixed_kf_poses[10 + (uint64_t)k] = se3(fixed + 7 * (size_t)k);
    prob.anchor_kf_id = 10;
    for (int j = 0; j < M; ++j) { const orbx::MapPointId id = 5000 + 3 * (uint64_t)j; prob.mp_ids.push_back(id); prob.local_mp_positions[id] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]}; }
    for (int i = 0; i < N; ++i) {
      const double* o = ob + 5 * (size_t)i;
      const int kf = (int)o[0], fx = (int)o[1], mp = (int)o[2];
      orbx::VisualObservation v;
      v.is_kf_optimized = kf >= 0;
      v.kf_id = kf >= 0 ? 1000 + 7 * (uint64_t)kf : 10 + (uint64_t)fx;
      v.mp_id = 5000 + 3 * (uint64_t)mp;
      v.observed_uv = {o[3], o[4]};
      prob.observations.push_back(v);
    }
    int calls = 0;
    auto res = orbx::solve_visual_ba(sp.handle(), prob, cam, orbx::LocalBAConfigLM{}, [&]() { return stop_after >= 0 && calls++ >= stop_after; });
    fo = fopen((out + "/ba_out.bin").c_str(), "wb");
    const int ok = res.has_value();
    put(fo, &ok, 1);
    if (ok) {
      const int it = (int)res->iterations;
      put(fo, &it, 1);
      put(fo, &res->initial_error, 1); put(fo, &res->final_error, 1);
      for (orbx::KeyFrameId id : prob.optimized_kf_ids) { const orbx::SE3& p = res->optimized_poses.at(id); put(fo, p.rotation.data(), 4); put(fo, p.translation.data(), 3); }
      for (orbx::MapPointId id : prob.mp_ids) put(fo, res->optimized_points.at(id).data(), 3);
    }
    fclose(fo);
    // ---- solve_global_ba (global_ba.rs:184-418): same file, fixed keyframe = fixed[0], observations of the other
    // fixed keyframes left out
    {
      orbx::GlobalBAProblemData gp;
      gp.fixed_kf_id = 10;
      gp.kf_ids.push_back(10); gp.kf_poses[10] = se3(fixed);
      for (int k = 0; k < K; ++k) { const orbx::KeyFrameId id = 1000 + 7 * (uint64_t)k; gp.kf_ids.push_back(id); gp.kf_poses[id] = se3(poses + 7 * (size_t)k); }
      for (int j = 0; j < M; ++j) { const orbx::MapPointId id = 5000 + 3 * (uint64_t)j; gp.mp_ids.push_back(id); gp.mp_positions[id] = {pts[3 * (size_t)j], pts[3 * (size_t)j + 1], pts[3 * (size_t)j + 2]}; }
      for (int i = 0; i < N; ++i) {
        const double* o = ob + 5 * (size_t)i;
        const int kf = (int)o[0], fx = (int)o[1], mp = (int)o[2];
        if (kf < 0 && fx != 0) continue;
        gp.observations.push_back({kf >= 0 ? 1000 + 7 * (uint64_t)kf : 10, 5000 + 3 * (uint64_t)mp, {o[3], o[4]}});
      }
      // tolerances tightened so that the run ends on the iteration cap: with one fixed keyframe the monocular scale is
      // free, the default 1e-6 stop test lands within 3 % of its threshold on this input and two correct solvers may
      // disagree by one iteration
      orbx::GlobalBAConfig gcfg;
      gcfg.param_tolerance = 1e-9; gcfg.gradient_tolerance = 1e-9;
      auto gr = orbx::solve_global_ba(sp.handle(), gp, cam, gcfg, nullptr);
      fo = fopen((out + "/gba_out.bin").c_str(), "wb");
      const int gok = gr.has_value();
      put(fo, &gok, 1);
      if (gok) {
        const int it = (int)gr->iterations;
        put(fo, &it, 1);
        put(fo, &gr->initial_error, 1); put(fo, &gr->final_error, 1);
        for (orbx::KeyFrameId id : gp.kf_ids) { const orbx::SE3& p = gr->optimized_poses.at(id); put(fo, p.rotation.data(), 4); put(fo, p.translation.data(), 3); }
        for (orbx::MapPointId id : gp.mp_ids) put(fo, gr->optimized_points.at(id).data(), 3);
      }
      fclose(fo);
      orbx::GlobalBAProblemData one;                                      // a single keyframe is None (:194-196)
      one.kf_ids = {1}; one.mp_ids = {2}; one.fixed_kf_id = 1;
      if (orbx::solve_global_ba(sp.handle(), one, cam, orbx::GlobalBAConfig{}, nullptr).has_value()) return 4;
    }
    // ---- solve_inertial_ba (local_inertial_ba.rs:1074-1275) through InertialBAProblemData keyed by ids
    {
      FILE* fi = fopen((in + "/iba.bin").c_str(), "rb");
      if (fi) {
        fclose(fi);
        std::vector<uint8_t> ib = slurp(in + "/iba.bin");
        const int* ih = (const int*)ib.data();
        const int iK = ih[0], iF = ih[1], iM = ih[2], iN = ih[3], iE = ih[4];
        const double* q = (const double*)(ib.data() + 32);
        const double* ipose = q; const double* ivel = ipose + 7 * (size_t)iK; const double* ibias = ivel + 3 * (size_t)iK;
        const double* ifix = ibias + 6 * (size_t)iK; const double* ipts = ifix + 7 * (size_t)iF; const double* iobs = ipts + 3 * (size_t)iM;
        const double* iedge = iobs + 6 * (size_t)iN; const double* ipre = iedge + 2 * (size_t)iE;
        orbx::InertialBAProblemData ip;
        for (int k = 0; k < iK; ++k) {
          const orbx::KeyFrameId id = 700 + 13 * (uint64_t)k;
          ip.opt_kf_ids.push_back(id);
          ip.kf_poses[id] = se3(ipose + 7 * (size_t)k);
          ip.kf_velocities[id] = {ivel[3 * (size_t)k], ivel[3 * (size_t)k + 1], ivel[3 * (size_t)k + 2]};
          orbx::ImuBias b;
          for (int a = 0; a < 3; ++a) { b.gyro[a] = ibias[6 * (size_t)k + a]; b.accel[a] = ibias[6 * (size_t)k + 3 + a]; }
          ip.kf_biases[id] = b;
        }
        for (int f2 = 0; f2 < iF; ++f2) ip.fixed_kf_poses[20 + (uint64_t)f2] = se3(ifix + 7 * (size_t)f2);
        for (int j = 0; j < iM; ++j) { const orbx::MapPointId id = 8000 + 5 * (uint64_t)j; ip.mp_ids.push_back(id); ip.mp_positions[id] = {ipts[3 * (size_t)j], ipts[3 * (size_t)j + 1], ipts[3 * (size_t)j + 2]}; }
        for (int i = 0; i < iN; ++i) {
          const double* o = iobs + 6 * (size_t)i;
          const int kf = (int)o[0], fx = (int)o[1];
          ip.visual_observations.push_back({kf >= 0 ? 700 + 13 * (uint64_t)kf : 20 + (uint64_t)fx, 8000 + 5 * (uint64_t)(int)o[2], {o[4], o[5]}, o[3] != 0.0, kf >= 0});
        }
        for (int e = 0; e < iE; ++e) {
          orbx::ImuEdgeData ed;
          ed.kf_i_id = 700 + 13 * (uint64_t)(int)iedge[2 * (size_t)e]; ed.kf_j_id = 700 + 13 * (uint64_t)(int)iedge[2 * (size_t)e + 1];
          const double* pr = ipre + 11 * (size_t)e;
          for (int a = 0; a < 4; ++a) ed.preint.delta_rot[a] = pr[a];
          for (int a = 0; a < 3; ++a) { ed.preint.delta_vel[a] = pr[4 + a]; ed.preint.delta_pos[a] = pr[7 + a]; }
          ed.preint.dt = pr[10];
          ip.imu_edges.push_back(ed);
        }
        auto ir = orbx::solve_inertial_ba(sp.handle(), ip, cam, orbx::LocalInertialBAConfig{}, nullptr);
        fo = fopen((out + "/iba_out.bin").c_str(), "wb");
        const int iok = ir.has_value();
        put(fo, &iok, 1);
        if (iok) {
          const int it = (int)ir->iterations, nrep = (int)ir->optimized_poses.size();
          put(fo, &it, 1); put(fo, &nrep, 1);
          const int padi = 0; put(fo, &padi, 1);
          put(fo, &ir->initial_error, 1); put(fo, &ir->final_error, 1);
          for (int k = 1; k < iK; ++k) {
            const orbx::KeyFrameId id = ip.opt_kf_ids[(size_t)k];
            const orbx::SE3& p = ir->optimized_poses.at(id);
            put(fo, p.rotation.data(), 4); put(fo, p.translation.data(), 3);
            put(fo, ir->optimized_velocities.at(id).data(), 3);
            put(fo, ir->optimized_biases.at(id).gyro.data(), 3); put(fo, ir->optimized_biases.at(id).accel.data(), 3);
          }
          for (orbx::MapPointId id : ip.mp_ids) put(fo, ir->optimized_points.at(id).data(), 3);
        }
        fclose(fo);
      }
    }
    // an empty problem is None, as local_ba_lm.rs:923-925
    orbx::VisualBAProblemData empty;
    if (orbx::solve_visual_ba(sp.handle(), empty, cam, orbx::LocalBAConfigLM{}, nullptr).has_value()) return 3;
  } catch (const orbx::Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  printf("HOST_MIRROR_OK\n");
  return 0;
}
