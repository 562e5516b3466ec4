/*
 * oracle.h — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under oracle/ is part of the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it, and only as the checker.  The product library
 * (orb-slam3-rust_amd/csrc) neither links nor includes anything from here except the
 * public ABI structs of include/orbx.h, which this header re-uses for data layout.
 *
 * Parity status:
 *   - Hamming distance, stereo matcher, triangulation, local BA: restated from the in-repo
 *     Rust (file:line cited at each function) and PINNED by the reference's own known-answer
 *     tests (vocabulary/mod.rs:429-441, corrector.rs:625-634, local_ba_lm.rs:1144-1243) plus
 *     the golden values of SURVEY Appendix D derived from the reference's formulas.
 *   - ORB extraction and the cross-check BF matcher live in OpenCV C++ (system library, version
 *     not pinned by the reference, source absent here, no reference test touches them):
 *     PARITY UNPINNED.  orb_ref.cpp is a written specification; the HIP
 *     kernels are held bit-exact to it, and it is "algorithmically equivalent to cv::ORB",
 *     not "bit-exact to OpenCV".
 */
#ifndef ORBX_ORACLE_H
#define ORBX_ORACLE_H

#include "../include/orbx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- matcher (match_ref.cpp) ---- */
uint32_t oracle_hamming256(const uint8_t* a, const uint8_t* b);
void oracle_hamming_batch(const uint8_t* a, const uint8_t* b, int n, uint32_t* out);
void oracle_disparity_bounds(const orbx_camera* cam, float* max_disp, float* min_disp);
int oracle_stereo_match(const orbx_camera* cam, const orbx_keypoint* kpL, const uint8_t* descL,
                        int nL, const orbx_keypoint* kpR, const uint8_t* descR, int nR,
                        orbx_dmatch* matches, double* points_cam, uint8_t* has_point);
int oracle_crosscheck_match(const uint8_t* q, int nq, const uint8_t* t, int nt, orbx_dmatch* out);
void oracle_guided_match(const orbx_keypoint* kp, const uint8_t* desc, int n, double img_w, double img_h,
                         const double* q_uv, const uint8_t* q_desc, int nq, double radius, int mode,
                         int* out_idx, uint32_t* out_dist);

void oracle_triangulation_geometry(const orbx_camera* cam, const double* pose1_wc, const double* pose2_wc,
                                   double* epipole2, double* F9);
int oracle_search_for_triangulation(const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1,
                                    const uint8_t* mp1, const uint8_t* stereo1, int n1, const orbx_keypoint* kp2,
                                    const uint8_t* desc2, const uint8_t* mp2, int n2, const double* pose1_wc,
                                    const double* pose2_wc, unsigned max_dist, int* out_pairs);

void oracle_fuse_search(const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                        const double* kf_poses_wc, const int* kf_feat_offset, const orbx_keypoint* kps,
                        const uint8_t* descs, int T, double radius_scale, unsigned desc_threshold, int* out_idx,
                        uint32_t* out_dist);

/* vocabulary (src/vocabulary/mod.rs): loader of the DBoW2 text format, per-descriptor tree descent */
void* oracle_vocab_load_text(const char* path);
void* oracle_vocab_from_arrays(int n_nodes, const uint32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                               const double* weight, int k, int l);
void oracle_vocab_free(void* v);
void oracle_vocab_info(void* v, int* k, int* l, int* n_nodes, int* n_words);
void oracle_vocab_arrays(void* v, uint32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight);
void oracle_bow_transform(void* v, const uint8_t* desc, int n, int levels_up, uint32_t* word, uint32_t* leaf, uint32_t* node,
                          double* weight);
int oracle_search_for_triangulation_bow(const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1,
                                        const uint8_t* mp1, const uint8_t* stereo1, const uint32_t* node1, int n1,
                                        const orbx_keypoint* kp2, const uint8_t* desc2, const uint8_t* mp2,
                                        const uint32_t* node2, int n2, const double* pose1_wc, const double* pose2_wc,
                                        unsigned max_dist, int* out_pairs);

/* ---- ORB extractor (orb_ref.cpp) ---- */
typedef struct {
  int n_levels;
  int w[8], h[8];
  float scale[8];
  int quota[8];
} oracle_orb_levels;
int oracle_orb_level_table(int w, int h, const orbx_orb_params* p, oracle_orb_levels* out);
void oracle_orb_umax(int* umax16);
/* stage outputs, for stage-by-stage parity tests of the kernels */
int oracle_orb_pyramid_level(const uint8_t* img, size_t stride, int w, int h,
                             const orbx_orb_params* p, int level, uint8_t* out /* w_l*h_l */);
int oracle_orb_blur_level(const uint8_t* img, size_t stride, int w, int h,
                          const orbx_orb_params* p, int level, uint8_t* out);
/* FAST+NMS+border candidates of one level: packed (score<<24 | y<<12 | x), sorted ascending
 * by (y,x); returns count (or -needed if cap too small). */
int oracle_orb_fast_level(const uint8_t* img, size_t stride, int w, int h,
                          const orbx_orb_params* p, int level, uint32_t* out, int cap);
float oracle_fast_atan2(float y, float x);
void oracle_sincos_deg(float angle_deg, float* c, float* s);
/* full extractor: returns number of keypoints, or -(needed) if cap is too small */
int oracle_orb_extract(const uint8_t* img, size_t stride, int w, int h, const orbx_orb_params* p,
                       orbx_keypoint* kp, uint8_t* desc, int cap);

/* ---- local BA (ba_ref.cpp) ---- */
void oracle_se3_to_params(const double* pose7, double* params6);
void oracle_se3_from_params(const double* params6, double* pose7);
void oracle_se3_inverse(const double* pose7, double* out7);
/* one observation: unweighted error e, sqrt(w), weighted residual r, J_pose (2x6 row-major,
 * already * sqrt(w)), J_point (2x3 row-major, * sqrt(w)) */
void oracle_ba_obs_terms(const orbx_camera* cam, double huber, const double* pose7,
                         const double* X, double u, double v, double* e2, double* sqrt_w,
                         double* r2, double* Jpose12, double* Jpoint6);
/* literal restatement: dense J, dense JtJ, partial-pivot LU (local_ba_lm.rs:912-1098) */
int oracle_ba_solve_dense(const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                          const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                          double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error,
                          double* final_error, double* trace /* optional [max_it*4] */);
/* structured restatement: block normal equations + Schur complement + Cholesky */
int oracle_ba_solve_schur(const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                          const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                          double* points, int N, const orbx_ba_obs* obs, int stop_after,
                          double* poses_wc_out, int* iterations, double* initial_error,
                          double* final_error, double* trace);
/* reduced system of one linearisation (for the multi-GPU partition test): S [6K*6K] row-major,
 * bs [6K], chi2; contributions of the given observations only, no damping of U when
 * `add_pose_damping` == 0 (rank partial) */
/* solve_global_ba (global_ba.rs:184-418): F = 1 (the anchor), Jacobian rows zero where z_c <= 0.001 (:561-563) */
int oracle_global_ba_solve_dense(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                                 const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs,
                                 int stop_after, double* poses_wc_out, int* iterations, double* initial_error,
                                 double* final_error, double* trace);
int oracle_global_ba_solve_schur(const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                                 const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs,
                                 int stop_after, double* poses_wc_out, int* iterations, double* initial_error,
                                 double* final_error, double* trace);
/* solve_inertial_ba (local_inertial_ba.rs:1074-1275), inertial_ba_ref.cpp */
void oracle_inertial_imu_residual(const double* state_i9, const double* state_j9, const double* preint11, double* r9);
int oracle_inertial_ba_solve(const orbx_camera* cam, const orbx_inertial_ba_config* cfg, int K, const double* poses_wc,
                             const double* velocities, const double* biases, int F, const double* fixed_poses_cw, int M,
                             double* points, int N, const orbx_ba_obs* obs, int E, const int* edge_kf, const double* preint,
                             int stop_after, double* poses_wc_out, double* vel_out, double* bias_out, int* iterations,
                             double* initial_error, double* final_error, double* trace);
int oracle_ba_reduced_system(const orbx_camera* cam, const orbx_ba_config* cfg, double lambda,
                             int K, const double* params_pose /*6K*/, int F,
                             const double* fixed_poses_cw, int M, const double* points, int N,
                             const orbx_ba_obs* obs, double* U /*K*36*/, double* gp /*6K*/,
                             double* Sred /*6K*6K: sum_j W V*^-1 W^T*/,
                             double* bred /*6K: sum_j W V*^-1 g_l*/, double* chi2);

#ifdef __cplusplus
}
#endif
#endif
