/*
 * orbx.h — C ABI of the MI355X-native hot path of jurmy24/orb-slam3-rust.
 *
 * This is the drop-in boundary.  The reference has no FFI layer of
 * its own: the path sits behind ordinary Rust `pub fn`s whose arguments are OpenCV
 * wrapper types.  Each entry point below names the reference interface it replaces
 * (file:line relative to the reference crate root); INTEGRATION.md shows the Rust
 * `extern "C"` block and the shim that re-creates the original signatures on top.
 *
 * Conventions
 *   - return 0 (ORBX_OK) on success, a negative ORBX_ERR_* otherwise;
 *     orbx_last_error(h) gives a human-readable message for the last failure
 *     (maps to `anyhow::Error` / `None` on the Rust side).
 *   - plain pointers and sizes only; no C++ or torch types.
 *   - the caller owns every buffer; capacities go in, counts come out.
 *     A result that does not fit its capacity is an error (ORBX_ERR_CAPACITY),
 *     never a silent truncation.
 *   - a handle is NOT thread-safe (mirrors `&mut self`, stereo.rs:52); use one
 *     handle per thread.  Each handle owns one HIP stream on its device.
 *   - `*_device` entry points take device pointers (memory of the handle's GPU)
 *     and are asynchronous on the handle's stream unless stated.  That stream is
 *     non-blocking: inputs produced on another stream must be complete (or ordered
 *     with an event against orbx_stream()) before the call, and must stay allocated
 *     until the work has run; the others take
 *     host pointers, copy in/out, and return when the results are in the buffers.
 *   - there is NO CPU fallback: every entry point fails with ORBX_ERR_NO_DEVICE
 *     when no gfx950 device can be opened (the host-only orbx_euroc_* / orbx_png_*
 *     input functions need no device).
 *   - two environment switches, both read at call time and both without effect on results:
 *     ORBX_NO_GRAPH=1 keeps orbx_process_stereo on eager launches instead of a captured hipGraph;
 *     ORBX_FORK_BLUR=1 lets batches of >= 16 images run the blur on a second stream beside the
 *     FAST -> Harris -> ordering chain (a few per cent faster; off by default because it makes
 *     per-kernel timings describe the overlap rather than the kernels).
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORBX_OK = 0,
  ORBX_ERR_INVALID = -1,   /* bad argument (null pointer, size out of range)        */
  ORBX_ERR_NO_DEVICE = -2, /* no usable HIP device / wrong architecture             */
  ORBX_ERR_HIP = -3,       /* a HIP runtime call failed (see orbx_last_error)       */
  ORBX_ERR_CAPACITY = -4,  /* a result did not fit the capacity the caller gave     */
  ORBX_ERR_NUMERIC = -5,   /* BA: reduced system not positive definite (LU failure
                              in the reference, local_ba_lm.rs:1036-1039)           */
  ORBX_ERR_EMPTY = -6      /* BA: no parameters or no residuals -> reference returns
                              None (local_ba_lm.rs:923-925)                         */
};

/* = tracking::frame::CameraModel, src/tracking/frame/camera.rs:3-10 */
typedef struct {
  double fx, fy, cx, cy, baseline;
} orbx_camera;

/* = the nine cv::ORB::create arguments, src/tracking/frame/stereo.rs:38-48.
 * The reference's configuration and its neighbourhood are implemented: any n_features,
 * 1 <= n_levels <= 8, scale_factor in (1, 1.5], fast_threshold 1..254; edge_threshold 31,
 * first_level 0, wta_k 2, score_type 0 (HARRIS_SCORE), patch_size 31 are fixed;
 * other values -> ORBX_ERR_INVALID. */
typedef struct {
  int n_features;
  float scale_factor;
  int n_levels, edge_threshold, first_level, wta_k, score_type, patch_size, fast_threshold;
} orbx_orb_params;

/* = cv::KeyPoint as seen through opencv::core::KeyPoint (28 bytes) */
typedef struct {
  float x, y, size, angle, response;
  int octave, class_id;
} orbx_keypoint;

/* = cv::DMatch (16 bytes); field order of the Rust struct literal at stereo.rs:149-154 */
typedef struct {
  int query_idx, train_idx, img_idx;
  float distance;
} orbx_dmatch;

typedef struct orbx_handle orbx_handle;

const char* orbx_version(void);
/* Layout version of this header's structs and entry points: bumped whenever a struct grows or a signature changes (2: orbx_ba_window
 * gained `obs32`, so its array stride changed).  orbx_abi_version() is what the loaded library was built with; a caller compiled against
 * another ORBX_ABI_VERSION must not call it (the C++ and Python mirrors check at handle creation / load time, and the Rust shim of
 * INTEGRATION.md does the same in StereoProcessor::new). */
#define ORBX_ABI_VERSION 2
int orbx_abi_version(void);
const char* orbx_last_error(const orbx_handle* h);

/* Fills *p with the reference's ORB configuration (stereo.rs:38-48) for n_features. */
void orbx_default_orb_params(int n_features, orbx_orb_params* p);

/* Replaces StereoProcessor::new (stereo.rs:37-50).  `max_batch` = the largest number of
 * stereo pairs one *_batch_device call will carry (>= 1); max_w/max_h bound image size. */
int orbx_create(const orbx_camera* cam, const orbx_orb_params* orb, int device, int max_w,
                int max_h, int max_batch, orbx_handle** out);
void orbx_destroy(orbx_handle* h);

/* The handle's hipStream_t (as void*), so a host framework can order its own work
 * (copies, collectives) against the library's. */
void* orbx_stream(orbx_handle* h);
int orbx_synchronize(orbx_handle* h);

/* ---- per-frame feature pipeline ------------------------------------------------ */

/* Replaces StereoProcessor::process (stereo.rs:52-66) for one stereo pair held in host
 * memory: extract L, extract R (cv::ORB::detectAndCompute, stereo.rs:68-78), match
 * (stereo.rs:80-161), triangulate (stereo.rs:186-216).
 *   left/right: CV_8UC1 rows of `w` pixels, row strides in bytes.
 *   kpL/kpR [cap_kp], descL/descR [cap_kp*32], matches [cap_kp], points_cam [cap_kp*3],
 *   has_point [cap_kp] (1 = Some, 0 = None; points_cam is (0,0,0) where 0). */
int orbx_process_stereo(orbx_handle* h, const uint8_t* left, size_t lstride, const uint8_t* right,
                        size_t rstride, int w, int h_px, orbx_keypoint* kpL, uint8_t* descL,
                        int* nL, orbx_keypoint* kpR, uint8_t* descR, int* nR, int cap_kp,
                        orbx_dmatch* matches, int* n_matches, double* points_cam,
                        uint8_t* has_point);

/* Same path for `batch` stereo pairs resident in device memory (the throughput form).
 *   d_images: [batch][2][h_px][stride] u8, left then right of each pair.
 *   d_kp [batch][2][cap_kp], d_desc [batch][2][cap_kp][32], d_nkp [batch][2],
 *   d_matches [batch][cap_kp], d_nmatches [batch], d_points [batch][cap_kp][3],
 *   d_has_point [batch][cap_kp].
 * Asynchronous; call orbx_check_status (synchronises) before trusting the results. */
int orbx_process_stereo_batch_device(orbx_handle* h, const uint8_t* d_images, int batch, int w,
                                     int h_px, size_t stride, orbx_keypoint* d_kp,
                                     uint8_t* d_desc, int* d_nkp, int cap_kp,
                                     orbx_dmatch* d_matches, int* d_nmatches, double* d_points,
                                     uint8_t* d_has_point);

/* The same path for `batch` stereo pairs held in HOST memory (layouts as above, all arrays on the host).
 * The batch is cut into chunks of at most `max_batch` pairs; the upload of chunk i+1, the kernels of chunk i and
 * the download of chunk i-1 run concurrently on three HIP streams (double-buffered device staging).  The copies
 * only overlap when the host buffers are page-locked: allocate them with orbx_host_alloc (or register them).
 * Synchronous: returns when every result is in the host arrays. */
int orbx_process_stereo_batch(orbx_handle* h, const uint8_t* images, int batch, int w, int h_px,
                              size_t stride, orbx_keypoint* kp, uint8_t* desc, int* nkp, int cap_kp,
                              orbx_dmatch* matches, int* nmatches, double* points, uint8_t* has_point);
/* Page-locked host memory for the call above (hipHostMalloc / hipHostFree). */
void* orbx_host_alloc(size_t bytes);
void orbx_host_free(void* p);

/* Extraction alone (= detect_features, stereo.rs:68-78) for `n_images` device images
 * [n_images][h_px][stride]; outputs as above with one slot per image. */
int orbx_extract_batch_device(orbx_handle* h, const uint8_t* d_images, int n_images, int w,
                              int h_px, size_t stride, orbx_keypoint* d_kp, uint8_t* d_desc,
                              int* d_nkp, int cap_kp);

/* Synchronises the stream and returns the sticky device-side status of the calls
 * since the last check (ORBX_OK or ORBX_ERR_CAPACITY ...), then clears it. */
int orbx_check_status(orbx_handle* h);

/* ---- matchers ------------------------------------------------------------------- */

/* = StereoProcessor::match_features + triangulate (stereo.rs:80-161, 186-216) on feature
 * sets the caller already has (host memory).  Uses the handle's CameraModel. */
int orbx_stereo_match(orbx_handle* h, const orbx_keypoint* kpL, const uint8_t* descL, int nL,
                      const orbx_keypoint* kpR, const uint8_t* descR, int nR,
                      orbx_dmatch* matches, int* n_matches, double* points_cam,
                      uint8_t* has_point);

/* Device form, `batch` independent pairs; per-pair slots of cap_kp as in
 * orbx_process_stereo_batch_device. */
int orbx_stereo_match_batch_device(orbx_handle* h, int batch, const orbx_keypoint* d_kp,
                                   const uint8_t* d_desc, const int* d_nkp, int cap_kp,
                                   orbx_dmatch* d_matches, int* d_nmatches, double* d_points,
                                   uint8_t* d_has_point);

/* = BFMatcher::new(NORM_HAMMING, crossCheck=true).train_match(query, train)
 * (src/tracking/tracker.rs:1001-1010): mutual nearest neighbours, ascending query index,
 * lowest index wins distance ties.  q [nq][32], t [nt][32], out [nq] (host memory). */
int orbx_hamming_match_crosscheck(orbx_handle* h, const uint8_t* q, int nq, const uint8_t* t,
                                  int nt, orbx_dmatch* out, int* n_out);
int orbx_hamming_match_crosscheck_device(orbx_handle* h, const uint8_t* d_q, int nq,
                                         const uint8_t* d_t, int nt, orbx_dmatch* d_out,
                                         int* d_n_out);

/* = descriptor_distance (stereo.rs:166-175) over n_pairs rows: out[i] = popcount(a[i]^b[i])
 * over 32 bytes.  Host memory. */
int orbx_hamming_batch(orbx_handle* h, const uint8_t* a, const uint8_t* b, int n_pairs,
                       uint32_t* out);
int orbx_hamming_batch_device(orbx_handle* h, const uint8_t* d_a, const uint8_t* d_b, int n_pairs,
                              uint32_t* d_out);

/* Guided matching = FeatureGrid::new + get_features_in_area (src/tracking/tracking_frame.rs:52-128,
 * 64x48 cells over img_w x img_h) followed by the tracker's descriptor search over the candidates:
 *   mode 0  track_with_motion_model (src/tracking/tracker.rs:1126-1157): smallest distance < 100;
 *   mode 1  track_local_map (tracker.rs:880-923): best <= 100 and, with more than one candidate,
 *           best <= 0.75 * second.
 * Ties go to the first candidate in the reference's visiting order (cells row-major, keypoint index
 * ascending inside a cell).  kp/desc: the frame's n features; q_uv [nq][2] f64 projected positions,
 * q_desc [nq][32] map-point descriptors; out_idx [nq] = keypoint index or -1, out_dist [nq].
 * Keypoint and query coordinates must be finite.  Host and device forms. */
int orbx_guided_match(orbx_handle* h, const orbx_keypoint* kp, const uint8_t* desc, int n, double img_w,
                      double img_h, const double* q_uv, const uint8_t* q_desc, int nq, double radius,
                      int mode, int* out_idx, uint32_t* out_dist);
int orbx_guided_match_device(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n,
                             double img_w, double img_h, const double* d_q_uv, const uint8_t* d_q_desc,
                             int nq, double radius, int mode, int* d_out_idx, uint32_t* d_out_dist);

/* = search_for_triangulation (src/local_mapping/triangulation.rs:401-527): matches between the features of two
 * keyframes that have no map point yet, inside a 100-px grid neighbourhood (32-px cells), gated by the distance to
 * the epipole (features without stereo depth) and to the epipolar line (chi2 3.84), smallest Hamming distance
 * < max_dist, greedy one-to-one in ascending index of keyframe 1.
 *   mp1 [n1] / mp2 [n2]: 1 = the feature already has a map point; stereo1 [n1]: 1 = points_cam is Some;
 *   pose1_wc / pose2_wc: 7 doubles (qw,qx,qy,qz,tx,ty,tz), camera-to-world as the Map stores them;
 *   out_pairs [n1][2] = (idx1, idx2) ascending idx1.  Host memory.  Uses the camera given here (image size is
 *   2cx x 2cy as in the reference, :434-435). */
int orbx_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* kp1,
                                  const uint8_t* desc1, const uint8_t* mp1, const uint8_t* stereo1, int n1,
                                  const orbx_keypoint* kp2, const uint8_t* desc2, const uint8_t* mp2, int n2,
                                  const double* pose1_wc, const double* pose2_wc, unsigned max_dist,
                                  int* out_pairs, int* n_out);

/* Device-resident form: every array in device memory (keypoints and descriptors as the extractor left them), poses on the
 * host; asynchronous on the handle's stream.  d_pairs [n1][2], d_n_out [1]. */
int orbx_search_for_triangulation_device(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* d_kp1,
                                         const uint8_t* d_desc1, const uint8_t* d_mp1, const uint8_t* d_stereo1, int n1,
                                         const orbx_keypoint* d_kp2, const uint8_t* d_desc2, const uint8_t* d_mp2, int n2,
                                         const double* pose1_wc, const double* pose2_wc, unsigned max_dist,
                                         int* d_pairs, int* d_n_out);

/* = search_for_triangulation_bow (src/local_mapping/triangulation.rs:541-658): as above, but the candidates of a
 * feature are the features of keyframe 2 in the same FeatureVector node instead of a grid neighbourhood.
 *   node1 [n1] / node2 [n2]: the FeatureVector key of each feature (the out_node of orbx_bow_transform), 0xffffffff
 *   for a feature in no list.  The reference walks feat_vec1 in HashMap order; features of different nodes never
 *   compete, so the set of pairs does not depend on that order — here they come out in ascending idx1. */
int orbx_search_for_triangulation_bow(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* kp1,
                                      const uint8_t* desc1, const uint8_t* mp1, const uint8_t* stereo1,
                                      const uint32_t* node1, int n1, const orbx_keypoint* kp2, const uint8_t* desc2,
                                      const uint8_t* mp2, const uint32_t* node2, int n2, const double* pose1_wc,
                                      const double* pose2_wc, unsigned max_dist, int* out_pairs, int* n_out);

/* ---- ORB vocabulary (src/vocabulary/mod.rs) ------------------------------------------------------------------
 * orbx_vocab_load_text = OrbVocabulary::load_from_text (:117-211): DBoW2 text format, "k L scoring weighting" then
 * one line per node "parent_id is_leaf d0 .. d31 weight"; lines with fewer than 35 fields are skipped, a field that
 * does not parse fails the load (ORBX_ERR_INVALID + message), node ids are sequential from 1, a node is linked to its
 * parent only when the parent id is smaller than its own.  orbx_vocab_create takes the same nodes as arrays
 * (index 0 = root, ignored).  The tables live on the handle's device until orbx_vocab_destroy. */
typedef struct orbx_vocabulary orbx_vocabulary;
int orbx_vocab_load_text(orbx_handle* h, const char* path, orbx_vocabulary** out);
int orbx_vocab_create(orbx_handle* h, int n_nodes, const uint32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                      const double* weight, int k, int l, orbx_vocabulary** out);
void orbx_vocab_destroy(orbx_vocabulary* v);
int orbx_vocab_info(const orbx_vocabulary* v, int* k, int* l, int* n_nodes, int* n_words);   /* params/num_nodes/num_words */
int orbx_vocab_nodes(const orbx_vocabulary* v, uint32_t* parent, uint8_t* is_leaf, uint8_t* desc, double* weight);

/* The per-descriptor part of OrbVocabulary::transform (:296-325): descent to the leaf taking the closest child at
 * every node (first child on ties, :230-248), then `levels_up` parents up (:262-275).
 *   out_word [n] word id (0 for a terminal node that is not flagged leaf, :247), out_leaf [n] leaf node id,
 *   out_node [n] FeatureVector key, out_weight [n] the leaf's weight.  BowVector = sum of out_weight per out_word,
 *   L1-normalised; FeatureVector = feature indices grouped by out_node — both left to the caller.
 * _device: descriptors and outputs in device memory, asynchronous on the handle's stream. */
int orbx_bow_transform(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* desc, int n, int levels_up,
                       uint32_t* out_word, uint32_t* out_leaf, uint32_t* out_node, double* out_weight);
int orbx_bow_transform_device(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* d_desc, int n, int levels_up,
                              uint32_t* d_word, uint32_t* d_leaf, uint32_t* d_node, double* d_weight);

/* The two maps OrbVocabulary::transform returns (mod.rs:296-325; transform_bow_only :327-355 is the first alone), built on the
 * device from the per-descriptor results above:
 *   BowVector     bow_word [n_bow] ascending word ids, bow_weight [n_bow]: per word the sum of its features' leaf weights in feature
 *                 order, then L1-normalised (:316-321; the norm is summed in ascending word order — the reference's HashMap order
 *                 is unspecified);
 *   FeatureVector fv_node [n_fv] ascending node ids, the features of node i = fv_index[fv_start[i] .. fv_start[i+1]) ascending.
 * Arrays are sized for n entries (fv_start n+1); at most 8192 descriptors per call.  _device: everything in device memory,
 * d_counts [2] = {n_bow, n_fv}, asynchronous.  orbx_bow_score = OrbVocabulary::score (:357-374) on two such BowVectors (host
 * arithmetic, no handle needed; word ids must ascend). */
int orbx_bow_vectors(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* desc, int n, int levels_up, uint32_t* bow_word,
                     double* bow_weight, int* n_bow, uint32_t* fv_node, int* fv_start, int* fv_index, int* n_fv);
int orbx_bow_vectors_device(orbx_handle* h, const orbx_vocabulary* v, const uint8_t* d_desc, int n, int levels_up,
                            uint32_t* d_bow_word, double* d_bow_weight, uint32_t* d_fv_node, int* d_fv_start, int* d_fv_index,
                            int* d_counts);
int orbx_bow_score(const uint32_t* word1, const double* weight1, int n1, const uint32_t* word2, const double* weight2, int n2,
                   double* score);

/* = the search of fuse_points_into_keyframes (src/local_mapping/search_in_neighbors.rs:273-343, with
 * KeyFrame::get_features_in_area, src/atlas/map/keyframe.rs:408-443) for every (map point, target keyframe) pair:
 * project the point with the keyframe's inverse pose, skip it behind the camera or outside [0,2cx)x[0,2cy), radius
 * = clamp(radius_scale * depth / fx, 10, 50), best = smallest Hamming distance < desc_threshold among the keyframe's
 * keypoints inside the circle (lowest index on ties).  The map mutation that consumes the result (:345-383) stays
 * on the host; it changes neither positions nor descriptors, so all pairs can be searched up front.
 *   positions [P][3], mp_desc [P][32]; kf_poses_wc [T][7] (qw,qx,qy,qz,tx,ty,tz; the Map's camera-to-world pose);
 *   kf_feat_offset [T+1]: keyframe t owns kps/descs[kf_feat_offset[t] .. kf_feat_offset[t+1]);
 *   radius_scale = config.radius_factor * scale_factor.powi(num_levels - 1) (:303);
 *   out_idx [P][T] = feature index inside the keyframe or -1; out_dist [P][T] (0 where -1).  Host memory. */
int orbx_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                     const double* kf_poses_wc, const int* kf_feat_offset, const orbx_keypoint* kps, const uint8_t* descs,
                     int T, double radius_scale, unsigned desc_threshold, int* out_idx, uint32_t* out_dist);

/* Device-resident form of orbx_fuse_search: positions, descriptors, keypoints, offsets and outputs in device memory, the
 * keyframe poses on the host (they are inverted there); asynchronous on the handle's stream. */
int orbx_fuse_search_device(orbx_handle* h, const orbx_camera* cam, const double* d_positions, const uint8_t* d_mp_desc, int P,
                            const double* kf_poses_wc, const int* d_kf_feat_offset, const orbx_keypoint* d_kps,
                            const uint8_t* d_descs, int T, double radius_scale, unsigned desc_threshold, int* d_out_idx,
                            uint32_t* d_out_dist);

/* ---- keyframe hand-off with device-resident features (src/system/messages.rs:19-51) --------------------------------
 * orbx_keyframe = the payload of NewKeyFrameMsg — keyframe id, timestamp, T_wc pose, keypoints, descriptors, stereo points
 * (`points_cam`, None where has_point is 0) and the map-point associations (`matched_map_points`) — with the four feature
 * arrays held in device memory.  It is made straight from the device outputs of the extractor (device-to-device copies on the
 * handle's stream, asynchronous: slot `b` of orbx_process_stereo_batch_device is d_kp + b*2*cap_kp etc., n = nkp[b][0]), so the
 * features a frame was just given never leave the GPU between `process` and the searches local mapping runs on them.
 * The ids of the associations are map bookkeeping and stay on the host; the device holds the is_some() flags the searches
 * read.  d_points_cam / d_has_point may both be NULL (monocular: every point None).  A keyframe belongs to the handle
 * that made it and must be destroyed before it.  */
typedef struct orbx_keyframe orbx_keyframe;
int orbx_keyframe_create(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n, const double* d_points_cam,
                         const uint8_t* d_has_point, uint64_t keyframe_id, uint64_t timestamp_ns, const double* pose_wc,
                         orbx_keyframe** out);
void orbx_keyframe_destroy(orbx_keyframe* kf);
int orbx_keyframe_info(const orbx_keyframe* kf, int* n_features, uint64_t* keyframe_id, uint64_t* timestamp_ns, double* pose_wc);
int orbx_keyframe_set_pose(orbx_keyframe* kf, const double* pose_wc);                 /* after BA / pose refinement       */
int orbx_keyframe_set_map_points(orbx_keyframe* kf, const int64_t* mp_ids);           /* [n], -1 = None                   */
int orbx_keyframe_get_map_points(const orbx_keyframe* kf, int64_t* mp_ids);
/* host copies for consumers that still want Vector<KeyPoint> / Mat (any pointer may be NULL) */
int orbx_keyframe_download(const orbx_keyframe* kf, orbx_keypoint* kp, uint8_t* desc, double* points_cam, uint8_t* has_point);
const orbx_keypoint* orbx_keyframe_device_keypoints(const orbx_keyframe* kf);
const uint8_t* orbx_keyframe_device_descriptors(const orbx_keyframe* kf);
/* The searches on device-resident keyframes; semantics, tie rules and results exactly those of orbx_guided_match,
 * orbx_search_for_triangulation (kf1 = the new keyframe, its stereo flags = has_point; poses = the keyframes' own) and
 * orbx_fuse_search (radius_scale as there) — only the small inputs (queries, map points) and the results cross PCIe. */
int orbx_keyframe_guided_match(orbx_handle* h, const orbx_keyframe* kf, double img_w, double img_h, const double* q_uv,
                               const uint8_t* q_desc, int nq, double radius, int mode, int* out_idx, uint32_t* out_dist);
int orbx_keyframe_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const orbx_keyframe* kf1,
                                           const orbx_keyframe* kf2, unsigned max_dist, int* out_pairs, int* n_out);
int orbx_keyframe_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                              const orbx_keyframe* const* kfs, int T, double radius_scale, unsigned desc_threshold,
                              int* out_idx, uint32_t* out_dist);

/* ---- input side (src/io/euroc.rs) ------------------------------------------------------------------------------
 * Host code: the EuRoC `mav0` reader of EurocDataset::new / len / frame_timestamp / stereo_pair (:64-132,
 * load_image_list :189-211, the camera part of load_stereo_calibration :325-360) and the PNG decode that
 * cv::imread(IMREAD_GRAYSCALE) does there.  No GPU involved; outputs feed orbx_process_stereo[_batch].
 *   orbx_png_decode_gray8: non-interlaced greyscale PNG (8 or 16 bit, optional alpha) -> 8-bit rows; out == NULL
 *     returns only the size.  Anything else (palette, RGB, interlaced, damaged) is ORBX_ERR_INVALID.
 *   orbx_euroc_open: ORBX_ERR_INVALID + message in err where the reference returns Err (missing file, a timestamp
 *     that does not parse, records of different lengths, cam0/cam1 of different lengths, bad yaml).
 *   orbx_euroc_calibration: left = {fx, fy, cx, cy of cam0, baseline = |t(T_cam1_body * T_cam0_body^-1)|}.
 *   orbx_euroc_read_pairs: frames [first, first+count) decoded by `threads` host threads into out[pair][2][h][w],
 *     the layout orbx_process_stereo_batch takes (use orbx_host_alloc memory to overlap the upload). */
typedef struct orbx_euroc orbx_euroc;
int orbx_png_decode_gray8(const uint8_t* file, size_t n, uint8_t* out, size_t stride, int* w, int* h);
int orbx_euroc_open(const char* mav0_dir, orbx_euroc** out, char* err, size_t err_cap);
void orbx_euroc_close(orbx_euroc* d);
int orbx_euroc_len(const orbx_euroc* d);
const char* orbx_euroc_last_error(const orbx_euroc* d);
int orbx_euroc_frame_timestamp(const orbx_euroc* d, int idx, uint64_t* timestamp_ns);
int orbx_euroc_calibration(const orbx_euroc* d, orbx_camera* left, double* k_right4, int* w, int* h);
int orbx_euroc_read_pairs(orbx_euroc* d, int first, int count, uint8_t* out, int threads);

/* ---- local bundle adjustment ------------------------------------------------------ */

/* = LocalBAConfigLM, src/optimizer/local_ba_lm.rs:96-119 */
typedef struct {
  int max_iterations;
  double param_tolerance, gradient_tolerance, huber_threshold;
  int max_covisible_keyframes;
} orbx_ba_config;

/* One VisualObservation (local_ba_lm.rs:68-78) after the id -> index re-keying the
 * reference does at :928-961.  kf_idx >= 0: optimised keyframe (is_kf_optimized);
 * kf_idx < 0: fixed keyframe `fixed_idx` (fixed_idx < 0 = unknown id -> identity pose,
 * local_ba_lm.rs:569).  _pad: 0 for the visual solvers; bit 0 = is_stereo for orbx_ba_solve_inertial. */
typedef struct {
  int32_t kf_idx, fixed_idx, mp_idx, _pad;
  double u, v;
} orbx_ba_obs;

void orbx_default_ba_config(orbx_ba_config* c);

/* = `&dyn Fn() -> bool` of solve_visual_ba; polled once at the top of each LM iteration
 * (local_ba_lm.rs:1013).  NULL = never stop. */
typedef int (*orbx_should_stop_fn)(void* user);

/* Collective hook for the point-partitioned multi-GPU form: sum `n`
 * doubles at device pointer `d_buf` in place over all ranks, ordered on `hip_stream`.
 * The host framework implements it (torch.distributed / RCCL).  NULL = single GPU. */
typedef int (*orbx_allreduce_fn)(void* user, void* d_buf, size_t n, void* hip_stream);
int orbx_ba_set_allreduce(orbx_handle* h, orbx_allreduce_fn fn, void* user);

/* The native collective of the point-partitioned solve: RCCL over xGMI, called by the library itself —
 * ncclAllReduce(ncclDouble, ncclSum) in place on the handle's stream — so a Rust (or C) host
 * needs no framework around it.  Either
 *   orbx_rccl_unique_id (rank 0; = ncclGetUniqueId, returns the id's size, 128) + the host's own way of handing the id to
 *   the other ranks + orbx_ba_init_rccl on every rank (= ncclCommInitRank; the library owns the communicator, one per
 *   handle, destroyed with it), or
 *   orbx_ba_set_rccl_comm with an ncclComm_t the host already has (caller-owned; NULL clears).
 * A communicator takes precedence over the orbx_ba_set_allreduce hook, which stays for hosts that bring another transport
 * (the CPU tests run it over gloo).  Every rank must call the solve with the same problem and its own partition; all
 * ranks issue the same sequence of collectives whatever their own should_stop() answers: the stop decision is part of
 * what is reduced (one rank asking is enough, and all ranks then stop before the same iteration), and so is "some rank
 * holds an observation with an index out of range" (every rank returns ORBX_ERR_INVALID). */
int orbx_rccl_unique_id(uint8_t* out, size_t cap);
int orbx_ba_init_rccl(orbx_handle* h, const uint8_t* unique_id, size_t id_bytes, int rank, int world);
int orbx_ba_set_rccl_comm(orbx_handle* h, void* nccl_comm);
/* What would carry the all-reduces of a partitioned solve on this handle right now: bit 0 = an RCCL communicator
 * (orbx_ba_init_rccl / orbx_ba_set_rccl_comm), bit 1 = the orbx_ba_set_allreduce hook; 0 = none — orbx_ba_solve_visual then
 * treats the observations it is given as the WHOLE problem, so a host that partitions must check before it calls.
 * librccl is bound on first use of the three calls above (dlopen by soname: a process that already mapped a librccl.so.1,
 * e.g. PyTorch's, gets that copy); a single-GPU process never loads it. */
int orbx_ba_has_collective(orbx_handle* h);
/* What the handle's RCCL communicator spans, from the communicator itself (ncclCommCount / ncclCommUserRank): the number of ranks its
 * all-reduce sums over and this rank's index.  ORBX_ERR_INVALID without a communicator.  (bench.py prints it, so that a multi-GPU record
 * shows how many ranks RCCL really saw.) */
int orbx_ba_rccl_world(orbx_handle* h, int* n_ranks, int* rank);

/* Replaces solve_visual_ba (local_ba_lm.rs:912-1098).
 *   poses_cw [K][7]  (qw,qx,qy,qz,tx,ty,tz) T_cw of the optimised keyframes (:966-977)
 *   fixed_poses_cw [F][7]  T_cw of anchor + other fixed observers
 *   points [M][3] in/out   map-point positions (:980-987 / :1079-1089)
 *   obs [N]
 *   poses_wc_out [K][7]    optimised poses inverted back to T_wc (:1062-1077)
 * Everything is f64.  With an allreduce hook set, `obs` holds this rank's partition of the
 * observations (all observations of the points it owns); poses, points and the scalars come
 * out identical on every rank. */
int orbx_ba_solve_visual(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                         const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                         double* points, int N, const orbx_ba_obs* obs,
                         orbx_should_stop_fn should_stop, void* user, double* poses_wc_out,
                         int* iterations, double* initial_error, double* final_error);

/* The same observation in 16 bytes, for callers whose pixel coordinates ARE f32 — the reference's always are: collect_visual_ba_data
 * widens `kp.pt()` (cv::Point2f) to f64 per observation (local_ba_lm.rs:870-872).  The library widens on the device instead, the same
 * exact conversion, so results are bit-identical to handing over the widened orbx_ba_obs — and half the bytes cross PCIe (the upload of
 * the observations is a fifth of a 32-window batch call).  kf_idx >= 0: optimised keyframe; kf_idx < 0: fixed observer -1 - kf_idx
 * (F = the identity pose, as fixed_idx -1 of orbx_ba_obs). */
typedef struct {
  int32_t kf_idx, mp_idx;
  float u, v;
} orbx_ba_obs32;
/* orbx_ba_solve_visual with the observations in the 16-byte form orbx_ba_obs32: f32 pixel coordinates — what the reference's are,
 * `kp.pt()` widened per observation at collect time (local_ba_lm.rs:870-872) — widened on the device, the same exact conversion, so the
 * result equals orbx_ba_solve_visual's on the widened observations bit for bit while half the bytes cross the host link (a 50-keyframe
 * window carries 2*10^5 observations: 3.2 MB instead of 6.5).  Not with an all-reduce hook / communicator installed: the partitioned
 * solve takes orbx_ba_obs. */
int orbx_ba_solve_visual_obs32(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                               const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                               double* points, int N, const orbx_ba_obs32* obs32,
                               orbx_should_stop_fn should_stop, void* user, double* poses_wc_out,
                               int* iterations, double* initial_error, double* final_error);

/* Many independent windows in ONE call ("many BA windows": one map per stream / per robot).  Each
 * window is exactly one orbx_ba_solve_visual problem (same arguments, same LM loop, local_ba_lm.rs:1012-1056) with its own
 * LM state on the device; all windows share every kernel launch — the window is the second grid dimension — so W reduced
 * systems are factored by W workgroups at once and the point / keyframe / Schur kernels of all windows fill the chip
 * together, instead of one window's chain of short launches using about 1 % of it.  Nothing is exchanged between windows,
 * and a window's arithmetic does not depend on what else is in the batch: every window's result equals the result of
 * orbx_ba_solve_visual on it bit for bit.  Windows may differ in size.  should_stop is polled once per iteration for the
 * whole batch (and while the work drains); a window that converges early simply stops taking part.
 *   status: ORBX_OK, or ORBX_ERR_EMPTY for a window the reference answers None for (:923-925) — the call still returns
 *   ORBX_OK and solves the others (also when that window is the only one of the batch).  An observation index out of range fails
 *   the whole call (ORBX_ERR_INVALID): every window's status then carries the error, and the in/out `points` / `poses_wc_out` of
 *   windows that had already finished (the other half of a two-stream batch) hold their results — hand in the original points
 *   again when retrying.  No C++ exception leaves this call or any other orbx_ba_* entry point: allocation or thread-creation failure is an error code.
 * The all-reduce hook / RCCL communicator is not used here (independent windows need no collective).
 * A batch of 16 or more windows without a should_stop callback runs as two halves at once: the second half on an internal second
 * stream with its own workspaces, driven by a helper thread for the duration of the call, so that one half's host preprocessing and
 * transfers run under the other half's kernels (+12 % LM iterations/s at 32 windows); the halves are cut where the observation
 * count is halved; if the second stream cannot be created the whole batch runs on the first.  Results do not depend on it.  With a callback
 * (which would otherwise be called from two threads), or with per-kernel profiling on, the call keeps to one stream.
 * Observations are handed over as they are: the index checks, the per-point grouping and the per-keyframe lists are made on the
 * device (four short launches per call shared by all windows), so the host makes no pass over them.  If `obs` lies in page-locked host
 * memory (orbx_host_alloc, hipHostMalloc or a registered range — the library asks the runtime) the copy engine reads it where it lies;
 * windows whose `obs` arrays follow each other in memory travel as ONE copy.  Pageable `obs` is first copied into the handle's pinned
 * staging blob (a plain copy, on worker threads when large).  The same holds for the one-window entry points. */

typedef struct {
  int K;                        /* in: optimised keyframes                         */
  const double* poses_cw;       /* in: [K][7]                                      */
  int F;
  const double* fixed_poses_cw; /* in: [F][7]                                      */
  int M;
  double* points;               /* in/out: [M][3]                                  */
  int N;
  const orbx_ba_obs* obs;       /* in: [N] (or NULL when obs32 is given)           */
  double* poses_wc_out;         /* out: [K][7]                                     */
  int status;                   /* out                                             */
  int iterations;               /* out                                             */
  double initial_error, final_error;   /* out                                      */
  const orbx_ba_obs32* obs32;   /* in: [N], optional: used instead of obs when not NULL */
} orbx_ba_window;
int orbx_ba_solve_visual_batch(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int n_windows,
                               orbx_ba_window* windows, orbx_should_stop_fn should_stop, void* user);

/* Replaces solve_global_ba (src/optimizer/global_ba.rs:184-418): the same LM loop over ALL keyframes of the map
 * with the first one (smallest id, :116-119) fixed, and one difference in the linearisation — an observation whose
 * point is not in front of its camera (z_c <= 0.001) keeps its 100-px residual but contributes zero Jacobian rows
 * (:561-563).
 *   poses_cw [K][7]: T_cw of the K = n_kfs - 1 optimised keyframes in kf_ids order with the fixed one removed
 *   (:214-220); fixed_pose_cw [7]: T_cw of the fixed keyframe; obs[i].kf_idx < 0 (fixed_idx 0) marks it.
 *   Tolerances come from cfg (GlobalBAConfig defaults: 10 iterations, 1e-6, 1e-6, sqrt(5.991), :36-45).
 *   poses_wc_out [K][7]; the fixed keyframe's result is its input pose (:386).
 * ORBX_ERR_EMPTY where the reference returns None (:194-196, :207-209) and for N = 0, which
 * collect_global_ba_data never produces (:176-178).  The all-reduce hook applies as in orbx_ba_solve_visual. */
int orbx_ba_solve_global(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                         const double* poses_cw, const double* fixed_pose_cw, int M, double* points, int N,
                         const orbx_ba_obs* obs, orbx_should_stop_fn should_stop, void* user,
                         double* poses_wc_out, int* iterations, double* initial_error, double* final_error);

/* orbx_ba_solve_global with 16-byte observations (see orbx_ba_solve_visual_obs32): kf_idx -1 marks the fixed keyframe. */
int orbx_ba_solve_global_obs32(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                               const double* poses_cw, const double* fixed_pose_cw, int M, double* points, int N,
                               const orbx_ba_obs32* obs32, orbx_should_stop_fn should_stop, void* user,
                               double* poses_wc_out, int* iterations, double* initial_error, double* final_error);

/* LocalInertialBAConfig (src/optimizer/local_inertial_ba.rs:109-141); orbx_default_inertial_ba_config = its Default:
 * 10 iterations, window 10, sqrt(5.991), sqrt(7.815), lambda 1e-2, gyro random-walk information 1e6, accel 1e4. */
typedef struct {
  int max_iterations;
  int window_size;               /* used by the caller when it collects the temporal window (:366-384) */
  double huber_threshold_mono, huber_threshold_stereo;
  double initial_lambda;
  double gyro_rw_info, accel_rw_info;
} orbx_inertial_ba_config;
void orbx_default_inertial_ba_config(orbx_inertial_ba_config* cfg);

/* Replaces solve_inertial_ba (src/optimizer/local_inertial_ba.rs:1074-1275): visual-inertial local BA over the K
 * keyframes of the temporal window, 15 parameters each (T_wc pose as axis-angle + translation, velocity, gyro bias,
 * accel bias), plus M map points.  Residuals: reprojection (Huber, threshold per observation mono/stereo,
 * obs[i]._pad bit 0 = is_stereo; 100-px penalty and no Jacobian where z_c <= 0.001), the 9-d preintegration
 * residual of imu_factors.rs:66-103 per IMU edge with forward-difference Jacobians (eps 1e-6, :806-861), and the
 * 6-d bias random walk per edge (:676-698).  LM as :1198-1243: initial lambda from cfg, stop on |gradient| < 1e-8
 * or a singular system, no step-size test; errors are |r| (not RMS).
 *   poses_wc [K][7], velocities [K][3], biases [K][6] (gyro xyz, accel xyz); fixed_poses_cw [F][7] (T_cw);
 *   obs: kf_idx = index into the window or -1 (+ fixed_idx); edge_kf [E][2] window indices (i earlier, j later);
 *   preint [E][11] = delta_rot (qw,qx,qy,qz), delta_vel, delta_pos, dt of PreintegratedState (preintegration.rs:85-98).
 *   Outputs for ALL K keyframes (the reference's result maps skip index 0, :1250 — the caller's business).
 * ORBX_ERR_EMPTY where the reference returns None (K < 2, :1080-1082). */
int orbx_ba_solve_inertial(orbx_handle* h, const orbx_camera* cam, const orbx_inertial_ba_config* cfg, int K,
                           const double* poses_wc, const double* velocities, const double* biases, int F,
                           const double* fixed_poses_cw, int M, double* points, int N, const orbx_ba_obs* obs, int E,
                           const int* edge_kf, const double* preint, orbx_should_stop_fn should_stop, void* user,
                           double* poses_wc_out, double* vel_out, double* bias_out, int* iterations,
                           double* initial_error, double* final_error);

/* Per-kernel device time for bench.py's roofline block.  While profiling is on
 * (orbx_set_profiling), every launch is bracketed by HIP events on the handle's stream;
 * orbx_get_kernel_times synchronises, fills up to `cap` entries with the durations summed
 * over all calls since the previous read (returns the number of kernels seen), and starts a
 * new accumulation window. */
typedef struct {
  char name[48];
  float ms;          /* summed duration of this kernel's launches in the window */
  int launches;
} orbx_kernel_time;
int orbx_set_profiling(orbx_handle* h, int on);
/* The same with the events around the launches of ONE kernel only (its name as orbx_get_kernel_times reports it): what a
 * timed region that needs one kernel's durations pays — two event records per launch of that kernel instead of two per
 * launch of every kernel (about 1.5 % of a 256-pair step).  NULL or "" = every kernel, as orbx_set_profiling(h, 1). */
int orbx_set_profiling_only(orbx_handle* h, const char* kernel_name);
int orbx_get_kernel_times(orbx_handle* h, orbx_kernel_time* out, int cap);

/* ---- stage inspection (tests, debugging) ---------------------------------------------------
 * Read back intermediate results of the LAST extraction call of this handle: one pyramid level
 * (which = 0) or its blurred version (which = 1) of image `image_index` as w_l*h_l tightly
 * packed bytes (out may be NULL to query the size), and the FAST+NMS candidates of a level packed
 * as (score<<24 | y<<12 | x) in no particular order.  Synchronous. */
int orbx_debug_read_level(orbx_handle* h, int image_index, int level, int which, uint8_t* out,
                          int* w_l, int* h_l);
int orbx_debug_read_candidates(orbx_handle* h, int image_index, int level, uint32_t* out, int cap,
                               int* n);

/* The per-observation terms of the BA linearisation at the given parameters, from the same device functions the solver's
 * kernels call: compute_residuals (local_ba_lm.rs:557-588) and the blocks of compute_jacobian (:591-639; jacobian_pose
 * :216-254, jacobian_point :257-288, huber_weight :291-297), in input order.
 *   out [N][20] = residual (2) | A = d r / d pose, row-major 2x6 (rot xyz, trans xyz) | B = d r / d point, row-major 2x3,
 *   all times sqrt(w).  global_mode != 0: the zero-Jacobian rule of solve_global_ba (global_ba.rs:561-563).
 * This is how the reference's Jacobian known answer (test_jacobian_pose_numerical, :1163-1243) is checked on the GPU. */
int orbx_debug_ba_blocks(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw,
                         int F, const double* fixed_poses_cw, int M, const double* points, int N, const orbx_ba_obs* obs,
                         int global_mode, double* out);

/* compute_imu_residual (src/optimizer/imu_factors.rs:66-103) of every IMU edge at the given keyframe states, from the device
 * function the inertial solver's ba_imu_kernel calls.
 *   poses_wc [K][7] (qw,qx,qy,qz,tx,ty,tz) T_wc, velocities [K][3], edge_kf [E][2] = (i, j), preint [E][11] as orbx_ba_solve_inertial
 *   (delta_rot qw,qx,qy,qz | delta_vel | delta_pos | dt);  out [E][9] = rotation | velocity | position residual.
 * This is how the reference's own known answer for this factor (test_imu_residual_zero_motion, imu_factors.rs:264-276: identical
 * states and an identity preintegration give a zero residual) is checked on the GPU. */
int orbx_debug_imu_residual(orbx_handle* h, int K, const double* poses_wc, const double* velocities, int E, const int* edge_kf,
                            const double* preint, double* out);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
