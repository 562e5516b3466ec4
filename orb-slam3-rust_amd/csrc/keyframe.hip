// keyframe.hip — orbx_keyframe: the payload of NewKeyFrameMsg (src/system/messages.rs:19-51) with the feature arrays kept
// in device memory, so that what Tracking hands to Local Mapping — keypoints, descriptors, stereo points, map-point
// associations of a processed frame — feeds the descriptor searches of local mapping (guided match, triangulation search,
// fuse search: SURVEY.md §8f rows 1 and 3) without a D2H / H2D round trip of 60 bytes per feature per search.
// Host code only: the searches themselves are the *_device entry points of orbx_api.hip.
#include <vector>

#include "orbx_internal.hpp"

struct orbx_keyframe {
  orbx_handle* h = nullptr;
  uint64_t id = 0, timestamp_ns = 0;
  double pose_wc[7] = {1, 0, 0, 0, 0, 0, 0};
  int n = 0;
  uint8_t* block = nullptr;          // one device allocation: kp | desc | points_cam | has_point | mp_flag
  orbx_keypoint* d_kp = nullptr;
  uint8_t* d_desc = nullptr;
  double* d_points = nullptr;
  uint8_t* d_has_point = nullptr;
  uint8_t* d_mp_flag = nullptr;
  std::vector<int64_t> mp_ids;       // matched_map_points (host side: ids are map bookkeeping), -1 = None
};

extern "C" {

int orbx_keyframe_create(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n, const double* d_points_cam,
                         const uint8_t* d_has_point, uint64_t keyframe_id, uint64_t timestamp_ns, const double* pose_wc,
                         orbx_keyframe** out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!out || n < 0 || (n > 0 && (!d_kp || !d_desc)) || ((d_points_cam == nullptr) != (d_has_point == nullptr)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_keyframe_create: bad argument");
  *out = nullptr;
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_keyframe* kf = new orbx_keyframe();
  kf->h = h; kf->id = keyframe_id; kf->timestamp_ns = timestamp_ns; kf->n = n;
  if (pose_wc) memcpy(kf->pose_wc, pose_wc, sizeof(kf->pose_wc));
  kf->mp_ids.assign((size_t)n, -1);
  const size_t n1 = (size_t)(n > 0 ? n : 1);
  const size_t o_kp = 0, o_desc = (o_kp + sizeof(orbx_keypoint) * n1 + 255) & ~(size_t)255, o_pts = (o_desc + 32 * n1 + 255) & ~(size_t)255,
               o_has = (o_pts + 24 * n1 + 255) & ~(size_t)255, o_mp = (o_has + n1 + 255) & ~(size_t)255, total = o_mp + ((n1 + 255) & ~(size_t)255);
  if (hipMalloc((void**)&kf->block, total) != hipSuccess) { delete kf; return orbx_fail(h, ORBX_ERR_HIP, "orbx_keyframe_create: out of device memory"); }
  kf->d_kp = (orbx_keypoint*)(kf->block + o_kp); kf->d_desc = kf->block + o_desc; kf->d_points = (double*)(kf->block + o_pts);
  kf->d_has_point = kf->block + o_has; kf->d_mp_flag = kf->block + o_mp;
  hipStream_t st = h->stream;
  hipError_t e = hipMemsetAsync(kf->block + o_pts, 0, total - o_pts, st);       // points (0,0,0), no stereo point, no map point
  if (e == hipSuccess && n > 0) e = hipMemcpyAsync(kf->d_kp, d_kp, sizeof(orbx_keypoint) * (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && n > 0) e = hipMemcpyAsync(kf->d_desc, d_desc, 32 * (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && n > 0 && d_points_cam) e = hipMemcpyAsync(kf->d_points, d_points_cam, 24 * (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess && n > 0 && d_has_point) e = hipMemcpyAsync(kf->d_has_point, d_has_point, (size_t)n, hipMemcpyDeviceToDevice, st);
  if (e != hipSuccess) { hipFree(kf->block); delete kf; return orbx_fail(h, ORBX_ERR_HIP, "orbx_keyframe_create: %s", hipGetErrorString(e)); }
  *out = kf;
  return ORBX_OK;
}

void orbx_keyframe_destroy(orbx_keyframe* kf) {
  if (!kf) return;
  hipSetDevice(kf->h->device);
  hipStreamSynchronize(kf->h->stream);
  if (kf->block) hipFree(kf->block);
  delete kf;
}

int orbx_keyframe_info(const orbx_keyframe* kf, int* n_features, uint64_t* keyframe_id, uint64_t* timestamp_ns, double* pose_wc) {
  if (!kf) return ORBX_ERR_INVALID;
  if (n_features) *n_features = kf->n;
  if (keyframe_id) *keyframe_id = kf->id;
  if (timestamp_ns) *timestamp_ns = kf->timestamp_ns;
  if (pose_wc) memcpy(pose_wc, kf->pose_wc, sizeof(kf->pose_wc));
  return ORBX_OK;
}

int orbx_keyframe_set_pose(orbx_keyframe* kf, const double* pose_wc) {
  if (!kf || !pose_wc) return ORBX_ERR_INVALID;
  memcpy(kf->pose_wc, pose_wc, sizeof(kf->pose_wc));
  return ORBX_OK;
}

int orbx_keyframe_set_map_points(orbx_keyframe* kf, const int64_t* mp_ids) {
  if (!kf || (kf->n > 0 && !mp_ids)) return ORBX_ERR_INVALID;
  orbx_handle* h = kf->h;
  ORBX_HIP(h, hipSetDevice(h->device));
  std::vector<uint8_t> flag((size_t)kf->n);
  for (int i = 0; i < kf->n; ++i) { kf->mp_ids[(size_t)i] = mp_ids[i]; flag[(size_t)i] = mp_ids[i] >= 0 ? 1 : 0; }
  if (kf->n > 0) {
    ORBX_HIP(h, hipMemcpyAsync(kf->d_mp_flag, flag.data(), (size_t)kf->n, hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(h, hipStreamSynchronize(h->stream));                     // flag is a local
  }
  return ORBX_OK;
}

int orbx_keyframe_get_map_points(const orbx_keyframe* kf, int64_t* mp_ids) {
  if (!kf || (kf->n > 0 && !mp_ids)) return ORBX_ERR_INVALID;
  if (kf->n > 0) memcpy(mp_ids, kf->mp_ids.data(), sizeof(int64_t) * (size_t)kf->n);
  return ORBX_OK;
}

int orbx_keyframe_download(const orbx_keyframe* kf, orbx_keypoint* kp, uint8_t* desc, double* points_cam, uint8_t* has_point) {
  if (!kf) return ORBX_ERR_INVALID;
  orbx_handle* h = kf->h;
  if (kf->n == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)kf->n;
  if (kp) ORBX_HIP(h, hipMemcpyAsync(kp, kf->d_kp, sizeof(orbx_keypoint) * n, hipMemcpyDeviceToHost, h->stream));
  if (desc) ORBX_HIP(h, hipMemcpyAsync(desc, kf->d_desc, 32 * n, hipMemcpyDeviceToHost, h->stream));
  if (points_cam) ORBX_HIP(h, hipMemcpyAsync(points_cam, kf->d_points, 24 * n, hipMemcpyDeviceToHost, h->stream));
  if (has_point) ORBX_HIP(h, hipMemcpyAsync(has_point, kf->d_has_point, n, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

const orbx_keypoint* orbx_keyframe_device_keypoints(const orbx_keyframe* kf) { return kf ? kf->d_kp : nullptr; }
const uint8_t* orbx_keyframe_device_descriptors(const orbx_keyframe* kf) { return kf ? kf->d_desc : nullptr; }

// ---- the searches of tracking / local mapping on device-resident keyframes ------------------------------------------------

int orbx_keyframe_guided_match(orbx_handle* h, const orbx_keyframe* kf, double img_w, double img_h, const double* q_uv,
                               const uint8_t* q_desc, int nq, double radius, int mode, int* out_idx, uint32_t* out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (!kf || kf->h != h || nq < 0 || (nq > 0 && (!q_uv || !q_desc || !out_idx || !out_dist)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_keyframe_guided_match: bad argument (a keyframe belongs to the handle that made it)");
  if (nq == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  // only the queries travel (48 B each up, 8 B each down); the frame's n x 60 B of features stay where they are
  if (int rc = orbx_reserve(h, h->ws_io[8], 16 * (size_t)nq + 32 * (size_t)nq + 8 * (size_t)nq)) return rc;
  double* d_uv = (double*)h->ws_io[8].p;
  uint8_t* d_qd = (uint8_t*)(d_uv + 2 * (size_t)nq);
  int* d_idx = (int*)(d_qd + 32 * (size_t)nq);
  uint32_t* d_dist = (uint32_t*)(d_idx + nq);
  ORBX_HIP(h, hipMemcpyAsync(d_uv, q_uv, 16 * (size_t)nq, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_qd, q_desc, 32 * (size_t)nq, hipMemcpyHostToDevice, h->stream));
  if (int rc = orbx_guided_match_device(h, kf->d_kp, kf->d_desc, kf->n, img_w, img_h, d_uv, d_qd, nq, radius, mode, d_idx, d_dist)) return rc;
  ORBX_HIP(h, hipMemcpyAsync(out_idx, d_idx, 4 * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_dist, d_dist, 4 * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

int orbx_keyframe_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const orbx_keyframe* kf1, const orbx_keyframe* kf2,
                                           unsigned max_dist, int* out_pairs, int* n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !kf1 || !kf2 || kf1->h != h || kf2->h != h || !n_out || (kf1->n > 0 && !out_pairs))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_keyframe_search_for_triangulation: bad argument");
  *n_out = 0;
  if (kf1->n == 0 || kf2->n == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  if (int rc = orbx_reserve(h, h->ws_io[8], sizeof(int) * (2 * (size_t)kf1->n + 4))) return rc;
  int* d_pairs = (int*)h->ws_io[8].p;
  int* d_n = d_pairs + 2 * (size_t)kf1->n;
  // map_point_ids[i].is_some() = the keyframe's map-point flags, points_cam[i].is_some() = its stereo flags (triangulation.rs:401-527)
  if (int rc = orbx_search_for_triangulation_device(h, cam, kf1->d_kp, kf1->d_desc, kf1->d_mp_flag, kf1->d_has_point, kf1->n, kf2->d_kp,
                                                    kf2->d_desc, kf2->d_mp_flag, kf2->n, kf1->pose_wc, kf2->pose_wc, max_dist, d_pairs, d_n))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (*n_out > 0) ORBX_HIP(h, hipMemcpy(out_pairs, d_pairs, sizeof(int) * 2 * (size_t)*n_out, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_keyframe_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                              const orbx_keyframe* const* kfs, int T, double radius_scale, unsigned desc_threshold, int* out_idx,
                              uint32_t* out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || P < 0 || T < 0 || (P > 0 && (!positions || !mp_desc)) || (T > 0 && !kfs) || (P > 0 && T > 0 && (!out_idx || !out_dist)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_keyframe_fuse_search: bad argument");
  if (P == 0 || T == 0) return ORBX_OK;
  std::vector<int> off((size_t)T + 1, 0);
  std::vector<double> poses(7 * (size_t)T);
  for (int t = 0; t < T; ++t) {
    if (!kfs[t] || kfs[t]->h != h) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_keyframe_fuse_search: keyframe %d is null or of another handle", t);
    off[(size_t)t + 1] = off[(size_t)t] + kfs[t]->n;
    memcpy(&poses[7 * (size_t)t], kfs[t]->pose_wc, 56);
  }
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t n = (size_t)std::max(off[(size_t)T], 1), pt = (size_t)P * T;
  // the target keyframes' features side by side (device-to-device, no host hop), the map points and the result block
  if (int rc = orbx_reserve(h, h->ws_io[8], sizeof(orbx_keypoint) * n + 32 * n + 32 * (size_t)P + 24 * (size_t)P + 4 * ((size_t)T + 1) + 8 * pt + 64)) return rc;
  uint8_t* b = (uint8_t*)h->ws_io[8].p;
  double* d_pos = (double*)b; b += 24 * (size_t)P;
  int* d_idx = (int*)b; b += 4 * pt; uint32_t* d_dist = (uint32_t*)b; b += 4 * pt;
  orbx_keypoint* d_kps = (orbx_keypoint*)b; b += sizeof(orbx_keypoint) * n;
  uint8_t* d_descs = b; b += 32 * n;
  uint8_t* d_mpd = b; b += 32 * (size_t)P;
  int* d_off = (int*)b;
  hipStream_t st = h->stream;
  for (int t = 0; t < T; ++t)
    if (kfs[t]->n > 0) {
      ORBX_HIP(h, hipMemcpyAsync(d_kps + off[(size_t)t], kfs[t]->d_kp, sizeof(orbx_keypoint) * (size_t)kfs[t]->n, hipMemcpyDeviceToDevice, st));
      ORBX_HIP(h, hipMemcpyAsync(d_descs + 32 * (size_t)off[(size_t)t], kfs[t]->d_desc, 32 * (size_t)kfs[t]->n, hipMemcpyDeviceToDevice, st));
    }
  ORBX_HIP(h, hipMemcpyAsync(d_pos, positions, 24 * (size_t)P, hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d_mpd, mp_desc, 32 * (size_t)P, hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d_off, off.data(), 4 * ((size_t)T + 1), hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipStreamSynchronize(st));                               // `off` is a local
  if (int rc = orbx_fuse_search_device(h, cam, d_pos, d_mpd, P, poses.data(), d_off, d_kps, d_descs, T, radius_scale, desc_threshold, d_idx, d_dist))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(out_idx, d_idx, 4 * pt, hipMemcpyDeviceToHost, st));
  ORBX_HIP(h, hipMemcpyAsync(out_dist, d_dist, 4 * pt, hipMemcpyDeviceToHost, st));
  ORBX_HIP(h, hipStreamSynchronize(st));
  return ORBX_OK;
}

}  // extern "C"
