// orbx_api.hip — the C ABI of include/orbx.h: handle lifetime, host<->device staging, status,
// per-kernel timing.  Host logic only; kernels live in match_kernels.hip / orb_kernels.hip /
// ba_kernels.hip.  There is no CPU fallback anywhere in this library.
#include <cstdlib>
#include <mutex>

#include <algorithm>
#include <vector>

#include <rccl/rccl.h>   // types only: the entry points are bound with dlopen on first use (rccl_api below)
#include <dlfcn.h>
#include <thread>
#include <system_error>
#include <new>

#include "orbx_internal.hpp"


static thread_local std::string g_create_error;

int orbx_fail(orbx_handle* h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_error = buf;
  return code;
}

int orbx_reserve(orbx_handle* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.bytes) return ORBX_OK;
  // any captured graph holds the addresses of the workspaces: a re-allocation invalidates it
  if (h->pair_graph) { hipGraphExecDestroy(h->pair_graph); h->pair_graph = nullptr; }
  h->pg_calls = 0;
  if (b.p) {
    ORBX_HIP(h, hipStreamSynchronize(h->stream));
    ORBX_HIP(h, hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
  }
  const size_t want = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
  ORBX_HIP(h, hipMalloc(&b.p, want));
  b.bytes = want;
  return ORBX_OK;
}

// ---- profiling --------------------------------------------------------------------------------------
static hipEvent_t prof_event(orbx_handle* h) {
  if (h->event_next == h->event_pool.size()) {
    hipEvent_t e;
    // timing only: no system-scope release (cache write-back) at every kernel boundary
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) hipEventCreate(&e);
    h->event_pool.push_back(e);
  }
  return h->event_pool[h->event_next++];
}
ProfScope::ProfScope(orbx_handle* h_, const char* name, hipStream_t stream_, bool chained)
    : h(h_), idx(-1), stream(stream_ ? stream_ : h_->stream) {
  if (!h->profiling) return;
  if (!h->prof_only.empty() && h->prof_only != name) { h->prof_tail = nullptr; return; }   // (launches follow that no scope brackets: the next bracketed one records its own start)
  for (size_t i = 0; i < h->timers.size(); ++i)
    if (h->timers[i].name == name) { idx = (int)i; break; }
  if (idx < 0) {
    KernelTimer t;
    t.name = name;
    h->timers.push_back(t);
    idx = (int)h->timers.size() - 1;
  }
  // chained: nothing was enqueued on this stream since the previous scope ended, so that scope's end event is this
  // one's start (half the events in a back-to-back kernel sequence: every event costs the stream ~1 us)
  if (chained && h->prof_tail && h->prof_tail_stream == stream) {
    h->timers[idx].ev.push_back(h->prof_tail);
    return;
  }
  hipEvent_t e = prof_event(h);
  hipEventRecord(e, stream);
  h->timers[idx].ev.push_back(e);
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  hipEvent_t e = prof_event(h);
  hipEventRecord(e, stream);
  h->timers[idx].ev.push_back(e);
  h->prof_tail = e;
  h->prof_tail_stream = stream;
}
void orbx_prof_begin_call(orbx_handle* h) { (void)h; }   // events accumulate until they are read
void orbx_prof_end_call(orbx_handle* h) { (void)h; }

extern "C" {

#define ORBX_STR2(x) #x
#define ORBX_STR(x) ORBX_STR2(x)
const char* orbx_version(void) { return "orbx-mi355x 0.1 (gfx950, abi " ORBX_STR(ORBX_ABI_VERSION) ")"; }
int orbx_abi_version(void) { return ORBX_ABI_VERSION; }

const char* orbx_last_error(const orbx_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void orbx_default_orb_params(int n_features, orbx_orb_params* p) {
  // stereo.rs:38-48
  p->n_features = n_features; p->scale_factor = 1.2f; p->n_levels = 8; p->edge_threshold = 31;
  p->first_level = 0; p->wta_k = 2; p->score_type = 0; p->patch_size = 31; p->fast_threshold = 20;
}

void orbx_default_ba_config(orbx_ba_config* c) {
  // LocalBAConfigLM::default, local_ba_lm.rs:109-119
  c->max_iterations = 10; c->param_tolerance = 1e-8; c->gradient_tolerance = 1e-8;
  c->huber_threshold = sqrt(5.991); c->max_covisible_keyframes = 20;
}

int orbx_create(const orbx_camera* cam, const orbx_orb_params* orb, int device, int max_w, int max_h,
                int max_batch, orbx_handle** out) {
  if (!cam || !orb || !out) return orbx_fail(nullptr, ORBX_ERR_INVALID, "null argument");
  *out = nullptr;
  if (max_w < 64 || max_h < 64 || max_w > 4095 || max_h > 4095 || max_batch < 1)
    return orbx_fail(nullptr, ORBX_ERR_INVALID, "image bounds must be 64..4095, max_batch >= 1");
  if (orb->n_levels < 1 || orb->n_levels > ORBX_MAX_LEVELS || orb->edge_threshold != 31 ||
      orb->first_level != 0 || orb->wta_k != 2 || orb->score_type != 0 || orb->patch_size != 31 ||
      orb->n_features < 0 || orb->fast_threshold < 1 || orb->fast_threshold > 254 ||
      !(orb->scale_factor > 1.0f) || orb->scale_factor > 1.5f)
    return orbx_fail(nullptr, ORBX_ERR_INVALID,
                     "only the reference's ORB configuration is implemented (stereo.rs:38-48): "
                     "n_levels<=8, scale factor in (1, 1.5], edge 31, first_level 0, WTA_K 2, HARRIS_SCORE, patch 31");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return orbx_fail(nullptr, ORBX_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU path");
  if (device < 0 || device >= ndev)
    return orbx_fail(nullptr, ORBX_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    return orbx_fail(nullptr, ORBX_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return orbx_fail(nullptr, ORBX_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                     device, prop.gcnArchName);
  orbx_handle* h = new orbx_handle();
  h->device = device; h->cam = *cam; h->orb = *orb;
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  h->max_w = max_w; h->max_h = max_h; h->max_batch = max_batch;
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&h->d_status, sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void**)&h->h_status, sizeof(unsigned)) != hipSuccess ||
      hipMemsetAsync(h->d_status, 0, sizeof(unsigned), h->stream) != hipSuccess) {
    const int rc = orbx_fail(nullptr, ORBX_ERR_HIP, "stream/status allocation failed: %s",
                             hipGetErrorString(hipGetLastError()));
    delete h;
    return rc;
  }
  *out = h;
  return ORBX_OK;
}

void orbx_destroy(orbx_handle* h) {
  if (!h) return;
  if (h->ba_aux) { orbx_destroy(h->ba_aux); h->ba_aux = nullptr; }
  delete h->ba_pool; h->ba_pool = nullptr;
  delete h->ba_helper; h->ba_helper = nullptr;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  DevBuf* bufs[] = {&h->resize_tab, &h->ws_pyr, &h->ws_blur, &h->ws_cand, &h->ws_counters,
                    &h->ws_sel, &h->ws_sel2, &h->ws_match, &h->ws_dtile};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  for (DevBuf& b : h->ws_io) if (b.p) hipFree(b.p);
  for (DevBuf& b : h->ws_ba) if (b.p) hipFree(b.p);
  for (auto& set : h->ws_pipe) for (DevBuf& b : set) if (b.p) hipFree(b.p);
  for (int i = 0; i < 2; ++i) { if (h->ev_in[i]) hipEventDestroy(h->ev_in[i]); if (h->ev_comp[i]) hipEventDestroy(h->ev_comp[i]); if (h->ev_out[i]) hipEventDestroy(h->ev_out[i]); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->ba_up_event) hipEventDestroy(h->ba_up_event);
  for (hipEvent_t e : h->ev_stag) if (e) hipEventDestroy(e);
  if (h->s_aux) hipStreamDestroy(h->s_aux);
  if (h->s_in) hipStreamDestroy(h->s_in);
  if (h->s_out) hipStreamDestroy(h->s_out);
  for (hipEvent_t e : h->event_pool) hipEventDestroy(e);
  if (h->pair_graph) hipGraphExecDestroy(h->pair_graph);
  orbx_rccl_drop(h);
  if (h->h_stage) hipHostFree(h->h_stage);
  if (h->h_ba_in) hipHostFree(h->h_ba_in);
  if (h->h_ba_out) hipHostFree(h->h_ba_out);
  if (h->h_abort) hipHostFree(h->h_abort);
  if (h->d_status) hipFree(h->d_status);
  if (h->h_status) hipHostFree(h->h_status);
  hipStreamDestroy(h->stream);
  delete h;
}

void* orbx_stream(orbx_handle* h) { return h ? (void*)h->stream : nullptr; }

int orbx_synchronize(orbx_handle* h) {
  if (!h) return ORBX_ERR_INVALID;
  ORBX_HIP(h, hipSetDevice(h->device));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

int orbx_check_status(orbx_handle* h) {
  if (!h) return ORBX_ERR_INVALID;
  ORBX_HIP(h, hipSetDevice(h->device));
  ORBX_HIP(h, hipMemcpyAsync(h->h_status, h->d_status, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemsetAsync(h->d_status, 0, sizeof(unsigned), h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  const unsigned st = *h->h_status;
  if (st & ORBX_ST_KP_OVERFLOW)
    return orbx_fail(h, ORBX_ERR_CAPACITY, "an image produced more keypoints than cap_kp");
  if (st) return orbx_fail(h, ORBX_ERR_HIP, "device status 0x%x", st);
  return ORBX_OK;
}

int orbx_set_profiling(orbx_handle* h, int on) {
  if (!h) return ORBX_ERR_INVALID;
  h->profiling = on != 0;
  h->prof_only.clear();
  h->event_next = 0;
  h->prof_tail = nullptr;
  for (auto& t : h->timers) t.ev.clear();
  return ORBX_OK;
}

int orbx_set_profiling_only(orbx_handle* h, const char* kernel_name) {
  if (!h) return ORBX_ERR_INVALID;
  const int rc = orbx_set_profiling(h, 1);
  if (rc != ORBX_OK) return rc;
  try { h->prof_only = kernel_name ? kernel_name : ""; } catch (...) { return orbx_fail(h, ORBX_ERR_HIP, "orbx_set_profiling_only: out of host memory"); }
  return ORBX_OK;
}

int orbx_get_kernel_times(orbx_handle* h, orbx_kernel_time* out, int cap) {
  if (!h) return ORBX_ERR_INVALID;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->s_aux) hipStreamSynchronize(h->s_aux);
  int n = 0;
  for (auto& t : h->timers) {
    if (t.ev.empty()) continue;
    t.ms = 0.f;
    t.launches = (int)t.ev.size() / 2;
    for (size_t i = 0; i + 1 < t.ev.size(); i += 2) {
      float ms = 0.f;
      hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]);
      t.ms += ms;
    }
    if (n < cap && out) {
      memset(&out[n], 0, sizeof(out[n]));
      strncpy(out[n].name, t.name.c_str(), sizeof(out[n].name) - 1);
      out[n].ms = t.ms;
      out[n].launches = t.launches;
    }
    ++n;
  }
  // reading resets: the next call starts a new accumulation window
  h->event_next = 0;
  h->prof_tail = nullptr;
  for (auto& t : h->timers) t.ev.clear();
  return n;
}

// ---- matchers -----------------------------------------------------------------------------------------

int orbx_stereo_match_batch_device(orbx_handle* h, int batch, const orbx_keypoint* d_kp,
                                   const uint8_t* d_desc, const int* d_nkp, int cap_kp,
                                   orbx_dmatch* d_matches, int* d_nmatches, double* d_points,
                                   uint8_t* d_has_point) {
  if (!h) return ORBX_ERR_INVALID;
  if (batch < 0 || cap_kp < 1 || !d_kp || !d_desc || !d_nkp || !d_matches || !d_nmatches || !d_points || !d_has_point)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_stereo_match_batch_device: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  return launch_stereo_match(h, batch, d_kp, d_desc, d_nkp, cap_kp, d_matches, d_nmatches, d_points, d_has_point);
}

int orbx_stereo_match(orbx_handle* h, const orbx_keypoint* kpL, const uint8_t* descL, int nL,
                      const orbx_keypoint* kpR, const uint8_t* descR, int nR, orbx_dmatch* matches,
                      int* n_matches, double* points_cam, uint8_t* has_point) {
  if (!h) return ORBX_ERR_INVALID;
  if (nL < 0 || nR < 0 || !n_matches || (nL > 0 && (!kpL || !descL || !matches || !points_cam || !has_point)) ||
      (nR > 0 && (!kpR || !descR)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_stereo_match: bad argument");
  *n_matches = 0;
  if (nL == 0) return ORBX_OK;   // stereo.rs:95 loop body never runs
  ORBX_HIP(h, hipSetDevice(h->device));
  const int cap = nL > nR ? nL : nR;
  const size_t kpb = sizeof(orbx_keypoint) * (size_t)cap, db = 32 * (size_t)cap;
  if (int rc = orbx_reserve(h, h->ws_io[0], 2 * kpb)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 2 * db)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], 4 * sizeof(int))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], sizeof(orbx_dmatch) * (size_t)cap)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[4], sizeof(double) * 3 * (size_t)cap)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[5], (size_t)cap)) return rc;
  orbx_keypoint* d_kp = (orbx_keypoint*)h->ws_io[0].p;
  uint8_t* d_desc = (uint8_t*)h->ws_io[1].p;
  int* d_n = (int*)h->ws_io[2].p;
  const int counts[2] = {nL, nR};
  ORBX_HIP(h, hipMemcpyAsync(d_kp, kpL, sizeof(orbx_keypoint) * (size_t)nL, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_desc, descL, 32 * (size_t)nL, hipMemcpyHostToDevice, h->stream));
  if (nR > 0) {
    ORBX_HIP(h, hipMemcpyAsync(d_kp + cap, kpR, sizeof(orbx_keypoint) * (size_t)nR, hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(h, hipMemcpyAsync(d_desc + db, descR, 32 * (size_t)nR, hipMemcpyHostToDevice, h->stream));
  }
  ORBX_HIP(h, hipMemcpyAsync(d_n, counts, sizeof(counts), hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));  // `counts` is a stack buffer
  orbx_prof_begin_call(h);
  if (int rc = launch_stereo_match(h, 1, d_kp, d_desc, d_n, cap, (orbx_dmatch*)h->ws_io[3].p, d_n + 2,
                                   (double*)h->ws_io[4].p, (uint8_t*)h->ws_io[5].p))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(n_matches, d_n + 2, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(points_cam, h->ws_io[4].p, sizeof(double) * 3 * (size_t)nL, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(has_point, h->ws_io[5].p, (size_t)nL, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (*n_matches > 0)
    ORBX_HIP(h, hipMemcpy(matches, h->ws_io[3].p, sizeof(orbx_dmatch) * (size_t)*n_matches, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_hamming_match_crosscheck_device(orbx_handle* h, const uint8_t* d_q, int nq, const uint8_t* d_t,
                                         int nt, orbx_dmatch* d_out, int* d_n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (nq < 0 || nt < 0 || !d_n_out || (nq > 0 && (!d_q || !d_out)) || (nt > 0 && !d_t))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_hamming_match_crosscheck_device: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  return launch_crosscheck(h, d_q, nq, d_t, nt, d_out, d_n_out);
}

int orbx_hamming_match_crosscheck(orbx_handle* h, const uint8_t* q, int nq, const uint8_t* t, int nt,
                                  orbx_dmatch* out, int* n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (nq < 0 || nt < 0 || !n_out || (nq > 0 && (!q || !out)) || (nt > 0 && !t))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_hamming_match_crosscheck: bad argument");
  *n_out = 0;
  if (nq == 0 || nt == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  if (int rc = orbx_reserve(h, h->ws_io[0], 32 * (size_t)nq)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 32 * (size_t)nt)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], sizeof(int))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], sizeof(orbx_dmatch) * (size_t)nq)) return rc;
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[0].p, q, 32 * (size_t)nq, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[1].p, t, 32 * (size_t)nt, hipMemcpyHostToDevice, h->stream));
  orbx_prof_begin_call(h);
  if (int rc = launch_crosscheck(h, (const uint8_t*)h->ws_io[0].p, nq, (const uint8_t*)h->ws_io[1].p, nt,
                                 (orbx_dmatch*)h->ws_io[3].p, (int*)h->ws_io[2].p))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(n_out, h->ws_io[2].p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (*n_out > 0)
    ORBX_HIP(h, hipMemcpy(out, h->ws_io[3].p, sizeof(orbx_dmatch) * (size_t)*n_out, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_hamming_batch_device(orbx_handle* h, const uint8_t* d_a, const uint8_t* d_b, int n_pairs,
                              uint32_t* d_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (n_pairs < 0 || (n_pairs > 0 && (!d_a || !d_b || !d_out)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_hamming_batch_device: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  return launch_hamming_batch(h, d_a, d_b, n_pairs, d_out);
}

int orbx_hamming_batch(orbx_handle* h, const uint8_t* a, const uint8_t* b, int n_pairs, uint32_t* out) {
  if (!h) return ORBX_ERR_INVALID;
  if (n_pairs < 0 || (n_pairs > 0 && (!a || !b || !out)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_hamming_batch: bad argument");
  if (n_pairs == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t bytes = 32 * (size_t)n_pairs;
  if (int rc = orbx_reserve(h, h->ws_io[0], bytes)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], bytes)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], sizeof(uint32_t) * (size_t)n_pairs)) return rc;
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[0].p, a, bytes, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[1].p, b, bytes, hipMemcpyHostToDevice, h->stream));
  orbx_prof_begin_call(h);
  if (int rc = launch_hamming_batch(h, (const uint8_t*)h->ws_io[0].p, (const uint8_t*)h->ws_io[1].p, n_pairs,
                                    (uint32_t*)h->ws_io[2].p))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(out, h->ws_io[2].p, sizeof(uint32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

int orbx_guided_match_device(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n, double img_w,
                             double img_h, const double* d_q_uv, const uint8_t* d_q_desc, int nq, double radius, int mode,
                             int* d_out_idx, uint32_t* d_out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (n < 0 || nq < 0 || (mode != 0 && mode != 1) || !(img_w > 0.0) || !(img_h > 0.0) || !(radius >= 0.0) ||
      (n > 0 && (!d_kp || !d_desc)) || (nq > 0 && (!d_q_uv || !d_q_desc || !d_out_idx || !d_out_dist)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_guided_match_device: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  return launch_guided_match(h, d_kp, d_desc, n, img_w, img_h, d_q_uv, d_q_desc, nq, radius, mode, d_out_idx, d_out_dist);
}

int orbx_guided_match(orbx_handle* h, const orbx_keypoint* kp, const uint8_t* desc, int n, double img_w, double img_h,
                      const double* q_uv, const uint8_t* q_desc, int nq, double radius, int mode, int* out_idx,
                      uint32_t* out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (n < 0 || nq < 0 || (mode != 0 && mode != 1) || !(img_w > 0.0) || !(img_h > 0.0) || !(radius >= 0.0) ||
      (n > 0 && (!kp || !desc)) || (nq > 0 && (!q_uv || !q_desc || !out_idx || !out_dist)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_guided_match: bad argument");
  if (nq == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t nn = (size_t)(n > 0 ? n : 1);
  if (int rc = orbx_reserve(h, h->ws_io[0], sizeof(orbx_keypoint) * nn)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 32 * nn)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], 16 * (size_t)nq)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], 32 * (size_t)nq)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[4], 8 * (size_t)nq)) return rc;
  if (n > 0) {
    ORBX_HIP(h, hipMemcpyAsync(h->ws_io[0].p, kp, sizeof(orbx_keypoint) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(h, hipMemcpyAsync(h->ws_io[1].p, desc, 32 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  }
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[2].p, q_uv, 16 * (size_t)nq, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[3].p, q_desc, 32 * (size_t)nq, hipMemcpyHostToDevice, h->stream));
  int* d_idx = (int*)h->ws_io[4].p;
  uint32_t* d_dist = (uint32_t*)(d_idx + nq);
  orbx_prof_begin_call(h);
  if (int rc = launch_guided_match(h, (const orbx_keypoint*)h->ws_io[0].p, (const uint8_t*)h->ws_io[1].p, n, img_w, img_h,
                                   (const double*)h->ws_io[2].p, (const uint8_t*)h->ws_io[3].p, nq, radius, mode, d_idx, d_dist))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(out_idx, d_idx, 4 * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_dist, d_dist, 4 * (size_t)nq, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

namespace {
struct Quat { double w, x, y, z; };
void quat_rotate(const Quat& q, const double* v, double* o) {            // nalgebra UnitQuaternion * Vector3
  const double t[3] = {2.0 * (q.y * v[2] - q.z * v[1]), 2.0 * (q.z * v[0] - q.x * v[2]), 2.0 * (q.x * v[1] - q.y * v[0])};
  const double c[3] = {q.y * t[2] - q.z * t[1], q.z * t[0] - q.x * t[2], q.x * t[1] - q.y * t[0]};
  for (int i = 0; i < 3; ++i) o[i] = t[i] * q.w + c[i] + v[i];
}
void mat3_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
// epipole of camera 1 in image 2 and F = K^-T [t12]x R12 K^-1, exactly as triangulation.rs:418-431, :661-683
void triangulation_geometry(const orbx_camera& cam, const double* p1, const double* p2, double* ep, double* F) {
  const Quat q1{p1[0], p1[1], p1[2], p1[3]}, q2{p2[0], p2[1], p2[2], p2[3]};
  const Quat q2i{q2.w, -q2.x, -q2.y, -q2.z}, q1i{q1.w, -q1.x, -q1.y, -q1.z};
  double r[3], c1[3], rt[3];
  quat_rotate(q2i, p2 + 4, r);
  const double t2i[3] = {-r[0], -r[1], -r[2]};                           // pose2.inverse().translation (se3.rs:56-63)
  quat_rotate(q2i, p1 + 4, c1);
  c1[0] += t2i[0]; c1[1] += t2i[1]; c1[2] += t2i[2];
  ep[0] = cam.fx * c1[0] / c1[2] + cam.cx;
  ep[1] = cam.fy * c1[1] / c1[2] + cam.cy;
  quat_rotate(q2i, p1 + 4, rt);
  const double t12[3] = {t2i[0] - rt[0], t2i[1] - rt[1], t2i[2] - rt[2]};
  const Quat r12{q2i.w * q1i.w - q2i.x * q1i.x - q2i.y * q1i.y - q2i.z * q1i.z, q2i.w * q1i.x + q2i.x * q1i.w + q2i.y * q1i.z - q2i.z * q1i.y,
                 q2i.w * q1i.y - q2i.x * q1i.z + q2i.y * q1i.w + q2i.z * q1i.x, q2i.w * q1i.z + q2i.x * q1i.y - q2i.y * q1i.x + q2i.z * q1i.w};
  const double w = r12.w, i = r12.x, j = r12.y, k = r12.z;
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k, ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0,
               jk = j * k * 2.0, wi = w * i * 2.0;
  const double R[9] = {ww + ii - jj - kk, ij - wk, wj + ik, wk + ij, ww - ii + jj - kk, jk - wi, ik - wj, wi + jk, ww - ii - jj + kk};
  const double tsk[9] = {0.0, -t12[2], t12[1], t12[2], 0.0, -t12[0], -t12[1], t12[0], 0.0};
  const double Ki[9] = {1.0 / cam.fx, 0.0, -cam.cx / cam.fx, 0.0, 1.0 / cam.fy, -cam.cy / cam.fy, 0.0, 0.0, 1.0};
  double E[9], KiT[9], T[9];
  mat3_mul(tsk, R, E);
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) KiT[a * 3 + b] = Ki[b * 3 + a];
  mat3_mul(KiT, E, T);
  mat3_mul(T, Ki, F);
}
}  // namespace

int orbx_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1,
                                  const uint8_t* mp1, const uint8_t* stereo1, int n1, const orbx_keypoint* kp2,
                                  const uint8_t* desc2, const uint8_t* mp2, int n2, const double* pose1_wc,
                                  const double* pose2_wc, unsigned max_dist, int* out_pairs, int* n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || n1 < 0 || n2 < 0 || n2 > 65535 * 64 || !pose1_wc || !pose2_wc || !n_out || max_dist > 256 ||
      (n1 > 0 && (!kp1 || !desc1 || !mp1 || !stereo1 || !out_pairs)) || (n2 > 0 && (!kp2 || !desc2 || !mp2)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_search_for_triangulation: bad argument");
  *n_out = 0;
  if (n1 == 0 || n2 == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  double ep[2], F[9];
  triangulation_geometry(*cam, pose1_wc, pose2_wc, ep, F);
  const size_t s1 = sizeof(orbx_keypoint) * (size_t)n1, s2 = sizeof(orbx_keypoint) * (size_t)n2;
  if (int rc = orbx_reserve(h, h->ws_io[0], s1 + s2)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 32 * ((size_t)n1 + n2))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], 2 * (size_t)n1 + n2)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], sizeof(int) * (2 * (size_t)n1 + 4))) return rc;
  orbx_keypoint* d_kp1 = (orbx_keypoint*)h->ws_io[0].p; orbx_keypoint* d_kp2 = d_kp1 + n1;
  uint8_t* d_d1 = (uint8_t*)h->ws_io[1].p; uint8_t* d_d2 = d_d1 + 32 * (size_t)n1;
  uint8_t* d_mp1 = (uint8_t*)h->ws_io[2].p; uint8_t* d_st1 = d_mp1 + n1; uint8_t* d_mp2 = d_st1 + n1;
  int* d_pairs = (int*)h->ws_io[3].p; int* d_n = d_pairs + 2 * (size_t)n1;
  ORBX_HIP(h, hipMemcpyAsync(d_kp1, kp1, s1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_kp2, kp2, s2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_d1, desc1, 32 * (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_d2, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_mp1, mp1, (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_st1, stereo1, (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_mp2, mp2, (size_t)n2, hipMemcpyHostToDevice, h->stream));
  orbx_prof_begin_call(h);
  if (int rc = launch_search_for_triangulation(h, cam, F, ep, d_kp1, d_d1, d_mp1, d_st1, n1, d_kp2, d_d2, d_mp2, n2, max_dist, d_pairs, d_n))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (*n_out > 0) ORBX_HIP(h, hipMemcpy(out_pairs, d_pairs, sizeof(int) * 2 * (size_t)*n_out, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_search_for_triangulation_device(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* d_kp1, const uint8_t* d_desc1,
                                         const uint8_t* d_mp1, const uint8_t* d_stereo1, int n1, const orbx_keypoint* d_kp2,
                                         const uint8_t* d_desc2, const uint8_t* d_mp2, int n2, const double* pose1_wc,
                                         const double* pose2_wc, unsigned max_dist, int* d_pairs, int* d_n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || n1 < 0 || n2 < 0 || n2 > 65535 * 64 || !pose1_wc || !pose2_wc || !d_n_out || max_dist > 256 ||
      (n1 > 0 && (!d_kp1 || !d_desc1 || !d_mp1 || !d_stereo1 || !d_pairs)) || (n2 > 0 && (!d_kp2 || !d_desc2 || !d_mp2)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_search_for_triangulation_device: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  double ep[2], F[9];
  triangulation_geometry(*cam, pose1_wc, pose2_wc, ep, F);
  orbx_prof_begin_call(h);
  return launch_search_for_triangulation(h, cam, F, ep, d_kp1, d_desc1, d_mp1, d_stereo1, n1, d_kp2, d_desc2, d_mp2, n2, max_dist, d_pairs,
                                         d_n_out);
}

int orbx_search_for_triangulation_bow(orbx_handle* h, const orbx_camera* cam, const orbx_keypoint* kp1, const uint8_t* desc1,
                                      const uint8_t* mp1, const uint8_t* stereo1, const uint32_t* node1, int n1,
                                      const orbx_keypoint* kp2, const uint8_t* desc2, const uint8_t* mp2, const uint32_t* node2, int n2,
                                      const double* pose1_wc, const double* pose2_wc, unsigned max_dist, int* out_pairs, int* n_out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || n1 < 0 || n2 < 0 || !pose1_wc || !pose2_wc || !n_out || max_dist > 256 ||
      (n1 > 0 && (!kp1 || !desc1 || !mp1 || !stereo1 || !node1 || !out_pairs)) || (n2 > 0 && (!kp2 || !desc2 || !mp2 || !node2)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_search_for_triangulation_bow: bad argument");
  *n_out = 0;
  if (n1 == 0 || n2 == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  double ep[2], F[9];
  triangulation_geometry(*cam, pose1_wc, pose2_wc, ep, F);
  // FeatureVector of keyframe 2 as one sorted array: (node, index) ascending = every node's list in push order (:309)
  std::vector<int> sorted((size_t)n2), lo((size_t)n1, 0), hi((size_t)n1, 0);
  for (int i = 0; i < n2; ++i) sorted[(size_t)i] = i;
  std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return node2[a] < node2[b]; });
  int m2 = n2;                                                           // features in no list sort last and are cut off
  while (m2 > 0 && node2[sorted[(size_t)m2 - 1]] == 0xffffffffu) --m2;
  for (int i = 0; i < n1; ++i) {
    if (node1[i] == 0xffffffffu) continue;
    const uint32_t key = node1[i];
    const auto b = std::lower_bound(sorted.begin(), sorted.begin() + m2, key, [&](int a, uint32_t k) { return node2[a] < k; });
    const auto e = std::upper_bound(b, sorted.begin() + m2, key, [&](uint32_t k, int a) { return k < node2[a]; });
    lo[(size_t)i] = (int)(b - sorted.begin()); hi[(size_t)i] = (int)(e - sorted.begin());
  }
  const size_t s1 = sizeof(orbx_keypoint) * (size_t)n1, s2 = sizeof(orbx_keypoint) * (size_t)n2;
  if (int rc = orbx_reserve(h, h->ws_io[0], s1 + s2)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 32 * ((size_t)n1 + n2))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], 2 * (size_t)n1 + n2)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], sizeof(int) * (2 * (size_t)n1 + 4))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[4], sizeof(int) * (2 * (size_t)n1 + (size_t)n2))) return rc;
  orbx_keypoint* d_kp1 = (orbx_keypoint*)h->ws_io[0].p; orbx_keypoint* d_kp2 = d_kp1 + n1;
  uint8_t* d_d1 = (uint8_t*)h->ws_io[1].p; uint8_t* d_d2 = d_d1 + 32 * (size_t)n1;
  uint8_t* d_mp1 = (uint8_t*)h->ws_io[2].p; uint8_t* d_st1 = d_mp1 + n1; uint8_t* d_mp2 = d_st1 + n1;
  int* d_pairs = (int*)h->ws_io[3].p; int* d_n = d_pairs + 2 * (size_t)n1;
  int* d_lo = (int*)h->ws_io[4].p; int* d_hi = d_lo + n1; int* d_sorted = d_hi + n1;
  ORBX_HIP(h, hipMemcpyAsync(d_kp1, kp1, s1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_kp2, kp2, s2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_d1, desc1, 32 * (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_d2, desc2, 32 * (size_t)n2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_mp1, mp1, (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_st1, stereo1, (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_mp2, mp2, (size_t)n2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_hi, hi.data(), sizeof(int) * (size_t)n1, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_sorted, sorted.data(), sizeof(int) * (size_t)n2, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));   // lo/hi/sorted are locals
  orbx_prof_begin_call(h);
  if (int rc = launch_search_for_triangulation_bow(h, F, ep, d_kp1, d_d1, d_mp1, d_st1, n1, d_kp2, d_d2, d_mp2, n2, d_sorted, d_lo, d_hi,
                                                   max_dist, d_pairs, d_n))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (*n_out > 0) ORBX_HIP(h, hipMemcpy(out_pairs, d_pairs, sizeof(int) * 2 * (size_t)*n_out, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_fuse_search_device(orbx_handle* h, const orbx_camera* cam, const double* d_positions, const uint8_t* d_mp_desc, int P,
                            const double* kf_poses_wc, const int* d_kf_feat_offset, const orbx_keypoint* d_kps, const uint8_t* d_descs, int T,
                            double radius_scale, unsigned desc_threshold, int* d_out_idx, uint32_t* d_out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || P < 0 || T < 0 || (P > 0 && (!d_positions || !d_mp_desc)) || (T > 0 && (!kf_poses_wc || !d_kf_feat_offset)) ||
      (P > 0 && T > 0 && (!d_out_idx || !d_out_dist || !d_kps || !d_descs)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_fuse_search_device: bad argument");
  if (P == 0 || T == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  std::vector<double> cw(7 * (size_t)T);                                 // kf.pose.inverse() (se3.rs:56-63)
  for (int t = 0; t < T; ++t) {
    const double* p = kf_poses_wc + 7 * (size_t)t;
    const Quat qi{p[0], -p[1], -p[2], -p[3]};
    double r[3];
    quat_rotate(qi, p + 4, r);
    double* o = cw.data() + 7 * (size_t)t;
    o[0] = qi.w; o[1] = qi.x; o[2] = qi.y; o[3] = qi.z; o[4] = -r[0]; o[5] = -r[1]; o[6] = -r[2];
  }
  if (int rc = orbx_reserve(h, h->ws_io[5], sizeof(double) * 7 * (size_t)T)) return rc;
  ORBX_HIP(h, hipMemcpyAsync(h->ws_io[5].p, cw.data(), sizeof(double) * 7 * (size_t)T, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));   // cw is a local
  orbx_prof_begin_call(h);
  return launch_fuse_search(h, cam, d_positions, d_mp_desc, P, (const double*)h->ws_io[5].p, d_kf_feat_offset, d_kps, d_descs, T, radius_scale,
                            desc_threshold, d_out_idx, d_out_dist);
}

int orbx_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* positions, const uint8_t* mp_desc, int P,
                     const double* kf_poses_wc, const int* kf_feat_offset, const orbx_keypoint* kps, const uint8_t* descs, int T,
                     double radius_scale, unsigned desc_threshold, int* out_idx, uint32_t* out_dist) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || P < 0 || T < 0 || (P > 0 && (!positions || !mp_desc)) || (T > 0 && (!kf_poses_wc || !kf_feat_offset)) ||
      (P > 0 && T > 0 && (!out_idx || !out_dist)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_fuse_search: bad argument");
  if (P == 0 || T == 0) return ORBX_OK;
  if (kf_feat_offset[0] != 0) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_fuse_search: kf_feat_offset[0] must be 0");
  for (int t = 0; t < T; ++t)
    if (kf_feat_offset[t + 1] < kf_feat_offset[t]) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_fuse_search: kf_feat_offset not ascending");
  const int n = kf_feat_offset[T];
  if (n > 0 && (!kps || !descs)) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_fuse_search: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  std::vector<double> cw(7 * (size_t)T);                                 // kf.pose.inverse() (se3.rs:56-63)
  for (int t = 0; t < T; ++t) {
    const double* p = kf_poses_wc + 7 * (size_t)t;
    const Quat qi{p[0], -p[1], -p[2], -p[3]};
    double r[3];
    quat_rotate(qi, p + 4, r);
    double* o = cw.data() + 7 * (size_t)t;
    o[0] = qi.w; o[1] = qi.x; o[2] = qi.y; o[3] = qi.z; o[4] = -r[0]; o[5] = -r[1]; o[6] = -r[2];
  }
  const size_t pt = (size_t)P * T;
  if (int rc = orbx_reserve(h, h->ws_io[0], sizeof(orbx_keypoint) * (size_t)std::max(n, 1))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[1], 32 * ((size_t)std::max(n, 1) + P))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[2], sizeof(double) * (3 * (size_t)P + 7 * (size_t)T) + sizeof(int) * ((size_t)T + 1))) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[3], 8 * pt)) return rc;
  orbx_keypoint* d_kps = (orbx_keypoint*)h->ws_io[0].p;
  uint8_t* d_descs = (uint8_t*)h->ws_io[1].p; uint8_t* d_mpd = d_descs + 32 * (size_t)std::max(n, 1);
  double* d_pos = (double*)h->ws_io[2].p; double* d_cw = d_pos + 3 * (size_t)P; int* d_off = (int*)(d_cw + 7 * (size_t)T);
  int* d_idx = (int*)h->ws_io[3].p; uint32_t* d_dist = (uint32_t*)(d_idx + pt);
  if (n > 0) {
    ORBX_HIP(h, hipMemcpyAsync(d_kps, kps, sizeof(orbx_keypoint) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    ORBX_HIP(h, hipMemcpyAsync(d_descs, descs, 32 * (size_t)n, hipMemcpyHostToDevice, h->stream));
  }
  ORBX_HIP(h, hipMemcpyAsync(d_mpd, mp_desc, 32 * (size_t)P, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_pos, positions, sizeof(double) * 3 * (size_t)P, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_cw, cw.data(), sizeof(double) * 7 * (size_t)T, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(d_off, kf_feat_offset, sizeof(int) * ((size_t)T + 1), hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));   // cw is a local; pageable copies are staged but keep it simple
  orbx_prof_begin_call(h);
  if (int rc = launch_fuse_search(h, cam, d_pos, d_mpd, P, d_cw, d_off, d_kps, d_descs, T, radius_scale, desc_threshold, d_idx, d_dist))
    return rc;
  ORBX_HIP(h, hipMemcpyAsync(out_idx, d_idx, sizeof(int) * pt, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipMemcpyAsync(out_dist, d_dist, sizeof(uint32_t) * pt, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  return ORBX_OK;
}

// ---- extraction + full per-frame path -------------------------------------------------------------------

int orbx_extract_batch_device(orbx_handle* h, const uint8_t* d_images, int n_images, int w, int h_px,
                              size_t stride, orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp, int cap_kp) {
  if (!h) return ORBX_ERR_INVALID;
  if (!d_images || !d_kp || !d_desc || !d_nkp || n_images < 0 || cap_kp < 1)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_extract_batch_device: bad argument");
  if (w < 64 || h_px < 64 || w > h->max_w || h_px > h->max_h || stride < (size_t)w)
    return orbx_fail(h, ORBX_ERR_INVALID, "image %dx%d (stride %zu) outside the handle's bounds %dx%d", w, h_px,
                     stride, h->max_w, h->max_h);
  if (n_images > 2 * h->max_batch)
    return orbx_fail(h, ORBX_ERR_INVALID, "n_images %d exceeds 2*max_batch %d", n_images, 2 * h->max_batch);
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  return launch_orb_extract(h, d_images, n_images, w, h_px, stride, d_kp, d_desc, d_nkp, cap_kp);
}

int orbx_process_stereo_batch_device(orbx_handle* h, const uint8_t* d_images, int batch, int w, int h_px,
                                     size_t stride, orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp,
                                     int cap_kp, orbx_dmatch* d_matches, int* d_nmatches, double* d_points,
                                     uint8_t* d_has_point) {
  if (!h) return ORBX_ERR_INVALID;
  if (batch < 0 || batch > h->max_batch)
    return orbx_fail(h, ORBX_ERR_INVALID, "batch %d outside 0..max_batch %d", batch, h->max_batch);
  if (!d_matches || !d_nmatches || !d_points || !d_has_point)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_process_stereo_batch_device: bad argument");
  // Two-stream form (ORBX_STAGGER=<chunks>, 2..8; experimental, round 5): the batch as `chunks` ranges of pairs dealt alternately to the
  // handle's stream and a second one, chunk c + 1 starting behind chunk c's pyramid launches, so that one range's latency-bound
  // launches (resize chain, Harris, ordering, the matcher's three) are in flight while the other range is in its issue-bound FAST /
  // describe kernels.  Pairs are independent and every workspace is indexed by image or pair: the results are those of one range over
  // the whole batch, bit for bit.  The handle's stream waits for the second one before the call returns (work the caller enqueues on
  // orbx_stream() afterwards sees every result).  MEASURED NEGATIVE (profiles/r05_stagger_two_streams_negative.txt): 145.2 k frames/s on one
  // stream, 144.6 k as 2 ranges, 139.5 k as 4 — under overlap every kernel stretches by what the other stream takes (describe x 2.1, FAST
  // x 1.6, Harris x 1.5, resize x 1.8): the latency-bound launches do not hide under the issue-bound ones, the chip is shared.  Off by default.
  const int stagger = [] { const char* e = getenv("ORBX_STAGGER"); const int v = e ? atoi(e) : 0; return v >= 2 && v <= 8 ? v : 0; }();   // (read per call, as ORBX_FORK_BLUR)
  static const bool stagger_prof = getenv("ORBX_STAGGER_PROFILE") != nullptr;    // (per-kernel events on both streams: they then time the overlap, not the kernels)
  static const bool desc_unfused = getenv("ORBX_DESC_UNFUSED") != nullptr;
  if (stagger && batch >= 2 * stagger && (!h->profiling || stagger_prof) && !desc_unfused) {
    if (!d_images || !d_kp || !d_desc || !d_nkp || cap_kp < 1)
      return orbx_fail(h, ORBX_ERR_INVALID, "orbx_process_stereo_batch_device: bad argument");
    if (w < 64 || h_px < 64 || w > h->max_w || h_px > h->max_h || stride < (size_t)w)
      return orbx_fail(h, ORBX_ERR_INVALID, "image %dx%d (stride %zu) outside the handle's bounds %dx%d", w, h_px, stride, h->max_w, h->max_h);
    ORBX_HIP(h, hipSetDevice(h->device));
    if (!h->s_aux) {
      ORBX_HIP(h, hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    for (auto& e : h->ev_stag) if (!e) ORBX_HIP(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (int rc = orb_extract_prepare(h, 2 * batch, w, h_px)) return rc;
    // (the matcher's workspace is sized before the streams fork: orbx_reserve may synchronise)
    if (int rc = launch_stereo_match_range(h, h->stream, batch, 0, 0, d_kp, d_desc, d_nkp, cap_kp, d_matches, d_nmatches, d_points, d_has_point)) return rc;
    ORBX_HIP(h, hipEventRecord(h->ev_fork, h->stream));
    ORBX_HIP(h, hipStreamWaitEvent(h->s_aux, h->ev_fork, 0));
    int p0 = 0;
    for (int c = 0; c < stagger; ++c) {
      const int p1 = (int)((long long)batch * (c + 1) / stagger);
      hipStream_t st = (c & 1) ? h->s_aux : h->stream;
      if (c > 0) ORBX_HIP(h, hipStreamWaitEvent(st, h->ev_stag[(c - 1) & 3], 0));     // behind the previous chunk's pyramid
      if (int rc = orb_extract_range(h, st, d_images, 2 * batch, 2 * p0, 2 * (p1 - p0), w, h_px, stride, d_kp, d_desc, d_nkp, cap_kp, h->ev_stag[c & 3])) return rc;
      if (int rc = launch_stereo_match_range(h, st, batch, p0, p1 - p0, d_kp, d_desc, d_nkp, cap_kp, d_matches, d_nmatches, d_points, d_has_point)) return rc;
      p0 = p1;
    }
    ORBX_HIP(h, hipEventRecord(h->ev_join, h->s_aux));
    ORBX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    return ORBX_OK;
  }
  if (int rc = orbx_extract_batch_device(h, d_images, 2 * batch, w, h_px, stride, d_kp, d_desc, d_nkp, cap_kp))
    return rc;
  return launch_stereo_match(h, batch, d_kp, d_desc, d_nkp, cap_kp, d_matches, d_nmatches, d_points, d_has_point);
}

int orbx_process_stereo(orbx_handle* h, const uint8_t* left, size_t lstride, const uint8_t* right,
                        size_t rstride, int w, int h_px, orbx_keypoint* kpL, uint8_t* descL, int* nL,
                        orbx_keypoint* kpR, uint8_t* descR, int* nR, int cap_kp, orbx_dmatch* matches,
                        int* n_matches, double* points_cam, uint8_t* has_point) {
  if (!h) return ORBX_ERR_INVALID;
  if (!left || !right || !kpL || !descL || !nL || !kpR || !descR || !nR || !matches || !n_matches ||
      !points_cam || !has_point || cap_kp < 1 || lstride < (size_t)w || rstride < (size_t)w)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_process_stereo: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  const size_t img = (size_t)w * h_px, cp = (size_t)cap_kp;
  // every output of the pair in ONE device block (and one pinned host mirror): a single D2H copy and a single
  // synchronisation per frame instead of one blocking copy per array
  const size_t o_pts = 0;                                   // double[cp*3]
  const size_t o_kp = o_pts + 24 * cp;                      // orbx_keypoint[2*cp]
  const size_t o_m = o_kp + sizeof(orbx_keypoint) * 2 * cp; // orbx_dmatch[cp]
  const size_t o_cnt = o_m + sizeof(orbx_dmatch) * cp;      // int nkp[2], nmatches, status
  const size_t o_desc = o_cnt + 16;                         // u8[2*cp*32]
  const size_t o_has = o_desc + 64 * cp;                    // u8[cp]
  const size_t total = (o_has + cp + 63) & ~(size_t)63;
  if (int rc = orbx_reserve(h, h->ws_io[6], 2 * img)) return rc;
  if (int rc = orbx_reserve(h, h->ws_io[7], total)) return rc;
  if (h->h_stage_bytes < total) {
    if (h->h_stage) ORBX_HIP(h, hipHostFree(h->h_stage));
    h->h_stage = nullptr; h->h_stage_bytes = 0;
    ORBX_HIP(h, hipHostMalloc((void**)&h->h_stage, total));
    h->h_stage_bytes = total;
  }
  uint8_t* d_img = (uint8_t*)h->ws_io[6].p;
  uint8_t* d_out = (uint8_t*)h->ws_io[7].p;
  ORBX_HIP(h, hipMemcpy2DAsync(d_img, w, left, lstride, w, h_px, hipMemcpyHostToDevice, h->stream));
  ORBX_HIP(h, hipMemcpy2DAsync(d_img + img, w, right, rstride, w, h_px, hipMemcpyHostToDevice, h->stream));
  int* d_cnt = (int*)(d_out + o_cnt);
  auto enqueue_device_part = [&]() -> int {
    if (int rc = orbx_process_stereo_batch_device(h, d_img, 1, w, h_px, (size_t)w, (orbx_keypoint*)(d_out + o_kp), d_out + o_desc,
                                                  d_cnt, cap_kp, (orbx_dmatch*)(d_out + o_m), d_cnt + 2, (double*)(d_out + o_pts),
                                                  d_out + o_has))
      return rc;
    ORBX_HIP(h, hipMemcpyAsync(d_cnt + 3, h->d_status, sizeof(unsigned), hipMemcpyDeviceToDevice, h->stream));
    ORBX_HIP(h, hipMemsetAsync(h->d_status, 0, sizeof(unsigned), h->stream));
    return ORBX_OK;
  };
  // The device part (memset + ~20 short kernels) is launch-bound for one pair: after two eager calls with an
  // unchanged configuration (all workspaces allocated, tables uploaded) it is captured into a hipGraph once and
  // replayed with a single launch per frame.  ORBX_NO_GRAPH=1 or profiling keeps the eager path.
  static const bool no_graph = getenv("ORBX_NO_GRAPH") != nullptr;
  // (the geometry test covers an extract / batch call at another image size in between: it rewrites the tables the
  // graph's kernels read — orb_prepare_geometry drops the graph itself, this keeps the eager-call count honest too)
  const bool same = h->pg_w == w && h->pg_h == h_px && h->pg_cap == cap_kp && h->pg_img == (void*)d_img && h->pg_out == (void*)d_out &&
                    h->geom_w == w && h->geom_h == h_px;
  if (!same) {
    if (h->pair_graph) { hipGraphExecDestroy(h->pair_graph); h->pair_graph = nullptr; }
    h->pg_w = w; h->pg_h = h_px; h->pg_cap = cap_kp; h->pg_img = d_img; h->pg_out = d_out; h->pg_calls = 0;
  }
  if (no_graph || h->profiling) {
    if (int rc = enqueue_device_part()) return rc;
  } else if (h->pair_graph) {
    ORBX_HIP(h, hipGraphLaunch(h->pair_graph, h->stream));
  } else if (h->pg_calls < 2) {
    ++h->pg_calls;
    if (int rc = enqueue_device_part()) return rc;
  } else {
    hipGraph_t graph = nullptr;
    ORBX_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_device_part();
    const hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc != ORBX_OK || e != hipSuccess || !graph) {
      if (graph) hipGraphDestroy(graph);
      h->pg_calls = -1000000;      // capture is not possible here: stay eager
      (void)hipGetLastError();
      if (int rc2 = enqueue_device_part()) return rc2;
    } else {
      const hipError_t ei = hipGraphInstantiate(&h->pair_graph, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (ei != hipSuccess) { h->pair_graph = nullptr; h->pg_calls = -1000000; (void)hipGetLastError(); if (int rc2 = enqueue_device_part()) return rc2; }
      else ORBX_HIP(h, hipGraphLaunch(h->pair_graph, h->stream));
    }
  }
  ORBX_HIP(h, hipMemcpyAsync(h->h_stage, d_out, total, hipMemcpyDeviceToHost, h->stream));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  const uint8_t* S = h->h_stage;
  const int* cnt = (const int*)(S + o_cnt);
  const unsigned st = (unsigned)cnt[3];
  if (st & ORBX_ST_KP_OVERFLOW) return orbx_fail(h, ORBX_ERR_CAPACITY, "an image produced more keypoints than cap_kp");
  if (st) return orbx_fail(h, ORBX_ERR_HIP, "device status 0x%x", st);
  *nL = cnt[0]; *nR = cnt[1]; *n_matches = cnt[2];
  memcpy(kpL, S + o_kp, sizeof(orbx_keypoint) * (size_t)cnt[0]);
  memcpy(kpR, S + o_kp + sizeof(orbx_keypoint) * cp, sizeof(orbx_keypoint) * (size_t)cnt[1]);
  memcpy(descL, S + o_desc, 32 * (size_t)cnt[0]);
  memcpy(descR, S + o_desc + 32 * cp, 32 * (size_t)cnt[1]);
  memcpy(matches, S + o_m, sizeof(orbx_dmatch) * (size_t)cnt[2]);
  memcpy(points_cam, S + o_pts, 24 * (size_t)cnt[0]);
  memcpy(has_point, S + o_has, (size_t)cnt[0]);
  return ORBX_OK;
}

void* orbx_host_alloc(size_t bytes) {
  void* p = nullptr;
  return hipHostMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}
void orbx_host_free(void* p) { if (p) hipHostFree(p); }

int orbx_process_stereo_batch(orbx_handle* h, const uint8_t* images, int batch, int w, int h_px, size_t stride,
                              orbx_keypoint* kp, uint8_t* desc, int* nkp, int cap_kp, orbx_dmatch* matches, int* nmatches,
                              double* points, uint8_t* has_point) {
  if (!h) return ORBX_ERR_INVALID;
  if (batch < 0 || cap_kp < 1 || !images || !kp || !desc || !nkp || !matches || !nmatches || !points || !has_point ||
      stride < (size_t)w)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_process_stereo_batch: bad argument");
  if (batch == 0) return ORBX_OK;
  ORBX_HIP(h, hipSetDevice(h->device));
  if (!h->s_in) {
    ORBX_HIP(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    ORBX_HIP(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_comp[i], hipEventDisableTiming));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_out[i], hipEventDisableTiming));
    }
  }
  const int C = h->max_batch < 32 ? h->max_batch : 32;   // pairs per chunk
  const size_t img_b = (size_t)h_px * stride * 2;          // one pair
  const size_t sz[8] = {img_b * C, sizeof(orbx_keypoint) * 2 * (size_t)cap_kp * C, 64 * (size_t)cap_kp * C, sizeof(int) * 2 * (size_t)C,
                        sizeof(orbx_dmatch) * (size_t)cap_kp * C, sizeof(int) * (size_t)C, 24 * (size_t)cap_kp * C, (size_t)cap_kp * C};
  for (int b = 0; b < 2; ++b)
    for (int i = 0; i < 8; ++i) if (int rc = orbx_reserve(h, h->ws_pipe[b][i], sz[i])) return rc;
  // workspaces of the extractor are sized before the pipeline starts (orbx_reserve may synchronise)
  const int n_chunks = (batch + C - 1) / C;
  for (int c = 0; c < n_chunks; ++c) {
    const int b = c & 1, p0 = c * C, np = (batch - p0 < C) ? batch - p0 : C;
    DevBuf* d = h->ws_pipe[b];
    if (c >= 2) ORBX_HIP(h, hipStreamWaitEvent(h->s_in, h->ev_comp[b], 0));    // image buffer b is free again
    ORBX_HIP(h, hipMemcpyAsync(d[0].p, images + (size_t)p0 * img_b, img_b * np, hipMemcpyHostToDevice, h->s_in));
    ORBX_HIP(h, hipEventRecord(h->ev_in[b], h->s_in));
    ORBX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_in[b], 0));
    if (c >= 2) ORBX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_out[b], 0));  // output buffers b have been downloaded
    if (int rc = orbx_process_stereo_batch_device(h, (const uint8_t*)d[0].p, np, w, h_px, stride, (orbx_keypoint*)d[1].p,
                                                  (uint8_t*)d[2].p, (int*)d[3].p, cap_kp, (orbx_dmatch*)d[4].p, (int*)d[5].p,
                                                  (double*)d[6].p, (uint8_t*)d[7].p))
      return rc;
    ORBX_HIP(h, hipEventRecord(h->ev_comp[b], h->stream));
    ORBX_HIP(h, hipStreamWaitEvent(h->s_out, h->ev_comp[b], 0));
    const size_t cp = (size_t)cap_kp;
    ORBX_HIP(h, hipMemcpyAsync(kp + 2 * cp * p0, d[1].p, sizeof(orbx_keypoint) * 2 * cp * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(desc + 64 * cp * p0, d[2].p, 64 * cp * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(nkp + 2 * (size_t)p0, d[3].p, sizeof(int) * 2 * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(matches + cp * p0, d[4].p, sizeof(orbx_dmatch) * cp * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(nmatches + p0, d[5].p, sizeof(int) * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(points + 3 * cp * p0, d[6].p, 24 * cp * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipMemcpyAsync(has_point + cp * p0, d[7].p, cp * np, hipMemcpyDeviceToHost, h->s_out));
    ORBX_HIP(h, hipEventRecord(h->ev_out[b], h->s_out));
  }
  ORBX_HIP(h, hipStreamSynchronize(h->s_out));
  return orbx_check_status(h);
}

// ---- local BA ---------------------------------------------------------------------------------------------

int orbx_ba_set_allreduce(orbx_handle* h, orbx_allreduce_fn fn, void* user) {
  if (!h) return ORBX_ERR_INVALID;
  h->allreduce = fn;
  h->allreduce_user = user;
  return ORBX_OK;
}

// ---- native collective of the point-partitioned solve: RCCL over xGMI -------------------------
// librccl is bound on first use (dlopen by soname), not at load time: a single-GPU user of this library never needs it, and a
// process in which PyTorch already mapped a librccl.so.1 gets THAT copy back from the loader — one RCCL per process, so a
// ncclComm_t handed over by orbx_ba_set_rccl_comm belongs to the library that will use it.
}  // extern "C"
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;            // (optional: only orbx_ba_rccl_world asks)
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  char err[256] = {0};
};
RcclApi g_rccl_api;
RcclApi* rccl_api() {
  static std::once_flag once;
  std::call_once(once, [] {
    RcclApi& api = g_rccl_api;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) { snprintf(api.err, sizeof(api.err), "librccl.so.1 not found (%s)", dlerror()); return; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
      snprintf(api.err, sizeof(api.err), "librccl: a collective entry point is missing");
      api.lib = nullptr;
    }
  });
  return g_rccl_api.lib ? &g_rccl_api : nullptr;
}
const char* rccl_why() { rccl_api(); return g_rccl_api.err[0] ? g_rccl_api.err : "RCCL is not available in this process"; }
}  // namespace
extern "C" {

int orbx_rccl_unique_id(uint8_t* out, size_t cap) {
  if (!out || cap < sizeof(ncclUniqueId)) return ORBX_ERR_INVALID;
  RcclApi* R = rccl_api();
  if (!R) return ORBX_ERR_HIP;
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return ORBX_ERR_HIP;
  memcpy(out, &id, sizeof(id));
  return (int)sizeof(id);
}

}  // extern "C"
void orbx_rccl_drop(orbx_handle* h) {
  if (h->rccl_comm && h->rccl_owned) { RcclApi* R = rccl_api(); if (R) R->CommDestroy((ncclComm_t)h->rccl_comm); }
  h->rccl_comm = nullptr; h->rccl_owned = false;
}
extern "C" {

int orbx_ba_init_rccl(orbx_handle* h, const uint8_t* unique_id, size_t id_bytes, int rank, int world) {
  if (!h) return ORBX_ERR_INVALID;
  if (!unique_id || id_bytes != sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_init_rccl: bad argument (the id is %zu bytes)", sizeof(ncclUniqueId));
  RcclApi* R = rccl_api();
  if (!R) return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_init_rccl: %s", rccl_why());
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_rccl_drop(h);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = R->CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return orbx_fail(h, ORBX_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString(r));
  h->rccl_comm = comm; h->rccl_owned = true;
  return ORBX_OK;
}

int orbx_ba_set_rccl_comm(orbx_handle* h, void* nccl_comm) {
  if (!h) return ORBX_ERR_INVALID;
  if (nccl_comm && !rccl_api()) return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_set_rccl_comm: %s", rccl_why());
  orbx_rccl_drop(h);
  h->rccl_comm = nccl_comm; h->rccl_owned = false;
  return ORBX_OK;
}

int orbx_ba_has_collective(orbx_handle* h) {
  if (!h) return 0;
  return (h->rccl_comm ? 1 : 0) | (h->allreduce ? 2 : 0);
}

int orbx_ba_rccl_world(orbx_handle* h, int* n_ranks, int* rank) {
  if (!h || !n_ranks || !rank) return ORBX_ERR_INVALID;
  *n_ranks = 0; *rank = -1;
  if (!h->rccl_comm) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_rccl_world: the handle holds no RCCL communicator");
  RcclApi* R = rccl_api();
  if (!R || !R->CommCount || !R->CommUserRank) return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_rccl_world: %s", R ? "librccl has no ncclCommCount" : rccl_why());
  ncclResult_t r = R->CommCount((ncclComm_t)h->rccl_comm, n_ranks);
  if (r == ncclSuccess) r = R->CommUserRank((ncclComm_t)h->rccl_comm, rank);
  if (r != ncclSuccess) return orbx_fail(h, ORBX_ERR_HIP, "ncclCommCount failed: %s", R->GetErrorString(r));
  return ORBX_OK;
}

}  // extern "C"
int orbx_rccl_allreduce_sum(orbx_handle* h, double* d_buf, size_t n, hipStream_t st) {
  RcclApi* R = rccl_api();
  if (!R) return orbx_fail(h, ORBX_ERR_HIP, "ncclAllReduce: %s", rccl_why());
  const ncclResult_t r = R->AllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, (ncclComm_t)h->rccl_comm, st);
  if (r != ncclSuccess) return orbx_fail(h, ORBX_ERR_HIP, "ncclAllReduce failed: %s", R->GetErrorString(r));
  return ORBX_OK;
}
extern "C" {

int orbx_ba_solve_visual(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                         const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                         double* points, int N, const orbx_ba_obs* obs, orbx_should_stop_fn should_stop,
                         void* user, double* poses_wc_out, int* iterations, double* initial_error,
                         double* final_error) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || F < 0 || M < 0 || N < 0 || !iterations || !initial_error || !final_error ||
      (K > 0 && (!poses_cw || !poses_wc_out)) || (F > 0 && !fixed_poses_cw) || (M > 0 && !points) ||
      (N > 0 && !obs))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  try {
    return ba_solve_visual(h, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, should_stop, user,
                           poses_wc_out, iterations, initial_error, final_error);
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual: unexpected C++ exception");
  }
}

int orbx_ba_solve_visual_obs32(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                               const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                               double* points, int N, const orbx_ba_obs32* obs32, orbx_should_stop_fn should_stop,
                               void* user, double* poses_wc_out, int* iterations, double* initial_error,
                               double* final_error) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || F < 0 || M < 0 || N < 0 || !iterations || !initial_error || !final_error ||
      (K > 0 && (!poses_cw || !poses_wc_out)) || (F > 0 && !fixed_poses_cw) || (M > 0 && !points) ||
      (N > 0 && !obs32))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual_obs32: bad argument");
  if (h->allreduce || h->rccl_comm) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual_obs32: the partitioned solve takes orbx_ba_obs (orbx_ba_solve_visual)");
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  try {
    return ba_solve_visual(h, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, nullptr, should_stop, user,
                           poses_wc_out, iterations, initial_error, final_error, false, nullptr, obs32);
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual_obs32: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual_obs32: unexpected C++ exception");
  }
}

int orbx_ba_solve_visual_batch(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int n_windows,
                               orbx_ba_window* windows, orbx_should_stop_fn should_stop, void* user) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || n_windows < 0 || (n_windows > 0 && !windows))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual_batch: bad argument");
  if (n_windows > 65535) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual_batch: at most 65535 windows per call");
  // (no C++ exception may cross this C boundary: allocation or thread-creation failure is reported as an error code)
  try {
    std::vector<BaWinHost> w((size_t)n_windows);
    for (int i = 0; i < n_windows; ++i) {
      orbx_ba_window& q = windows[i];
      if (q.K < 0 || q.F < 0 || q.M < 0 || q.N < 0 || (q.K > 0 && (!q.poses_cw || !q.poses_wc_out)) || (q.F > 0 && !q.fixed_poses_cw) ||
          (q.M > 0 && !q.points) || (q.N > 0 && !q.obs && !q.obs32))
        return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_visual_batch: window %d: bad argument", i);
      w[i] = BaWinHost{q.K, q.F, q.M, q.N, q.poses_cw, q.fixed_poses_cw, q.points, q.obs, q.poses_wc_out, &q.iterations,
                       &q.initial_error, &q.final_error, ORBX_OK, q.obs32};
    }
    ORBX_HIP(h, hipSetDevice(h->device));
    orbx_prof_begin_call(h);
    orbx_allreduce_fn saved = h->allreduce;          // independent windows: no collective
    void* saved_comm = h->rccl_comm;
    h->allreduce = nullptr; h->rccl_comm = nullptr;
    struct Restore { orbx_handle* h; orbx_allreduce_fn f; void* c; ~Restore() { h->allreduce = f; h->rccl_comm = c; } } restore{h, saved, saved_comm};
    // A large batch runs as two halves at once — this handle and a second, internal one (its own stream, workspaces and pinned blobs), the
    // second half driven by a helper thread — so that one half's host preprocessing, upload and download run under the other half's kernels
    // and the latency-bound launches of one fill the gaps of the other.  Windows are independent and a window's arithmetic never depends on
    // the batch it travels in, so the results are the same bit for bit.  Not with a should_stop callback (it would be called from two
    // threads), not while per-kernel profiling is on (the kernel times belong to one handle).  (Three / four parts on three / four
    // streams: 39 / 37 k LM iterations/s against 46 k with two and 41 k with one, at 32 windows.)  The halves are cut where the
    // OBSERVATION count is halved, not the window count: the streaming kernels' time follows the observations.
    // If a half fails the call fails as a whole (every window's status = the error) — the in/out `points` and `poses_wc_out` of windows in
    // the half that did finish then already hold its results; a caller that retries must hand in the original points again (orbx.h).
    static const bool no_split = getenv("ORBX_BA_NO_SPLIT") != nullptr;
    int rc;
    bool split = !no_split && n_windows >= 16 && !should_stop && !h->profiling;
    if (split && !h->ba_aux) {
      const int rc_aux = orbx_create(&h->cam, &h->orb, h->device, h->max_w, h->max_h, 1, &h->ba_aux);
      if (rc_aux != ORBX_OK) { h->ba_aux = nullptr; split = false; }      // no second stream: the whole batch on this one
    }
    if (split) {
      size_t total = 0, run = 0;
      for (int i = 0; i < n_windows; ++i) total += (size_t)w[i].N;
      // the first part's upload is the one nothing hides (the second part's travels under the first part's kernels): ORBX_BA_SPLIT_FRAC
      // (default 0.5) moves the cut for A/B runs
      static const double split_frac = [] { const char* e = getenv("ORBX_BA_SPLIT_FRAC"); const double f = e ? atof(e) : 0.5; return f > 0.05 && f < 0.95 ? f : 0.5; }();
      int n0 = 0;
      while (n0 < n_windows - 1 && (double)(run + (size_t)w[n0].N) - 0.5 * (double)w[n0].N <= split_frac * (double)total) run += (size_t)w[n0++].N;
      n0 = std::max(1, std::min(n_windows - 1, n0));
      int rc1 = ORBX_OK;
      // the two halves preprocess at the same time: half the cores each  (one after the other with all the cores each, so that the first
      // half's kernels run under the second half's preprocessing, was built: 16 threads sort a half in 0.33 ms where 8 take 0.41, so the
      // second half was ready at 0.7 ms instead of 0.45 — no gain)
      const int half_cores = std::max(1, (int)std::thread::hardware_concurrency() / 2);
      h->ba_pool_cap = half_cores; h->ba_aux->ba_pool_cap = half_cores;
      h->ba_peer_windows = n_windows - n0; h->ba_aux->ba_peer_windows = n0;
      // one PCIe link: the second half's uploads go behind the first half's (orbx_internal.hpp: ba_gate_*)
      std::atomic<int> gate{0};
      if (!h->ba_up_event && hipEventCreateWithFlags(&h->ba_up_event, hipEventDisableTiming) != hipSuccess) h->ba_up_event = nullptr;
      h->ba_gate_signal = &gate; h->ba_aux->ba_gate_wait = &gate; h->ba_aux->ba_gate_event = h->ba_up_event;
      struct Uncap { orbx_handle* h; ~Uncap() { h->ba_pool_cap = 0; h->ba_aux->ba_pool_cap = 0; h->ba_peer_windows = 0; h->ba_aux->ba_peer_windows = 0;
                                                h->ba_gate_signal = nullptr; h->ba_aux->ba_gate_wait = nullptr; h->ba_aux->ba_gate_event = nullptr; } } uncap{h};
      if (!h->ba_helper) {
        try { h->ba_helper = new OrbxHelperThread(); } catch (...) { h->ba_helper = nullptr; }
      }
      if (h->ba_helper) {
        h->ba_helper->start([&] {
          hipSetDevice(h->device);
          try { rc1 = ba_solve_batch(h->ba_aux, cam, cfg, n_windows - n0, w.data() + n0, nullptr, nullptr); }
          catch (...) { rc1 = ORBX_ERR_HIP; }
        });
        try { rc = ba_solve_batch(h, cam, cfg, n0, w.data(), nullptr, nullptr); }
        catch (...) { rc = orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual_batch: out of host memory"); }
        gate.store(1, std::memory_order_release);                              // (however the first half ended, the second must not wait for it)
        h->ba_helper->wait();
      } else {                                                              // no thread: the whole batch here, on one stream
        h->ba_gate_signal = nullptr; h->ba_peer_windows = 0; h->ba_pool_cap = 0;
        rc = ba_solve_batch(h, cam, cfg, n_windows, w.data(), nullptr, nullptr);
      }
      if (rc == ORBX_OK && rc1 != ORBX_OK) rc = orbx_fail(h, rc1, "(windows %d..%d, numbered from %d) %s", n0, n_windows - 1, n0, orbx_last_error(h->ba_aux));
    } else rc = ba_solve_batch(h, cam, cfg, n_windows, w.data(), should_stop, user);
    for (int i = 0; i < n_windows; ++i) windows[i].status = rc == ORBX_OK ? w[i].status : rc;
    return rc;
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual_batch: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_visual_batch: unexpected C++ exception");
  }
}

int orbx_debug_ba_blocks(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                         const double* fixed_poses_cw, int M, const double* points, int N, const orbx_ba_obs* obs, int global_mode,
                         double* out) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || F < 0 || M < 0 || N < 0 || (K > 0 && !poses_cw) || (F > 0 && !fixed_poses_cw) || (M > 0 && !points) ||
      (N > 0 && (!obs || !out)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_debug_ba_blocks: bad argument");
  if (K > 128) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_debug_ba_blocks: at most 128 keyframes");
  ORBX_HIP(h, hipSetDevice(h->device));
  return ba_debug_blocks(h, cam, cfg, K, poses_cw, F, fixed_poses_cw, M, points, N, obs, global_mode, out);
}

int orbx_debug_imu_residual(orbx_handle* h, int K, const double* poses_wc, const double* velocities, int E, const int* edge_kf,
                            const double* preint, double* out) {
  if (!h) return ORBX_ERR_INVALID;
  if (K < 0 || E < 0 || (K > 0 && (!poses_wc || !velocities)) || (E > 0 && (!edge_kf || !preint || !out)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_debug_imu_residual: bad argument");
  ORBX_HIP(h, hipSetDevice(h->device));
  return ba_debug_imu_residual(h, K, poses_wc, velocities, E, edge_kf, preint, out);
}

void orbx_default_inertial_ba_config(orbx_inertial_ba_config* c) {       // local_inertial_ba.rs:126-141
  if (!c) return;
  c->max_iterations = 10; c->window_size = 10;
  c->huber_threshold_mono = std::sqrt(5.991); c->huber_threshold_stereo = std::sqrt(7.815);
  c->initial_lambda = 1e-2; c->gyro_rw_info = 1e6; c->accel_rw_info = 1e4;
}

int orbx_ba_solve_inertial(orbx_handle* h, const orbx_camera* cam, const orbx_inertial_ba_config* cfg, int K, const double* poses_wc,
                           const double* velocities, const double* biases, int F, const double* fixed_poses_cw, int M, double* points,
                           int N, const orbx_ba_obs* obs, int E, const int* edge_kf, const double* preint,
                           orbx_should_stop_fn should_stop, void* user, double* poses_wc_out, double* vel_out, double* bias_out,
                           int* iterations, double* initial_error, double* final_error) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || F < 0 || M < 0 || N < 0 || E < 0 || !iterations || !initial_error || !final_error ||
      (K > 0 && (!poses_wc || !velocities || !biases || !poses_wc_out || !vel_out || !bias_out)) || (F > 0 && !fixed_poses_cw) ||
      (M > 0 && !points) || (N > 0 && !obs) || (E > 0 && (!edge_kf || !preint)))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_inertial: bad argument");
  *iterations = 0; *initial_error = 0.0; *final_error = 0.0;
  if (K < 2) return orbx_fail(h, ORBX_ERR_EMPTY, "inertial BA needs two keyframes in the window");   // local_inertial_ba.rs:1080-1082
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  orbx_ba_config vc{cfg->max_iterations, 0.0, 1e-8, cfg->huber_threshold_mono, 0};
  BaInertialHost in{cfg, velocities, biases, E, edge_kf, preint, vel_out, bias_out};
  try {
    return ba_solve_visual(h, cam, &vc, K, poses_wc, F, fixed_poses_cw, M, points, N, obs, should_stop, user, poses_wc_out, iterations,
                           initial_error, final_error, false, &in);
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_inertial: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_inertial: unexpected C++ exception");
  }
}

int orbx_ba_solve_global(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw,
                         const double* fixed_pose_cw, int M, double* points, int N, const orbx_ba_obs* obs,
                         orbx_should_stop_fn should_stop, void* user, double* poses_wc_out, int* iterations,
                         double* initial_error, double* final_error) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || M < 0 || N < 0 || !iterations || !initial_error || !final_error || !fixed_pose_cw ||
      (K > 0 && (!poses_cw || !poses_wc_out)) || (M > 0 && !points) || (N > 0 && !obs))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_global: bad argument");
  *iterations = 0; *initial_error = 0.0; *final_error = 0.0;
  if (K < 1 || M == 0) return orbx_fail(h, ORBX_ERR_EMPTY, "global BA needs two keyframes and a map point");   // global_ba.rs:194-196
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  try {
    return ba_solve_visual(h, cam, cfg, K, poses_cw, 1, fixed_pose_cw, M, points, N, obs, should_stop, user, poses_wc_out,
                           iterations, initial_error, final_error, true);
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_global: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_global: unexpected C++ exception");
  }
}

int orbx_ba_solve_global_obs32(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw,
                               const double* fixed_pose_cw, int M, double* points, int N, const orbx_ba_obs32* obs32,
                               orbx_should_stop_fn should_stop, void* user, double* poses_wc_out, int* iterations,
                               double* initial_error, double* final_error) {
  if (!h) return ORBX_ERR_INVALID;
  if (!cam || !cfg || K < 0 || M < 0 || N < 0 || !iterations || !initial_error || !final_error || !fixed_pose_cw ||
      (K > 0 && (!poses_cw || !poses_wc_out)) || (M > 0 && !points) || (N > 0 && !obs32))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_global_obs32: bad argument");
  if (h->allreduce || h->rccl_comm) return orbx_fail(h, ORBX_ERR_INVALID, "orbx_ba_solve_global_obs32: the partitioned solve takes orbx_ba_obs (orbx_ba_solve_global)");
  *iterations = 0; *initial_error = 0.0; *final_error = 0.0;
  if (K < 1 || M == 0) return orbx_fail(h, ORBX_ERR_EMPTY, "global BA needs two keyframes and a map point");   // global_ba.rs:194-196
  ORBX_HIP(h, hipSetDevice(h->device));
  orbx_prof_begin_call(h);
  try {
    return ba_solve_visual(h, cam, cfg, K, poses_cw, 1, fixed_pose_cw, M, points, N, nullptr, should_stop, user, poses_wc_out,
                           iterations, initial_error, final_error, true, nullptr, obs32);
  } catch (const std::bad_alloc&) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_global_obs32: out of host memory");
  } catch (...) {
    return orbx_fail(h, ORBX_ERR_HIP, "orbx_ba_solve_global_obs32: unexpected C++ exception");
  }
}

}  // extern "C"
