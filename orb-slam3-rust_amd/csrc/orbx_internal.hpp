// orbx_internal.hpp — shared host-side declarations of the HIP library (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbx.h"

#define ORBX_MAX_LEVELS 8

// Persistent host workers of a handle (the batch BA call's per-window preprocessing: creating and joining 16 threads per call cost
// more than the 0.4 ms of work each of them then did).  run(items, want, f) calls f(0..items-1), each index once, on the calling thread
// and up to `want` workers, and returns when all are done; an exception inside f is reported by the return value (false), never thrown
// across the workers.  Not reentrant: one run() at a time per pool (a handle is used by one thread at a time, orbx.h).
class OrbxWorkPool {
 public:
  explicit OrbxWorkPool(int workers) {
    for (int i = 0; i < workers; ++i) th_.emplace_back([this, i] { loop(i); });      // std::system_error if a thread cannot be created
  }
  ~OrbxWorkPool() {
    { std::lock_guard<std::mutex> g(m_); quit_ = true; ++gen_; }
    go_.notify_all();
    for (auto& t : th_) t.join();
  }
  int workers() const { return (int)th_.size(); }
  bool run(int items, int want, const std::function<void(int)>& f) {
    want = want < 0 ? 0 : (want > (int)th_.size() ? (int)th_.size() : want);
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f; items_ = items; want_ = want; active_ = want; failed_ = false;
      next_.store(0, std::memory_order_relaxed);
      ++gen_;
    }
    if (want > 0) go_.notify_all();
    take();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return active_ == 0; });
    job_ = nullptr;
    return !failed_;
  }

 private:
  void take() {
    try { for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < items_;) (*job_)(i); }
    catch (...) { std::lock_guard<std::mutex> g(m_); failed_ = true; }
  }
  void loop(int idx) {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(m_);
        go_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
        if (idx >= want_) continue;                                     // this round runs on fewer workers
      }
      take();
      { std::lock_guard<std::mutex> g(m_); if (--active_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable go_, done_;
  const std::function<void(int)>* job_ = nullptr;
  std::atomic<int> next_{0};
  int items_ = 0, want_ = 0, active_ = 0;
  unsigned long long gen_ = 0;
  bool quit_ = false, failed_ = false;
};

// One persistent helper thread of a handle: start(f) hands it a job and returns, wait() blocks until the job is done.  (The second half of
// a large BA batch runs on it: creating and joining a std::thread per call was 40-70 us of a 3.5 ms call.)  One job at a time.
class OrbxHelperThread {
 public:
  OrbxHelperThread() : th_([this] { loop(); }) {}                          // std::system_error if the thread cannot be created
  ~OrbxHelperThread() {
    { std::lock_guard<std::mutex> g(m_); quit_ = true; }
    cv_.notify_all();
    th_.join();
  }
  void start(std::function<void()> f) {
    { std::lock_guard<std::mutex> g(m_); job_ = std::move(f); busy_ = true; }
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> g(m_);
    cv_.wait(g, [&] { return !busy_; });
  }

 private:
  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return quit_ || (busy_ && job_); });
        if (quit_) return;
        f = std::move(job_); job_ = nullptr;
      }
      try { f(); } catch (...) {}                                        // (the job reports its own errors; nothing may escape a thread)
      { std::lock_guard<std::mutex> g(m_); busy_ = false; }
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool busy_ = false, quit_ = false;
  std::thread th_;                                                        // last: starts when every other member exists
};

// ---- launch descriptors shared by host code and kernels ------------------------------------------
// Geometry of one pyramid level inside the per-image pyramid / blur slots (identical layout).
struct OrbLevelGeom {
  int w, h;            // level size
  int pitch;           // row pitch in bytes inside the slots (multiple of 64)
  int quota;           // n_l, features wanted on this level (Appendix A.3)
  float scale;         // scaleFactor^l as f32
  unsigned off;        // byte offset of the level inside one image's slot (multiple of 256)
  unsigned cand_off;   // element offset of the level's candidate region inside one image's slot
  unsigned cand_cap;   // worst-case number of NMS survivors of the level
  int btiles_x, btile_start;   // blur tiles (64x16 over the whole level)
  int ftiles_x, ftile_start;   // FAST tiles (64x16 over the border-filtered region [31,w-31)x[31,h-31))
  // describe tiles (describe_tile_kernel): the keypoint region [31, w-32] x [31, h-32] cut into dt_nx x dt_ny rectangles of dt_tw x dt_th
  // keypoint positions; tile of a keypoint = ((x - 31) / dt_tw, (y - 31) / dt_th), the divisions by __umulhi with dt_mx / dt_my
  int dt_nx, dt_ny, dt_tw, dt_th, dt_start;
  unsigned dt_mx, dt_my;
};
struct OrbGeom {
  int n_levels;
  int btiles_total, ftiles_total;
  int fast_threshold;
  unsigned slot_bytes;   // bytes of one image's pyramid (and blur) slot
  unsigned cand_total;   // candidate slots per image (sum of cand_cap)
  int dt_total;          // describe tiles per image (all levels); 0: the image size does not admit them (describe_fused_kernel then)
  OrbLevelGeom lv[ORBX_MAX_LEVELS];
};
// Where level images live: level 0 is the caller's image when it is 4-byte aligned with a pitch
// that is a multiple of 4, otherwise a copy in the pyramid slot; levels >= 1 are in the slot.
struct OrbSrc {
  const uint8_t* l0;
  size_t l0_img_stride;   // bytes between consecutive images at level 0
  int l0_pitch;
  int pad_;
  uint8_t* pyr;
  uint8_t* blur;
};

// device status word bits (sticky, cleared by orbx_check_status)
#define ORBX_ST_KP_OVERFLOW 1u   // more keypoints than cap_kp
#define ORBX_ST_INTERNAL 2u

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct KernelTimer {
  std::string name;
  std::vector<hipEvent_t> ev;  // start/stop pairs of the current call
  float ms = 0.f;
  int launches = 0;
};

struct orbx_handle {
  int device = -1;
  int n_cu = 256;                 // compute units of the device (persistent launches size their grids by it)
  hipStream_t stream = nullptr;
  orbx_camera cam{};
  orbx_orb_params orb{};
  int max_w = 0, max_h = 0, max_batch = 0;
  std::string err;
  unsigned* d_status = nullptr;
  unsigned* h_status = nullptr;   // pinned
  uint8_t* h_stage = nullptr;     // pinned mirror of the single-pair output block (orbx_process_stereo)
  size_t h_stage_bytes = 0;
  // hipGraph of the device part of orbx_process_stereo (launch-bound: ~20 short launches per frame); valid for
  // one (w, h, cap, buffer addresses) configuration, re-captured when any of them changes
  hipGraphExec_t pair_graph = nullptr;
  int pg_w = 0, pg_h = 0, pg_cap = 0, pg_calls = 0;
  void* pg_img = nullptr;
  void* pg_out = nullptr;
  // cached level geometry + resize tables for the last image size
  int geom_w = 0, geom_h = 0;
  OrbGeom geom{};
  DevBuf resize_tab;                     // per level l>=1: xtab[w_l], ytab[h_l] packed (ofs<<16 | c1)
  std::vector<unsigned> resize_tab_off;  // element offsets: [2*l] x table, [2*l+1] y table
  unsigned btile_tab_off = 0, ftile_tab_off = 0, dtile_tab_off = 0;   // tile -> (level, tx, ty) tables of the blur / FAST / describe launches, same buffer
  OrbSrc last_src{};                               // where the level images of the last extraction live, and how many images it held:
  int last_n_images = 0;                           // orbx_debug_read_level(which = 1) blurs them on demand (the product path keeps no blurred pyramid)
  // grow-only workspaces
  DevBuf ws_pyr, ws_blur, ws_cand, ws_counters, ws_sel, ws_sel2, ws_match, ws_io[12];
  DevBuf ws_dtile;                       // [image][describe tile] (begin, end) inside the level's spatially ordered keypoint list (rank_select_kernel)
  DevBuf ws_ba[28];
  // pipelined host-batch path: copy streams, events, double-buffered staging
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
  DevBuf ws_pipe[2][8];
  // second stream of the extractor: the blur runs beside the FAST -> Harris -> ordering chain (launch_orb_extract)
  hipStream_t s_aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_stag[4] = {nullptr, nullptr, nullptr, nullptr};   // two-stream form of orbx_process_stereo_batch_device: chunk c + 1 starts behind chunk c's pyramid
  // BA
  orbx_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  void* rccl_comm = nullptr;                               // ncclComm_t of the point-partitioned solve (orbx_ba_init_rccl / orbx_ba_set_rccl_comm)
  orbx_handle* ba_aux = nullptr;                           // second stream + workspaces of orbx_ba_solve_visual_batch (half of a large batch runs there)
  bool rccl_owned = false;
  void* h_ba_in = nullptr;   size_t h_ba_in_bytes = 0;    // pinned mirrors of the batch input / output blobs (ba_solve_batch)
  void* h_ba_out = nullptr;  size_t h_ba_out_bytes = 0;
  int* h_abort = nullptr;    int* d_abort = nullptr;       // pinned, device-visible: should_stop() seen while the iterations drain
  OrbxWorkPool* ba_pool = nullptr;                         // host workers of the batch preprocessing (created by the first large batch)
  OrbxHelperThread* ba_helper = nullptr;                   // drives the second half of a large batch (orbx_ba_solve_visual_batch)
  int ba_pool_cap = 0;                                     // > 0: at most this many threads for the next preprocessing (two halves share the cores)
  int ba_peer_windows = 0;                                 // windows of the other half of a batch, solved at the same time on the peer handle's stream (launch-shape heuristics count them)
  // the two halves of a batch share one PCIe link: the second half's uploads are ordered behind the first half's (its kernels then start
  // as early as they can, and the second half's bytes travel under them).  The first half records ba_up_event on its stream once its
  // uploads are enqueued and sets *ba_gate_signal; the second half waits for *ba_gate_wait, then makes its stream wait for ba_gate_event.
  hipEvent_t ba_up_event = nullptr;
  std::atomic<int>* ba_gate_signal = nullptr;
  std::atomic<int>* ba_gate_wait = nullptr;
  hipEvent_t ba_gate_event = nullptr;
  // profiling
  bool profiling = false;
  std::string prof_only;   // non-empty: only this kernel's launches are bracketed (orbx_set_profiling_only)
  std::vector<KernelTimer> timers;
  std::vector<hipEvent_t> event_pool;
  size_t event_next = 0;
  hipEvent_t prof_tail = nullptr;          // end event of the last profiling scope (may start the next one)
  hipStream_t prof_tail_stream = nullptr;
};

int orbx_fail(orbx_handle* h, int code, const char* fmt, ...);
int orbx_reserve(orbx_handle* h, DevBuf& b, size_t bytes);

#define ORBX_HIP(h, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return orbx_fail((h), ORBX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                       __FILE__, __LINE__);                                                 \
  } while (0)

// profiling scope: records a start/stop event pair around a launch when profiling is on
struct ProfScope {
  orbx_handle* h;
  int idx;
  hipStream_t stream;   // the stream the bracketed launches go to (default: the handle's)
  ProfScope(orbx_handle* h, const char* name, hipStream_t stream = nullptr, bool chained = false);
  ~ProfScope();
};
void orbx_prof_begin_call(orbx_handle* h);
void orbx_prof_end_call(orbx_handle* h);

// ---- kernel launchers (defined in the .hip files) --------------------------------------------------
// matcher (match_kernels.hip)
int launch_stereo_match(orbx_handle* h, int batch, const orbx_keypoint* d_kp, const uint8_t* d_desc,
                        const int* d_nkp, int cap_kp, orbx_dmatch* d_matches, int* d_nmatches,
                        double* d_points, uint8_t* d_has_point);
int launch_stereo_match_range(orbx_handle* h, hipStream_t st, int batch_total, int pair0, int batch, const orbx_keypoint* d_kp,
                              const uint8_t* d_desc, const int* d_nkp, int cap_kp, orbx_dmatch* d_matches, int* d_nmatches,
                              double* d_points, uint8_t* d_has_point);
int launch_crosscheck(orbx_handle* h, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                      orbx_dmatch* d_out, int* d_n_out);
int launch_hamming_batch(orbx_handle* h, const uint8_t* d_a, const uint8_t* d_b, int n, uint32_t* d_out);
int launch_guided_match(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n, double img_w, double img_h,
                        const double* d_q_uv, const uint8_t* d_q_desc, int nq, double radius, int mode, int* d_out_idx,
                        uint32_t* d_out_dist);
int launch_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const double* F9, const double* epipole,
                                    const orbx_keypoint* d_kp1, const uint8_t* d_desc1, const uint8_t* d_mp1,
                                    const uint8_t* d_stereo1, int n1, const orbx_keypoint* d_kp2, const uint8_t* d_desc2,
                                    const uint8_t* d_mp2, int n2, unsigned max_dist, int* d_pairs, int* d_n_out);
int launch_search_for_triangulation_bow(orbx_handle* h, const double* F9, const double* epipole, const orbx_keypoint* d_kp1,
                                        const uint8_t* d_desc1, const uint8_t* d_mp1, const uint8_t* d_stereo1, int n1,
                                        const orbx_keypoint* d_kp2, const uint8_t* d_desc2, const uint8_t* d_mp2, int n2,
                                        const int* d_sorted_idx, const int* d_rng_lo, const int* d_rng_hi, unsigned max_dist,
                                        int* d_pairs, int* d_n_out);
int launch_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* d_positions, const uint8_t* d_mp_desc, int P,
                       const double* d_kf_pose_cw, const int* d_kf_off, const orbx_keypoint* d_kps, const uint8_t* d_descs,
                       int T, double radius_scale, unsigned desc_threshold, int* d_out_idx, uint32_t* d_out_dist);
// extractor (orb_kernels.hip)
int orb_prepare_geometry(orbx_handle* h, int w, int h_px);
int launch_orb_extract(orbx_handle* h, const uint8_t* d_images, int n_images, int w, int h_px,
                       size_t stride, orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp, int cap_kp);
// The same over the images [img0, img0 + n) of a call that carries n_images in all, on stream `st`: orb_extract_prepare sizes the
// workspaces for the whole call and clears its counters (on the handle's stream), orb_extract_range then runs one range of it; `after_resize`
// (optional) is recorded on `st` behind the range's pyramid launches.
int orb_extract_prepare(orbx_handle* h, int n_images, int w, int h_px);
int orb_extract_range(orbx_handle* h, hipStream_t st, const uint8_t* d_images, int n_images, int img0, int n, int w, int h_px, size_t stride,
                      orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp, int cap_kp, hipEvent_t after_resize);
// BA (ba_kernels.hip)
int ba_solve_visual(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                    const double* poses_cw, int F, const double* fixed_poses_cw, int M,
                    double* points, int N, const orbx_ba_obs* obs, orbx_should_stop_fn should_stop,
                    void* user, double* poses_wc_out, int* iterations, double* initial_error,
                    double* final_error, bool global_mode = false, const struct BaInertialHost* inr = nullptr,
                    const orbx_ba_obs32* obs32 = nullptr);   // obs32: the 16-byte form of the observations, used instead of obs when given
// one window of a batch as the caller hands it over (orbx_ba_solve_visual_batch); status: ORBX_OK or ORBX_ERR_EMPTY
struct BaWinHost {
  int K, F, M, N;
  const double* poses_cw; const double* fixed_poses_cw; double* points; const orbx_ba_obs* obs;
  double* poses_wc_out; int* iterations; double* initial_error; double* final_error;
  int status;
  const orbx_ba_obs32* obs32 = nullptr;       // the 16-byte form of the observations (orbx.h): used instead of obs when given
};
int ba_solve_batch(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int W, BaWinHost* win,
                   orbx_should_stop_fn should_stop, void* user, bool global_mode = false, const struct BaInertialHost* inr = nullptr,
                   bool single_call = false);
// in-place sum of `n` doubles over the ranks of the handle's communicator, ordered on `st` (ncclAllReduce, orbx_api.hip)
int orbx_rccl_allreduce_sum(orbx_handle* h, double* d_buf, size_t n, hipStream_t st);
// destroys the handle's communicator if the library owns it, and clears it
void orbx_rccl_drop(orbx_handle* h);
int ba_debug_imu_residual(orbx_handle* h, int K, const double* poses_wc, const double* velocities, int E, const int* edge_kf,
                          const double* preint, double* out);
int ba_debug_blocks(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                    const double* fixed_poses_cw, int M, const double* points, int N, const orbx_ba_obs* obs, int global_mode,
                    double* out);
// the extra inputs / outputs of solve_inertial_ba (local_inertial_ba.rs:1074-1275) for ba_solve_visual's inertial mode
struct BaInertialHost {
  const orbx_inertial_ba_config* cfg;
  const double* velocities;   // [K][3]
  const double* biases;       // [K][6]
  int E;
  const int* edge_kf;         // [E][2]
  const double* preint;       // [E][11]
  double* vel_out;            // [K][3]
  double* bias_out;           // [K][6]
};
