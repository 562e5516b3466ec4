// ba_kernels.hip — visual local bundle adjustment on gfx950, all f64.
//
// Replaces solve_visual_ba (src/optimizer/local_ba_lm.rs:912-1098): Levenberg-Marquardt with the
// reference's damping / accept / stop rules (:1004-1056), the reference's residual (:192-212,
// :557-588), Huber weighting (:291-297) and g2o-style Jacobian blocks (:216-288).  The reference
// forms a dense 2N x P Jacobian, a dense J^T J and solves with LU (:1019-1039); the same normal
// equations are built here in their block structure (last bullet):
//
// Six launches per LM iteration (every block derives the keyframe rotations it needs itself):
//
//   ba_build_kernel      one 32- or 16-lane group per map point: residual + Jacobian blocks of its
//                        observations, V_j = sum B^T B, g_l, damped inverse V*_j^-1, V*_j^-1 g_l; per observation only six
//                        numbers are stored (oP [N][6] = x, y, 1/z, sqrt w, r0, r1: 48 B) — A, B and W = A^T B are rebuilt
//                        from them where they are needed, without a division or a square root
//   ba_kf_schur_kernel   keyframe partials (U_k = sum A^T A, g_p, b_red = sum A^T (B V*^-1 g_l), fixed summation order) and, in
//                        the same launch, S_red = Y^T W ((6K x 3M) x (3M x 6K)) on v_mfma_f64_16x16x4_f64: a block owns a
//                        (128-column block pair, k-split), fills 24-row operand tiles in LDS (W from oP, Y = W V*^-1) and
//                        accumulates its upper 16x16 tiles in registers — the one dense contraction of the path
//                        (in a batch: ba_kf_kernel and ba_schur_kernel, their own launches)
//   ba_gather_kernel     fixed-order reduction of the split-K and keyframe-split partials
//   ba_solve_lds_kernel  one workgroup: S = U* - S_red in LDS, blocked right-looking Cholesky, one phase per 16-column panel: wave 0
//                        factors the diagonal block (DPP row_newbcast) and publishes its columns, the other waves apply the previous
//                        panel's MFMA update and solve the rows below (right-hand side as one more row) a pivot step behind it;
//                        branch-free backward substitution over the zeroed upper triangle, delta_p
//                        (128 < n <= 176, inertial windows: ba_solve_tiled_kernel / ba_solve_inertial_tiled_kernel, lower tiles in LDS)
//                        (176 < n <= 320, S in global memory: ba_big_assemble_kernel + ba_big_factor_kernel — one workgroup, left-looking,
//                        one panel of look-ahead, the backward substitution in the same launch; beyond 320: one ba_big_step_kernel per panel
//                        + ba_big_back_kernel)
//   ba_backsub_kernel    delta_l = V*^-1 (-g_l - W^T delta_p), trial parameters and trial residuals
//   ba_decide_kernel     sums, accept / reject, lambda, stop tests — the LM state lives on the device
//
// plus ba_chi2_kernel (initial error), ba_sum3_kernel (partitioned runs) and the inertial kernels (ba_imu_*,
// ba_inertial_*) for solve_inertial_ba.
//
// Every reduction has a fixed order: results are run-to-run deterministic.  With an all-reduce hook
// (orbx_ba_set_allreduce) each rank holds a partition of the map points; [S_red, U, g_p, b_red,
// chi2, |g_l|^2] is summed over ranks before the solve and [chi2_trial, |delta_l|^2, |p_l|^2] after
// the back-substitution — two small latency-bound collectives per iteration.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <mutex>
#include <numeric>
#include <chrono>
#include <thread>

#include "orbx_internal.hpp"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

struct BaDims {
  int K, F, M, N;
  int P;        // 6K padded to a multiple of 16
  int rows;     // 3 * pps * ksplit: the k dimension of the Schur product, padded
  int ksplit;
  int ntile;    // P/16
  int pps;      // map points per k-split, a multiple of 16 (= 48 rows = one LDS operand tile of the producer / consumer body, two of the general body)
  int ncb;      // 128-column blocks of the reduced system: ceil(ntile / 8)
};

struct BaCam {
  double fx, fy, cx, cy, huber;
  int zero_behind;   // solve_global_ba: Jacobian rows are zero where z_c <= 0.001 (global_ba.rs:561-563)
  int inertial;      // solve_inertial_ba: T_wc pose parameters, the Jacobian forms of local_inertial_ba.rs:735-804,
                     // no Huber on the 100-px penalty, Huber threshold per observation (o_flag bit 0 = stereo)
  double huber_stereo;
  const int* o_flag;
};

// Levenberg-Marquardt state kept on the device (local_ba_lm.rs:1004-1056): the host only polls should_stop and
// enqueues; accept/reject, lambda, the stop tests and the iteration count are decided by ba_decide_kernel, and
// every kernel of an iteration returns at once when `done` is set.  sel picks which of the two parameter buffers
// is "current" (the other one receives the trial step).
struct BaState {
  double lambda, cur_sq, final_sq, gtol, ptol;
  int done, iters, sel;
  int bad;      // 0, or INT_MAX - (index of the first observation with an index out of range): ba_prep_count_kernel; such a window never runs
  // the fused step (ba_step_kernel, BaWin::dbl): which of the two sets of per-observation / per-point build results belongs to the
  // current parameters, and whether the current set's lambda-dependent point matrices are the ones of the last REJECTED step (BaWin::rej)
  int bsel, psel;
};
__device__ __forceinline__ double* ba_cur(const BaState* S, double* P0, double* P1) { return S->sel ? P1 : P0; }
__device__ __forceinline__ double* ba_trial(const BaState* S, double* P0, double* P1) { return S->sel ? P0 : P1; }

// One window of a batch (orbx_ba_solve_visual_batch; a single solve is a batch of one).  The array lives in device
// memory; every kernel of the LM loop takes it and picks its window by blockIdx.y, so that W independent windows share
// each launch: W workgroups factor W reduced systems at once instead of one CU working while 255 idle, and the
// keyframe / Schur / point kernels of all windows fill the chip together.  Nothing is shared between windows, and a
// window's arithmetic does not depend on W: every window of a batch equals its single-window result bit for bit.
struct BaWin {
  BaDims d;
  int n;                 // size of the reduced system the solve kernels see (6K; 15K for the inertial window)
  int use_lds;           // reduced system factored in LDS by ba_solve_lds_kernel (else the ba_big_* kernels)
  int part_sums;         // the Schur launch leaves one partial tile set per gather share (its k-splits already added in the gather's order) instead of one per k-split
  BaState* S;
  double *P0, *P1;       // the two parameter buffers [6K | 3M (| 9K)]
  const double* Rt_fix;  // [F+1][12]
  const int *pt_start, *o_kf;       // o_kf: the optimised keyframe of an observation, or -1 - f for the fixed observer f (f = F: the identity, :569)
  const double* o_uv;
  const int* kf_start;              // keyframe CSR over the point-major order: [K + 1] (host)
  int *kf_obs, *kf_pt;              // ... observation index and its map point, [kf_start[K]] each (ba_kflist_kernel, once per call)
  double *Vinv, *gl, *vg;           // per point: V*^-1 (9), g_l (3), V*^-1 g_l (3)
  double *pt_chi2, *pt_glsq, *pt_dsq, *pt_psq;
  double *oP;            // [N][6] per observation: x, y of the point in the camera frame, 1/z, sqrt(Huber weight), weighted residual (2) — what A, B, W are rebuilt from
  double *Rt_cur;        // [K][12] R|t of the optimised keyframes at the current parameters (written by block 0 of ba_build_kernel)
  int *slot_first;       // [M][K] first observation (point-major index) of point j in optimised keyframe k, or -1   } built on the device once
  int *obs_next;         // [N] next observation of the same (point, keyframe), or -1 (a point seen twice by one keyframe) } per call: ba_slots_kernel
  double *kfpart, *part, *rb;
  double *dp;            // step of the reduced system [n pad 16]
  double *Sg, *bvec, *ginv;   // global-memory factorisation (n > ~135 and the inertial system)
  double *res;           // [16] result block, see ba_decide_kernel
  // the observation CSR is built on the device once per call (ba_prep_*_kernel) from the caller's observations as they were handed over
  const orbx_ba_obs* obs_raw;   // [N], input order (obs32 == 0), or
  int obs32;             // ... the same array in the 16-byte form orbx_ba_obs32 (kf_idx < 0: fixed observer -1 - kf_idx; u, v as f32)
  int pad_;
  int *pt_fill;          // [M] per-point counters: observations of the point (count pass), then its fill position (place pass)
  int *obs_tmp;          // [N] observation indices grouped by map point, in arrival order within a point (sorted by ba_prep_order_kernel)
  int *o_flag;           // [N] inertial: orbx_ba_obs::_pad (bit 0 = stereo) in point-major order, else null
  double *Mz;            // [M][6] per point: the inverse of the Cholesky factor of V* (lower: m00, m10, m11, m20, m21, m22) — V*^-1 = M^T M, the one-operand Schur product's Z = W M^T
  // dbl: oP, Rt_cur, Mz, Vinv, gl, vg, pt_chi2, pt_glsq and Vraw exist twice, set 1 right behind set 0 (ba_set): the step kernel writes
  // the build results of the TRIAL parameters into the set that is not current while the current one is still being read
  double *Vraw;          // [M][6] per point: V before damping (dbl only: the rejected step's matrices are rebuilt from it)
  double *rej;           // [M][18] per point: Mz (6) | Vinv (9) | vg (3) of the CURRENT set for the lambda a rejection leaves (dbl only)
  int dbl, pad2_;
};

// The pointers a kernel takes out of its window descriptor are generic to the compiler (loaded from memory, not kernel arguments): every
// access through them was a FLAT load or store, which counts on lgkmcnt as well as vmcnt and may complete out of order with LDS traffic —
// each wait became a full drain of both counters, and no load stayed in flight across an LDS access.  (A cast to the global address
// space and back is folded away before the address-space inference runs, and llvm.assume(!is_shared & !is_private) is not picked up by
// this compiler; a pointer that is LOADED as a global pointer is.)  So a kernel reads its descriptor through a view of the same bytes
// whose pointer members are declared in the global address space — ba_win_global() — and works on the copy it returns.
#define BA_AS1 __attribute__((address_space(1)))
struct BaWinView {
  BaDims d;
  int n, use_lds, part_sums;
  BA_AS1 BaState* S;
  BA_AS1 double *P0, *P1;
  BA_AS1 const double* Rt_fix;
  BA_AS1 const int *pt_start, *o_kf;
  BA_AS1 const double* o_uv;
  BA_AS1 const int* kf_start;
  BA_AS1 int *kf_obs, *kf_pt;
  BA_AS1 double *Vinv, *gl, *vg;
  BA_AS1 double *pt_chi2, *pt_glsq, *pt_dsq, *pt_psq;
  BA_AS1 double *oP;
  BA_AS1 double *Rt_cur;
  BA_AS1 int *slot_first;
  BA_AS1 int *obs_next;
  BA_AS1 double *kfpart, *part, *rb;
  BA_AS1 double *dp;
  BA_AS1 double *Sg, *bvec, *ginv;
  BA_AS1 double *res;
  BA_AS1 const orbx_ba_obs* obs_raw;
  int obs32, pad_;
  BA_AS1 int *pt_fill;
  BA_AS1 int *obs_tmp;
  BA_AS1 int *o_flag;
  BA_AS1 double *Mz;
  BA_AS1 double *Vraw;
  BA_AS1 double *rej;
  int dbl, pad2_;
};
static_assert(sizeof(BaWinView) == sizeof(BaWin) && offsetof(BaWinView, S) == offsetof(BaWin, S) && offsetof(BaWinView, oP) == offsetof(BaWin, oP) &&
              offsetof(BaWinView, res) == offsetof(BaWin, res) && offsetof(BaWinView, o_flag) == offsetof(BaWin, o_flag) && offsetof(BaWinView, Mz) == offsetof(BaWin, Mz) &&
              offsetof(BaWinView, rej) == offsetof(BaWin, rej) && offsetof(BaWinView, dbl) == offsetof(BaWin, dbl),
              "BaWinView is BaWin with its pointers in the global address space");
__device__ __forceinline__ BaWin ba_win_global(const BaWin* __restrict__ wins, int i) {
  const BaWinView& v = ((const BaWinView*)wins)[i];
  BaWin g;
  g.d = v.d; g.n = v.n; g.use_lds = v.use_lds; g.part_sums = v.part_sums;
  g.S = (BaState*)v.S; g.P0 = (double*)v.P0; g.P1 = (double*)v.P1; g.Rt_fix = (const double*)v.Rt_fix;
  g.pt_start = (const int*)v.pt_start; g.o_kf = (const int*)v.o_kf; g.o_uv = (const double*)v.o_uv; g.kf_start = (const int*)v.kf_start;
  g.kf_obs = (int*)v.kf_obs; g.kf_pt = (int*)v.kf_pt; g.Vinv = (double*)v.Vinv; g.gl = (double*)v.gl; g.vg = (double*)v.vg;
  g.pt_chi2 = (double*)v.pt_chi2; g.pt_glsq = (double*)v.pt_glsq; g.pt_dsq = (double*)v.pt_dsq; g.pt_psq = (double*)v.pt_psq;
  g.oP = (double*)v.oP; g.Rt_cur = (double*)v.Rt_cur; g.slot_first = (int*)v.slot_first; g.obs_next = (int*)v.obs_next;
  g.kfpart = (double*)v.kfpart; g.part = (double*)v.part; g.rb = (double*)v.rb; g.dp = (double*)v.dp;
  g.Sg = (double*)v.Sg; g.bvec = (double*)v.bvec; g.ginv = (double*)v.ginv; g.res = (double*)v.res;
  g.obs_raw = (const orbx_ba_obs*)v.obs_raw; g.obs32 = v.obs32; g.pad_ = 0; g.pt_fill = (int*)v.pt_fill; g.obs_tmp = (int*)v.obs_tmp; g.o_flag = (int*)v.o_flag;
  g.Mz = (double*)v.Mz; g.Vraw = (double*)v.Vraw; g.rej = (double*)v.rej; g.dbl = v.dbl; g.pad2_ = 0;
  return g;
}

// ---- small device math ---------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double w = q[0], i = q[1], j = q[2], k = q[3];
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  const double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0;
  const double ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
  R[0] = ww + ii - jj - kk; R[1] = ij - wk;           R[2] = wj + ik;
  R[3] = wk + ij;           R[4] = ww - ii + jj - kk; R[5] = jk - wi;
  R[6] = ik - wj;           R[7] = wi + jk;           R[8] = ww - ii - jj + kk;
}

// local_ba_lm.rs:648-662 then R|t (12 doubles)
// nalgebra UnitQuaternion::from_scaled_axis = exp of the pure quaternion r/2 (identity when |r/2|^2 <= eps^2)
__device__ __forceinline__ void dev_q_from_scaled_axis(const double* r, double* q) {
  const double v0 = r[0] / 2.0, v1 = r[1] / 2.0, v2 = r[2] / 2.0;
  const double nn = v0 * v0 + v1 * v1 + v2 * v2;
  const double eps = 2.220446049250313e-16;
  if (nn <= eps * eps) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; return; }
  const double n = sqrt(nn), s = 1.0 * sin(n) / n;
  q[0] = 1.0 * cos(n); q[1] = v0 * s; q[2] = v1 * s; q[3] = v2 * s;
}
__device__ __forceinline__ void dev_q_rot(const double* q, const double* v, double* o) {   // UnitQuaternion * Vector3
  const double t0 = 2.0 * (q[2] * v[2] - q[3] * v[1]), t1 = 2.0 * (q[3] * v[0] - q[1] * v[2]), t2 = 2.0 * (q[1] * v[1] - q[2] * v[0]);
  const double c0 = q[2] * t2 - q[3] * t1, c1 = q[3] * t0 - q[1] * t2, c2 = q[1] * t1 - q[2] * t0;
  o[0] = t0 * q[0] + c0 + v[0]; o[1] = t1 * q[0] + c1 + v[1]; o[2] = t2 * q[0] + c2 + v[2];
}
__device__ __forceinline__ void dev_q_mul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}

// pose parameters -> R|t of T_cw (12 doubles).  Visual solvers: se3_from_params (local_ba_lm.rs:648-662) of T_cw parameters;
// inertial: extract_pose(..).inverse() of T_wc parameters (local_inertial_ba.rs:584-591, :634; se3.rs:56-63)
__device__ __forceinline__ void pose_to_Rt(const double* p, int inertial, double* o) {
  if (inertial) {
    double qwc[4], rt[3];
    dev_q_from_scaled_axis(p, qwc);
    const double qcw[4] = {qwc[0], -qwc[1], -qwc[2], -qwc[3]};
    dev_q_rot(qcw, p + 3, rt);
    quat_to_R(qcw, o);
    o[9] = -rt[0]; o[10] = -rt[1]; o[11] = -rt[2];
    return;
  }
  double q[4];
  const double angle = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  if (angle > 1e-10) {
    double a0 = p[0] / angle, a1 = p[1] / angle, a2 = p[2] / angle;
    const double n = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    a0 /= n; a1 /= n; a2 /= n;
    const double s = sin(angle / 2.0), c = cos(angle / 2.0);
    q[0] = c; q[1] = a0 * s; q[2] = a1 * s; q[3] = a2 * s;
  } else {
    q[0] = 1; q[1] = q[2] = q[3] = 0;
  }
  quat_to_R(q, o);
  o[9] = p[3]; o[10] = p[4]; o[11] = p[5];
}
constexpr int BA_MAX_K = 128;   // keyframes whose R|t a block keeps in LDS (= BA_MAX_N / 6)

// Every block of the build / chi2 kernels derives the K rotations it needs itself (K <= 128 sin/cos pairs, a few
// hundred instructions) instead of reading them from a separate pose kernel: one launch less on a loop whose
// iterations are a chain of ~6 us launches.
__device__ __forceinline__ void block_poses(const double* params, int K, int inertial, double* sRt) {
  for (int k = threadIdx.x; k < K; k += blockDim.x) pose_to_Rt(params + 6 * (size_t)k, inertial, sRt + 12 * k);
  __syncthreads();
}

// iteration counter only (windows without points never launch the build kernel)
__global__ void ba_iter_kernel(const BaWin* __restrict__ wins, int iter) {
  BaState* S = ba_win_global(wins, blockIdx.y).S;
  if (S->done) return;
  S->iters = iter + 1;                                                  // local_ba_lm.rs:1017
}

// ---- the observation CSR, built on the device once per call -------------------------------------------------------------------
// Until round 4 the host made two passes over every window's observations per call (index checks + per-point counts, then a stable
// scatter into point-major order): 0.07 ms per 32 k-observation window on one core, and in a 32-window batch — spread over a pool of
// workers that each also enqueued their window's upload — 1.4-2.8 ms per half in which the GPU stood still (ORBX_BA_TIMING, round 3).
// Now the caller's observations go up as they are (straight out of the caller's memory when that is pinned) and four short launches
// per call, shared by all windows, build what the solver reads: pt_start / kf_start, and o_kf / o_uv (/ o_flag) in point-major order with
// the observations of a point in INPUT order — the order the host's stable scatter produced, so every sum downstream keeps its order
// and its bits.  The order is made deterministic without a stable sort: the place pass groups observation INDICES by point in whatever
// order its atomics land, and the order pass sorts each point's (short) index list.
//   count: index checks (local_ba_lm.rs has none: an index out of range is a caller's bug — the window is flagged and never runs, the
//          call fails with ORBX_ERR_INVALID as before), observations per point and per optimised keyframe
//   scan:  pt_start, kf_start
//   place: observation index -> its point's segment of obs_tmp
//   order: per point, sort the segment; gather the observations into o_kf / o_uv / o_flag
constexpr int BA_PREP_OPB = 1024;   // observations per block of the count / place passes (256 threads x 4)
// (kf_idx, fixed_idx, mp_idx, _pad) of observation i in either wire format; the 16-byte form carries the fixed observer in kf_idx
__device__ __forceinline__ int4 ba_obs_ids(const BaWin& win, int i) {
  if (win.obs32) {
    const int2 q = *reinterpret_cast<const int2*>(reinterpret_cast<const orbx_ba_obs32*>(win.obs_raw) + i);
    const int f = -1 - q.x;                                                // fixed observer f; f == F: the identity pose (= fixed_idx -1 of orbx_ba_obs)
    return make_int4(q.x >= 0 ? q.x : -1, (q.x >= 0 || f == win.d.F) ? -1 : f, q.y, 0);
  }
  return *reinterpret_cast<const int4*>(&win.obs_raw[i]);
}
__global__ __launch_bounds__(256) void ba_prep_count_kernel(const BaWin* __restrict__ wins) {
  __shared__ int s_kf[BA_MAX_K];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int K = win.d.K, F = win.d.F, M = win.d.M, N = win.d.N;
  const int i0 = blockIdx.x * BA_PREP_OPB;
  if (win.S->done || i0 >= N) return;
  const int tid = threadIdx.x;
  for (int k = tid; k < K; k += 256) s_kf[k] = 0;
  __syncthreads();
  int* __restrict__ pt_fill = win.pt_fill;
#pragma unroll
  for (int r = 0; r < BA_PREP_OPB / 256; ++r) {
    const int i = i0 + r * 256 + tid;
    if (i >= N) break;
    const int4 q = ba_obs_ids(win, i);                                    // kf_idx, fixed_idx, mp_idx, _pad
    if (q.z < 0 || q.z >= M || q.x >= K || (q.x < 0 && q.y >= F)) { atomicMax(&win.S->bad, 0x7fffffff - i); continue; }
    atomicAdd(&pt_fill[q.z], 1);
    if (q.x >= 0) atomicAdd(&s_kf[q.x], 1);
  }
  __syncthreads();
  int* kf_cnt = const_cast<int*>(win.kf_start) + 1;                      // kf_start[k + 1] = count of keyframe k until the scan
  for (int k = tid; k < K; k += 256) if (s_kf[k]) atomicAdd(&kf_cnt[k], s_kf[k]);
}

// one block per window: exclusive scans.  pt_fill is left zeroed for the place pass.
__global__ __launch_bounds__(1024) void ba_prep_scan_kernel(const BaWin* __restrict__ wins) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  const BaWin win = ba_win_global(wins, blockIdx.y);
  BaState* S = win.S;
  if (S->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (S->bad != 0) {                                                      // (uniform: written by the previous launch)
    __syncthreads();
    if (tid == 0) S->done = 1;
    return;
  }
  const int K = win.d.K, M = win.d.M;
  int* __restrict__ pt_start = const_cast<int*>(win.pt_start);
  int* __restrict__ pt_fill = win.pt_fill;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int j0 = 0; j0 < M; j0 += 1024) {
    const int j = j0 + tid;
    const int c = j < M ? pt_fill[j] : 0;
    if (j < M) pt_fill[j] = 0;
    int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = s_carry;
    for (int w2 = 0; w2 < wave; ++w2) base += s_wave[w2];
    if (j < M) pt_start[j] = base + inc - c;
    __syncthreads();
    if (tid == 1023) s_carry = base + inc;
    __syncthreads();
  }
  if (tid == 0) pt_start[M] = s_carry;
  if (wave == 0) {
    // kf_start[k + 1] = observations of the keyframes 0 .. k: a wave scan, two keyframes per lane (K <= 128) — one thread's loop of K
    // dependent read-modify-writes was most of this launch at 49 keyframes (16 us)
    int* kf_start = const_cast<int*>(win.kf_start);                      // (kf_start[0] = 0: zeroed with the counters)
    static_assert(BA_MAX_K <= 128, "two keyframes per lane of one wave");
    const int k0 = 2 * lane, c0 = k0 < K ? kf_start[k0 + 1] : 0, c1 = k0 + 1 < K ? kf_start[k0 + 2] : 0;
    int inc = c0 + c1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    const int before = inc - c0 - c1;
    if (k0 < K) kf_start[k0 + 1] = before + c0;
    if (k0 + 1 < K) kf_start[k0 + 2] = before + c0 + c1;
  }
}

__global__ __launch_bounds__(256) void ba_prep_place_kernel(const BaWin* __restrict__ wins) {
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int N = win.d.N;
  const int i0 = blockIdx.x * BA_PREP_OPB;
  if (win.S->done || i0 >= N) return;
  const int* __restrict__ pt_start = win.pt_start;
#pragma unroll
  for (int r = 0; r < BA_PREP_OPB / 256; ++r) {
    const int i = i0 + r * 256 + threadIdx.x;
    if (i >= N) break;
    const int mp = win.obs32 ? reinterpret_cast<const orbx_ba_obs32*>(win.obs_raw)[i].mp_idx : win.obs_raw[i].mp_idx;
    win.obs_tmp[pt_start[mp] + atomicAdd(&win.pt_fill[mp], 1)] = i;
  }
}

// 16 lanes per map point: the point's observation indices into ascending (= input) order by counting, for each, the smaller ones
// (a track is ~16 observations; a long one costs its square over 16 lanes), then the observations themselves into that order.
__global__ __launch_bounds__(256) void ba_prep_order_kernel(const BaWin* __restrict__ wins) {
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int F = win.d.F, M = win.d.M;
  const int j = (blockIdx.x * 256 + threadIdx.x) >> 4, li = threadIdx.x & 15;
  if (win.S->done || j >= M) return;
  const int s = win.pt_start[j], n = win.pt_start[j + 1] - s;
  const int* __restrict__ tmp = win.obs_tmp + s;
  const orbx_ba_obs* __restrict__ obs = win.obs_raw;
  int* __restrict__ o_kf = const_cast<int*>(win.o_kf);
  double* __restrict__ o_uv = const_cast<double*>(win.o_uv);
  int* __restrict__ o_flag = win.o_flag;
  for (int e = li; e < n; e += 16) {
    const int idx = tmp[e];
    int rank = 0;
    for (int k = 0; k < n; ++k) rank += tmp[k] < idx ? 1 : 0;
    const int4 q = ba_obs_ids(win, idx);
    double2_t uv;
    if (win.obs32) {                                                       // f32 -> f64: exact, what the host's `as f64` (local_ba_lm.rs:870-872) does
      const float2 f = *reinterpret_cast<const float2*>(&reinterpret_cast<const orbx_ba_obs32*>(win.obs_raw)[idx].u);
      uv[0] = (double)f.x; uv[1] = (double)f.y;
    } else uv = *reinterpret_cast<const double2_t*>(&obs[idx].u);
    const size_t t = (size_t)(s + rank);
    o_kf[t] = q.x >= 0 ? q.x : -1 - (q.y >= 0 ? q.y : F);                 // a fixed observer f as -1 - f; slot F = identity (:569)
    *reinterpret_cast<double2_t*>(&o_uv[2 * t]) = uv;
    if (o_flag) o_flag[t] = q.w;
  }
}

// (point, keyframe) -> its observation(s): what fills the operand tiles of the Schur product.  One thread per map point walks the
// point's observations in order (they are contiguous, point-major): slot (j, k) = the first observation of point j by optimised
// keyframe k; a point seen twice by one keyframe (two features of it carry the same map point) has its observations chained in
// order through obs_next — the tile slot then holds the sum of their W blocks, as J^T J does.  Built here once per call instead of
// on the host: 4 (M K + N) bytes per window less to prepare and to upload (a fifth of the input blob at 20 keyframes / 2000 points).
// Round 5: sixteen lanes per point instead of one thread walking its ~16-50 observations through dependent global round trips (2000
// points = 8 workgroups, 15 us per call; 8000 points 38 us).  The lanes take the point's observations in turn and note the smallest
// observation index per keyframe in LDS (atomicMin); a keyframe that comes up twice — rare: two features of one keyframe on one map
// point — sends the whole point down the serial walk below, by one lane, exactly as before.  Same slot map, same chains.
constexpr int BA_SLOT_LANES = 16, BA_SLOT_PPB = 256 / BA_SLOT_LANES;
__global__ __launch_bounds__(256) void ba_slots_kernel(const BaWin* __restrict__ wins) {
  __shared__ int s_first[BA_SLOT_PPB][BA_MAX_K];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int K = win.d.K, M = win.d.M;
  const int grp = threadIdx.x / BA_SLOT_LANES, li = threadIdx.x % BA_SLOT_LANES;
  const int j = blockIdx.x * BA_SLOT_PPB + grp;
  if (win.S->done || K <= 0 || K > BA_MAX_K) return;  // (a window the reference answers None for was never prepared: its index arrays hold nothing; K is checked by the host)
  const bool live = j < M;
  int* __restrict__ slot = win.slot_first + (size_t)(live ? j : 0) * K;
  int* __restrict__ nxt = win.obs_next;
  const int* __restrict__ o_kf = win.o_kf;
  int* first = s_first[grp];
  const int t0 = live ? win.pt_start[j] : 0, t1 = live ? win.pt_start[j + 1] : 0;
  for (int k = li; k < K; k += BA_SLOT_LANES) first[k] = 0x7fffffff;
  // (a group is part of one wave: its LDS operations are performed in issue order, no barrier between these steps)
  bool dup = false;
  for (int t = t0 + li; t < t1; t += BA_SLOT_LANES) {
    const int k = o_kf[t];
    if (k >= 0 && atomicMin(&first[k], t) != 0x7fffffff) dup = true;     // somebody noted this keyframe before: a second observation of (point, keyframe)
  }
  const unsigned long long bal = __ballot(dup);
  const bool any_dup = ((bal >> ((threadIdx.x & 63) & ~(BA_SLOT_LANES - 1))) & ((1ull << BA_SLOT_LANES) - 1ull)) != 0ull;
  if (!live) return;
  if (!any_dup) {
    for (int t = t0 + li; t < t1; t += BA_SLOT_LANES) nxt[t] = -1;
    for (int k = li; k < K; k += BA_SLOT_LANES) { const int f = first[k]; slot[k] = f == 0x7fffffff ? -1 : f; }
    return;
  }
  if (li != 0) return;
  for (int k = 0; k < K; ++k) slot[k] = -1;
  for (int t = t0; t < t1; ++t) {
    nxt[t] = -1;
    const int k = o_kf[t];
    if (k < 0) continue;
    int i = slot[k];
    if (i < 0) { slot[k] = t; continue; }
    while (nxt[i] >= 0) i = nxt[i];                                      // (chains hold two, rarely three observations)
    nxt[i] = t;
  }
}

// The keyframe lists (for each optimised keyframe the observations it makes, in point-major order, and their map points) from the slot
// map: block k walks the points in chunks of 256, a thread counts its point's chain for keyframe k (0, 1, rarely more), a block-wide
// exclusive scan places them behind the keyframe's host-computed start.  Until round 3's end the host built these 8 bytes per observation
// and uploaded them with every call.  Same order as the host's pass produced (ascending observation index within a keyframe).
// (Round 5: 1024 threads — the block walks the points in chunks of its size with three barriers per chunk; with the sixteen-lane slot kernel
// the two launches together 27.5 -> 11.8 us per call at 20 keyframes / 2000 points, 82 -> 29.4 us at 50 / 8000: profiles/r05_ba_setup_kernels_ab.txt)
constexpr int BA_KFL_THREADS = 1024;
__global__ __launch_bounds__(BA_KFL_THREADS) void ba_kflist_kernel(const BaWin* __restrict__ wins) {
  __shared__ int s_wave[BA_KFL_THREADS / 64];
  __shared__ int s_base;
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int K = win.d.K, M = win.d.M, k = blockIdx.x;
  if (win.S->done || k >= K) return;
  const int* __restrict__ slot_first = win.slot_first; const int* __restrict__ nxt = win.obs_next;
  int* __restrict__ kf_obs = win.kf_obs; int* __restrict__ kf_pt = win.kf_pt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = win.kf_start[k];
  __syncthreads();
  for (int j0 = 0; j0 < M; j0 += BA_KFL_THREADS) {
    const int j = j0 + tid;
    const int first = j < M ? slot_first[(size_t)j * K + k] : -1;
    int cnt = 0;
    for (int i = first; i >= 0; i = nxt[i]) ++cnt;
    int inc = cnt;                                                          // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int pos = s_base + inc - cnt;
    for (int w2 = 0; w2 < wave; ++w2) pos += s_wave[w2];
    for (int i = first; i >= 0; i = nxt[i]) { kf_obs[pos] = i; kf_pt[pos] = j; ++pos; }
    __syncthreads();
    if (tid == BA_KFL_THREADS - 1) s_base = pos;                            // (the last thread's end = the chunk's end)
    __syncthreads();
  }
}

struct ObsOut { double r0, r1, A[12], B[6]; double px, py, piz, psw; double chi; };

// The 2x6 pose block A and the 2x3 point block B of one observation (both * sqrt(w)) from its four "projective" numbers —
// x, y of the point in the camera frame, 1/z, and the square root of the Huber weight — and the keyframe's R|t: no division, no
// square root.  piz == 0 marks an observation whose Jacobian rows are zero (|z| < 1e-6, or behind the camera in the global /
// inertial forms).  obs_terms itself goes through this function, so every kernel that rebuilds the blocks from the stored
// (x, y, 1/z, sqrt w) gets the same bits the build kernel used for V and g_l.
// J_pose (:239-254), J_point (:281-287); inertial: local_inertial_ba.rs:735-804.
// Round 4: the entries as products of six shared factors (f_x sqrt w, f_y sqrt w, those times 1/z, x/z, y/z) instead of each entry's own
// chain of four or five multiplications as the reference writes them — 28 double-precision operations where there were 70, in every
// per-observation kernel (these kernels issue f64 operations and little else).  The values agree with the literal form to rounding.
// The inertial form's point block is the same expression; its pose block is the visual one with the opposite sign (the perturbation
// is applied on the other side), so one body serves both.  A[4] and A[9] are structurally zero: the callers skip their products.
__device__ __forceinline__ void obs_jac_from_proj(const BaCam& cam, const double* Rt, double x, double y, double piz, double sw,
                                                  double* __restrict__ A, double* __restrict__ B) {
  // (piz == 0 comes with sw == 0 — obs_terms sets both or neither, an empty Schur slot is all zeros — and then every entry below is
  // 0 * finite: no test, no branch, so that two observations' blocks can be scheduled into each other)
  const double fs = cam.fx * sw, gs = cam.fy * sw, fz = fs * piz, gz = gs * piz, xz = x * piz, yz = y * piz;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    B[c] = fz * fma(xz, Rt[6 + c], -Rt[c]);                // -(1/z) (f_x R_0c - f_x (x/z) R_2c) sqrt w
    B[3 + c] = gz * fma(yz, Rt[6 + c], -Rt[3 + c]);
  }
  const double sg = cam.inertial ? -1.0 : 1.0;
  const double fa = sg * fs, ga = sg * gs, fza = sg * fz, gza = sg * gz;
  A[2] = yz * fa;                    A[0] = xz * A[2];                  A[1] = -(fma(xz, xz, 1.0) * fa);
  A[3] = -fza;                       A[4] = 0.0;                        A[5] = xz * fza;
  A[8] = -(xz * ga);                 A[7] = yz * A[8];                  A[6] = fma(yz, yz, 1.0) * ga;
  A[9] = 0.0;                        A[10] = -gza;                      A[11] = yz * gza;
}
// (the pose block's structural zeros: column 4 of the u row, column 3 of the v row)
__device__ __forceinline__ constexpr bool ba_a0_zero(int a) { return a == 4; }
__device__ __forceinline__ constexpr bool ba_a1_zero(int a) { return a == 3; }
// A[a] p + A[6 + a] q without the products of the structural zeros
__device__ __forceinline__ double ba_arow(const double* A, int a, double p, double q) {
  if (ba_a0_zero(a)) return A[6 + a] * q;
  if (ba_a1_zero(a)) return A[a] * p;
  return fma(A[a], p, A[6 + a] * q);
}
// acc + A[a] p + A[6 + a] q
__device__ __forceinline__ double ba_arow_acc(double acc, const double* A, int a, double p, double q) {
  if (ba_a0_zero(a)) return fma(A[6 + a], q, acc);
  if (ba_a1_zero(a)) return fma(A[a], p, acc);
  return fma(A[a], p, fma(A[6 + a], q, acc));
}

// The one-operand form of the Schur product (round 4).  S_red = sum_j W_j V_j*^-1 W_j^T, and with V* = L L^T, M = L^-1 (lower triangular):
// V*^-1 = M^T M, so W V*^-1 W^T = Z Z^T with Z = W M^T — ONE operand tile instead of Y = W V*^-1 and W, and Z = A^T (B M^T) never forms W:
// per slot the 2 x 3 block B M^T (12 operations) and A^T of it (32) where Y and W took 54 + 90.  z[c * 6 + a], M = (m00, m10, m11, m20, m21, m22).
template <bool ACC>
__device__ __forceinline__ void obs_z_from_stored(const BaCam& cam, const double* Rt, const double* __restrict__ q, const double* __restrict__ M, double* __restrict__ z) {
  double A[12], B[6];
  obs_jac_from_proj(cam, Rt, q[0], q[1], q[2], q[3], A, B);
  double t[6];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const double* b = B + 3 * r;
    t[3 * r] = b[0] * M[0]; t[3 * r + 1] = fma(b[0], M[1], b[1] * M[2]); t[3 * r + 2] = fma(b[0], M[3], fma(b[1], M[4], b[2] * M[5]));
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int c = 0; c < 3; ++c) z[6 * c + a] = ACC ? ba_arow_acc(z[6 * c + a], A, a, t[c], t[3 + c]) : ba_arow(A, a, t[c], t[3 + c]);
  }
}

// error (:192-212), Huber (:291-297), and with want_jac the Jacobian blocks and the four numbers they follow from.
// o.chi = w |e|^2 = |e|^2 inside the Huber threshold, threshold * |e| beyond it — every sum of squared residuals (the build kernel's and
// the trial's) takes this one expression, which needs neither the weight's division nor its square root; r0, r1 (= sqrt w e, for g_l)
// only exist with want_jac.  The projection divides once (1/z) and multiplies twice where the reference writes x / z, y / z: a
// double-precision division is ~30 issue slots of a kernel that does little else.
__device__ __forceinline__ void obs_terms(const BaCam& cam, const double* Rt, const double* X, double u, double v,
                                          bool want_jac, ObsOut& o, int flag = 0) {
  // X_c = R X + t.  (The reference rotates with the quaternion form v + w t + q x t; R X is the same
  // rotation — agreement is to rounding, well inside the stated tolerance.)
  const double x = fma(Rt[0], X[0], fma(Rt[1], X[1], fma(Rt[2], X[2], Rt[9])));
  const double y = fma(Rt[3], X[0], fma(Rt[4], X[1], fma(Rt[5], X[2], Rt[10])));
  const double z = fma(Rt[6], X[0], fma(Rt[7], X[1], fma(Rt[8], X[2], Rt[11])));
  o.px = x; o.py = y; o.piz = 0.0; o.psw = 0.0;
  // local_inertial_ba.rs:633-659 (residual), :735-804 (Jacobian rows): behind the camera -> (100, 100), no Jacobian; the visual form
  // (:192-212) takes the same residual there and keeps the Jacobian unless |z| < 1e-6 (or zero_behind: the global form)
  const bool behind = cam.inertial ? !(z > 0.001) : z <= 0.001;
  const double thr = cam.inertial && (flag & 1) ? cam.huber_stereo : cam.huber;
  if (cam.inertial && behind) {                       // (its (100, 100) is returned before the Huber weight: :640-642)
    o.r0 = 100.0; o.r1 = 100.0; o.chi = 20000.0;
    if (want_jac) obs_jac_from_proj(cam, Rt, x, y, 0.0, 0.0, o.A, o.B);
    return;
  }
  const double zi = 1.0 / z;
  double e0 = 100.0, e1 = 100.0;
  if (!behind) { e0 = u - fma(cam.fx * x, zi, cam.cx); e1 = v - fma(cam.fy * y, zi, cam.cy); }
  const double s2 = fma(e0, e0, e1 * e1);
  double sw = 1.0;
  if (s2 <= thr * thr) o.chi = s2;
  else {
    const double en = sqrt(s2);
    o.chi = thr * en;
    if (want_jac) sw = sqrt(thr / en);
  }
  if (!want_jac) return;
  o.r0 = e0 * sw; o.r1 = e1 * sw;
  const bool no_jac = !cam.inertial && (fabs(z) < 1e-6 || (cam.zero_behind && behind));
  if (!no_jac) { o.piz = zi; o.psw = sw; }
  obs_jac_from_proj(cam, Rt, x, y, o.piz, o.psw, o.A, o.B);
}

// The arrays that exist once per SET of build results (BaWin::dbl): set s of a window.  Without dbl there is one set and s is 0.
struct BaSet { double *oP, *Rt_cur, *Mz, *Vinv, *gl, *vg, *pt_chi2, *pt_glsq, *Vraw; };
__device__ __forceinline__ BaSet ba_set(const BaWin& w, int s) {
  const size_t n1 = (size_t)max(w.d.N, 1), m1 = (size_t)max(w.d.M, 1), k1 = (size_t)max(w.d.K, 1), q = w.dbl ? (size_t)s : 0;
  BaSet b;
  b.oP = w.oP + q * 6 * n1; b.Rt_cur = w.Rt_cur + q * 12 * k1; b.Mz = w.Mz + q * 6 * m1; b.Vinv = w.Vinv + q * 9 * m1;
  b.gl = w.gl + q * 3 * m1; b.vg = w.vg + q * 3 * m1; b.pt_chi2 = w.pt_chi2 + q * m1; b.pt_glsq = w.pt_glsq + q * m1;
  b.Vraw = w.Vraw ? w.Vraw + q * 6 * m1 : nullptr;
  return b;
}
// The lambda-dependent point matrices the consumers read: the current set's own, or — after a rejected step — the ones the step kernel
// left for the lambda of the rejection (BaWin::rej: Mz | Vinv | vg)
struct BaPointMats { const double *Mz, *Vinv, *vg; };
__device__ __forceinline__ BaPointMats ba_point_mats(const BaWin& w, const BaState* S, const BaSet& cur) {
  BaPointMats m;
  if (w.dbl && S->psel) { const size_t m1 = (size_t)max(w.d.M, 1); m.Mz = w.rej; m.Vinv = w.rej + 6 * m1; m.vg = w.rej + 15 * m1; }
  else { m.Mz = cur.Mz; m.Vinv = cur.Vinv; m.vg = cur.vg; }
  return m;
}

// One observation of the build pass: residual + Jacobian at (Rt, X), its six numbers stored, chi2 / V / g_l accumulated
__device__ __forceinline__ void ba_build_obs(const BaCam& cam, const double* Rt, const double (&X)[3], double u, double v, int flag,
                                             double* __restrict__ q, double (&V)[6], double (&g)[3], double& chi) {
  ObsOut o;
  obs_terms(cam, Rt, X, u, v, true, o, flag);
  q[0] = o.px; q[1] = o.py; q[2] = o.piz; q[3] = o.psw; q[4] = o.r0; q[5] = o.r1;
  chi += o.chi;
  V[0] = fma(o.B[0], o.B[0], fma(o.B[3], o.B[3], V[0]));
  V[1] = fma(o.B[0], o.B[1], fma(o.B[3], o.B[4], V[1]));
  V[2] = fma(o.B[0], o.B[2], fma(o.B[3], o.B[5], V[2]));
  V[3] = fma(o.B[1], o.B[1], fma(o.B[4], o.B[4], V[3]));
  V[4] = fma(o.B[1], o.B[2], fma(o.B[4], o.B[5], V[4]));
  V[5] = fma(o.B[2], o.B[2], fma(o.B[5], o.B[5], V[5]));
  g[0] = fma(o.B[0], o.r0, fma(o.B[3], o.r1, g[0]));
  g[1] = fma(o.B[1], o.r0, fma(o.B[4], o.r1, g[1]));
  g[2] = fma(o.B[2], o.r0, fma(o.B[5], o.r1, g[2]));
}
// damped V* = V + lambda*max(diag,1e-6) (:1031-1034); V* = L L^T (V* is positive definite: V is a sum of B^T B and every diagonal entry is
// damped), M = L^-1 (m00, m10, m11, m20, m21, m22), I = V*^-1 = M^T M, vg = V*^-1 g.  The three pivots' reciprocal square roots are the only
// transcendental steps.  (First form: three square roots and five divisions for M beside the closed-form inverse's one division: 47.2 us
// per 32-window launch against 42.8, profiles/r04_ba_mz_rsqrt_ab.txt.)
__device__ __forceinline__ void ba_point_matrices(const double (&V)[6], const double (&g)[3], double lambda, double (&M)[6], double (&I)[9], double (&vgo)[3]) {
  const double a_ = V[0] + lambda * fmax(V[0], 1e-6), b_ = V[1], c_ = V[2];
  const double d_ = V[3] + lambda * fmax(V[3], 1e-6), e_ = V[4], f_ = V[5] + lambda * fmax(V[5], 1e-6);
  const double m00 = rsqrt(a_), l10 = b_ * m00, l20 = c_ * m00;
  const double m11 = rsqrt(fmax(d_ - l10 * l10, 0.0)), l21 = (e_ - l20 * l10) * m11;
  const double m22 = rsqrt(fmax(f_ - l20 * l20 - l21 * l21, 0.0));
  const double m10 = -(l10 * m00) * m11, m21 = -(l21 * m11) * m22, m20 = -(l20 * m00 + l21 * m10) * m22;
  M[0] = m00; M[1] = m10; M[2] = m11; M[3] = m20; M[4] = m21; M[5] = m22;
  I[0] = m00 * m00 + m10 * m10 + m20 * m20; I[1] = m10 * m11 + m20 * m21; I[2] = m20 * m22;
  I[3] = I[1]; I[4] = m11 * m11 + m21 * m21; I[5] = m21 * m22;
  I[6] = I[2]; I[7] = I[5]; I[8] = m22 * m22;
  // V*^-1 g_l: what the keyframe partials need of this point for b_red = sum W V*^-1 g_l = sum A^T (B V*^-1 g_l)
  vgo[0] = I[0] * g[0] + I[1] * g[1] + I[2] * g[2];
  vgo[1] = I[3] * g[0] + I[4] * g[1] + I[5] * g[2];
  vgo[2] = I[6] * g[0] + I[7] * g[1] + I[8] * g[2];
}

// Lanes per map point in the per-observation kernels: 32 for one window (twice the blocks for its latency chains), 16 in a batch of
// >= 8 windows (points average 16 observations: in a 32-lane group half the lanes idle through the Jacobian arithmetic, and these
// kernels are f64-VALU-bound there).  The group size must not enter the sums, so the 16-lane form keeps TWO accumulators per sum —
// a for the observations a 32-lane group's lane l would take (l, l + 32, ...), b for lane l + 16's (l + 16, l + 48, ...) — and starts
// its shuffle tree with a + b, which is the 32-lane tree's first step; the remaining four steps are the same lanes in the same order.
// (Round 2/3 tried 16 lanes as ONE compile-time value for both: batch +4 %, single window -3.5 %.)
template <int LANES> __device__ __forceinline__ double group_sum(double a, double b) {   // sum over the point's lane group, fixed tree
  static_assert(LANES == 32 || LANES == 16, "a power of two dividing the wave");
  double v = a;
  if (LANES == 32) v += __shfl_xor(v, 16); else v = a + b;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// One 32-lane group per point (2 points per wave).  Observations are stored point-major (CSR).
// (round 3: 115 VGPRs; caps for 3 / 5 blocks per CU: 103.3 -> 102.5 / 134 us per batch iteration.  Round 4, 16 lanes per point with the
// residual's Huber branch: 178 VGPRs uncapped = 2 waves per SIMD, 53.6 us per 32-window launch; capped at 3 blocks (168): 40.1; at 4 (128, spills): 56.9)
#ifndef ORBX_BUILD_MINBLOCKS
#define ORBX_BUILD_MINBLOCKS 3
#endif
template <int LANES>
__global__ __launch_bounds__(256, ORBX_BUILD_MINBLOCKS) void ba_build_kernel(const BaWin* __restrict__ wins, BaCam cam, int iter) {
  __shared__ double sRt[12 * BA_MAX_K];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaDims d = win.d;
  BaState* S = win.S;
  if (S->done || (int)blockIdx.x * (256 / LANES) >= max(d.M, 1)) return;   // (block 0 of a window without points still counts the iteration)
  double *P0 = win.P0, *P1 = win.P1;
  const double* __restrict__ Rt_fix = win.Rt_fix;
  const int* __restrict__ pt_start = win.pt_start; const int* __restrict__ o_kf = win.o_kf;
  const double* __restrict__ o_uv = win.o_uv;
  const BaSet out = ba_set(win, S->bsel);                                // (the fused loop launches this kernel for iteration 0 only: set 0)
  double* __restrict__ Vinv = out.Vinv /*M*9*/; double* __restrict__ gl = out.gl /*M*3*/; double* __restrict__ vg = out.vg /*M*3*/;
  double* __restrict__ pt_chi2 = out.pt_chi2 /*M*/; double* __restrict__ pt_glsq = out.pt_glsq /*M*/;
  double* __restrict__ oP = out.oP;
  const double lambda = S->lambda;
  const double* params = ba_cur(S, P0, P1);
  if (blockIdx.x == 0 && threadIdx.x == 0) S->iters = iter + 1;          // local_ba_lm.rs:1017
  block_poses(params, d.K, cam.inertial, sRt);
  if (blockIdx.x == 0) for (int a = threadIdx.x; a < 12 * d.K; a += blockDim.x) out.Rt_cur[a] = sRt[a];   // for the kernels that rebuild the blocks
  const int lane32 = threadIdx.x & (LANES - 1);
  // a group takes the points g0, g0 + (groups of the launch), ...: one point per group for a single window (as many short blocks as
  // possible for its latency chain), several in a large batch, where the block's prologue (K poses) is then paid once for all of
  // them.  A point's arithmetic does not depend on which block or round handles it.
  const int g0 = (blockIdx.x * blockDim.x + threadIdx.x) / LANES, gstride = (int)gridDim.x * (256 / LANES);
  // Software pipeline over a group's points (a large batch gives a group several): a point's observation range and position are
  // requested two points ahead, the first observation of each lane (its keyframe and pixel) one point ahead — the chain pt_start ->
  // o_kf / o_uv -> arithmetic was two exposed memory round trips per point, in a kernel whose arithmetic takes a quarter of its time.
  struct PtPre { int s, e; double X[3]; int k0; double u0, v0; };
  auto load_pt = [&](int j, PtPre& q) {
    const int jj = min(j, d.M - 1);                                                // (past the last point: a harmless repeat, never used)
    q.s = pt_start[jj]; q.e = pt_start[jj + 1];
    q.X[0] = params[6 * (size_t)d.K + 3 * (size_t)jj]; q.X[1] = params[6 * (size_t)d.K + 3 * (size_t)jj + 1]; q.X[2] = params[6 * (size_t)d.K + 3 * (size_t)jj + 2];
  };
  auto load_obs = [&](PtPre& q) {
    const int i = q.s + lane32;
    const bool in = i < q.e;
    q.k0 = in ? o_kf[i] : 0; q.u0 = in ? o_uv[2 * (size_t)i] : 0.0; q.v0 = in ? o_uv[2 * (size_t)i + 1] : 0.0;
  };
  PtPre pa, pb, pc;
  load_pt(g0, pa); load_pt(g0 + gstride, pb);
  load_obs(pa);
  for (int j = g0; j < d.M; j += gstride) {   // group-uniform
  load_obs(pb);
  load_pt(j + 2 * gstride, pc);
  const double X[3] = {pa.X[0], pa.X[1], pa.X[2]};
  const int s = pa.s, e = pa.e;
  double Va[6] = {0, 0, 0, 0, 0, 0}, ga[3] = {0, 0, 0}, chia = 0.0, Vb[6] = {0, 0, 0, 0, 0, 0}, gb[3] = {0, 0, 0}, chib = 0.0;
  // residual + Jacobian, accumulate V, g_l (lane-strided, then a fixed shuffle tree); per observation only the six numbers the
  // consumers rebuild A, B and W = A^T B from are stored (48 B; until round 3: W itself, 144 B, written in a second pass)
  auto one = [&](int i, int k, double u, double v, double (&V)[6], double (&g)[3], double& chi) {
    double Rt[12];
    if (k >= 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = sRt[12 * k + a];
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = Rt_fix[12 * (size_t)(-1 - k) + a];
    }
    ba_build_obs(cam, Rt, X, u, v, cam.o_flag ? cam.o_flag[i] : 0, oP + 6 * (size_t)i, V, g, chi);
  };
  for (int i = s + lane32; i < e; i += 32) {
    if (i == s + lane32) one(i, pa.k0, pa.u0, pa.v0, Va, ga, chia);               // (the lane's first observation came with the pipeline)
    else one(i, o_kf[i], o_uv[2 * (size_t)i], o_uv[2 * (size_t)i + 1], Va, ga, chia);
    if (LANES == 16 && i + 16 < e) one(i + 16, o_kf[i + 16], o_uv[2 * (size_t)(i + 16)], o_uv[2 * (size_t)(i + 16) + 1], Vb, gb, chib);
  }
  double V[6], g[3];
#pragma unroll
  for (int a = 0; a < 6; ++a) V[a] = group_sum<LANES>(Va[a], Vb[a]);
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = group_sum<LANES>(ga[a], gb[a]);
  const double chi = group_sum<LANES>(chia, chib);
  double Mm[6], I[9], vgo[3];
  ba_point_matrices(V, g, lambda, Mm, I, vgo);
  if (lane32 == 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) out.Mz[6 * (size_t)j + a] = Mm[a];
#pragma unroll
    for (int a = 0; a < 9; ++a) Vinv[9 * (size_t)j + a] = I[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) { gl[3 * (size_t)j + a] = g[a]; vg[3 * (size_t)j + a] = vgo[a]; }
    if (out.Vraw) {
#pragma unroll
      for (int a = 0; a < 6; ++a) out.Vraw[6 * (size_t)j + a] = V[a];
    }
    pt_chi2[j] = chi;
    pt_glsq[j] = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
  }
  pa = pb; pb = pc;
  }
}

#ifndef ORBX_BA_PPS
#define ORBX_BA_PPS 32
#endif
constexpr int BA_PPS_TARGET = ORBX_BA_PPS;   // map points per k-split of the Schur product (multiple of 8).  Round 3 (48-byte observations): 24 / 32 / 48 / 64 points: one window 1.417 / 1.443 / 1.483 / 1.569 ms per solve, 32-window batch 69.2 / 75.6 / 76.5 / 79.9 k LM it/s device-only — the batch wants few, long blocks (half the partials to write and gather), one window many short ones; 32 stays.  (Having it both ways — k-splits summed PAIRS FIRST, the batch's blocks owning a pair and parking the first partial in their own slot of the buffer, one window's gather forming the pairs: same bits either way — was built and passed every batch-equals-single test, but the parked partial's read-modify-write at the end of each block cost what the gather saved: Schur 117.7 -> 138.0, gather 38.9 -> 22.5 us per iteration; withdrawn.)  Earlier sweeps:  24 / 32 / 48 / 64: single window 6.41 / 6.36 / 6.10 / 5.83 k LM it/s, 32-window batch 38.6 / 41.3 / 41.3 / 41.4 k
#ifndef ORBX_BA_KFSPLIT
#define ORBX_BA_KFSPLIT 1
#endif
#ifndef ORBX_BA_KF_THREADS
#define ORBX_BA_KF_THREADS 256
#endif
constexpr int BA_KFSPLIT = ORBX_BA_KFSPLIT;   // round 2 (stored blocks): 16 / 8 / 4 / 2 blocks per keyframe: 32-window batch 36.2 / 39.0 / 40.2 / 40.4 k LM it/s, single window unchanged: the 33 shuffle-tree reductions per block outweigh the observations a block adds up.  Round 3 (recomputed blocks): 8 / 4 / 2 / 1 blocks: keyframe partials 134 / 82.5 / 61.7 / 55.7 us per 32-window iteration, one window 22.4 us throughout: 1
// BA_KFSPLIT blocks per optimised keyframe: partial U_k (21 unique), g_p (6), b_red (6) over a slice of its
// observations; the gather kernel adds the partials in a fixed order.
// NT threads per keyframe block (a template argument: the thread count fixes which observations a thread adds up and the width of the
// final sum over waves, i.e. the rounding — every launch of one build uses the same NT)
// (128 / 256 / 512 threads: keyframe partials of a 32-window batch 27.6 / 34.9 / 49.6 us, one window's fused launch 20.5 / 18.6 / 18.8 us,
// configs[4]'s 131 / 126 / 152 us: 256)
constexpr int BA_KF_THREADS = ORBX_BA_KF_THREADS;
template <int NT>
__device__ __forceinline__ void ba_kf_body(int bx, const BaWin& win, const BaCam& cam) {
  __shared__ double red[NT / 64][33];
  const int* __restrict__ kf_start = win.kf_start; const int* __restrict__ kf_obs = win.kf_obs; const int* __restrict__ kf_pt = win.kf_pt;
  const BaSet cset = ba_set(win, win.S->bsel);
  const double* __restrict__ oP = cset.oP; const double* __restrict__ vg = ba_point_mats(win, win.S, cset).vg;
  double* __restrict__ kfpart = win.kfpart;   /*[K][BA_KFSPLIT][33]*/
  const int k = bx / BA_KFSPLIT, sp = bx % BA_KFSPLIT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double Rt[12];                                                          // this block's keyframe, as the build kernel derived it
#pragma unroll
  for (int a = 0; a < 12; ++a) Rt[a] = cset.Rt_cur[12 * (size_t)k + a];
  const int s0 = kf_start[k], len = kf_start[k + 1] - s0;
  const int s = s0 + (int)((long long)len * sp / BA_KFSPLIT), e = s0 + (int)((long long)len * (sp + 1) / BA_KFSPLIT);
  double acc[33];   // 21 unique U entries, 6 g_p, 6 b_red
#pragma unroll
  for (int a = 0; a < 33; ++a) acc[a] = 0.0;
  // Software pipeline over this thread's observations t = s + tid, + 256, ...: the indices run two rounds ahead, the observation's
  // six numbers and its point's V*^-1 g_l one round — a round's two dependent gathers (index -> data) are in flight under the arithmetic
  // of the rounds before it instead of standing in front of it (the loop was two exposed memory round trips per observation: 55.7 us
  // per 32-window launch with 608 blocks that all run at once, i.e. the time of ONE block's seven rounds).
  int t = s + tid;
  int i1 = -1, j1 = 0, i2 = -1, j2 = 0;                                   // next round's / the round after's indices (-1: none)
  double q0[6], v0[3], q1[6], v1[3];
  bool have0 = t < e;
  if (have0) {
    const int i0 = kf_obs[t], j0 = kf_pt[t];
#pragma unroll
    for (int a = 0; a < 6; ++a) q0[a] = oP[6 * (size_t)i0 + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) v0[a] = vg[3 * (size_t)j0 + a];
  }
  if (t + NT < e) { i1 = kf_obs[t + NT]; j1 = kf_pt[t + NT]; }
  if (t + 2 * NT < e) { i2 = kf_obs[t + 2 * NT]; j2 = kf_pt[t + 2 * NT]; }
  while (have0) {
    const bool have1 = i1 >= 0;
    if (have1) {                                                             // next round's data: issued before this round's arithmetic
#pragma unroll
      for (int a = 0; a < 6; ++a) q1[a] = oP[6 * (size_t)i1 + a];
#pragma unroll
      for (int a = 0; a < 3; ++a) v1[a] = vg[3 * (size_t)j1 + a];
    }
    i1 = i2; j1 = j2;
    if (t + 3 * NT < e) { i2 = kf_obs[t + 3 * NT]; j2 = kf_pt[t + 3 * NT]; } else i2 = -1;
    double A[12], B[6];
    obs_jac_from_proj(cam, Rt, q0[0], q0[1], q0[2], q0[3], A, B);         // no division, no square root: the build kernel stored what they gave
    const double r0 = q0[4], r1 = q0[5];
    int qi = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = a; b < 6; ++b) {                                          // U_ab += A_0a A_0b + A_1a A_1b, the structural zeros' products left out
        double v = acc[qi];
        if (!ba_a1_zero(a) && !ba_a1_zero(b)) v = fma(A[6 + a], A[6 + b], v);
        if (!ba_a0_zero(a) && !ba_a0_zero(b)) v = fma(A[a], A[b], v);
        acc[qi++] = v;
      }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] = ba_arow_acc(acc[21 + a], A, a, r0, r1);
    // W V*^-1 g_l = A^T (B (V*^-1 g_l)): the 2-vector B vg, then A^T of it
    const double t0 = fma(B[0], v0[0], fma(B[1], v0[1], B[2] * v0[2])), t1 = fma(B[3], v0[0], fma(B[4], v0[1], B[5] * v0[2]));
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[27 + a] = ba_arow_acc(acc[27 + a], A, a, t0, t1);
    have0 = have1;
#pragma unroll
    for (int a = 0; a < 6; ++a) q0[a] = q1[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) v0[a] = v1[a];
    t += NT;
  }
#pragma unroll
  for (int a = 0; a < 33; ++a) {
    double v = acc[a];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wave][a] = v;
  }
  __syncthreads();
  if (tid < 33) {
    double v = red[0][tid];
#pragma unroll
    for (int w2 = 1; w2 < NT / 64; ++w2) v += red[w2][tid];                 // (((w0 + w1) + w2) + ...)
    kfpart[((size_t)k * BA_KFSPLIT + sp) * 33 + tid] = v;
  }
}

// S_red partials.  A block owns (column-block pair (bi <= bj) of 128 columns each, k-split ks): the rows (= 3 per map point) of
// its k-split pass through LDS 24 at a time as two operand tiles, Y (columns of block bi) and W (columns of block bj), filled
// straight from the per-observation W blocks — slot (point j, keyframe k) = the sum of the W of its observations (almost
// always one; zero if k does not see j), Y = W V*^-1 with the build kernel's own expression — and every wave accumulates its
// upper 16x16 tiles of the 8 x 8 tile block in registers over all rows: wave q owns tile rows q and 7 - q (9 tiles each in
// a diagonal block, 16 in an off-diagonal one).  v_mfma_f64_16x16x4_f64: lane l feeds A[l&15][l>>4], B[l>>4][l&15];
// D[(l>>4)+4r][l&15], r = 0..3.  The dense [3M][6K] operands never exist in memory: per iteration the Schur product reads
// 144 B per observation instead of 2 x 8 B x 3M x 6K (393 MB at 32 windows of 20 keyframes / 2000 points).
constexpr int SCH_R = 24;                 // rows per LDS tile (8 points, 6 MFMA k-steps)
constexpr int SCH_PITCH = 136;            // doubles per LDS row: 128 + 16 — the two 16-lane row groups of a ds_read_b64 half-wave land 32 banks apart
// one tile step of ba_schur_body's product for wave Q (tile rows Q and 7 - Q of the block pair; DG: a diagonal pair, upper tiles only)
// CM: the tile columns of block bj that exist (8, or fewer in the last block: n = 294 is 8 + 8 + 3 tiles — multiplying the 5 padded
// ones as well was 37 % of the launch's MFMAs at configs[4]'s size).  A compile-time bound like Q and DG, for the same reason.
template <int Q, bool DG, int CM>
__device__ __forceinline__ void ba_schur_gen_mfma(double4_t (&acc)[2][8], const double* __restrict__ sY, const double* __restrict__ sW, int lane) {
#pragma unroll
  for (int kq = 0; kq < SCH_R / 4; ++kq) {
    const int row = 4 * kq + (lane >> 4);
    const double a0 = sY[row * SCH_PITCH + Q * 16 + (lane & 15)];
    const double a1 = sY[row * SCH_PITCH + (7 - Q) * 16 + (lane & 15)];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
      const double bv = (DG ? sY : sW)[row * SCH_PITCH + c * 16 + (lane & 15)];     // (a diagonal pair multiplies its one Z tile with itself)
      if (!DG || c >= Q) acc[0][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][c], 0, 0, 0);
      if (!DG || c >= 7 - Q) acc[1][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][c], 0, 0, 0);
    }
  }
}
template <int CM>
__device__ __forceinline__ void ba_schur_gen_step(int wave, bool diag, double4_t (&acc)[2][8], const double* __restrict__ sY, const double* __restrict__ sW, int lane) {
  switch (2 * wave + (diag ? 1 : 0)) {
    case 0: ba_schur_gen_mfma<0, false, CM>(acc, sY, sW, lane); break;
    case 1: ba_schur_gen_mfma<0, true, CM>(acc, sY, sW, lane); break;
    case 2: ba_schur_gen_mfma<1, false, CM>(acc, sY, sW, lane); break;
    case 3: ba_schur_gen_mfma<1, true, CM>(acc, sY, sW, lane); break;
    case 4: ba_schur_gen_mfma<2, false, CM>(acc, sY, sW, lane); break;
    case 5: ba_schur_gen_mfma<2, true, CM>(acc, sY, sW, lane); break;
    case 6: ba_schur_gen_mfma<3, false, CM>(acc, sY, sW, lane); break;
    default: ba_schur_gen_mfma<3, true, CM>(acc, sY, sW, lane); break;
  }
}

__device__ __forceinline__ void ba_schur_body(int bx, const BaWin& win, const BaCam& cam, double* __restrict__ sY, double* __restrict__ sW) {
  const BaDims& d = win.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npair = d.ncb * (d.ncb + 1) / 2;
  if (bx >= npair * d.ksplit) return;
  const int pr = bx / d.ksplit, ks = bx - pr * d.ksplit;
  int bi = 0, rem = pr;
  while (rem >= d.ncb - bi) { rem -= d.ncb - bi; ++bi; }
  const int bj = bi + rem;
  const bool diag = bi == bj;
  const int j_begin = ks * d.pps;                                   // first map point of this k-split
  const BaSet cset = ba_set(win, win.S->bsel);
  const double* __restrict__ oP = cset.oP; const double* __restrict__ Rt_cur = cset.Rt_cur; const double* __restrict__ Mzp = ba_point_mats(win, win.S, cset).Mz;
  const int* __restrict__ slot_first = win.slot_first; const int* __restrict__ obs_next = win.obs_next;
  double4_t acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[a][c] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int trow[2] = {wave, 7 - wave};                             // this wave's two tile rows inside the block
  const int cm = min(8, d.ntile - 8 * bj);                          // tile columns of block bj that exist
  const int kf_lo_i = (128 * bi) / 6, kf_lo_j = (128 * bj) / 6;   // first keyframe whose 6 columns reach into the block; at most 23 do
  constexpr int NPT = SCH_R / 3, NSLOT = NPT * 23;
  // columns past 6K and slots of keyframes that do not exist are never written below: zero both tiles once
  for (int u = tid; u < 2 * SCH_R * SCH_PITCH; u += 256) sY[u] = 0.0;   // (sW follows sY)
  const int nside = diag ? 1 : 2;
  // slot u of this thread (at most two: 8 points x 23 keyframes x sides over 256 threads) -> first observation of (point,
  // keyframe), fetched one tile ahead so that the fill is ONE dependent round trip (oW) instead of two
  auto slot_index = [&](int u, int j0) -> int {
    if (u >= nside * NSLOT) return -2;
    const int side = u / NSLOT, v = u - side * NSLOT;
    const int pj = v / 23, kk = v - pj * 23;
    const int k = (side ? kf_lo_j : kf_lo_i) + kk, j = j0 + pj;
    const int cb = (side ? bj : bi) * 128;
    if (k >= d.K || 6 * k >= cb + 128 || 6 * k + 6 <= cb) return -2;   // no such keyframe, or none of its columns in the block
    return j < d.M ? slot_first[(size_t)j * d.K + k] : -1;
  };
  int nxt[2] = {slot_index(tid, j_begin), slot_index(tid + 256, j_begin)};
  // (the whole loop once per column count: two multiply variants inside one loop body cost 67 spilled registers)
  auto run = [&](auto cm_) {
  constexpr int CM = decltype(cm_)::value;
  for (int j0 = j_begin; j0 < j_begin + d.pps; j0 += NPT) {
    // ---- fill: one thread per (side, point, keyframe): side 0 = the Y tile (columns of block bi; in a diagonal block also the W
    // tile, same columns), side 1 = the W tile of block bj
    __syncthreads();                                                // the previous tile has been consumed (and the zero fill is done)
    const int cur[2] = {nxt[0], nxt[1]};
    if (j0 + NPT < j_begin + d.pps) { nxt[0] = slot_index(tid, j0 + NPT); nxt[1] = slot_index(tid + 256, j0 + NPT); }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (cur[q] == -2) continue;
      const int u = tid + 256 * q;
      const int side = u / NSLOT, v = u - side * NSLOT;
      const int pj = v / 23, kk = v - pj * 23;
      const int k = (side ? kf_lo_j : kf_lo_i) + kk, j = j0 + pj;
      const int cb = (side ? bj : bi) * 128;
      double w[18];
#pragma unroll
      for (int t = 0; t < 18; ++t) w[t] = 0.0;
      const double* Rk = Rt_cur + 12 * (size_t)k;                     // (read where used: 16 accumulator tiles leave no registers to park it in)
      double Mj[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) Mj[t] = j < d.M ? Mzp[6 * (size_t)j + t] : 0.0;
      for (int i = cur[q]; i >= 0; i = obs_next[i]) obs_z_from_stored<true>(cam, Rk, oP + 6 * (size_t)i, Mj, w);   // Z of the slot (0 + Z for the first)
      double* __restrict__ dstT = side == 0 ? sY : sW;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int col = 6 * k + a - cb;
        if (col < 0 || col >= 128) continue;
        dstT[(3 * pj + 0) * SCH_PITCH + col] = w[a]; dstT[(3 * pj + 1) * SCH_PITCH + col] = w[6 + a]; dstT[(3 * pj + 2) * SCH_PITCH + col] = w[12 + a];
      }
    }
    __syncthreads();
    // ---- multiply: 6 k-steps of 4 rows (wave index and diagonal flag as template arguments: with run-time predicates around the
    // MFMAs a tile step was dozens of basic blocks and no operand read could be scheduled above the matrix instructions, as in the
    // one-column-block body before its consumer loop was specialised)
    ba_schur_gen_step<CM>(wave, diag, acc, sY, sW, lane);
  }
  };
  if (cm <= 2) run(std::integral_constant<int, 2>{});
  else if (cm <= 4) run(std::integral_constant<int, 4>{});
  else if (cm <= 6) run(std::integral_constant<int, 6>{});
  else run(std::integral_constant<int, 8>{});
  // ---- partials out: the gather kernel's layout, part[(upper tile, ks)][16][16]
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ti = bi * 8 + trow[a];
    if (ti >= d.ntile) continue;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int tj = bj * 8 + c;
      if (tj >= d.ntile || tj < ti) continue;
      const int tile = ti * d.ntile - ti * (ti - 1) / 2 + (tj - ti);
      double* o = win.part + ((size_t)tile * d.ksplit + ks) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[a][c][r];
    }
  }
}

#ifdef ORBX_SCHUR_STAMPS
// debug build only (-DORBX_SCHUR_STAMPS, scripts/ba_schur_stamps.py): s_memtime ticks per phase of the consumer wave 0 of every ba_schur_diag_ws_body workgroup, summed; [6] / [12] = the wave's life in s_memrealtime ticks (100 MHz) / s_memtime ticks (shader clock), [7] = workgroups, [8..11] = producer wave 4
__device__ unsigned long long g_schur_stamps[16];
#define SCHUR_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += now_ - t_prev_; t_prev_ = now_; } while (0)   /* per-thread sums, flushed once at the end: an atomic per stamp distorted what it measured */
#define SCHUR_STAMP_FLUSH(a, b) do { if (threadIdx.x == 0 || threadIdx.x == 256) for (int k_ = (a); k_ < (b); ++k_) atomicAdd(&g_schur_stamps[k_], st_acc_[k_]); } while (0)
extern "C" int orbx_debug_schur_stamps(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_schur_stamps), 128) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; z[2] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(g_schur_stamps), z, 128) != hipSuccess) return -1; }
  return 0;
}
#else
#define SCHUR_STAMP(k) do { } while (0)
#define SCHUR_STAMP_FLUSH(a, b) do { } while (0)
#endif
// The same product for windows whose reduced system fits ONE 128-column block (K <= 21 optimised keyframes: every local-BA window of
// the reference, max_covisible_keyframes = 20): only the diagonal block pair exists, so a wave's 9 upper tiles share 9 accumulators —
// slot c holds tile (q, c) for c >= q and tile (7-q, 7-c) for c < q, the ninth the diagonal tile (7-q, 7-q): 72 registers instead of the
// 128 of the general body — and every slot thread fills at most one (point, keyframe) slot per tile.
// PRODUCER and CONSUMER waves of one 512-thread workgroup per CU: waves 4-7 build tile T + 1 (the slot's W = A^T B from its 32 stored
// bytes, Y = W V*^-1; the slot index fetched two tiles ahead, the stored numbers one) into the second of two LDS buffers while waves 0-3
// run the 54 MFMAs of tile T out of the first — one barrier per tile — and a workgroup walks `spb` consecutive k-splits (one for a
// single window, enough in a batch that the launch is one workgroup per CU), writing each split's partial in its own slot.  Same slots,
// same operands, same k order, same v_mfma_f64_16x16x4_f64 sequence per tile as ba_schur_body: bit-identical partials.
// What it bought and what it showed (phase stamps of a consumer and a producer wave, scripts/ba_schur_stamps.py): one window 23.5 ->
// 18.3 us per fused launch (its 63 workgroups own a CU each either way; the prologue and the fill left the consumers' path).  The
// 32-window launch stayed at 117-118 us, and the stamps say why: per tile a consumer wave spends 3.56 k cycles in its 54 MFMAs — 64
// cycles each: the matrix pipe is saturated for the length of the phase — and then waits 3.1 k cycles at the barrier for its SIMD's
// producer wave, which needed 5.6 k cycles for a tile of ~280 instructions: 3.5 k of them standing still while the MFMAs ran, the rest
// once they had finished.  f64 VALU instructions and f64 MFMAs do not execute side by side on a SIMD — MI355X quotes the same 78.6
// TFLOP/s for vector and matrix f64: one double-precision datapath — so the fill's arithmetic serialises with the matrix instructions
// whichever wave issues it, and a tile costs the SIMD 3.5 k (MFMA) + ~2 k (fill) + barrier and store ≈ 9 k cycles in this body and in
// its predecessor (fill -> barrier -> MFMA -> barrier in the same four waves, two workgroups per CU) alike.  The clock stays at
// 2.25 GHz throughout (s_memtime against s_memrealtime inside the kernel).  Neither a second register set of slot data (numbers two
// tiles ahead) nor halving the producers' instruction stream moved the launch: 39 % of the f64 MFMA peak is what a product whose
// operands take a third as much f64 arithmetic to BUILD as to multiply can reach on this datapath; the remaining lever is the
// arithmetic of the fill itself.
#ifndef ORBX_SCHW_PRODUCER_PRIO
#define ORBX_SCHW_PRODUCER_PRIO 1
#endif
#ifndef ORBX_SCHW_ROWS
#define ORBX_SCHW_ROWS 48
#endif
constexpr int SCHW_THREADS = 512;      // 4 consumer + 4 producer waves
constexpr int SCHW_R = ORBX_SCHW_ROWS;                                 // rows of a tile of the producer / consumer body: 16 map points, 12 MFMA k-steps
constexpr int SCHW_NPT = SCHW_R / 3, SCHW_SPT = SCHW_NPT / 8;          // points per tile; slots per producer thread (8 points x 23 keyframes of threads)
constexpr int SCHW_TILE = SCHW_R * SCH_PITCH;
static_assert(SCHW_R % 24 == 0 && SCHW_R <= 48, "whole passes of the producers' 8 x 23 slot threads; two tiles of 48 rows are 104 KB");
#ifndef ORBX_BA_GATHER_MAX_BLOCKS
#define ORBX_BA_GATHER_MAX_BLOCKS 2048
#endif
constexpr int BA_GATHER_MAX_BLOCKS = ORBX_BA_GATHER_MAX_BLOCKS;   // blocks per window of the gather launch (256 / 1024 / 2048: configs[4] 23.0 / 21.3 / 19.8 us; a 20-keyframe window needs 203)
constexpr int BA_GATHER_LANES = 8;     // shares of a window's k-splits (ba_gather_kernel's lanes per element = the Schur launch's workgroups per window in a large batch)
constexpr size_t SCHW_LDS_BYTES = 2 * (size_t)SCHW_TILE * sizeof(double);      // two Z buffers: 104 448 B at 48 rows
template <int Q>
__device__ __forceinline__ void ba_schur_ws_consume(const BaWin& win, const double* __restrict__ lds, int ks0, int nT, int tps, bool sums) {
  const BaDims& d = win.d;
  const int lane = threadIdx.x & 63;
#ifdef ORBX_SCHUR_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(), st_acc_[16] = {0};
  const unsigned long long t_real0_ = __builtin_amdgcn_s_memrealtime(), t_clk0_ = t_prev_;
#endif
  double4_t acc[8], accx = double4_t{0.0, 0.0, 0.0, 0.0}, run[9];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 9; ++c) run[c] = double4_t{0.0, 0.0, 0.0, 0.0};
  // this wave's nine partial tiles: element offsets inside a k-split's 256-double tile slots (tile index x ksplit x 256), or -1
  const int lofs = ((lane >> 4)) * 16 + (lane & 15);
  auto tile_of = [&](int ti, int tj) -> long long {
    if (ti >= d.ntile || tj >= d.ntile) return -1;
    return (long long)(ti * d.ntile - ti * (ti - 1) / 2 + (tj - ti)) * d.ksplit * 256 + lofs;
  };
  long long tofs[9];
#pragma unroll
  for (int c = 0; c < 8; ++c) tofs[c] = c >= Q ? tile_of(Q, c) : tile_of(7 - Q, 7 - c);
  tofs[8] = tile_of(7 - Q, 7 - Q);
  auto store = [&](int ks, long long o, const double4_t& v) {
    if (o < 0) return;
    double* p = win.part + o + (size_t)ks * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[64 * r] = v[r];                    // rows (lane >> 4) + 4 r of the tile
  };
  __syncthreads();                                                    // zero fill done
  __syncthreads();                                                    // tile 0 is in buffer 0
  SCHUR_STAMP(0);
  for (int T = 0; T < nT; ++T) {
    const double* __restrict__ sZ = lds + (size_t)(T & 1) * SCHW_TILE;
#ifndef ORBX_SCHW_NO_MFMA          // (timing experiments only: the producers alone)
#pragma unroll
    for (int kq = 0; kq < SCHW_R / 4; ++kq) {
      const int row = 4 * kq + (lane >> 4);
      const double a0 = sZ[row * SCH_PITCH + Q * 16 + (lane & 15)];
      const double a1 = sZ[row * SCH_PITCH + (7 - Q) * 16 + (lane & 15)];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const double bv = sZ[row * SCH_PITCH + c * 16 + (lane & 15)];
        if (c >= Q) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[c], 0, 0, 0);              // tile (Q, c)
        if (c > 7 - Q) acc[7 - c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[7 - c], 0, 0, 0);  // tile (7-Q, c) in slot 7-c < Q
        if (c == 7 - Q) accx = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, accx, 0, 0, 0);              // tile (7-Q, 7-Q)
      }
    }
#endif
    SCHUR_STAMP(4);
    if ((T + 1) % tps == 0) {                                         // the k-split is complete
      const int ks = ks0 + T / tps;
      if (!sums) {                                                    // its partial out, accumulators cleared
#pragma unroll
        for (int c = 0; c < 8; ++c) { store(ks, tofs[c], acc[c]); acc[c] = double4_t{0.0, 0.0, 0.0, 0.0}; }
        store(ks, tofs[8], accx);
        accx = double4_t{0.0, 0.0, 0.0, 0.0};
      } else {
        // this workgroup owns one of the gather's shares: it adds the share's partials itself, in the order the gather would (running
        // sum + next partial, starting from the first), and leaves the ONE sum in the first k-split's slot — 1/spb of the bytes
        // (the three cases around the whole set of nine, not inside it: as a test per accumulator this was eighteen scalar branches
        // per k-split, 2.1 k cycles of a consumer's 20 k per k-split)
        const bool first = T + 1 == tps, last = T + 1 == nT;
        if (first) {
#pragma unroll
          for (int c = 0; c < 9; ++c) run[c] = c < 8 ? acc[c] : accx;
        } else {
#pragma unroll
          for (int c = 0; c < 9; ++c) {
            const double4_t& a = c < 8 ? acc[c] : accx;
            double4_t& r = run[c];
            r[0] = r[0] + a[0]; r[1] = r[1] + a[1]; r[2] = r[2] + a[2]; r[3] = r[3] + a[3];
          }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
        accx = double4_t{0.0, 0.0, 0.0, 0.0};
        if (last) {
#pragma unroll
          for (int c = 0; c < 9; ++c) store(ks0, tofs[c], run[c]);
        }
      }
      SCHUR_STAMP(5);
    }
    __syncthreads();
    SCHUR_STAMP(1);
  }
#ifdef ORBX_SCHUR_STAMPS
  SCHUR_STAMP_FLUSH(0, 6);
  if (threadIdx.x == 0) { const unsigned long long t_real1_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_schur_stamps[12], __builtin_amdgcn_s_memtime() - t_clk0_); atomicAdd(&g_schur_stamps[6], t_real1_ - t_real0_); atomicAdd(&g_schur_stamps[7], 1ull);
    atomicMin(&g_schur_stamps[2], t_real0_); atomicMax(&g_schur_stamps[3], t_real1_); }   // ([2], [3]: first start / last end of the launches since the reset, 100 MHz)
#endif
}

__device__ __forceinline__ void ba_schur_diag_ws_body(int bx, int spb, const BaWin& win, const BaCam& cam, double* __restrict__ lds) {
  const BaDims& d = win.d;
  if (win.part_sums) spb = (d.ksplit + BA_GATHER_LANES - 1) / BA_GATHER_LANES;   // one workgroup per share of the gather
  const int ks0 = bx * spb, ks1 = min(ks0 + spb, d.ksplit);
  if (ks0 >= d.ksplit) return;                                        // (the whole workgroup, before any barrier)
  const int tid = threadIdx.x;
  const int tps = d.pps / SCHW_NPT;                                   // tiles per k-split (pps is a multiple of 16)
  const int nT = (ks1 - ks0) * tps;
  const int j_begin = ks0 * d.pps;
  for (int u = tid; u < 2 * SCHW_TILE; u += SCHW_THREADS) lds[u] = 0.0;    // columns past 6K and slots of absent keyframes stay zero in both buffers
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);            // (a scalar: the two roles and the consumer's tile rows branch on SGPRs; 108.4 -> 107.2 us per 32-window launch)
  if (wv >= 4) {
#ifdef ORBX_SCHUR_STAMPS
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(), st_acc_[16] = {0};
#endif
    // ---- producers: thread (pj, k) fills the slots (point j0 + pj + 8 h, keyframe k), h = 0 .. SCHW_SPT - 1, of every tile: 8 x 23
    // threads of the four waves.  The waves dispatched second lose every arbitration for the SIMD's issue slots to the consumers on age
    // (MI355X_MICROARCH.md, "two waves per SIMD"), and these are the waves the tile waits for: one s_setprio for the whole loop.
#if ORBX_SCHW_PRODUCER_PRIO
    __builtin_amdgcn_s_setprio(ORBX_SCHW_PRODUCER_PRIO);
#endif
    const int f = tid - 256, pj = f / 23, k = f - pj * 23;
    const bool slot = pj < 8 && k < d.K && 6 * k < 128;
    const BaSet cset = ba_set(win, win.S->bsel);
    const double* __restrict__ oP = cset.oP; const double* __restrict__ Mzp = ba_point_mats(win, win.S, cset).Mz;
    const int* __restrict__ slot_first = win.slot_first; const int* __restrict__ obs_next = win.obs_next;
    // (the slot index runs two tiles ahead, the stored numbers one; a second register set — numbers two tiles ahead — measured no
    // difference in round 3 and again in round 4, 86.1-88.9 against 86.3-87.0 us: the loads are not what the producers wait for)
    struct SlotData { double pq[SCHW_SPT][4], M[SCHW_SPT][6]; int chain[SCHW_SPT], idx[SCHW_SPT]; } sd;    // per slot: (x, y, 1/z, sqrt w) and the point's M; empty slot: zeros
    double Rk[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) Rk[t] = slot ? cset.Rt_cur[12 * (size_t)k + t] : 0.0;
    auto slot_of = [&](int T, int h) -> int {                         // first observation of (point of tile T, keyframe k), or -1
      const int j = j_begin + T * SCHW_NPT + pj + 8 * h;
      return (slot && T < nT && j < d.M) ? slot_first[(size_t)j * d.K + k] : -1;
    };
    auto fetch = [&](int T) {                                          // tile T's stored numbers (32 B) and M (48 B) -> registers, by sd.idx
#pragma unroll
      for (int h = 0; h < SCHW_SPT; ++h) {
        const int j = j_begin + T * SCHW_NPT + pj + 8 * h, i = sd.idx[h];
#pragma unroll
        for (int t = 0; t < 4; ++t) sd.pq[h][t] = i >= 0 ? oP[6 * (size_t)i + t] : 0.0;
#pragma unroll
        for (int t = 0; t < 6; ++t) sd.M[h][t] = (slot && T < nT && j < d.M) ? Mzp[6 * (size_t)j + t] : 0.0;
        sd.chain[h] = i >= 0 ? obs_next[i] : -1;
      }
    };
    auto fill = [&](double* __restrict__ sZ) {
      if (!slot) return;
#ifdef ORBX_SCHW_NO_FILL           // (timing experiments only: the consumers alone)
      return;
#endif
      // Z = A^T (B M^T) of each of the thread's slots: straight-line code (an empty slot's zeros go through the same arithmetic), so
      // the slots' dependent chains are scheduled into each other; then the rare second observation of a point by this keyframe
      double w[SCHW_SPT][18];
#pragma unroll
      for (int h = 0; h < SCHW_SPT; ++h) obs_z_from_stored<false>(cam, Rk, sd.pq[h], sd.M[h], w[h]);
#pragma unroll
      for (int h = 0; h < SCHW_SPT; ++h)
        for (int i = sd.chain[h]; i >= 0; i = obs_next[i]) obs_z_from_stored<true>(cam, Rk, oP + 6 * (size_t)i, sd.M[h], w[h]);
      // (columns 6 k + a <= 125 for k <= 20: no bounds test; the stores pair up into ds_write_b128)
#pragma unroll
      for (int h = 0; h < SCHW_SPT; ++h)
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const int col = 6 * k + a, r0 = 3 * (pj + 8 * h);
          sZ[(r0 + 0) * SCH_PITCH + col] = w[h][a]; sZ[(r0 + 1) * SCH_PITCH + col] = w[h][6 + a]; sZ[(r0 + 2) * SCH_PITCH + col] = w[h][12 + a];
        }
    };
    // build tile T into buffer T & 1, then request tile T + 1's numbers and tile T + 2's slot indices
    auto step = [&](int T) {
      fill(lds + (size_t)(T & 1) * SCHW_TILE);
      SCHUR_STAMP(8);
      fetch(T + 1);                                                    // (by the indices loaded a step ago; past the last tile: zeros)
#pragma unroll
      for (int h = 0; h < SCHW_SPT; ++h) sd.idx[h] = slot_of(T + 2, h);
      SCHUR_STAMP(9);
    };
#pragma unroll
    for (int h = 0; h < SCHW_SPT; ++h) sd.idx[h] = slot_of(0, h);
    fetch(0);
#pragma unroll
    for (int h = 0; h < SCHW_SPT; ++h) sd.idx[h] = slot_of(1, h);
    __syncthreads();                                                  // zero fill done
    step(0);                                                          // tile 0 -> buffer 0
    __syncthreads();
    SCHUR_STAMP(11);
    for (int T = 0; T < nT; ++T) {                                    // (the consumers work on tile T)
      if (T + 1 < nT) step(T + 1);
      __syncthreads();
      SCHUR_STAMP(10);
    }
    SCHUR_STAMP_FLUSH(8, 12);
    return;
  }
  // ---- consumers: wave q owns tile rows q and 7 - q (slot c: tile (q, c) for c >= q, tile (7-q, 7-c) for c < q; the ninth: (7-q, 7-q)).
  // The wave index is a template argument of the loop: with q a run-time value every `if (c >= q)` around an MFMA was a branch, a tile
  // step was some forty basic blocks, and no ds_read of the next k-step could be scheduled above the MFMAs of this one (the matrix pipe
  // ran 3456 of a step's ~6000 cycles).
  switch (wv) {
    case 0: ba_schur_ws_consume<0>(win, lds, ks0, nT, tps, win.part_sums != 0); break;
    case 1: ba_schur_ws_consume<1>(win, lds, ks0, nT, tps, win.part_sums != 0); break;
    case 2: ba_schur_ws_consume<2>(win, lds, ks0, nT, tps, win.part_sums != 0); break;
    default: ba_schur_ws_consume<3>(win, lds, ks0, nT, tps, win.part_sums != 0); break;
  }
}

// One window: keyframe partials and Schur blocks in ONE launch (both consume the build kernel's output and feed the solve;
// a launch less on a latency-bound chain, and ~200 blocks do not compete for LDS or registers).  Same block bodies: same results.
// (DIAG: 512-thread workgroups with 102 KB of dynamic LDS for the producer / consumer Schur body; the keyframe blocks use the first 256
// threads; the general body keeps 256-thread workgroups and its static tiles)
constexpr int BA_KFS_GEN_THREADS = BA_KF_THREADS > 256 ? BA_KF_THREADS : 256;   // the fused launch with the general Schur body (which uses the first 256)
static_assert(BA_KF_THREADS % 64 == 0 && BA_KF_THREADS >= 64 && BA_KF_THREADS <= SCHW_THREADS, "keyframe blocks: whole waves, at most the fused launch's workgroup");
template <bool DIAG>
__global__ __launch_bounds__(DIAG ? SCHW_THREADS : BA_KFS_GEN_THREADS, DIAG ? 1 : (BA_KFS_GEN_THREADS > 256 ? 1 : 2)) void ba_kf_schur_kernel(const BaWin* __restrict__ wins, BaCam cam) {
  extern __shared__ __align__(16) double s_dyn[];
  __shared__ double s_tiles[DIAG ? 1 : 2 * SCH_R * SCH_PITCH];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  if (win.S->done) return;
  const int nkf = win.d.K * BA_KFSPLIT;
  if ((int)blockIdx.x < nkf) { if (threadIdx.x < BA_KF_THREADS) ba_kf_body<BA_KF_THREADS>((int)blockIdx.x, win, cam); }
  else if (DIAG) ba_schur_diag_ws_body((int)blockIdx.x - nkf, 1, win, cam, s_dyn);
  else if (threadIdx.x < 256) ba_schur_body((int)blockIdx.x - nkf, win, cam, s_tiles, s_tiles + SCH_R * SCH_PITCH);
}

// (their own launches in a batch: 51 KB of LDS and ~200 VGPRs per block would otherwise throttle the thousands of small keyframe blocks too)
// (242 VGPRs, 2 waves per SIMD: a cap of 128 VGPRs / 4 waves per SIMD measured 61.1 -> 61.5 us per batch iteration, 64 VGPRs with spills 78 us:
// the kernel is not occupancy-bound)
// (end of round 3, 182 VGPRs once its loads were global loads: the 640 workgroups of a 32-window batch are 1.25 rounds at 2 per CU — a cap
// of 168 VGPRs (2 spilled) holds 3 per CU, one round: 36.8 -> 32.3 us per batch iteration)
#ifndef ORBX_KF_MINBLOCKS
#define ORBX_KF_MINBLOCKS 3
#endif
// Workgroups go to the 8 XCDs round-robin in launch order, and a window's keyframe blocks share cache lines: consecutive observations
// (48 B each) belong to the same point and to DIFFERENT keyframes, so a 128-byte line of the store is wanted by two or three blocks.  With
// the plain (keyframe, window) grid those sit on different XCDs and every L2 fetches the line for itself; when the window count is a
// multiple of 8 the launch order is re-read so that all blocks of a window land on XCD (window mod 8) and walk the store side by side under
// one L2.  Which block computes a keyframe's partial does not enter its value.
#ifndef ORBX_KF_XCD
#define ORBX_KF_XCD 1
#endif
__global__ __launch_bounds__(BA_KF_THREADS, ORBX_KF_MINBLOCKS) void ba_kf_kernel(const BaWin* __restrict__ wins, BaCam cam) {
  int bx = blockIdx.x, by = blockIdx.y;
  if (ORBX_KF_XCD && (gridDim.y & 7) == 0) {
    const int lin = by * (int)gridDim.x + bx, xcd = lin & 7, slot = lin >> 3;      // slot: this block's place in its XCD's queue
    by = xcd + 8 * (slot / (int)gridDim.x);
    bx = slot - (slot / (int)gridDim.x) * (int)gridDim.x;
  }
  const BaWin win = ba_win_global(wins, by);
  if (win.S->done || bx >= win.d.K * BA_KFSPLIT) return;
  ba_kf_body<BA_KF_THREADS>(bx, win, cam);
}

// DIAG: every window of the launch has a reduced system of at most 128 columns (host-checked)
// (three blocks per CU for the one-column-block body; the general body holds 2 x 8 accumulator tiles = 256 VGPRs and would spill under
// that cap: as (256, 3) its launch over four 50-keyframe windows ran 0.94 ms, 13 % of the f64 matrix peak)
// (their own launch in a batch.  DIAG — every window one column block —: the producer / consumer body, `spb` consecutive k-splits per
// workgroup, one 512-thread workgroup per CU; until round 3's second half: ba_schur_diag_body, two 256-thread workgroups per CU, 118 us
// per 32-window launch.  Otherwise the general body.)
template <bool DIAG>
__global__ __launch_bounds__(DIAG ? SCHW_THREADS : 256, DIAG ? 1 : 2) void ba_schur_kernel(const BaWin* __restrict__ wins, BaCam cam, int spb) {
  extern __shared__ __align__(16) double s_dyn[];
  __shared__ double s_tiles[DIAG ? 1 : 2 * SCH_R * SCH_PITCH];     // Y and W operand tiles of the general body (51 KB)
  const BaWin win = ba_win_global(wins, blockIdx.y);
  if (win.S->done) return;
  if (DIAG) ba_schur_diag_ws_body((int)blockIdx.x, spb, win, cam, s_dyn);
  else ba_schur_body((int)blockIdx.x, win, cam, s_tiles, s_tiles + SCH_R * SCH_PITCH);   // (blocks beyond this window's need return inside)
}

// reduce-buffer layout (doubles): [Sred n*n | U 36K | gp n | bred n | chi2 | glsq], n = 6K
__global__ __launch_bounds__(256) void ba_gather_kernel(const BaWin* __restrict__ wins, int iter_mark) {
  const BaWin win = ba_win_global(wins, blockIdx.y);
  if (win.S->done) return;
  if (iter_mark >= 0 && blockIdx.x == 0 && threadIdx.x == 0) win.S->iters = iter_mark + 1;   // local_ba_lm.rs:1017 — the fused loop has no build launch to count its iterations 1, 2, ...
  const BaDims d = win.d;
  const double* __restrict__ part = win.part; const double* __restrict__ kfpart = win.kfpart;
  const BaSet cset = ba_set(win, win.S->bsel);
  const double* __restrict__ pt_chi2 = cset.pt_chi2; const double* __restrict__ pt_glsq = cset.pt_glsq;
  double* __restrict__ rb = win.rb;
  const int n = 6 * d.K;
  const size_t nn = (size_t)n * n;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  // S_red = sum of the k-split partials, upper triangle computed, mirrored on store.  BA_GATHER_LANES lanes per element: each adds
  // its share of the k-splits in order (ceil(ksplit / lanes) consecutive ones, up to sixteen loads in flight), then a fixed xor tree
  // over the shares — the same for every launch shape.  (One lane per element walked 40-80 dependent-latency loads: a quarter of a
  // batch's iteration.)  When `win.part_sums` is set the Schur launch has already added each share's partials in this same order
  // (ba_schur_diag_ws_body with one workgroup per share) and left ONE tile set per share: the lane reads that instead — a sixteenth of the bytes.
  // A lane adds its share for TWO neighbouring elements (i, j) and (i, j + 1), j even — one 16-byte load per k-split, a wave covers 16
  // elements x 8 shares with 128 contiguous bytes per (share, k-split), half the load instructions of an element per lane.
  constexpr int GL = BA_GATHER_LANES;
  const int ksq = (d.ksplit + GL - 1) / GL;
  const int qd = threadIdx.x & (GL - 1);
  const int k_lo = min(qd * ksq, d.ksplit), k_hi = min(k_lo + ksq, d.ksplit);
  const int npr = (n + 1) >> 1;                                      // element pairs per row
  const size_t npairs = (size_t)n * npr;
  for (size_t pidx = (size_t)(tid / GL); pidx < ((npairs + 63) & ~(size_t)63); pidx += (size_t)(nth / GL)) {   // whole waves stay in the loop (shuffles below)
    const bool in = pidx < npairs;
    const int i = in ? (int)(pidx / npr) : 0, j = in ? 2 * (int)(pidx - (size_t)i * npr) : 0;
    const bool v0 = in && i <= j, v1 = in && i <= j + 1 && j + 1 < n;
    double s0 = 0.0, s1 = 0.0;
    if (v0 || v1) {
      // elements (i, j), (i, j + 1) sit at row i%16, cols j%16, j%16 + 1 of the upper tile (i/16, j/16)
      const int ti = i >> 4, tj = j >> 4;
      const int tile = ti * d.ntile - ti * (ti - 1) / 2 + (tj - ti);
      const double2_t* p = (const double2_t*)(part + (size_t)tile * d.ksplit * 256 + (i & 15) * 16 + (j & 15));
      if (win.part_sums) { if (k_lo < k_hi) { const double2_t v = p[(size_t)k_lo * 128]; s0 = v[0]; s1 = v[1]; } }   // the share's sum sits in its first k-split's slot
      else {
        int ks = k_lo;
        for (; ks + 8 <= k_hi; ks += 8) {
          double2_t v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(ks + q) * 128];
#pragma unroll
          for (int q = 0; q < 8; ++q) { s0 += v[q][0]; s1 += v[q][1]; }
        }
        for (; ks < k_hi; ++ks) { const double2_t v = p[(size_t)ks * 128]; s0 += v[0]; s1 += v[1]; }
      }
    }
#pragma unroll
    for (int off = 1; off < GL; off <<= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if (qd == 0) {
      if (v0) { rb[(size_t)i * n + j] = s0; rb[(size_t)j * n + i] = s0; }
      if (v1) { rb[(size_t)i * n + j + 1] = s1; rb[(size_t)(j + 1) * n + i] = s1; }
    }
  }
  double* q = rb + nn;                 // U [36K] | gp [n] | bred [n] | chi2 | glsq
  for (int i = tid; i < 33 * d.K; i += nth) {
    const int k = i / 33, a = i - 33 * k;
    double v = 0.0;
    for (int sp = 0; sp < BA_KFSPLIT; ++sp) v += kfpart[((size_t)k * BA_KFSPLIT + sp) * 33 + a];
    if (a < 21) {
      int r = 0, c = 0, t = a;
      for (r = 0; r < 6; ++r) { if (t < 6 - r) { c = r + t; break; } t -= 6 - r; }
      q[36 * (size_t)k + r * 6 + c] = v;
      q[36 * (size_t)k + c * 6 + r] = v;
    } else if (a < 27) q[36 * (size_t)d.K + 6 * (size_t)k + (a - 21)] = v;
    else q[36 * (size_t)d.K + n + 6 * (size_t)k + (a - 27)] = v;
  }
  q += 36 * (size_t)d.K;
  if (blockIdx.x == 0) {
    // chi2 and |g_l|^2 over the points, fixed order (one block, tree over 256 partials)
    __shared__ double sh[2][256];
    double c = 0.0, g = 0.0;
    for (int jx = threadIdx.x; jx < d.M; jx += 256) { c += pt_chi2[jx]; g += pt_glsq[jx]; }
    sh[0][threadIdx.x] = c; sh[1][threadIdx.x] = g;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
      if ((int)threadIdx.x < s2) { sh[0][threadIdx.x] += sh[0][threadIdx.x + s2]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s2]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) { q[2 * n] = sh[0][0]; q[2 * n + 1] = sh[1][0]; }
  }
}

// result block (doubles): [0] chi2  [1] gnorm  [2] chol_ok  [3] |dp|^2  [4] |p_pose|^2
// One workgroup.  S lives in LDS (n <= ~135; larger systems take the multi-kernel path below); the right-hand
// side always in LDS.  Right-looking Cholesky: 2 barriers per column, trailing update on a 16x16 thread grid; the two
// triangular solves run in wave 0 alone as row dot products with shuffle reductions (no block barriers).
#ifndef ORBX_BA_SOLVE_THREADS
#define ORBX_BA_SOLVE_THREADS 1024
#endif
constexpr int BA_SOLVE_THREADS = ORBX_BA_SOLVE_THREADS;   // 256 / 512 / 1024 threads: 43.9 / 36.7 / 35.2 us per solve at n = 114 (end of round 3)
constexpr int BA_MAX_N = 768;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// sum over the block in a fixed order: shuffle tree inside each wave, then thread 0 adds the wave totals in wave order
// (two barriers instead of the ten of a shared-memory tree over 1024 threads)
__device__ __forceinline__ double block_sum_fixed(double v, double* s_w /* [blockDim/64] */) {
  v = wave_sum(v);
  __syncthreads();                                        // s_w may still be read from a previous call
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_w[w];
  return t;                                               // same value in every thread
}

// 1/sqrt(x): hardware estimate (v_rsq_f64) + two Newton steps y <- y + y*(1 - x y^2)/2 (full f64 accuracy)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // y <- y + y (e / 2), e = 1 - x y^2.  Written with h = y / 2 (exact) as fma(h, e, y): the same real product h e = y (e / 2) under the
    // same single rounding as fma(y, 0.5 * e, y), but h does not wait for e — three dependent operations per step instead of four on the
    // factorisation's pivot chain.
    const double h = 0.5 * y;
    const double e = fma(-x * y, y, 1.0);
    y = fma(h, e, y);
  }
  return y;
}

// Lane broadcasts inside a row of 16 lanes by DPP row_newbcast (gfx90a+): the f64 forms v_mov_b64_dpp / v_fmac_f64_dpp take the
// value of lane L of the reader's own 16-lane row as source 0 — one instruction where a v_readlane pair through SGPRs plus the fma
// took three (and their SGPR hazards).  The leading s_nop covers the "VALU write -> DPP read of that VGPR" wait states, which the
// compiler's hazard recogniser does not see inside inline assembly.  All 64 lanes must be active.
template <int L> __device__ __forceinline__ double row_bcast_f64(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(L));
  return r;
}
// acc - (bsrc of lane L of this lane's row) * m, one rounding (the same value as fma(-m, that, acc))
template <int L> __device__ __forceinline__ double fnma_row_bcast_f64(double acc, double bsrc, double m) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m), "n"(L));
  return acc;
}
// the same without the wait states: for a caller whose `bsrc` was written at least two instructions ago — the second and later updates of
// a pivot step, which all read the column the first one read (an s_nop is an instruction like any other to a wave that issues one every
// ~8 cycles: 105 of them were a fifth of the 16 x 16 factorisation)
// `after`: the result of the update that carried the wait states — an operand the instruction does not read, there so that the compiler
// cannot schedule this one ahead of it (round 3: in ba_big_factor_kernel's register allocation it did, and column 15 read a stale pivot column)
template <int L> __device__ __forceinline__ double fnma_row_bcast_f64_settled(double acc, double bsrc, double m, double after) {
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(m), "n"(L), "v"(after));
  return acc;
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// Cholesky of a 16x16 block in one wave.  Lane j of every 16-lane row (the four rows of the wave carry the same block) owns row j:
// Lr[i] = a[j][i] for i <= j (rows >= nb: the identity).  Step jj: the pivot and then, for every later column kk, the entry
// L[kk][jj] reach all lanes of the row by DPP row_newbcast — 1 + (15 - jj) instructions where readlane pairs through SGPRs took
// 2 + 3 (15 - jj).  On return Lr holds row j of L; rv[jj] = 1 / L[jj][jj] (written by thread jj; 1.0 past nb).  Returns 0 when a live
// pivot is not positive (uniform).
// pub(jj, column, 1 / L[jj][jj]) is called once per pivot step, right after column jj of L is final (lane j holds L[j][jj]; rows >= jj
// matter): the LDS solve hands each column to the waves that solve the rows below the block while the factorisation is still running.
struct Chol16NoPub { template <typename J> __device__ __forceinline__ void operator()(J, double, double) const {} };
template <typename Pub = Chol16NoPub>
__device__ __forceinline__ int chol16_rows_dpp(double (&Lr)[16], int nb, int tid, double* __restrict__ rv, Pub&& pub = Pub{}) {
  const int j = tid & 15;
  int good = 1;
  double myri = 1.0;
  static_for<16>([&](auto jj_) {
    constexpr int jj = decltype(jj_)::value;
    if (jj >= nb) return;                                                     // (uniform) the identity rows of a short last block: nothing to do
    const double d = row_bcast_f64<jj>(Lr[jj]);
    if (jj < nb && !(d > 0.0)) good = 0;                                    // uniform
    const double ri = rsqrt_nr(d);                                            // (after a bad pivot: NaNs, which nobody stores)
    myri = (j == jj) ? ri : myri;                                             // (stored once, after the last step)
    Lr[jj] = Lr[jj] * ri;                                                     // column jj of L (rows >= jj matter; row jj itself holds d: d / sqrt d)
    pub(jj_, Lr[jj], ri);
    static_for<16>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      // a[j][kk] -= L[j][jj] L[kk][jj] (used for j >= kk); Lr[jj] was written just above: the first update waits for it, the others need not
      if constexpr (kk == jj + 1) Lr[kk] = fnma_row_bcast_f64<kk>(Lr[kk], Lr[jj], Lr[jj]);
      else if constexpr (kk > jj + 1) Lr[kk] = fnma_row_bcast_f64_settled<kk>(Lr[kk], Lr[jj], Lr[jj], Lr[jj + 1]);
    });
  });
  if (tid < 16) rv[tid] = myri;
  return good;
}

// force-inlined into both kernels so that the address space of S (LDS vs global) is known: as an
// out-of-line function it took a generic pointer and every access became a slow flat_load/flat_store.
//
// Blocked right-looking Cholesky with panels of 16 columns (described at the loop); srinv[c] = 1 / L_cc.

#ifdef ORBX_SOLVE_STAMPS
// debug build only (-DORBX_SOLVE_STAMPS, scripts/build_stamps.sh + scripts/ba_solve_stamps.py): s_memtime ticks per phase of the
// one-workgroup solve, summed over solves.  (Ticks are only comparable within one run: the counter's rate is not the shader clock's.)
__device__ unsigned long long g_solve_stamps[16];
#define SOLVE_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += now_ - t_prev_; t_prev_ = now_; } while (0)   /* per-thread sums, flushed once (thread 0): a global read-modify-write per stamp cost 300-500 cycles and made every phase look longer */
#define SOLVE_STAMP_FLUSH() do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 15; ++k_) g_solve_stamps[k_] += st_acc_[k_]; g_solve_stamps[15] += 1; } } while (0)
extern "C" int orbx_debug_solve_stamps(unsigned long long* out16, int reset) {
  if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_solve_stamps), 128) != hipSuccess) return -1;
  if (reset) { const unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_solve_stamps), z, 128) != hipSuccess) return -1; }
  return 0;
}
#else
#define SOLVE_STAMP(k) do { } while (0)
#define SOLVE_STAMP_FLUSH() do { } while (0)
#endif
// (inertial terms: src/optimizer/local_inertial_ba.rs:661-698, :806-880; the kernels that fill these records are further down)
struct BaInertialDev {
  int K, M, E;
  double gw, aw;                 // sqrt of the random-walk informations
  const int* edge_kf;            // [E][2]
  const double* preint;          // [E][11]
};
constexpr int IMU_REC = 18 * 18 + 18 + 2;   // per edge: H (18x18) | g (18) | chi2(imu + random walk) | spare

// Where the reduced system lives in LDS.  LaySquare: row-major n x n, n <= 128 (every local-BA window of the reference).  LayTiled: the
// LOWER 16 x 16 tiles only, tile (ti, tj <= ti) at ((ti (ti + 1)) / 2 + tj) * 256 doubles — 55 tiles = 110 KB at n = 150 (the 15-d
// system of a 10-keyframe inertial window), 66 = 132 KB at n = 176 — for systems that do not fit as a square but need not go through
// the multi-launch path; a 16-column panel's rows are contiguous 16-double tile rows in both.  rd(): an entry right of the diagonal
// TILE does not exist in the tiled layout and reads as the zero the square layout stores there.
struct LaySquare {
  double* p; int n;
  static constexpr bool TILED = false;
  __device__ __forceinline__ double* at(int i, int j) const { return p + (size_t)i * n + j; }
  __device__ __forceinline__ const double* rd(int i, int j, const double*) const { return p + (size_t)i * n + j; }
};
struct LayTiled {
  double* p;
  static constexpr bool TILED = true;
  __device__ __forceinline__ double* at(int i, int j) const {
    const int ti = i >> 4, tj = j >> 4;
    return p + (size_t)((ti * (ti + 1)) / 2 + tj) * 256 + (i & 15) * 16 + (j & 15);
  }
  __device__ __forceinline__ const double* rd(int i, int j, const double* zero) const { return (j >> 4) > (i >> 4) ? zero : at(i, j); }
};
constexpr int BA_TILED_MAX_N = 176;    // 66 lower tiles = 132 KB beside the 20-23 KB of static arrays
constexpr size_t BA_TILED_LDS_MAX = 8 * 256 * (size_t)(((BA_TILED_MAX_N + 15) / 16) * ((BA_TILED_MAX_N + 15) / 16 + 1) / 2);

// MODE 0: S = blockdiag(U*) - S_red and b = -g_p + b_red are formed here from the gather's buffer `rb`.  MODE 1: the system was
// assembled in global memory (win.Sg, lower triangle, and win.bvec: ba_big_assemble_kernel) and is copied in.  MODE 2: the 15-d system
// of an inertial window is assembled here, in the tiles, from `rb`, the IMU edge records and the bias random walk — what
// ba_inertial_assemble_kernel does in global memory with one 256-thread block (26 us at 10 keyframes) — same terms, same order per entry.
template <typename LAY, int MODE>
__device__ __forceinline__ void solve_body(int n, const BaState* St, double* P0, double* P1, const double* __restrict__ rb, int K, LAY S,
                           double* __restrict__ dp, double* __restrict__ res, const double* __restrict__ Sg = nullptr, const double* __restrict__ bvec = nullptr,
                           const BaInertialDev* inp = nullptr, const double* __restrict__ imu_buf = nullptr) {
  constexpr bool FROM_SG = MODE != 0;                                           // (the system does not come from the visual assembly below)
  if (St->done) return;
#ifdef ORBX_SOLVE_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(), st_acc_[16] = {0};
#endif
  const double lambda = St->lambda;
  const double* params = ba_cur(St, P0, P1);
  __shared__ __align__(16) double sb[BA_MAX_N];
  __shared__ double srinv[BA_MAX_N];
  __shared__ double s_red[BA_SOLVE_THREADS];
  const int tid = threadIdx.x, nth = BA_SOLVE_THREADS, lane = tid & 63;
  constexpr int NRW = LAY::TILED ? 3 : 2;                                      // 64-row register sets of the one-wave phases (n <= 64 NRW)
  __shared__ double s_zero;
  if (tid == 0) s_zero = 0.0;
  const bool small_n = n <= 64 * NRW;
  const size_t nn = (size_t)n * n;
  const double* U = rb + nn;                                                     // (FROM_SG: rb unused)
  const double* gp = U + 36 * (size_t)K;
  const double* bred = gp + n;
  if constexpr (!FROM_SG) {
  // S = blockdiag(U*) - S_red ; b = -g_p + b_red.  The factorisation reads and writes the lower triangle (j <= i) only; the upper
  // triangle is written as ZERO here and stays zero, which the backward substitution relies on.  Thread (ty, tx) of a 32 x 32 grid
  // takes the entries (ty + 32 a, tx + 32 b): all of its loads are issued before the first is used (the nested loops with their
  // run-time bounds waited for each global load in turn: 6 us of a 63 us solve at n = 114).
  {
    // (n <= 128 on this path: 160 KB of LDS less the static arrays, and n even.  Blocks of 32 x 32 entries right of the diagonal blocks
    // take their zeros without a load or a compare: with all 25 blocks of a 160 x 160 grid treated alike the 16 waves issued ~6 k
    // instructions' worth of cycles for 6.5 k live entries)
    constexpr int AS = 128 / 32;
    constexpr int TYN = BA_SOLVE_THREADS / 32, ASR = (128 + TYN - 1) / TYN, RPB = 32 / TYN;   // thread rows of the launch, row rounds, rounds per 32-row block
    static_assert(TYN <= 32 && 32 % TYN == 0, "a 32-entry block row is a whole number of thread-row rounds");
    const int ty = tid >> 5, tx = tid & 31;
    double rv_[ASR][AS], uv_[ASR][AS];
#pragma unroll
    for (int a = 0; a < ASR; ++a)
#pragma unroll
      for (int b2 = 0; b2 < AS; ++b2) {
        if (b2 > a / RPB) continue;                                            // (compile time) right of the diagonal block
        const int i = ty + TYN * a, j = tx + 32 * b2;
        const bool low = i < n && j <= i;
        rv_[a][b2] = low ? rb[(size_t)i * n + j] : 0.0;
        uv_[a][b2] = (low && i / 6 == j / 6) ? U[36 * (size_t)(i / 6) + (i % 6) * 6 + (j % 6)] : 0.0;
      }
#pragma unroll
    for (int a = 0; a < ASR; ++a)
#pragma unroll
      for (int b2 = 0; b2 < AS; ++b2) {
        const int i = ty + TYN * a, j = tx + 32 * b2;
        if (i < n && j < n) {
          double v = 0.0;
          if (b2 <= a / RPB && j <= i) {
            v = -rv_[a][b2];
            if (i / 6 == j / 6) {
              double u = uv_[a][b2];
              if (i == j) u += lambda * fmax(u, 1e-6);
              v += u;
            }
          }
          *S.at(i, j) = v;
        }
      }
  }
  // b and |g_p|^2: the n <= 128 entries sit in the first two waves — each sums its own, the barrier in front of the factorisation
  // publishes the two sums (a block-wide sum took two barriers of its own)
  double gs = 0.0;
  for (int i = tid; i < n; i += nth) { sb[i] = -gp[i] + bred[i]; gs += gp[i] * gp[i]; }
  if (small_n) { if (tid < 64 * NRW) { gs = wave_sum(gs); if ((tid & 63) == 0) s_red[tid >> 6] = gs; } }
  else { gs = block_sum_fixed(gs, s_red); if (tid == 0) { res[0] = bred[n]; res[1] = sqrt(gs + bred[n + 1]); } }
  } else if constexpr (MODE == 2) {
    static_assert(LAY::TILED, "the inertial assembly writes tiles");
    const BaInertialDev& in = *inp;
    const int Kk = in.K, n6 = 6 * Kk;
    const double* Uv = rb + (size_t)n6 * n6;
    const double* gpv = Uv + 36 * (size_t)Kk;
    const double* bredv = gpv + n6;
    const double* ex = params + 6 * (size_t)Kk + 3 * (size_t)in.M;
    double* gfull = s_red;                                                       // [n] gradient over the keyframe states (n <= 176 < nth)
    const int ntl = (n + 15) >> 4;
    for (int u = tid; u < ntl * (ntl + 1) / 2 * 256; u += nth) S.p[u] = 0.0;
    for (int i = tid; i < n; i += nth) { sb[i] = 0.0; gfull[i] = 0.0; }
    __syncthreads();
    // J^T J of the IMU and random-walk rows, edge after edge (fixed order, no atomics); lower triangle only
    for (int e = 0; e < in.E; ++e) {
      const int kk[2] = {in.edge_kf[2 * e], in.edge_kf[2 * e + 1]};
      const double* rec = imu_buf + (size_t)e * IMU_REC;
      if (tid < 18 * 18) {
        const int a = tid / 18, b2 = tid % 18;
        const int i = 15 * kk[a / 9] + a % 9, j = 15 * kk[b2 / 9] + b2 % 9;
        if (j <= i) *S.at(i, j) += rec[tid];
      }
      if (tid >= 512 && tid < 512 + 18) { const int t = tid - 512; gfull[15 * kk[t / 9] + t % 9] += rec[18 * 18 + t]; }
      if (tid >= 576 && tid < 576 + 6) {                                        // :863-880 (the bias rows 9..14 of a state: no entry of the 18 x 18 block above)
        const int k = tid - 576;
        const double w = k < 3 ? in.gw : in.aw;
        const int ii = 15 * kk[0] + 9 + k, jj = 15 * kk[1] + 9 + k;
        const double r = (ex[9 * (size_t)kk[1] + 3 + k] - ex[9 * (size_t)kk[0] + 3 + k]) * w;
        *S.at(ii, ii) += w * w; *S.at(jj, jj) += w * w;
        *S.at(max(ii, jj), min(ii, jj)) -= w * w;
        gfull[ii] += -w * r; gfull[jj] += w * r;
      }
      __syncthreads();
    }
    // visual part: U on the keyframe diagonal blocks, the gradient, damping with the full diagonal, then the Schur term
    for (int t = tid; t < 36 * Kk; t += nth) {
      const int k = t / 36, a = (t % 36) / 6, b2 = t % 6;
      if (b2 <= a) *S.at(15 * k + a, 15 * k + b2) += Uv[t];
    }
    for (int t = tid; t < n6; t += nth) gfull[15 * (t / 6) + t % 6] += gpv[t];
    __syncthreads();
    for (int i = tid; i < n; i += nth) { double* dgp = S.at(i, i); *dgp += lambda * fmax(*dgp, 1e-6); }
    __syncthreads();
    for (int idx = tid; idx < n6 * n6; idx += nth) {
      const int i = idx / n6, j = idx - i * n6;
      if (j <= i) *S.at(15 * (i / 6) + i % 6, 15 * (j / 6) + j % 6) -= rb[idx];
    }
    for (int i = tid; i < n; i += nth) {
      double b = -gfull[i];
      if (i % 15 < 6) b += bredv[6 * (i / 15) + i % 15];
      sb[i] = b;
    }
    // |gradient|^2 as ba_inertial_assemble_kernel forms it: 256 strided partials, a tree over them
    __shared__ double red256[256];
    if (tid < 256) { double gs2 = 0.0; for (int i = tid; i < n; i += 256) gs2 += gfull[i] * gfull[i]; red256[tid] = gs2; }
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) { if (tid < s2) red256[tid] += red256[tid + s2]; __syncthreads(); }
    if (tid == 0) {
      double chi = bredv[n6];                                                    // visual chi2 (gather kernel)
      for (int e = 0; e < in.E; ++e) chi += imu_buf[(size_t)e * IMU_REC + 18 * 18 + 18];
      res[0] = chi;
      res[1] = sqrt(red256[0] + bredv[n6 + 1]);                                  // |gradient| over keyframe states and points (:1213)
    }
  } else {
    // copy the lower triangle in (entries right of the diagonal inside the diagonal tiles: zeros), all of a thread's loads first
    constexpr int TYN = BA_SOLVE_THREADS / 32, CR = (BA_TILED_MAX_N + TYN - 1) / TYN, CC = (BA_TILED_MAX_N + 31) / 32;
    const int ty = tid >> 5, tx = tid & 31;
    double v_[CR][CC];
#pragma unroll
    for (int a = 0; a < CR; ++a)
#pragma unroll
      for (int b2 = 0; b2 < CC; ++b2) {
        const int i = ty + TYN * a, j = tx + 32 * b2;
        v_[a][b2] = (i < n && j <= i) ? Sg[(size_t)i * n + j] : 0.0;
      }
#pragma unroll
    for (int a = 0; a < CR; ++a)
#pragma unroll
      for (int b2 = 0; b2 < CC; ++b2) {
        const int i = ty + TYN * a, j = tx + 32 * b2;
        if (i < n && j < n && (!LAY::TILED || (j >> 4) <= (i >> 4))) *S.at(i, j) = v_[a][b2];
      }
    for (int i = tid; i < n; i += nth) sb[i] = bvec[i];
  }
  // Right-looking blocked Cholesky, panels of 16 columns, S in LDS — ONE phase and one block barrier per panel p (first column c0):
  //   - wave 0 brings the diagonal tile (c0, c0) up to date with the previous panel's rank-16 update, factors it (lane j owns row j in
  //     registers, DPP row_newbcast: chol16_rows_dpp) and PUBLISHES every column of L11 the moment its pivot step is done: the column
  //     into s_col, then 1 / L_tt into srinv, which starts as zeros and so doubles as the flag (LDS performs one wave's operations in
  //     issue order: no wait in between);
  //   - waves 1..15 apply the previous panel's update S22 -= L21 L21^T to the other lower tiles (v_mfma_f64_16x16x4_f64, one wave per
  //     tile) and b -= L21 y to the right-hand side below it, and count themselves on s_upd when their part is written;
  //   - the threads that own the rows below the panel — and the right-hand side as one more row: forward substitution L y = b rides
  //     along — wait until all fifteen have counted (their row's 16 entries may be any wave's tile), then solve x = a L11^-T column by
  //     column one step behind wave 0's pivots (flag and column read in one round trip, one step ahead; x_t *= 1/L_tt, x_jx -= x_t
  //     L[jx][t] for jx > t: per entry the same fma sequence in the same order as a row-oriented substitution, same bits).
  // The counter only grows (panel p waits for 15 p) and a flag is written once per solve, so nothing is reset between phases; a wave never waits before
  // it has counted itself and wave 0 never waits at all, and every wait is bounded all the same (a solve that ran into the bound
  // reports failure instead of hanging the queue).  Until the end of round 3: update phase (with wave 0's factor inside it as a
  // look-ahead) and row-solve phase, two barriers per panel, the 15 waves idle through most of the first and 13 through the second:
  // 9.0 k cycles per panel at n = 114.
  __shared__ int s_ok, s_upd;
  __shared__ __align__(16) double s_rv[16];
  __shared__ __align__(16) double s_col[16][16];                               // s_col[t][jx] = L[jx][t] of the block in flight
  SOLVE_STAMP(0);
  if (tid == 0) { s_ok = 1; s_upd = 0; }
  for (int i = tid; i < n; i += nth) srinv[i] = 0.0;                             // (0 = column not published yet)
  int ok = 1;
  constexpr int SPIN_MAX = 1 << 22;
  // the 16x16 diagonal block at c0_ (nb_ live columns), by wave 0 alone: reads and writes only that block of S, s_col, srinv, s_rv, s_ok
  auto factor_diag = [&](int c0_, int nb_, int step0) {
      const int j = tid & 15;
      double Lr[16];
      if (nb_ == 16) {
        // a full block: whole rows as eight ds_read_b128 — no masks, the entries right of the diagonal are the zeros the assembly wrote
        // (rows 912 bytes apart land on distinct banks; the wave's four 16-lane rows read the same addresses)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const double2_t p2 = *(const double2_t*)S.at(c0_ + j, c0_ + i);
          Lr[i] = p2[0]; Lr[i + 1] = p2[1];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) Lr[i] = (j < nb_ && i <= j) ? *S.at(c0_ + j, c0_ + i) : 0.0;   // (rows >= nb_ take no pivot step: chol16_rows_dpp; a 1.0 on their diagonal was 16 loop-invariant constants held in — and spilled from — registers)
      }
      SOLVE_STAMP(8);
      const int good = chol16_rows_dpp(Lr, nb_, tid, s_rv, [&](auto jj_, double col, double ri) {
        constexpr int jj = decltype(jj_)::value;
        if (tid < 16) {                                                            // (the wave's other three 16-lane rows hold copies)
          s_col[jj][j] = col;
          asm volatile("" ::: "memory");                                           // the column first; the LDS queue keeps the order
          // 1 / L_tt doubles as the "column t is there" flag: srinv starts as zeros and a reciprocal pivot is never zero (a bad pivot
          // leaves a NaN, which also compares unequal to zero) — two ds_writes per pivot step instead of four with a counter and s_rv
          __hip_atomic_store(&srinv[c0_ + jj], ri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      });
      SOLVE_STAMP(9);
      if (tid == 0 && !good) s_ok = 0;
      if (good && nb_ == 16) {
        // (the registers right of the diagonal hold the mirror entries' updates by now: zeros go back, the backward substitution counts on
        // them — and on a zero ON the diagonal: L_cc itself is never read again, 1 / L_cc is in srinv)
        if (tid < 16) {
#pragma unroll
          for (int i = 0; i < 16; i += 2)
            *(double2_t*)S.at(c0_ + j, c0_ + i) = double2_t{i < j ? Lr[i] : 0.0, i + 1 < j ? Lr[i + 1] : 0.0};
        }
      } else if (good && tid < nb_) {
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i <= j) *S.at(c0_ + j, c0_ + i) = i < j ? Lr[i] : 0.0;
      }
  };
  auto wait_for = [&](int* ctr, int want, int& seen, int& spins) {                // until *ctr >= want (what was read last is kept in `seen`)
    while (seen < want && spins < SPIN_MAX) {
      seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (seen < want) { __builtin_amdgcn_s_sleep(1); ++spins; }
    }
    asm volatile("" ::: "memory");                                                 // what the counter vouches for is read after it, not before
  };
  __syncthreads();
  if constexpr (!FROM_SG) {
    if (small_n && tid == nth - 1) {
      double g2 = s_red[0] + s_red[1];
      if (NRW > 2) g2 += s_red[2];
      res[0] = bred[n]; res[1] = sqrt(g2 + bred[n + 1]);
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);                       // the wave index as a scalar: the roles below branch on SGPRs
  for (int c0 = 0, p = 0; c0 < n; c0 += 16, ++p) {
    const int nb = min(16, n - c0), prev = c0 - 16;
    // ---- the previous panel's update of the trailing matrix [c0, n) and of the right-hand side below it
    if (prev >= 0) {
      const int m = n - c0;
      if (tid >= nth - 192 && tid - (nth - 192) < m) {                         // b_below -= L21 y_panel (the last three waves: not in wave 0's way)
        const int rr = c0 + tid - (nth - 192);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fma(*S.at(rr, prev + k), sb[prev + k], acc);
        sb[rr] -= acc;
      }
      const int nt = (m + 15) / 16, units = nt * (nt + 1) / 2, nw = (nth >> 6) - 1;
      // wave 0: tile (c0, c0) only — it factors that block next; waves 1..15: the other lower tiles  (keeping them off waves 4, 8 and 12,
      // which share wave 0's SIMD and its f64 datapath — what the one-launch global factorisation needs — measured 32.4 -> 32.8 us at
      // n = 114 and 80.6 -> 82.6 at n = 162: a tile here is four MFMAs, the extra round costs more than the contention)
      for (int unit = wv; unit < units; unit += wv == 0 ? units : nw) {        // wave-uniform, and a scalar to the compiler
        int ti = 0, rem = unit;
        while (rem > ti) { rem -= ti + 1; ++ti; }                              // unit = ti (ti + 1) / 2 + tj, tj <= ti
        const int tj = rem;
        const int ra = min(c0 + 16 * ti + (lane & 15), n - 1), rb2 = min(c0 + 16 * tj + (lane & 15), n - 1);
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
        double av[4], bv[4], old_[4];
        const int col = c0 + 16 * tj + (lane & 15);
        // (a diagonal tile's two operands are the same rows: read once; the tile's current values are requested with the operands, not
        // after the matrix instructions: they are on wave 0's chain to the next factorisation)
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *S.at(ra, prev + (lane >> 4) + 4 * q);
        if (ti != tj) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = *S.at(rb2, prev + (lane >> 4) + 4 * q);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[q] = av[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = c0 + 16 * ti + (lane >> 4) + 4 * q;
          old_[q] = (row < n && col <= row) ? *S.at(row, col) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = c0 + 16 * ti + (lane >> 4) + 4 * q;
          if (row < n && col <= row) *S.at(row, col) = old_[q] - acc[q];
        }
      }
      if (wv != 0 && lane == 0) {                                                // this wave's part of the update is written
        asm volatile("" ::: "memory");
        __hip_atomic_fetch_add(&s_upd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    SOLVE_STAMP(7);
    // ---- the panel: factor (wave 0) and, a pivot step behind it, the rows below
    if (wv == 0) { factor_diag(c0, nb, 16 * p); SOLVE_STAMP(10); }
    else if (nb == 16) {
      // rows below the panel (a panel with rows below it is a full one), the right-hand side as row n; both layouts keep a row's 16
      // entries of the panel contiguous and 16-byte aligned (n = 6K even and c0 a multiple of 16 / tile rows)
      const int r = c0 + 16 + tid - 64;
      if (r <= n) {
        int seen_u = 0, spins = 0;
        if (prev >= 0) wait_for(&s_upd, ((nth >> 6) - 1) * p, seen_u, spins);
        double* row = r < n ? S.at(r, c0) : &sb[c0];
        double x[16];
#pragma unroll
        for (int jx = 0; jx < 16; jx += 2) {
          const double2_t p2 = *(const double2_t*)&row[jx];
          x[jx] = p2[0]; x[jx + 1] = p2[1];
        }
        // column t = its flag (1 / L_tt, zero until published) and L[jx][t], jx > t (pairs from the even index at or below t + 1), read
        // in ONE round trip: the flag first, the entries right behind it without waiting — LDS performs a wave's reads in issue order, so
        // a non-zero flag vouches for the entries read after it — and one step AHEAD of the arithmetic (two dependent round trips per
        // step, poll then column, made the rows slower than the factorisation they follow)
        auto fetch = [&](auto t_, double& rt, double (&lc)[16]) {
          constexpr int t = decltype(t_)::value;
          rt = __hip_atomic_load(&srinv[c0 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int jx = (t + 1) & ~1; jx < 16; jx += 2) {
            const double2_t p2 = *(const double2_t*)&s_col[t][jx];
            lc[jx] = p2[0]; lc[jx + 1] = p2[1];
          }
          asm volatile("" ::: "memory");
        };
        double rta = 0.0, rtb = 0.0, lca[16], lcb[16];
        auto step = [&](auto t_, double& rt, double (&lc)[16], double& rtn, double (&lcn)[16]) {
          constexpr int t = decltype(t_)::value;
          while (rt == 0.0 && spins < SPIN_MAX) { __builtin_amdgcn_s_sleep(1); ++spins; fetch(t_, rt, lc); }   // not there yet when it was read: again
          if constexpr (t + 1 < 16) fetch(std::integral_constant<int, t + 1>{}, rtn, lcn);
          x[t] = x[t] * rt;
#pragma unroll
          for (int jx = t + 1; jx < 16; ++jx) x[jx] = fma(-x[t], lc[jx], x[jx]);
          // the step ends here: without this the compiler polls for all 16 columns first, parks every column in scratch and does the
          // arithmetic afterwards
          asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                            "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) :: "memory");
        };
        fetch(std::integral_constant<int, 0>{}, rta, lca);
        static_for<8>([&](auto h_) {
          constexpr int t = 2 * decltype(h_)::value;
          step(std::integral_constant<int, t>{}, rta, lca, rtb, lcb);
          step(std::integral_constant<int, t + 1>{}, rtb, lcb, rta, lca);
        });
        if (spins >= SPIN_MAX) s_ok = 0;                                           // never seen: fail the solve rather than trust x
#pragma unroll
        for (int jx = 0; jx < 16; jx += 2) *(double2_t*)&row[jx] = double2_t{x[jx], x[jx + 1]};
      }
    }
    __syncthreads();
    SOLVE_STAMP(2);
    ok = s_ok;
    if (!ok) break;
    if (nb < 16 && tid == nth - 1) {
      // the short last panel has no rows below it: only the right-hand side, entry by entry (nothing after it in this loop reads sb)
      double x[16];
#pragma unroll
      for (int jx = 0; jx < 16; ++jx) x[jx] = (jx < nb) ? sb[c0 + jx] : 0.0;
#pragma unroll
      for (int jx = 0; jx < 16; ++jx) {
        if (jx < nb) {
          double v = x[jx];
#pragma unroll
          for (int t = 0; t < jx; ++t) v = fma(-x[t], *S.at(c0 + jx, c0 + t), v);
          x[jx] = v * srinv[c0 + jx];
        }
      }
#pragma unroll
      for (int jx = 0; jx < 16; ++jx) if (jx < nb) sb[c0 + jx] = x[jx];
    }
  }
  if (n & 15) __syncthreads();                                                    // (uniform) the short last panel's right-hand side was solved behind the loop's barrier
  SOLVE_STAMP(4);
  // (A panel-blocked backward substitution — the panel's 16 unknowns by DPP row_newbcast steps in wave 0, the rows above it one
  // thread each, same operations in the same order, bit-identical — was built and measured: 15.0 us against the 12.2 us of the
  // single wave below at n = 114; its 16 block barriers and LDS round trips cost more than the readlane pairs they replace.
  // v_fmac_f64_dpp / v_mov_b64_dpp issue every 16 cycles, a v_readlane_b32 every 4: profiles/r03_valu_issue_probe2.txt.)
  if (ok && tid < 64 && small_n) {
    // backward L^T x = y (y is already in sb: the forward substitution ran inside the factorisation), column oriented: lane holds
    // rows `lane`, `lane + 64` (and `lane + 128` in the tiled layout) of the right-hand side in registers; step c: x_c = b_c / L_cc
    // reaches all lanes by a v_readlane pair, b_r -= L_cr x_c for r < c.  ONE wave issues an instruction every ~8.4 cycles whatever the
    // instruction (profiles/r03_valu_issue_probe2.txt, 1 wave/SIMD), so what counts here is the instruction count of a step — address
    // add, ds_read, multiply, two readlanes, fma:
    //   - rows r >= c need no mask: they read the upper triangle and the diagonal of S, which hold zeros (above; in the tiled layout a
    //     shared zero for tiles that do not exist), so the fma leaves them alone — lane c keeps b_c and is scaled once at the end (the
    //     same product that was broadcast);
    //   - 1 / L_cc is not fetched per step: every lane multiplies its own entry by its own 1 / L_rr and the readlane picks lane c's;
    //   - the columns of each 64-row set run as a loop of their own (the pivot sits in that set's register, only the sets up to it are
    //     touched), chunks of columns fetched one chunk ahead, registers alternating (no copies); what does not fill a pair of chunks
    //     goes first, step by step.
    // Until round 3 a step was 22 instructions with two exec-mask regions and a uniform branch: 11.8 us of a 68.7 us solve at n = 114.
    // Same operations on the same operands in the same order: same bits.
    int rr_[NRW];
    double bx[NRW], rix[NRW];
#pragma unroll
    for (int q = 0; q < NRW; ++q) {
      rr_[q] = min(lane + 64 * q, n - 1);
      bx[q] = lane + 64 * q < n ? sb[lane + 64 * q] : 0.0;
      rix[q] = srinv[rr_[q]];
    }
    auto bcast = [&](double t, int src) -> double {
      const int lo = __builtin_amdgcn_readlane(__double2loint(t), src), hi = __builtin_amdgcn_readlane(__double2hiint(t), src);
      return __hiloint2double(hi, lo);
    };
    auto run = [&](auto p_, auto ch_) {                                         // columns of register set P, downwards
      constexpr int P = decltype(p_)::value, CH = decltype(ch_)::value;
      const int c_hi = min(n, 64 * P + 64) - 1, c_lo = 64 * P;
      if (c_hi < c_lo) return;
      double la[P + 1][CH], lb[P + 1][CH];
      auto fetch = [&](int ctop, double (&l)[P + 1][CH]) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int cc = max(ctop - j, 0);                                      // (the prefetch past the last chunk: any row will do)
#pragma unroll
          for (int q = 0; q <= P; ++q) l[q][j] = *S.rd(cc, rr_[q], &s_zero);
        }
      };
      auto chunk = [&](int ctop, const double (&l)[P + 1][CH]) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const double xc = bcast(bx[P] * rix[P], (ctop - j) & 63);              // column c lives in lane c mod 64
#pragma unroll
          for (int q = 0; q <= P; ++q) bx[q] = fma(-l[q][j], xc, bx[q]);
        }
      };
      const int cnt = c_hi - c_lo + 1, pairs = cnt / (2 * CH);
      int c = c_hi;
      for (; c >= c_lo + pairs * 2 * CH; --c) {                                   // the odd part first, one column at a time
        const double xc = bcast(bx[P] * rix[P], c & 63);
#pragma unroll
        for (int q = 0; q <= P; ++q) bx[q] = fma(-*S.rd(c, rr_[q], &s_zero), xc, bx[q]);
      }
      if (pairs > 0) fetch(c, la);
      for (int q2 = 0; q2 < pairs; ++q2, c -= 2 * CH) {
        fetch(c - CH, lb);
        chunk(c, la);
        fetch(c - 2 * CH, la);
        chunk(c - CH, lb);
      }
    };
    if constexpr (NRW > 2) run(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    run(std::integral_constant<int, 1>{}, std::integral_constant<int, 8>{});
    run(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
    // the solution is in this wave's registers: delta_p, |delta_p|^2 and |p|^2 from here, no block-wide pass behind it
    double d2 = 0.0, p2s = 0.0;
#pragma unroll
    for (int q = 0; q < NRW; ++q) {
      const bool in = lane + 64 * q < n;
      const double xq = in ? bx[q] * rix[q] : 0.0, pq = in ? params[lane + 64 * q] : 0.0;
      if (in) dp[lane + 64 * q] = xq;
      if (q == 0) { d2 = xq * xq; p2s = pq * pq; } else { d2 = d2 + xq * xq; p2s = p2s + pq * pq; }
    }
    const double dsq = wave_sum(d2), psq = wave_sum(p2s);
    if (lane == 0) { res[2] = 1.0; res[3] = dsq; res[4] = psq; }
  } else if (ok && tid < 64) {
    // general n: row dot products with shuffle reductions
    for (int r = n - 1; r >= 0; --r) {
      double acc = 0.0;
      for (int k = r + 1 + lane; k < n; k += 64) acc = fma(*S.at(k, r), sb[k], acc);
      acc = wave_sum(acc);
      if (lane == 0) sb[r] = (sb[r] - acc) * srinv[r];
      __builtin_amdgcn_wave_barrier();
    }
  }
  SOLVE_STAMP(5);
  if (ok && small_n) {                                                          // (uniform) wave 0 has written everything
    SOLVE_STAMP_FLUSH();
    return;
  }
  __syncthreads();
  double dsq = 0.0, psq = 0.0;
  for (int i = tid; i < n; i += nth) {
    const double v = ok ? sb[i] : 0.0;
    dp[i] = v;
    dsq += v * v;
    psq += params[i] * params[i];
  }
  dsq = block_sum_fixed(dsq, s_red);
  psq = block_sum_fixed(psq, s_red);
  if (tid == 0) { res[2] = (double)ok; res[3] = dsq; res[4] = psq; }
  SOLVE_STAMP(6);
  SOLVE_STAMP_FLUSH();
}

__global__ __launch_bounds__(BA_SOLVE_THREADS) void ba_solve_lds_kernel(const BaWin* __restrict__ wins) {
  extern __shared__ __align__(16) double dyn[];
  const BaWin win = ba_win_global(wins, blockIdx.y);       // one workgroup per window
  if (win.use_lds != 1) return;
  solve_body<LaySquare, 0>(win.n, win.S, win.P0, win.P1, win.rb, win.d.K, LaySquare{dyn, win.n}, win.dp, win.res);
}

// The same factorisation and solve for systems of up to BA_TILED_MAX_N unknowns that do not fit LDS as a square (use_lds == 2: a visual
// window of 22..29 keyframes, the 15-d system of an inertial window of up to 11): assembled in global memory by ba_big_assemble_kernel /
// ba_inertial_assemble_kernel as for the multi-launch path, copied into lower 16 x 16 tiles, ONE launch instead of n / 16 + 1
// (inertial BA, 10 keyframes: 11 launches, 129 us per iteration for the solve).
__global__ __launch_bounds__(BA_SOLVE_THREADS) void ba_solve_tiled_kernel(const BaWin* __restrict__ wins) {
  extern __shared__ __align__(16) double dyn[];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  if (win.use_lds != 2) return;
  solve_body<LayTiled, 1>(win.n, win.S, win.P0, win.P1, win.rb, win.d.K, LayTiled{dyn}, win.dp, win.res, win.Sg, win.bvec);
}

// the 15-d system of an inertial window (one window per call), assembled in the tiles and solved in the same launch
__global__ __launch_bounds__(BA_SOLVE_THREADS) void ba_solve_inertial_tiled_kernel(const BaWin* __restrict__ wins, BaInertialDev in,
                                                                                   const double* __restrict__ imu_buf) {
  extern __shared__ __align__(16) double dyn[];
  const BaWin win = ba_win_global(wins, 0);
  solve_body<LayTiled, 2>(win.n, win.S, win.P0, win.P1, win.rb, in.K, LayTiled{dyn}, win.dp, win.res, nullptr, nullptr, &in, imu_buf);
}

// ---- large reduced systems (n > ~135: S does not fit LDS) -------------------------------------------------------
// Right-looking blocked Cholesky over several launches, panels of 16 columns, S in global memory (L2 resident):
//   ba_big_assemble_kernel   S = blockdiag(U*) - S_red, b, |g|                                   (many blocks)
//   per panel:  ba_big_step_kernel    trailing update of the previous panel (one wave per lower 16x16 tile, four
//                                     v_mfma_f64_16x16x4_f64) + this panel: diagonal block, rows below it, its part of L y = b
//   ba_big_back_kernel       panel-blocked backward substitution, |dp|^2, |p|^2                   (1 block)
constexpr int BB_NB = 16;

__global__ __launch_bounds__(256) void ba_big_assemble_kernel(const BaWin* __restrict__ wins) {
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaState* St = win.S;
  if (win.use_lds == 1 || St->done) return;                                // (the tiled LDS solve, use_lds == 2, reads what this kernel assembles)
  const int n = win.n, K = win.d.K;
  const double* __restrict__ rb = win.rb;
  double* __restrict__ Sg = win.Sg; double* __restrict__ bvec = win.bvec; double* __restrict__ res = win.res;
  const double lambda = St->lambda;
  const size_t nn = (size_t)n * n;
  const double* U = rb + nn;
  const double* gp = U + 36 * (size_t)K;
  const double* bred = gp + n;
  const int gt = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (size_t idx = gt; idx < nn; idx += nth) {
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    double v = -rb[idx];
    if (i / 6 == j / 6) {
      double u = U[36 * (size_t)(i / 6) + (i % 6) * 6 + (j % 6)];
      if (i == j) u += lambda * fmax(u, 1e-6);
      v += u;
    }
    Sg[idx] = v;
  }
  for (int i = gt; i < n; i += nth) bvec[i] = -gp[i] + bred[i];
  if (blockIdx.x == 0) {
    __shared__ double sh[256];
    double gs = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) gs += gp[i] * gp[i];
    sh[threadIdx.x] = gs;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) { if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2]; __syncthreads(); }
    if (threadIdx.x == 0) { res[0] = bred[n]; res[1] = sqrt(sh[0] + bred[n + 1]); res[2] = 1.0; }
  }
}

// One launch per panel (c0 = first column of the panel).  Block 0 = the panel block: it first brings the panel's own column block
// up to date with the rank-16 update of the PREVIOUS panel (the tiles (ti, 0) of the trailing matrix) and the right-hand side below
// it (forward substitution rides along: b_r -= L[r][prev..] y_prev), then factors the 16x16 diagonal block (wave 0, DPP), solves
// the rows below against it and the panel's 16 entries of the right-hand side.  The other blocks apply the previous panel's update
// to the rest of the trailing matrix, one wave per lower 16x16 tile (ti, tj), tj >= 1.  Both read only the previous panel's
// columns, which nobody writes in this launch, and write disjoint tiles.  (Round 2: a panel kernel, an update kernel per panel and
// forward AND backward substitution in a one-block kernel afterwards — 38 + 1 dependent launches at n = 294, 381 us per solve; now
// 19 + 1.)
// (1024-thread blocks — the panel block's column tiles in two rounds, every row below in one — measured slower: 327 -> 438 us per solve.)
#ifndef ORBX_BB_STEP_THREADS
#define ORBX_BB_STEP_THREADS 512
#endif
constexpr int BB_STEP_THREADS = ORBX_BB_STEP_THREADS, BB_STEP_WAVES = BB_STEP_THREADS / 64;   // 256 / 512 threads: 211.8 / 204.9 us per solve at n = 294 (round 1's 1024: slower)
static_assert(BB_STEP_THREADS >= 256 && BB_STEP_THREADS % 64 == 0 && BB_STEP_THREADS <= 1024, "the panel block stages its 16 x 16 tile with 256 threads");
constexpr int BB_COL_TILES = (BA_MAX_N / 16 + BB_STEP_WAVES - 1) / BB_STEP_WAVES;      // column tiles per wave of the panel block, all in flight at once
__global__ __launch_bounds__(BB_STEP_THREADS) void ba_big_step_kernel(const BaWin* __restrict__ wins, int c0, int one_launch_max_n) {
  __shared__ __align__(16) double D[BB_NB][BB_NB + 2];                    // (even pitch: the row solves read L11 in pairs)
  __shared__ __align__(16) double rinv[BB_NB];
  __shared__ double ys[BB_NB];
  __shared__ int s_ok;
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int n = win.n;
  double* __restrict__ Sg = win.Sg; double* __restrict__ ginv = win.ginv; double* __restrict__ bvec = win.bvec; double* __restrict__ res = win.res;
  if (win.use_lds || n <= one_launch_max_n || c0 >= n || win.S->done || res[2] == 0.0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prev = c0 - BB_NB;                                            // the finished panel whose update is applied here (none at c0 = 0)
  const int nt = (n - c0 + 15) / 16;                                      // tile rows of the trailing matrix [c0, n)
  if (blockIdx.x > 0) {
    if (prev < 0) return;
    const int unit = ((int)blockIdx.x - 1) * BB_STEP_WAVES + wave;                     // tiles (ti, tj) with 1 <= tj <= ti < nt: unit = ti (ti - 1) / 2 + tj - 1
    if (unit >= nt * (nt - 1) / 2) return;
    int ti = 1, rem = unit;
    while (rem >= ti) { rem -= ti; ++ti; }
    const int tj = rem + 1;
    // S[ti][tj] -= L[ti][prev] L[tj][prev]^T, lower part, one wave
    const int ra = min(c0 + 16 * ti + (lane & 15), n - 1), rb_ = min(c0 + 16 * tj + (lane & 15), n - 1);
    const double* pa = Sg + (size_t)ra * n + prev + (lane >> 4);
    const double* pb = Sg + (size_t)rb_ * n + prev + (lane >> 4);
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    double av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { av[q] = pa[4 * q]; bv[q] = pb[4 * q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
    const int col = c0 + 16 * tj + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = c0 + 16 * ti + (lane >> 4) + 4 * q;
      if (row < n && col <= row) Sg[(size_t)row * n + col] -= acc[q];
    }
    return;
  }
  // ---- the panel block
  const int nb = min(BB_NB, n - c0);
  if (tid == 0) s_ok = 1;
  if (tid < BB_NB * BB_NB) {
    const int i = tid / BB_NB, j = tid % BB_NB;
    D[i][j] = i == j ? 1.0 : 0.0;                                           // rows / columns beyond nb: the identity
  }
  if (prev >= 0) {
    // Column tiles (ti, 0), ti = wave, wave + 4, ...: EVERY operand of all of a wave's tiles (the shared tile-0 rows of the previous
    // panel, each tile's own rows, each tile's current values) and the right-hand side's rows are requested before the first is
    // used — one L2 round trip for the whole phase instead of two per tile, five tiles deep.
    if (tid < BB_NB) ys[tid] = bvec[prev + tid];
    const int rb_ = min(c0 + (lane & 15), n - 1);
    double bv[4], av[BB_COL_TILES][4], cv[BB_COL_TILES][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = Sg[(size_t)rb_ * n + prev + (lane >> 4) + 4 * q];
    const int col = c0 + (lane & 15);
#pragma unroll
    for (int t = 0; t < BB_COL_TILES; ++t) {
      const int ti = wave + BB_STEP_WAVES * t;
      if (ti < nt) {                                                       // wave-uniform
        const int ra = min(c0 + 16 * ti + (lane & 15), n - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[t][q] = Sg[(size_t)ra * n + prev + (lane >> 4) + 4 * q];
          const int row = c0 + 16 * ti + (lane >> 4) + 4 * q;
          cv[t][q] = (row < n && col <= row) ? Sg[(size_t)row * n + col] : 0.0;
        }
      }
    }
    __syncthreads();                                                       // ys
    // right-hand side below the previous panel: b_r -= L[r][prev ..] y_prev
    for (int r = c0 + tid; r < n; r += BB_STEP_THREADS) {
      const double* row = Sg + (size_t)r * n + prev;
      double v = bvec[r];
#pragma unroll
      for (int j = 0; j < BB_NB; ++j) v = fma(-row[j], ys[j], v);
      bvec[r] = v;
    }
#pragma unroll
    for (int t = 0; t < BB_COL_TILES; ++t) {
      const int ti = wave + BB_STEP_WAVES * t;
      if (ti < nt) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t][q], bv[q], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = c0 + 16 * ti + (lane >> 4) + 4 * q;
          if (row < n && col <= row) {
            const double v = cv[t][q] - acc[q];
            if (ti == 0) D[(lane >> 4) + 4 * q][lane & 15] = v;            // the diagonal tile goes to the factor through LDS
            else Sg[(size_t)row * n + col] = v;
          }
        }
      }
    }
  } else if (tid < BB_NB * BB_NB) {
    const int i = tid / BB_NB, j = tid % BB_NB;
    if (i < nb && j <= i) D[i][j] = Sg[(size_t)(c0 + i) * n + c0 + j];
  }
  __syncthreads();                                                         // the column block and the right-hand side are up to date (block scope)
  // rows below the block: their 16 entries are final now; request them before the factor so that they travel under it.  Row n = the
  // right-hand side (its 16 entries of this panel): one more row of the same solve (a full panel; the short last one has no rows below
  // and solves its right-hand side on its own further down)
  const int r0 = c0 + BB_NB + tid;
  const bool rows_here = nb == BB_NB;
  double xr[BB_NB];
  if (rows_here && r0 <= n) {
    const double* src = r0 < n ? Sg + (size_t)r0 * n + c0 : bvec + c0;
#pragma unroll
    for (int j = 0; j < BB_NB; ++j) xr[j] = src[j];
  }
  if (tid < 64) {
    const int j = tid & 15;
    double Lr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) Lr[i] = (j < nb && i <= j) ? D[j][i] : (i == j ? 1.0 : 0.0);
    const int good = chol16_rows_dpp(Lr, nb, tid, rinv);
    if (tid == 0 && !good) s_ok = 0;
    if (good && tid < BB_NB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        D[j][i] = i <= j ? Lr[i] : 0.0;
        if (j < nb && i <= j) Sg[(size_t)(c0 + j) * n + c0 + i] = Lr[i];
      }
    }
  }
  __syncthreads();
  if (!s_ok) { if (tid == 0) res[2] = 0.0; return; }
  if (tid < nb) ginv[c0 + tid] = rinv[tid];
  if (!rows_here) {
    if (tid == BB_STEP_THREADS - 1) {                                      // the short last panel: y of this panel, L11 y = b, entry by entry
      double x[BB_NB];
#pragma unroll
      for (int jx = 0; jx < BB_NB; ++jx) x[jx] = jx < nb ? bvec[c0 + jx] : 0.0;
#pragma unroll
      for (int jx = 0; jx < BB_NB; ++jx) {
        if (jx < nb) {
          double v = x[jx];
#pragma unroll
          for (int t = 0; t < jx; ++t) v = fma(-D[jx][t], x[t], v);
          x[jx] = v * rinv[jx];
        }
      }
#pragma unroll
      for (int jx = 0; jx < BB_NB; ++jx) if (jx < nb) bvec[c0 + jx] = x[jx];
    }
    return;
  }
  // rows below the block (and the right-hand side): x = a L11^-T, column oriented as in ba_solve_lds_kernel — once x_t is final every
  // later entry takes its term, per entry the same fma sequence as a row-oriented substitution; L11 in pairs from LDS
  for (int r = r0; r <= n; r += BB_STEP_THREADS) {
    double* row = r < n ? Sg + (size_t)r * n + c0 : bvec + c0;
    if (r != r0) {
#pragma unroll
      for (int j = 0; j < BB_NB; ++j) xr[j] = row[j];
    }
#pragma unroll
    for (int t = 0; t < BB_NB; t += 2) {
      double la[BB_NB], lb[BB_NB];                                           // L[jx][t], L[jx][t + 1] for jx > t
#pragma unroll
      for (int jx = t + 1; jx < BB_NB; ++jx) {
        const double2_t p2 = *(const double2_t*)&D[jx][t];
        la[jx] = p2[0]; lb[jx] = p2[1];
      }
      const double2_t q2 = *(const double2_t*)&rinv[t];
      xr[t] = xr[t] * q2[0];
#pragma unroll
      for (int jx = t + 1; jx < BB_NB; ++jx) xr[jx] = fma(-xr[t], la[jx], xr[jx]);
      xr[t + 1] = xr[t + 1] * q2[1];
#pragma unroll
      for (int jx = t + 2; jx < BB_NB; ++jx) xr[jx] = fma(-xr[t + 1], lb[jx], xr[jx]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < BB_NB; ++j) row[j] = xr[j];
  }
}

// The same factorisation in ONE launch for systems of up to BF_MAX_N unknowns (configs[4]: n = 294): one workgroup of 16 waves per window,
// LEFT-looking with one panel of look-ahead.  Panel p (columns c0 .. c0 + 15) needs S[c0.., panel] - L[c0.., 0..c0) L[panel, 0..c0)^T, one
// wave per 16-row tile on the f64 MFMA.  All of that sum but its last 16 columns is formed one panel EARLIER, while wave 0 factors the
// previous diagonal block: the tile's rows of L straight from L2 (128 contiguous bytes per row and 16 columns, three chunks of loads in
// flight), the panel's own rows as the B operand from an LDS copy (Bp, filled another phase earlier by the waves that have no row to
// solve — 15 waves fetching the same 16 rows was half of the CU's L1 bandwidth), the columns of the panel just before from the LDS copy
// of that panel (Lp).  The last 16 columns — what the row solves have just produced — come from Lp as well, so the dependent chain of a
// panel is four MFMAs, the 16 x 16 factor (wave 0, DPP), the row solves (one thread per row, the arithmetic of ba_big_step_kernel's panel
// block, the right-hand side as row n) and three barriers, with no global-memory round trip in it.  The tiles belong to the twelve waves
// that do not share wave 0's SIMD: f64 MFMA and f64 VALU use one datapath, and the factor is the chain everything waits for.  Every entry
// of L is written once and only read afterwards, by the workgroup that wrote it.  Larger systems keep the multi-launch path: their
// trailing updates want more than one CU.
constexpr int BF_THREADS = 1024, BF_MAX_N = 320;
constexpr int BF_CPITCH = 356;                                            // rows per column of the updated panel (>= BF_MAX_N + 16; = 4 mod 32)
constexpr int BF_LPITCH = 18;                                             // doubles per row of the LDS copy of a panel of L (144 bytes: 16 rows x 16 bytes fill the banks once)
constexpr int BF_LROWS = BF_MAX_N + 16;                                   // rows below a panel, the right-hand side and the clamped rows of the last tile
constexpr int BF_BPITCH = 290;                                            // doubles per row of the B operand (>= BF_MAX_N - 32; = 2 mod 32)
constexpr int BF_BQ = (BF_MAX_N - 2 * 16 + 127) / 128;                         // slices of 64 column pairs in a row of Bp
constexpr int BF_TILE_WAVES = 12, BF_TILES = 2;                           // tile ti of a panel: slot ti / 12 of the ti % 12-th wave that is not on wave 0's SIMD
// (Round 4: the four row waves, all on wave 0's SIMD, take 4-5 k cycles per panel for what is one wave's 1.2 k — so the rows were spread over
// waves 0-3, one per SIMD, nine tile waves with three slots each on SIMDs 1-3, waves 4 / 8 / 12 staging Bp only: 128.8-129.4 -> 135.4-135.8 us
// per solve at n = 294.  The stamps say why: the phase ends when the tile waves' look-ahead does — 9 waves x 3 tiles need longer than
// 12 x 2 — not when the rows are solved.  Withdrawn; profiles/r04_ba_big_factor_ab.txt.)
static_assert((BF_MAX_N + 1 + 15) / 16 <= BF_TILES * BF_TILE_WAVES, "every tile of a panel needs an owner");
static_assert(BF_MAX_N + 1 - 16 <= 2 * 256, "the rows below a panel: at most two per thread of the four row waves");
constexpr size_t BF_LDS_BYTES = (size_t)(16 * BF_CPITCH + BF_LROWS * BF_LPITCH + 16 * BF_BPITCH) * 8;
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));  // (rows of an odd-n system start 8 bytes off)
// Loads of L and S in ba_big_factor_kernel, as GLOBAL loads: the row pointers come out of a select between two members of BaWin and were
// generic to the compiler — flat loads, which count on lgkmcnt as well, so that every wait for an LDS read also waited for the prefetched
// chunks of L.
typedef __attribute__((address_space(1))) const double2_u bf_g2_t;
typedef __attribute__((address_space(1))) const double bf_g1_t;
__device__ __forceinline__ double2_t bf_ldg2(const double* p) { return *(bf_g2_t*)p; }
__device__ __forceinline__ double bf_ldg1(const double* p) { return *(bf_g1_t*)p; }
#ifdef ORBX_BF_DEBUG
__device__ unsigned long long g_bf_stamps[8];
#define BF_STAMP(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); bf_acc_[k] += now_ - bf_prev_; bf_prev_ = now_; } } while (0)
#else
#define BF_STAMP(k) do { } while (0)
#endif
__global__ __launch_bounds__(BF_THREADS) void ba_big_factor_kernel(const BaWin* __restrict__ wins) {
  extern __shared__ __align__(16) double dyn[];
  __shared__ __align__(16) double D[BB_NB][BB_NB + 2];
  __shared__ __align__(16) double rinv[BB_NB];
  __shared__ int s_ok;
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const int n = win.n;
  if (win.use_lds || n > BF_MAX_N || win.S->done) return;
  double* Sg = win.Sg; double* ginv = win.ginv; double* bvec = win.bvec; double* res = win.res;
  double* Linv = ginv + 2 * ((n + 15) & ~15);                               // [panel][16][16] rows of the diagonal blocks' L11^-T (behind 1/L_jj and the inertial system's gradient)
  double* Cs = dyn;
  double* Lp = Cs + 16 * BF_CPITCH;
  double* Bp = Lp + BF_LROWS * BF_LPITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, g = lane >> 4;   // (the wave index as a scalar — the roles and tiles derived from it branch on SGPRs: 138.1 -> 136.0 us)
  const bool row_role = (wave & 3) == 0;
  const int widx = (wave & 3) ? wave - 1 - (wave >> 2) : 1000;              // 0 .. 11 among the tile waves
  if (tid == 0) s_ok = 1;
  auto rowp = [&](int r) -> double* { return r < n ? Sg + (size_t)r * n : bvec; };    // row n (and the clamped rows past it): the right-hand side
  auto mfma4 = [](double4_t a, double2_t a0, double2_t a1, double2_t b0, double2_t b1) {
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], a, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], a, 0, 0, 0);
  };
  // 16 columns of a tile's sum from the LDS copy of the panel that holds them; lrow0: the tile's first row, brow0: the B rows' first, as Lp numbers them
  auto lp_chunk = [&](double4_t a, int lrow0, int brow0) {
    const double* ap = Lp + (lrow0 + m) * BF_LPITCH + 4 * g;
    const double* bp = Lp + (brow0 + m) * BF_LPITCH + 4 * g;
    return mfma4(a, *(const double2_t*)ap, *(const double2_t*)(ap + 2), *(const double2_t*)bp, *(const double2_t*)(bp + 2));
  };
  double4_t acc[BF_TILES];
  double cv[BF_TILES][4];
  // (The last panels have few tiles and the longest sums: splitting a tile's columns over the idle tile waves — 2 to 4 parts, the partial
  // sums handed to the tile's owner through LDS — was built, exact to 7e-15 at every size, and measured 137.9 -> 141.5 us per iteration at
  // n = 294: what the look-ahead gained in the last six panels, the owner's extra LDS round trip at the head of every such panel's chain gave back.)
  // A tile's sum over columns k0 .. k1 - 1 (multiples of 16) for the panel starting at cn: the tile's rows of L from global memory, the
  // panel's own rows (the B operand) from Bp.  (One run over all the columns with a RAW s_barrier half way, so that the loads in flight
  // survive the panel's second barrier instead of two runs that each begin with an exposed L2 round trip, was built: 143.8 -> 147.3 us.)
  auto bulk = [&](double4_t a, int cn, int ti, int k0, int k1) {
    if (k1 <= k0) return a;
    const double* ap = rowp(min(cn + 16 * ti + m, n)) + 4 * g;              // lane (m, g): row m of the tile, columns kb + 4 g .. + 3 of each 16
    const double* bp = Bp + m * BF_BPITCH + 4 * g;
    const int k1c = k1 - 16, ka = min(k0 + 16, k1c);
    // Three register sets, the loop unrolled three times: a chunk's loads are issued two chunks before its MFMAs and nothing is copied
    // (a rotation by moves made every copy wait for its load).  The offsets are opaque to the compiler (the empty asm): nothing in the
    // loop stores, so it had replaced the carried registers by a fresh load of the same address right before the MFMAs — no prefetch
    // left, one exposed L2 round trip per chunk.  Offsets past the end are clamped: a repeat, never used.
    auto ld = [&](int k, double2_t& x0, double2_t& x1) {
      k = min(k, k1c);
      asm volatile("" : "+s"(k));
      x0 = bf_ldg2(ap + k); x1 = bf_ldg2(ap + k + 2);
    };
    auto mm = [&](int kb, const double2_t& x0, const double2_t& x1) { a = mfma4(a, x0, x1, *(const double2_t*)(bp + kb), *(const double2_t*)(bp + kb + 2)); };
    double2_t r0a, r0b, r1a, r1b, r2a, r2b;
    ld(k0, r0a, r0b); ld(ka, r1a, r1b);
    for (int kb = k0; kb < k1; kb += 48) {
      ld(kb + 32, r2a, r2b); mm(kb, r0a, r0b);
      if (kb + 16 >= k1) break;
      ld(kb + 48, r0a, r0b); mm(kb + 16, r1a, r1b);
      if (kb + 32 >= k1) break;
      ld(kb + 64, r1a, r1b); mm(kb + 32, r2a, r2b);
    }
    return a;
  };
  // The panel starting at column cn = c0 + 16, while panel c0 is being factored and the rows below it solved: its entries of S and its
  // sum over the columns before c0, for this wave's tiles (kept in registers until the panel's turn).  Under the factor: columns
  // c0 - 16 .. c0 - 1 from Lp (which numbers its rows from c0 and is rewritten by the row solves) and the first half of the columns before
  // them; beside the row solves: the other half.
  auto ahead_factor = [&](int c0) {
    const int cn = c0 + BB_NB, ntn = (n + 1 - cn + 15) / 16, kend = c0 - BB_NB, ks = kend > 0 ? (kend >> 5) << 4 : 0;
#pragma unroll
    for (int t = 0; t < BF_TILES; ++t) {
      const int ti = widx + BF_TILE_WAVES * t;
      if (ti < ntn) {                                                       // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = cn + 16 * ti + g + 4 * q;
          cv[t][q] = bf_ldg1(rowp(min(r, n)) + min(cn + m, n - 1));         // (clamped: rows past the right-hand side and columns past n are never stored)
        }
        double4_t a = {0.0, 0.0, 0.0, 0.0};
        if (c0 > 0) a = lp_chunk(a, BB_NB + 16 * ti, BB_NB);
        acc[t] = bulk(a, cn, ti, 0, ks);
      }
    }
  };
  auto ahead_rows = [&](int c0) {
    const int cn = c0 + BB_NB, ntn = (n + 1 - cn + 15) / 16, kend = c0 - BB_NB, ks = kend > 0 ? (kend >> 5) << 4 : 0;
#pragma unroll
    for (int t = 0; t < BF_TILES; ++t) {
      const int ti = widx + BF_TILE_WAVES * t;
      if (ti < ntn) acc[t] = bulk(acc[t], cn, ti, ks, kend);
    }
  };
#ifdef ORBX_BF_DEBUG
  unsigned long long bf_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bf_prev_ = __builtin_amdgcn_s_memtime();
#endif
  // Two roles, two loops, the same three barriers per panel: what a tile wave keeps in registers from one panel to the next (its tiles'
  // sums and entries of S) is not live in the code of the waves that factor and solve rows, and the other way round.
  if (!row_role) {
    // ---- the tile waves
    ahead_factor(-BB_NB);                                                  // panel 0: its entries of S
    // Bp for the look-ahead two panels on — rows c0 + 32 .. + 47 of L, the columns before c0 — is requested beside a panel's row solves and
    // written at the start of the next panel, when nobody reads the old one; registers in between.  16 rows x BF_BQ slices of 64 column
    // pairs, four (row, slice) units per tile wave.
    static_assert(16 * BF_BQ <= 4 * BF_TILE_WAVES, "Bp staging: four units per tile wave");
    double2_t bq[4];
    for (int c0 = 0; c0 < n; c0 += BB_NB) {
      const int nb = min(BB_NB, n - c0);
      const int nt = (n + 1 - c0 + 15) / 16;
      if (c0 > BB_NB && c0 + BB_NB < n) {                                  // (requested in the panel before; there is a look-ahead to use it)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int unit = 4 * widx + i, k2 = lane + 64 * (unit >> 4);
          if (2 * k2 < c0 - BB_NB) *(double2_t*)&Bp[(unit & 15) * BF_BPITCH + 2 * k2] = bq[i];
        }
      }
      // the last 16 columns of the panel's sum from the LDS copy of the previous panel (which numbers its rows from this panel's first row)
#pragma unroll
      for (int t = 0; t < BF_TILES; ++t) {
        const int ti = widx + BF_TILE_WAVES * t;
        if (ti < nt) {
          double4_t a = acc[t];
          if (c0 > 0) a = lp_chunk(a, 16 * ti, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int lr = 16 * ti + g + 4 * q;
            const double v = cv[t][q] - a[q];
            if (lr < nb) D[lr][m] = v;                                     // the diagonal block goes to the factor
            else Cs[m * BF_CPITCH + lr] = v;
          }
        }
      }
      __syncthreads();
      if (c0 + BB_NB < n) ahead_factor(c0);                                // under the factor: the first part of the next panel's sum over the columns before this panel
      __syncthreads();
      if (!s_ok) break;
      // (every panel and every lane, at clamped addresses: a conditional load would keep the old value alive through the whole next panel)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int unit = 4 * widx + i;
        bq[i] = bf_ldg2(rowp(min(c0 + 2 * BB_NB + (unit & 15), n)) + 2 * min(lane + 64 * (unit >> 4), max(c0 / 2 - 1, 0)));
      }
      if (c0 + BB_NB < n) ahead_rows(c0);                                  // beside the row solves: the rest of it
      __syncthreads();
    }
  } else {
    // ---- wave 0 (the factor) and the three waves that share its SIMD: the rows below a block, one thread each (two for the first panels
    // of a system beyond 271 unknowns)
    const int ridx = (wave >> 2) * 64 + lane;
    for (int c0 = 0; c0 < n; c0 += BB_NB) {
      const int nb = min(BB_NB, n - c0);
      __syncthreads();
      BF_STAMP(1);
      if (tid < 64) {
        const int j = tid & 15;
        double Lr[16];
        if (nb == BB_NB) {
          // a full block: whole rows in pairs, no masks — what stands right of the diagonal never reaches an entry at or left of it
          // (a per-lane identity there was 16 loop-invariant constants, spilled and reloaded one scratch round trip at a time: 8 of the
          // 14 thousand cycles of this phase)
#pragma unroll
          for (int i = 0; i < 16; i += 2) { const double2_t p2 = *(const double2_t*)&D[j][i]; Lr[i] = p2[0]; Lr[i + 1] = p2[1]; }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) Lr[i] = (j < nb && i <= j) ? D[j][i] : 0.0;   // (rows >= nb take no pivot step)
        }
        const int good = chol16_rows_dpp(Lr, nb, tid, rinv);
        if (tid == 0 && !good) s_ok = 0;
        if (good && tid < BB_NB) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            D[j][i] = i <= j ? Lr[i] : 0.0;
            if (j < nb && i <= j) Sg[(size_t)(c0 + j) * n + c0 + i] = Lr[i];
          }
        }
        BF_STAMP(2);
      }
      __syncthreads();
      BF_STAMP(3);
      if (!s_ok) break;
      if (tid < nb) ginv[c0 + tid] = rinv[tid];
      // rows below the block, r = c0 + nb .. n (a short last panel: the right-hand side only), and behind them the sixteen rows of the
      // identity: e_r L11^-T = row r of L11^-T, which turns the backward substitution's panel step (below) into a 16 x 16 matrix-vector
      // product — no dependent chain there
      const int nreal = n - c0 - nb + 1;
      for (int qr = ridx; qr < nreal + BB_NB; qr += 256) {
        const bool pseudo = qr >= nreal;
        const int lr = nb + qr;
        double xr[BB_NB];
#pragma unroll
        for (int j = 0; j < BB_NB; ++j) xr[j] = pseudo ? (j == qr - nreal ? 1.0 : 0.0) : Cs[j * BF_CPITCH + lr];
#pragma unroll
        for (int t = 0; t < BB_NB; t += 2) {
          double la[BB_NB], lb[BB_NB];
#pragma unroll
          for (int jx = t + 1; jx < BB_NB; ++jx) {
            const double2_t p2 = *(const double2_t*)&D[jx][t];
            la[jx] = p2[0]; lb[jx] = p2[1];
          }
          const double2_t q2 = *(const double2_t*)&rinv[t];
          xr[t] = xr[t] * q2[0];
#pragma unroll
          for (int jx = t + 1; jx < BB_NB; ++jx) xr[jx] = fma(-xr[t], la[jx], xr[jx]);
          xr[t + 1] = xr[t + 1] * q2[1];
#pragma unroll
          for (int jx = t + 2; jx < BB_NB; ++jx) xr[jx] = fma(-xr[t + 1], lb[jx], xr[jx]);
          // (pins the pair of steps: without it all 72 LDS reads of the block are issued first and 128 VGPRs do not hold them)
          asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(xr[4]), "+v"(xr[5]), "+v"(xr[6]), "+v"(xr[7]), "+v"(xr[8]), "+v"(xr[9]),
                       "+v"(xr[10]), "+v"(xr[11]), "+v"(xr[12]), "+v"(xr[13]), "+v"(xr[14]), "+v"(xr[15]) :: "memory");
        }
        if (pseudo) {
          double* li = Linv + ((size_t)(c0 >> 4) * BB_NB + (qr - nreal)) * BB_NB;
#pragma unroll
          for (int j = 0; j < BB_NB; j += 2) { const double2_t v2 = {xr[j], xr[j + 1]}; *(double2_t*)(li + j) = v2; }
          continue;
        }
        double* dst = rowp(c0 + lr) + c0;
        if (nb == BB_NB) {
          double* lp = Lp + (lr - BB_NB) * BF_LPITCH;                        // the next panel numbers its rows from c0 + 16
#pragma unroll
          for (int j = 0; j < BB_NB; j += 2) { const double2_t v2 = {xr[j], xr[j + 1]}; *(double2_t*)(lp + j) = v2; *(double2_u*)(dst + j) = v2; }
        } else {
#pragma unroll
          for (int j = 0; j < BB_NB; ++j) if (j < nb) dst[j] = xr[j];
        }
      }
      __syncthreads();                                                     // the panel's columns of L: the LDS copy for the next panel, global memory for the ones after
      BF_STAMP(4);
    }
  }
  // ---- backward substitution L^T x = y, |dp|^2, |p|^2 (until round 4 a launch of its own, ba_big_back_kernel: 31 us at n = 294 — a
  // 16-step dependent chain per panel on one wave).  Every wave arrives here behind the same barrier (a failed pivot leaves both loops
  // after a panel's second one).  Per panel: x_p = L11^-T v_p by sixteen lanes of the last wave from the rows the row solves left in
  // Linv, then v_i -= sum_j L[c0 + j][i] x_j for the rows above, one thread each; both read what they need of the NEXT panel from
  // global memory before this panel's barriers.
  const bool ok = s_ok != 0;
  // Six waves stay: one thread per row of v (n <= 320: five waves) and the first sixteen lanes of the sixth for the panel's product;
  // the other ten leave — a block barrier costs by the waves that must arrive, and the loop below has two per panel.
  constexpr int BK_THREADS = 384;
  if (tid >= BK_THREADS) return;
  double* sb = Cs;                                                         // v, then x (n <= BF_MAX_N entries)
  double* yv = Cs + BF_MAX_N + 16;                                         // the panel's x
  double* red = yv + BB_NB;                                                // per-wave partial sums
  static_assert(BF_MAX_N <= BK_THREADS - 64 && BF_MAX_N + 16 + BB_NB + 2 * (BK_THREADS / 64) <= 16 * BF_CPITCH, "a thread per row, a wave for the panel's product; the epilogue's vectors fit the panel buffer");
  const BaState* St = win.S;
  const double* params = ba_cur(St, win.P0, win.P1);
  double* __restrict__ dp = win.dp;
  if (tid == 0 && !ok) res[2] = 0.0;
  if (ok) {
    const int c_last = ((n - 1) / BB_NB) * BB_NB;
    const int mt = tid - (BK_THREADS - 64);                                // 0 .. 15: this lane's row of the panel's L11^-T (lanes of the sixth wave — never a row of v)
    const bool mv = mt >= 0 && mt < BB_NB;
    // (FULL: a 16-column panel — everything but, possibly, the last one; a compile-time flag, because a run-time `j < nb` around every load
    // and multiply-add of the unrolled loops became a scalar branch each: some sixty basic blocks per panel, 2.7 k cycles)
    auto load_panel = [&](int c0, double (&a)[BB_NB], auto full_) {
      constexpr bool FULL = decltype(full_)::value;
      if (c0 < 0) return;
      const int nb = FULL ? BB_NB : min(BB_NB, n - c0);
      if (mv) {
#pragma unroll
        for (int j = 0; j < BB_NB; j += 2) { const double2_t v2 = bf_ldg2(Linv + ((size_t)(c0 >> 4) * BB_NB + mt) * BB_NB + j); a[j] = v2[0]; a[j + 1] = v2[1]; }
      } else if (tid < c0) {
#pragma unroll
        for (int j = 0; j < BB_NB; ++j) a[j] = (FULL || j < nb) ? bf_ldg1(Sg + (size_t)(c0 + j) * n + tid) : 0.0;
      }
    };
    // (a third register set — a panel's numbers requested TWO panels ahead, nothing copied — does not fit the 128 VGPRs of a 1024-thread
    // workgroup: 25 spills)
    double cur[BB_NB], nxt[BB_NB];
#pragma unroll
    for (int j = 0; j < BB_NB; ++j) { cur[j] = 0.0; nxt[j] = 0.0; }
    auto panel = [&](int c0, auto full_) {
      constexpr bool FULL = decltype(full_)::value;
      const int nb = FULL ? BB_NB : min(BB_NB, n - c0);
      load_panel(c0 - BB_NB, nxt, std::true_type{});                       // (the panels below the last are full)
      if (mv) {
        // four partial sums (the row's sixteen products, every fourth to one sum), then (s0 + s1) + (s2 + s3): four dependent steps instead of sixteen
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < BB_NB; ++j) s4[j & 3] = fma(cur[j], (FULL || j < nb) ? sb[c0 + j] : 0.0, s4[j & 3]);   // (row mt of L11^-T is zero left of its diagonal; rows >= nb of a short panel are not used)
        yv[mt] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
      }
      __syncthreads();
      if (tid < c0) {
        double v0 = sb[tid], v1 = 0.0;                                     // two chains of eight
#pragma unroll
        for (int j = 0; j < BB_NB; j += 2) {
          v0 = fma(-cur[j], (FULL || j < nb) ? yv[j] : 0.0, v0);
          v1 = fma(-cur[j + 1], (FULL || j + 1 < nb) ? yv[j + 1] : 0.0, v1);
        }
        sb[tid] = v0 + v1;
      } else if (tid < c0 + nb) sb[tid] = yv[tid - c0];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < BB_NB; ++j) cur[j] = nxt[j];
    };
    if (tid < n) sb[tid] = bf_ldg1(bvec + tid);                            // y: the row solves left it in the right-hand side's row
    load_panel(c_last, cur, std::false_type{});
    __syncthreads();
    panel(c_last, std::false_type{});
    for (int c0 = c_last - BB_NB; c0 >= 0; c0 -= BB_NB) panel(c0, std::true_type{});
  }
  double dsq = 0.0, psq = 0.0;
  if (tid < n) {
    const double v = ok ? sb[tid] : 0.0, pv = params[tid];
    dp[tid] = v;
    dsq = v * v; psq = pv * pv;
  }
  dsq = wave_sum(dsq); psq = wave_sum(psq);
  if ((tid & 63) == 0) { red[tid >> 6] = dsq; red[BK_THREADS / 64 + (tid >> 6)] = psq; }
  __syncthreads();
  if (tid == 0) {
    double a = red[0], c = red[BK_THREADS / 64];
#pragma unroll
    for (int w2 = 1; w2 < BK_THREADS / 64; ++w2) { a += red[w2]; c += red[BK_THREADS / 64 + w2]; }
    res[3] = a; res[4] = c;
  }
  BF_STAMP(5);
#ifdef ORBX_BF_DEBUG
  if (tid == 0) for (int k = 0; k < 8; ++k) g_bf_stamps[k] = bf_acc_[k];
#endif
}

#ifdef ORBX_BF_DEBUG
// debug build only: the one-launch factorisation alone on a host matrix (scripts/bf_debug.py).  S: n x n row-major, b: n; out: L in S, y in b, 1/L_jj in ginv
extern "C" int orbx_debug_big_factor(double* S, double* b, double* ginv_out, int n) {
  double* d = nullptr; BaWin* dw = nullptr; BaState* ds = nullptr;
  const size_t nn = (size_t)n * n;
  if (hipMalloc(&d, (nn + 21 * (size_t)n + 700) * 8) != hipSuccess || hipMalloc(&dw, sizeof(BaWin)) != hipSuccess || hipMalloc(&ds, sizeof(BaState)) != hipSuccess) return -1;
  hipMemset(ds, 0, sizeof(BaState));
  BaWin w; memset(&w, 0, sizeof(w));
  w.n = n; w.use_lds = 0; w.S = ds; w.Sg = d; w.bvec = d + nn; w.ginv = w.bvec + n; w.res = w.ginv + 19 * (size_t)n + 600; w.dp = w.res + 16; w.P0 = w.P1 = w.bvec;   // (Linv behind ginv; the epilogue's dp and |p|^2 go to scratch)
  hipMemcpy(d, S, nn * 8, hipMemcpyHostToDevice); hipMemcpy(w.bvec, b, (size_t)n * 8, hipMemcpyHostToDevice);
  const double one[3] = {0, 0, 1.0}; hipMemcpy(w.res, one, 24, hipMemcpyHostToDevice);
  hipMemcpy(dw, &w, sizeof(w), hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)ba_big_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES);
  for (int rep = 0; rep < 3; ++rep) {                                      // (the stamps of the last, warm, run are reported)
    hipMemcpy(d, S, nn * 8, hipMemcpyHostToDevice); hipMemcpy(w.bvec, b, (size_t)n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ba_big_factor_kernel, dim3(1, 1), dim3(BF_THREADS), BF_LDS_BYTES, 0, dw);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
  }
  double r3[3]; hipMemcpy(r3, w.res, 24, hipMemcpyDeviceToHost);
  hipMemcpy(S, d, nn * 8, hipMemcpyDeviceToHost); hipMemcpy(b, w.bvec, (size_t)n * 8, hipMemcpyDeviceToHost); hipMemcpy(ginv_out, w.ginv, (size_t)n * 8, hipMemcpyDeviceToHost);
  hipFree(d); hipFree(dw); hipFree(ds);
  unsigned long long st[8];
  if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_bf_stamps), 64) == hipSuccess)
    fprintf(stderr, "  ticks: first loads %llu | last chunk + barrier %llu | factor %llu | look-ahead past the factor %llu | row solves + barrier %llu | backward substitution + norms %llu\n", st[0], st[1], st[2], st[3], st[4], st[5]);
  return r3[2] != 0.0 ? 0 : 1;
}
#endif

// backward substitution L^T x = y (y = bvec after the last ba_big_step_kernel), |dp|^2, |p|^2: one block
__global__ __launch_bounds__(256) void ba_big_back_kernel(const BaWin* __restrict__ wins, int one_launch_max_n) {
  __shared__ double sb[BA_MAX_N];
  __shared__ double y[BB_NB];
  __shared__ double red[256];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaState* St = win.S;
  if (win.use_lds || St->done || win.n <= one_launch_max_n) return;      // (the one-launch factorisation substitutes backward itself)
  const int n = win.n;
  const double* __restrict__ Sg = win.Sg; const double* __restrict__ ginv = win.ginv; const double* __restrict__ bvec = win.bvec;
  double* __restrict__ dp = win.dp; double* __restrict__ res = win.res;
  const double* params = ba_cur(St, win.P0, win.P1);
  const int tid = threadIdx.x;
  const int ok = res[2] != 0.0;
  if (ok) {
    for (int i = tid; i < n; i += 256) sb[i] = bvec[i];
    __syncthreads();
    // The 16x16 diagonal block of each panel is staged in LDS by the whole block before one thread runs its dependent
    // chain: read straight from global memory that chain was ~270 serial L2 round trips per panel (n = 300: 289 us)
    __shared__ double Dg[BB_NB][BB_NB + 1];
    __shared__ double gi[BB_NB];
    // Everything a panel needs from global memory — its diagonal block, 1 / L_jj, and for every row above it the 16 entries of L
    // that couple it to the panel — is requested BEFORE the previous panel's dependent chain is waited for (the block element one
    // panel ahead, the row entries ahead of the chain): one L2 round trip per panel on the critical path instead of three.
    constexpr int BACK_ROWS = (BA_MAX_N + 255) / 256;                     // rows above a panel per thread
    const int c_last = ((n - 1) / BB_NB) * BB_NB;
    const int di = tid / BB_NB, dj = tid % BB_NB;
    auto block_elem = [&](int c0) -> double { const int nb_ = min(BB_NB, n - c0); return (c0 >= 0 && di < nb_ && dj <= di) ? Sg[(size_t)(c0 + di) * n + c0 + dj] : 0.0; };
    auto inv_elem = [&](int c0) -> double { return (c0 >= 0 && tid < min(BB_NB, n - c0)) ? ginv[c0 + tid] : 0.0; };
    double d_next = block_elem(c_last), g_next = inv_elem(c_last);
    for (int c0 = c_last; c0 >= 0; c0 -= BB_NB) {                         // backward: L^T x = y
      const int nb = min(BB_NB, n - c0);
      Dg[di][dj] = d_next;
      if (tid < BB_NB) gi[tid] = g_next;
      double Lrow[BACK_ROWS][BB_NB];
#pragma unroll
      for (int q = 0; q < BACK_ROWS; ++q) {
        const int i = tid + 256 * q;
        if (i < c0) {
#pragma unroll
          for (int j = 0; j < BB_NB; ++j) Lrow[q][j] = j < nb ? Sg[(size_t)(c0 + j) * n + i] : 0.0;
        }
      }
      d_next = block_elem(c0 - BB_NB); g_next = inv_elem(c0 - BB_NB);
      __syncthreads();
      if (tid < 64) {                                  // wave 0: lane j owns column j of the block (= row j of L^T)
        // step i: x_i = v_i / L_ii reaches every lane by a readlane pair, v_j -= L_ij x_i for j < i.  As in ba_solve_lds_kernel the step is
        // multiply, two readlanes, fma and nothing else: entries with j >= i are zeros in Lc (so the fma leaves those lanes alone), lane i
        // keeps v_i and is scaled once at the end by the same 1 / L_ii, which every lane applies to its own entry before the readlane
        // picks lane i's (until round 3: a select, two exec-mask regions and an LDS read of 1 / L_ii per step).
        const int j = tid & 15;
        double Lc[BB_NB];
#pragma unroll
        for (int i = 0; i < BB_NB; ++i) Lc[i] = (i < nb && i > j) ? Dg[i][j] : 0.0;
        double v = j < nb ? sb[c0 + j] : 0.0;
        const double gj = j < nb ? gi[j] : 0.0;
#pragma unroll
        for (int i = BB_NB - 1; i >= 0; --i) {
          const double t = v * gj;
          const double xi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), i), __builtin_amdgcn_readlane(__double2loint(t), i));
          v = fma(-Lc[i], xi, v);                      // (steps i >= nb: Lc = 0 and t = 0)
        }
        v = v * gj;
        if (tid < nb) { y[tid] = v; sb[c0 + tid] = v; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < BACK_ROWS; ++q) {
        const int i = tid + 256 * q;
        if (i < c0) {
          double v = sb[i];
#pragma unroll
          for (int j = 0; j < BB_NB; ++j) if (j < nb) v = fma(-Lrow[q][j], y[j], v);
          sb[i] = v;
        }
      }
      __syncthreads();
    }
  }
  double dsq = 0.0, psq = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double v = ok ? sb[i] : 0.0;
    dp[i] = v;
    dsq += v * v;
    psq += params[i] * params[i];
  }
  red[tid] = dsq;
  __syncthreads();
  for (int s2 = 128; s2 >= 1; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
  if (tid == 0) res[3] = red[0];
  __syncthreads();
  red[tid] = psq;
  __syncthreads();
  for (int s2 = 128; s2 >= 1; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
  if (tid == 0) res[4] = red[0];
}

// delta_l = V*^-1 (-g_l - sum_k W_kj^T delta_p_k); trial = params + delta; per-point |delta_l|^2, |p_l|^2.
// One 32-lane group per point: lanes take the point's observations, fixed shuffle tree for the three sums.
template <int LANES>
__global__ __launch_bounds__(256) void ba_backsub_kernel(const BaWin* __restrict__ wins, BaCam cam, const double* __restrict__ dp_override,
                                                         int owned_only) {
  // + the trial residuals of the point (local_ba_lm.rs:1047-1048): the group that back-substitutes a point already holds
  // its trial position, and the trial rotations follow from the pose step alone, so chi2(trial) needs no launch of its own
  __shared__ double sRt[12 * BA_MAX_K];      // trial poses
  __shared__ double sRt0[12 * BA_MAX_K];     // current poses (the linearisation point: ba_build_kernel's, read back)
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaDims d = win.d;
  const BaState* S = win.S;
  if (S->done || (int)blockIdx.x * 256 >= max(LANES * d.M, 6 * d.K)) return;
  double *P0 = win.P0, *P1 = win.P1;
  const double* __restrict__ dp = dp_override ? dp_override : win.dp;     // (inertial: the 6-d pose steps scattered out of the 15-d solve)
  const int* __restrict__ pt_start = win.pt_start; const int* __restrict__ o_kf = win.o_kf;
  const double* __restrict__ o_uv = win.o_uv; const double* __restrict__ Rt_fix = win.Rt_fix;
  const double* __restrict__ oP = win.oP; const double* __restrict__ Vinv = win.Vinv; const double* __restrict__ gl = win.gl;
  double* __restrict__ pt_dsq = win.pt_dsq; double* __restrict__ pt_psq = win.pt_psq; double* __restrict__ pt_chi2 = win.pt_chi2;
  const double* params = ba_cur(S, P0, P1);
  double* trial = ba_trial(S, P0, P1);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gtid < 6 * d.K) trial[gtid] = params[gtid] + dp[gtid];
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) {
    double p6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) p6[a] = params[6 * (size_t)k + a] + dp[6 * (size_t)k + a];   // = the trial pose written above
    pose_to_Rt(p6, cam.inertial, sRt + 12 * k);
  }
  for (int a = threadIdx.x; a < 12 * d.K; a += blockDim.x) sRt0[a] = win.Rt_cur[a];
  __syncthreads();
  const int lane32 = threadIdx.x & (LANES - 1);
  const int gstride = (int)gridDim.x * (256 / LANES);
  for (int j = gtid / LANES; j < d.M; j += gstride) {   // group-uniform (several points per group in a large batch, as ba_build_kernel)
  const double X0[3] = {params[6 * (size_t)d.K + 3 * (size_t)j], params[6 * (size_t)d.K + 3 * (size_t)j + 1], params[6 * (size_t)d.K + 3 * (size_t)j + 2]};
  const int s = pt_start[j], e = pt_start[j + 1];
  double acca[3] = {0.0, 0.0, 0.0}, accb[3] = {0.0, 0.0, 0.0};
  // sum_k W_kj^T delta_p_k = sum over the point's observations of B^T (A delta_p_k), with A and B rebuilt from the observation's
  // stored (x, y, 1/z, sqrt w) and the keyframe's current R|t — 32 B read per observation where the W block was 144.  (Round 3's
  // first attempt at this recomputed the projection itself, with its divisions and square roots: 68 -> 88 us per batch iteration.)
  auto wtdp = [&](int i, double (&acc)[3]) {
    const int k = o_kf[i];
    if (k < 0) return;
    const double* q = oP + 6 * (size_t)i;
    double A[12], B[6];
    obs_jac_from_proj(cam, sRt0 + 12 * k, q[0], q[1], q[2], q[3], A, B);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double dk = dp[6 * (size_t)k + a];
      if (!ba_a0_zero(a)) s0 = fma(A[a], dk, s0);
      if (!ba_a1_zero(a)) s1 = fma(A[6 + a], dk, s1);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = fma(B[c], s0, fma(B[3 + c], s1, acc[c]));
  };
  for (int i = s + lane32; i < e; i += 32) {
    wtdp(i, acca);
    if (LANES == 16 && i + 16 < e) wtdp(i + 16, accb);
  }
  double acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = group_sum<LANES>(acca[c], accb[c]);
  // every lane of the group forms the trial position (same arithmetic, same value); lane 0 stores it
  const double rhs[3] = {-gl[3 * (size_t)j] - acc[0], -gl[3 * (size_t)j + 1] - acc[1], -gl[3 * (size_t)j + 2] - acc[2]};
  const double* I = Vinv + 9 * (size_t)j;
  double X[3], dsq = 0.0, psq = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double dl = I[a * 3] * rhs[0] + I[a * 3 + 1] * rhs[1] + I[a * 3 + 2] * rhs[2];
    const double p = X0[a];
    X[a] = p + dl;
    dsq += dl * dl; psq += p * p;
  }
  double chia = 0.0, chib = 0.0;
  auto trial_chi = [&](int i, double& chi) {
    const int k = o_kf[i];
    double Rt[12];
    if (k >= 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = sRt[12 * k + a];
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = Rt_fix[12 * (size_t)(-1 - k) + a];
    }
    ObsOut o;
    obs_terms(cam, Rt, X, o_uv[2 * (size_t)i], o_uv[2 * (size_t)i + 1], false, o, cam.o_flag ? cam.o_flag[i] : 0);
    chi += o.chi;
  };
  for (int i = s + lane32; i < e; i += 32) {
    trial_chi(i, chia);
    if (LANES == 16 && i + 16 < e) trial_chi(i + 16, chib);
  }
  const double chi = group_sum<LANES>(chia, chib);
  if (lane32 != 0) continue;
#pragma unroll
  for (int a = 0; a < 3; ++a) trial[6 * (size_t)d.K + 3 * (size_t)j + a] = X[a];
  // partitioned over ranks: a point's |p|^2 is counted by the rank that holds its observations
  if (owned_only && pt_start[j + 1] == pt_start[j]) psq = 0.0;
  pt_dsq[j] = dsq; pt_psq[j] = psq; pt_chi2[j] = chi;
  }
}

// The back-substitution of iteration i and the build pass of iteration i + 1 in ONE pass over the observations (round 4; the visual
// solve on one GPU: BaWin::dbl).  An accepted step makes the trial parameters current, and what ba_build_kernel would then compute from them —
// residuals, the stored numbers of every observation, V, g_l, V*^-1, M — is what the trial's chi2 already needs most of: so the group
// that back-substitutes a point builds it at its trial position right away, into the set of build results that is NOT current (ba_set),
// with the lambda an acceptance leaves (lambda / 10).  ba_decide_kernel then only flips BaState::bsel.  A rejected step keeps the current
// set — same parameters, same residuals and Jacobians — and needs its lambda-dependent point matrices for lambda * 10: the group writes
// those too (BaWin::rej, from the V and g_l the set keeps), and BaState::psel makes the consumers read them.  Same arithmetic on the same
// operands in the same order as the two kernels it replaces: same bits.  One launch and one pass over o_kf / o_uv less per iteration.
template <int LANES>
__global__ __launch_bounds__(256, ORBX_BUILD_MINBLOCKS) void ba_step_kernel(const BaWin* __restrict__ wins, BaCam cam) {
  __shared__ double sRt[12 * BA_MAX_K];      // trial poses
  __shared__ double sRt0[12 * BA_MAX_K];     // current poses (the linearisation point)
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaDims d = win.d;
  const BaState* S = win.S;
  if (S->done || (int)blockIdx.x * 256 >= max(LANES * d.M, 6 * d.K)) return;
  double *P0 = win.P0, *P1 = win.P1;
  const double* __restrict__ dp = win.dp;
  const int* __restrict__ pt_start = win.pt_start; const int* __restrict__ o_kf = win.o_kf;
  const double* __restrict__ o_uv = win.o_uv; const double* __restrict__ Rt_fix = win.Rt_fix;
  const int bs = S->bsel;
  const BaSet cur = ba_set(win, bs), oth = ba_set(win, bs ^ 1);
  const double* __restrict__ oP = cur.oP; const double* __restrict__ Vinv = ba_point_mats(win, S, cur).Vinv; const double* __restrict__ gl = cur.gl;
  double* __restrict__ pt_dsq = win.pt_dsq; double* __restrict__ pt_psq = win.pt_psq;
  const size_t m1 = (size_t)max(d.M, 1);
  double* __restrict__ rMz = win.rej; double* __restrict__ rVinv = win.rej + 6 * m1; double* __restrict__ rvg = win.rej + 15 * m1;
  const double lambda = S->lambda, lam_a = fmax(lambda * 0.1, 1e-10), lam_r = fmin(lambda * 10.0, 1e10);   // what ba_decide_kernel will set (:1050-1055)
  const double* params = ba_cur(S, P0, P1);
  double* trial = ba_trial(S, P0, P1);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gtid < 6 * d.K) trial[gtid] = params[gtid] + dp[gtid];
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) {
    double p6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) p6[a] = params[6 * (size_t)k + a] + dp[6 * (size_t)k + a];   // = the trial pose written above
    pose_to_Rt(p6, cam.inertial, sRt + 12 * k);
  }
  for (int a = threadIdx.x; a < 12 * d.K; a += blockDim.x) sRt0[a] = cur.Rt_cur[a];
  __syncthreads();
  if (blockIdx.x == 0) for (int a = threadIdx.x; a < 12 * d.K; a += blockDim.x) oth.Rt_cur[a] = sRt[a];
  const int lane32 = threadIdx.x & (LANES - 1);
  const int gstride = (int)gridDim.x * (256 / LANES);
  for (int j = gtid / LANES; j < d.M; j += gstride) {   // group-uniform
  const double X0[3] = {params[6 * (size_t)d.K + 3 * (size_t)j], params[6 * (size_t)d.K + 3 * (size_t)j + 1], params[6 * (size_t)d.K + 3 * (size_t)j + 2]};
  const int s = pt_start[j], e = pt_start[j + 1];
  // ---- the back-substitution of ba_backsub_kernel: delta_l = V*^-1 (-g_l - sum_k W_kj^T delta_p_k)
  double acca[3] = {0.0, 0.0, 0.0}, accb[3] = {0.0, 0.0, 0.0};
  auto wtdp = [&](int i, double (&acc)[3]) {
    const int k = o_kf[i];
    if (k < 0) return;
    const double* q = oP + 6 * (size_t)i;
    double A[12], B[6];
    obs_jac_from_proj(cam, sRt0 + 12 * k, q[0], q[1], q[2], q[3], A, B);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double dk = dp[6 * (size_t)k + a];
      if (!ba_a0_zero(a)) s0 = fma(A[a], dk, s0);
      if (!ba_a1_zero(a)) s1 = fma(A[6 + a], dk, s1);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = fma(B[c], s0, fma(B[3 + c], s1, acc[c]));
  };
  for (int i = s + lane32; i < e; i += 32) {
    wtdp(i, acca);
    if (LANES == 16 && i + 16 < e) wtdp(i + 16, accb);
  }
  double acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = group_sum<LANES>(acca[c], accb[c]);
  const double rhs[3] = {-gl[3 * (size_t)j] - acc[0], -gl[3 * (size_t)j + 1] - acc[1], -gl[3 * (size_t)j + 2] - acc[2]};
  const double* Ic = Vinv + 9 * (size_t)j;
  double X[3], dsq = 0.0, psq = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double dl = Ic[a * 3] * rhs[0] + Ic[a * 3 + 1] * rhs[1] + Ic[a * 3 + 2] * rhs[2];
    const double p = X0[a];
    X[a] = p + dl;
    dsq += dl * dl; psq += p * p;
  }
  // ---- the build pass of ba_build_kernel at the trial position and the trial poses, into the other set
  double Va[6] = {0, 0, 0, 0, 0, 0}, ga[3] = {0, 0, 0}, chia = 0.0, Vb[6] = {0, 0, 0, 0, 0, 0}, gb[3] = {0, 0, 0}, chib = 0.0;
  auto one = [&](int i, double (&V)[6], double (&g)[3], double& chi) {
    const int k = o_kf[i];
    double Rt[12];
    if (k >= 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = sRt[12 * k + a];
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = Rt_fix[12 * (size_t)(-1 - k) + a];
    }
    ba_build_obs(cam, Rt, X, o_uv[2 * (size_t)i], o_uv[2 * (size_t)i + 1], cam.o_flag ? cam.o_flag[i] : 0, oth.oP + 6 * (size_t)i, V, g, chi);
  };
  for (int i = s + lane32; i < e; i += 32) {
    one(i, Va, ga, chia);
    if (LANES == 16 && i + 16 < e) one(i + 16, Vb, gb, chib);
  }
  double V[6], g[3];
#pragma unroll
  for (int a = 0; a < 6; ++a) V[a] = group_sum<LANES>(Va[a], Vb[a]);
#pragma unroll
  for (int a = 0; a < 3; ++a) g[a] = group_sum<LANES>(ga[a], gb[a]);
  const double chi = group_sum<LANES>(chia, chib);
  double Mm[6], I[9], vgo[3];
  ba_point_matrices(V, g, lam_a, Mm, I, vgo);
  // the current set's matrices for the lambda a rejection leaves
  double Vc[6], gc[3], Mr[6], Ir[9], vgr[3];
#pragma unroll
  for (int a = 0; a < 6; ++a) Vc[a] = cur.Vraw[6 * (size_t)j + a];
#pragma unroll
  for (int a = 0; a < 3; ++a) gc[a] = gl[3 * (size_t)j + a];
  ba_point_matrices(Vc, gc, lam_r, Mr, Ir, vgr);
  if (lane32 != 0) continue;
#pragma unroll
  for (int a = 0; a < 6; ++a) { oth.Mz[6 * (size_t)j + a] = Mm[a]; oth.Vraw[6 * (size_t)j + a] = V[a]; rMz[6 * (size_t)j + a] = Mr[a]; }
#pragma unroll
  for (int a = 0; a < 9; ++a) { oth.Vinv[9 * (size_t)j + a] = I[a]; rVinv[9 * (size_t)j + a] = Ir[a]; }
#pragma unroll
  for (int a = 0; a < 3; ++a) { oth.gl[3 * (size_t)j + a] = g[a]; oth.vg[3 * (size_t)j + a] = vgo[a]; rvg[3 * (size_t)j + a] = vgr[a]; }
  oth.pt_chi2[j] = chi;
  oth.pt_glsq[j] = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
#pragma unroll
  for (int a = 0; a < 3; ++a) trial[6 * (size_t)d.K + 3 * (size_t)j + a] = X[a];
  pt_dsq[j] = dsq; pt_psq[j] = psq;
  }
}

// out[i] = a[i] - b[i]  /  a[i] += b[i]   (merging point updates across ranks)
__global__ void ba_diff_kernel(size_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = a[i] - b[i];
}

// chi2 of a parameter vector: per-point partial sums (fixed order)
__global__ __launch_bounds__(256) void ba_chi2_kernel(const BaWin* __restrict__ wins, BaCam cam, int which) {
  __shared__ double sRt[12 * BA_MAX_K];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaDims d = win.d;
  const BaState* S = win.S;
  if (S->done || (int)blockIdx.x * (256 / 32) >= d.M) return;
  double *P0 = win.P0, *P1 = win.P1;
  const double* __restrict__ Rt_fix = win.Rt_fix;
  const int* __restrict__ pt_start = win.pt_start; const int* __restrict__ o_kf = win.o_kf;
  const double* __restrict__ o_uv = win.o_uv;
  double* __restrict__ pt_chi2 = win.pt_chi2;
  const double* params = which ? ba_trial(S, P0, P1) : ba_cur(S, P0, P1);
  block_poses(params, d.K, cam.inertial, sRt);
  // one 32-lane group per point (as ba_build_kernel): 2000 points alone would fill 8 blocks
  const int lane32 = threadIdx.x & 31;
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  if (j >= d.M) return;   // whole 32-lane group leaves together
  const double X[3] = {params[6 * (size_t)d.K + 3 * (size_t)j], params[6 * (size_t)d.K + 3 * (size_t)j + 1],
                       params[6 * (size_t)d.K + 3 * (size_t)j + 2]};
  double chia = 0.0, chib = 0.0;
  auto trial_chi = [&](int i, double& chi) {
    const int k = o_kf[i];
    double Rt[12];
    if (k >= 0) {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = sRt[12 * k + a];
    } else {
#pragma unroll
      for (int a = 0; a < 12; ++a) Rt[a] = Rt_fix[12 * (size_t)(-1 - k) + a];
    }
    ObsOut o;
    obs_terms(cam, Rt, X, o_uv[2 * (size_t)i], o_uv[2 * (size_t)i + 1], false, o, cam.o_flag ? cam.o_flag[i] : 0);
    chi += o.chi;
  };
  for (int i = pt_start[j] + lane32; i < pt_start[j + 1]; i += 32) trial_chi(i, chia);
  const double chi = group_sum<32>(chia, chib);
  if (lane32 == 0) pt_chi2[j] = chi;
}

// out[0..2] = sums over points of up to three per-point arrays, one block, fixed tree
// three = 0: out = res + 12 <- sum pt_chi2 (the initial error); three = 1: res + 5 <- sums of pt_chi2, pt_dsq, pt_psq
__global__ __launch_bounds__(256) void ba_sum3_kernel(const BaWin* __restrict__ wins, int three) {
  __shared__ double sh[3][256];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaState* S = win.S;
  if (S->done) return;
  const int M = win.d.M;
  const double* __restrict__ a = win.pt_chi2; const double* __restrict__ b = three ? win.pt_dsq : nullptr;
  const double* __restrict__ c = three ? win.pt_psq : nullptr;
  double* __restrict__ out = win.res + (three ? 5 : 12);
  double x = 0.0, y = 0.0, z = 0.0;
  for (int j = threadIdx.x; j < M; j += 256) { x += a[j]; if (b) y += b[j]; if (c) z += c[j]; }
  sh[0][threadIdx.x] = x; sh[1][threadIdx.x] = y; sh[2][threadIdx.x] = z;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s]; sh[1][threadIdx.x] += sh[1][threadIdx.x + s]; sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sh[0][0]; out[1] = sh[1][0]; out[2] = sh[2][0]; }
}

// (Taking this decision in the LAST block of the back-substitution launch — every block counts itself on a device-scope atomic, the one that
// sees all others counted reads their per-point sums — was built and passed every test, and cost 25 us per launch for one window, 117 us for
// 32: the results of the other blocks sit in the L2s of other XCDs, and the __threadfence() that makes them visible device-wide is an L2
// write-back per block.  A kernel boundary does that once.  7.5 us for this launch it is.)
// The tail of one LM iteration, local_ba_lm.rs:1022-1055, on the device.  res: [0] chi2(cur) [1] |g| [2] chol ok
// [3] |dp|^2 [4] |p_pose|^2 [5] chi2(trial) [6] |dl|^2 [7] |p_points|^2.
// With M >= 0 the kernel first forms res[5..7] itself — the fixed-order sums of the per-point chi2(trial), |delta_l|^2 and
// |p_l|^2 the back-substitution left (what ba_sum3_kernel does), plus the IMU / random-walk chi2 of the trial state — so
// that the single-GPU loop ends in one launch; M < 0: res[5..7] were reduced (and all-reduced) by the caller.
// abort: device-visible word the host sets when should_stop() turns true while the enqueued iterations drain — the solve then
// ends at the next iteration boundary, as the reference's poll at the top of the loop (:1013) does.  stop_vote (partitioned
// runs): res[8] holds the all-reduced stop votes of the ranks for the NEXT iteration, so that every rank leaves together.
#ifndef ORBX_BA_DECIDE_THREADS
#define ORBX_BA_DECIDE_THREADS 1024
#endif
constexpr int BA_DECIDE_THREADS = ORBX_BA_DECIDE_THREADS;   // (the thread count fixes the order of the three sums; 256 / 512 / 1024: one window 7.5 / 6.6 / 6.3 us, configs[4] 15.1 / 10.6 / 8.6, 32-window batch 9.4 / 8.8 / 9.8)
__global__ __launch_bounds__(BA_DECIDE_THREADS) void ba_decide_kernel(const BaWin* __restrict__ wins, int reduce_here, const double* __restrict__ imu_buf, int E,
                                                        const volatile int* __restrict__ abort_flag, int stop_vote, int first_is_initial) {
  __shared__ double sh[3][BA_DECIDE_THREADS];
  const BaWin win = ba_win_global(wins, blockIdx.y);
  BaState* S = win.S;
  if (S->done) return;
  double* __restrict__ res = win.res;
  const int M = reduce_here ? win.d.M : -1;
  // (the fused loop: the trial's per-point chi2 is the OTHER set's, written by ba_step_kernel)
  const double* __restrict__ pt_chi2 = win.dbl ? ba_set(win, S->bsel ^ 1).pt_chi2 : win.pt_chi2;
  const double* __restrict__ pt_dsq = win.pt_dsq; const double* __restrict__ pt_psq = win.pt_psq;
  const int tid = threadIdx.x;
  if (M >= 0) {
    double x = 0.0, y = 0.0, z = 0.0;
    for (int j = tid; j < M; j += BA_DECIDE_THREADS) { x += pt_chi2[j]; y += pt_dsq[j]; z += pt_psq[j]; }
    // (shuffle trees inside the waves and the wave totals in wave order — one barrier instead of log2(threads) — measured 9.2 -> 12.5 us
    // per 32-window launch, 6.0 -> 7.0 for one window: three f64 shuffle trees cost more than the LDS tree's barriers.  Round 4, withdrawn.)
    sh[0][tid] = x; sh[1][tid] = y; sh[2][tid] = z;
    __syncthreads();
    for (int s2 = BA_DECIDE_THREADS / 2; s2 >= 1; s2 >>= 1) {
      if (tid < s2) { sh[0][tid] += sh[0][tid + s2]; sh[1][tid] += sh[1][tid + s2]; sh[2][tid] += sh[2][tid + s2]; }
      __syncthreads();
    }
  }
  if (tid != 0) return;
  if (M >= 0) {
    double c = sh[0][0];
    if (E > 0) {
      double ci = 0.0;
      for (int e = 0; e < E; ++e) ci += imu_buf[(size_t)e * (18 * 18 + 18 + 2) + 18 * 18 + 18];
      c += ci;
    }
    res[5] = c; res[6] = sh[1][0]; res[7] = sh[2][0];
  }
  const double cur_sq = res[0];
  if (first_is_initial && S->iters == 1) res[12] = cur_sq;                        // the initial error (:1000-1001) is the first iteration's current error: no pass of its own
  S->cur_sq = cur_sq;
  S->final_sq = cur_sq;
  if (res[1] < S->gtol) { S->done = 1; return; }                                  // :1027-1029
  if (res[2] == 0.0) { S->done = 1; return; }                                     // :1036-1039
  const double dnorm = sqrt(res[3] + res[6]), pnorm = sqrt(res[4] + res[7]);
  if (dnorm < S->ptol * (pnorm + S->ptol)) { S->done = 1; return; }               // :1041-1044
  if (res[5] < cur_sq) {                                                          // :1050-1055
    S->sel ^= 1;
    S->final_sq = res[5];
    S->lambda = fmax(S->lambda * 0.1, 1e-10);
    if (win.dbl) { S->bsel ^= 1; S->psel = 0; }                                   // the build results of the trial parameters are current now
  } else {
    S->lambda = fmin(S->lambda * 10.0, 1e10);
    if (win.dbl) S->psel = 1;                                                     // same set, the point matrices of the larger lambda (BaWin::rej)
  }
  // should_stop at the top of the next iteration (:1013)
  if ((abort_flag && *abort_flag) || (stop_vote && res[8] > 0.0)) S->done = 1;
}

// ---- inertial terms (src/optimizer/local_inertial_ba.rs:661-698, :806-880; src/optimizer/imu_factors.rs:66-103) --------------------
// Device parameter layout in inertial mode: [6K T_wc pose | 3M points | 9K velocity, gyro bias, accel bias].

__device__ __forceinline__ void dev_scaled_axis(const double* q, double* o) {   // nalgebra UnitQuaternion::scaled_axis
  double v0 = q[1], v1 = q[2], v2 = q[3];
  if (!(q[0] >= 0.0)) { v0 = -v0; v1 = -v1; v2 = -v2; }
  const double n = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
  if (n > 0.0) {
    const double ang = atan2(n, fabs(q[0])) * 2.0;
    o[0] = v0 / n * ang; o[1] = v1 / n * ang; o[2] = v2 / n * ang;
  } else { o[0] = o[1] = o[2] = 0.0; }
}

// si / sj: pose (6) + velocity (3) of the two keyframes
__device__ __forceinline__ void imu_residual_dev(const double* si, const double* sj, const double* pre, double* r9) {
  const double dt = pre[10];
  double ri[4], rj[4];
  dev_q_from_scaled_axis(si, ri);
  dev_q_from_scaled_axis(sj, rj);
  const double ric[4] = {ri[0], -ri[1], -ri[2], -ri[3]}, drc[4] = {pre[0], -pre[1], -pre[2], -pre[3]};
  double t[4], err[4];
  dev_q_mul(drc, ric, t);
  dev_q_mul(t, rj, err);                                                  // imu_factors.rs:85
  dev_scaled_axis(err, r9);
  const double g[3] = {0.0, 0.0, -9.81};                                  // imu/sample.rs:6
  double a[3], b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) a[i] = sj[6 + i] - si[6 + i] - g[i] * dt;   // :89
  dev_q_rot(ric, a, b);
#pragma unroll
  for (int i = 0; i < 3; ++i) r9[3 + i] = b[i] - pre[4 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) a[i] = sj[3 + i] - si[3 + i] - si[6 + i] * dt - 0.5 * g[i] * dt * dt;   // :93-94
  dev_q_rot(ric, a, b);
#pragma unroll
  for (int i = 0; i < 3; ++i) r9[6 + i] = b[i] - pre[7 + i];
}

// One block (64 threads) per IMU edge.  full: residual, forward-difference Jacobian (eps 1e-6, 18 columns = pose and
// velocity of both keyframes), J^T J, J^T r; always: chi2 of the IMU and bias-random-walk residuals of the edge.
__global__ __launch_bounds__(64) void ba_imu_kernel(const BaState* S, double* P0, double* P1, int which, int full, BaInertialDev in,
                                                    double* __restrict__ imu_buf) {
  __shared__ double st[2][9];
  __shared__ double r[19][9];
  __shared__ double J[9][18];
  if (S->done) return;
  const double* params = which ? ba_trial(S, P0, P1) : ba_cur(S, P0, P1);
  const int e = blockIdx.x, tid = threadIdx.x;
  const int ki = in.edge_kf[2 * e], kj = in.edge_kf[2 * e + 1];
  const double* ex = params + 6 * (size_t)in.K + 3 * (size_t)in.M;
  if (tid < 18) {
    const int s = tid / 9, j = tid % 9, k = s ? kj : ki;
    st[s][j] = j < 6 ? params[6 * (size_t)k + j] : ex[9 * (size_t)k + (j - 6)];
  }
  __syncthreads();
  const double* pre = in.preint + 11 * (size_t)e;
  if (tid < (full ? 19 : 1)) {
    double a[9], b[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { a[j] = st[0][j]; b[j] = st[1][j]; }
    if (tid > 0) {                                                        // params_plus[col] += eps (:822-824)
      const int c = tid - 1;
      if (c < 9) a[c] += 1e-6; else b[c - 9] += 1e-6;
      if (ki == kj) { if (c < 9) b[c] += 1e-6; else a[c - 9] += 1e-6; }  // one parameter vector: both views move
    }
    double rr[9];
    imu_residual_dev(a, b, pre, rr);
#pragma unroll
    for (int k = 0; k < 9; ++k) r[tid][k] = rr[k];
  }
  __syncthreads();
  double* out = imu_buf + (size_t)e * IMU_REC;
  if (tid == 0) {
    double chi = 0.0;
    for (int k = 0; k < 9; ++k) chi += r[0][k] * r[0][k];
    for (int k = 0; k < 6; ++k) {                                         // :676-698
      const double w = k < 3 ? in.gw : in.aw;
      const double d = (ex[9 * (size_t)kj + 3 + k] - ex[9 * (size_t)ki + 3 + k]) * w;
      chi += d * d;
    }
    out[18 * 18 + 18] = chi;
  }
  if (!full) return;
  for (int t = tid; t < 9 * 18; t += 64) { const int k = t / 18, c = t % 18; J[k][c] = (r[c + 1][k] - r[0][k]) / 1e-6; }   // :833-836
  __syncthreads();
  for (int t = tid; t < 18 * 18; t += 64) {
    const int a = t / 18, b = t % 18;
    double h = 0.0;
    for (int k = 0; k < 9; ++k) h += J[k][a] * J[k][b];
    out[t] = h;
  }
  if (tid < 18) {
    double g = 0.0;
    for (int k = 0; k < 9; ++k) g += J[k][tid] * r[0][k];
    out[18 * 18 + tid] = g;
  }
}

// The 9-d preintegration residual of every edge at the given states, from the device function ba_imu_kernel calls (orbx_debug_imu_residual)
__global__ __launch_bounds__(64) void ba_debug_imu_kernel(int E, const int* __restrict__ edge_kf, const double* __restrict__ states9,
                                                          const double* __restrict__ preint, double* __restrict__ out) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= E) return;
  double a[9], b[9], rr[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { a[j] = states9[9 * (size_t)edge_kf[2 * e] + j]; b[j] = states9[9 * (size_t)edge_kf[2 * e + 1] + j]; }
  imu_residual_dev(a, b, preint + 11 * (size_t)e, rr);
#pragma unroll
  for (int k = 0; k < 9; ++k) out[9 * (size_t)e + k] = rr[k];
}

// res[slot] += sum over the edges of their chi2 (fixed order)
__global__ void ba_imu_addchi_kernel(const BaState* S, int E, const double* __restrict__ imu_buf, double* __restrict__ res, int slot) {
  if (S->done) return;
  double c = 0.0;
  for (int e = 0; e < E; ++e) c += imu_buf[(size_t)e * IMU_REC + 18 * 18 + 18];
  res[slot] += c;
}

// The damped normal equations of the 15K keyframe states after the points are eliminated (one block):
//   H = scatter(U - W V*^-1 W^T) + sum_e J_e^T J_e + random-walk blocks,  H_ii += lambda * max(JtJ_ii, 1e-6) with the
//   FULL J^T J diagonal (visual U_ii + IMU + random walk, local_inertial_ba.rs:1217-1221),  rhs = -(g - W V*^-1 g_l).
// rb: the visual reduce buffer [Sred n6^2 | U 36K | gp n6 | bred n6 | chi2 | glsq].  State order inside a keyframe as
// the reference: pose 0..5, velocity 6..8, gyro bias 9..11, accel bias 12..14.
__global__ __launch_bounds__(256) void ba_inertial_assemble_kernel(const BaState* St, double* P0, double* P1, BaInertialDev in,
                                                                   const double* __restrict__ rb, const double* __restrict__ imu_buf,
                                                                   double* __restrict__ Sg, double* __restrict__ bvec,
                                                                   double* __restrict__ gfull, double* __restrict__ res) {
  __shared__ double red[256];
  if (St->done) return;
  const double lambda = St->lambda;
  const double* params = ba_cur(St, P0, P1);
  const int K = in.K, n6 = 6 * K, n = 15 * K, tid = threadIdx.x;
  const double* U = rb + (size_t)n6 * n6;
  const double* gp = U + 36 * (size_t)K;
  const double* bred = gp + n6;
  const double* ex = params + 6 * (size_t)K + 3 * (size_t)in.M;
  for (size_t idx = tid; idx < (size_t)n * n; idx += 256) Sg[idx] = 0.0;
  for (int i = tid; i < n; i += 256) { bvec[i] = 0.0; gfull[i] = 0.0; }
  __syncthreads();
  // J^T J of the IMU and random-walk rows, edge after edge (fixed order, no atomics)
  for (int e = 0; e < in.E; ++e) {
    const int kk[2] = {in.edge_kf[2 * e], in.edge_kf[2 * e + 1]};
    const double* rec = imu_buf + (size_t)e * IMU_REC;
    for (int t = tid; t < 18 * 18; t += 256) {
      const int a = t / 18, b = t % 18;
      Sg[(size_t)(15 * kk[a / 9] + a % 9) * n + 15 * kk[b / 9] + b % 9] += rec[t];
    }
    if (tid < 18) gfull[15 * kk[tid / 9] + tid % 9] += rec[18 * 18 + tid];
    if (tid >= 32 && tid < 38) {                                          // :863-880
      const int k = tid - 32;
      const double w = k < 3 ? in.gw : in.aw;
      const int ii = 15 * kk[0] + 9 + k, jj = 15 * kk[1] + 9 + k;
      const double r = (ex[9 * (size_t)kk[1] + 3 + k] - ex[9 * (size_t)kk[0] + 3 + k]) * w;
      Sg[(size_t)ii * n + ii] += w * w; Sg[(size_t)jj * n + jj] += w * w;
      Sg[(size_t)ii * n + jj] -= w * w; Sg[(size_t)jj * n + ii] -= w * w;
      gfull[ii] += -w * r; gfull[jj] += w * r;
    }
    __syncthreads();
  }
  // visual part: U on the keyframe diagonal blocks, the gradient, damping with the full diagonal, then the Schur term
  for (int t = tid; t < 36 * K; t += 256) {
    const int k = t / 36, a = (t % 36) / 6, b = t % 6;
    Sg[(size_t)(15 * k + a) * n + 15 * k + b] += U[t];
  }
  for (int t = tid; t < n6; t += 256) gfull[15 * (t / 6) + t % 6] += gp[t];
  __syncthreads();
  for (int i = tid; i < n; i += 256) Sg[(size_t)i * n + i] += lambda * fmax(Sg[(size_t)i * n + i], 1e-6);
  __syncthreads();
  for (size_t idx = tid; idx < (size_t)n6 * n6; idx += 256) {
    const int i = (int)(idx / n6), j = (int)(idx - (size_t)i * n6);
    Sg[(size_t)(15 * (i / 6) + i % 6) * n + 15 * (j / 6) + j % 6] -= rb[idx];
  }
  for (int i = tid; i < n; i += 256) {
    double b = -gfull[i];
    if (i % 15 < 6) b += bred[6 * (i / 15) + i % 15];
    bvec[i] = b;
  }
  double gs = 0.0;
  for (int i = tid; i < n; i += 256) gs += gfull[i] * gfull[i];
  red[tid] = gs;
  __syncthreads();
  for (int s2 = 128; s2 >= 1; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
  if (tid == 0) {
    double chi = bred[n6];                                                // visual chi2 (gather kernel)
    for (int e = 0; e < in.E; ++e) chi += imu_buf[(size_t)e * IMU_REC + 18 * 18 + 18];
    res[0] = chi;
    res[1] = sqrt(red[0] + bred[n6 + 1]);                                 // |gradient| over keyframe states and points (:1213)
    res[2] = 1.0;
  }
}

// after the solve: dp15 -> the 6-d pose steps the point back-substitution reads, and the trial velocity / bias block
__global__ void ba_inertial_scatter_kernel(const BaState* S, double* P0, double* P1, int K, int M, const double* __restrict__ dp15,
                                           double* __restrict__ dp6) {
  if (S->done) return;
  const double* params = ba_cur(S, P0, P1);
  double* trial = ba_trial(S, P0, P1);
  const size_t off = 6 * (size_t)K + 3 * (size_t)M;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 15 * K; i += gridDim.x * blockDim.x) {
    const int k = i / 15, a = i % 15;
    if (a < 6) dp6[6 * k + a] = dp15[i];
    else trial[off + 9 * (size_t)k + (a - 6)] = params[off + 9 * (size_t)k + (a - 6)] + dp15[i];
  }
}

// End of a solve: the window's state, result block and CURRENT parameters into one contiguous record of the output blob
// (one D2H copy for the whole batch).  out: [0] iterations [1] final_sq [2] chi2 of the initial parameters [3] sel
// [4] done [5] bad-index code (BaState::bad) [8...] parameters.
__global__ __launch_bounds__(256) void ba_finish_kernel(const BaWin* __restrict__ wins, double* __restrict__ out_base,
                                                        const size_t* __restrict__ out_off, int np_extra_per_kf) {
  const BaWin win = ba_win_global(wins, blockIdx.y);
  const BaState* S = win.S;
  double* out = out_base + out_off[blockIdx.y];
  const size_t np = 6 * (size_t)win.d.K + 3 * (size_t)win.d.M + (size_t)np_extra_per_kf * win.d.K;
  const double* cur = ba_cur(S, win.P0, win.P1);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = (double)S->iters; out[1] = S->final_sq; out[2] = win.res[12]; out[3] = (double)S->sel; out[4] = (double)S->done;
    out[5] = (double)S->bad;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < np; i += (size_t)gridDim.x * blockDim.x) out[8 + i] = cur[i];
}

// Stage inspection (orbx_debug_ba_blocks): residual, pose block A (2x6) and point block B (2x3) of every observation at
// the given parameters, by the same obs_terms / pose_to_Rt the solver's kernels use.  One thread per observation.
__global__ __launch_bounds__(256) void ba_debug_blocks_kernel(BaCam cam, int K, int N, const double* __restrict__ params,
                                                              const double* __restrict__ Rt_fix, const int* __restrict__ o_kf,
                                                              const int* __restrict__ o_fix, const int* __restrict__ o_mp,
                                                              const double* __restrict__ o_uv, double* __restrict__ out /*N*20*/) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double Rt[12];
  const int k = o_kf[i];
  if (k >= 0) pose_to_Rt(params + 6 * (size_t)k, cam.inertial, Rt);
  else {
#pragma unroll
    for (int a = 0; a < 12; ++a) Rt[a] = Rt_fix[12 * (size_t)o_fix[i] + a];
  }
  const double* X = params + 6 * (size_t)K + 3 * (size_t)o_mp[i];
  ObsOut o;
  obs_terms(cam, Rt, X, o_uv[2 * (size_t)i], o_uv[2 * (size_t)i + 1], true, o, 0);
  double* q = out + 20 * (size_t)i;
  q[0] = o.r0; q[1] = o.r1;
#pragma unroll
  for (int a = 0; a < 12; ++a) q[2 + a] = o.A[a];
#pragma unroll
  for (int a = 0; a < 6; ++a) q[14 + a] = o.B[a];
}

// ---- host helpers -----------------------------------------------------------------------------------------------
void host_quat_rotate(const double* q, const double* v, double* o) {
  // nalgebra `UnitQuaternion * Vector3`
  const double t[3] = {2.0 * (q[2] * v[2] - q[3] * v[1]), 2.0 * (q[3] * v[0] - q[1] * v[2]), 2.0 * (q[1] * v[1] - q[2] * v[0])};
  const double c[3] = {q[2] * t[2] - q[3] * t[1], q[3] * t[0] - q[1] * t[2], q[1] * t[1] - q[2] * t[0]};
  for (int i = 0; i < 3; ++i) o[i] = t[i] * q[0] + c[i] + v[i];
}
void host_quat_to_R(const double* q, double* R) {
  const double w = q[0], i = q[1], j = q[2], k = q[3];
  const double ww = w * w, ii = i * i, jj = j * j, kk = k * k;
  const double ij = i * j * 2.0, wk = w * k * 2.0, wj = w * j * 2.0, ik = i * k * 2.0, jk = j * k * 2.0, wi = w * i * 2.0;
  R[0] = ww + ii - jj - kk; R[1] = ij - wk; R[2] = wj + ik;
  R[3] = wk + ij; R[4] = ww - ii + jj - kk; R[5] = jk - wi;
  R[6] = ik - wj; R[7] = wi + jk; R[8] = ww - ii - jj + kk;
}
// local_ba_lm.rs:642-645 + nalgebra scaled_axis()
void host_se3_to_params(const double* pose7, double* p6) {
  const double w = pose7[0];
  double v[3] = {pose7[1], pose7[2], pose7[3]};
  if (!(w >= 0.0)) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n > 0.0) {
    const double ang = std::atan2(n, std::fabs(w)) * 2.0;
    for (int i = 0; i < 3; ++i) p6[i] = v[i] / n * ang;
  } else p6[0] = p6[1] = p6[2] = 0.0;
  p6[3] = pose7[4]; p6[4] = pose7[5]; p6[5] = pose7[6];
}
// local_ba_lm.rs:648-662 then SE3::inverse (se3.rs:56-63) -> T_wc (:1062-1077)
void host_params_to_pose_wc(const double* p6, double* out7) {
  double q[4];
  const double angle = std::sqrt(p6[0] * p6[0] + p6[1] * p6[1] + p6[2] * p6[2]);
  if (angle > 1e-10) {
    double a[3] = {p6[0] / angle, p6[1] / angle, p6[2] / angle};
    const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const double s = std::sin(angle / 2.0), c = std::cos(angle / 2.0);
    q[0] = c; q[1] = a[0] / n * s; q[2] = a[1] / n * s; q[3] = a[2] / n * s;
  } else { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  const double qi[4] = {q[0], -q[1], -q[2], -q[3]};
  double rt[3];
  host_quat_rotate(qi, p6 + 3, rt);
  out7[0] = qi[0]; out7[1] = qi[1]; out7[2] = qi[2]; out7[3] = qi[3];
  out7[4] = -rt[0]; out7[5] = -rt[1]; out7[6] = -rt[2];
}

}  // namespace

// ---- host side of the solver ------------------------------------------------------------------------------------------
namespace {

constexpr size_t BA_LDS_STATIC = 22784;                              // static LDS of ba_solve_lds_kernel (sb, srinv, s_red, s_col, s_rv, ...), rounded up
constexpr size_t BA_LDS_DYN_MAX = 160 * 1024 - BA_LDS_STATIC;
static_assert(BA_TILED_LDS_MAX + BA_LDS_STATIC <= 160 * 1024 && BF_LDS_BYTES + 8 * 1024 <= 160 * 1024 && SCHW_LDS_BYTES + 8 * 1024 <= 160 * 1024,
              "dynamic LDS of the tiled solve / one-launch factorisation / Schur kernels beside their static arrays (the exact check, against the code object, runs once per device in ba_solve_batch)");

struct Carve {                                                       // byte offsets inside one buffer, 256-byte aligned pieces
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; }
};

// everything about one window that is known from its sizes alone
struct WinPlan {
  BaDims d{};
  int n = 0, use_lds = 0, skip = 0;
  size_t np = 0, n_upper = 0, rb_len = 0, lds_need = 0;
  // byte offsets: input blob (LM state, parameters, fixed poses; the caller's observations behind all of them), counter zone,
  // scratch arena, output blob
  size_t i_state, i_params, i_rtfix, i_obs;
  size_t c_fill, c_kfstart;
  size_t a_p1, a_oP, a_rtcur, a_slot, a_next, a_kfobs, a_kfpt, a_vinv, a_gl, a_vg, a_pt, a_kfpart, a_part, a_rb, a_solve, a_res, a_vraw, a_rej;
  size_t a_ptstart, a_okf, a_ouv, a_oflag, a_tmp, a_mz;
  size_t o_out;
  int n_kfobs = 0;
  double n_res = 0.0;
};
static_assert(sizeof(orbx_ba_obs) == 32 && offsetof(orbx_ba_obs, mp_idx) == 8 && offsetof(orbx_ba_obs, u) == 16,
              "ba_prep_*_kernel read an observation as one int4 (kf_idx, fixed_idx, mp_idx, _pad) and one double2 (u, v)");
static_assert(sizeof(orbx_ba_obs32) == 16 && offsetof(orbx_ba_obs32, mp_idx) == 4 && offsetof(orbx_ba_obs32, u) == 8, "... or one int2 and one float2");

// Is [p, p + bytes) pinned (hipHostMalloc / hipHostRegister) host memory — something the copy engine can read where it lies?
bool host_is_pinned(const void* p, size_t bytes) {
  if (!p || bytes == 0) return false;
  auto one = [](const void* q) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // (older runtimes answer an error for pageable memory)
    return a.type == hipMemoryTypeHost;
  };
  return one(p) && one((const char*)p + bytes - 1);
}

int pinned_reserve(orbx_handle* h, void** p, size_t* have, size_t need) {
  if (need <= *have) return ORBX_OK;
  if (*p) { ORBX_HIP(h, hipStreamSynchronize(h->stream)); ORBX_HIP(h, hipHostFree(*p)); *p = nullptr; *have = 0; }
  const size_t want = (need + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
  ORBX_HIP(h, hipHostMalloc(p, want));
  *have = want;
  return ORBX_OK;
}

// The O(K + M) host part of one window, straight into the (pinned) input blob: pose parameters, map points, fixed poses.  Everything
// per observation happens on the device (ba_prep_*_kernel).
void prep_window(const BaWinHost& w, const WinPlan& pl, uint8_t* blob, bool inertial, const BaInertialHost* inr) {
  const int K = w.K, F = w.F, M = w.M;
  double* params = (double*)(blob + pl.i_params);
  double* Rt_fix = (double*)(blob + pl.i_rtfix);
  for (int k = 0; k < K; ++k) host_se3_to_params(w.poses_cw + 7 * (size_t)k, &params[6 * (size_t)k]);   // scaled axis + translation
  if (M > 0) memcpy(&params[6 * (size_t)K], w.points, 24 * (size_t)M);
  if (inertial)                                                          // :1154-1173
    for (int k = 0; k < K; ++k) {
      double* ex = &params[6 * (size_t)K + 3 * (size_t)M + 9 * (size_t)k];
      for (int i = 0; i < 3; ++i) ex[i] = inr->velocities[3 * (size_t)k + i];
      for (int i = 0; i < 6; ++i) ex[3 + i] = inr->biases[6 * (size_t)k + i];
    }
  for (int f = 0; f <= F; ++f) {
    const double ident[7] = {1, 0, 0, 0, 0, 0, 0};
    const double* p = f < F ? w.fixed_poses_cw + 7 * (size_t)f : ident;
    host_quat_to_R(p, &Rt_fix[12 * (size_t)f]);
    Rt_fix[12 * (size_t)f + 9] = p[4]; Rt_fix[12 * (size_t)f + 10] = p[5]; Rt_fix[12 * (size_t)f + 11] = p[6];
  }
}

}  // namespace

// W windows at once (W = 1: orbx_ba_solve_visual / global / inertial).  The all-reduce hook and the inertial mode apply to a
// single window only.
int ba_solve_batch(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int W, BaWinHost* win,
                   orbx_should_stop_fn should_stop, void* user, bool global_mode, const BaInertialHost* inr, bool single_call) {
  // single_call: the one-window entry points (orbx_ba_solve_visual / global / inertial) answer ORBX_ERR_EMPTY as their return
  // value where the reference returns None; the batch entry point reports it in the window's status and returns ORBX_OK,
  // also for a batch of one (orbx.h)
  // inr != nullptr: solve_inertial_ba (local_inertial_ba.rs:1074-1275).  `poses_cw` then holds the T_wc poses of the window
  // and cfg carries max_iterations only; the point elimination, the per-keyframe 6x6 blocks and the Schur product are the
  // visual solver's kernels, the 15-d keyframe states are assembled and solved on top of them.
  const bool inertial = inr != nullptr;
  const bool have_coll = h->allreduce != nullptr || h->rccl_comm != nullptr;
  const bool dist = have_coll && !inertial && W == 1;
  if (W <= 0) return ORBX_OK;
  if ((inertial || have_coll) && W != 1)
    return orbx_fail(h, ORBX_ERR_INVALID, "the inertial mode and the all-reduce hook take one window per call");
  for (int w = 0; w < W; ++w) { *win[w].iterations = 0; *win[w].initial_error = 0.0; *win[w].final_error = 0.0; win[w].status = ORBX_OK; }
  hipStream_t st = h->stream;
  // ORBX_BA_TIMING=1: host-side phase times of this call on stderr (plan, preprocessing, descriptors + upload enqueue, launch
  // enqueue, drain, unpack)
  static const bool timing = getenv("ORBX_BA_TIMING") != nullptr;
  double t_mark[8] = {0};
  auto mark = [&](int i) { if (timing) t_mark[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  mark(0);

  // ---- plan: dimensions and the layout of the buffers
  std::vector<WinPlan> plan(W);
  Carve cin, car, cout, ccnt;
  // the fused loop (ba_step_kernel: back-substitution + the next iteration's build pass in one launch): the visual solve on one GPU.
  // ORBX_BA_FUSED=0 keeps the six-launch iteration (A/B runs); the partitioned solve (its trial chi2 is a collective) keeps it always.
  // (The inertial solve scatters its 15-d steps into the pose steps before this pass and adds its IMU terms in the decision: both outside it.)
  static const bool fused_off = [] { const char* e = getenv("ORBX_BA_FUSED"); return e && e[0] == '0'; }();
  // In a batch of 8 windows or more (16 lanes per point, several points per lane group) the two separate kernels stay.  Two fused forms
  // were measured there, both the same bits: point by point (no room for the build kernel's software pipeline over a group's points at
  // the 168 VGPRs three blocks per CU allow: 10 spilled) 79.1 us per 32-window launch against 28.3 + 37.6; and as two sweeps over a
  // group's points inside one launch (back-substitution of all of them, trial positions parked in LDS, then the build kernel's own
  // pipelined loop; 161 VGPRs, no spills) 67.0 against 28.8 + 37.8 — the trial-chi2 pass it drops is what the second set's footprint
  // and the rejected step's matrices cost.  For one window the fused pass is 10.4 us against 6.1 + 7.7 and a launch less.
  const bool fused = !fused_off && !dist && W < 8;
  const size_t dbl = fused ? 2 : 1;
  const size_t i_wins = cin.take(sizeof(BaWin) * (size_t)W);
  const size_t i_wins15 = cin.take(sizeof(BaWin));                      // inertial: the same window seen by the 15-d solve
  const size_t i_outoff = cin.take(sizeof(size_t) * (size_t)W);
  int live = 0, maxN = 0;
  size_t obs_bytes = 0;
  for (int w = 0; w < W; ++w) {
    const BaWinHost& ww = win[w];
    WinPlan& pl = plan[w];
    const int K = ww.K, F = ww.F, M = ww.M, N = ww.N;
    // local_ba_lm.rs:923-925 (with a partition the local N may be 0 while the global problem is not)
    if ((6 * (size_t)K + 3 * (size_t)M == 0) || (N == 0 && !dist && !inertial)) {
      win[w].status = ORBX_ERR_EMPTY;
      pl.skip = 1;
    } else { ++live; maxN = std::max(maxN, N); }
    if (6 * K > BA_MAX_N || K > BA_MAX_K) return orbx_fail(h, ORBX_ERR_INVALID, "window %d: at most %d optimised keyframes per window", w, BA_MAX_N / 6);
    BaDims& d = pl.d;
    d.K = K; d.F = F; d.M = M; d.N = N;
    d.P = std::max(16, (6 * K + 15) & ~15);
    d.ntile = d.P / 16;
    d.ncb = (d.ntile + 7) / 8;
    // k-splits of the Schur product: ~32 map points each, at most 128 of them; a function of the window alone (never of the
    // batch it travels in: the split fixes the summation order of S_red)
    // (24 points per split measured best for one window of 20 keyframes / 2000 points — 8, 16, 24, 32, 48, 96, 192 tried: the
    // Schur blocks take 8.5 us + 3.6 us per 8 points, the reduction of the partials grows with the number of splits)
    d.ksplit = std::max(1, std::min(128, (M + BA_PPS_TARGET - 1) / BA_PPS_TARGET));
    // (configs[4], 6 column-block pairs x 128 splits = 768 workgroups at 2 per CU: capping the splits at 100 / 85 / 64 / 48 / 32 to fit one
    // round of workgroups measured 95.7 / 99.2 / 102.8 / 120.7 / 180.0 us against 94.1 — many short workgroups it stays)
    d.pps = std::max(16, (((M + d.ksplit - 1) / d.ksplit) + 15) & ~15);   // (whole tiles of both Schur bodies: 8 and 16 points)
    d.rows = 3 * d.pps * d.ksplit;
    pl.n = 6 * K;
    pl.np = 6 * (size_t)K + 3 * (size_t)M + (inertial ? 9 * (size_t)K : 0);
    pl.n_upper = (size_t)d.ntile * (d.ntile + 1) / 2;
    pl.rb_len = (size_t)pl.n * pl.n + 36 * (size_t)K + 2 * (size_t)pl.n + 2;
    pl.lds_need = 8 * ((size_t)pl.n * pl.n + BA_SOLVE_THREADS);
    pl.use_lds = (pl.lds_need <= BA_LDS_DYN_MAX && pl.n % 2 == 0 && pl.n <= 128) ? 1 : 0;   // (the LDS solve reads pairs of entries — even n, which 6K is — and lays its threads over a 128 x 128 grid)
    if (!pl.use_lds && pl.n <= BA_TILED_MAX_N) {                          // lower 16 x 16 tiles in LDS: one launch where the multi-launch path takes n / 16 + 1
      const size_t nt = (size_t)(pl.n + 15) / 16;
      pl.use_lds = 2;
      pl.lds_need = 8 * 256 * (nt * (nt + 1) / 2);
    }
    pl.n_res = 2.0 * (double)N;
    const size_t n1 = (size_t)std::max(N, 1), m1 = (size_t)std::max(M, 1), k1 = (size_t)std::max(K, 1);
    pl.i_state = cin.take(sizeof(BaState));
    pl.i_params = cin.take(8 * std::max<size_t>(pl.np, 1));
    pl.i_rtfix = cin.take(8 * 12 * (size_t)(F + 1));
    pl.i_obs = obs_bytes; obs_bytes += (ww.obs32 ? sizeof(orbx_ba_obs32) : sizeof(orbx_ba_obs)) * (size_t)(pl.skip ? 0 : N);   // (relative to the observation region, packed; multiples of 16)
    pl.c_fill = ccnt.take(4 * m1); pl.c_kfstart = ccnt.take(4 * (k1 + 1));
    pl.a_ptstart = car.take(4 * (m1 + 1)); pl.a_okf = car.take(4 * n1); pl.a_ouv = car.take(16 * n1);
    pl.a_oflag = inertial ? car.take(4 * n1) : 0; pl.a_tmp = car.take(4 * n1);
    pl.a_p1 = car.take(8 * std::max<size_t>(pl.np, 1));
    pl.a_oP = car.take(48 * n1 * dbl); pl.a_rtcur = car.take(96 * k1 * dbl);
    pl.a_slot = car.take(4 * m1 * k1); pl.a_next = car.take(4 * n1); pl.a_kfobs = car.take(4 * n1); pl.a_kfpt = car.take(4 * n1);
    pl.a_mz = car.take(48 * m1 * dbl);
    pl.a_vinv = car.take(72 * m1 * dbl); pl.a_gl = car.take(24 * m1 * dbl); pl.a_vg = car.take(24 * m1 * dbl);
    pl.a_pt = car.take(8 * (2 * dbl + 2) * m1);                            // pt_chi2 [dbl][m1] | pt_glsq [dbl][m1] | pt_dsq [m1] | pt_psq [m1]
    pl.a_vraw = fused ? car.take(48 * m1 * dbl) : 0; pl.a_rej = fused ? car.take(8 * 18 * m1) : 0;
    pl.a_kfpart = car.take(8 * 33 * BA_KFSPLIT * k1);
    pl.a_part = car.take(8 * pl.n_upper * d.ksplit * 256);
    pl.a_rb = car.take(8 * (pl.rb_len + 8));
    pl.a_solve = car.take(8 * ((size_t)pl.n * pl.n + 21 * (size_t)pl.n + 640 + BA_SOLVE_THREADS));   // dp [n pad 16] | S [n*n] | b [n] | 1/L_jj [n pad 16] | (gradient slot [n pad 16]) | L11^-T rows [n / 16 + 1][256]
    pl.a_res = car.take(8 * 16);
    pl.o_out = cout.take(8 * (8 + pl.np));
  }
  if (live == 0) {
    if (single_call) return orbx_fail(h, ORBX_ERR_EMPTY, "no parameters or no residuals");
    return ORBX_OK;                                                      // every window reports ORBX_ERR_EMPTY in its status
  }
  const size_t small_bytes = cin.take(0);                                // descriptors, LM states, parameters, fixed poses of every window: ONE upload
  const size_t a_cnt = car.take(ccnt.off);                               // per-point / per-keyframe counters of every window: ONE memset
  // Where each window's observations come from: straight out of the caller's memory when that is pinned (the copy engine reads it where
  // it lies: no host pass over the observations at all), else through the pinned blob (a plain copy, on the handle's workers when it is
  // large).  Windows whose observations follow each other in memory travel as one copy.
  static const bool force_stage = getenv("ORBX_BA_STAGE_OBS") != nullptr;   // (A/B and tests: never read the caller's memory directly)
  struct Run { int w0, w1; size_t off, bytes; const void* src; bool direct; };
  std::vector<Run> runs;
  bool any_stage = false;
  for (int w = 0; w < W; ++w) {
    if (plan[w].skip || win[w].N == 0) continue;
    const void* src = win[w].obs32 ? (const void*)win[w].obs32 : (const void*)win[w].obs;
    const size_t bytes = (win[w].obs32 ? sizeof(orbx_ba_obs32) : sizeof(orbx_ba_obs)) * (size_t)win[w].N;
    if (!runs.empty() && (const char*)runs.back().src + runs.back().bytes == (const char*)src && runs.back().off + runs.back().bytes == plan[w].i_obs) {
      runs.back().bytes += bytes; runs.back().w1 = w + 1;
    } else runs.push_back(Run{w, w + 1, plan[w].i_obs, bytes, src, false});
  }
  for (Run& r : runs) {
    r.direct = !force_stage && host_is_pinned(r.src, r.bytes);
    if (!r.direct) any_stage = true;
  }
  enum { B_IN, B_ARENA, B_OUT, B_IMU, B_S15 };
  const int K0 = win[0].K, M0 = win[0].M;                               // inertial / partitioned: the one window
  const int n15 = 15 * K0;
  if (inertial && n15 > BA_MAX_N) return orbx_fail(h, ORBX_ERR_INVALID, "at most %d keyframes per inertial window", BA_MAX_N / 15);
  // (pure host validation comes before the first byte of the caller's memory is handed to the copy engine: ADVICE r4)
  if (inertial)
    for (int e = 0; e < inr->E; ++e)
      if (inr->edge_kf[2 * e] < 0 || inr->edge_kf[2 * e] >= K0 || inr->edge_kf[2 * e + 1] < 0 || inr->edge_kf[2 * e + 1] >= K0)
        return orbx_fail(h, ORBX_ERR_INVALID, "IMU edge %d: keyframe index out of range", e);
  if (int rc = orbx_reserve(h, h->ws_ba[B_IN], small_bytes + obs_bytes + 256)) return rc;
  if (int rc = orbx_reserve(h, h->ws_ba[B_ARENA], car.off)) return rc;
  if (int rc = orbx_reserve(h, h->ws_ba[B_OUT], cout.off)) return rc;
  // inertial: edges (int [E][2]) | preint [E][11] | per-edge J^T J records;  dp15 [n15 pad 16] | S [n15^2] | b | 1/L_jj | gradient
  if (int rc = orbx_reserve(h, h->ws_ba[B_IMU], inertial ? 8 * ((size_t)inr->E * (1 + 11 + IMU_REC) + 8) : 8)) return rc;
  if (int rc = orbx_reserve(h, h->ws_ba[B_S15], inertial ? 8 * ((size_t)n15 * n15 + 21 * (size_t)n15 + 640) : 8)) return rc;
  if (int rc = pinned_reserve(h, &h->h_ba_in, &h->h_ba_in_bytes, small_bytes + (any_stage ? obs_bytes : 0))) return rc;
  if (int rc = pinned_reserve(h, &h->h_ba_out, &h->h_ba_out_bytes, cout.off)) return rc;
  if (!h->h_abort) {
    ORBX_HIP(h, hipHostMalloc((void**)&h->h_abort, 64));
    ORBX_HIP(h, hipHostGetDevicePointer((void**)&h->d_abort, h->h_abort, 0));
  }
  *h->h_abort = 0;
  uint8_t* hin = (uint8_t*)h->h_ba_in;
  uint8_t* din = (uint8_t*)h->ws_ba[B_IN].p;
  uint8_t* dar = (uint8_t*)h->ws_ba[B_ARENA].p;
  double* dout = (double*)h->ws_ba[B_OUT].p;
  uint8_t* dobs = din + small_bytes;                                     // the observation region of the device blob
  uint8_t* hobs = hin + small_bytes;                                     // ... and of the pinned one (staged windows only)

  // The second half of a batch (orbx_ba_solve_visual_batch) orders its uploads behind the first half's: both share one PCIe link, and the
  // first half's kernels start when ITS bytes have arrived.  Whatever happens below, the first half opens the gate when it leaves.
  struct GateOpen { std::atomic<int>* g; ~GateOpen() { if (g) g->store(1, std::memory_order_release); } } gate_open{h->ba_gate_signal};
  bool direct_sent = false, drained = false;
  // Once the copy engine has been pointed at the caller's page-locked observations, no return path may leave them in flight: an error
  // return below (a HIP failure, the pool, the attribute check, the collective) first drains the stream, so the caller may reuse or free
  // its buffer the moment the call is back, whatever it answered.
  struct DrainOnLeave { hipStream_t st; const bool* sent; const bool* done; ~DrainOnLeave() { if (*sent && !*done) (void)hipStreamSynchronize(st); } } drain_on_leave{st, &direct_sent, &drained};
  auto send_direct = [&]() -> int {
    if (h->ba_gate_wait) {
      while (h->ba_gate_wait->load(std::memory_order_acquire) == 0) std::this_thread::yield();
      if (h->ba_gate_event) ORBX_HIP(h, hipStreamWaitEvent(st, h->ba_gate_event, 0));
    }
    direct_sent = true;                                                  // (set first: a failure part-way has already enqueued the earlier runs)
    for (const Run& r : runs)
      if (r.direct) ORBX_HIP(h, hipMemcpyAsync(dobs + r.off, r.src, r.bytes, hipMemcpyHostToDevice, st));
    return ORBX_OK;
  };
  mark(1);
  if (!h->ba_gate_wait) { if (int rc = send_direct()) return rc; }        // the copy engine starts at once; the host's O(K + M) part runs under it

  // ---- host part, one window per task: LM state, pose parameters, points, fixed poses — and the observations of a window whose memory
  // the copy engine cannot read, copied into the pinned blob (threads when that is large enough to pay for them)
  {
    auto work = [&](int w) {
      const WinPlan& pl = plan[w];
      BaState& s0 = *(BaState*)(hin + pl.i_state);
      memset(&s0, 0, sizeof(s0));
      s0.lambda = inertial ? inr->cfg->initial_lambda : 1e-3;            // :1006-1010 / local_inertial_ba.rs:1195
      s0.gtol = inertial ? 1e-8 : cfg->gradient_tolerance;              // local_inertial_ba.rs:1213
      s0.ptol = inertial ? 0.0 : cfg->param_tolerance;                  // the inertial loop has no step-size test
      s0.done = pl.skip;                                                 // a window the reference answers None for (:923-925) never runs
      if (!pl.skip) prep_window(win[w], pl, hin, inertial, inr);
    };
    size_t staged = 0;
    std::vector<int> stage_w;
    for (const Run& r : runs) if (!r.direct) { staged += r.bytes; for (int w = r.w0; w < r.w1; ++w) if (!plan[w].skip && win[w].N > 0) stage_w.push_back(w); }
    auto stage = [&](int i) {
      const int w = stage_w[(size_t)i];
      if (win[w].obs32) memcpy(hobs + plan[w].i_obs, win[w].obs32, sizeof(orbx_ba_obs32) * (size_t)win[w].N);
      else memcpy(hobs + plan[w].i_obs, win[w].obs, sizeof(orbx_ba_obs) * (size_t)win[w].N);
    };
    int nthr = (int)std::min<size_t>({stage_w.size(), (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, staged / (1u << 20) + 1});
    if (h->ba_pool_cap > 0) nthr = std::min(nthr, h->ba_pool_cap);
    if (nthr > 1 && !h->ba_pool) {
      try { h->ba_pool = new OrbxWorkPool((int)std::min<size_t>(15, std::max(1u, std::thread::hardware_concurrency()) - 1)); }
      catch (...) { h->ba_pool = nullptr; }                              // no workers: this thread does it all
    }
    if (nthr > 1 && h->ba_pool) {
      // the workers copy observations while this thread does the small per-window part: item 0 = that part, items 1.. = staged windows
      const std::function<void(int)> job = [&](int i) { if (i == 0) { for (int w = 0; w < W; ++w) work(w); } else stage(i - 1); };
      if (!h->ba_pool->run((int)stage_w.size() + 1, nthr - 1, job)) return orbx_fail(h, ORBX_ERR_HIP, "batch preprocessing: out of host memory");
    } else {
      for (int w = 0; w < W; ++w) work(w);
      for (int i = 0; i < (int)stage_w.size(); ++i) stage(i);
    }
  }
  mark(2);
  if (!direct_sent) { if (int rc = send_direct()) return rc; }
  for (const Run& r : runs)
    if (!r.direct) ORBX_HIP(h, hipMemcpyAsync(dobs + r.off, hobs + r.off, r.bytes, hipMemcpyHostToDevice, st));
  // window descriptors + LM states into the blob
  BaWin* hw = (BaWin*)(hin + i_wins);
  size_t* hoff = (size_t*)(hin + i_outoff);
  size_t lds_max = 0, tiled_lds_max = 0;
  int any_tiled = 0;
  // points per 32-lane group of the build / back-substitution launches: 1 for a few windows, 4 in a large batch (one block prologue —
  // the K rotations — per 32 points instead of per 8; 32-window batch: build 0.103 -> 0.096, back-substitution 0.068 -> 0.064 ms per iteration)
  const int ppg = W >= 8 ? 4 : 1;
  const int ptl = W >= 8 ? 16 : 32;   // lanes per point in those two kernels (group_sum<>: the sums do not depend on it)
  static const int one_launch_max_n = getenv("ORBX_BA_BIG_STEPS") ? 0 : BF_MAX_N;   // (ORBX_BA_BIG_STEPS: the multi-launch factorisation for every size, for A/B runs)
  int any_one = 0;
  int any_lds = 0, any_big = 0, n_big_max = 0, maxM = 0, maxK = 0, max_schur_blocks = 1, all_diag = 1, max_gather = 1, max_back = 1, max_asm = 1;
  for (int w = 0; w < W; ++w) {
    const WinPlan& pl = plan[w];
    BaWin& b = hw[w];
    memset(&b, 0, sizeof(b));
    b.d = pl.d; b.n = pl.n; b.use_lds = pl.use_lds;
    b.S = (BaState*)(din + pl.i_state);
    b.P0 = (double*)(din + pl.i_params); b.P1 = (double*)(dar + pl.a_p1);
    b.Rt_fix = (const double*)(din + pl.i_rtfix);
    b.pt_start = (const int*)(dar + pl.a_ptstart); b.o_kf = (const int*)(dar + pl.a_okf);
    b.o_uv = (const double*)(dar + pl.a_ouv);
    b.kf_start = (const int*)(dar + a_cnt + pl.c_kfstart); b.kf_obs = (int*)(dar + pl.a_kfobs); b.kf_pt = (int*)(dar + pl.a_kfpt);
    b.obs_raw = (const orbx_ba_obs*)(dobs + pl.i_obs); b.obs32 = win[w].obs32 ? 1 : 0; b.pt_fill = (int*)(dar + a_cnt + pl.c_fill); b.obs_tmp = (int*)(dar + pl.a_tmp);
    b.o_flag = inertial ? (int*)(dar + pl.a_oflag) : nullptr;
    b.Mz = (double*)(dar + pl.a_mz);
    b.Vinv = (double*)(dar + pl.a_vinv); b.gl = (double*)(dar + pl.a_gl); b.vg = (double*)(dar + pl.a_vg);
    const size_t m1 = (size_t)std::max(pl.d.M, 1);
    b.pt_chi2 = (double*)(dar + pl.a_pt); b.pt_glsq = b.pt_chi2 + dbl * m1; b.pt_dsq = b.pt_glsq + dbl * m1; b.pt_psq = b.pt_dsq + m1;
    b.dbl = fused ? 1 : 0; b.pad2_ = 0;
    b.Vraw = fused ? (double*)(dar + pl.a_vraw) : nullptr; b.rej = fused ? (double*)(dar + pl.a_rej) : nullptr;
    b.oP = (double*)(dar + pl.a_oP); b.Rt_cur = (double*)(dar + pl.a_rtcur);
    b.slot_first = (int*)(dar + pl.a_slot); b.obs_next = (int*)(dar + pl.a_next);
    b.kfpart = (double*)(dar + pl.a_kfpart); b.part = (double*)(dar + pl.a_part); b.rb = (double*)(dar + pl.a_rb);
    b.dp = (double*)(dar + pl.a_solve);
    b.Sg = b.dp + ((pl.n + 15) & ~15); b.bvec = b.Sg + (size_t)pl.n * pl.n; b.ginv = b.bvec + pl.n;
    b.res = (double*)(dar + pl.a_res);
    hoff[w] = pl.o_out / 8;
    if (pl.skip) continue;                                             // (its LM state says done: set with the window's preprocessing)
    if (pl.use_lds == 1 && !inertial) { any_lds = 1; lds_max = std::max(lds_max, pl.lds_need); }
    else if (pl.use_lds == 2 && !inertial) { any_tiled = 1; tiled_lds_max = std::max(tiled_lds_max, pl.lds_need); }
    else if (!inertial) { any_big = 1; if (pl.n > one_launch_max_n) n_big_max = std::max(n_big_max, pl.n); else any_one = 1; }
    maxM = std::max(maxM, pl.d.M); maxK = std::max(maxK, pl.d.K);
    max_schur_blocks = std::max(max_schur_blocks, pl.d.ncb * (pl.d.ncb + 1) / 2 * pl.d.ksplit);
    if (pl.d.ncb != 1) all_diag = 0;
    max_gather = std::max(max_gather, std::min(BA_GATHER_MAX_BLOCKS, (BA_GATHER_LANES * pl.n * ((pl.n + 1) / 2) + 255) / 256));
    max_back = std::max(max_back, (std::max(ptl * ((pl.d.M + ppg - 1) / ppg), pl.n) + 255) / 256);
    max_asm = std::max(max_asm, std::min(512, (pl.n * pl.n + 255) / 256));
  }
  // the build launch holds 3 workgroups per CU (166 VGPRs): a grid past that runs a second, mostly empty round of workgroups, so a large
  // batch gives each group as many points as keep the grid (with the peer half's windows) within one round
  int ppg_build = ppg;
  if (ptl == 16) {
    const size_t cap = 3 * (size_t)h->n_cu, wt = (size_t)(W + h->ba_peer_windows);
    while (ppg_build < 16 && wt * (size_t)((((maxM + ppg_build - 1) / ppg_build) * ptl + 255) / 256) > cap) ++ppg_build;
    if (const char* e = getenv("ORBX_BA_PPG")) if (*e) ppg_build = std::max(1, atoi(e));
  }
  const int pt_groups = (maxM + ppg_build - 1) / ppg_build;
  // a batch large enough to fill the chip with one workgroup per (window, gather share): the Schur workgroups add their share's partials
  // themselves (BaWin::part_sums) and the gather reads one tile set per share
  // (the windows of a batch's other half, running on the peer stream at the same time, count: together they fill the chip)
  const bool schur_sums = all_diag && W > 1 && (size_t)(W + h->ba_peer_windows) * BA_GATHER_LANES >= (size_t)h->n_cu;
  if (schur_sums) for (int w = 0; w < W; ++w) hw[w].part_sums = 1;
  BaCam bc{cam->fx, cam->fy, cam->cx, cam->cy, inertial ? inr->cfg->huber_threshold_mono : cfg->huber_threshold, global_mode ? 1 : 0,
           inertial ? 1 : 0, inertial ? inr->cfg->huber_threshold_stereo : 0.0, nullptr};
  BaInertialDev ind{};
  double* imu_buf = nullptr;
  BaWin* d_wins = (BaWin*)(din + i_wins);
  BaWin* d_wins15 = (BaWin*)(din + i_wins15);
  double *dp15 = nullptr, *gfull = nullptr;
  if (inertial) {
    bc.o_flag = hw[0].o_flag;
    int* d_edges = (int*)h->ws_ba[B_IMU].p;
    double* d_pre = (double*)h->ws_ba[B_IMU].p + inr->E;               // 2 ints per edge = 1 double slot per edge
    imu_buf = d_pre + 11 * (size_t)inr->E;
    if (inr->E > 0) {
      ORBX_HIP(h, hipMemcpyAsync(d_edges, inr->edge_kf, 8 * (size_t)inr->E, hipMemcpyHostToDevice, st));
      ORBX_HIP(h, hipMemcpyAsync(d_pre, inr->preint, 88 * (size_t)inr->E, hipMemcpyHostToDevice, st));
    }
    ind.K = K0; ind.M = M0; ind.E = inr->E;
    ind.gw = std::sqrt(inr->cfg->gyro_rw_info); ind.aw = std::sqrt(inr->cfg->accel_rw_info);
    ind.edge_kf = d_edges; ind.preint = d_pre;
    // the 15-d system: dp15 [n15 pad 16] | S [n15^2] | b [n15] | 1/L_jj [n15] | gradient [n15]
    BaWin& b15 = *(BaWin*)(hin + i_wins15);
    b15 = hw[0];
    b15.n = n15; b15.use_lds = n15 <= BA_TILED_MAX_N ? 2 : 0;             // (the 15-d system of up to 11 keyframes: the tiled LDS solve)
    if (b15.use_lds == 2) { const size_t nt = (size_t)(n15 + 15) / 16; tiled_lds_max = 8 * 256 * (nt * (nt + 1) / 2); }
    dp15 = (double*)h->ws_ba[B_S15].p;
    b15.dp = dp15;
    b15.Sg = dp15 + ((n15 + 15) & ~15); b15.bvec = b15.Sg + (size_t)n15 * n15; b15.ginv = b15.bvec + n15;
    gfull = b15.ginv + n15;
  }
  // the window descriptors, LM states, parameters and fixed poses of every window: one upload
  ORBX_HIP(h, hipMemcpyAsync(din, hin, small_bytes, hipMemcpyHostToDevice, st));
  if (h->ba_gate_signal) {                                                // this half's bytes are on their way: the other half's may follow
    if (h->ba_up_event) ORBX_HIP(h, hipEventRecord(h->ba_up_event, st));
    h->ba_gate_signal->store(1, std::memory_order_release);
  }
  mark(3);
  {
    // process-wide function attributes, set once per device to the largest size any window may ask for.  The result is kept per
    // device: a failed attempt is reported by every call that finds it, and retried — never a launch with the attribute unset.
    static std::mutex attr_m;
    static int attr_state[64];                                         // 0 = not yet, 1 = set
    std::lock_guard<std::mutex> g(attr_m);
    if (attr_state[h->device & 63] != 1) {
      hipError_t e_attr = hipFuncSetAttribute((const void*)ba_solve_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BA_LDS_DYN_MAX);
      if (e_attr == hipSuccess) e_attr = hipFuncSetAttribute((const void*)ba_solve_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BA_TILED_LDS_MAX);
      if (e_attr == hipSuccess) e_attr = hipFuncSetAttribute((const void*)ba_solve_inertial_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BA_TILED_LDS_MAX);
      if (e_attr == hipSuccess) e_attr = hipFuncSetAttribute((const void*)ba_kf_schur_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCHW_LDS_BYTES);
      if (e_attr == hipSuccess) e_attr = hipFuncSetAttribute((const void*)ba_schur_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCHW_LDS_BYTES);
      if (e_attr == hipSuccess) e_attr = hipFuncSetAttribute((const void*)ba_big_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_LDS_BYTES);
      ORBX_HIP(h, e_attr);
      // what the kernels declare statically plus the largest dynamic size must fit the CU's 160 KB — checked against the code object itself,
      // so a static array that grows fails here, by name, and not as a launch error of the one window size that needs the maximum
      const struct { const void* f; size_t dyn; const char* name; } lds[] = {
          {(const void*)ba_solve_lds_kernel, BA_LDS_DYN_MAX, "ba_solve_lds_kernel"}, {(const void*)ba_solve_tiled_kernel, BA_TILED_LDS_MAX, "ba_solve_tiled_kernel"},
          {(const void*)ba_solve_inertial_tiled_kernel, BA_TILED_LDS_MAX, "ba_solve_inertial_tiled_kernel"}, {(const void*)ba_kf_schur_kernel<true>, SCHW_LDS_BYTES, "ba_kf_schur_kernel"},
          {(const void*)ba_schur_kernel<true>, SCHW_LDS_BYTES, "ba_schur_kernel"}, {(const void*)ba_big_factor_kernel, BF_LDS_BYTES, "ba_big_factor_kernel"}};
      for (const auto& k : lds) {
        hipFuncAttributes fa;
        ORBX_HIP(h, hipFuncGetAttributes(&fa, k.f));
        if (fa.sharedSizeBytes + k.dyn > 160 * 1024)
          return orbx_fail(h, ORBX_ERR_INVALID, "internal: %s needs %zu B static + %zu B dynamic LDS, more than the 160 KB of a CU", k.name, (size_t)fa.sharedSizeBytes, k.dyn);
      }
      attr_state[h->device & 63] = 1;
    }
  }
  // k-splits per workgroup of the batch's one-column-block Schur launch: as many as make the launch about one workgroup per CU (a
  // workgroup owns its CU: 102 KB of LDS); the partials do not depend on it
  int schur_spb = all_diag ? std::max(1, std::min(16, (int)(((size_t)(W + h->ba_peer_windows) * max_schur_blocks + h->n_cu - 1) / std::max(1, h->n_cu)))) : 1;
  if (schur_sums) schur_spb = 1;                                          // (a workgroup's range is its window's share: set in the kernel)
  const BaWin& w0 = hw[0];                                              // host copies of the device pointers of window 0
  double* res0 = w0.res;
  const dim3 gW1(1, W);
  auto allreduce = [&](double* dptr, size_t cnt) -> int {
    if (!dist) return ORBX_OK;
    if (h->rccl_comm) return orbx_rccl_allreduce_sum(h, dptr, cnt, st);       // native: ncclAllReduce(ncclDouble, ncclSum) on this stream
    if (h->allreduce(h->allreduce_user, dptr, cnt, (void*)st) != 0) return orbx_fail(h, ORBX_ERR_HIP, "all-reduce hook failed");
    return ORBX_OK;
  };
  auto chi2_sum = [&](int which, int three) {
    hipLaunchKernelGGL(ba_sum3_kernel, gW1, dim3(256), 0, st, d_wins, three);
    if (inertial && inr->E > 0) {                                        // + IMU and bias-random-walk residuals (:661-698)
      hipLaunchKernelGGL(ba_imu_kernel, dim3(inr->E), dim3(64), 0, st, w0.S, w0.P0, w0.P1, which, 0, ind, imu_buf);
      hipLaunchKernelGGL(ba_imu_addchi_kernel, dim3(1), dim3(1), 0, st, w0.S, inr->E, imu_buf, res0 + (three ? 5 : 12), 0);
    }
  };
  // Partitioned over ranks: every rank must issue the same sequence of collectives, whatever its own should_stop() or its
  // own partition says — so the decisions travel with the collectives.  [total residual count | stop votes for
  // iteration 0 | ranks with a bad observation index] first (host-read), then the votes for iteration i+1 ride on the second
  // all-reduce of iteration i (res[8], read by ba_decide_kernel); a rank whose should_stop() fired keeps enqueueing.
  double n_res = plan[0].n_res;
  std::vector<double> votes((size_t)std::max(cfg->max_iterations, 0) + 2, 0.0);
  bool my_stop = false;
  // the observation CSR of every window (see ba_prep_count_kernel)
  ORBX_HIP(h, hipMemsetAsync(dar + a_cnt, 0, ccnt.off, st));
  if (maxN > 0) {
    ProfScope ps(h, "ba_prep");
    const dim3 go((maxN + BA_PREP_OPB - 1) / BA_PREP_OPB, W);
    hipLaunchKernelGGL(ba_prep_count_kernel, go, dim3(256), 0, st, d_wins);
    hipLaunchKernelGGL(ba_prep_scan_kernel, gW1, dim3(1024), 0, st, d_wins);
    hipLaunchKernelGGL(ba_prep_place_kernel, go, dim3(256), 0, st, d_wins);
    if (maxM > 0) hipLaunchKernelGGL(ba_prep_order_kernel, dim3((maxM * 16 + 255) / 256, W), dim3(256), 0, st, d_wins);
  }
  if (dist) {
    my_stop = should_stop && should_stop(user);
    int bad0 = 0;                                                         // this rank's index check, from the device (the stream is drained here anyway)
    ORBX_HIP(h, hipMemcpyAsync(&bad0, &w0.S->bad, sizeof(int), hipMemcpyDeviceToHost, st));
    ORBX_HIP(h, hipStreamSynchronize(st));
    const double head[3] = {n_res, my_stop ? 1.0 : 0.0, bad0 != 0 ? 1.0 : 0.0};
    ORBX_HIP(h, hipMemcpyAsync(res0 + 8, head, 24, hipMemcpyHostToDevice, st));
    ORBX_HIP(h, hipStreamSynchronize(st));
    if (int rc = allreduce(res0 + 8, 3)) return rc;
    double tot[3];
    ORBX_HIP(h, hipMemcpyAsync(tot, res0 + 8, 24, hipMemcpyDeviceToHost, st));
    ORBX_HIP(h, hipStreamSynchronize(st));
    if (tot[2] > 0.0) return orbx_fail(h, ORBX_ERR_INVALID, "an observation index is out of range on %d rank(s)%s", (int)tot[2], bad0 != 0 ? " (this one too)" : "");
    n_res = tot[0];
    if (n_res == 0.0) return orbx_fail(h, ORBX_ERR_EMPTY, "no residuals on any rank");
    my_stop = tot[1] > 0.0;                                              // the collective decision for iteration 0
  }

  if (maxM > 0 && maxK > 0) {
    ProfScope ps(h, "ba_slots_kflist");
    hipLaunchKernelGGL(ba_slots_kernel, dim3((maxM + BA_SLOT_PPB - 1) / BA_SLOT_PPB, W), dim3(256), 0, st, d_wins);
    hipLaunchKernelGGL(ba_kflist_kernel, dim3(maxK, W), dim3(BA_KFL_THREADS), 0, st, d_wins);
  }
  // initial error (:1000-1001) -> res[12].  The visual solve on one GPU takes it from its first iteration (the current error the gather
  // assembles from the build kernel's per-point sums, ba_decide_kernel) — a pass over every observation and a reduction less per call;
  // a call that enqueues no iteration, the partitioned solve (the sum is a collective) and the inertial one (IMU terms) compute it here.
  const bool init_from_first = !dist && !inertial;
  auto initial_error_pass = [&]() {
    ProfScope ps(h, "ba_chi2");
    if (maxM > 0) hipLaunchKernelGGL(ba_chi2_kernel, dim3((maxM * 32 + 255) / 256, W), dim3(256), 0, st, d_wins, bc, 0);
    chi2_sum(0, 0);
  };
  if (!init_from_first) {
    initial_error_pass();
    if (int rc = allreduce(res0 + 12, 1)) return rc;
  }
  bool any_iteration = false;

  // The loop only polls should_stop (:1013) and enqueues; nothing below waits for the GPU.
  bool stopped = my_stop;
  for (int iter = 0; iter < cfg->max_iterations && !(dist && iter == 0 && my_stop); ++iter) {           // :1012
    if (!dist && should_stop && should_stop(user)) { stopped = true; break; }                           // :1013
    any_iteration = true;
    const bool fused_loop = fused && maxM > 0;                               // (iterations 1, 2, ...: the step kernel of the iteration before has built them)
    if (!fused_loop || iter == 0) {
      ProfScope ps(h, "ba_build_kernel");
      if (maxM > 0) {
        const dim3 g((pt_groups * ptl + 255) / 256, W);
        if (ptl == 16) hipLaunchKernelGGL(ba_build_kernel<16>, g, dim3(256), 0, st, d_wins, bc, iter);
        else hipLaunchKernelGGL(ba_build_kernel<32>, g, dim3(256), 0, st, d_wins, bc, iter);
      }
      else hipLaunchKernelGGL(ba_iter_kernel, gW1, dim3(1), 0, st, d_wins, iter);
    }
    if (maxK > 0 && W == 1) {
      ProfScope ps(h, "ba_kf_schur_kernel");
      const dim3 g(maxK * BA_KFSPLIT + max_schur_blocks, 1);
      if (all_diag) hipLaunchKernelGGL(ba_kf_schur_kernel<true>, g, dim3(SCHW_THREADS), SCHW_LDS_BYTES, st, d_wins, bc);
      else hipLaunchKernelGGL(ba_kf_schur_kernel<false>, g, dim3(BA_KFS_GEN_THREADS), 0, st, d_wins, bc);
    } else if (maxK > 0) {
      {
        ProfScope ps(h, "ba_kf_kernel");
        hipLaunchKernelGGL(ba_kf_kernel, dim3(maxK * BA_KFSPLIT, W), dim3(BA_KF_THREADS), 0, st, d_wins, bc);
      }
      ProfScope ps(h, "ba_schur_kernel", nullptr, true);
      if (all_diag) hipLaunchKernelGGL(ba_schur_kernel<true>, dim3(schur_sums ? BA_GATHER_LANES : (max_schur_blocks + schur_spb - 1) / schur_spb, W), dim3(SCHW_THREADS), SCHW_LDS_BYTES, st, d_wins, bc, schur_spb);
      else hipLaunchKernelGGL(ba_schur_kernel<false>, dim3(max_schur_blocks, W), dim3(256), 0, st, d_wins, bc, 1);
    }
    {
      ProfScope ps(h, "ba_gather_kernel");
      // (with the shares already summed by the Schur launch a lane reads ONE pair per element: a block per 32 element pairs was 7 200 blocks of a
      // few hundred cycles each for 32 windows; ORBX_BA_GATHER_DIV elements per thread: 1 / 2 / 4 / 8 / 16 measured 14.7 / 13.4 / 12.8 / 15.1 / 18.5 us)
      static const int gather_div = [] { const char* e = getenv("ORBX_BA_GATHER_DIV"); const int v = e ? atoi(e) : 4; return v >= 1 && v <= 64 ? v : 4; }();
      hipLaunchKernelGGL(ba_gather_kernel, dim3(schur_sums ? std::max(1, max_gather / gather_div) : max_gather, W), dim3(256), 0, st, d_wins, fused_loop && iter > 0 ? iter : -1);
    }
    if (int rc = allreduce(w0.rb, plan[0].rb_len)) return rc;
    if (inertial) {
      ProfScope ps(h, "ba_inertial_solve");
      const BaWin& b15 = *(const BaWin*)(hin + i_wins15);
      if (inr->E > 0) hipLaunchKernelGGL(ba_imu_kernel, dim3(inr->E), dim3(64), 0, st, w0.S, w0.P0, w0.P1, 0, 1, ind, imu_buf);
      if (n15 <= BA_TILED_MAX_N) hipLaunchKernelGGL(ba_solve_inertial_tiled_kernel, dim3(1, 1), dim3(BA_SOLVE_THREADS), tiled_lds_max, st, d_wins15, ind, (const double*)imu_buf);
      else {
        hipLaunchKernelGGL(ba_inertial_assemble_kernel, dim3(1), dim3(256), 0, st, w0.S, w0.P0, w0.P1, ind, w0.rb, imu_buf, b15.Sg, b15.bvec, gfull, res0);
        if (n15 <= one_launch_max_n) hipLaunchKernelGGL(ba_big_factor_kernel, dim3(1, 1), dim3(BF_THREADS), BF_LDS_BYTES, st, d_wins15);
        else for (int c0 = 0; c0 < n15; c0 += BB_NB) {
          const int nt = (n15 - c0 + 15) / 16, units = c0 > 0 ? nt * (nt - 1) / 2 : 0;
          hipLaunchKernelGGL(ba_big_step_kernel, dim3(1 + (units + BB_STEP_WAVES - 1) / BB_STEP_WAVES, 1), dim3(BB_STEP_THREADS), 0, st, d_wins15, c0, one_launch_max_n);
        }
        if (n15 > one_launch_max_n) hipLaunchKernelGGL(ba_big_back_kernel, dim3(1, 1), dim3(256), 0, st, d_wins15, one_launch_max_n);
      }
      hipLaunchKernelGGL(ba_inertial_scatter_kernel, dim3(1), dim3(256), 0, st, w0.S, w0.P0, w0.P1, K0, M0, dp15, w0.dp);
    } else {
      ProfScope ps(h, "ba_solve_kernel");
      if (any_lds) hipLaunchKernelGGL(ba_solve_lds_kernel, gW1, dim3(BA_SOLVE_THREADS), lds_max, st, d_wins);
      if (any_big || any_tiled) hipLaunchKernelGGL(ba_big_assemble_kernel, dim3(max_asm, W), dim3(256), 0, st, d_wins);
      if (any_tiled) hipLaunchKernelGGL(ba_solve_tiled_kernel, gW1, dim3(BA_SOLVE_THREADS), tiled_lds_max, st, d_wins);
      if (any_big) {
        if (any_one) hipLaunchKernelGGL(ba_big_factor_kernel, gW1, dim3(BF_THREADS), BF_LDS_BYTES, st, d_wins);
        for (int c0 = 0; c0 < n_big_max; c0 += BB_NB) {
          const int nt = (n_big_max - c0 + 15) / 16, units = c0 > 0 ? nt * (nt - 1) / 2 : 0;
          hipLaunchKernelGGL(ba_big_step_kernel, dim3(1 + (units + BB_STEP_WAVES - 1) / BB_STEP_WAVES, W), dim3(BB_STEP_THREADS), 0, st, d_wins, c0, one_launch_max_n);
        }
        if (n_big_max > 0) hipLaunchKernelGGL(ba_big_back_kernel, gW1, dim3(256), 0, st, d_wins, one_launch_max_n);
      }
    }
    if (fused_loop) {
      ProfScope ps(h, "ba_step_kernel");
      if (ptl == 16) hipLaunchKernelGGL(ba_step_kernel<16>, dim3(max_back, W), dim3(256), 0, st, d_wins, bc);
      else hipLaunchKernelGGL(ba_step_kernel<32>, dim3(max_back, W), dim3(256), 0, st, d_wins, bc);
    } else {
      ProfScope ps(h, "ba_backsub_kernel");
      if (ptl == 16) hipLaunchKernelGGL(ba_backsub_kernel<16>, dim3(max_back, W), dim3(256), 0, st, d_wins, bc, (const double*)nullptr, dist ? 1 : 0);
      else hipLaunchKernelGGL(ba_backsub_kernel<32>, dim3(max_back, W), dim3(256), 0, st, d_wins, bc, (const double*)nullptr, dist ? 1 : 0);
    }
    // trial residuals (:1047-1048): res[5] = chi2(trial), res[6] = |delta_l|^2, res[7] = |p_l|^2 from the per-point parts the
    // back-substitution kernel left; then the accept / reject / stop decision (:1041-1055)
    if (dist) {
      {
        ProfScope ps(h, "ba_chi2");
        chi2_sum(1, 1);
      }
      // this rank's vote on running iteration iter + 1 (polled here, once per iteration as :1013)
      if (!stopped && iter + 1 < cfg->max_iterations && should_stop && should_stop(user)) stopped = true;
      votes[(size_t)iter] = stopped ? 1.0 : 0.0;
      ORBX_HIP(h, hipMemcpyAsync(res0 + 8, &votes[(size_t)iter], 8, hipMemcpyHostToDevice, st));
      if (int rc = allreduce(res0 + 5, 4)) return rc;
      ProfScope ps(h, "ba_decide_kernel");
      hipLaunchKernelGGL(ba_decide_kernel, gW1, dim3(BA_DECIDE_THREADS), 0, st, d_wins, 0, (const double*)nullptr, 0, (const volatile int*)nullptr, 1, 0);
    } else {
      ProfScope ps(h, "ba_decide_kernel");
      const int E = inertial ? inr->E : 0;
      if (E > 0) hipLaunchKernelGGL(ba_imu_kernel, dim3(E), dim3(64), 0, st, w0.S, w0.P0, w0.P1, 1, 0, ind, imu_buf);
      hipLaunchKernelGGL(ba_decide_kernel, gW1, dim3(BA_DECIDE_THREADS), 0, st, d_wins, 1, (const double*)imu_buf, E, (const volatile int*)h->d_abort, 0, init_from_first ? 1 : 0);
    }
  }
  if (init_from_first && !any_iteration) initial_error_pass();
  {
    size_t np_max = 1;
    for (int w = 0; w < W; ++w) np_max = std::max(np_max, plan[w].np);
    hipLaunchKernelGGL(ba_finish_kernel, dim3((unsigned)std::min<size_t>(64, (np_max + 255) / 256), W), dim3(256), 0, st, d_wins, dout,
                       (const size_t*)(din + i_outoff), inertial ? 9 : 0);
  }
  double* hout = (double*)h->h_ba_out;
  ORBX_HIP(h, hipMemcpyAsync(hout, dout, cout.off, hipMemcpyDeviceToHost, st));   // ONE download
  mark(4);
  // While the enqueued iterations drain, keep asking should_stop: a stop requested now (the local mapper's "new keyframe
  // arrived") ends the solve at the next iteration boundary, as in the reference, instead of being seen only by the polls
  // at enqueue time, which are all over within the first few hundred microseconds.
  if (should_stop && !dist && !stopped) {
    while (hipStreamQuery(st) == hipErrorNotReady)
      if (should_stop(user)) { *(volatile int*)h->h_abort = 1; break; }
  }
  ORBX_HIP(h, hipStreamSynchronize(st));
  drained = true;
  ORBX_HIP(h, hipGetLastError());
  mark(5);

  for (int w = 0; w < W; ++w) {                                          // an observation index out of range (found by ba_prep_count_kernel): the call fails, nothing is written
    if (plan[w].skip) continue;
    const int code = (int)hout[plan[w].o_out / 8 + 5];
    if (code == 0) continue;
    const int i = 0x7fffffff - code;
    int kf, fx, mp;
    if (win[w].obs32) { const orbx_ba_obs32& o = win[w].obs32[i]; kf = o.kf_idx >= 0 ? o.kf_idx : -1; fx = o.kf_idx >= 0 ? -1 : -1 - o.kf_idx; mp = o.mp_idx; }
    else { const orbx_ba_obs& o = win[w].obs[i]; kf = o.kf_idx; fx = o.fixed_idx; mp = o.mp_idx; }
    return orbx_fail(h, ORBX_ERR_INVALID, "window %d observation %d: index out of range (kf %d/%d, fixed %d/%d, mp %d/%d)", w, i, kf,
                     win[w].K, fx, win[w].F, mp, win[w].M);
  }
  for (int w = 0; w < W; ++w) {
    if (plan[w].skip) continue;
    const WinPlan& pl = plan[w];
    const BaWinHost& ww = win[w];
    const int K = ww.K, M = ww.M;
    const double* o = hout + pl.o_out / 8;
    const int iters = (int)o[0];
    *ww.iterations = iters;
    const double init_sq = o[2], final_sq = iters > 0 ? o[1] : o[2];
    if (inertial) {                                                      // |r|, not RMS (local_inertial_ba.rs:1191, :1244)
      *ww.initial_error = std::sqrt(init_sq);
      *ww.final_error = std::sqrt(final_sq);
    } else {
      const double nr = dist ? n_res : pl.n_res;
      *ww.initial_error = std::sqrt(init_sq) / std::sqrt(nr);
      *ww.final_error = std::sqrt(final_sq) / std::sqrt(nr);           // :1059-1060
    }
    std::vector<double> merged;
    const double* params = o + 8;
    if (dist && M > 0) {
      // every point moved only on the rank that owns it: sum the per-rank updates
      const double* init_pts = (const double*)(hin + pl.i_params) + 6 * (size_t)K;   // the blob still holds the initial parameters
      double* cur = o[3] != 0.0 ? w0.P1 : w0.P0;
      double* other = o[3] != 0.0 ? w0.P0 : w0.P1;
      ORBX_HIP(h, hipMemcpyAsync(other, init_pts, 8 * 3 * (size_t)M, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(ba_diff_kernel, dim3(64), dim3(256), 0, st, 3 * (size_t)M, cur + 6 * (size_t)K, other, other + 3 * (size_t)M);
      if (int rc = allreduce(other + 3 * (size_t)M, 3 * (size_t)M)) return rc;
      std::vector<double> upd(3 * (size_t)M);
      ORBX_HIP(h, hipMemcpyAsync(upd.data(), other + 3 * (size_t)M, 8 * 3 * (size_t)M, hipMemcpyDeviceToHost, st));
      ORBX_HIP(h, hipStreamSynchronize(st));
      merged.assign(params, params + pl.np);
      for (size_t j = 0; j < 3 * (size_t)M; ++j) merged[6 * (size_t)K + j] = init_pts[j] + upd[j];
      params = merged.data();
    }
    if (inertial) {                                                      // extract_pose / velocity / bias (:584-608, :1250-1254)
      for (int k = 0; k < K; ++k) {
        const double* p6 = &params[6 * (size_t)k];
        double* q = ww.poses_wc_out + 7 * (size_t)k;
        const double v[3] = {p6[0] / 2.0, p6[1] / 2.0, p6[2] / 2.0};
        const double nn = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], eps = 2.220446049250313e-16;
        if (nn <= eps * eps) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
        else { const double nv = std::sqrt(nn), sn = 1.0 * std::sin(nv) / nv; q[0] = 1.0 * std::cos(nv); q[1] = v[0] * sn; q[2] = v[1] * sn; q[3] = v[2] * sn; }
        q[4] = p6[3]; q[5] = p6[4]; q[6] = p6[5];
        const double* ex = &params[6 * (size_t)K + 3 * (size_t)M + 9 * (size_t)k];
        for (int i = 0; i < 3; ++i) inr->vel_out[3 * (size_t)k + i] = ex[i];
        for (int i = 0; i < 6; ++i) inr->bias_out[6 * (size_t)k + i] = ex[3 + i];
      }
    } else {
      for (int k = 0; k < K; ++k) host_params_to_pose_wc(&params[6 * (size_t)k], ww.poses_wc_out + 7 * (size_t)k);
    }
    for (int j = 0; j < 3 * M; ++j) ww.points[j] = params[6 * (size_t)K + j];
  }
  if (timing) {
    mark(6);
    fprintf(stderr, "[orbx ba] W=%d in=%.1f MB: plan %.3f | prep %.3f | descriptors+upload enqueue %.3f | iteration enqueue %.3f | drain %.3f | unpack %.3f ms\n",
            W, cin.off / 1e6, t_mark[1] - t_mark[0], t_mark[2] - t_mark[1], t_mark[3] - t_mark[2], t_mark[4] - t_mark[3], t_mark[5] - t_mark[4], t_mark[6] - t_mark[5]);
  }
  return ORBX_OK;
}

int ba_solve_visual(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K,
                    const double* poses_cw, int F, const double* fixed_poses_cw, int M, double* points, int N,
                    const orbx_ba_obs* obs, orbx_should_stop_fn should_stop, void* user, double* poses_wc_out,
                    int* iterations, double* initial_error, double* final_error, bool global_mode, const BaInertialHost* inr, const orbx_ba_obs32* obs32) {
  BaWinHost w{K, F, M, N, poses_cw, fixed_poses_cw, points, obs, poses_wc_out, iterations, initial_error, final_error, ORBX_OK, obs32};
  return ba_solve_batch(h, cam, cfg, 1, &w, should_stop, user, global_mode, inr, true);
}

// compute_imu_residual (imu_factors.rs:66-103) of every edge: out [E][9] = rotation | velocity | position residual
int ba_debug_imu_residual(orbx_handle* h, int K, const double* poses_wc, const double* velocities, int E, const int* edge_kf,
                          const double* preint, double* out) {
  if (E == 0) return ORBX_OK;
  for (int e = 0; e < E; ++e)
    if (edge_kf[2 * e] < 0 || edge_kf[2 * e] >= K || edge_kf[2 * e + 1] < 0 || edge_kf[2 * e + 1] >= K)
      return orbx_fail(h, ORBX_ERR_INVALID, "IMU edge %d: keyframe index out of range", e);
  std::vector<double> st(9 * (size_t)K);
  for (int k = 0; k < K; ++k) {
    host_se3_to_params(poses_wc + 7 * (size_t)k, &st[9 * (size_t)k]);      // scaled axis + translation of T_wc, as the solver's parameters
    for (int i = 0; i < 3; ++i) st[9 * (size_t)k + 6 + i] = velocities[3 * (size_t)k + i];
  }
  Carve c;
  const size_t o_s = c.take(8 * st.size()), o_e = c.take(8 * (size_t)E), o_p = c.take(88 * (size_t)E), o_o = c.take(72 * (size_t)E);
  if (int rc = orbx_reserve(h, h->ws_ba[6], c.off)) return rc;
  uint8_t* d = (uint8_t*)h->ws_ba[6].p;
  hipStream_t s = h->stream;
  ORBX_HIP(h, hipMemcpyAsync(d + o_s, st.data(), 8 * st.size(), hipMemcpyHostToDevice, s));
  ORBX_HIP(h, hipMemcpyAsync(d + o_e, edge_kf, 8 * (size_t)E, hipMemcpyHostToDevice, s));
  ORBX_HIP(h, hipMemcpyAsync(d + o_p, preint, 88 * (size_t)E, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ba_debug_imu_kernel, dim3((E + 63) / 64), dim3(64), 0, s, E, (const int*)(d + o_e), (const double*)(d + o_s),
                     (const double*)(d + o_p), (double*)(d + o_o));
  ORBX_HIP(h, hipMemcpyAsync(out, d + o_o, 72 * (size_t)E, hipMemcpyDeviceToHost, s));
  ORBX_HIP(h, hipStreamSynchronize(s));
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

// residual / Jacobian blocks of every observation at the given parameters (input order): out [N][20] = r (2) | A (12) | B (6)
int ba_debug_blocks(orbx_handle* h, const orbx_camera* cam, const orbx_ba_config* cfg, int K, const double* poses_cw, int F,
                    const double* fixed_poses_cw, int M, const double* points, int N, const orbx_ba_obs* obs, int global_mode,
                    double* out) {
  if (N == 0) return ORBX_OK;
  for (int i = 0; i < N; ++i) {
    const orbx_ba_obs& o = obs[i];
    if (o.mp_idx < 0 || o.mp_idx >= M || o.kf_idx >= K || (o.kf_idx < 0 && o.fixed_idx >= F))
      return orbx_fail(h, ORBX_ERR_INVALID, "observation %d: index out of range", i);
  }
  std::vector<double> params(6 * (size_t)K + 3 * (size_t)M), Rt_fix(12 * (size_t)(F + 1)), uv(2 * (size_t)N);
  std::vector<int> okf(N), ofix(N), omp(N);
  for (int k = 0; k < K; ++k) host_se3_to_params(poses_cw + 7 * (size_t)k, &params[6 * (size_t)k]);
  for (int j = 0; j < 3 * M; ++j) params[6 * (size_t)K + j] = points[j];
  for (int f = 0; f <= F; ++f) {
    const double ident[7] = {1, 0, 0, 0, 0, 0, 0};
    const double* p = f < F ? fixed_poses_cw + 7 * (size_t)f : ident;
    host_quat_to_R(p, &Rt_fix[12 * (size_t)f]);
    Rt_fix[12 * (size_t)f + 9] = p[4]; Rt_fix[12 * (size_t)f + 10] = p[5]; Rt_fix[12 * (size_t)f + 11] = p[6];
  }
  for (int i = 0; i < N; ++i) {
    okf[i] = obs[i].kf_idx >= 0 ? obs[i].kf_idx : -1;
    ofix[i] = obs[i].kf_idx >= 0 ? 0 : (obs[i].fixed_idx >= 0 ? obs[i].fixed_idx : F);
    omp[i] = obs[i].mp_idx; uv[2 * (size_t)i] = obs[i].u; uv[2 * (size_t)i + 1] = obs[i].v;
  }
  Carve c;
  const size_t o_p = c.take(8 * params.size()), o_r = c.take(8 * Rt_fix.size()), o_k = c.take(4 * (size_t)N), o_f = c.take(4 * (size_t)N),
               o_m = c.take(4 * (size_t)N), o_u = c.take(16 * (size_t)N), o_o = c.take(160 * (size_t)N);
  if (int rc = orbx_reserve(h, h->ws_ba[6], c.off)) return rc;
  uint8_t* d = (uint8_t*)h->ws_ba[6].p;
  hipStream_t st = h->stream;
  ORBX_HIP(h, hipMemcpyAsync(d + o_p, params.data(), 8 * params.size(), hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d + o_r, Rt_fix.data(), 8 * Rt_fix.size(), hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d + o_k, okf.data(), 4 * (size_t)N, hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d + o_f, ofix.data(), 4 * (size_t)N, hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d + o_m, omp.data(), 4 * (size_t)N, hipMemcpyHostToDevice, st));
  ORBX_HIP(h, hipMemcpyAsync(d + o_u, uv.data(), 16 * (size_t)N, hipMemcpyHostToDevice, st));
  BaCam bc{cam->fx, cam->fy, cam->cx, cam->cy, cfg->huber_threshold, global_mode ? 1 : 0, 0, 0.0, nullptr};
  hipLaunchKernelGGL(ba_debug_blocks_kernel, dim3((N + 255) / 256), dim3(256), 0, st, bc, K, N, (const double*)(d + o_p), (const double*)(d + o_r),
                     (const int*)(d + o_k), (const int*)(d + o_f), (const int*)(d + o_m), (const double*)(d + o_u), (double*)(d + o_o));
  ORBX_HIP(h, hipMemcpyAsync(out, d + o_o, 160 * (size_t)N, hipMemcpyDeviceToHost, st));
  ORBX_HIP(h, hipStreamSynchronize(st));
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}
