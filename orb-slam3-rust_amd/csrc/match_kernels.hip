// match_kernels.hip — 256-bit Hamming matchers for gfx950 (wave64).
//
// Replaces, bit-exactly:
//   src/tracking/frame/stereo.rs:80-161   match_features (gated brute force, best/second, ratio)
//   src/tracking/frame/stereo.rs:166-175  descriptor_distance
//   src/tracking/frame/stereo.rs:186-216  triangulate
//   src/tracking/tracker.rs:1001-1010     BFMatcher(NORM_HAMMING, crossCheck=true).train_match
//
// Layout: descriptors are rows of 32 bytes read as 4 x u64 per lane; distances are
// 4 x (xor + v_bcnt via __popcll); best/second/index reductions run across the 64 lanes of a
// wave with __shfl_xor and reproduce the reference's sequential scan (strict '<': the lowest
// right index wins a tie, `second` is the second smallest of the multiset).  All of this is
// integer/byte work bound by L2/HBM reads, not by MFMA.
#include "orbx_internal.hpp"

namespace {

constexpr int kWave = 64;
constexpr int SM_THREADS = 256;
constexpr int SM_LEFT_PER_WAVE = 4;
constexpr int SM_LEFT_PER_BLOCK = SM_LEFT_PER_WAVE * (SM_THREADS / kWave);
constexpr int SM_RCHUNK = 2048;
constexpr unsigned TH_HIGH = 100;  // stereo.rs:10

struct Desc256 {
  unsigned long long w[4];
};

__device__ __forceinline__ Desc256 load_desc(const uint8_t* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  Desc256 d;
  d.w[0] = q[0]; d.w[1] = q[1]; d.w[2] = q[2]; d.w[3] = q[3];
  return d;
}
__device__ __forceinline__ unsigned hamming(const Desc256& a, const Desc256& b) {
  return (unsigned)(__popcll(a.w[0] ^ b.w[0]) + __popcll(a.w[1] ^ b.w[1]) +
                    __popcll(a.w[2] ^ b.w[2]) + __popcll(a.w[3] ^ b.w[3]));
}

// merge two (best, best_idx, second) triples: two smallest of the union, lowest index on ties
__device__ __forceinline__ void merge_top2(unsigned& b, int& bi, unsigned& s, unsigned ob, int obi,
                                           unsigned os) {
  const bool take = (ob < b) || (ob == b && obi < bi);
  const unsigned loser = take ? b : ob;
  if (take) { b = ob; bi = obi; }
  s = min(min(s, os), loser);
}

// One wave = SM_LEFT_PER_WAVE left keypoints; right (x,y) staged through LDS in chunks.
__global__ __launch_bounds__(SM_THREADS) void stereo_match_kernel(
    const orbx_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
    const int* __restrict__ nkp, int cap, float max_disp, float min_disp,
    int2* __restrict__ tmp) {
  __shared__ float2 sR[SM_RCHUNK];
  const int pair = blockIdx.y;
  const orbx_keypoint* kpL = kp + (size_t)(2 * pair) * cap;
  const orbx_keypoint* kpR = kpL + cap;
  const uint8_t* dL = desc + (size_t)(2 * pair) * cap * 32;
  const uint8_t* dR = dL + (size_t)cap * 32;
  const int nL = min(nkp[2 * pair], cap), nR = min(nkp[2 * pair + 1], cap);
  const int l0 = blockIdx.x * SM_LEFT_PER_BLOCK;
  if (l0 >= nL) return;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;

  float ul[SM_LEFT_PER_WAVE], vl[SM_LEFT_PER_WAVE], min_u[SM_LEFT_PER_WAVE], max_u[SM_LEFT_PER_WAVE];
  Desc256 dl[SM_LEFT_PER_WAVE];
  unsigned best[SM_LEFT_PER_WAVE], second[SM_LEFT_PER_WAVE];
  int bidx[SM_LEFT_PER_WAVE];
  bool valid[SM_LEFT_PER_WAVE];
#pragma unroll
  for (int q = 0; q < SM_LEFT_PER_WAVE; ++q) {
    const int li = l0 + wave * SM_LEFT_PER_WAVE + q;
    valid[q] = li < nL;
    const int lc = valid[q] ? li : 0;
    ul[q] = kpL[lc].x;
    vl[q] = kpL[lc].y;
    min_u[q] = fmaxf(ul[q] - max_disp, 0.0f);                             // stereo.rs:100
    const float lim = ((float)nR * ul[q]) / (float)nL;                    // stereo.rs:102
    max_u[q] = fminf(ul[q] - min_disp, lim);                              // stereo.rs:101
    dl[q] = load_desc(dL + (size_t)lc * 32);
    best[q] = TH_HIGH; second[q] = TH_HIGH; bidx[q] = 0x7fffffff;
  }

  for (int r0 = 0; r0 < nR; r0 += SM_RCHUNK) {
    const int cnt = min(SM_RCHUNK, nR - r0);
    __syncthreads();
    for (int i = tid; i < cnt; i += SM_THREADS) sR[i] = make_float2(kpR[r0 + i].x, kpR[r0 + i].y);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SM_LEFT_PER_WAVE; ++q) {
      if (!valid[q]) continue;  // wave-uniform
      for (int i = lane; i < cnt; i += kWave) {
        const float2 r = sR[i];
        if (fabsf(vl[q] - r.y) > 2.0f) continue;                          // stereo.rs:117
        if (r.x < min_u[q] || r.x > max_u[q]) continue;                   // stereo.rs:122
        if (ul[q] <= r.x) continue;                                       // stereo.rs:127
        const int ri = r0 + i;
        const unsigned d = hamming(dl[q], load_desc(dR + (size_t)ri * 32));
        if (d < best[q]) { second[q] = best[q]; best[q] = d; bidx[q] = ri; }   // stereo.rs:135-141
        else if (d < second[q]) { second[q] = d; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < SM_LEFT_PER_WAVE; ++q) {
    unsigned b = best[q], s = second[q];
    int bi = bidx[q];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned ob = __shfl_xor(b, off);
      const int obi = __shfl_xor(bi, off);
      const unsigned os = __shfl_xor(s, off);
      merge_top2(b, bi, s, ob, obi, os);
    }
    if (lane == 0 && valid[q]) {
      const int li = l0 + wave * SM_LEFT_PER_WAVE + q;
      const bool has = bi != 0x7fffffff;
      const bool emit = has && (((float)b < 0.9f * (float)s) || s == TH_HIGH);   // stereo.rs:145-148
      tmp[(size_t)pair * cap + li] = emit ? make_int2(bi, (int)b) : make_int2(-1, 0);
    }
  }
}

// One block per stereo pair: ordered compaction of the per-left results into DMatch rows
// (ascending query_idx, stereo.rs:149-156) and stereo triangulation (stereo.rs:186-216, f64).
__global__ __launch_bounds__(256) void stereo_compact_kernel(
    const orbx_keypoint* __restrict__ kp, const int* __restrict__ nkp, int cap,
    const int2* __restrict__ tmp, orbx_camera cam, orbx_dmatch* __restrict__ matches,
    int* __restrict__ nmatches, double* __restrict__ points, uint8_t* __restrict__ has_point) {
  __shared__ int wave_tot[4];
  __shared__ int running;
  const int pair = blockIdx.x;
  const orbx_keypoint* kpL = kp + (size_t)(2 * pair) * cap;
  const orbx_keypoint* kpR = kpL + cap;
  const int nL = min(nkp[2 * pair], cap);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < nL; base += 256) {
    const int li = base + tid;
    int2 t = make_int2(-1, 0);
    if (li < nL) t = tmp[(size_t)pair * cap + li];
    const bool flag = t.x >= 0;
    const unsigned long long m = __ballot(flag);
    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (li < nL) {
      uint8_t hp = 0;
      if (flag) {
        orbx_dmatch dm;
        dm.query_idx = li; dm.train_idx = t.x; dm.img_idx = 0; dm.distance = (float)t.y;
        matches[(size_t)pair * cap + off + prefix] = dm;
        const double lx = (double)kpL[li].x, ly = (double)kpL[li].y, rx = (double)kpR[t.x].x;
        const double disparity = lx - rx;                                 // stereo.rs:204
        if (!(fabs(disparity) < 0.5)) {                                   // stereo.rs:205
          const double z = cam.fx * cam.baseline / disparity;             // stereo.rs:208
          const double x = (lx - cam.cx) * z / cam.fx;                    // stereo.rs:209
          const double y = (ly - cam.cy) * z / cam.fy;                    // stereo.rs:210
          double* P = points + ((size_t)pair * cap + li) * 3;
          P[0] = x; P[1] = y; P[2] = z;
          hp = 1;
        }
      }
      has_point[(size_t)pair * cap + li] = hp;
    }
    __syncthreads();
    if (tid == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (tid == 0) nmatches[pair] = running;
}

// ---- brute-force nearest neighbour (one direction of the cross-check matcher) -----------------------
constexpr int NN_Q = 16;  // query rows per block, kept in LDS; every lane owns one train row per tile
__global__ __launch_bounds__(256) void nn_kernel(const uint8_t* __restrict__ q, int nq,
                                                 const uint8_t* __restrict__ t, int nt,
                                                 int* __restrict__ nn_idx, unsigned* __restrict__ nn_dist) {
  __shared__ unsigned long long sq[NN_Q][4];
  __shared__ unsigned long long red[4][NN_Q];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * NN_Q;
  if (tid < NN_Q * 4) {
    const int r = tid >> 2, c = tid & 3;
    const int qi = min(q0 + r, nq - 1);
    sq[r][c] = reinterpret_cast<const unsigned long long*>(q + (size_t)qi * 32)[c];
  }
  __syncthreads();
  unsigned bd[NN_Q];
  int bj[NN_Q];
#pragma unroll
  for (int r = 0; r < NN_Q; ++r) { bd[r] = 0xffffffffu; bj[r] = 0x7fffffff; }
  for (int j = tid; j < nt; j += 256) {
    const Desc256 tr = load_desc(t + (size_t)j * 32);
#pragma unroll
    for (int r = 0; r < NN_Q; ++r) {
      const unsigned d = (unsigned)(__popcll(tr.w[0] ^ sq[r][0]) + __popcll(tr.w[1] ^ sq[r][1]) +
                                    __popcll(tr.w[2] ^ sq[r][2]) + __popcll(tr.w[3] ^ sq[r][3]));
      if (d < bd[r]) { bd[r] = d; bj[r] = j; }   // ascending j per lane + strict '<' = first minimum
    }
  }
#pragma unroll
  for (int r = 0; r < NN_Q; ++r) {
    unsigned long long key = ((unsigned long long)bd[r] << 32) | (unsigned)bj[r];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(key, off);
      key = o < key ? o : key;
    }
    if (lane == 0) red[wave][r] = key;
  }
  __syncthreads();
  if (tid < NN_Q && q0 + tid < nq) {
    unsigned long long key = red[0][tid];
    for (int w = 1; w < 4; ++w) key = red[w][tid] < key ? red[w][tid] : key;
    nn_idx[q0 + tid] = (int)(unsigned)(key & 0xffffffffull);
    nn_dist[q0 + tid] = (unsigned)(key >> 32);
  }
}

// emit (i, fwd[i], d) iff bwd[fwd[i]] == i, ascending i — one block, ordered compaction
__global__ __launch_bounds__(256) void crosscheck_compact_kernel(const int* __restrict__ fwd,
                                                                 const unsigned* __restrict__ fdist,
                                                                 const int* __restrict__ bwd, int nq,
                                                                 orbx_dmatch* __restrict__ out,
                                                                 int* __restrict__ n_out) {
  __shared__ int wave_tot[4];
  __shared__ int running;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int base = 0; base < nq; base += 256) {
    const int i = base + tid;
    bool flag = false;
    int j = -1;
    if (i < nq) { j = fwd[i]; flag = bwd[j] == i; }
    const unsigned long long m = __ballot(flag);
    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (flag) {
      orbx_dmatch dm;
      dm.query_idx = i; dm.train_idx = j; dm.img_idx = 0; dm.distance = (float)fdist[i];
      out[off + prefix] = dm;
    }
    __syncthreads();
    if (tid == 0) running += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  if (tid == 0) *n_out = running;
}

__global__ __launch_bounds__(256) void hamming_batch_kernel(const uint8_t* __restrict__ a,
                                                            const uint8_t* __restrict__ b, int n,
                                                            uint32_t* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = hamming(load_desc(a + (size_t)i * 32), load_desc(b + (size_t)i * 32));
}

}  // namespace

int launch_stereo_match(orbx_handle* h, int batch, const orbx_keypoint* d_kp, const uint8_t* d_desc,
                        const int* d_nkp, int cap_kp, orbx_dmatch* d_matches, int* d_nmatches,
                        double* d_points, uint8_t* d_has_point) {
  if (batch <= 0) return ORBX_OK;
  if (int rc = orbx_reserve(h, h->ws_match, sizeof(int2) * (size_t)batch * cap_kp)) return rc;
  // stereo.rs:84-90: f64 product/quotient, then `as f32`
  const float max_disp = (float)(h->cam.fx * h->cam.baseline / 0.1);
  const float min_disp = (float)(h->cam.fx * h->cam.baseline / 40.0);
  {
    ProfScope ps(h, "stereo_match_kernel");
    dim3 grid((cap_kp + SM_LEFT_PER_BLOCK - 1) / SM_LEFT_PER_BLOCK, batch);
    hipLaunchKernelGGL(stereo_match_kernel, grid, dim3(SM_THREADS), 0, h->stream, d_kp, d_desc, d_nkp,
                       cap_kp, max_disp, min_disp, (int2*)h->ws_match.p);
  }
  {
    ProfScope ps(h, "stereo_compact_kernel");
    hipLaunchKernelGGL(stereo_compact_kernel, dim3(batch), dim3(256), 0, h->stream, d_kp, d_nkp, cap_kp,
                       (const int2*)h->ws_match.p, h->cam, d_matches, d_nmatches, d_points, d_has_point);
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_crosscheck(orbx_handle* h, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                      orbx_dmatch* d_out, int* d_n_out) {
  if (nq <= 0 || nt <= 0) {
    ORBX_HIP(h, hipMemsetAsync(d_n_out, 0, sizeof(int), h->stream));
    return ORBX_OK;
  }
  const size_t need = sizeof(int) * (size_t)(2 * nq + 2 * nt);
  if (int rc = orbx_reserve(h, h->ws_match, need)) return rc;
  int* fwd = (int*)h->ws_match.p;
  unsigned* fdist = (unsigned*)(fwd + nq);
  int* bwd = (int*)(fdist + nq);
  unsigned* bdist = (unsigned*)(bwd + nt);
  {
    ProfScope ps(h, "nn_kernel");
    hipLaunchKernelGGL(nn_kernel, dim3((nq + NN_Q - 1) / NN_Q), dim3(256), 0, h->stream, d_q, nq, d_t, nt, fwd, fdist);
    hipLaunchKernelGGL(nn_kernel, dim3((nt + NN_Q - 1) / NN_Q), dim3(256), 0, h->stream, d_t, nt, d_q, nq, bwd, bdist);
  }
  {
    ProfScope ps(h, "crosscheck_compact_kernel");
    hipLaunchKernelGGL(crosscheck_compact_kernel, dim3(1), dim3(256), 0, h->stream, fwd, fdist, bwd, nq, d_out, d_n_out);
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_hamming_batch(orbx_handle* h, const uint8_t* d_a, const uint8_t* d_b, int n, uint32_t* d_out) {
  if (n <= 0) return ORBX_OK;
  ProfScope ps(h, "hamming_batch_kernel");
  const int blocks = min((n + 255) / 256, 2048);
  hipLaunchKernelGGL(hamming_batch_kernel, dim3(blocks), dim3(256), 0, h->stream, d_a, d_b, n, d_out);
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}
