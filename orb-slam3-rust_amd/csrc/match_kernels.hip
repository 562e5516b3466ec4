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
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "orbx_internal.hpp"

namespace {

constexpr int kWave = 64;
constexpr int SM_THREADS = 256;
#ifndef ORBX_SM_LPW
#define ORBX_SM_LPW 8
#endif
constexpr int SM_LEFT_PER_WAVE = ORBX_SM_LPW;   // multiple of 4: four left keypoints share a wave
constexpr int SM_LEFT_PER_BLOCK = SM_LEFT_PER_WAVE * (SM_THREADS / kWave);
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

// --- stereo matcher, bucketed ------------------------------------------------------------------------
// The reference scans every right keypoint for every left one (stereo.rs:95-141); ~99 % fail the
// vertical gate |vl - vr| <= 2 (:117).  Here the right keypoints of a pair are first counting-sorted by
// image row (stereo_bucket_kernel, one block per pair, histogram + scan + scatter in LDS); a left
// keypoint then examines only the rows floor(vl)-3 .. floor(vl)+3 and applies the reference's exact f32
// gates to those.  The result is the same set of admissible candidates; best / second / best_idx are
// order-independent functions of that set (two smallest distances of the multiset, lowest right index
// among the minima — what the sequential strict '<' scan yields), so the matches are bit-identical.
// Keypoint coordinates must be finite.
constexpr int SB_ROWS = 4096;            // row buckets (images are at most 4095 rows, orbx_create)
constexpr int SB_THREADS = 1024;

__device__ __forceinline__ int row_bucket(float v) {
  return v >= 0.0f ? (v < (float)(SB_ROWS - 1) ? (int)v : SB_ROWS - 1) : 0;
}

__global__ __launch_bounds__(SB_THREADS) void stereo_bucket_kernel(const orbx_keypoint* __restrict__ kp,
                                                                   const int* __restrict__ nkp, int cap,
                                                                   int* __restrict__ bstart /*[pair][SB_ROWS+1]*/,
                                                                   int* __restrict__ sidx /*[pair][cap]*/,
                                                                   float2* __restrict__ sxy /*[pair][cap]*/) {
  __shared__ int cnt[SB_ROWS];
  __shared__ int wsum[SB_THREADS / 64];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const orbx_keypoint* kpR = kp + (size_t)(2 * pair + 1) * cap;
  const int nR = min(nkp[2 * pair + 1], cap);
  for (int i = tid; i < SB_ROWS; i += SB_THREADS) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < nR; i += SB_THREADS) atomicAdd(&cnt[row_bucket(kpR[i].y)], 1);
  __syncthreads();
  // exclusive scan of 4096 counters: 4 per thread, wave scan, cross-wave offsets
  int c[4], tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { c[k] = cnt[4 * tid + k]; tot += c[k]; }
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int wv = 0; wv < wave; ++wv) base += wsum[wv];
  __syncthreads();
  int* bs = bstart + (size_t)pair * (SB_ROWS + 1);
#pragma unroll
  for (int k = 0; k < 4; ++k) { bs[4 * tid + k] = base; cnt[4 * tid + k] = base; base += c[k]; }
  if (tid == SB_THREADS - 1) bs[SB_ROWS] = base;
  __syncthreads();
  for (int i = tid; i < nR; i += SB_THREADS) {
    const float x = kpR[i].x, y = kpR[i].y;
    const int pos = atomicAdd(&cnt[row_bucket(y)], 1);     // order inside a bucket is irrelevant
    sidx[(size_t)pair * cap + pos] = i;
    sxy[(size_t)pair * cap + pos] = make_float2(x, y);
  }
}

// candidate update that does not depend on visiting order
__device__ __forceinline__ void push_top2(unsigned& b, int& bi, unsigned& s, unsigned d, int ri) {
  if (d < b || (d == b && ri < bi)) { s = b; b = d; bi = ri; }
  else s = min(s, d);
}

// Sixteen lanes per left keypoint, four left keypoints per wave at a time (a left keypoint has ~30 right keypoints
// in its 7 rows, ~10 of them inside the gates: a whole wave per keypoint left most lanes idle and paid a 6-step
// reduction); SM_LEFT_PER_WAVE keypoints per wave in turn.
__global__ __launch_bounds__(SM_THREADS) void stereo_match_kernel(
    const orbx_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
    const int* __restrict__ nkp, int cap, float max_disp, float min_disp,
    const int* __restrict__ bstart, const int* __restrict__ sidx, const float2* __restrict__ sxy,
    int2* __restrict__ tmp) {
  const int pair = blockIdx.y;
  const orbx_keypoint* kpL = kp + (size_t)(2 * pair) * cap;
  const uint8_t* dL = desc + (size_t)(2 * pair) * cap * 32;
  const uint8_t* dR = dL + (size_t)cap * 32;
  const int nL = min(nkp[2 * pair], cap), nR = min(nkp[2 * pair + 1], cap);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  const int* bs = bstart + (size_t)pair * (SB_ROWS + 1);
  const int* si = sidx + (size_t)pair * cap;
  const float2* sx = sxy + (size_t)pair * cap;
  // The kernel is a chain of dependent memory round trips per left keypoint (keypoint -> row buckets -> candidate position -> its index
  // -> its descriptor), a few dozen waves per SIMD living through it one after the other: what it costs is the LENGTH of that chain.
  // So: the keypoints, descriptors and bucket bounds of ALL of a wave's rounds are requested before the first is used; a candidate's
  // index travels with its position (both are functions of t), not after the gates; and two candidates per lane (t, t + 16) are in
  // flight at a time, their descriptors requested together.  (Round 2's form took 8 round trips per round of 4 keypoints, two rounds
  // per wave in turn: 0.130 ms per 256 pairs.  Round 4: the right descriptors copied into bucket order by stereo_bucket_kernel, so that a
  // group's sixteen lanes read adjacent descriptors and the index leaves the address chain: this kernel 0.0895 -> 0.0865 ms, the bucket
  // kernel 0.011 -> 0.022 — withdrawn.)
  constexpr int NR = SM_LEFT_PER_WAVE / 4;
  if (nL <= 0) return;
  int li_[NR]; bool live[NR]; float ul_[NR], vl_[NR]; Desc256 dl_[NR]; int lo_[NR], hi_[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    li_[r] = blockIdx.x * SM_LEFT_PER_BLOCK + wave * SM_LEFT_PER_WAVE + 4 * r + grp;
    live[r] = li_[r] < nL;                                                   // uniform over the 16-lane group
    const int lc = live[r] ? li_[r] : 0;
    ul_[r] = kpL[lc].x; vl_[r] = kpL[lc].y;
    dl_[r] = load_desc(dL + (size_t)lc * 32);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int rb = row_bucket(vl_[r]);
    lo_[r] = bs[max(rb - 3, 0)]; hi_[r] = live[r] ? bs[min(rb + 3, SB_ROWS - 1) + 1] : 0;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const float ul = ul_[r], vl = vl_[r];
    const float min_u = fmaxf(ul - max_disp, 0.0f);                          // stereo.rs:100
    const float lim = ((float)nR * ul) / (float)nL;                          // stereo.rs:102
    const float max_u = fminf(ul - min_disp, lim);                           // stereo.rs:101
    const int hi = hi_[r];
    unsigned b = TH_HIGH, s = TH_HIGH;
    int bi = 0x7fffffff;
    auto gates = [&](const float2& rr) -> bool {
      return !(fabsf(vl - rr.y) > 2.0f) &&                                   // stereo.rs:117
             !(rr.x < min_u || rr.x > max_u) &&                              // stereo.rs:122
             !(ul <= rr.x);                                                  // stereo.rs:127
    };
    for (int t = lo_[r] + gl; t < hi; t += 32) {
      const int t2 = t + 16;
      const bool in2 = t2 < hi;
      const float2 r1 = sx[t];
      const int i1 = si[t];
      const float2 r2 = sx[in2 ? t2 : t];
      const int i2 = si[in2 ? t2 : t];
      const bool p1 = gates(r1), p2 = in2 && gates(r2);
      Desc256 d1, d2;
      if (p1) d1 = load_desc(dR + (size_t)i1 * 32);
      if (p2) d2 = load_desc(dR + (size_t)i2 * 32);
      if (p1) { const unsigned d = hamming(dl_[r], d1); if (d < TH_HIGH) push_top2(b, bi, s, d, i1); }   // stereo.rs:132-141 (d >= 100 never enters)
      if (p2) { const unsigned d = hamming(dl_[r], d2); if (d < TH_HIGH) push_top2(b, bi, s, d, i2); }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {                                 // xor < 16 stays inside the group
      const unsigned ob = __shfl_xor(b, off);
      const int obi = __shfl_xor(bi, off);
      const unsigned os = __shfl_xor(s, off);
      merge_top2(b, bi, s, ob, obi, os);
    }
    if (gl == 0 && live[r]) {
      const bool has = bi != 0x7fffffff;
      const bool emit = has && (((float)b < 0.9f * (float)s) || s == TH_HIGH);   // stereo.rs:145-148
      tmp[(size_t)pair * cap + li_[r]] = emit ? make_int2(bi, (int)b) : make_int2(-1, 0);
    }
  }
}

// The same matcher with the pair's right image resident in LDS (round 5): one 1024-thread workgroup per stereo pair copies the bucket-ordered right
// keypoints — position, index, and the DESCRIPTORS gathered into bucket order — and the row-bucket bounds into LDS (about 107 KB at 2064 keypoints),
// then walks the left keypoints 64 per round (sixteen lanes each, as above).  What stereo_match_kernel pays per left keypoint is a chain of six
// dependent round trips to L2; here the chain runs on LDS, the right descriptors leave HBM once per pair (the line-granular gathers of the first
// form fetched 4.9 times the compulsory bytes), and the group reduction is four DPP row rotations instead of twelve LDS swizzles.  Same candidate
// sets, same order-independent top-2: the matches are bit-identical (tests/test_matcher_gpu.py, test_extract_gpu.py).  Large batches only: a pair
// occupies ONE CU for its ~31 rounds, so a call needs at least as many pairs as CUs to fill the chip (launch_stereo_match_range picks).
constexpr int SML_THREADS = 1024;
#ifndef ORBX_SML_LPK
#define ORBX_SML_LPK 4
#endif
// lanes per left keypoint: with LDS latencies there is no chain of L2 round trips to spread over sixteen lanes, and what a round costs beside its
// candidates (the group reduction, the ratio test, the next keypoint's loads) is paid per group — per 512 pairs, same box: stereo_match_kernel 0.1755 ms;
// this kernel with 16 / 8 / 4 lanes per keypoint and gates + distance in one loop body 0.1455 / 0.1206 / 0.1213; with the gates into a bit mask first
// and the row range of vl -+ 2.01 instead of seven rows 0.1428 / 0.1110 / 0.1050 (profiles/r05_stereo_match_lds.txt)
constexpr int SML_LPK = ORBX_SML_LPK;
constexpr int SML_LEFT_PER_ROUND = SML_THREADS / SML_LPK;
static_assert(SML_LPK == 4 || SML_LPK == 8 || SML_LPK == 16, "lanes per left keypoint");
template <int CTRL>
__device__ __forceinline__ void top2_dpp_step(unsigned& b, int& bi, unsigned& s) {
  const unsigned ob = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int obi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false);
  const unsigned os = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s, CTRL, 0xf, 0xf, false);
  merge_top2(b, bi, s, ob, obi, os);
}
// FUSED: the whole of match_features + triangulate for the pair in this one workgroup — the counting sort of the right keypoints by row (what
// stereo_bucket_kernel does) straight into the LDS arrays, the per-left-keypoint results in LDS, and the ordered compaction + f64 triangulation (what
// stereo_compact_kernel does) behind the last round: one launch instead of three, the bucket tables and the (best, distance) pairs never leave the CU.
// 52 B per keypoint + 32 KB of LDS (140 KB at 2064 keypoints); above that the three-launch form with this kernel in the middle.
template <bool FUSED>
__global__ __launch_bounds__(SML_THREADS) void stereo_match_lds_kernel(
    const orbx_keypoint* __restrict__ kp, const uint8_t* __restrict__ desc,
    const int* __restrict__ nkp, int cap, float max_disp, float min_disp,
    const int* __restrict__ bstart, const int* __restrict__ sidx, const float2* __restrict__ sxy,
    int2* __restrict__ tmp, orbx_camera cam, orbx_dmatch* __restrict__ matches,
    int* __restrict__ nmatches, double* __restrict__ points, uint8_t* __restrict__ has_point) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sml_smem[];
  uint4* s_desc = reinterpret_cast<uint4*>(sml_smem);                                  // [cap][2]
  float2* s_xy = reinterpret_cast<float2*>(sml_smem + (size_t)cap * 32);               // [cap]
  int* s_idx = reinterpret_cast<int*>(sml_smem + (size_t)cap * 40);                    // [cap]
  int* s_bs = s_idx + cap;                                                             // [SB_ROWS + 1]
  int* s_cur = s_bs + (SB_ROWS + 1);                                                   // FUSED: [SB_ROWS] counters / cursors, then the wave totals of the compaction
  int2* s_tmp = reinterpret_cast<int2*>(sml_smem + (((size_t)cap * 44 + (size_t)(2 * SB_ROWS + 1) * 4 + 7) & ~(size_t)7));   // FUSED: [cap]
  const int pair = blockIdx.x;
  const orbx_keypoint* kpL = kp + (size_t)(2 * pair) * cap;
  const uint8_t* dL = desc + (size_t)(2 * pair) * cap * 32;
  const uint8_t* dR = dL + (size_t)cap * 32;
  const int nL = min(nkp[2 * pair], cap), nR = min(nkp[2 * pair + 1], cap);
  const int tid = threadIdx.x;
  const int gl = tid & (SML_LPK - 1);
  const int* bs = bstart + (size_t)pair * (SB_ROWS + 1);
  const int* si = sidx + (size_t)pair * cap;
  const float2* sx = sxy + (size_t)pair * cap;
  if (nL <= 0) {
    if (FUSED && tid == 0) nmatches[pair] = 0;
    return;
  }
  // the first round's left keypoint travels under the staging
  int li = tid / SML_LPK;
  bool live = li < nL;
  float2 uv = make_float2(kpL[live ? li : 0].x, kpL[live ? li : 0].y);
  Desc256 dl = load_desc(dL + (size_t)(live ? li : 0) * 32);
  if (FUSED) {
    // counting sort of the right keypoints by image row (stereo_bucket_kernel's steps; the order inside a bucket is what the atomics give)
    const orbx_keypoint* kpR = kpL + cap;
    static_assert(SB_ROWS == 4 * SML_THREADS, "four counters per thread in the scan");
    const int lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SB_ROWS; i += SML_THREADS) s_cur[i] = 0;
    __syncthreads();
    for (int i = tid; i < nR; i += SML_THREADS) atomicAdd(&s_cur[row_bucket(kpR[i].y)], 1);
    __syncthreads();
    int c[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = s_cur[4 * tid + k]; tot += c[k]; }
    int inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
    int* wsum = reinterpret_cast<int*>(s_tmp);                                          // (free until the rounds write their results)
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - tot;
    for (int wv = 0; wv < wave; ++wv) base += wsum[wv];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_bs[4 * tid + k] = base; s_cur[4 * tid + k] = base; base += c[k]; }
    if (tid == SML_THREADS - 1) s_bs[SB_ROWS] = base;
    __syncthreads();
    for (int i = tid; i < nR; i += SML_THREADS) {
      const float x = kpR[i].x, y = kpR[i].y;
      const uint4* g = reinterpret_cast<const uint4*>(dR + (size_t)i * 32);
      const uint4 d0 = g[0], d1 = g[1];
      const int pos = atomicAdd(&s_cur[row_bucket(y)], 1);
      s_idx[pos] = i; s_xy[pos] = make_float2(x, y);
      s_desc[2 * pos] = d0; s_desc[2 * pos + 1] = d1;
    }
  } else {
    for (int i = tid; i < nR; i += SML_THREADS) {
      const int ix = si[i];
      const uint4* g = reinterpret_cast<const uint4*>(dR + (size_t)ix * 32);
      const uint4 d0 = g[0], d1 = g[1];
      s_xy[i] = sx[i]; s_idx[i] = ix;
      s_desc[2 * i] = d0; s_desc[2 * i + 1] = d1;
    }
    for (int i = tid; i <= SB_ROWS; i += SML_THREADS) s_bs[i] = bs[i];
  }
  __syncthreads();
  for (;;) {
    // the next round's left keypoint
    const int li_n = li + SML_LEFT_PER_ROUND;
    const bool live_n = li_n < nL;
    const float2 uv_n = make_float2(kpL[live_n ? li_n : 0].x, kpL[live_n ? li_n : 0].y);
    const Desc256 dl_n = load_desc(dL + (size_t)(live_n ? li_n : 0) * 32);
    const float ul = uv.x, vl = uv.y;
    // row buckets that can hold a candidate of the vertical gate: |vl - vr| <= 2 in f32 means vr within 2.0000003 of vl, so the buckets of
    // vl -+ 2.01 cover it (five or six rows; stereo_match_kernel scans seven) — the gates below still decide
    const int lo = s_bs[row_bucket(vl - 2.01f)], hi = live ? s_bs[row_bucket(vl + 2.01f) + 1] : 0;
    const float min_u = fmaxf(ul - max_disp, 0.0f);                          // stereo.rs:100
    const float lim = ((float)nR * ul) / (float)nL;                          // stereo.rs:102
    const float max_u = fminf(ul - min_disp, lim);                           // stereo.rs:101
    unsigned b = TH_HIGH, s = TH_HIGH;
    int bi = 0x7fffffff;
    auto gates = [&](const float2& rr) -> bool {
      return !(fabsf(vl - rr.y) > 2.0f) &&                                   // stereo.rs:117
             !(rr.x < min_u || rr.x > max_u) &&                              // stereo.rs:122
             !(ul <= rr.x);                                                  // stereo.rs:127
    };
    auto visit = [&](int t) {
      const uint4 a0 = s_desc[2 * t], a1 = s_desc[2 * t + 1];
      Desc256 d1;
      d1.w[0] = a0.x | ((unsigned long long)a0.y << 32); d1.w[1] = a0.z | ((unsigned long long)a0.w << 32);
      d1.w[2] = a1.x | ((unsigned long long)a1.y << 32); d1.w[3] = a1.z | ((unsigned long long)a1.w << 32);
      const unsigned d = hamming(dl, d1);
      if (d < TH_HIGH) push_top2(b, bi, s, d, s_idx[t]);                      // stereo.rs:132-141 (d >= 100 never enters)
    };
    // Two steps per 32 candidates of a lane: the gates of all of them into a bit mask (a third pass), then the descriptors of the set bits only —
    // with the gates and the distance in one loop body every lane of the wave pays for a distance whenever any lane's candidate passes
    for (int base = lo + gl; base < hi; base += 32 * SML_LPK) {
      unsigned mask = 0;
      int t = base;
      for (int k = 0; k < 32 && t < hi; ++k, t += SML_LPK) mask |= gates(s_xy[t]) ? 1u << k : 0u;
      while (mask) {
        const int k = __ffs((int)mask) - 1;
        mask &= mask - 1u;
        visit(base + SML_LPK * k);
      }
    }
    // all-reduce over the group's lanes on the DPP path (the sets merged at every step are disjoint)
    if (SML_LPK == 16) { top2_dpp_step<0x128>(b, bi, s); top2_dpp_step<0x124>(b, bi, s); top2_dpp_step<0x122>(b, bi, s); top2_dpp_step<0x121>(b, bi, s); }   // row_ror 8, 4, 2, 1
    if (SML_LPK == 8) top2_dpp_step<0x141>(b, bi, s);                          // row_half_mirror: i <-> 7 - i
    if (SML_LPK <= 8) { top2_dpp_step<0xB1>(b, bi, s); top2_dpp_step<0x4E>(b, bi, s); }   // quad_perm [1,0,3,2], [2,3,0,1]
    if (gl == 0 && live) {
      const bool has = bi != 0x7fffffff;
      const bool emit = has && (((float)b < 0.9f * (float)s) || s == TH_HIGH);   // stereo.rs:145-148
      const int2 res = emit ? make_int2(bi, (int)b) : make_int2(-1, 0);
      if (FUSED) s_tmp[li] = res; else tmp[(size_t)pair * cap + li] = res;
    }
    if (li_n - (li_n & (SML_LEFT_PER_ROUND - 1)) >= nL) break;                  // block-uniform: the next round holds no live group
    li = li_n; live = live_n; uv = uv_n; dl = dl_n;
  }
  if (!FUSED) return;
  // ordered compaction of the matches and the f64 triangulation (stereo_compact_kernel's steps over 1024 threads)
  {
    const orbx_keypoint* kpR = kpL + cap;
    const int lane = tid & 63, wave = tid >> 6;
    int* wave_tot = s_cur;                                                               // (the cursors are spent)
    int running = 0;
    __syncthreads();
    for (int base = 0; base < nL; base += SML_THREADS) {
      const int lq = base + tid;
      int2 t = make_int2(-1, 0);
      if (lq < nL) t = s_tmp[lq];
      const bool flag = t.x >= 0;
      const unsigned long long m = __ballot(flag);
      const int prefix = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_tot[wave] = __popcll(m);
      __syncthreads();
      int off = running, all = 0;
      for (int w = 0; w < SML_THREADS / 64; ++w) { const int wt = wave_tot[w]; off += w < wave ? wt : 0; all += wt; }
      if (lq < nL) {
        uint8_t hp = 0;
        if (flag) {
          orbx_dmatch dm;
          dm.query_idx = lq; dm.train_idx = t.x; dm.img_idx = 0; dm.distance = (float)t.y;
          matches[(size_t)pair * cap + off + prefix] = dm;
          const double lx = (double)kpL[lq].x, ly = (double)kpL[lq].y, rx = (double)kpR[t.x].x;
          const double disparity = lx - rx;                                 // stereo.rs:204
          if (!(fabs(disparity) < 0.5)) {                                   // stereo.rs:205
            const double z = cam.fx * cam.baseline / disparity;             // stereo.rs:208
            const double x = (lx - cam.cx) * z / cam.fx;                    // stereo.rs:209
            const double y = (ly - cam.cy) * z / cam.fy;                    // stereo.rs:210
            double* P = points + ((size_t)pair * cap + lq) * 3;
            P[0] = x; P[1] = y; P[2] = z;
            hp = 1;
          }
        }
        if (!hp) {   // None: defined contents (zeros) so that outputs are reproducible byte for byte
          double* P = points + ((size_t)pair * cap + lq) * 3;
          P[0] = 0.0; P[1] = 0.0; P[2] = 0.0;
        }
        has_point[(size_t)pair * cap + lq] = hp;
      }
      running += all;
      __syncthreads();
    }
    if (tid == 0) nmatches[pair] = running;
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
      if (!hp) {   // None: defined contents (zeros) so that outputs are reproducible byte for byte
        double* P = points + ((size_t)pair * cap + li) * 3;
        P[0] = 0.0; P[1] = 0.0; P[2] = 0.0;
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

// ---- guided matching: FeatureGrid (tracking_frame.rs:52-128) + the tracker's two search rules -----------------
constexpr int GG_COLS = 64, GG_ROWS = 48, GG_CELLS = GG_COLS * GG_ROWS;   // tracking_frame.rs:43-44

// Rust `f64 as usize`: truncation, negative / NaN -> 0
__device__ __forceinline__ int sat_cell(double v, int last) {
  return (v > 0.0) ? (v >= (double)(last + 1) ? last : (int)v) : 0;
}
// Rust `f64 as i32`: saturating, NaN -> 0
__device__ __forceinline__ int sat_i32(double v) {
  if (v != v) return 0;
  if (v <= -2147483648.0) return INT_MIN;
  if (v >= 2147483647.0) return INT_MAX;
  return (int)v;
}

// One block: counting sort of the keypoints by grid cell (CSR: cell_start[GG_CELLS+1], sorted_idx[n]) and the
// cell of every keypoint.  The order inside a cell is irrelevant: ties are broken on (cell, index) explicitly.
__global__ __launch_bounds__(1024) void grid_build_kernel(const orbx_keypoint* __restrict__ kp, int n, double winv, double hinv,
                                                          int* __restrict__ cell_start, int* __restrict__ sorted_idx,
                                                          unsigned short* __restrict__ cell_of) {
  __shared__ int cnt[GG_CELLS];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < GG_CELLS; i += 1024) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int cx = sat_cell(((double)kp[i].x - 0.0) * winv, GG_COLS - 1);     // tracking_frame.rs:66-75
    const int cy = sat_cell(((double)kp[i].y - 0.0) * hinv, GG_ROWS - 1);
    const int c = cy * GG_COLS + cx;
    cell_of[i] = (unsigned short)c;
    atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  int c3[3], tot = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { c3[k] = cnt[3 * tid + k]; tot += c3[k]; }
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int wv = 0; wv < wave; ++wv) base += wsum[wv];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) { cell_start[3 * tid + k] = base; cnt[3 * tid + k] = base; base += c3[k]; }
  if (tid == 1023) cell_start[GG_CELLS] = base;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) sorted_idx[atomicAdd(&cnt[cell_of[i]], 1)] = i;
}

// One wave per query.  key = (distance << 48 | cell << 32 | index): the minimum key is the smallest distance,
// and among equal distances the first candidate in the reference's visiting order (cells row-major, indices
// ascending inside a cell).  `second` is the second smallest distance of the multiset.
__global__ __launch_bounds__(256) void guided_match_kernel(const uint8_t* __restrict__ desc, const int* __restrict__ cell_start,
                                                           const int* __restrict__ sorted_idx,
                                                           const unsigned short* __restrict__ cell_of,
                                                           const double* __restrict__ q_uv, const uint8_t* __restrict__ q_desc,
                                                           int nq, double radius, double winv, double hinv, int mode,
                                                           int* __restrict__ out_idx, uint32_t* __restrict__ out_dist) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const double x = q_uv[2 * (size_t)q], y = q_uv[2 * (size_t)q + 1];
  // tracking_frame.rs:107-117, including `(max as usize).min(cols - 1)`: a negative max wraps -> last cell
  const int mnx = sat_i32(floor((x - 0.0 - radius) * winv)), mxx = sat_i32(ceil((x - 0.0 + radius) * winv));
  const int mny = sat_i32(floor((y - 0.0 - radius) * hinv)), mxy = sat_i32(ceil((y - 0.0 + radius) * hinv));
  const int x0 = max(mnx, 0), y0 = max(mny, 0);
  const int x1 = (mxx < 0 || mxx > GG_COLS - 1) ? GG_COLS - 1 : mxx;
  const int y1 = (mxy < 0 || mxy > GG_ROWS - 1) ? GG_ROWS - 1 : mxy;
  const Desc256 dq = load_desc(q_desc + (size_t)q * 32);
  unsigned long long bk = ~0ull;
  unsigned s = 0xffffffffu;
  int total = 0;
  if (x0 <= x1) {
    for (int cy = y0; cy <= y1; ++cy) {
      const int lo = cell_start[cy * GG_COLS + x0], hi = cell_start[cy * GG_COLS + x1 + 1];
      total += hi - lo;
      for (int t = lo + lane; t < hi; t += kWave) {
        const int i = sorted_idx[t];
        const unsigned d = hamming(dq, load_desc(desc + (size_t)i * 32));
        if (mode == 0 && d >= TH_HIGH) continue;                          // tracker.rs:1146
        const unsigned long long key = ((unsigned long long)d << 48) | ((unsigned long long)cell_of[i] << 32) | (unsigned)i;
        if (key < bk) { s = min(s, (unsigned)(bk >> 48)); if (bk == ~0ull) s = 0xffffffffu; bk = key; }
        else s = min(s, d);
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long ok = __shfl_xor(bk, off);
    const unsigned os = __shfl_xor(s, off);
    const unsigned long long mn = ok < bk ? ok : bk, mx = ok < bk ? bk : ok;
    unsigned ns = min(s, os);
    if (mx != ~0ull) ns = min(ns, (unsigned)(mx >> 48));
    bk = mn; s = ns;
  }
  if (lane == 0) {
    int res = -1;
    unsigned rd = 0;
    if (bk != ~0ull) {
      const unsigned best = (unsigned)(bk >> 48);
      const int bi = (int)(unsigned)(bk & 0xffffffffull);
      if (mode == 0) { res = bi; rd = best; }
      else if (total > 0 && best <= TH_HIGH &&                                        // tracker.rs:884-886, :907-909
               !(total > 1 && (float)best > 0.75f * (float)s)) { res = bi; rd = best; }   // :911-915
    }
    out_idx[q] = res;
    out_dist[q] = rd;
  }
}

// ---- search_for_triangulation (src/local_mapping/triangulation.rs:339-527, 661-705) -----------------------------
// Epipolar-gated, grid-limited, greedy one-to-one matching between two keyframes.  The reference walks the
// features of keyframe 1 in index order and removes every matched feature of keyframe 2 from later searches
// (a serial dictatorship).  Exact parallel form: (1) every feature of keyframe 1 proposes its best admissible
// partner in parallel, ignoring the others; (2) one block walks the proposals in index order — a proposal whose
// partner is still free is final; one whose partner was taken by an earlier feature is recomputed cooperatively
// by the whole block against the current `taken` set, then the walk continues.  Only conflicts cost extra work.
struct TriArgs {
  double F[9];                 // fundamental matrix, row-major (host, triangulation.rs:670-683)
  double epx, epy;             // epipole of camera 1 in image 2 (:418-426)
  int cols, rows;              // 32-px grid over image 2 (:339-346, :437-438)
  unsigned max_dist;
  int n1, n2;
  const orbx_keypoint* kp1; const uint8_t* desc1; const uint8_t* mp1; const uint8_t* stereo1;
  const orbx_keypoint* kp2; const uint8_t* desc2;
  const int* cell_start; const int* sorted_idx; const unsigned short* cell_of;
  uint8_t* taken;              // per feature of keyframe 2: has a map point (mp2) or has been matched
  // FeatureVector mode (search_for_triangulation_bow, :541-658): candidates of feature i1 are sorted_idx[rng_lo[i1] ..
  // rng_hi[i1]) = the features of keyframe 2 in the same vocabulary node, ascending; no grid (cell_of == nullptr)
  const int* rng_lo; const int* rng_hi;
};

__device__ __forceinline__ int f32_as_cell(float v, int last) {          // Rust `f32 as usize` then .min(last)
  return (v > 0.0f) ? (v >= (float)(last + 1) ? last : (int)v) : 0;
}

__global__ __launch_bounds__(1024) void tri_grid_build_kernel(const orbx_keypoint* __restrict__ kp, int n, int cols, int rows,
                                                              const uint8_t* __restrict__ mp2, int* __restrict__ cell_start,
                                                              int* __restrict__ sorted_idx, unsigned short* __restrict__ cell_of,
                                                              uint8_t* __restrict__ taken) {
  __shared__ int cnt[4096];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncell = cols * rows;
  for (int i = tid; i < 4096; i += 1024) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int c = f32_as_cell(kp[i].x / 32.0f, cols - 1), r = f32_as_cell(kp[i].y / 32.0f, rows - 1);   // :357-358
    const int cell = r * cols + c;
    cell_of[i] = (unsigned short)cell;
    taken[i] = mp2[i];
    atomicAdd(&cnt[cell], 1);
  }
  __syncthreads();
  int c4[4], tot = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { c4[k] = cnt[4 * tid + k]; tot += c4[k]; }
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off); if (lane >= off) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int wv = 0; wv < wave; ++wv) base += wsum[wv];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (4 * tid + k <= ncell) cell_start[4 * tid + k] = base; cnt[4 * tid + k] = base; base += c4[k]; }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) sorted_idx[atomicAdd(&cnt[cell_of[i]], 1)] = i;
}

// best admissible partner of feature i1 over the candidates this thread visits (t, t+stride, ...):
// key = (distance << 48 | cell << 32 | index), ~0 = none.  Ties go to the first candidate in the reference's
// visiting order (cells row-major, indices ascending inside a cell).
template <typename TakenPtr>
__device__ __forceinline__ unsigned long long tri_scan(const TriArgs& A, TakenPtr taken, int i1, int t, int stride) {
  const float x = A.kp1[i1].x, y = A.kp1[i1].y;
  unsigned long long best = ~0ull;
  const double p0 = (double)x, p1 = (double)y;
  const double l0 = A.F[0] * p0 + A.F[1] * p1 + A.F[2] * 1.0;           // l2 = F p1 (:685-687)
  const double l1 = A.F[3] * p0 + A.F[4] * p1 + A.F[5] * 1.0;
  const double l2 = A.F[6] * p0 + A.F[7] * p1 + A.F[8] * 1.0;
  const double den = sqrt(l0 * l0 + l1 * l1);                            // :692
  if (den < 1e-10) return best;                                          // :694-696
  const bool mono = A.stereo1[i1] == 0;
  const Desc256 d1 = load_desc(A.desc1 + (size_t)i1 * 32);
  auto visit = [&](int lo, int hi) {
    for (int p = lo + t; p < hi; p += stride) {
      const int i2 = A.sorted_idx[p];
      if (taken[i2]) continue;                                           // :479-481
      const double x2 = (double)A.kp2[i2].x, y2 = (double)A.kp2[i2].y;
      if (mono) {                                                        // :489-497
        const double dx = A.epx - x2, dy = A.epy - y2;
        if (dx * dx + dy * dy < 100.0) continue;
      }
      const double num = fabs(l0 * x2 + l1 * y2 + l2 * 1.0);             // :691
      const double dl = num / den;
      if (!(dl * dl < 3.84)) continue;                                   // :698-702
      const unsigned d = hamming(d1, load_desc(A.desc2 + (size_t)i2 * 32));
      if (d >= A.max_dist) continue;                                     // :516 with best_dist starting at max_dist
      const unsigned cell = A.cell_of ? A.cell_of[i2] : 0u;
      const unsigned long long key = ((unsigned long long)d << 48) | ((unsigned long long)cell << 32) | (unsigned)i2;
      best = key < best ? key : best;
    }
  };
  if (A.rng_lo) {                                                        // same vocabulary node (:577-581)
    visit(A.rng_lo[i1], A.rng_hi[i1]);
    return best;
  }
  const int c0 = f32_as_cell(fmaxf(floorf((x - 100.0f) / 32.0f), 0.0f), 1 << 30);   // :376-379, radius 100 (:442)
  const int c1 = min(f32_as_cell(ceilf((x + 100.0f) / 32.0f), 1 << 30), A.cols - 1);
  const int r0 = f32_as_cell(fmaxf(floorf((y - 100.0f) / 32.0f), 0.0f), 1 << 30);
  const int r1 = min(f32_as_cell(ceilf((y + 100.0f) / 32.0f), 1 << 30), A.rows - 1);
  if (c0 > c1) return best;
  for (int r = r0; r <= r1; ++r) visit(A.cell_start[r * A.cols + c0], A.cell_start[r * A.cols + c1 + 1]);
  return best;
}

__global__ __launch_bounds__(256) void tri_propose_kernel(TriArgs A, int* __restrict__ prop) {
  const int lane = threadIdx.x & 63;
  const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i1 >= A.n1) return;
  if (A.mp1[i1]) { if (lane == 0) prop[i1] = -1; return; }               // :449-452
  unsigned long long key = tri_scan(A, (const uint8_t*)A.taken, i1, lane, kWave);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(key, off); key = o < key ? o : key; }
  if (lane == 0) prop[i1] = key == ~0ull ? -1 : (int)(unsigned)(key & 0xffffffffull);
}

// Ordered resolve.  A round looks at the next 256 features at once: a proposal is safe when its partner is free
// and no earlier feature of the round wants the same partner (owner[] = smallest proposing index, atomicMin).
// Everything before the first unsafe feature f is committed in order; f is recomputed by the whole block against
// the current `taken` set and committed; the next round starts at f+1.  Rounds = ceil(n1/256) + #conflicts.
// owner[] entries left behind by features that are re-examined later stay valid: a feature only changes its
// proposal when it is the first unsafe one, and then its old partner is already taken.
template <bool kLds>
__global__ __launch_bounds__(256) void tri_resolve_kernel(TriArgs A, int* __restrict__ prop, int* __restrict__ owner_g,
                                                          int* __restrict__ pairs, int* __restrict__ n_out) {
  extern __shared__ __align__(16) unsigned char tri_smem[];
  __shared__ int s_i1, s_n, s_first[4], s_cnt[4];
  __shared__ unsigned long long red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* owner_l = (int*)tri_smem;
  uint8_t* taken_l = tri_smem + sizeof(int) * (size_t)A.n2;
  if (kLds) {
    for (int i = tid; i < A.n2; i += 256) { owner_l[i] = INT_MAX; taken_l[i] = A.taken[i]; }
  } else {
    for (int i = tid; i < A.n2; i += 256) owner_g[i] = INT_MAX;
  }
  if (tid == 0) { s_i1 = 0; s_n = 0; }
  __syncthreads();
  for (;;) {
    const int base = s_i1, nbase = s_n;
    if (base >= A.n1) break;
    const int i = base + tid;
    const int p = i < A.n1 ? prop[i] : -1;
    const bool want = p >= 0;
    bool fre = false;
    if (want) {
      fre = kLds ? taken_l[p] == 0 : A.taken[p] == 0;
      if (fre) { if (kLds) atomicMin(&owner_l[p], i); else atomicMin(&owner_g[p], i); }
    }
    __syncthreads();
    const bool unsafe = want && (!fre || (kLds ? owner_l[p] : owner_g[p]) != i);
    const unsigned long long ub = __ballot(unsafe), wb = __ballot(want);
    if (lane == 0) s_first[wave] = ub ? wave * 64 + __ffsll((long long)ub) - 1 : 256;
    __syncthreads();
    const int f = min(min(s_first[0], s_first[1]), min(s_first[2], s_first[3]));
    // ordered commit of the safe prefix [0, f)
    const unsigned long long pre = wave * 64 >= f ? 0ull : (f - wave * 64 >= 64 ? wb : wb & ((1ull << (f - wave * 64)) - 1ull));
    if (lane == 0) s_cnt[wave] = __popcll(pre);
    __syncthreads();
    int off = nbase;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    if (want && tid < f) {
      const int o = off + __popcll(pre & ((1ull << lane) - 1ull));
      pairs[2 * o] = i; pairs[2 * o + 1] = p;                              // :522-525
      if (kLds) taken_l[p] = 1; else A.taken[p] = 1;
    }
    const int ncommit = nbase + s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    __syncthreads();
    if (f >= 256) {
      if (tid == 0) { s_i1 = base + 256; s_n = ncommit; }
      __syncthreads();
      continue;
    }
    const int i1 = base + f;                       // its partner went to an earlier feature: recompute
    unsigned long long key = kLds ? tri_scan(A, (const uint8_t*)taken_l, i1, tid, 256) : tri_scan(A, (const uint8_t*)A.taken, i1, tid, 256);
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) { const unsigned long long o = __shfl_xor(key, o2); key = o < key ? o : key; }
    if (lane == 0) red[wave] = key;
    __syncthreads();
    if (tid == 0) {
      unsigned long long k = red[0];
      for (int w = 1; w < 4; ++w) k = red[w] < k ? red[w] : k;
      int n = ncommit;
      if (k != ~0ull) {
        const int q = (int)(unsigned)(k & 0xffffffffull);
        pairs[2 * n] = i1; pairs[2 * n + 1] = q; ++n;
        if (kLds) taken_l[q] = 1; else A.taken[q] = 1;
      }
      s_i1 = i1 + 1; s_n = n;
    }
    __syncthreads();
  }
  if (tid == 0) *n_out = s_n;
}

// ---- fuse search (src/local_mapping/search_in_neighbors.rs:273-343, src/atlas/map/keyframe.rs:408-443) ---------
// Every (map point, target keyframe) pair: project, clamp the depth-scaled radius to [10, 50] px, scan the
// keyframe's keypoints inside the circle, keep the smallest Hamming distance below the threshold (lowest index on
// ties).  Block = 256 map points x one keyframe; the keyframe's keypoint coordinates stream through LDS and every
// thread tests all of them (the reference's linear scan, O(P*T*N), in f64 so that the <= r^2 gate is exact).
#define FUSE_CHUNK 2048
#define FUSE_THREADS 128
__global__ __launch_bounds__(FUSE_THREADS) void fuse_search_kernel(orbx_camera cam, const double* __restrict__ positions,
                                                          const uint8_t* __restrict__ mp_desc, int P,
                                                          const double* __restrict__ kf_pose_cw, const int* __restrict__ kf_off,
                                                          const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ descs,
                                                          int T, double radius_scale, unsigned thr, int* __restrict__ out_idx,
                                                          uint32_t* __restrict__ out_dist) {
  __shared__ double2 xy[FUSE_CHUNK];   // widened once per block: the circle test is f64 (keyframe.rs:435-437)
  const int t = blockIdx.y, p = blockIdx.x * FUSE_THREADS + threadIdx.x;
  const int f0 = kf_off[t], n = kf_off[t + 1] - f0;
  bool valid = p < P;
  double u = 0.0, v = 0.0, r2 = -1.0;
  Desc256 dq{};
  if (valid) {
    const double* q = kf_pose_cw + 7 * (size_t)t;                          // inverse pose from the host (se3.rs:56-63)
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double px = positions[3 * (size_t)p], py = positions[3 * (size_t)p + 1], pz = positions[3 * (size_t)p + 2];
    const double t0 = 2.0 * (qy * pz - qz * py), t1 = 2.0 * (qz * px - qx * pz), t2 = 2.0 * (qx * py - qy * px);
    const double c0 = qy * t2 - qz * t1, c1 = qz * t0 - qx * t2, c2 = qx * t1 - qy * t0;
    const double x = (t0 * qw + c0 + px) + q[4], y = (t1 * qw + c1 + py) + q[5], z = (t2 * qw + c2 + pz) + q[6];
    valid = !(z <= 0.0);                                                   // :286
    if (valid) {
      u = cam.fx * x / z + cam.cx;                                         // :291-292
      v = cam.fy * y / z + cam.cy;
      const double width = cam.cx * 2.0, height = cam.cy * 2.0;
      valid = !(u < 0.0 || u >= width || v < 0.0 || v >= height);          // :297
      const double radius = radius_scale * z / cam.fx;                     // :303
      const double sr = fmax(fmin(radius, 50.0), 10.0);                    // :304
      r2 = sr * sr;
      dq = load_desc(mp_desc + 32 * (size_t)p);
    }
  }
  // Hits are rare (about one per query) but each costs a dependent 32-byte global load; they are parked in a
  // 4-entry list (LDS, one column per thread) and scored together so that a wave stalls once per flush instead of
  // once per hit.  The scan itself is unrolled by 8 so that the LDS reads and the f64 chains of 8 tests overlap.
  __shared__ unsigned short hits[4][FUSE_THREADS];
  unsigned long long best = ~0ull;
  int nh = 0, cbase = 0;
  auto flush = [&]() {
    Desc256 dd[4];
    int id[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < nh) { id[k] = cbase + hits[k][threadIdx.x]; dd[k] = load_desc(descs + 32 * (size_t)(f0 + id[k])); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < nh) {
        const unsigned d = hamming(dq, dd[k]);
        if (d < thr) {                                                     // :336
          const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)id[k];
          best = key < best ? key : best;
        }
      }
    }
    nh = 0;
  };
  for (int c = 0; c < n; c += FUSE_CHUNK) {
    const int m = min(FUSE_CHUNK, n - c);
    if (nh) flush();                                                       // list entries are relative to cbase
    cbase = c;
    __syncthreads();
    for (int i = threadIdx.x; i < FUSE_CHUNK; i += FUSE_THREADS)
      xy[i] = i < m ? make_double2((double)kps[f0 + c + i].x, (double)kps[f0 + c + i].y) : make_double2(1e300, 1e300);
    __syncthreads();
    if (valid) {
      for (int i = 0; i < m; i += 8) {                                     // padding entries never pass the gate
        unsigned mask = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double2 q = xy[i + k];
          const double du = q.x - u, dv = q.y - v;
          mask |= (du * du + dv * dv <= r2 ? 1u : 0u) << k;
        }
        while (mask) {
          const int k = __ffs(mask) - 1;
          mask &= mask - 1;
          hits[nh][threadIdx.x] = (unsigned short)(i + k);
          if (++nh == 4) flush();
        }
      }
    }
  }
  if (nh) flush();
  if (p < P) {
    out_idx[(size_t)p * T + t] = best == ~0ull ? -1 : (int)(unsigned)(best & 0xffffffffull);
    out_dist[(size_t)p * T + t] = best == ~0ull ? 0u : (unsigned)(best >> 32);
  }
}

__global__ __launch_bounds__(256) void hamming_batch_kernel(const uint8_t* __restrict__ a,
                                                            const uint8_t* __restrict__ b, int n,
                                                            uint32_t* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = hamming(load_desc(a + (size_t)i * 32), load_desc(b + (size_t)i * 32));
}

}  // namespace

// `batch_total` pairs size the workspace; this call matches the `batch` pairs that start at pair `pair0` of the caller's arrays on stream
// `st` (nullptr = the handle's).  Pairs are independent and every array — the workspace's too — is indexed by pair, so a batch processed
// as several ranges (orbx_process_stereo_batch_device's two-stream form) gives the bits of one call over the whole batch.
int launch_stereo_match_range(orbx_handle* h, hipStream_t st, int batch_total, int pair0, int batch, const orbx_keypoint* d_kp, const uint8_t* d_desc,
                              const int* d_nkp, int cap_kp, orbx_dmatch* d_matches, int* d_nmatches,
                              double* d_points, uint8_t* d_has_point) {
  if (batch_total <= 0) return ORBX_OK;
  if (!st) st = h->stream;
  // workspace: tmp int2[batch*cap] | bstart int[batch*(SB_ROWS+1)] | sidx int[batch*cap] | sxy float2[batch*cap]
  const size_t n_tmp = (size_t)batch_total * cap_kp;
  const size_t bytes = sizeof(int2) * n_tmp + sizeof(int) * (size_t)batch_total * (SB_ROWS + 1) + sizeof(int) * n_tmp +
                       sizeof(float2) * n_tmp + 64;
  if (int rc = orbx_reserve(h, h->ws_match, bytes)) return rc;
  if (batch <= 0) return ORBX_OK;                      // (a call that only sizes the workspace)
  const size_t o = (size_t)pair0 * cap_kp;
  int2* tmp = (int2*)h->ws_match.p;
  float2* sxy = (float2*)(tmp + n_tmp);
  int* sidx = (int*)(sxy + n_tmp);
  int* bstart = sidx + n_tmp + (size_t)pair0 * (SB_ROWS + 1);
  tmp += o; sxy += o; sidx += o;
  d_kp += 2 * o; d_desc += 64 * o; d_nkp += 2 * (size_t)pair0;
  d_matches += o; d_nmatches += pair0; d_points += 3 * o; d_has_point += o;
  // stereo.rs:84-90: f64 product/quotient, then `as f32`
  const float max_disp = (float)(h->cam.fx * h->cam.baseline / 0.1);
  const float min_disp = (float)(h->cam.fx * h->cam.baseline / 40.0);
  // Large batches: the pair's right image in LDS, one workgroup per pair (stereo_match_lds_kernel); ORBX_SM_LDS=0 / 1 forces the choice (A/B runs, tests),
  // ORBX_SM_LDS=2 the three-launch form with the LDS matcher in the middle where the fused one would fit
  const size_t lds_need = (size_t)cap_kp * 44 + (size_t)(SB_ROWS + 1) * 4;
  const size_t lds_fused = (((size_t)cap_kp * 44 + (size_t)(2 * SB_ROWS + 1) * 4 + 7) & ~(size_t)7) + (size_t)cap_kp * 8;
  static const int sm_env = [] { const char* e = getenv("ORBX_SM_LDS"); return e ? atoi(e) : -1; }();
  static const bool sm_attr = hipFuncSetAttribute((const void*)stereo_match_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                              hipFuncSetAttribute((const void*)stereo_match_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
  const bool use_lds = sm_attr && lds_need <= 160 * 1024 && (sm_env >= 1 || (sm_env < 0 && batch >= h->n_cu));
  if (use_lds && lds_fused <= 160 * 1024 && sm_env != 2) {
    ProfScope ps(h, "stereo_match_kernel", st, true);
    hipLaunchKernelGGL(stereo_match_lds_kernel<true>, dim3(batch), dim3(SML_THREADS), lds_fused, st, d_kp, d_desc, d_nkp,
                       cap_kp, max_disp, min_disp, bstart, sidx, sxy, tmp, h->cam, d_matches, d_nmatches, d_points, d_has_point);
    ORBX_HIP(h, hipGetLastError());
    return ORBX_OK;
  }
  {
    ProfScope ps(h, "stereo_bucket_kernel", st);
    hipLaunchKernelGGL(stereo_bucket_kernel, dim3(batch), dim3(SB_THREADS), 0, st, d_kp, d_nkp, cap_kp, bstart, sidx, sxy);
  }
  if (use_lds) {
    ProfScope ps(h, "stereo_match_kernel", st, true);
    hipLaunchKernelGGL(stereo_match_lds_kernel<false>, dim3(batch), dim3(SML_THREADS), lds_need, st, d_kp, d_desc, d_nkp,
                       cap_kp, max_disp, min_disp, bstart, sidx, sxy, tmp, h->cam, d_matches, d_nmatches, d_points, d_has_point);
  } else {
    ProfScope ps(h, "stereo_match_kernel", st, true);
    dim3 grid((cap_kp + SM_LEFT_PER_BLOCK - 1) / SM_LEFT_PER_BLOCK, batch);
    hipLaunchKernelGGL(stereo_match_kernel, grid, dim3(SM_THREADS), 0, st, d_kp, d_desc, d_nkp,
                       cap_kp, max_disp, min_disp, bstart, sidx, sxy, tmp);
  }
  {
    ProfScope ps(h, "stereo_compact_kernel", st, true);
    hipLaunchKernelGGL(stereo_compact_kernel, dim3(batch), dim3(256), 0, st, d_kp, d_nkp, cap_kp,
                       (const int2*)tmp, h->cam, d_matches, d_nmatches, d_points, d_has_point);
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_stereo_match(orbx_handle* h, int batch, const orbx_keypoint* d_kp, const uint8_t* d_desc,
                        const int* d_nkp, int cap_kp, orbx_dmatch* d_matches, int* d_nmatches,
                        double* d_points, uint8_t* d_has_point) {
  return launch_stereo_match_range(h, nullptr, batch, 0, batch, d_kp, d_desc, d_nkp, cap_kp, d_matches, d_nmatches, d_points, d_has_point);
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

int launch_guided_match(orbx_handle* h, const orbx_keypoint* d_kp, const uint8_t* d_desc, int n, double img_w, double img_h,
                        const double* d_q_uv, const uint8_t* d_q_desc, int nq, double radius, int mode, int* d_out_idx,
                        uint32_t* d_out_dist) {
  if (nq <= 0) return ORBX_OK;
  // workspace: cell_start int[GG_CELLS+1] | sorted_idx int[n] | cell_of u16[n]
  const size_t bytes = sizeof(int) * (GG_CELLS + 1 + (size_t)std::max(n, 1)) + sizeof(unsigned short) * (size_t)std::max(n, 1) + 64;
  if (int rc = orbx_reserve(h, h->ws_match, bytes)) return rc;
  int* cell_start = (int*)h->ws_match.p;
  int* sorted_idx = cell_start + GG_CELLS + 1;
  unsigned short* cell_of = (unsigned short*)(sorted_idx + std::max(n, 1));
  const double winv = (double)GG_COLS / (img_w - 0.0), hinv = (double)GG_ROWS / (img_h - 0.0);   // tracking_frame.rs:58-59
  {
    ProfScope ps(h, "grid_build_kernel");
    hipLaunchKernelGGL(grid_build_kernel, dim3(1), dim3(1024), 0, h->stream, d_kp, n, winv, hinv, cell_start, sorted_idx, cell_of);
  }
  {
    ProfScope ps(h, "guided_match_kernel");
    hipLaunchKernelGGL(guided_match_kernel, dim3((nq + 3) / 4), dim3(256), 0, h->stream, d_desc, cell_start, sorted_idx, cell_of,
                       d_q_uv, d_q_desc, nq, radius, winv, hinv, mode, d_out_idx, d_out_dist);
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_search_for_triangulation(orbx_handle* h, const orbx_camera* cam, const double* F9, const double* epipole,
                                    const orbx_keypoint* d_kp1, const uint8_t* d_desc1, const uint8_t* d_mp1,
                                    const uint8_t* d_stereo1, int n1, const orbx_keypoint* d_kp2, const uint8_t* d_desc2,
                                    const uint8_t* d_mp2, int n2, unsigned max_dist, int* d_pairs, int* d_n_out) {
  if (n1 <= 0 || n2 <= 0) {
    ORBX_HIP(h, hipMemsetAsync(d_n_out, 0, sizeof(int), h->stream));
    return ORBX_OK;
  }
  auto f64_as_u32 = [](double v) -> unsigned { return v > 0 ? (v >= 4294967295.0 ? 4294967295u : (unsigned)v) : 0u; };
  const unsigned iw = f64_as_u32(cam->cx * 2.0), ih = f64_as_u32(cam->cy * 2.0);               // triangulation.rs:434-435
  const float fc = std::ceil((float)iw / 32.0f), fr = std::ceil((float)ih / 32.0f);
  const int cols = (int)std::min(64.0f, std::max(fc, 0.0f)), rows = (int)std::min(64.0f, std::max(fr, 0.0f));   // :437-438
  if (cols < 1 || rows < 1) {
    ORBX_HIP(h, hipMemsetAsync(d_n_out, 0, sizeof(int), h->stream));
    return ORBX_OK;
  }
  // workspace: cell_start int[4100] | sorted_idx int[n2] | prop int[n1] | owner int[n2] | cell_of u16[n2] | taken u8[n2]
  const size_t bytes = sizeof(int) * (4100 + 2 * (size_t)n2 + (size_t)n1) + 2 * (size_t)n2 + (size_t)n2 + 64;
  if (int rc = orbx_reserve(h, h->ws_match, bytes)) return rc;
  int* cell_start = (int*)h->ws_match.p;
  int* sorted_idx = cell_start + 4100;
  int* prop = sorted_idx + n2;
  int* owner = prop + n1;
  unsigned short* cell_of = (unsigned short*)(owner + n2);
  uint8_t* taken = (uint8_t*)(cell_of + n2);
  TriArgs A{};
  for (int i = 0; i < 9; ++i) A.F[i] = F9[i];
  A.epx = epipole[0]; A.epy = epipole[1];
  A.cols = cols; A.rows = rows; A.max_dist = max_dist; A.n1 = n1; A.n2 = n2;
  A.kp1 = d_kp1; A.desc1 = d_desc1; A.mp1 = d_mp1; A.stereo1 = d_stereo1; A.kp2 = d_kp2; A.desc2 = d_desc2;
  A.cell_start = cell_start; A.sorted_idx = sorted_idx; A.cell_of = cell_of; A.taken = taken;
  {
    ProfScope ps(h, "tri_grid_build_kernel");
    hipLaunchKernelGGL(tri_grid_build_kernel, dim3(1), dim3(1024), 0, h->stream, d_kp2, n2, cols, rows, d_mp2, cell_start, sorted_idx, cell_of, taken);
  }
  {
    ProfScope ps(h, "tri_propose_kernel");
    hipLaunchKernelGGL(tri_propose_kernel, dim3((n1 + 3) / 4), dim3(256), 0, h->stream, A, prop);
  }
  {
    ProfScope ps(h, "tri_resolve_kernel");
    const size_t lds = 5 * (size_t)n2 + 16;
    if (lds <= 150 * 1024) {
      if (lds > 64 * 1024)   // more than 64 KB of dynamic LDS needs the opt-in
        ORBX_HIP(h, hipFuncSetAttribute((const void*)tri_resolve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      hipLaunchKernelGGL(tri_resolve_kernel<true>, dim3(1), dim3(256), lds, h->stream, A, prop, owner, d_pairs, d_n_out);
    } else {
      hipLaunchKernelGGL(tri_resolve_kernel<false>, dim3(1), dim3(256), 0, h->stream, A, prop, owner, d_pairs, d_n_out);
    }
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_fuse_search(orbx_handle* h, const orbx_camera* cam, const double* d_positions, const uint8_t* d_mp_desc, int P,
                       const double* d_kf_pose_cw, const int* d_kf_off, const orbx_keypoint* d_kps, const uint8_t* d_descs,
                       int T, double radius_scale, unsigned desc_threshold, int* d_out_idx, uint32_t* d_out_dist) {
  if (P <= 0 || T <= 0) return ORBX_OK;
  if (T > 65535) return orbx_fail(h, ORBX_ERR_INVALID, "fuse search: more than 65535 target keyframes");
  ProfScope ps(h, "fuse_search_kernel");
  hipLaunchKernelGGL(fuse_search_kernel, dim3((P + FUSE_THREADS - 1) / FUSE_THREADS, T), dim3(FUSE_THREADS), 0, h->stream, *cam, d_positions, d_mp_desc, P,
                     d_kf_pose_cw, d_kf_off, d_kps, d_descs, T, radius_scale, desc_threshold, d_out_idx, d_out_dist);
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_search_for_triangulation_bow(orbx_handle* h, const double* F9, const double* epipole, const orbx_keypoint* d_kp1,
                                        const uint8_t* d_desc1, const uint8_t* d_mp1, const uint8_t* d_stereo1, int n1,
                                        const orbx_keypoint* d_kp2, const uint8_t* d_desc2, const uint8_t* d_mp2, int n2,
                                        const int* d_sorted_idx, const int* d_rng_lo, const int* d_rng_hi, unsigned max_dist,
                                        int* d_pairs, int* d_n_out) {
  if (n1 <= 0 || n2 <= 0) {
    ORBX_HIP(h, hipMemsetAsync(d_n_out, 0, sizeof(int), h->stream));
    return ORBX_OK;
  }
  // workspace: prop int[n1] | owner int[n2] | taken u8[n2]
  const size_t bytes = sizeof(int) * ((size_t)n1 + (size_t)n2) + (size_t)n2 + 64;
  if (int rc = orbx_reserve(h, h->ws_match, bytes)) return rc;
  int* prop = (int*)h->ws_match.p;
  int* owner = prop + n1;
  uint8_t* taken = (uint8_t*)(owner + n2);
  ORBX_HIP(h, hipMemcpyAsync(taken, d_mp2, (size_t)n2, hipMemcpyDeviceToDevice, h->stream));   // :606-608
  TriArgs A{};
  for (int i = 0; i < 9; ++i) A.F[i] = F9[i];
  A.epx = epipole[0]; A.epy = epipole[1];
  A.cols = 0; A.rows = 0; A.max_dist = max_dist; A.n1 = n1; A.n2 = n2;
  A.kp1 = d_kp1; A.desc1 = d_desc1; A.mp1 = d_mp1; A.stereo1 = d_stereo1; A.kp2 = d_kp2; A.desc2 = d_desc2;
  A.cell_start = nullptr; A.sorted_idx = d_sorted_idx; A.cell_of = nullptr; A.taken = taken;
  A.rng_lo = d_rng_lo; A.rng_hi = d_rng_hi;
  {
    ProfScope ps(h, "tri_propose_kernel");
    hipLaunchKernelGGL(tri_propose_kernel, dim3((n1 + 3) / 4), dim3(256), 0, h->stream, A, prop);
  }
  {
    ProfScope ps(h, "tri_resolve_kernel");
    const size_t lds = 5 * (size_t)n2 + 16;
    if (lds <= 150 * 1024) {
      if (lds > 64 * 1024)
        ORBX_HIP(h, hipFuncSetAttribute((const void*)tri_resolve_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      hipLaunchKernelGGL(tri_resolve_kernel<true>, dim3(1), dim3(256), lds, h->stream, A, prop, owner, d_pairs, d_n_out);
    } else {
      hipLaunchKernelGGL(tri_resolve_kernel<false>, dim3(1), dim3(256), 0, h->stream, A, prop, owner, d_pairs, d_n_out);
    }
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}
