// orb_kernels.hip — batched ORB extractor for gfx950: cv::ORB::detectAndCompute as configured by
// the reference (src/tracking/frame/stereo.rs:38-48, :68-78), restated per SURVEY.md Appendix A.
//
// One call processes n_images images (left and right of every stereo pair of a batch); every kernel
// covers all images, and all pyramid levels where there is no level-to-level dependency:
//
//   copy_l0_kernel        (only if the caller's rows are not 4-byte aligned)
//   resize_kernel  x7     level l from level l-1, INTER_LINEAR_EXACT fixed point: 4 px x 6 rows per thread from an LDS-staged source tile,
//                         8-byte source windows, v_dot4 taps                                  (A.4)
//   blur_kernel           7x7 sigma-2 fixed-point Gaussian of every level: register window walking down
//                         column strips (no LDS), DPP neighbours, v_dot4 / v_dot2 taps       (A.8)
//   fast_kernel           FAST-9/16 score + 3x3 NMS + border filter on 62x62 LDS tiles, 3 tiles per block with the next
//                         tile's pixels prefetched: byte-parallel compass pre-test (v_bitop3_b32), compacted survivors,
//                         arc score on f16 denormals (v_pk_minimum3/maximum3_f16), one append per chain; per-level
//                         candidate lists + score histograms                                  (A.5)
//   harris_select_kernel  retainBest(2 n_l) by FAST score via the histogram, Harris response
//                         of the survivors                                                    (A.6)
//   rank_select_kernel    canonical order (response desc, y, x) by bitonic sort in LDS, retainBest(n_l)
//                         incl. ties with the n-th; the kept keypoints again in spatial order  (A.6)
//   describe_kernel       16 lanes per keypoint, 4 keypoints per wave, walking the spatial list:
//                         intensity-centroid angle from the level, 37x37 blurred patch in LDS ->
//                         256 steered BRIEF tests, one __ballot per 16 bits of 4 descriptors (A.7, A.8)
//
// Everything is integer/byte work except the Harris response, the angle and the pattern rotation,
// which are f32/f64 written one IEEE operation at a time (no contraction) so that the results are
// bit-identical to the CPU specification.  Byte/integer stencil and scan work: no MFMA here by design; what
// binds each kernel (VALU issue for FAST and blur) is in DESIGN.md §4 and §8.
#include <cfloat>
#include <cmath>

#include "orbx_internal.hpp"

namespace {

#include "orb_pattern_31.inc"

constexpr int EDGE = 31;             // edgeThreshold, stereo.rs:42

__constant__ __attribute__((aligned(16))) signed char c_pattern[256 * 4];

// XCD-aware block -> (image, block-within-image) map.  Workgroups are dealt round-robin over the 8
// XCDs by linear id, so with a 1-D grid of 8*ceil(n_img/8)*per_img blocks, id = (grp*per_img + b)*8 + x
// puts every block of image grp*8+x on XCD x (whichever physical XCD that is): an image's pyramid
// (1.2 MB) then stays in ONE 4 MB L2 while its tiles / keypoints are processed, instead of being
// fetched by up to 8 L2s.  (A 2-D grid with the image group in y is NOT dealt this way: measured, it
// loses the locality.)  The division by per_img uses the host's floor(2^32/per_img) and one fix-up
// step — these kernels launch 100k+ short blocks and a generic division is ~40 scalar instructions
// of prologue.  Speed only; any placement gives the same results.
struct XcdMap { unsigned per_img, magic; };
__device__ __forceinline__ bool xcd_decode_at(XcdMap m, int n_img, unsigned L, int& img, int& b) {
  const unsigned x = L & 7u, slot = L >> 3;
  unsigned grp = __umulhi(slot, m.magic);
  unsigned r = slot - grp * m.per_img;
  if (r >= m.per_img) { ++grp; r -= m.per_img; }
  b = (int)r;
  img = (int)(grp * 8u + x);
  return img < n_img;
}
__device__ __forceinline__ bool xcd_decode(XcdMap m, int n_img, int& img, int& b) { return xcd_decode_at(m, n_img, blockIdx.x, img, b); }
static inline XcdMap xcd_map(int per_img) {
  return XcdMap{(unsigned)per_img, per_img > 1 ? (unsigned)(0x100000000ull / (unsigned)per_img) : 0xffffffffu};
}
static inline dim3 xcd_grid(int per_img, int n_img) { return dim3(8u * (unsigned)((n_img + 7) / 8) * (unsigned)per_img); }

__device__ __forceinline__ const uint8_t* level_ptr(const OrbSrc& s, const OrbGeom& g, int img, int l,
                                                    int& pitch) {
  if (l == 0) { pitch = s.l0_pitch; return s.l0 + (size_t)img * s.l0_img_stride; }
  pitch = g.lv[l].pitch;
  return s.pyr + (size_t)img * g.slot_bytes + g.lv[l].off;
}

// ---- level 0 copy (unaligned caller images only) -------------------------------------------------------
__global__ __launch_bounds__(256) void copy_l0_kernel(const uint8_t* __restrict__ src, size_t img_stride,
                                                      size_t row_stride, int w, int h, uint8_t* __restrict__ pyr,
                                                      unsigned slot, int pitch) {
  const int img = blockIdx.z;
  const int y = blockIdx.y;
  const uint8_t* s = src + (size_t)img * img_stride + (size_t)y * row_stride;
  uint8_t* d = pyr + (size_t)img * slot + (size_t)y * pitch;
  for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) d[x] = s[x];
}

// ---- A.4 resize: 4 destination pixels per thread, one aligned u32 store ----------------------------------
// The four source offsets of a thread span at most 6 bytes (scale 1.2), so each source row is read as
// one unaligned 8-byte window; a tap pair is (window >> 8*o) and the 8.8 horizontal sum is one
// v_dot4_u32_u8:  cx0*p0 + cx1*p1 = (cx0-1)*p0 + cx1*p1 + p0  (cx0 = 256-cx1 can be 256, cx0-1 fits a byte).
__device__ __forceinline__ unsigned ld_u32(const uint8_t* p);
#ifndef ORBX_RESIZE_ROWS
#define ORBX_RESIZE_ROWS 6
#endif
// output rows per thread (y, y+16, ...).  Direct-load form: 2 / 3 / 4 rows 0.362 / 0.334 / 0.336 ms per 256 pairs; LDS-staged form
// (below): 2 / 3 / 4 / 6 / 8 rows 0.343 / 0.300 / 0.291 / 0.260 / ~0.27
constexpr int RESIZE_ROWS = ORBX_RESIZE_ROWS;
// byte 2 of four dwords as one dword: three v_perm_b32 / or instead of four shifts and three shift-ors
__device__ __forceinline__ unsigned pack_byte2(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_amdgcn_perm(b, a, 0x0c0c0602u) | __builtin_amdgcn_perm(d, c, 0x06020c0cu);
}

__device__ __forceinline__ unsigned resize_h(unsigned long long win, int o, unsigned coef) {
  const unsigned pr = (unsigned)(win >> (8 * o));
  return __builtin_amdgcn_udot4(pr, coef, pr & 0xffu, false);
}

// What bound the first form of this kernel (every thread loading its own unaligned 8-byte windows from global memory, 6 per thread)
// was the texture addresser: TA busy 73 % at 2.5 TB/s of traffic, VALU half idle (profiles/r01l_pmc_sq_b256.txt).  Now the block
// stages its source rectangle — at most 73 rows x 128 bytes for scale factors up to 1.5 — with aligned 16-byte loads (a quarter of
// the lane requests) and the windows come from LDS as three aligned dwords + two v_alignbyte_b32 (unaligned LDS reads are slow).
constexpr int RZ_LP = 128;                                               // LDS row pitch: 8 x 16 bytes (60 * 1.5 + 8 + 15 = 113)
constexpr int rz_src_rows(int rows) { return (16 * rows - 1) * 3 / 2 + 3; }   // source rows under 16*rows output rows at scale 1.5
constexpr int RESIZE_ROWS_SMALL = 2;     // small batches (a pair): more, shorter blocks — latency, not throughput
#ifndef ORBX_RESIZE_CHAIN
#define ORBX_RESIZE_CHAIN 1
#endif
// consecutive tiles per block with the next one's source loads in flight under the current one's arithmetic: 1 / 2 / 3 tiles
// 0.260 / 0.304 / 0.334 ms per 256 pairs — the upper levels have too few tiles to give any away (level 7: 8 per image)
constexpr int RESIZE_CHAIN = ORBX_RESIZE_CHAIN;
template <int RESIZE_ROWS>
__global__ __launch_bounds__(256) void resize_kernel(OrbSrc s, OrbGeom g, int l, int n_img, XcdMap xm, int tiles_x, int n_tiles,
                                                     const unsigned* __restrict__ xtab,
                                                     const unsigned* __restrict__ ytab) {
  constexpr int RZ_SRC_ROWS = rz_src_rows(RESIZE_ROWS);
  __shared__ __attribute__((aligned(16))) uint8_t st[RZ_SRC_ROWS * RZ_LP + 16];
  int img, chain;
  if (!xcd_decode(xm, n_img, img, chain)) return;
  int sp;
  const uint8_t* src = level_ptr(s, g, img, l - 1, sp);
  const int sh = g.lv[l - 1].h;
  const int w = g.lv[l].w, h = g.lv[l].h, dp = g.lv[l].pitch;
  uint8_t* dst = s.pyr + (size_t)img * g.slot_bytes + g.lv[l].off;
  const int tid = threadIdx.x;
  constexpr int NP = (RZ_SRC_ROWS + 31) / 32;          // staging passes: every pass's load is issued before the first store waits for one
  struct Tile { int X0, Y0, ax, ncol, sy_lo, nrow, off; };
  uint4 v[NP];
  // (the rectangle's bounds from the scale in f32 with margins instead of the four scalar table loads: 0.264 -> 0.274 ms)
  // a tile's source rectangle (block-uniform): columns from the first pixel's left tap (16-byte aligned, inside the row) to the
  // last active thread's window end, rows from the first output row's upper tap to the last one's lower tap; and its loads
  auto fetch = [&](int tb, Tile& t) {
    const int tby = tb / tiles_x, tbx = tb - tby * tiles_x;
    t.X0 = tbx * 64; t.Y0 = tby * (16 * RESIZE_ROWS);
    const int sx_lo = (int)(xtab[t.X0] >> 16);
    const int xb_last = min((int)(xtab[min(t.X0 + 60, ((w + 3) & ~3) - 4)] >> 16), sp - 8);
    t.ax = min(sx_lo & ~15, sp - 16);
    t.ncol = (xb_last + 8 - t.ax + 15) >> 4;                          // 16-byte columns, at most 8
    t.sy_lo = (int)(ytab[t.Y0] >> 16);
    const int sy_hi = min((int)(ytab[min(t.Y0 + 16 * RESIZE_ROWS - 1, h - 1)] >> 16) + 1, sh - 1);
    t.nrow = sy_hi - t.sy_lo + 1;
    const int c = tid & 7;
    // (a column that would run past the row end is pulled back inside it: it then repeats bytes of its neighbour)
    const int cs = min(t.ax + 16 * c, sp - 16);
    t.off = cs - t.ax;
    const uint8_t* gp = src + (unsigned)(__umul24((unsigned)(t.sy_lo + (tid >> 3)), (unsigned)sp) + (unsigned)cs);
    const unsigned gstep = 32u * (unsigned)sp;
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (c < t.ncol && (tid >> 3) + 32 * k < t.nrow) __builtin_memcpy(&v[k], gp + (unsigned)k * gstep, 16);
  };
  const int tb0 = chain * RESIZE_CHAIN, tb_end = min(tb0 + RESIZE_CHAIN, n_tiles);
  Tile cur, nxt;
  fetch(tb0, cur);
  for (int tb = tb0; tb < tb_end; ++tb) {
    const int x0 = cur.X0 + (tid & 15) * 4;
    const int yb = cur.Y0 + (tid >> 4);
    const bool active = yb < h && x0 < w;
    uint4 xt4 = {0u, 0u, 0u, 0u};
    unsigned yt[RESIZE_ROWS];
    if (active) {
      xt4 = *reinterpret_cast<const uint4*>(xtab + x0);      // table padded to a multiple of 4 entries
#pragma unroll
      for (int r = 0; r < RESIZE_ROWS; ++r) yt[r] = ytab[min(yb + 16 * r, h - 1)];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int r = (tid >> 3) + 32 * k;
      if ((tid & 7) < cur.ncol && r < cur.nrow) {
        if ((cur.off & 15) == 0) *reinterpret_cast<uint4*>(&st[r * RZ_LP + cur.off]) = v[k];
        else {
          unsigned* q = reinterpret_cast<unsigned*>(&st[r * RZ_LP + cur.off]);
          q[0] = v[k].x; q[1] = v[k].y; q[2] = v[k].z; q[3] = v[k].w;
        }
      }
    }
    __syncthreads();
    if (tb + 1 < tb_end) fetch(tb + 1, nxt);
    if (active) {
      const unsigned xt[4] = {xt4.x, xt4.y, xt4.z, xt4.w};
      // window start, pulled back at the right edge so that the 8 bytes stay inside the row (offsets then reach 7; the
      // second tap of a pixel clamped at the edge has weight 0, so the zero shifted in for it is never used)
      const int xb = min((int)(xt[0] >> 16), sp - 8);
      const int col = xb - cur.ax, sh8 = col & 3;
      const uint8_t* lp = st + (col & ~3);
#pragma unroll
      for (int r = 0; r < RESIZE_ROWS; ++r) {
        const int y = yb + 16 * r;
        if (y >= h) break;
        const int y0 = (int)(yt[r] >> 16), y1 = min(y0 + 1, sh - 1);
        const unsigned* p0 = reinterpret_cast<const unsigned*>(lp + (y0 - cur.sy_lo) * RZ_LP);
        const unsigned* p1 = reinterpret_cast<const unsigned*>(lp + (y1 - cur.sy_lo) * RZ_LP);
        const unsigned a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
        const unsigned long long w0 = (unsigned long long)__builtin_amdgcn_alignbyte(a1, a0, (unsigned)sh8) |
                                      ((unsigned long long)__builtin_amdgcn_alignbyte(a2, a1, (unsigned)sh8) << 32);
        const unsigned long long w1 = (unsigned long long)__builtin_amdgcn_alignbyte(b1, b0, (unsigned)sh8) |
                                      ((unsigned long long)__builtin_amdgcn_alignbyte(b2, b1, (unsigned)sh8) << 32);
        const unsigned cy1 = yt[r] & 0xffffu, cy0 = 256u - cy1;
        unsigned vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                           // (pixels past the row end compute garbage into the row padding)
          const int o = (int)(xt[k] >> 16) - xb;               // 0..7
          const unsigned cx1 = xt[k] & 0xffffu;
          const unsigned coef = (255u - cx1) | (cx1 << 8);     // (cx0 - 1, cx1)
          const unsigned h0 = resize_h(w0, o, coef), h1 = resize_h(w1, o, coef);   // 8.8
          vv[k] = __umul24(cy0, h0) + __umul24(cy1, h1) + 32768u;                                        // 16.16, rounded
        }
        const unsigned packed = pack_byte2(vv[0], vv[1], vv[2], vv[3]);
        *reinterpret_cast<unsigned*>(dst + (unsigned)(__umul24((unsigned)y, (unsigned)dp) + (unsigned)x0)) = packed;
      }
    }
    __syncthreads();   // the tile is consumed before the next one overwrites it
    cur = nxt;
  }
}

// tile -> (level, tile x, tile y) from the table orb_prepare_geometry builds (one scalar load instead of a
// search over the levels and a division in every block's prologue): l | tx << 3 | ty << 17
__device__ __forceinline__ void decode_tile(const unsigned* __restrict__ tab, int tile, int& l, int& tx, int& ty) {
  const unsigned e = tab[tile];
  l = (int)(e & 7u); tx = (int)((e >> 3) & 0x3fffu); ty = (int)(e >> 17);
}


// ---- A.8 Gaussian blur 7x7 sigma 2, taps {18,34,48,56,48,34,18}/256, 8.8 then 16.16 ------------------
// No LDS: a wave owns a 256-px-wide column strip (lane = 4 consecutive pixels = one dword) and walks
// down BLUR_STRIP rows.  Per input row: one coalesced 256-B dword load per wave, neighbour dwords by
// DPP wave shifts (+2 edge lanes loading), horizontal taps with v_alignbyte + v_dot4_u32_u8 (8.8 sums),
// a ring of row PAIRS of those sums in registers, vertical taps by v_dot2_u32_u16 (16.16), one dword store.
// Lanes whose 10-px window crosses the image edge rebuild their three dwords with BORDER_REFLECT_101 from the
// same registers (v_perm_b32, no loads).
#ifndef ORBX_BLUR_PF
#define ORBX_BLUR_PF 2
#endif
constexpr int BLUR_PF = ORBX_BLUR_PF;   // input rows in flight per lane
#ifndef ORBX_BLUR_STRIP
#define ORBX_BLUR_STRIP 32
#endif
#if defined(ORBX_BLUR_NOARITH) && ORBX_BLUR_NOARITH == 2   // measurement build: 256-px strips on line boundaries, no halo lanes (results meaningless)
constexpr int BLUR_W = 256, BLUR_HALO = 0;
#else
constexpr int BLUR_W = 248, BLUR_HALO = 1;
#endif
constexpr int BLUR_STRIP = ORBX_BLUR_STRIP, BLUR_H = 4 * BLUR_STRIP;   // lanes 1..62 produce output, 0 and 63 are halo;
// rows per wave 16 / 32 / 48 / 64 / 96 / 128: 0.307 / 0.300 / 0.302 / 0.315 / 0.366 / 0.369 ms per 256 pairs

__device__ __forceinline__ void blur_hsum(unsigned d0, unsigned d1, unsigned d2, unsigned (&hs)[4]) {
  const unsigned G0 = 18u | (34u << 8) | (48u << 16) | (56u << 24);   // px x-3..x
  const unsigned G1 = 48u | (34u << 8) | (18u << 16);                 // px x+1..x+3
  hs[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 1), G1,
                                 __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 1), G0, 0u, false), false);
  hs[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 2), G1,
                                 __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 2), G0, 0u, false), false);
  hs[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d2, d1, 3), G1,
                                 __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d1, d0, 3), G0, 0u, false), false);
  hs[3] = __builtin_amdgcn_udot4(d2, G1, __builtin_amdgcn_udot4(d1, G0, 0u, false), false);
}

// BORDER_REFLECT_101 at the left / right image edge without touching memory: every byte an edge lane needs lies in
// the three dwords the wave already holds (its own and its two neighbours'), so the reflected window is three
// v_perm_b32 with per-lane selectors computed once (levels are at least 8 px wide, so a lane has at most one edge).
struct BlurEdge {
  unsigned s1, s2;      // selectors of d1' = perm(d1, d0, s1) and d2' = perm(hi ? d2 : d1, hi ? d1 : d0, s2)
  bool left, hi;        // left: d0' = (px4, px3, px2, px1) = perm(d2, d1, 0x01020304)
};
__device__ __forceinline__ BlurEdge blur_edge_setup(int x0, int w, bool active) {
  BlurEdge e;
  e.s1 = 0x07060504u; e.s2 = 0x07060504u; e.hi = true;
  e.left = active && x0 == 0;
  const int rem = w - x0;                       // pixels of the image at and right of x0
  if (active && rem <= 6) {                     // the window reaches x0+6
    unsigned s1 = 0, s2 = 0;
    if (rem >= 4) {                             // d1 whole; d2 byte j = px x0+4+j, reflected px = x0 + (2 rem - 6 - j)
      s1 = 0x07060504u;
#pragma unroll
      for (int j = 0; j < 4; ++j) s2 |= (unsigned)(j < rem - 4 ? 4 + j : 2 * rem - 6 - j) << (8 * j);   // over (d2, d1)
    } else {                                    // d1 byte j >= rem reflects to x0 + (2 rem - 2 - j); d2 to x0 + (2 rem - 6 - j)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1 |= (unsigned)(j < rem ? 4 + j : 4 + 2 * rem - 2 - j) << (8 * j);                             // over (d1, d0)
        const int o = 2 * rem - 6 - j;          // below -4: only feeds pixels past the edge, any byte will do
        s2 |= (unsigned)(o >= -4 ? 4 + o : 0) << (8 * j);
      }
      e.hi = false;
    }
    e.s1 = s1; e.s2 = s2;
  }
  return e;
}

// d1 = this lane's dword of the row, loaded two rows ahead by the caller so that the load latency overlaps the
// arithmetic of the rows in between; EDGE = some lane of this wave sits on the image edge (wave-uniform: two copies of the row
// loop — as a run-time flag the compiler turned the branch into 3 v_perm + 5 v_cndmask in every wave's every row, a sixth of it)
template <bool EDGE>
__device__ __forceinline__ void blur_row(unsigned d1, const BlurEdge& e, unsigned (&hs)[4]) {
  unsigned d0 = __builtin_amdgcn_update_dpp(0u, d1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  unsigned d2 = __builtin_amdgcn_update_dpp(0u, d1, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  if (EDGE) {
    const unsigned n0 = e.left ? __builtin_amdgcn_perm(d2, d1, 0x01020304u) : d0;
    const unsigned n2 = __builtin_amdgcn_perm(e.hi ? d2 : d1, e.hi ? d1 : d0, e.s2);
    d1 = __builtin_amdgcn_perm(d1, d0, e.s1);
    d0 = n0; d2 = n2;
  }
  blur_hsum(d0, d1, d2, hs);
}

// BORDER_REFLECT_101 for an index at most n-1 outside [0, n) (levels are at least 8 px, the kernel reaches 3 px out)
__device__ __forceinline__ int reflect101_once(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// One wave = M row groups of 64/M lanes: group k walks rows [ys + k*R, ys + (k+1)*R), R = BLUR_STRIP/M, of a strip (64/M - 2)*4 px
// wide (first and last lane of a group are halo lanes).  M = 1 is the full 248-px strip; M = 2 (120 px) and M = 4
// (56 px) take the narrow remainder at the right of a level in 22 resp. 14 row steps instead of 38, so a remainder of
// 8 px (752 = 3*248 + 8) no longer costs a whole strip.  For M > 1 the row index is per lane (VALU), for M = 1 scalar.
template <int M, bool EDGE>
__device__ __forceinline__ void blur_strip_e(const uint8_t* __restrict__ src, int pitch, int h, uint8_t* __restrict__ dst, int dpitch,
                                             const BlurEdge& be, int x0, bool active, int ys, int nrows, int nrows_l) {
  const int xl = max(0, min(x0, pitch - 4));
  // input rows q = 0 .. nrows+5 are image rows reflect(ys-3+q); p0/p1 hold this lane's dword of rows q and q+1
  const int last = (M > 1 ? max(nrows_l, 1) : nrows) + 5;
  // 32-bit byte offsets from the (wave-uniform) level base: scalar base + one VGPR offset per access instead of 64-bit
  // per-lane pointer arithmetic (v_mad_u64_u32 is a quarter-rate instruction: four of them per row were a fifth of the row)
  auto rowp = [&](int q) { return src + (unsigned)(__umul24((unsigned)reflect101_once(ys - 3 + min(q, last), h), (unsigned)pitch) + (unsigned)xl); };
  // this lane's dwords of the next BLUR_PF input rows are in flight.  2 / 3 / 4 / 6 rows ahead: 0.298 / 0.298 / 0.298 / 0.303 ms per
  // 256 pairs, and dropping the edge code from the waves without an edge lane (a sixth of their VALU work) changed nothing either:
  // at 1.5 GB of counter traffic in 0.298 ms = 5.0 TB/s the kernel sits on what HBM delivers for a half-read half-write stream
  unsigned pq[BLUR_PF];
#pragma unroll
  for (int k = 0; k < BLUR_PF; ++k) pq[k] = *reinterpret_cast<const unsigned*>(rowp(k));
  // Vertical taps on PAIRS of rows: the 8.8 horizontal sums fit 16 bits, so two consecutive rows of one pixel share a
  // register and v_dot2_u32_u16 applies two taps at once: out(y) = (w0,w1).(18,34) + (w2,w3).(48,56) + (w4,w5).(48,34)
  // + 18 w6 — three dot2 and one mad instead of three adds and four multiplies.  pr[q % 6] = rows (q, q+1).
  typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));
  unsigned pr[6][4], hprev[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const unsigned cur = pq[0];
#pragma unroll
    for (int k = 0; k + 1 < BLUR_PF; ++k) pq[k] = pq[k + 1];
    pq[BLUR_PF - 1] = *reinterpret_cast<const unsigned*>(rowp(i + BLUR_PF));
    unsigned hs[4];
#ifdef ORBX_BLUR_NOARITH   // measurement build only (scripts/build_variant.sh): the kernel's loads and stores without its arithmetic
    hs[0] = hs[1] = hs[2] = hs[3] = cur;
#else
    blur_row<EDGE>(cur, be, hs);
#endif
    if (i > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) pr[i - 1][k] = hprev[k] | (hs[k] << 16);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) hprev[k] = hs[k];
  }
  const blur_us2 T0 = {18, 34}, T1 = {48, 56}, T2 = {48, 34};
  for (int y0 = 0; y0 < nrows; y0 += 6) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int y = y0 + i;
      if (y < nrows) {   // wave-uniform
        // input row q = y+6 (image row ys+y+3) arrives; window rows are q = y .. y+6
        const unsigned cur = pq[0];
#pragma unroll
        for (int k = 0; k + 1 < BLUR_PF; ++k) pq[k] = pq[k + 1];
        pq[BLUR_PF - 1] = *reinterpret_cast<const unsigned*>(rowp(y + 6 + BLUR_PF));
        unsigned hs[4];
#ifdef ORBX_BLUR_NOARITH
        const unsigned packed = cur ^ hprev[0];
        hprev[0] = cur;
#else
        blur_row<EDGE>(cur, be, hs);
        unsigned vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          pr[(i + 5) % 6][k] = hprev[k] | (hs[k] << 16);      // rows (y+5, y+6)
          hprev[k] = hs[k];
          unsigned v = __umul24(18u, hs[k]) + 32768u;          // (hs < 2^16) rounding constant rides on the first multiply-add
          v = __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, pr[i % 6][k]), T0, v, false);
          v = __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, pr[(i + 2) % 6][k]), T1, v, false);
          vv[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, pr[(i + 4) % 6][k]), T2, v, false);
        }
        const unsigned packed = pack_byte2(vv[0], vv[1], vv[2], vv[3]);   // (v >> 16) of the four 16.16 sums
#endif
        if (active && (M == 1 || y < nrows_l)) *reinterpret_cast<unsigned*>(dst + (unsigned)(__umul24((unsigned)(ys + y), (unsigned)dpitch) + (unsigned)x0)) = packed;
      }
    }
  }
}

template <int M>
__device__ __forceinline__ void blur_strip(const uint8_t* __restrict__ src, int pitch, int w, int h, uint8_t* __restrict__ dst,
                                           int dpitch, int xbase, int ys_wave, int lane) {
  constexpr int GL = 64 / M, R = BLUR_STRIP / M;
  const int gl = lane & (GL - 1);
  int ys = ys_wave + (lane / GL) * R;
  const int nrows = min(R, h - ys_wave);          // wave-uniform: group 0 has the most rows
  int nrows_l = min(R, h - ys);                    // this lane's group (M > 1: can be <= 0 below the image)
  if (M > 1 && nrows_l <= 0) { ys = ys_wave; nrows_l = 0; }   // idle group: walk group 0's rows, store nothing
  const int x0 = xbase + (gl - BLUR_HALO) * 4;
  const bool active = gl >= BLUR_HALO && gl <= GL - 1 - BLUR_HALO && x0 < w && nrows_l > 0;
  const BlurEdge be = blur_edge_setup(x0, w, active);
  // some lane of this wave sits on the image edge (wave-uniform)
  if (__ballot(be.left || !be.hi || be.s2 != 0x07060504u) != 0ull) blur_strip_e<M, true>(src, pitch, h, dst, dpitch, be, x0, active, ys, nrows, nrows_l);
  else blur_strip_e<M, false>(src, pitch, h, dst, dpitch, be, x0, active, ys, nrows, nrows_l);
}

// 8 waves/SIMD (<= 64 VGPRs, no scratch since the edge lanes stopped gathering bytes from memory — that gather, 12
// dependent byte loads per row in every wave holding an edge lane, was the latency the kernel waited on: 0.308 ->
// 0.198 ms; 7 waves: 0.198, 8 waves: 0.195)
__attribute__((amdgpu_waves_per_eu(8, 8)))
__global__ __launch_bounds__(256) void blur_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, const unsigned* __restrict__ tile_tab) {
  int img, tile;
  if (!xcd_decode(xm, n_img, img, tile)) return;
  // strip table of orb_prepare_geometry: level | mode << 3 | (x / 4) << 5 | tile row << 18
  const unsigned e = tile_tab[tile];
  const int l = (int)(e & 7u), mode = (int)((e >> 3) & 3u), xbase = (int)((e >> 5) & 0x1fffu) * 4, ty = (int)(e >> 18);
  int pitch;
  const uint8_t* src = level_ptr(s, g, img, l, pitch);
  const int w = g.lv[l].w, h = g.lv[l].h;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: row arithmetic stays on the SALU
  const int ys = ty * BLUR_H + wave * BLUR_STRIP;
  if (ys >= h) return;
  uint8_t* dst = s.blur + (size_t)img * g.slot_bytes + g.lv[l].off;
  const int dpitch = g.lv[l].pitch;
  if (mode == 0) blur_strip<1>(src, pitch, w, h, dst, dpitch, xbase, ys, lane);
  else if (mode == 1) blur_strip<2>(src, pitch, w, h, dst, dpitch, xbase, ys, lane);
  else blur_strip<4>(src, pitch, w, h, dst, dpitch, xbase, ys, lane);
}

// ---- A.5 FAST-9/16 ------------------------------------------------------------------------------------------
// score = (max over the 16 arcs of 9 contiguous ring pixels of min(centre - ring), or of
// min(ring - centre)) - 1 when that maximum exceeds the threshold, else 0.
typedef short fast_s2 __attribute__((ext_vector_type(2)));

// One polarity only: max over the 16 arcs of min over the arc of s*(centre - ring), s = +1 (ring darker) or -1 (ring
// brighter), on packed 16-bit pairs.  A polarity whose compass pre-test fails cannot exceed the threshold (every
// 9-arc holds two adjacent compass points), so phase 2 evaluates only the polarity (rarely both) that passes.
// Packing: X[j] = (d[j], d[j+8]) — a register and its half-swap hold all 16 ring positions, so "position + k" is
// another register, half-swapped when it wraps past 8; the swap is the op_sel modifier of v_pk_min_i16 (free).
__device__ __forceinline__ fast_s2 pk_min_swapped(fast_s2 a, fast_s2 b) {   // min(a, (b.hi, b.lo))
  fast_s2 d;
  asm("v_pk_min_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ int fast_arc_min_pk(int v, const int (&r)[16], bool brighter) {
  fast_s2 X[8], A[8], B[8];
  const short sv = (short)(brighter ? -v : v), sm = (short)(brighter ? 1 : -1);
  const fast_s2 c = {sv, sv}, m = {sm, sm};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned rp = (unsigned)r[j] | ((unsigned)r[j + 8] << 16);
    X[j] = __builtin_bit_cast(fast_s2, rp) * m + c;                      // s*(centre - ring): one v_pk_mad_i16
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) A[j] = j + 1 < 8 ? __builtin_elementwise_min(X[j], X[j + 1]) : pk_min_swapped(X[j], X[j - 7]);   // window 2
#pragma unroll
  for (int j = 0; j < 8; ++j) B[j] = j + 2 < 8 ? __builtin_elementwise_min(A[j], A[j + 2]) : pk_min_swapped(A[j], A[j - 6]);   // 4
#pragma unroll
  for (int j = 0; j < 8; ++j) A[j] = j + 4 < 8 ? __builtin_elementwise_min(B[j], B[j + 4]) : pk_min_swapped(B[j], B[j - 4]);   // 8
  fast_s2 best = pk_min_swapped(A[0], X[0]);                                                                                   // 9
#pragma unroll
  for (int j = 1; j < 8; ++j) best = __builtin_elementwise_max(best, pk_min_swapped(A[j], X[j]));
  return max((int)best[0], (int)best[1]);
}

// The same maximum over the 16 arcs in packed f16 with the three-input minimum / maximum of gfx950 (v_pk_minimum3_f16,
// v_pk_maximum3_f16: one issue slot like every packed instruction, two comparisons per lane instead of one).  A pixel byte p IS
// an f16 bit pattern: the denormal p * 2^-24 (the kernels run with f16 denormals kept, .amdhsa_float_denorm_mode_16_64 3), and
// denormals order and add exactly like the integers they hold.  The score of a polarity s (+1 brighter, -1 darker) is
//     max over the 16 arcs of min over the arc of s*(ring - centre)  =  [max over arcs of min over the arc of s*ring] - s*centre,
// so the centre leaves the window ladder: s*ring is the sign bit (one v_or_b32, 2-cycle class, instead of a v_pk_fma_f16 per pair),
// 9-windows are min3 of min3 (X[j..j+2], then windows j, j+3, j+6: 16 instructions instead of the 2-4-8(+1) ladder's 32), their
// maximum is 5 maximum3, and one v_add_f16 of -s*centre gives the score, whose bits read as an int16 are the score itself when it
// is positive and a negative number otherwise (sign-magnitude, -0 included).
typedef _Float16 fast_h2 __attribute__((ext_vector_type(2)));
// minimum3 with the 2nd / 3rd operand half-swapped (ring position + 8 lives in the other half of the register)
__device__ __forceinline__ fast_h2 pk_min3_00(fast_h2 a, fast_h2 b, fast_h2 c) { fast_h2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ fast_h2 pk_min3_01(fast_h2 a, fast_h2 b, fast_h2 c) { fast_h2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ fast_h2 pk_min3_11(fast_h2 a, fast_h2 b, fast_h2 c) { fast_h2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ fast_h2 pk_max3(fast_h2 a, fast_h2 b, fast_h2 c) { fast_h2 d; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// low half = max(a.lo, a.hi) (high half unused)
__device__ __forceinline__ fast_h2 pk_max_halves(fast_h2 a) { fast_h2 d; asm("v_pk_maximum3_f16 %0, %1, %1, %1 op_sel:[0,1,1]" : "=v"(d) : "v"(a)); return d; }
// P[j] = ring[j] | ring[j + 8] << 16; returns the arc score of the polarity when it is positive, a negative number otherwise
__device__ __forceinline__ int fast_arc_score_h(unsigned v, const unsigned (&P)[8], bool brighter) {
  fast_h2 X[8], B[8], W[8];
  const unsigned sgn = brighter ? 0u : 0x80008000u;
#pragma unroll
  for (int j = 0; j < 8; ++j) X[j] = __builtin_bit_cast(fast_h2, P[j] ^ sgn);
#pragma unroll
  for (int j = 0; j < 8; ++j)                                         // windows of 3: positions j, j+1, j+2
    B[j] = j + 2 < 8 ? pk_min3_00(X[j], X[j + 1], X[j + 2]) : (j + 1 < 8 ? pk_min3_01(X[j], X[j + 1], X[j - 6]) : pk_min3_11(X[j], X[j - 7], X[j - 6]));
#pragma unroll
  for (int j = 0; j < 8; ++j)                                         // windows of 9: windows of 3 at j, j+3, j+6
    W[j] = j + 6 < 8 ? pk_min3_00(B[j], B[j + 3], B[j + 6]) : (j + 3 < 8 ? pk_min3_01(B[j], B[j + 3], B[j - 2]) : pk_min3_11(B[j], B[j - 5], B[j - 2]));
  const fast_h2 m1 = pk_max3(W[0], W[1], W[2]), m2 = pk_max3(W[3], W[4], W[5]), m3 = pk_max3(W[6], W[7], m1);
  const fast_h2 best = pk_max_halves(pk_max3(m2, m3, m3));
  const _Float16 mc = __builtin_bit_cast(_Float16, (unsigned short)(brighter ? (v | 0x8000u) : v));   // -s * centre
  const _Float16 b = best[0] + mc;
  return (int)__builtin_bit_cast(short, b);
}

// Two-phase per tile.  Score region 64 x (FT_H + 2) positions (inner 62 x FT_H + 1-position NMS frame), pixel tile
// 72 x (FT_H + 8) bytes staged so that position i sits at byte i+4 of its LDS row (dword aligned):
//   1. every position, 4 per thread from 5 dword LDS reads: compass pre-test — a 9-arc always contains
//      two adjacent compass points (0,4,8,12), so a corner needs two adjacent ones both brighter than
//      v+t or both darker than v-t; survivors (~10 %) are compacted into an LDS list;
//   2. full arc score only for the listed positions, written into the LDS score tile;
//   3. 3x3 NMS over the corners phase 2 found (second, much shorter list), block-aggregated append to the
//      level's candidates.
#ifndef ORBX_FT_H
#define ORBX_FT_H 62
#endif
#ifndef ORBX_FT_THREADS
#define ORBX_FT_THREADS 256
#endif
constexpr int FT_THREADS = ORBX_FT_THREADS;
// inner tile 62 x FT_H.  Round 1 (packed-i16 kernel): heights 30 / 46 / 62: 0.837 / 0.821 / 0.835 ms per 256 pairs.  After the round-2 work
// (chains, deferred append) with 256 threads: 46 / 54 / 58 / 62 / 66 / 70 / 78: 0.537 / 0.528 / 0.503 / 0.498 / 0.526 / 0.527 / 0.540 — fewer tile
// prologues and halo rows per pixel until the LDS footprint (21 KB at 62) takes a block away from the CU; 512 threads on 62 x 94: 0.568
constexpr int FT_W = 62, FT_H = ORBX_FT_H;
constexpr int FS_W = 64, FS_H = FT_H + 2;          // score region
#ifndef ORBX_FP_PITCH
#define ORBX_FP_PITCH 72
#endif
// (pixel-tile row pitch 72 / 76 / 80 / 88 / 104 bytes: 0.537 / 0.540 / 0.538 / 0.537 / 0.538 ms per 256 pairs — the 31 % bank-conflict share of
// the LDS pipe's cycles does not bind the kernel)
constexpr int FP_PITCH = ORBX_FP_PITCH, FP_ROWS = FS_H + 6;
#ifndef ORBX_FAST_CHAIN
#define ORBX_FAST_CHAIN 3
#endif
constexpr int FAST_CHAIN = ORBX_FAST_CHAIN;   // tiles per block
// The envelope of (ORBX_FT_H, ORBX_FT_THREADS, ORBX_FP_PITCH, ORBX_FAST_CHAIN) the kernel's index arithmetic is written for; a -D outside it
// fails to compile instead of running.  (Round 2 swept these macros with the guards removed and one run of the 62-row / 512-thread
// build ended in a GPU memory fault — DESIGN.md §4 "the f62t512 fault".)
constexpr int FAST_ST_RPP = FT_THREADS / 9;                                   // rows staged per pass: 9 lanes x 8 bytes per 72-byte row
constexpr int FAST_ST_PASS = (FP_ROWS + FAST_ST_RPP - 1) / FAST_ST_RPP;       // passes (one 8-byte register per pass and thread)
constexpr int FAST_LIST_CAP = ((FT_W + 1) / 2) * ((FT_H + 1) / 2);            // 3x3-NMS survivors of a tile: at most one per 2x2 positions
constexpr int FAST_LDS_BYTES = FP_ROWS * FP_PITCH + FS_H * FS_W + 2 * FS_W * FS_H + 4 * (FAST_LIST_CAP + 1) + 12;
static_assert(FT_W == 62 && FS_W == 64, "a position is stored as j * 64 + i and split by >> 6 / & 63; a task is 16 positions of a 64-wide row");
static_assert(FT_THREADS % 64 == 0 && FT_THREADS >= 128 && FT_THREADS <= 1024, "whole waves; the wave-level scans and ballots assume full waves");
static_assert(FT_H >= 8 && FT_H % 2 == 0 && FT_H <= 126, "tile height: even (2x2 NMS bound), score rows j < 128 so that j * 64 + i fits the 16-bit list");
static_assert(FP_PITCH >= 72 && FP_PITCH % 8 == 0, "a pixel row holds 9 aligned 8-byte chunks (positions 0..63 at bytes 4..67, ring reach +-3)");
static_assert(FAST_ST_RPP >= 1 && FAST_ST_RPP * FAST_ST_PASS >= FP_ROWS && FAST_ST_PASS <= 4, "the staging passes cover every pixel row with at most 4 prefetch registers");
static_assert(16 * ((FS_W + 15) / 16) * FS_H <= FS_W * FS_H && FS_W * FS_H <= 65536, "s_pos holds every position of the score region; entries are 16 bits");
static_assert(4 * FS_H < 65536 / 4 && 16 * FS_H < 65536 / 16, "task / tasks-per-row by the 16.16 reciprocal is exact below 65536 / tasks-per-row tasks");
static_assert(FP_PITCH >= 16 * ((FS_W + 15) / 16) + 8, "the byte-parallel pre-test reads dwords 4g .. 4g+5 of a pixel row");
static_assert((FS_W * FS_H) % 16 == 0, "the score tile is cleared with 16-byte stores");
static_assert(FAST_LDS_BYTES <= 65536, "LDS per block (160 KB per CU / this = resident blocks: 7 at the shipped 62 x 62, 256 threads)");
static_assert(FAST_CHAIN >= 1 && FAST_CHAIN <= 16, "tiles per block");

__device__ __forceinline__ unsigned ld_u32(const uint8_t* p);

// ---- byte-parallel compass pre-test (thresholds below 128) ---------------------------------------------------------
// The probe (profiles/r02_valu_issue_probe.txt) shows two issue classes on this chip: the packed-i16 / v_perm / v_cmp code of the
// pre-test above issues every 4 cycles, plain add / sub / and / or / xor / shift-right and v_bitop3_b32 every 2.  The same test on
// the four pixels of a dword at once, in those 2-cycle instructions only (the compiler folds the boolean expressions into
// v_bitop3_b32): per byte  hi = min(v + t, 255), lo = max(v - t, 0)  and the unsigned comparisons  hi >= r,  r >= lo  by the
// borrow of (a | 0x80) - (b & 0x7f) combined with the top bits.  Result: bit 7 of byte k set iff position k passes — exactly
// the positions the packed-i16 form passes (r > v + t can only hold for v + t < 255, r < v - t only for v - t > 0).
// v_bitop3_b32 look-up tables: bit (a << 2 | b << 1 | c) of the immediate is f(a, b, c).  (Written as C expressions the compiler
// emits separate and / or / xor / not instructions: 182 instead of 88 for two dwords.)
#define ORBX_BITOP3(a, b, c, lut) __builtin_amdgcn_bitop3_b32((a), (b), (c), (lut))
__device__ __forceinline__ unsigned swar_ge(unsigned a, unsigned aH /* a | 0x80.. */, unsigned b, unsigned bL /* b & 0x7f.. */) {
  const unsigned d = aH - bL;                      // per byte (128 + a_lo) - b_lo >= 1: no borrow crosses a byte; bit 7 = [a_lo >= b_lo]
  return ORBX_BITOP3(a, b, d, 0xB2);               // (a & ~b) | (~(a ^ b) & d): bit 7 of every byte = [a >= b] (other bits unused)
}
#ifndef ORBX_SWAR_SATURATED
// The thresholds WITHOUT saturation and every comparison in the form [threshold >= r], so a ring dword is prepared once (r & 0x7f..):
//   brighter:  r > v + t.   hi8 = (v + t) mod 256 = s ^ vH; where the byte overflows (o) nothing is brighter, and o joins the combine
//   darker:    r < v - t  <=>  v - t - 1 >= r  where v - t - 1 >= 0 (valid), nothing is darker elsewhere; T1 = (t + 1) per byte
// 34 instead of 42 vector instructions per dword, all in the 2-cycle class; the same positions pass (tests compare with the oracle).
__device__ __forceinline__ unsigned swar_compass_pass(unsigned v, unsigned r0, unsigned r4, unsigned r8, unsigned r12, unsigned T, unsigned T1) {
  const unsigned H = 0x80808080u, L = 0x7f7f7f7fu;
  const unsigned vH = v & H;
  const unsigned s = (v & L) + T;                  // low seven bits + t <= 0xfe: no carry crosses a byte
  const unsigned o = ORBX_BITOP3(v, s, H, 0x80);   // v & s & H: v + t >= 256 in this byte
  const unsigned hi = s ^ vH;                      // (v + t) mod 256
  const unsigned hiH = s | H;                      // hi | H
  const unsigned d = (v | H) - T1;                 // (128 + v_lo) - (t + 1) >= 0
  const unsigned w = ORBX_BITOP3(v, d, H, 0xA8);   // (v | d) & H: v - t - 1 >= 0 in this byte
  const unsigned lo = ORBX_BITOP3(d, vH, L, 0xE0); // d & (vH | L) = v - t - 1 where valid
  const unsigned loH = d | H;                      // lo | H
  const unsigned q0 = r0 & L, q4 = r4 & L, q8 = r8 & L, q12 = r12 & L;
  const unsigned g0 = ORBX_BITOP3(hi, r0, hiH - q0, 0xB2), g8 = ORBX_BITOP3(hi, r8, hiH - q8, 0xB2);      // [v + t >= r]: not brighter
  const unsigned g4 = ORBX_BITOP3(hi, r4, hiH - q4, 0xB2), g12 = ORBX_BITOP3(hi, r12, hiH - q12, 0xB2);
  const unsigned f0 = ORBX_BITOP3(lo, r0, loH - q0, 0xB2), f8 = ORBX_BITOP3(lo, r8, loH - q8, 0xB2);      // [v - t - 1 >= r]: darker
  const unsigned f4 = ORBX_BITOP3(lo, r4, loH - q4, 0xB2), f12 = ORBX_BITOP3(lo, r12, loH - q12, 0xB2);
  const unsigned nb = ORBX_BITOP3(ORBX_BITOP3(g0, g8, o, 0xEA), g4, g12, 0xF8);    // o | (g0 & g8) | (g4 & g12): no two adjacent compass points brighter
  const unsigned dk = ORBX_BITOP3(ORBX_BITOP3(f0, f8, w, 0xA8), f4, f12, 0xE0);    // w & (f0 | f8) & (f4 | f12): two adjacent ones darker
  return ORBX_BITOP3(nb, dk, H, 0x8A);             // (~nb | dk) & H
}
#else
__device__ __forceinline__ unsigned swar_compass_pass(unsigned v, unsigned r0, unsigned r4, unsigned r8, unsigned r12, unsigned T, unsigned) {
  const unsigned H = 0x80808080u, L = 0x7f7f7f7fu;
  const unsigned vH = v & H;
  const unsigned s = (v & L) + T;                  // low seven bits + t <= 0xfe: no carry crosses a byte
  const unsigned o = ORBX_BITOP3(v, s, H, 0x80);   // v & s & H: the byte overflows (v >= 128 and low sum >= 128)
  const unsigned hi = ORBX_BITOP3(s, vH, o - (o >> 7), 0xFE);          // s | vH | 0x7f there: min(v + t, 255)
  const unsigned d = (v | H) - T;                  // (128 + v_lo) - t >= 1
  const unsigned w = ORBX_BITOP3(v, d, H, 0xA8);   // (v | d) & H: v - t >= 0 in this byte
  const unsigned lo = ORBX_BITOP3(d, w - (w >> 7), vH, 0xE0);          // d & (mask | vH): max(v - t, 0)
  const unsigned hiH = hi | H, loL = lo & L;
  const unsigned g0 = swar_ge(hi, hiH, r0, r0 & L), g8 = swar_ge(hi, hiH, r8, r8 & L);
  const unsigned g4 = swar_ge(hi, hiH, r4, r4 & L), g12 = swar_ge(hi, hiH, r12, r12 & L);
  const unsigned e0 = swar_ge(r0, r0 | H, lo, loL), e8 = swar_ge(r8, r8 | H, lo, loL);
  const unsigned e4 = swar_ge(r4, r4 | H, lo, loL), e12 = swar_ge(r12, r12 | H, lo, loL);
  const unsigned nb = ORBX_BITOP3(g0 & g8, g4, g12, 0xF8);             // (r0 <= hi and r8 <= hi) or (r4 <= hi and r12 <= hi): not brighter
  const unsigned nd = ORBX_BITOP3(e0 & e8, e4, e12, 0xF8);             // likewise not darker
  return ORBX_BITOP3(nb, nd, H, 0x2A);             // ~(nb & nd) & H: two adjacent compass points brighter than v + t, or darker than v - t
}
#endif
// One returning LDS add by the calling lane(s) as written.  (atomicAdd of a wave-uniform value under `if (lane == 0)` goes through the
// compiler's atomic optimizer all the same: mbcnt of the one-lane exec mask, s_bcnt1, a v_mul_lo_u32 of the value by the lane's rank —
// eleven instructions around the ds_add_rtn_u32, in every round of fast_kernel's phase 1.)
__device__ __forceinline__ int lds_add_rtn(int* p, int v) {
  typedef __attribute__((address_space(3))) int lds_int;
  const unsigned a = (unsigned)(size_t)(lds_int*)p;
  int r;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory");
  return r;
}
// bits 7, 15, 23, 31 -> bits 0..3
__device__ __forceinline__ unsigned swar_movemask(unsigned p) { return (p * 0x00204081u) >> 28; }
// inclusive prefix sum over the 64 lanes of a wave (DPP row shifts + row broadcasts)
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
  return v;
}

// SWAR: phase 1 in the byte-parallel form, sixteen positions per task (fast_threshold < 128; launch_orb_extract picks the variant)
// (waves_per_eu(8, 8) keeps the register allocator at 64 VGPRs although 21 KB of LDS admit 7 blocks per CU: with (7, 8) it takes
// more registers and the kernel runs 0.512 instead of 0.498 ms.  The per-(image, level) retainBest thresholds precomputed by a small
// kernel instead of in each of harris_select_kernel's eight blocks: 0.149 -> 0.146 ms, not kept.)
template <bool SWAR>
__attribute__((amdgpu_waves_per_eu(8, 8)))
__global__ __launch_bounds__(FT_THREADS) void fast_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, const unsigned* __restrict__ tile_tab,
                                                   unsigned* __restrict__ cand, unsigned* __restrict__ cand_count,
                                                   unsigned* __restrict__ hist, int n_tiles, int chain_len, unsigned* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t sp[FP_ROWS][FP_PITCH];
  __shared__ __attribute__((aligned(16))) uint8_t ss[FS_H][FS_W];
  __shared__ __attribute__((aligned(4))) unsigned short s_pos[FS_W * FS_H];
  __shared__ unsigned s_list[FAST_LIST_CAP + 1];   // NMS survivors: at most one per 2x2 positions
  __shared__ int s_npos, s_cnt;
  __shared__ unsigned s_base;
  // A block works through FAST_CHAIN consecutive tiles of one image and loads the NEXT tile's pixels into registers while it works
  // on the current one: with one tile per block the chain  kernel arguments -> tile table -> pixels  (about 2 us) is exposed in
  // every block (phase-skipping builds measured staging alone at 0.18 ms of 0.63 per 256 pairs, added to the compute phases rather
  // than hidden under the other blocks of the CU).  Short chains keep the hardware's dynamic block dispatch as the load balancer:
  // fully persistent blocks measured worse both ways — a fixed stride 0.71 ms (tile costs differ, the slowest of 2048 fixed shares
  // sets the time) and tickets from 8 per-XCD queues 2.0 ms (170k same-address device atomics serialise at ~100 ns each).
  const int tid = threadIdx.x, lane = tid & 63;
  const int st_r0 = tid / 9, st_c = tid - 9 * st_r0;               // staging: 9 lanes x 8 bytes per 72-byte row, FT_THREADS / 9 rows per pass
  constexpr int ST_RPP = FAST_ST_RPP, ST_PASS = FAST_ST_PASS;
  unsigned long long pf[ST_PASS];
  struct Tile { int img, l, x0, y0, w, h, aw, ah; };
  Tile cur, nxt;
  // tile -> geometry, and the loads of its pixels into pf
  auto fetch = [&](int img_, int tile, Tile& c) {
    c.img = img_;
    int tx, ty;
    decode_tile(tile_tab, tile, c.l, tx, ty);
    int pitch;
    const uint8_t* src = level_ptr(s, g, c.img, c.l, pitch);
    c.w = g.lv[c.l].w; c.h = g.lv[c.l].h;
    c.x0 = EDGE + tx * FT_W; c.y0 = EDGE + ty * FT_H;   // first inner pixel; score position (i,j) = pixel (x0-1+i, y0-1+j)
    // score positions this tile needs (inner part that lies inside the border-filtered region + the NMS frame): tiles on
    // the right / bottom edge of a level are partial — 28 % of all tile area at 752x480 — and only pay for what they hold
    c.aw = min(FT_W, c.w - EDGE - c.x0) + 2; c.ah = min(FT_H, c.h - EDGE - c.y0) + 2;
    // pixel tile: LDS row r = image row y0-4+r, byte b = image column x0-5+b
#pragma unroll
    for (int k = 0; k < ST_PASS; ++k) {
      const int r = st_r0 + ST_RPP * k;
      if (st_r0 < ST_RPP && r < c.ah + 6) {
        const int gy = min(c.y0 - 4 + r, c.h - 1), gx = min(c.x0 - 5 + 8 * st_c, pitch - 8);
        __builtin_memcpy(&pf[k], src + (size_t)gy * pitch + gx, 8);
      }
    }
  };
  int img0, chain;
  if (!xcd_decode(xm, n_img, img0, chain)) return;
  const int tile0 = chain * chain_len, tile_end = min(tile0 + chain_len, n_tiles);
  fetch(img0, tile0, cur);
  // The corners of a chain's tiles gather in s_list and go to their level's list TOGETHER: the append needs a returning global atomic (about
  // 2 us) — per tile it cost 0.086 of the kernel's 0.91 ms per 512 pairs (profiles/r05_fast_phase1_steps.txt) although it was consumed one phase
  // into the next tile.  s_list holds FAST_LIST_CAP entries; before a tile's phases 2 and 3 the gathered corners are appended first if they belong
  // to another level or might not fit beside this tile's (at most min(npos, FAST_LIST_CAP) strict maxima: every one is a pre-test survivor).
  int acc_cnt = 0, acc_il = 0;
  unsigned acc_off = 0, acc_cap = 0;
  auto append = [&]() {                                              // block-uniform: every thread calls it
    if (tid == 0) s_base = atomicAdd(&cand_count[acc_il], (unsigned)acc_cnt);
    __syncthreads();
    for (int q = tid; q < acc_cnt; q += FT_THREADS) {
      const unsigned c = s_list[q];
      const unsigned slot = s_base + (unsigned)q;
      // (cannot happen: a level's list is sized for the worst case of 3x3-NMS survivors — a slot beyond it is reported, never written)
      if (slot >= acc_cap) { atomicOr(status, ORBX_ST_INTERNAL); continue; }
      cand[(size_t)img0 * g.cand_total + acc_off + slot] = c;
      atomicAdd(&hist[(size_t)acc_il * 256 + (c >> 24)], 1u);
    }
    __syncthreads();                                                 // s_list is read before phase 3 appends to it again
    if (tid == 0) s_cnt = 0;                                         // (phase 3 sits behind another barrier)
    acc_cnt = 0;
  };
  if (tid == 0) s_cnt = 0;
  for (int tile = tile0; tile < tile_end; ++tile) {
  const int img = cur.img, l = cur.l, x0 = cur.x0, y0 = cur.y0, w = cur.w, h = cur.h, aw = cur.aw, ah = cur.ah;
  if (tid == 0) s_npos = 0;
  const int qpr = (aw + 3) >> 2;                                   // 4-position tasks per row, 1..16
  const int ntask = qpr * ah;
  const unsigned inv = (unsigned)(65536.f / (float)qpr) + 1u;      // task / qpr = (task * inv) >> 16, exact for task < 512
#pragma unroll
  for (int k = 0; k < ST_PASS; ++k) {
    const int r = st_r0 + ST_RPP * k;
    if (st_r0 < ST_RPP && r < ah + 6) *reinterpret_cast<unsigned long long*>(&sp[r][8 * st_c]) = pf[k];
  }
#pragma unroll
  for (int i = tid; i < FS_W * FS_H / 16; i += FT_THREADS) reinterpret_cast<uint4*>(&ss[0][0])[i] = uint4{0u, 0u, 0u, 0u};   // 16-byte stores clear the score tile
  __syncthreads();
  // the next tile's pixels travel while this one is processed (the barriers below wait for LDS only, not for these loads)
  if (tile + 1 < tile_end) fetch(img0, tile + 1, nxt);
  const int t = g.fast_threshold;
  if (SWAR) {
    // phase 1, byte-parallel: a task = 16 consecutive positions of a row = the four centre dwords 4g+1 .. 4g+4 of LDS row j+3 (position
    // i sits at byte i+4; dwords 4g and 4g+5 supply the x-3 / x+3 neighbours), so a full tile is ONE round of the block: one wave prefix
    // sum, one returning LDS add and one walk over each lane's set bits per 16 positions (with 8 positions per task — two rounds — those
    // were a quarter of the phase).  The positions that pass go to the list by the prefix sum of the lanes' counts (DPP scan) and a short
    // loop over each lane's set bits — no ballot per position
    const int qpr16 = (aw + 15) >> 4;                                // 16-position tasks per row, 1..4
    const int ntask16 = qpr16 * ah;
    const unsigned inv16 = (unsigned)(65536.f * __builtin_amdgcn_rcpf((float)qpr16)) + 1u;   // (v_rcp_f32 is within 1 ulp: the floor is the quotient's for 1..4)
    const unsigned T = (unsigned)t * 0x01010101u;
    const unsigned T1 = T + 0x01010101u;                             // t + 1 <= 128 per byte
    for (int task0 = 0; task0 < ntask16; task0 += FT_THREADS) {
      const int task = task0 + tid;
      unsigned m16 = 0;
      int j = 0, gq = 0;
      if (task < ntask16) {
        j = (int)(__umul24((unsigned)task, inv16) >> 16); gq = task - (int)__umul24((unsigned)j, (unsigned)qpr16);
        const uint2* crow = reinterpret_cast<const uint2*>(&sp[j + 3][16 * gq]);           // dwords 4g .. 4g+5 (8-byte aligned: the pitch is a multiple of 8)
        const uint2 c01 = crow[0], c23 = crow[1], c45 = crow[2];
        const unsigned* upr = reinterpret_cast<const unsigned*>(&sp[j][16 * gq + 4]);
        const unsigned* dnr = reinterpret_cast<const unsigned*>(&sp[j + 6][16 * gq + 4]);
        const unsigned up0 = upr[0], up1 = upr[1], up2 = upr[2], up3 = upr[3], dn0 = dnr[0], dn1 = dnr[1], dn2 = dnr[2], dn3 = dnr[3];
        const unsigned p0 = swar_compass_pass(c01.y, dn0, __builtin_amdgcn_alignbyte(c23.x, c01.y, 3), up0, __builtin_amdgcn_alignbyte(c01.y, c01.x, 1), T, T1);
        const unsigned p1 = swar_compass_pass(c23.x, dn1, __builtin_amdgcn_alignbyte(c23.y, c23.x, 3), up1, __builtin_amdgcn_alignbyte(c23.x, c01.y, 1), T, T1);
        const unsigned p2 = swar_compass_pass(c23.y, dn2, __builtin_amdgcn_alignbyte(c45.x, c23.y, 3), up2, __builtin_amdgcn_alignbyte(c23.y, c23.x, 1), T, T1);
        const unsigned p3 = swar_compass_pass(c45.x, dn3, __builtin_amdgcn_alignbyte(c45.y, c45.x, 3), up3, __builtin_amdgcn_alignbyte(c45.x, c23.y, 1), T, T1);
        // bits 7, 15, 23, 31 of p0 -> bits 0..3, of p1 -> bits 4..7 by ONE multiply: (p0 >> 4) | p1 holds position b at bit 8 b + 3 and
        // position 4 + b at bit 8 b + 7, the partial products of 0x00204081 (shifts 21 - 7 b) put them at bits 24 + b and 28 + b, no two
        // partial products share a bit and no other one reaches bits 24..31
        m16 = ((((p0 >> 4) | p1) * 0x00204081u) >> 24) | (((((p2 >> 4) | p3) * 0x00204081u) >> 16) & 0xff00u);
      }
      const int cnt = __popc(m16);
      const int incl = wave_scan_incl(cnt);
      const int wtot = __builtin_amdgcn_readlane(incl, 63);
      if (wtot) {                                                    // wave-uniform
        int base = 0;
        if (lane == 0) base = lds_add_rtn(&s_npos, wtot);
        int off = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
        const unsigned short pbase = (unsigned short)(j * FS_W + 16 * gq);
        for (unsigned mm = m16; mm; mm &= mm - 1u) s_pos[off++] = (unsigned short)(pbase + (__ffs((int)mm) - 1));
      }
    }
  } else
  // phase 1: compass pre-test, 4 positions per task
  for (int task = tid; task < ntask; task += FT_THREADS) {
    const int j = (int)(__umul24((unsigned)task, inv) >> 16), tq = task - (int)__umul24((unsigned)j, (unsigned)qpr);   // (v_mul_lo_u32 is quarter rate)
    const unsigned* rowc = reinterpret_cast<const unsigned*>(&sp[j + 3][0]) + tq;
    const unsigned c0 = rowc[0], c1 = rowc[1], c2 = rowc[2];
    const unsigned up = reinterpret_cast<const unsigned*>(&sp[j][0])[tq + 1];       // y-3
    const unsigned dn = reinterpret_cast<const unsigned*>(&sp[j + 6][0])[tq + 1];   // y+3
    // the 4 positions as two packed i16 pairs (v_perm_b32 widens bytes, v_pk_min/max/sub_i16 test two positions
    // per instruction)
    const unsigned xm3 = __builtin_amdgcn_alignbyte(c1, c0, 1);   // x-3 of the 4 positions: bytes 4tq+1 .. 4tq+4
    const unsigned xp3 = __builtin_amdgcn_alignbyte(c2, c1, 3);   // x+3: bytes 4tq+7 .. 4tq+10
    const fast_s2 tt = {(short)t, (short)t};
    unsigned pass[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const unsigned sel = hf ? 0x0c030c02u : 0x0c010c00u;
      const fast_s2 v2 = __builtin_bit_cast(fast_s2, __builtin_amdgcn_perm(0u, c1, sel));
      const fast_s2 r0 = __builtin_bit_cast(fast_s2, __builtin_amdgcn_perm(0u, dn, sel));
      const fast_s2 r4 = __builtin_bit_cast(fast_s2, __builtin_amdgcn_perm(0u, xp3, sel));
      const fast_s2 r8 = __builtin_bit_cast(fast_s2, __builtin_amdgcn_perm(0u, up, sel));
      const fast_s2 r12 = __builtin_bit_cast(fast_s2, __builtin_amdgcn_perm(0u, xm3, sel));
      // every adjacent compass pair takes one point of {0, 8} and one of {4, 12}, and every such pair is adjacent:
      // "two adjacent ones brighter than v+t"  <=>  max(r0,r8) > v+t and max(r4,r12) > v+t; darker likewise with min
      const fast_s2 B = __builtin_elementwise_min(__builtin_elementwise_max(r0, r8), __builtin_elementwise_max(r4, r12));
      const fast_s2 D = __builtin_elementwise_max(__builtin_elementwise_min(r0, r8), __builtin_elementwise_min(r4, r12));
      const fast_s2 eb = (v2 + tt) - B, ed = D - (v2 - tt);   // negative exactly where brighter / darker passes
      pass[hf] = (__builtin_bit_cast(unsigned, eb) | __builtin_bit_cast(unsigned, ed)) & 0x80008000u;
    }
    // wave-level compaction by ballots (list order is free: phases 2/3 only need the set)
    const bool pk0 = pass[0] & 0x8000u, pk1 = pass[0] & 0x80000000u, pk2 = pass[1] & 0x8000u, pk3 = pass[1] & 0x80000000u;
    const unsigned long long b0 = __ballot(pk0), b1 = __ballot(pk1), b2 = __ballot(pk2), b3 = __ballot(pk3);
    const int n0 = __popcll(b0), n1 = __popcll(b1), n2 = __popcll(b2), n3 = __popcll(b3);
    const int wtot = n0 + n1 + n2 + n3;
    if (wtot) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_npos, wtot);
      base = __builtin_amdgcn_readfirstlane(base);
      const unsigned short p0 = (unsigned short)(j * FS_W + 4 * tq);
      if (pk0) s_pos[base + __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u))] = p0;
      base += n0;
      if (pk1) s_pos[base + __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u))] = p0 + 1;
      base += n1;
      if (pk2) s_pos[base + __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u))] = p0 + 2;
      base += n2;
      if (pk3) s_pos[base + __builtin_amdgcn_mbcnt_hi((unsigned)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b3, 0u))] = p0 + 3;
    }
  }
  __syncthreads();
  const int npos = s_npos;
  if (acc_cnt > 0 && (acc_il != img * g.n_levels + l || acc_cnt + min(npos, FAST_LIST_CAP) > FAST_LIST_CAP)) append();
  // phase 2: full score of the pre-test survivors
  for (int q = tid; q < npos; q += FT_THREADS) {
    const int p = s_pos[q], j = p >> 6, i = p & 63;
    const int cy = j + 3, cx = i + 4;
    const unsigned v = sp[cy][cx];
    unsigned r[16];
    r[0] = sp[cy + 3][cx];      r[1] = sp[cy + 3][cx + 1];  r[2] = sp[cy + 2][cx + 2];
    r[3] = sp[cy + 1][cx + 3];  r[4] = sp[cy][cx + 3];      r[5] = sp[cy - 1][cx + 3];
    r[6] = sp[cy - 2][cx + 2];  r[7] = sp[cy - 3][cx + 1];  r[8] = sp[cy - 3][cx];
    r[9] = sp[cy - 3][cx - 1];  r[10] = sp[cy - 2][cx - 2]; r[11] = sp[cy - 1][cx - 3];
    r[12] = sp[cy][cx - 3];     r[13] = sp[cy + 1][cx - 3]; r[14] = sp[cy + 2][cx - 2];
    r[15] = sp[cy + 3][cx - 1];
    // (one unaligned LDS read per ring row, 2 x b32 + 5 x b64, plus a v_perm_b32 per pair measured 1.10 ms against 0.63: the
    // unaligned reads are slow)
    // which polarity passed the compass pre-test (the other one cannot exceed the threshold)
    const int hi = (int)v + t, lo = (int)v - t;
    const bool brighter = (int)min(max(r[0], r[8]), max(r[4], r[12])) > hi;
    const bool darker = (int)max(min(r[0], r[8]), min(r[4], r[12])) < lo;
    unsigned P[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) P[k] = r[k] | (r[k + 8] << 16);
#ifdef ORBX_FAST_NO_LADDER
    // measurement build (NOT the specification): the 16 ring reads and their packing stay, the min / max ladder of the arc score is replaced by
    // two instructions — what ANY cheaper test in front of the ladder (a contiguity bit-mask) could save at the very most
    int b = (int)((P[0] ^ P[1] ^ P[2] ^ P[3] ^ P[4] ^ P[5] ^ P[6] ^ P[7]) & 0xffu) + (brighter ? 1 : 0);
#else
    int b = fast_arc_score_h(v, P, brighter);
    if (brighter && darker) b = max(b, fast_arc_score_h(v, P, false));    // both passed (rare)
#endif
    const int sc = b > t ? b - 1 : 0;
    ss[j][i] = (uint8_t)sc;
  }
  __syncthreads();
  // phase 3: NMS (strictly greater than the 8 neighbours) + runByImageBorder, over the same list: corners inside the tile and the
  // border-filtered region (a few per cent of the survivors — the pre-test also passes every straight edge)
  for (int q = tid; q < npos; q += FT_THREADS) {
    const int p = s_pos[q];
    const int j = p >> 6, i = p & 63;
    const int x = x0 + i - 1, y = y0 + j - 1;
    const int sc = ss[j][i];
    const bool corner = sc > 0 && i >= 1 && i <= FT_W && j >= 1 && j <= FT_H && x < w - EDGE && y < h - EDGE;
    if (corner) {
      const int m = max(max(max((int)ss[j - 1][i - 1], (int)ss[j - 1][i]), max((int)ss[j - 1][i + 1], (int)ss[j][i - 1])),
                        max(max((int)ss[j][i + 1], (int)ss[j + 1][i - 1]), max((int)ss[j + 1][i], (int)ss[j + 1][i + 1])));
      if (sc > m) {
        const int pos = atomicAdd(&s_cnt, 1);
        if (pos < FAST_LIST_CAP) s_list[pos] = ((unsigned)sc << 24) | ((unsigned)y << 12) | (unsigned)x;
        else atomicOr(status, ORBX_ST_INTERNAL);    // (cannot happen: strict maxima of a 62 x FT_H tile are at most one per 2x2)
      }
    }
  }
  __syncthreads();
  acc_cnt = min(s_cnt, FAST_LIST_CAP);                               // (s_cnt keeps counting across the chain's tiles until an append resets it)
  acc_il = img * g.n_levels + l;
  acc_off = g.lv[l].cand_off;
  acc_cap = g.lv[l].cand_cap;
  // (no barrier here: the next round writes pixels, scores and s_npos before its first barrier, the position list after it, and
  // s_cnt / s_base / s_list only behind that barrier)
  cur = nxt;
  }
  if (acc_cnt > 0) append();
}

__device__ __forceinline__ unsigned ld_u32(const uint8_t* p) {   // unaligned dword load (global_load_dword)
  unsigned v;
  __builtin_memcpy(&v, p, 4);
  return v;
}

// ---- A.6 Harris response (blockSize 7, k 0.04) of one candidate, one thread ------------------------------
__device__ __forceinline__ float harris_response(const uint8_t* __restrict__ img, int pitch, int x0, int y0) {
  int a = 0, b = 0, c = 0;
  // 9x9 window as 9 rows x 3 unaligned dwords (27 loads instead of 81 byte loads; one unaligned global_load_dwordx3 per row — 9 loads —
  // measured 0.159 against 0.150 ms per 256 pairs, round 4: the 12-byte form does not go through the address coalescer as three dwords do)
  int rowm[9], row0[9], rowp[9];
  // Round 5: each row as ONE 12-byte load on a 4-byte boundary (level rows start on 4-byte boundaries and x0 - 4 - sh + 11 < w), the window
  // shifted into place by v_alignbyte_b32: 9 aligned loads instead of 27 dwords at odd addresses, 0.142 -> 0.139 ms per 256 pairs
  // (ORBX_HARRIS_UNALIGNED: the three dwords at x0 - 4 as before, A/B builds)
  unsigned w[9][3];
#ifdef ORBX_HARRIS_UNALIGNED
  const uint8_t* p = img + (size_t)(y0 - 4) * pitch + (x0 - 4);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const uint8_t* pr = p + (size_t)r * pitch;
    w[r][0] = ld_u32(pr); w[r][1] = ld_u32(pr + 4); w[r][2] = ld_u32(pr + 8);
  }
#else
  const unsigned sh = (unsigned)(x0 - 4) & 3u;
  const uint8_t* p = img + (unsigned)(__umul24((unsigned)(y0 - 4), (unsigned)pitch) + ((unsigned)(x0 - 4) & ~3u));
  struct alignas(4) Row12 { unsigned d[3]; };
  Row12 rw[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) rw[r] = *reinterpret_cast<const Row12*>(p + (unsigned)r * (unsigned)pitch);
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    w[r][0] = __builtin_amdgcn_alignbyte(rw[r].d[1], rw[r].d[0], sh);
    w[r][1] = __builtin_amdgcn_alignbyte(rw[r].d[2], rw[r].d[1], sh);
    w[r][2] = rw[r].d[2] >> (8u * sh);
  }
#endif
#define ORBX_ROW(dst, r)                                                                    \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                             \
    dst[k] = (int)((w[r][0] >> (8 * k)) & 0xffu);                                             \
    dst[4 + k] = (int)((w[r][1] >> (8 * k)) & 0xffu);                                         \
  }                                                                                           \
  dst[8] = (int)(w[r][2] & 0xffu);
  ORBX_ROW(rowm, 0)
  ORBX_ROW(row0, 1)
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    ORBX_ROW(rowp, i + 2)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int Ix = (row0[j + 2] - row0[j]) * 2 + (rowm[j + 2] - rowm[j]) + (rowp[j + 2] - rowp[j]);
      const int Iy = (rowp[j + 1] - rowm[j + 1]) * 2 + (rowp[j] - rowm[j]) + (rowp[j + 2] - rowm[j + 2]);
      a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { rowm[k] = row0[k]; row0[k] = rowp[k]; }
  }
#undef ORBX_ROW
  const float scale = 1.f / (4 * 7 * 255.f);
  const float scale_sq_sq = scale * scale * scale * scale;   // folded at compile time, IEEE
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  const float t1 = __fmul_rn(fa, fb);
  const float t2 = __fmul_rn(fc, fc);
  const float sm = __fadd_rn(fa, fb);
  const float t3 = __fmul_rn(__fmul_rn(0.04f, sm), sm);
  float r = __fmul_rn(__fsub_rn(__fsub_rn(t1, t2), t3), scale_sq_sq);
  if (r == 0.f) r = 0.f;
  return r;
}

// (Round 4: the same response on packed i16 pairs — v_perm_b32 widening, packed column sums / differences, v_dot2_i32_i16 for the three sums
// of products — was written and compiled: 665 vector instructions in the kernel against 777, the compiler already shares the column sums of this
// scalar form; at 45 % VALU share of a 0.15 ms kernel that is below 1 % of the step.  Not kept.)
// float -> u32 whose unsigned order is the float order
__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// blocks of harris_select_kernel per image: level l owns blocks start[l] .. start[l + 1] - 1, each a contiguous share of the level's candidates
struct HarrisPlan { int start[ORBX_MAX_LEVELS + 1]; };
// retainBest(2*n_l) by FAST score (threshold from the histogram), then Harris of each survivor.
// key = (~orderable(response) << 32) | y << 16 | x : ascending key = canonical order.
__global__ __launch_bounds__(256) void harris_select_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, const unsigned* __restrict__ cand,
                                                            const unsigned* __restrict__ cand_count,
                                                            const unsigned* __restrict__ hist,
                                                            unsigned long long* __restrict__ sel,
                                                            unsigned* __restrict__ sel_count, HarrisPlan hp) {
  __shared__ unsigned sh[256];
  __shared__ int s_thr;
  int img, bb;
  if (!xcd_decode(xm, n_img, img, bb)) return;
  int l = 0;
#pragma unroll
  for (int k = 1; k < ORBX_MAX_LEVELS; ++k) l += (bb >= hp.start[k]) ? 1 : 0;             // (start[] is non-decreasing; levels beyond n_levels start at the total)
  const int chunk = bb - hp.start[l], n_chunks = hp.start[l + 1] - hp.start[l];
  const int il = img * g.n_levels + l;
  const int tid = threadIdx.x;
  const unsigned count = min(cand_count[il], g.lv[l].cand_cap);   // (fast_kernel never writes beyond the level's list)
  const unsigned want = 2u * (unsigned)g.lv[l].quota;
  // threshold = largest score sc with #(score >= sc) >= want (the n-th best score; ties with it are kept), by a
  // parallel suffix sum over the 256 histogram bins (thread = bin)
  {
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned hv = hist[(size_t)il * 256 + tid];
    const int incl = wave_scan_incl((int)hv);                     // (DPP; six __shfl_down steps were six LDS round trips at the head of every block's chain)
    unsigned suf = (unsigned)(__builtin_amdgcn_readlane(incl, 63) - incl) + hv;   // suffix sum inside the wave
    if (lane == 0) sh[wave] = suf;
    if (tid == 0) s_thr = 0;
    __syncthreads();
    for (int w2 = wave + 1; w2 < 4; ++w2) suf += sh[w2];
    const unsigned long long m = __ballot(suf >= want);
    if (want == 0) { if (tid == 0) s_thr = 256; }                 // retainBest(0) clears
    else if (count > want && m && lane == 0) atomicMax(&s_thr, wave * 64 + 63 - __clzll((long long)m));
    __syncthreads();
  }
  const unsigned thr = (unsigned)s_thr;
  int pitch;
  const uint8_t* src = level_ptr(s, g, img, l, pitch);
  const unsigned* cl = cand + (size_t)img * g.cand_total + g.lv[l].cand_off;
  unsigned long long* out = sel + (size_t)img * g.cand_total + g.lv[l].cand_off;
  // Two steps per round of 1024 candidates: (1) the ones at or above the threshold (typically a third) are compacted into
  // an LDS list, (2) the list is worked off densely — the 9x9 response costs ~750 instructions, so a wave should not
  // carry lanes that failed the threshold.  Appends to the level's survivor list take one atomic per wave.
  __shared__ unsigned s_pass[1024];
  __shared__ unsigned s_np;
  const int lane = tid & 63;
  const unsigned per = (count + (unsigned)n_chunks - 1u) / (unsigned)n_chunks;         // this block's contiguous share
  const unsigned lim = min(count, (chunk + 1) * per);
  for (unsigned base = chunk * per; base < lim; base += 1024u) {                       // block-uniform bounds
    if (tid == 0) s_np = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned i = base + k * 256 + tid;
      const unsigned c = i < lim ? cl[i] : 0u;
      const bool pass = i < lim && (c >> 24) >= thr;
      const unsigned long long m = __ballot(pass);
      if (m) {
        unsigned p0 = 0;
        if (lane == 0) p0 = (unsigned)lds_add_rtn(reinterpret_cast<int*>(&s_np), __popcll(m));   // (as written: see lds_add_rtn)
        p0 = __builtin_amdgcn_readfirstlane(p0);
        if (pass) s_pass[p0 + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = c;
      }
    }
    __syncthreads();
    const unsigned np = s_np;
    // the round's output slots are reserved by one atomic per wave issued BEFORE the responses are computed (every wave knows its
    // share of the list: entries wave*64 + 256 m), so its round trip (about 2 us) runs under their loads and arithmetic
    unsigned mine = 0;                                                                 // this wave's entries of the round
    for (unsigned q0 = (unsigned)(tid & ~63); q0 < np; q0 += 256) mine += min(64u, np - q0);
    unsigned pos0 = 0;
    if (lane == 0 && mine) pos0 = atomicAdd(&sel_count[il], mine);
    for (unsigned q0 = 0; q0 < np; q0 += 256) {                                        // block-uniform bound
      const unsigned q = q0 + tid;
      if (q < np) {
        const unsigned c = s_pass[q];
        const int x = (int)(c & 0xfffu), y = (int)((c >> 12) & 0xfffu);
        const float r = harris_response(src, pitch, x, y);
        const unsigned long long key = ((unsigned long long)(~orderable(r)) << 32) | ((unsigned)y << 16) | (unsigned)x;
        // wave's slots: its earlier passes hold 64 entries each (only the last one can be partial)
        out[__builtin_amdgcn_readfirstlane(pos0) + (q0 >> 8) * 64u + (unsigned)lane] = key;
      }
    }
    __syncthreads();
  }
}

// (Round 5: ONE 1024-thread workgroup per image — all levels' candidates in one flat index space, survivors in one LDS list with slots from the
// histograms' suffix sums, dense responses — was built, bit-exact, and measured 0.152 against 0.138 ms per 256 pairs:
// profiles/r05_harris_per_image_negative.txt.  Withdrawn.)
// Canonical order of the survivors of one (image, level) + retainBest(n_l) with ties.  Up to 2048 keys: bitonic
// sort in LDS (rank_select_kernel); more (heavy ties): rank sort — keys are unique (x,y differ), so rank = number
// of smaller keys is a permutation.  E owned keys per thread per pass.
constexpr int RK_NT = 256;   // threads of rank_select_kernel: one block per (image, level) — most levels hold a few hundred keys
template <int E>
__device__ __forceinline__ void rank_pass(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                          unsigned M, unsigned g0, unsigned long long* chunk, int quota, unsigned* s_thr) {
  constexpr int CH = 2048;
  const int tid = threadIdx.x;
  unsigned long long my[E];
  unsigned rank[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const unsigned idx = g0 + e * RK_NT + tid;
    my[e] = idx < M ? in[idx] : ~0ull;
    rank[e] = 0;
  }
  for (unsigned c0 = 0; c0 < M; c0 += CH) {
    const unsigned n = min((unsigned)CH, M - c0);
    __syncthreads();
    for (unsigned i = tid; i < n; i += RK_NT) chunk[i] = in[c0 + i];
    __syncthreads();
    for (unsigned k = 0; k < n; ++k) {
      const unsigned long long v = chunk[k];
#pragma unroll
      for (int e = 0; e < E; ++e) rank[e] += (v < my[e]) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const unsigned idx = g0 + e * RK_NT + tid;
    if (idx < M) {
      out[rank[e]] = my[e];
      if ((int)rank[e] == quota - 1) *s_thr = (unsigned)(my[e] >> 32);
    }
  }
}

// Besides the canonical list (sel2, the order of the output slots) the kernel leaves the kept keypoints of the level a
// second time in `sel` — which it has consumed by then — in SPATIAL order: by 128-pixel column band, then by row, each
// entry x | y << 16 | slot-in-level << 32.  describe_kernel walks that list, so the 16 keypoints of a block are
// neighbours in one band and share the cache lines of their patch rows (results go to the keypoint's slot: the order
// of processing is free).
// exclusive prefix sum over skey[0..1024) in place (all RK_NT threads; skey[1024..1024+RK_NT/64) is scratch): 1024 / RK_NT counters per
// thread, scan inside each wave, totals of the waves before
__device__ __forceinline__ void rank_prefix_1024(unsigned* skey, int tid) {
  constexpr int CE = 1024 / RK_NT;
  unsigned cv[CE], tot = 0;
#pragma unroll
  for (int q = 0; q < CE; ++q) { cv[q] = skey[CE * tid + q]; tot += cv[q]; }
  unsigned inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned t = __shfl_up(inc, off);
    if ((tid & 63) >= off) inc += t;
  }
  if ((tid & 63) == 63) skey[1024 + (tid >> 6)] = inc;
  __syncthreads();
  unsigned run = inc - tot;
  for (int w = 0; w < (tid >> 6); ++w) run += skey[1024 + w];
#pragma unroll
  for (int q = 0; q < CE; ++q) { skey[CE * tid + q] = run; run += cv[q]; }
}

// With describe tiles (g.dt_total > 0) the spatial order is by describe tile — cell = tile index inside the level — and the kernel also
// leaves each tile's (begin, end) range of that list in tile_rng: describe_tile_kernel's blocks own one tile each.
__global__ __launch_bounds__(RK_NT) void rank_select_kernel(OrbGeom g, int n_img, XcdMap xm, unsigned long long* __restrict__ sel,
                                                           const unsigned* __restrict__ sel_count,
                                                           unsigned long long* __restrict__ sel2,
                                                           unsigned* __restrict__ kept, uint2* __restrict__ tile_rng) {
  // (Round 5: the Harris responses inside this kernel — survivors compacted into the key array, responses computed densely, sorted in place —
  // took 0.215-0.218 ms against 0.139 + 0.074 for the two launches: profiles/r05_select_fused_negative.txt.  Withdrawn.)
  __shared__ unsigned long long chunk[2048];
  __shared__ unsigned skey[1024 + 16];
  __shared__ unsigned s_thr, s_keep;
  int img, l;
  if (!xcd_decode(xm, n_img, img, l)) return;
  const int il = img * g.n_levels + l;
  const int tid = threadIdx.x;
  const unsigned M = sel_count[il];
  const int quota = g.lv[l].quota;
  unsigned long long* in = sel + (size_t)img * g.cand_total + g.lv[l].cand_off;
  unsigned long long* out = sel2 + (size_t)img * g.cand_total + g.lv[l].cand_off;
  const bool tiled = g.dt_total > 0;
  const unsigned ntl = tiled ? (unsigned)(g.lv[l].dt_nx * g.lv[l].dt_ny) : 0u;     // at most 1024 (orb_prepare_geometry)
  uint2* rng = tile_rng + (size_t)img * (size_t)g.dt_total + g.lv[l].dt_start;
  const unsigned dt_nx = (unsigned)g.lv[l].dt_nx, dt_mx = g.lv[l].dt_mx, dt_my = g.lv[l].dt_my;
  auto tile_of = [&](unsigned xy) { return __umulhi((xy >> 16) - (unsigned)EDGE, dt_my) * dt_nx + __umulhi((xy & 0xffffu) - (unsigned)EDGE, dt_mx); };
  if (tid == 0) { s_thr = 0xffffffffu; s_keep = 0; }
  __syncthreads();
  if (M == 0 || quota == 0) {
    if (tid == 0) kept[il] = 0;
    for (unsigned t = tid; t < ntl; t += RK_NT) rng[t] = uint2{0u, 0u};
    return;
  }
  if (M <= 2048) {
    // the usual case (M ~ 2 n_l): bitonic sort of the keys in LDS, padded with ~0 to a power of two.  A thread owns the
    // compare-exchanges c = tid, tid + RK_NT, ... of a step; steps with stride <= 64 stay inside the 128 keys of one
    // (wave, c / RK_NT) pair, so only the strides >= 128 need a block barrier (10 of the 55 steps at 1024 keys).
    unsigned P = 128;
    while (P < M) P <<= 1;
    for (unsigned i = tid; i < P; i += RK_NT) chunk[i] = i < M ? in[i] : ~0ull;
    __syncthreads();
    unsigned pj = 128;
    for (unsigned k = 2; k <= P; k <<= 1) {
      for (unsigned j = k >> 1; j > 0; j >>= 1) {
        if (j >= 128 || pj >= 128) __syncthreads();
        else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // same wave: LDS is in order, keep the compiler honest
        pj = j;
        for (unsigned c = tid; c < P / 2; c += RK_NT) {
          const unsigned i = ((c & ~(j - 1)) << 1) | (c & (j - 1));
          const unsigned ixj = i | j;
          const unsigned long long x = chunk[i], y = chunk[ixj];
          if ((x > y) == ((i & k) == 0)) { chunk[i] = y; chunk[ixj] = x; }
        }
      }
    }
    __syncthreads();
    for (unsigned i = tid; i < M; i += RK_NT) out[i] = chunk[i];
    if (M <= (unsigned)quota) {
      if (tid == 0) { kept[il] = M; s_keep = M; }
    } else {
      const unsigned thr = (unsigned)(chunk[quota - 1] >> 32);   // ~orderable(response of the n-th best); ties with it are kept
      for (unsigned i = tid; i < M; i += RK_NT)
        if ((unsigned)(chunk[i] >> 32) <= thr && (i + 1 >= M || (unsigned)(chunk[i + 1] >> 32) > thr)) { kept[il] = i + 1; s_keep = i + 1; }
    }
    __syncthreads();
    // spatial order of the kept keypoints: counting sort by cell = (128-px column band, block of rows) — a walk down
    // each band; the order inside a cell is whatever the atomics give (processing order only, results are unaffected)
    const unsigned K = s_keep;
    int rsh = 4;                                                   // 16-row blocks, coarser if the level is huge
    while ((unsigned)(((g.lv[l].w + 127) >> 7) * ((g.lv[l].h >> rsh) + 1)) > 1024u) ++rsh;
    const unsigned nrb = (unsigned)(g.lv[l].h >> rsh) + 1u;
    for (int i = tid; i < 1024; i += RK_NT) skey[i] = 0u;          // histogram in skey[0..1024), wave totals behind it
    __syncthreads();
    constexpr int KE = 2048 / RK_NT;
    unsigned mycell[KE], mypos[KE];
#pragma unroll
    for (int e = 0; e < KE; ++e) {
      const unsigned i = tid + (unsigned)RK_NT * e;
      if (i < K) {
        const unsigned xy = (unsigned)chunk[i];
        mycell[e] = tiled ? tile_of(xy) : ((xy & 0xffffu) >> 7) * nrb + ((xy >> 16) >> rsh);
        mypos[e] = atomicAdd(&skey[mycell[e]], 1u);
      }
    }
    __syncthreads();
    rank_prefix_1024(skey, tid);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < KE; ++e) {
      const unsigned i = tid + (unsigned)RK_NT * e;
      if (i < K) in[skey[mycell[e]] + mypos[e]] = (chunk[i] & 0xffffffffull) | ((unsigned long long)i << 32);
    }
    for (unsigned t = tid; t < ntl; t += RK_NT) rng[t] = uint2{skey[t], t + 1 < ntl ? skey[t + 1] : K};
    return;
  }
  for (unsigned g0 = 0; g0 < M; g0 += 4 * RK_NT) rank_pass<4>(in, out, M, g0, chunk, quota, &s_thr);
  __syncthreads();
  unsigned K;
  if (M <= (unsigned)quota) {
    if (tid == 0) kept[il] = M;
    K = M;
  } else {
    const unsigned thr = s_thr;   // ~orderable(response of the n-th best); keep hi <= thr
    unsigned cnt = 0;
    for (unsigned i = tid; i < M; i += RK_NT) cnt += ((unsigned)(in[i] >> 32) <= thr) ? 1u : 0u;
    atomicAdd(&s_keep, cnt);
    __syncthreads();
    if (tid == 0) kept[il] = s_keep;
    K = min(s_keep, M);
  }
  __threadfence();
  __syncthreads();
  if (!tiled) {
    // (more than 2048 candidates on one level: no spatial sort, the list is the canonical order)
    for (unsigned i = tid; i < K; i += RK_NT) in[i] = (out[i] & 0xffffffffull) | ((unsigned long long)i << 32);
    return;
  }
  // the same list by describe tile: counting sort of the K kept entries (canonical order in out[]) with the counters in LDS
  unsigned* cursor = reinterpret_cast<unsigned*>(chunk);
  for (int i = tid; i < 1024; i += RK_NT) { skey[i] = 0u; cursor[i] = 0u; }
  __syncthreads();
  for (unsigned i = tid; i < K; i += RK_NT) atomicAdd(&skey[tile_of((unsigned)out[i])], 1u);
  __syncthreads();
  rank_prefix_1024(skey, tid);
  __syncthreads();
  for (unsigned i = tid; i < K; i += RK_NT) {
    const unsigned long long v = out[i];
    const unsigned c = tile_of((unsigned)v);
    in[skey[c] + atomicAdd(&cursor[c], 1u)] = (v & 0xffffffffull) | ((unsigned long long)i << 32);
  }
  for (unsigned t = tid; t < ntl; t += RK_NT) rng[t] = uint2{skey[t], t + 1 < ntl ? skey[t + 1] : K};
}

// ---- A.7 orientation + A.8 descriptor: one wave per keypoint ---------------------------------------------
// cv::fastAtan2, scalar form, f32, one IEEE op at a time
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// cos/sin of the keypoint angle: radians in f32, then a deterministic f64 evaluation (quadrant
// reduction with a two-part pi/2, Taylor polynomials to r^17 / r^16), rounded to f32.
__device__ __forceinline__ void sincos_deg(float angle_deg, float& c_out, float& s_out) {
  const float rad_f = __fmul_rn(angle_deg, (float)(3.14159265358979323846 / 180.f));
  const double t = (double)rad_f;
  const int k = (int)__dadd_rn(__dmul_rn(t, 0.63661977236758134308), 0.5);
  const double kd = (double)k;
  const double r = __dsub_rn(__dsub_rn(t, __dmul_rn(kd, 1.57079632673412561417e+00)), __dmul_rn(kd, 6.07710050650619224932e-11));
  const double z = __dmul_rn(r, r);
  double S = 1.0 / 355687428096000.0;
  S = __dsub_rn(__dmul_rn(S, z), 1.0 / 1307674368000.0);
  S = __dadd_rn(__dmul_rn(S, z), 1.0 / 6227020800.0);
  S = __dsub_rn(__dmul_rn(S, z), 1.0 / 39916800.0);
  S = __dadd_rn(__dmul_rn(S, z), 1.0 / 362880.0);
  S = __dsub_rn(__dmul_rn(S, z), 1.0 / 5040.0);
  S = __dadd_rn(__dmul_rn(S, z), 1.0 / 120.0);
  S = __dsub_rn(__dmul_rn(S, z), 1.0 / 6.0);
  const double sr = __dadd_rn(r, __dmul_rn(r, __dmul_rn(z, S)));
  double C = 1.0 / 20922789888000.0;
  C = __dsub_rn(__dmul_rn(C, z), 1.0 / 87178291200.0);
  C = __dadd_rn(__dmul_rn(C, z), 1.0 / 479001600.0);
  C = __dsub_rn(__dmul_rn(C, z), 1.0 / 3628800.0);
  C = __dadd_rn(__dmul_rn(C, z), 1.0 / 40320.0);
  C = __dsub_rn(__dmul_rn(C, z), 1.0 / 720.0);
  C = __dadd_rn(__dmul_rn(C, z), 1.0 / 24.0);
  C = __dsub_rn(__dmul_rn(C, z), 0.5);
  const double cr = __dadd_rn(1.0, __dmul_rn(z, C));
  double cs, sn;
  switch (k & 3) {
    case 0: cs = cr; sn = sr; break;
    case 1: cs = -sr; sn = cr; break;
    case 2: cs = -cr; sn = -sr; break;
    default: cs = sr; sn = -cr; break;
  }
  c_out = (float)cs;
  s_out = (float)sn;
}

#ifndef ORBX_PB_PITCH
#define ORBX_PB_PITCH 10
#endif
constexpr int PB_ROWS = 37, PB_PITCH = ORBX_PB_PITCH;   // blurred 37x37 patch: 10 dwords per row (26.7 KB per block with the tables: 6 blocks per CU)

// Intensity-centroid weights (Appendix A.7): the 31x31 patch as 31 rows x 8 dwords (task t = r*8 + c,
// pixel column 4c+b); per task one dword of 0/1 disc-membership bytes and one of (column index)*membership
// bytes, so that v_dot4_u32_u8 yields sum(I) and sum((u+15) I) of four pixels at once.
__constant__ unsigned c_ic_ones[256];
__constant__ unsigned c_ic_col[256];

// What binds this kernel (PMC, 128-pair batch): the texture addresser — TA busy 86 %, VALU 50 %, LDS 28 % — i.e. the
// scattered row segments of the two patches, not arithmetic or latency.  Measured and dropped: all pixel loads hoisted
// in front of their first use (7 -> 2 dependent round trips: +-0), 6 waves/SIMD by a VGPR cap (slower), 16-byte lane
// loads (unaligned wide accesses split: 0.31 -> 0.36 ms), the test pattern as floats in LDS (ds_read_b128: slower),
// 8-byte loads at aligned addresses with the misalignment undone in the arithmetic (24 loads instead of 29, bit-exact,
// 0.31 -> 0.32 ms; per-lane task constants must then be kept from being hoisted out of the keypoint loop, ~50 VGPRs).
// Round 2: not requesting the tenth of the patch / disc bytes that can never be read (rows |dy| >= 13 need fewer 8-byte columns)
// — as predicated loads 0.485 -> 0.62 ms (the branches break the load schedule), as loads redirected to a needed neighbour address 0.54.
// Four keypoints per wave, 16 lanes each: the per-keypoint work that every lane would otherwise repeat (slot and
// key decode, the three centroid reductions, fastAtan2, the f64 sin/cos) is shared by 4 keypoints per instruction.
// Lane li of a group owns centroid tasks t = it*16 + li and descriptor bits r*16 + li (it, r = 0..15); a ballot
// delivers 16 bits of each of the four descriptors at once.
constexpr int DG_PER_WAVE = 4, DG_PER_BLOCK = 16;   // 16 lanes per keypoint
typedef float desc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int row16_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x124 /* row_ror:4 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x122 /* row_ror:2 */, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x121 /* row_ror:1 */, 0xf, 0xf, false);
  return v;
}

// (round 2 re-check of the occupancy: 4 / 5 / 6 waves per SIMD — 5 is what the 95 VGPRs give — 0.546 / 0.490 / 0.540 ms per 256 pairs)
#ifdef ORBX_DESC_WAVES
__attribute__((amdgpu_waves_per_eu(ORBX_DESC_WAVES, ORBX_DESC_WAVES)))
#endif
__global__ __launch_bounds__(256) void describe_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, int blocks_per_img,
                                                       const unsigned long long* __restrict__ sel2,
                                                       const unsigned long long* __restrict__ spatial,
                                                       const unsigned* __restrict__ kept,
                                                       orbx_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                       int* __restrict__ nkp, int cap_kp, float patch_size,
                                                       unsigned* __restrict__ status) {
  __shared__ unsigned pb[DG_PER_BLOCK][PB_ROWS * PB_PITCH];
  int img, bx;
  if (!xcd_decode(xm, n_img, img, bx)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15;
  // block-wide tables in LDS (registers are better spent on occupancy): 256 rBRIEF test pairs, 256 centroid tasks
  __shared__ int s_pat[256];
  __shared__ __attribute__((aligned(8))) unsigned s_ones[256], s_col[256];
  s_pat[tid] = reinterpret_cast<const int*>(c_pattern)[tid];
  s_ones[tid] = c_ic_ones[tid];
  s_col[tid] = c_ic_col[tid];
  __syncthreads();
  unsigned start[ORBX_MAX_LEVELS + 1];
  start[0] = 0;
#pragma unroll
  for (int l = 0; l < ORBX_MAX_LEVELS; ++l)
    start[l + 1] = start[l] + (l < g.n_levels ? kept[img * g.n_levels + l] : 0u);
  const unsigned total = start[ORBX_MAX_LEVELS];
  const unsigned limit = min(total, (unsigned)cap_kp);
  if (bx == 0 && tid == 0) {
    nkp[img] = (int)limit;
    if (total > (unsigned)cap_kp) atomicOr(status, ORBX_ST_KP_OVERFLOW);
  }
  unsigned* myp = pb[wave * DG_PER_WAVE + grp];
  const uint8_t* pbb = reinterpret_cast<const uint8_t*>(myp);
  for (unsigned base = (bx * 4 + wave) * DG_PER_WAVE; base < total; base += blocks_per_img * DG_PER_BLOCK) {
    // position `pos` of the image's spatially ordered walk (levels one after the other) -> level, entry of the level's
    // spatial list -> the keypoint and its output slot
    const unsigned pos_raw = base + grp;
    const unsigned pos = pos_raw < total ? pos_raw : base;  // idle groups shadow the wave's first keypoint
    int l = 0;
    unsigned lbase = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; ++i) if (pos >= start[i]) { l = i; lbase = start[i]; }
    // (levels beyond n_levels have start == total > pos, so l < n_levels)
    const size_t lofs = (size_t)img * g.cand_total + g.lv[l].cand_off;
    const unsigned long long ent = spatial[lofs + (pos - lbase)];
    const int kx = (int)(ent & 0xffffu), ky = (int)((ent >> 16) & 0xffffu);
    const unsigned j2 = (unsigned)(ent >> 32) & 0xffffu;
    const unsigned slot = lbase + j2;
    const bool active = pos_raw < total && slot < limit;
    const float resp = from_orderable(~(unsigned)(sel2[lofs + j2] >> 32));   // only needed for the output record
    int pitch;
    const uint8_t* src = level_ptr(s, g, img, l, pitch);
    const uint8_t* blr = s.blur + (size_t)img * g.slot_bytes + g.lv[l].off;
    const int bpitch = g.lv[l].pitch;
    // stage the blurred 37x37 patch (rows of 40 bytes starting at kx-18) into this group's LDS patch: 5 lanes x 8 bytes
    // per row, 3 rows per step (no division, half the loads of a dword-per-lane loop)
    {
      const int sub = li / 5, c2 = li - 5 * sub;           // li = 15 idles
      // (row pointers by 32-bit steps: a 64-bit multiply-add per load is a quarter-rate instruction)
      const uint8_t* b0 = blr + (unsigned)(__umul24((unsigned)(ky - 18 + sub), (unsigned)bpitch) + (unsigned)(kx - 18 + 8 * c2));
      const unsigned bstep = 3u * (unsigned)bpitch;
#pragma unroll
      for (int it = 0; it < 13; ++it) {
        const int r = 3 * it + sub;
        if (sub < 3 && r < PB_ROWS) {
          unsigned long long v;
          __builtin_memcpy(&v, b0 + (unsigned)it * bstep, 8);
          myp[r * PB_PITCH + 2 * c2] = (unsigned)v;
          myp[r * PB_PITCH + 2 * c2 + 1] = (unsigned)(v >> 32);
        }
      }
    }
    // intensity centroid over the 749-pixel disc straight from the level image (integer, order independent)
    int sA = 0, sB = 0, sC = 0;
    {
      // lane li: rows 4 it + (li >> 2), dwords 2 (li & 3) and 2 (li & 3) + 1 of the row as ONE 8-byte load (row 31 does not exist:
      // zero weights, re-reads row 30).  Builds without the centroid loads / without the patch loads run 0.38 / 0.35 ms against 0.505:
      // the addresser's time follows the bytes, but not quite — an 8-byte instruction costs 1.5x a 4-byte one, so 8 rounds of 8-byte
      // loads instead of 16 of dwords: 0.502 -> 0.485 ms per 256 pairs
      const uint8_t* a0 = src + (unsigned)(__umul24((unsigned)(ky - 15 + (li >> 2)), (unsigned)pitch) + (unsigned)(kx - 15 + 8 * (li & 3)));
      const unsigned astep = 4u * (unsigned)pitch;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = 4 * it + (li >> 2);
        const int r = min(row, 30);
        unsigned long long px;
        __builtin_memcpy(&px, a0 + ((unsigned)it * astep - (row > 30 ? (unsigned)pitch : 0u)), 8);
        const int t0 = row * 8 + 2 * (li & 3);
        const uint2 w1 = *reinterpret_cast<const uint2*>(&s_ones[t0]);
        const uint2 wc = *reinterpret_cast<const uint2*>(&s_col[t0]);
        const unsigned sI = __builtin_amdgcn_udot4((unsigned)(px >> 32), w1.y, __builtin_amdgcn_udot4((unsigned)px, w1.x, 0u, false), false);
        sA += (int)__builtin_amdgcn_udot4((unsigned)(px >> 32), wc.y, __builtin_amdgcn_udot4((unsigned)px, wc.x, 0u, false), false);
        sB += (int)sI;
        sC += (r - 15) * (int)sI;
      }
    }
    // sums over the 16 lanes of the group = one DPP row: four rotate-and-add steps, no LDS crossbar
    sA = row16_sum(sA); sB = row16_sum(sB); sC = row16_sum(sC);
    const int m10 = sA - 15 * sB, m01 = sC;
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    float ca, sa;
    sincos_deg(angle, ca, sa);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);   // staging stores visible to the whole wave before the reads
    // 16 test rounds, 16 lanes x 4 keypoints each: one ballot per round holds 16 descriptor bits of each of the four
    // keypoints.  Fully unrolled so that the word / shift a round's bits go to are compile-time constants (with a
    // partial unroll the four 64-bit words were updated through selects: ~16 extra VALU instructions per round).
    unsigned long long word[4] = {0ull, 0ull, 0ull, 0ull};
    const desc_f2 ca2 = {ca, ca}, sa2 = {sa, sa}, magic2 = {12582912.f, 12582912.f};
    // byte (row + 18) * pitch + col + 18 of the patch from the raw float bits: the 24-bit multiply sees 0x400000 + row, the
    // column term carries the whole 0x4B400000 + col
    constexpr unsigned kBias = 0x400000u * (PB_PITCH * 4) + 0x4B400000u - (18u * (PB_PITCH * 4) + 18u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // rotated test points, both ends of the pair per packed-f32 instruction (each product and sum rounded on its own,
      // as the reference's scalar code); rint by the 1.5*2^23 trick: the low 24 bits of (x + magic) are 0x400000 + rint(x)
      const int pr = s_pat[r * 16 + li];
      const desc_f2 X = {(float)(signed char)(pr & 0xff), (float)(signed char)((pr >> 16) & 0xff)};
      const desc_f2 Y = {(float)(signed char)((pr >> 8) & 0xff), (float)(signed char)((pr >> 24) & 0xff)};
      const desc_f2 fx = X * ca2 - Y * sa2 + magic2;
      const desc_f2 fy = X * sa2 + Y * ca2 + magic2;
      const unsigned a0 = __umul24(__float_as_uint(fy[0]), PB_PITCH * 4) + __float_as_uint(fx[0]) - kBias;
      const unsigned a1 = __umul24(__float_as_uint(fy[1]), PB_PITCH * 4) + __float_as_uint(fx[1]) - kBias;
      const int t0 = pbb[a0];
      const int t1 = pbb[a1];
      const unsigned long long bal = __ballot(t0 < t1);
      const unsigned chunk = (unsigned)(bal >> (16 * grp)) & 0xffffu;        // this keypoint's bits 16r .. 16r+15
      word[r >> 2] |= (unsigned long long)chunk << (16 * (r & 3));
    }
    if (active && li < 4) {
      const unsigned long long wv = li == 0 ? word[0] : li == 1 ? word[1] : li == 2 ? word[2] : word[3];
      reinterpret_cast<unsigned long long*>(desc_out + ((size_t)img * cap_kp + slot) * 32)[li] = wv;
    }
    if (active && li == 4) {
      const float sc = g.lv[l].scale;
      orbx_keypoint o;
      o.x = __fmul_rn((float)kx, sc);
      o.y = __fmul_rn((float)ky, sc);
      o.size = __fmul_rn(patch_size, sc);
      o.angle = angle;
      o.response = resp;
      o.octave = l;
      o.class_id = -1;
      kp_out[(size_t)img * cap_kp + slot] = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- A.7 + A.8 with the blur inside (round 4): no blurred pyramid ----------------------------------------------------------------
// The blurred levels exist for ONE reader, this kernel, which takes a 37 x 37 patch per keypoint out of them — so the whole-level blur
// (0.31 ms per 256 pairs, 1.39 GB of HBM traffic per launch) and the second image per keypoint that the texture addresser had to walk
// (31 + 37 row segments) are replaced by ONE 43-row window of the unblurred level per keypoint, staged in LDS, from which the kernel
// takes the intensity centroid and computes the patch's blur itself.  The blur is exact integer arithmetic whichever way it is summed —
// the written CPU specification of A.8 (orb_ref.cpp:119-133 of the test checker): an 8.8 horizontal pass that fits 16 bits without rounding, a 16.16 vertical pass, ONE rounding
// (v + 32768) >> 16 — i.e. blur(y, x) = (sum_ij g_i g_j p(y + i - 3, x + j - 3) + 32768) >> 16, g = {18, 34, 48, 56, 48, 34, 18}.  On the
// VALU the two passes cost ~9 instructions per patch pixel and the patches of an image overlap 3.3-fold: 0.44 ms of issue slots, more
// than the kernel it would replace.  On the matrix pipe they are two banded products on v_mfma_i32_16x16x64_i8 per keypoint:
//   H' = (W - 128) T1      W: 48 x 64 window bytes (row pitch 48: the k-slots 48..63 of a row are the next row's bytes, under zero taps),
//                          T1[c][x] = g[c - x - 2]: 9 MFMAs (3 row blocks x 3 column blocks);  H' = H - 32768 fits 16 bits
//   V' = T2 H'             H' split into a signed high byte and a low byte (xor 0x80 -> signed): two i8 products per tile, 18 MFMAs,
//                          V = 256 S_hi + S_lo + 2^15 + 2^23,  byte = (V + 32768) >> 16
// with NO data movement between the two: the accumulator of the first product has its column on the lane and four consecutive rows in its
// registers, and the k index of the second product is a summation index — the k-slot (lane group q, byte j) of the second product is DEFINED
// as row 16 (j >> 2) + 4 q + (j & 3) of H', which is what lane group q holds, and the tap operand T2 is laid out to match (a table built by
// the host: orb_prepare_geometry).  Only the A/B pairing of equal (lane group, byte) slots is assumed of the instruction's k layout.
// The result goes back to LDS column-major (a lane's register = four vertically adjacent bytes of one column = one dword), over the
// keypoint's own window, which is dead once its three A operands are in registers; the test points read byte (x + 18) * 48 + (y + 18).
// Per keypoint 27 MFMAs (16 cycles each) beside ~110 VALU instructions of glue, against ~1400 VALU instructions for the two passes.
constexpr int DF_WP = 48;                         // window / patch pitch in bytes; the window is 48 rows (43 staged), the patch 48 columns (37 used)
constexpr int DF_WIN_BYTES = 48 * DF_WP;          // 2304 B per keypoint
constexpr int DF_X0 = 23, DF_Y0 = 21;             // window origin = keypoint - (23, 21): the centroid's 31-px rows start at window byte 8 (aligned 8-byte LDS reads)
typedef int df_i4 __attribute__((ext_vector_type(4)));
__constant__ __attribute__((aligned(16))) unsigned c_blur_band[6 * 64 * 4];   // [operand set: T1 for column block 0..2, T2 for row block 0..2][lane][4 dwords]
#ifndef ORBX_DF_ITERS
#define ORBX_DF_ITERS 2        // keypoint groups per wave: the block prologue (tables, level starts) is paid once per ORBX_DF_ITERS x 16 keypoints (1 / 2: 0.705 / 0.699 ms)
#endif
// Measured on the way (round 4, same box, per 256 pairs; every variant bit-exact on the frozen digests):
//   whole-level blur + describe_kernel 0.299 + 0.484 = 0.783 ms;  this kernel as first written 0.785 — the compiler put the MFMA destinations in
//   AGPRs and every VALU consumer behind a v_accvgpr_read_b32 (108 of 261 vector instructions per keypoint); with -mllvm
//   -amdgpu-mfma-vgpr-form=1 (Makefile, this file only) 0.705;  the patch bytes stored one by one straight from bits 16..23 of the sums
//   (ds_write_b8_d16_hi by inline asm — as C++ byte stores the compiler merges them back into a dword with MORE VALU work — 27 v_perm_b32
//   fewer per keypoint, 36 LDS stores instead of 9): 0.853, the LDS pipe pays more than the VALU saves;  the test pattern as f16 pairs
//   (one v_cvt_f32_f16 per coordinate): no change, the byte table already converts with one SDWA instruction per coordinate.
// Round 5: the tests in describe_tile_kernel's form (32 lanes per keypoint, a lane's eight tests converted once per four keypoints, no ballot):
//   1.366 -> 1.337 ms per 512 pairs at 752x480, 0.735 -> 0.72 per 128 pairs at 1920x1080.
// The kernel is bound by VALU issue: per 4 keypoints ~1200 vector instructions + 108 MFMAs that hold the issue port for 8 cycles each.

__global__ __launch_bounds__(256) void describe_fused_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, int blocks_per_img,
                                                             const unsigned long long* __restrict__ sel2,
                                                             const unsigned long long* __restrict__ spatial,
                                                             const unsigned* __restrict__ kept,
                                                             orbx_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                             int* __restrict__ nkp, int cap_kp, float patch_size,
                                                             unsigned* __restrict__ status) {
  // 16 windows, then the tables: the last window's A operand reads 16 bytes past its end (k-slots under zero taps) — into the tables
  constexpr int kPatBytes = 1024;
  __shared__ __attribute__((aligned(16))) unsigned char s_win[DG_PER_BLOCK * DF_WIN_BYTES + kPatBytes + 2 * 1024];
  static_assert(sizeof(s_win) <= 40960, "four blocks per CU (160 KB of LDS)");
  int* s_pat = reinterpret_cast<int*>(s_win + DG_PER_BLOCK * DF_WIN_BYTES);
  unsigned* s_ones = reinterpret_cast<unsigned*>(s_win + DG_PER_BLOCK * DF_WIN_BYTES + kPatBytes);
  unsigned* s_col = s_ones + 256;
  int img, bx;
  if (!xcd_decode(xm, n_img, img, bx)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, li = lane & 15;
  s_pat[tid] = reinterpret_cast<const int*>(c_pattern)[tid];
  s_ones[tid] = c_ic_ones[tid];
  s_col[tid] = c_ic_col[tid];
  __syncthreads();
  unsigned start[ORBX_MAX_LEVELS + 1];
  start[0] = 0;
#pragma unroll
  for (int l = 0; l < ORBX_MAX_LEVELS; ++l)
    start[l + 1] = start[l] + (l < g.n_levels ? kept[img * g.n_levels + l] : 0u);
  const unsigned total = start[ORBX_MAX_LEVELS];
  const unsigned limit = min(total, (unsigned)cap_kp);
  if (bx == 0 && tid == 0) {
    nkp[img] = (int)limit;
    if (total > (unsigned)cap_kp) atomicOr(status, ORBX_ST_KP_OVERFLOW);
  }
  unsigned char* wwin = s_win + wave * DG_PER_WAVE * DF_WIN_BYTES;          // the wave's four windows
  unsigned char* mywin = wwin + grp * DF_WIN_BYTES;
  // staging tasks: 43 rows x 6 chunks of 8 bytes = 258, task t = 16 it + li -> (row t / 6, chunk t % 6); the pattern repeats every 3
  // rounds (48 tasks = 8 rows), so a lane keeps three (row, chunk) pairs and adds 8 rows per period
  unsigned st_g[3], st_l[3];
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) {
    const unsigned t0 = 16u * ph + (unsigned)li, r0 = (t0 * 171u) >> 10, c6 = t0 - 6u * r0;
    st_g[ph] = r0; st_l[ph] = r0 * DF_WP + 8u * c6;                         // (st_g: row; the byte offset follows from st_l)
  }
  const int l32 = lane & 31, hf = lane >> 5;
  for (unsigned base = (bx * 4 + wave) * DG_PER_WAVE; base < total; base += blocks_per_img * DG_PER_BLOCK) {
    const unsigned pos_raw = base + grp;
    const unsigned pos = pos_raw < total ? pos_raw : base;  // idle groups shadow the wave's first keypoint
    int l = 0;
    unsigned lbase = 0;
#pragma unroll
    for (int i = 1; i < ORBX_MAX_LEVELS; ++i) if (pos >= start[i]) { l = i; lbase = start[i]; }
    const size_t lofs = (size_t)img * g.cand_total + g.lv[l].cand_off;
    const unsigned long long ent = spatial[lofs + (pos - lbase)];
    const int kx = (int)(ent & 0xffffu), ky = (int)((ent >> 16) & 0xffffu);
    const unsigned j2 = (unsigned)(ent >> 32) & 0xffffu;
    const unsigned slot = lbase + j2;
    const bool active = pos_raw < total && slot < limit;
    const float resp = from_orderable(~(unsigned)(sel2[lofs + j2] >> 32));   // only needed for the output record
    int pitch;
    const uint8_t* src = level_ptr(s, g, img, l, pitch);
    // ---- the 43 x 48 window of the level around the keypoint (rows ky - 21 .. ky + 21, bytes kx - 23 .. kx + 24) into LDS
    {
      const uint8_t* w0 = src + (unsigned)(__umul24((unsigned)(ky - DF_Y0), (unsigned)pitch) + (unsigned)(kx - DF_X0));
      const unsigned p8 = 8u * (unsigned)pitch;
      unsigned go[3];
#pragma unroll
      for (int ph = 0; ph < 3; ++ph) go[ph] = __umul24(st_g[ph], (unsigned)pitch) + (st_l[ph] - st_g[ph] * DF_WP);
      unsigned long long v[17];
#pragma unroll
      for (int it = 0; it < 17; ++it) {
        const int ph = it % 3, k = it / 3;
        if (it < 16 || li < 2) __builtin_memcpy(&v[it], w0 + (go[ph] + (unsigned)k * p8), 8);
      }
#pragma unroll
      for (int it = 0; it < 17; ++it) {
        const int ph = it % 3, k = it / 3;
        // stored as p - 128 (xor 0x80 per byte): the signed bytes the matrix pipe multiplies; the centroid below takes signed dot products of
        // the same bytes — the disc is symmetric about its centre, so the offset drops out of both moments (sum of (u - 15) and of (r - 15)
        // over the disc are zero).  34 xors per wave and group of 4 keypoints here instead of 48 on the blur's operands
        if (it < 16 || li < 2) *reinterpret_cast<unsigned long long*>(mywin + st_l[ph] + (unsigned)k * (8u * DF_WP)) = v[it] ^ 0x8080808080808080ull;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);   // the four windows are in LDS for the whole wave
    // ---- intensity centroid over the 749-pixel disc, from the window (integer, order independent)
    int sA = 0, sB = 0, sC = 0;
    {
      const unsigned char* c0 = mywin + ((li >> 2) + 6) * DF_WP + 8 + 8 * (li & 3);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = 4 * it + (li >> 2);
        const int r = min(row, 30);
        const unsigned long long px = *reinterpret_cast<const unsigned long long*>(c0 + (it * 4 - (row > 30 ? 1 : 0)) * DF_WP);
        const int t0 = row * 8 + 2 * (li & 3);
        const uint2 w1 = *reinterpret_cast<const uint2*>(&s_ones[t0]);
        const uint2 wc = *reinterpret_cast<const uint2*>(&s_col[t0]);
        const int sI = __builtin_amdgcn_sdot4((int)(px >> 32), (int)w1.y, __builtin_amdgcn_sdot4((int)px, (int)w1.x, 0, false), false);
        sA += __builtin_amdgcn_sdot4((int)(px >> 32), (int)wc.y, __builtin_amdgcn_sdot4((int)px, (int)wc.x, 0, false), false);
        sB += sI;
        sC += (r - 15) * sI;
      }
    }
    sA = row16_sum(sA); sB = row16_sum(sB); sC = row16_sum(sC);
    const int m10 = sA - 15 * sB, m01 = sC;   // (of p - 128: equal to the moments of p, see the staging)
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    float ca, sa;
    sincos_deg(angle, ca, sa);
    // ---- the blurred patch of each of the wave's four keypoints, by the whole wave, on the matrix pipe
    {
      const int m16 = lane & 15, q = lane >> 4;
      const df_i4* band = reinterpret_cast<const df_i4*>(c_blur_band) + lane;
#pragma unroll 1
      for (int kp = 0; kp < DG_PER_WAVE; ++kp) {
        unsigned char* win = wwin + kp * DF_WIN_BYTES;
        df_i4 a1[3];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb) {
          a1[mb] = *reinterpret_cast<const df_i4*>(win + (16 * mb + m16) * DF_WP + 16 * q);   // (p - 128 as signed bytes: flipped when staged)
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): the window is in registers — its bytes may be overwritten
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const df_i4 t1 = band[64 * nb];
          df_i4 hh[3];
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) hh[mb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[mb], t1, (df_i4){0, 0, 0, 0}, 0, 0, 0);
          // split: k-slot (q, j = 4 mb + i) = row 16 mb + 4 q + i of H' = register i of hh[mb]
          df_i4 bhi, blo;
#pragma unroll
          for (int mb = 0; mb < 3; ++mb) {
            const unsigned p01 = __builtin_amdgcn_perm((unsigned)hh[mb][1], (unsigned)hh[mb][0], 0x05010400u);
            const unsigned p23 = __builtin_amdgcn_perm((unsigned)hh[mb][3], (unsigned)hh[mb][2], 0x05010400u);
            blo[mb] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
            bhi[mb] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
          }
          blo[3] = 0; bhi[3] = 0;
#pragma unroll
          for (int mb2 = 0; mb2 < 3; ++mb2) {
            const df_i4 t2 = band[64 * (3 + mb2)];
            const df_i4 shi = __builtin_amdgcn_mfma_i32_16x16x64_i8(t2, bhi, (df_i4){0, 0, 0, 0}, 0, 0, 0);
            const df_i4 slo = __builtin_amdgcn_mfma_i32_16x16x64_i8(t2, blo, (df_i4){8454144, 8454144, 8454144, 8454144}, 0, 0, 0);   // 2^16 + 2^23
            const unsigned v0 = ((unsigned)shi[0] << 8) + (unsigned)slo[0], v1 = ((unsigned)shi[1] << 8) + (unsigned)slo[1];
            const unsigned v2 = ((unsigned)shi[2] << 8) + (unsigned)slo[2], v3 = ((unsigned)shi[3] << 8) + (unsigned)slo[3];
            const unsigned u01 = __builtin_amdgcn_perm(v1, v0, 0x00000602u), u23 = __builtin_amdgcn_perm(v3, v2, 0x00000602u);
            *reinterpret_cast<unsigned*>(win + (16 * nb + m16) * DF_WP + 16 * mb2 + 4 * q) = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);   // the four patches are in LDS for the whole wave
    // the 256 tests, two keypoints per round (32 lanes each): a lane's eight tests — test 8 l + r, its own descriptor byte — converted once per
    // four keypoints, the sign of t0 - t1 shifted into the byte (describe_tile_kernel's form)
    typedef float desc_f4 __attribute__((ext_vector_type(4)));
    desc_f4 pat[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int pr = s_pat[8 * l32 + r];
      pat[r] = (desc_f4){(float)(signed char)(pr & 0xff), (float)(signed char)((pr >> 16) & 0xff), (float)(signed char)((pr >> 8) & 0xff), (float)(signed char)((pr >> 24) & 0xff)};
    }
    const desc_f2 magic2 = {12582912.f, 12582912.f};
    constexpr unsigned kBias = 0x400000u * DF_WP + 0x4B400000u - (18u * DF_WP + 18u);
    const unsigned slot_a = active ? slot : 0xffffffffu;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int kp = 2 * k2 + hf;                                  // the half-wave's keypoint of this round
      const int from = (16 * kp) << 2;
      const float cak = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(ca)));
      const float sak = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(sa)));
      const unsigned slotk = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)slot_a);
      const desc_f2 ca2 = {cak, cak}, sa2 = {sak, sak};
      const unsigned char* pbb = wwin + kp * DF_WIN_BYTES;
      const unsigned kbase = 0u - kBias;
      unsigned acc = 0;
#pragma unroll
      for (int r = 7; r >= 0; --r) {
        const desc_f2 X = {pat[r][0], pat[r][1]}, Y = {pat[r][2], pat[r][3]};
        const desc_f2 fx = X * ca2 - Y * sa2 + magic2;
        const desc_f2 fy = X * sa2 + Y * ca2 + magic2;
        unsigned a0, a1;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a0) : "v"(__float_as_uint(fx[0])), "v"((unsigned)DF_WP), "v"(__float_as_uint(fy[0])));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a1) : "v"(__float_as_uint(fx[1])), "v"((unsigned)DF_WP), "v"(__float_as_uint(fy[1])));
        const unsigned t0 = pbb[a0 + kbase];
        const unsigned t1 = pbb[a1 + kbase];
        acc = __builtin_amdgcn_alignbit(acc, t0 - t1, 31);
      }
      if (slotk != 0xffffffffu) desc_out[((size_t)img * cap_kp + slotk) * 32 + l32] = (uint8_t)acc;
    }
    if (active && li == 4) {
      const float sc = g.lv[l].scale;
      orbx_keypoint o;
      o.x = __fmul_rn((float)kx, sc);
      o.y = __fmul_rn((float)ky, sc);
      o.size = __fmul_rn(patch_size, sc);
      o.angle = angle;
      o.response = resp;
      o.octave = l;
      o.class_id = -1;
      kp_out[(size_t)img * cap_kp + slot] = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- A.7 + A.8 per TILE (round 5): the patch blur shared by the keypoints of a tile ---------------------------------------------------
// describe_fused_kernel blurs a 48 x 48 window per keypoint, and the windows of an image's 2000 keypoints cover its pyramid more than three
// times over (a build without the blur: 0.425 of the kernel's 0.680 ms per 256 pairs).  Here a workgroup owns one describe tile — a rectangle
// of at most DT_TW_MAX x DT_TH_MAX keypoint positions of one level (orb_prepare_geometry cuts each level's keypoint region into equal tiles;
// rank_select_kernel orders the kept keypoints by tile and leaves each tile's range) — and
//   1. blurs the tile's rectangle + 18 px once into LDS, column-major, with the same two banded products on v_mfma_i32_16x16x64_i8
//      as the per-keypoint form: a window is 64 rows x 64 bytes of the level read straight from global memory as the A operands
//      (row 16 mb + m, bytes 16 q .. 16 q + 15: no staging) and yields 48 x 48 blurred pixels (30 MFMAs); windows sit 48 apart, the last one
//      of a row / column pulled back inside the level (it then repeats pixels of its neighbour: same integers);
//   2. walks the tile's keypoints in chunks of 16 per wave: intensity centroids from the level (global loads: the level's bytes are not staged) four
//      keypoints per round, ONE evaluation of the angle and its sin / cos for the chunk, then the 256 tests on the LDS tile at
//      (kx - ox + dx) * DT_PITCH + (ky - oy + dy) two keypoints per round, 32 lanes each, a lane's eight tests in registers as floats.
// Same integers as the per-keypoint blur and as the whole-level specification (A.8), so the descriptors are bit-identical; tiles without
// keypoints return at once.  Levels too small for a 64 x 64 window use describe_fused_kernel (g.dt_total == 0).
#ifndef ORBX_DT_NWX
#define ORBX_DT_NWX 4
#endif
#ifndef ORBX_DT_NWY
#define ORBX_DT_NWY 4
#endif
constexpr int DT_NWX = ORBX_DT_NWX, DT_NWY = ORBX_DT_NWY;     // windows per tile at most
#ifndef ORBX_DT_PAD
#define ORBX_DT_PAD 4
#endif
// LDS tile: DT_COLS columns of DT_PITCH bytes (column-major: a lane's four result rows are one dword).  The pitch is an ODD number of dwords:
// the 16 columns of a result store then fall into 16 different banks (at 48 dwords they share 2)
constexpr int DT_PITCH = 48 * DT_NWY + ORBX_DT_PAD, DT_COLS = 48 * DT_NWX;
constexpr int DT_TILE_BYTES = DT_PITCH * DT_COLS;
// keypoint positions per tile: the blurred rectangle is 36 larger, and the first window starts up to 3 bytes / rows earlier — columns on 4-byte
// boundaries (a window's 16-byte row loads at an odd address cost 0.15 ms of this kernel's 0.70 per 256 pairs: the unaligned form goes through
// the address path lane by lane), rows so that every window row is a multiple of 4 below the clamp row
constexpr int DT_TW_MAX = 48 * DT_NWX - 39;
constexpr int DT_TH_MAX = 48 * DT_NWY - 39;
static_assert(DT_NWX >= 1 && DT_NWY >= 1 && DT_TILE_BYTES + 3072 <= 65536, "LDS per workgroup");
__constant__ __attribute__((aligned(16))) unsigned c_blur_band_t[6 * 64 * 4];   // T1 for output-column block 0..2, T2 for output-row block 0..2 (64 real k-slots)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void describe_tile_kernel(OrbSrc s, OrbGeom g, int n_img, XcdMap xm, const unsigned* __restrict__ tile_tab,
                                                            const uint2* __restrict__ tile_rng,
                                                            const unsigned long long* __restrict__ sel2,
                                                            const unsigned long long* __restrict__ spatial,
                                                            const unsigned* __restrict__ kept,
                                                            orbx_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                            int* __restrict__ nkp, int cap_kp, float patch_size,
                                                            unsigned* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) unsigned char s_tile[DT_TILE_BYTES + 3 * 1024];
  int* s_pat = reinterpret_cast<int*>(s_tile + DT_TILE_BYTES + 2048);
  unsigned* s_ones = reinterpret_cast<unsigned*>(s_tile + DT_TILE_BYTES);
  unsigned* s_col = s_ones + 256;
  int img, tile;
  if (!xcd_decode(xm, n_img, img, tile)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  int l, tx, ty;
  decode_tile(tile_tab, tile, l, tx, ty);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- the tile's rectangle of the level (block-uniform)
  int pitch;
  const uint8_t* src = level_ptr(s, g, img, l, pitch);
  const int lw = g.lv[l].w, lh = g.lv[l].h;
  const int X0 = EDGE + tx * g.lv[l].dt_tw, X1 = min(X0 + g.lv[l].dt_tw - 1, lw - EDGE - 1);
  const int Y0 = EDGE + ty * g.lv[l].dt_th, Y1 = min(Y0 + g.lv[l].dt_th - 1, lh - EDGE - 1);
  const int xlim = pitch - 64, ylim = lh - 64;                  // last admissible window origin (every byte a window reads lies inside the level's rows)
  const int ox = min(X0 - 21, xlim) & ~3;                       // origin of the first window: blurred pixel (ox + 3, oy + 3) is byte 0 of the LDS tile; xlim and 48 are multiples of 4
  int oy = min(Y0 - 21, ylim);
  oy -= (4 - ((ylim - oy) & 3)) & 3;                            // window rows stay dword-aligned in the tile when the last one is pulled back to ylim
  const int nwx = min((X1 + 18 - (ox + 3) + 1 + 47) / 48, DT_NWX), nwy = min((Y1 + 18 - (oy + 3) + 1 + 47) / 48, DT_NWY);
  // ---- the first window's loads leave before anything else is waited for (the tile's keypoint range, the list entries)
  const int m16 = lane & 15, q = lane >> 4;
  const df_i4* band = reinterpret_cast<const df_i4*>(c_blur_band_t) + lane;
  const int nwin = nwx * nwy;
  // a window's origin and its A operands (the loads of window wi + 4 travel under the products of window wi)
  const unsigned p16 = 16u * (unsigned)pitch;
  auto origin = [&](int wi, int& c, int& r) {
    const int wy = wi / nwx, wx = wi - wy * nwx;              // (scalar)
    c = min(ox + 48 * wx, xlim); r = min(oy + 48 * wy, ylim);
  };
  auto fetch = [&](int wi, df_i4 (&a)[4]) {
    int c, r;
    origin(wi, c, r);
    unsigned off = __umul24((unsigned)(r + m16), (unsigned)pitch) + (unsigned)(c + 16 * q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) { __builtin_memcpy(&a[mb], src + off, 16); off += p16; }
  };
  df_i4 anx[4];
  if (wave < nwin) fetch(wave, anx);
  unsigned lbase = 0, total = 0;
#pragma unroll
  for (int i = 0; i < ORBX_MAX_LEVELS; ++i) {
    const unsigned k = i < g.n_levels ? kept[img * g.n_levels + i] : 0u;
    lbase += i < l ? k : 0u;
    total += k;
  }
  const unsigned limit = min(total, (unsigned)cap_kp);
  if (tile == 0 && tid == 0) {
    nkp[img] = (int)limit;
    if (total > (unsigned)cap_kp) atomicOr(status, ORBX_ST_KP_OVERFLOW);
  }
  const uint2 rng = tile_rng[(size_t)img * (size_t)g.dt_total + tile];
  const unsigned n_kp = rng.y - rng.x;
  if (n_kp == 0) return;
  s_pat[tid] = reinterpret_cast<const int*>(c_pattern)[tid];
  s_ones[tid] = c_ic_ones[tid];
  s_col[tid] = c_ic_col[tid];
  const int grp = lane >> 4, li = lane & 15;
  const size_t lofs = (size_t)img * g.cand_total + g.lv[l].cand_off;
  const unsigned ngrp = (n_kp + DG_PER_WAVE - 1) / DG_PER_WAVE;
  const unsigned long long* sp0 = spatial + lofs + rng.x;
  auto entry = [&](unsigned gi) {
    const unsigned pos_raw = gi * DG_PER_WAVE + (unsigned)grp;
    return sp0[pos_raw < n_kp ? pos_raw : (gi < ngrp ? gi * DG_PER_WAVE : 0u)];   // idle groups shadow the wave's first keypoint
  };
  unsigned long long ent = entry((unsigned)wave), ent_n = entry((unsigned)wave + 4u);   // (requested here: they arrive under the blur)
  const unsigned long long ent_2 = entry((unsigned)wave + 8u), ent_3 = entry((unsigned)wave + 12u);
  // ---- 1. the blurred rectangle, one 64 x 64 window per wave and round
  {
    // (s_setprio 1 / 3 for the blur phase, 0 for the keypoint phase: 1.07 against 1.047 ms per 512 pairs — profiles/r05_describe_tile_steps.txt)
    for (int wi = wave; wi < nwin; wi += 4) {
      int c, r;
      origin(wi, c, r);
      df_i4 a1[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) a1[mb] = anx[mb] ^ (df_i4){(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};   // p - 128: the signed bytes the matrix pipe multiplies
      if (wi + 4 < nwin) fetch(wi + 4, anx);
      unsigned char* out0 = s_tile + (unsigned)((c - ox + m16) * DT_PITCH + (r - oy) + 4 * q);
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        const df_i4 t1 = band[64 * nb];
        df_i4 hh[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) hh[mb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[mb], t1, (df_i4){0, 0, 0, 0}, 0, 0, 0);
        // split H' = H - 32768 into a signed high byte and a low byte (xor 0x80 -> signed): k-slot (q, j = 4 mb + i) = row 16 mb + 4 q + i = register i of hh[mb]
        df_i4 bhi, blo;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          const unsigned p01 = __builtin_amdgcn_perm((unsigned)hh[mb][1], (unsigned)hh[mb][0], 0x05010400u);
          const unsigned p23 = __builtin_amdgcn_perm((unsigned)hh[mb][3], (unsigned)hh[mb][2], 0x05010400u);
          blo[mb] = (int)(__builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u);
          bhi[mb] = (int)__builtin_amdgcn_perm(p23, p01, 0x07060302u);
        }
#pragma unroll
        for (int mb2 = 0; mb2 < 3; ++mb2) {
          const df_i4 t2 = band[64 * (3 + mb2)];
          const df_i4 shi = __builtin_amdgcn_mfma_i32_16x16x64_i8(t2, bhi, (df_i4){0, 0, 0, 0}, 0, 0, 0);
          const df_i4 slo = __builtin_amdgcn_mfma_i32_16x16x64_i8(t2, blo, (df_i4){8454144, 8454144, 8454144, 8454144}, 0, 0, 0);   // 2^16 + 2^23
          const unsigned v0 = ((unsigned)shi[0] << 8) + (unsigned)slo[0], v1 = ((unsigned)shi[1] << 8) + (unsigned)slo[1];
          const unsigned v2 = ((unsigned)shi[2] << 8) + (unsigned)slo[2], v3 = ((unsigned)shi[3] << 8) + (unsigned)slo[3];
          const unsigned u01 = __builtin_amdgcn_perm(v1, v0, 0x00000602u), u23 = __builtin_amdgcn_perm(v3, v2, 0x00000602u);
          *reinterpret_cast<unsigned*>(out0 + (16 * nb) * DT_PITCH + 16 * mb2) = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
        }
      }
    }
  }
  __syncthreads();
  // ---- 2. the tile's keypoints, four per wave and round (16 lanes each)
  // Two keypoint groups ahead: the list entry of group gi + 8 and the centroid pixels of group gi + 4 are requested while group gi is worked on
  // (list entry -> pixels -> arithmetic is two dependent round trips to L2 per group otherwise, with four waves per SIMD to hide them).
  // Intensity centroid over the 749-pixel disc straight from the level image (integer, order independent): lane li takes rows
  // 4 it + (li >> 2), dwords 2 (li & 3) and 2 (li & 3) + 1 of the row as ONE 8-byte load (row 31 does not exist: zero weights, re-reads row 30)
  const unsigned astep = 4u * (unsigned)pitch;
  // (each 8-byte piece as ONE 12-byte load on a 4-byte boundary, shifted into place by two v_alignbyte_b32: loads at odd addresses go through
  // the address path lane by lane — 0.065 of this kernel's 0.55 ms per 256 pairs; ORBX_DT_CENTROID_UNALIGNED keeps the 8-byte form for A/B builds)
  struct alignas(4) Row12 { unsigned d[3]; };
  auto pixels = [&](unsigned long long e, unsigned long long (&px)[8]) {
    const int kx = (int)(e & 0xffffu), ky = (int)((e >> 16) & 0xffffu);
#ifdef ORBX_DT_CENTROID_UNALIGNED
    const uint8_t* a0 = src + (unsigned)(__umul24((unsigned)(ky - 15 + (li >> 2)), (unsigned)pitch) + (unsigned)(kx - 15 + 8 * (li & 3)));
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 4 * it + (li >> 2);
      __builtin_memcpy(&px[it], a0 + ((unsigned)it * astep - (row > 30 ? (unsigned)pitch : 0u)), 8);
    }
#else
    const unsigned xb = (unsigned)(kx - 15 + 8 * (li & 3)), sh = xb & 3u;     // (the last piece's 12 bytes end at kx + 20 at most: inside the row)
    // 32-bit offsets from the (uniform) level base, stepped by additions: scalar base + VGPR offset loads, no 64-bit address arithmetic
    unsigned off = __umul24((unsigned)(ky - 15 + (li >> 2)), (unsigned)pitch) + (xb & ~3u);
    Row12 rw[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      rw[it] = *reinterpret_cast<const Row12*>(src + (it == 7 && (li >> 2) == 3 ? off - (unsigned)pitch : off));   // (row 31 does not exist: re-reads row 30)
      off += astep;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it)
      px[it] = (unsigned long long)__builtin_amdgcn_alignbyte(rw[it].d[1], rw[it].d[0], sh) |
               ((unsigned long long)__builtin_amdgcn_alignbyte(rw[it].d[2], rw[it].d[1], sh) << 32);
#endif
  };
  // Chunks of 16 keypoints per wave:
  //  (a) the centroid sums, four keypoints per round (16 lanes each), every group's sums of round k kept by its lane k;
  //  (b) ONE evaluation of the angle and its f64 sin / cos for the chunk (lanes 0..3 of each group), and the keypoint records from those lanes;
  //  (c) the 256 tests, two keypoints per round (32 lanes each): a lane keeps its eight tests — test 8 l + r, the lane's own descriptor byte —
  //      in registers as floats (32 VGPRs, read from the LDS byte table and converted ONCE per chunk: not live under the centroid's loads; the
  //      same floats from a global table cost 0.97 against 0.895 ms per 512 pairs — a chunk's first round waits for them) and shifts the sign
  //      of t0 - t1 into its byte: no conversions, no pattern reads and no ballot inside the rounds.
  typedef float desc_f4 __attribute__((ext_vector_type(4)));
  auto centroid = [&](const unsigned long long (&px)[8], int& m10, int& m01) {
    int sA = 0, sB = 0, sC = 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 4 * it + (li >> 2);
      const int r = min(row, 30);
      const int t0 = row * 8 + 2 * (li & 3);
      const uint2 w1 = *reinterpret_cast<const uint2*>(&s_ones[t0]);
      const uint2 wc = *reinterpret_cast<const uint2*>(&s_col[t0]);
      const unsigned sI = __builtin_amdgcn_udot4((unsigned)(px[it] >> 32), w1.y, __builtin_amdgcn_udot4((unsigned)px[it], w1.x, 0u, false), false);
      sA += (int)__builtin_amdgcn_udot4((unsigned)(px[it] >> 32), wc.y, __builtin_amdgcn_udot4((unsigned)px[it], wc.x, 0u, false), false);
      sB += (int)sI;
      sC += (r - 15) * (int)sI;
    }
    sA = row16_sum(sA); sB = row16_sum(sB); sC = row16_sum(sC);
    m10 = sA - 15 * sB; m01 = sC;
  };
  const int l32 = lane & 31, hf = lane >> 5;
  unsigned long long e0 = ent, e1 = ent_n, e2 = ent_2, e3 = ent_3;
  for (unsigned g0 = (unsigned)wave; g0 < ngrp; g0 += 16) {
    int sel10, sel01;
    {
      unsigned long long pxa[8], pxb[8];
      int m10, m01;
      pixels(e0, pxa);
      if (g0 + 4 < ngrp) pixels(e1, pxb);
      centroid(pxa, m10, m01);
      sel10 = m10; sel01 = m01;                                // (lane 0 of the group is what matters of round 0; lanes 1..3 are overwritten or unused)
      if (g0 + 4 < ngrp) {
        if (g0 + 8 < ngrp) pixels(e2, pxa);
        centroid(pxb, m10, m01);
        if (li == 1) { sel10 = m10; sel01 = m01; }
        if (g0 + 8 < ngrp) {
          if (g0 + 12 < ngrp) pixels(e3, pxb);
          centroid(pxa, m10, m01);
          if (li == 2) { sel10 = m10; sel01 = m01; }
          if (g0 + 12 < ngrp) {
            centroid(pxb, m10, m01);
            if (li == 3) { sel10 = m10; sel01 = m01; }
          }
        }
      }
    }
    // the lane's tests (the index goes through an empty asm so that the reads and conversions stay inside the chunk loop)
    int pidx = 8 * l32;
    asm volatile("" : "+v"(pidx));
    desc_f4 pat[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int pr = s_pat[pidx + r];
      pat[r] = (desc_f4){(float)(signed char)(pr & 0xff), (float)(signed char)((pr >> 16) & 0xff), (float)(signed char)((pr >> 8) & 0xff), (float)(signed char)((pr >> 24) & 0xff)};
    }
    // (b) lane k < 4 of group grp stands for the keypoint of round k of that group
    const unsigned long long es = li == 1 ? e1 : li == 2 ? e2 : li == 3 ? e3 : e0;
    const unsigned j2_s = (unsigned)(es >> 32) & 0xffffu;
    const unsigned gi_s = g0 + 4u * (unsigned)(li & 3);
    const int kx_s = (int)(es & 0xffffu), ky_s = (int)((es >> 16) & 0xffffu);
    const unsigned slot_s = lbase + j2_s;
    const bool active_s = li < 4 && gi_s < ngrp && gi_s * DG_PER_WAVE + (unsigned)grp < n_kp && slot_s < limit;
    const float resp = from_orderable(~(unsigned)(sel2[lofs + j2_s] >> 32));   // only needed for the output record
    // the next chunk's list entries travel under the tests
    const unsigned long long n0 = entry(g0 + 16u), n1 = entry(g0 + 20u);
    const float angle16 = fast_atan2_deg((float)sel01, (float)sel10);
    float ca16, sa16;
    sincos_deg(angle16, ca16, sa16);
    // byte (kx - (ox + 3) + col) * DT_PITCH + (ky - (oy + 3) + row) of the column-major tile from the raw float bits: the 24-bit multiply sees
    // 0x400000 + col, the row term carries the whole 0x4B400000 + row
    constexpr unsigned kBias = 0x400000u * DT_PITCH + 0x4B400000u;
    const unsigned kbase16 = (unsigned)((kx_s - (ox + 3)) * DT_PITCH + (ky_s - (oy + 3))) - kBias;
    const unsigned slot16 = active_s ? slot_s : 0xffffffffu;
    if (active_s) {
      const float sc = g.lv[l].scale;
      orbx_keypoint o;
      o.x = __fmul_rn((float)kx_s, sc);
      o.y = __fmul_rn((float)ky_s, sc);
      o.size = __fmul_rn(patch_size, sc);
      o.angle = angle16;
      o.response = resp;
      o.octave = l;
      o.class_id = -1;
      kp_out[(size_t)img * cap_kp + slot_s] = o;
    }
    // (c) round k2: the half-wave hf takes the chunk's keypoint 2 k2 + hf = round k2 >> 1 of group 2 (k2 & 1) + hf
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      if (g0 + 4u * (unsigned)(k2 >> 1) >= ngrp) break;
      const int from = (16 * (2 * (k2 & 1) + hf) + (k2 >> 1)) << 2;
      const float ca = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(ca16)));
      const float sa = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(sa16)));
      const unsigned kbase = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)kbase16);
      const unsigned slot = (unsigned)__builtin_amdgcn_ds_bpermute(from, (int)slot16);
      const desc_f2 ca2 = {ca, ca}, sa2 = {sa, sa}, magic2 = {12582912.f, 12582912.f};
      unsigned acc = 0;
#pragma unroll
      for (int r = 7; r >= 0; --r) {
        const desc_f2 X = {pat[r][0], pat[r][1]}, Y = {pat[r][2], pat[r][3]};
        const desc_f2 fx = X * ca2 - Y * sa2 + magic2;
        const desc_f2 fy = X * sa2 + Y * ca2 + magic2;
        unsigned a0, a1;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a0) : "v"(__float_as_uint(fx[0])), "v"((unsigned)DT_PITCH), "v"(__float_as_uint(fy[0])));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a1) : "v"(__float_as_uint(fx[1])), "v"((unsigned)DT_PITCH), "v"(__float_as_uint(fy[1])));
        const unsigned t0 = s_tile[a0 + kbase];
        const unsigned t1 = s_tile[a1 + kbase];
        acc = __builtin_amdgcn_alignbit(acc, t0 - t1, 31);       // (acc << 1) | (t0 < t1): test 8 l + r ends as bit r
      }
      if (slot != 0xffffffffu) desc_out[((size_t)img * cap_kp + slot) * 32 + l32] = (uint8_t)acc;
    }
    e0 = n0; e1 = n1; e2 = entry(g0 + 24u); e3 = entry(g0 + 28u);
  }
}

// ---- host: geometry, tables, launch sequence ---------------------------------------------------------------

void build_resize_tab(int src, int dst, std::vector<unsigned>& out) {
  // f = (d + 0.5) * (src/dst) - 0.5 in f64; i = floor(f); c1 = round-half-even((f - i) * 256)
  const double inv_scale = (double)dst / (double)src;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; ++d) {
    const double f = scale * ((double)d + 0.5) - 0.5;
    int i = (int)std::floor(f);
    unsigned ofs, c1;
    if (i < 0) { ofs = 0; c1 = 0; }
    else if (i >= src - 1) { ofs = (unsigned)(src - 1); c1 = 0; }
    else { ofs = (unsigned)i; c1 = (unsigned)lrint((f - (double)i) * 256.0); }
    if (c1 == 256u) { ofs += 1; c1 = 0; }   // weight (0,256) on (i,i+1) == weight (256,0) on (i+1,i+2): keeps c1 in a byte
    out.push_back((ofs << 16) | c1);
  }
  while (out.size() % 4) out.push_back(out.back());   // uint4 loads of the x table
}

// Column strips of one level for blur_kernel: (first column, mode).  Full 248-px strips while they fit; the remainder
// goes to the cheapest cover by narrower strips (cost = row steps of a wave: rows per group + 6 halo rows).
std::vector<std::pair<int, int>> blur_strips(int w) {
  std::vector<std::pair<int, int>> out;
  int x = 0;
  while (w - x >= BLUR_W) { out.push_back({x, 0}); x += BLUR_W; }
  const int rem = w - x;
  if (rem <= 0) return out;
  const int wid[3] = {BLUR_W, BLUR_W / 2 - 4 * BLUR_HALO, BLUR_W / 4 - 6 * BLUR_HALO}, cost[3] = {BLUR_STRIP + 6, BLUR_STRIP / 2 + 6, BLUR_STRIP / 4 + 6};
  int best_cost = 1 << 30, best[3] = {1, 0, 0};
  for (int a = 0; a <= 1; ++a)
    for (int b = 0; b <= 2; ++b)
      for (int c = 0; c <= 4; ++c) {
        if (a * wid[0] + b * wid[1] + c * wid[2] < rem) continue;
        const int cst = a * cost[0] + b * cost[1] + c * cost[2];
        if (cst < best_cost) { best_cost = cst; best[0] = a; best[1] = b; best[2] = c; }
      }
  for (int m = 0; m < 3; ++m)
    for (int k = 0; k < best[m] && x < w; ++k) { out.push_back({x, m}); x += wid[m]; }
  return out;
}

}  // namespace

int orb_prepare_geometry(orbx_handle* h, int w, int h_px) {
  if (h->geom_w == w && h->geom_h == h_px) return ORBX_OK;
  // A new image size rewrites the resize / tile tables IN PLACE (the buffer rarely grows, so orbx_reserve does not see
  // it): work of earlier asynchronous calls may still be reading them — the blocking copy below runs on the null stream,
  // which the handle's non-blocking streams do not order against — and a captured hipGraph of orbx_process_stereo has the
  // old OrbGeom and table offsets baked into its kernel nodes.  Drain the streams and drop the graph first.
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (h->s_aux) ORBX_HIP(h, hipStreamSynchronize(h->s_aux));
  if (h->pair_graph) { hipGraphExecDestroy(h->pair_graph); h->pair_graph = nullptr; }
  h->pg_calls = 0;
  h->geom_w = 0; h->geom_h = 0;     // nothing valid until the new tables are up
  const orbx_orb_params& p = h->orb;
  OrbGeom g{};
  g.n_levels = p.n_levels;
  g.fast_threshold = p.fast_threshold;
  // level sizes and quotas: OpenCV orb.cpp (Appendix A.2/A.3)
  const double sf = (double)p.scale_factor;
  for (int l = 0; l < p.n_levels; ++l) {
    g.lv[l].scale = (float)std::pow(sf, (double)(l - p.first_level));
    g.lv[l].w = (int)lrintf((float)w / g.lv[l].scale);
    g.lv[l].h = (int)lrintf((float)h_px / g.lv[l].scale);
  }
  const float factor = (float)(1.0 / sf);
  float nper = (float)p.n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.n_levels));
  int sum = 0;
  for (int l = 0; l < p.n_levels - 1; ++l) {
    g.lv[l].quota = (int)lrintf(nper);
    sum += g.lv[l].quota;
    nper *= factor;
  }
  g.lv[p.n_levels - 1].quota = p.n_features - sum > 0 ? p.n_features - sum : 0;
  unsigned off = 0, coff = 0;
  int bt = 0, ft = 0;
  for (int l = 0; l < p.n_levels; ++l) {
    OrbLevelGeom& L = g.lv[l];
    if (L.w < 8 || L.h < 8) return orbx_fail(h, ORBX_ERR_INVALID, "image too small for %d pyramid levels", p.n_levels);
    L.pitch = (L.w + 63) & ~63;
    L.off = off;
    off += (unsigned)(((size_t)L.pitch * L.h + 255) & ~(size_t)255);
    const int iw = L.w - 2 * EDGE, ih = L.h - 2 * EDGE;   // border-filtered region
    L.cand_off = coff;
    L.cand_cap = (iw > 0 && ih > 0) ? (unsigned)(((iw + 1) / 2) * ((ih + 1) / 2)) : 0u;
    coff += (L.cand_cap + 63u) & ~63u;
    bt += (int)blur_strips(L.w).size() * ((L.h + BLUR_H - 1) / BLUR_H);
    L.ftiles_x = iw > 0 ? (iw + FT_W - 1) / FT_W : 0;
    L.ftile_start = ft;
    ft += (iw > 0 && ih > 0) ? L.ftiles_x * ((ih + FT_H - 1) / FT_H) : 0;
    if (L.ftiles_x == 0) L.ftiles_x = 1;
  }
  g.slot_bytes = off;
  g.cand_total = coff;
  g.btiles_total = bt;
  g.ftiles_total = ft;
  // describe tiles: each level's keypoint region [31, w-32] x [31, h-32] cut into equal rectangles of at most DT_TW_MAX x DT_TH_MAX positions,
  // the cut that needs the fewest 48 x 48 blur windows (a tile's blurred rectangle is 36 larger than its keypoint rectangle).  Needs every
  // level that can hold keypoints to hold a 64 x 64 window, and at most 1024 tiles per level (rank_select_kernel's counters).
  {
    static const int dt_env = [] { const char* e = getenv("ORBX_DESC_TILE"); return e ? atoi(e) : -1; }();   // 0: never, 1: whenever admissible, unset: by the rule below
    bool ok = dt_env != 0;
    int dt = 0;
    long windows = 0;
    auto cut = [](int k, int tmax, int margin, int& n_out, int& t_out) {
      int best = 1 << 30;
      const int n0 = (k + tmax - 1) / tmax;
      for (int n = n0; n <= n0 + 4; ++n) {
        const int t = (k + n - 1) / n;
        const int wins = n * ((t + margin + 47) / 48);
        if (wins < best) { best = wins; n_out = n; t_out = t; }
      }
      return best;
    };
    for (int l = 0; l < p.n_levels; ++l) {
      OrbLevelGeom& L = g.lv[l];
      const int kw = L.w - 2 * EDGE, kh = L.h - 2 * EDGE;
      L.dt_nx = L.dt_ny = 0; L.dt_tw = L.dt_th = 1; L.dt_start = dt; L.dt_mx = L.dt_my = 0;
      if (kw <= 0 || kh <= 0) continue;
      if (L.w < 64 || L.h < 64) { ok = false; continue; }
      windows += (long)cut(kw, DT_TW_MAX, 39, L.dt_nx, L.dt_tw) * cut(kh, DT_TH_MAX, 39, L.dt_ny, L.dt_th);
      if (L.dt_nx * L.dt_ny > 1024) ok = false;
      L.dt_mx = (unsigned)(0x100000000ull / (unsigned)L.dt_tw) + 1u;     // floor(n / d) = umulhi(n, floor(2^32 / d) + 1) for n, d < 2^16
      L.dt_my = (unsigned)(0x100000000ull / (unsigned)L.dt_th) + 1u;
      dt += L.dt_nx * L.dt_ny;
    }
    // The tile form blurs `windows` windows per image whatever the keypoints, the per-keypoint form one per keypoint: per 752x480 image 734 - 758
    // windows against 2000 keypoints (0.535 against 0.680 ms per 256 pairs), per 1920x1080 image ~4400 against 4000 (0.520 against 0.363 per 64
    // pairs).  By the measured costs — 0.45 ns per window, 0.35 against 0.66 ns per keypoint — the forms break even at windows = 0.7 n_features.
    if (dt_env < 0 && windows * 10 > (long)p.n_features * 7) ok = false;
    g.dt_total = ok ? dt : 0;
  }
  // resize tables
  std::vector<unsigned> tab;
  h->resize_tab_off.assign(2 * ORBX_MAX_LEVELS, 0);
  for (int l = 1; l < p.n_levels; ++l) {
    h->resize_tab_off[2 * l] = (unsigned)tab.size();
    build_resize_tab(g.lv[l - 1].w, g.lv[l].w, tab);
    h->resize_tab_off[2 * l + 1] = (unsigned)tab.size();
    build_resize_tab(g.lv[l - 1].h, g.lv[l].h, tab);
  }
  // tile tables of the blur (strips, see blur_kernel) and FAST (level, tx, ty) launches, behind the resize tables
  if (g.lv[0].w > 8191 * 4 || g.lv[0].h > 16383 * 16) return orbx_fail(h, ORBX_ERR_INVALID, "image too large for the tile tables");
  h->btile_tab_off = (unsigned)tab.size();
  for (int l = 0; l < p.n_levels; ++l) {
    const std::vector<std::pair<int, int>> strips = blur_strips(g.lv[l].w);
    for (int ty = 0; ty < (g.lv[l].h + BLUR_H - 1) / BLUR_H; ++ty)
      for (const auto& st : strips)
        tab.push_back((unsigned)l | ((unsigned)st.second << 3) | ((unsigned)(st.first / 4) << 5) | ((unsigned)ty << 18));
  }
  if ((int)(tab.size() - h->btile_tab_off) != bt) return orbx_fail(h, ORBX_ERR_INVALID, "internal: blur tile table size mismatch");
  h->ftile_tab_off = (unsigned)tab.size();
  for (int l = 0; l < p.n_levels; ++l) {
    const OrbLevelGeom& L = g.lv[l];
    const int iw = L.w - 2 * EDGE, ih = L.h - 2 * EDGE;
    const int ny = (iw > 0 && ih > 0) ? (ih + FT_H - 1) / FT_H : 0;
    for (int ty = 0; ty < ny; ++ty)
      for (int tx = 0; tx < L.ftiles_x; ++tx) tab.push_back((unsigned)l | ((unsigned)tx << 3) | ((unsigned)ty << 17));
  }
  if ((int)(tab.size() - h->ftile_tab_off) != ft) return orbx_fail(h, ORBX_ERR_INVALID, "internal: FAST tile table size mismatch");
  h->dtile_tab_off = (unsigned)tab.size();
  if (g.dt_total > 0)
    for (int l = 0; l < p.n_levels; ++l)
      for (int ty = 0; ty < g.lv[l].dt_ny; ++ty)
        for (int tx = 0; tx < g.lv[l].dt_nx; ++tx) tab.push_back((unsigned)l | ((unsigned)tx << 3) | ((unsigned)ty << 17));
  if ((int)(tab.size() - h->dtile_tab_off) != g.dt_total) return orbx_fail(h, ORBX_ERR_INVALID, "internal: describe tile table size mismatch");
  if (int rc = orbx_reserve(h, h->resize_tab, sizeof(unsigned) * (tab.size() + 1))) return rc;
  ORBX_HIP(h, hipMemcpy(h->resize_tab.p, tab.data(), sizeof(unsigned) * tab.size(), hipMemcpyHostToDevice));
  static_assert(sizeof(kPattern31) == 256 * 4 * sizeof(int), "pattern table");
  signed char pat[1024];
  for (int i = 0; i < 256; ++i)
    for (int k = 0; k < 4; ++k) pat[4 * i + k] = (signed char)kPattern31[i][k];
  ORBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), pat, sizeof(pat)));
  // intensity-centroid dot4 weights; umax (Appendix A.7) = half-widths of the 31-px disc
  static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  unsigned ones[256], colw[256];
  for (int t = 0; t < 256; ++t) {
    ones[t] = colw[t] = 0;
    const int r = t >> 3, c = t & 7;
    if (r > 30) continue;
    for (int b = 0; b < 4; ++b) {
      const int col = 4 * c + b, u = col - 15, v = r - 15;
      if (col <= 30 && std::abs(u) <= umax[std::abs(v)]) {
        ones[t] |= 1u << (8 * b);
        colw[t] |= (unsigned)col << (8 * b);
      }
    }
  }
  ORBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_ic_ones), ones, sizeof(ones)));
  ORBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_ic_col), colw, sizeof(colw)));
  // tap operands of describe_fused_kernel's two banded products (A.8 taps {18, 34, 48, 56, 48, 34, 18}), per lane (m = lane & 15, g = lane >> 4):
  //   sets 0..2  T1 for output-column block nb: byte j of the lane's 16 = tap[c - x - 2], c = 16 g + j (window byte), x = 16 nb + m
  //   sets 3..5  T2 for output-row block mb:    byte j = tap[rho - y], rho = 16 (j >> 2) + 4 g + (j & 3) (the row of H' that k-slot (g, j)
  //              stands for: what lane group g's accumulator registers hold), y = 16 mb + m; k-slots of the fourth register set (rows >= 48): 0
  {
    static const int tap[7] = {18, 34, 48, 56, 48, 34, 18};
    std::vector<unsigned> band(6 * 64 * 4, 0u);
    for (int set = 0; set < 6; ++set)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 16; ++j) {
          const int m = lane & 15, gq = lane >> 4;
          int idx;
          if (set < 3) idx = (16 * gq + j) - (16 * set + m) - 2;
          else idx = (j >> 2) < 3 ? (16 * (j >> 2) + 4 * gq + (j & 3)) - (16 * (set - 3) + m) : -1;
          const unsigned v = (idx >= 0 && idx <= 6) ? (unsigned)tap[idx] : 0u;
          band[((size_t)set * 64 + lane) * 4 + (j >> 2)] |= v << (8 * (j & 3));
        }
    ORBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_blur_band), band.data(), band.size() * sizeof(unsigned)));
    // describe_tile_kernel's operands: windows of 64 rows x 64 bytes, output (x, y) from window bytes x .. x + 6 of rows y .. y + 6, all 64 k-slots real:
    //   sets 0..2  T1: byte j = tap[c - x], c = 16 g + j, x = 16 nb + m;   sets 3..5  T2: byte j = tap[rho - y], rho = 16 (j >> 2) + 4 g + (j & 3), y = 16 mb + m
    std::vector<unsigned> band_t(6 * 64 * 4, 0u);
    for (int set = 0; set < 6; ++set)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 16; ++j) {
          const int m = lane & 15, gq = lane >> 4;
          const int idx = set < 3 ? (16 * gq + j) - (16 * set + m) : (16 * (j >> 2) + 4 * gq + (j & 3)) - (16 * (set - 3) + m);
          const unsigned v = (idx >= 0 && idx <= 6) ? (unsigned)tap[idx] : 0u;
          band_t[((size_t)set * 64 + lane) * 4 + (j >> 2)] |= v << (8 * (j & 3));
        }
    ORBX_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(c_blur_band_t), band_t.data(), band_t.size() * sizeof(unsigned)));
  }
  h->geom = g;
  h->geom_w = w;
  h->geom_h = h_px;
  return ORBX_OK;
}

int orb_extract_prepare(orbx_handle* h, int n_images, int w, int h_px) {
  if (n_images <= 0) return ORBX_OK;
  if (int rc = orb_prepare_geometry(h, w, h_px)) return rc;
  const OrbGeom& g = h->geom;
  const size_t n_il = (size_t)n_images * g.n_levels;
  if (int rc = orbx_reserve(h, h->ws_pyr, (size_t)g.slot_bytes * n_images)) return rc;
  static const bool unfused_ws = getenv("ORBX_DESC_UNFUSED") != nullptr;
  if (unfused_ws) { if (int rc = orbx_reserve(h, h->ws_blur, (size_t)g.slot_bytes * n_images)) return rc; }
  if (int rc = orbx_reserve(h, h->ws_cand, sizeof(unsigned) * (size_t)g.cand_total * n_images)) return rc;
  if (int rc = orbx_reserve(h, h->ws_sel, sizeof(unsigned long long) * (size_t)g.cand_total * n_images)) return rc;
  if (int rc = orbx_reserve(h, h->ws_sel2, sizeof(unsigned long long) * (size_t)g.cand_total * n_images)) return rc;
  if (int rc = orbx_reserve(h, h->ws_dtile, sizeof(uint2) * (size_t)std::max(g.dt_total, 1) * n_images)) return rc;
  // counters: cand_count[n_il], sel_count[n_il], kept[n_il], hist[n_il*256]
  const size_t n_cnt = n_il * (3 + 256);
  if (int rc = orbx_reserve(h, h->ws_counters, sizeof(unsigned) * n_cnt)) return rc;
  ORBX_HIP(h, hipMemsetAsync(h->ws_counters.p, 0, sizeof(unsigned) * n_cnt, h->stream));
  return ORBX_OK;
}

int orb_extract_range(orbx_handle* h, hipStream_t st, const uint8_t* d_images, int n_images, int img0, int n, int w, int h_px, size_t stride,
                      orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp, int cap_kp, hipEvent_t after_resize) {
  if (n <= 0) return ORBX_OK;
  const OrbGeom& g = h->geom;
  const int nl = g.n_levels;
  const size_t n_il = (size_t)n_images * nl;
  unsigned* cand_count = (unsigned*)h->ws_counters.p;
  unsigned* sel_count = cand_count + n_il;
  unsigned* kept = sel_count + n_il;
  unsigned* hist = kept + n_il;
  const bool aligned = ((uintptr_t)d_images % 4 == 0) && (stride % 4 == 0) && (((size_t)h_px * stride) % 4 == 0);
  const unsigned* tab = (const unsigned*)h->resize_tab.p;
  {
    OrbSrc s{};
    s.pyr = (uint8_t*)h->ws_pyr.p + (size_t)img0 * g.slot_bytes;
    s.blur = (uint8_t*)h->ws_blur.p + (size_t)img0 * g.slot_bytes;
    const uint8_t* imgs = d_images + (size_t)img0 * h_px * stride;
    if (aligned) {
      s.l0 = imgs; s.l0_img_stride = (size_t)h_px * stride; s.l0_pitch = (int)stride;
    } else {
      ProfScope ps(h, "copy_l0_kernel", st);
      hipLaunchKernelGGL(copy_l0_kernel, dim3(1, h_px, n), dim3(256), 0, st, imgs, (size_t)h_px * stride, stride, w, h_px, s.pyr,
                         g.slot_bytes, g.lv[0].pitch);
      s.l0 = s.pyr; s.l0_img_stride = g.slot_bytes; s.l0_pitch = g.lv[0].pitch;
    }
    if (img0 == 0) { h->last_src = s; h->last_n_images = n_images; }     // (orbx_debug_read_level addresses the whole call's images)
    unsigned* cc = cand_count + (size_t)img0 * nl;
    unsigned* sc = sel_count + (size_t)img0 * nl;
    unsigned* kp = kept + (size_t)img0 * nl;
    unsigned* hs = hist + (size_t)img0 * nl * 256;
    unsigned* cand = (unsigned*)h->ws_cand.p + (size_t)img0 * g.cand_total;
    unsigned long long* sel = (unsigned long long*)h->ws_sel.p + (size_t)img0 * g.cand_total;
    unsigned long long* sel2 = (unsigned long long*)h->ws_sel2.p + (size_t)img0 * g.cand_total;
    uint2* dtile = (uint2*)h->ws_dtile.p + (size_t)img0 * (size_t)g.dt_total;
    {
      ProfScope ps(h, "resize_kernel", st);
      for (int l = 1; l < nl; ++l) {
        const bool small = n < 64;                       // up to 32 pairs: latency counts, keep the blocks short and many (4 / 8 / 16 pairs: 1-2 % over the 6-row form)
        // (rows chosen per level to waste least of its last tile row, among 3..6: 0.262 against 0.258 ms with 6 everywhere)
        const int rows = small ? RESIZE_ROWS_SMALL : RESIZE_ROWS;
        const int tx = (g.lv[l].w + 63) / 64, ty = (g.lv[l].h + 16 * rows - 1) / (16 * rows);
        const int chains = (tx * ty + RESIZE_CHAIN - 1) / RESIZE_CHAIN;
        if (small)
          hipLaunchKernelGGL(resize_kernel<RESIZE_ROWS_SMALL>, xcd_grid(chains, n), dim3(256), 0, st, s, g, l, n, xcd_map(chains), tx, tx * ty,
                             tab + h->resize_tab_off[2 * l], tab + h->resize_tab_off[2 * l + 1]);
        else
          hipLaunchKernelGGL(resize_kernel<RESIZE_ROWS>, xcd_grid(chains, n), dim3(256), 0, st, s, g, l, n, xcd_map(chains), tx, tx * ty,
                             tab + h->resize_tab_off[2 * l], tab + h->resize_tab_off[2 * l + 1]);
      }
    }
    if (after_resize) ORBX_HIP(h, hipEventRecord(after_resize, st));
    // blur (latency-bound, ~50 % VALU-busy) and the FAST -> Harris -> ordering chain (issue-bound) both depend only on the
    // pyramid and meet again at describe: with ORBX_FORK_BLUR=1 in the environment the blur runs beside the chain on a
    // second stream (+2-3 % frames/s).  Off by default and never while per-kernel profiling is on: two kernels sharing the
    // chip stretch each other's duration, so per-kernel times (HIP events, rocprof) would describe the overlap instead of
    // the kernels and no longer agree between runs.
    // Round 4: describe_fused_kernel blurs each keypoint's patch itself (matrix pipe): no blurred pyramid, no blur launch.  ORBX_DESC_UNFUSED=1
    // keeps the two-kernel form (whole-level blur + describe_kernel) for A/B runs; orbx_debug_read_level(which = 1) blurs on demand.
    static const bool unfused = getenv("ORBX_DESC_UNFUSED") != nullptr;
    const bool fork = unfused && n >= 16 && !h->profiling && getenv("ORBX_FORK_BLUR") != nullptr;
    if (fork && !h->s_aux) {
      ORBX_HIP(h, hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      ORBX_HIP(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    if (fork) {
      ORBX_HIP(h, hipEventRecord(h->ev_fork, st));
      ORBX_HIP(h, hipStreamWaitEvent(h->s_aux, h->ev_fork, 0));
    }
    if (unfused) {
      ProfScope ps(h, "blur_kernel", fork ? h->s_aux : st, true);
      hipLaunchKernelGGL(blur_kernel, xcd_grid(g.btiles_total, n), dim3(256), 0, fork ? h->s_aux : st, s, g, n, xcd_map(g.btiles_total),
                         tab + h->btile_tab_off);
    }
    if (fork) ORBX_HIP(h, hipEventRecord(h->ev_join, h->s_aux));
    if (g.ftiles_total > 0) {
      ProfScope ps(h, "fast_kernel", st, true);
      // chains of FAST_CHAIN tiles per block once there are blocks to spare (each CU holds 8): for a pair one tile per block
      const int chain_len = (size_t)g.ftiles_total * n >= (size_t)4 * 8 * h->n_cu ? FAST_CHAIN : 1;
      const int chains = (g.ftiles_total + chain_len - 1) / chain_len;
      if (g.fast_threshold < 128)
        hipLaunchKernelGGL(fast_kernel<true>, xcd_grid(chains, n), dim3(FT_THREADS), 0, st, s, g, n, xcd_map(chains), tab + h->ftile_tab_off, cand, cc, hs, g.ftiles_total, chain_len, h->d_status);
      else
        hipLaunchKernelGGL(fast_kernel<false>, xcd_grid(chains, n), dim3(FT_THREADS), 0, st, s, g, n, xcd_map(chains), tab + h->ftile_tab_off, cand, cc, hs, g.ftiles_total, chain_len, h->d_status);
    }
    {
      ProfScope ps(h, "harris_select_kernel", st, true);
      // (round 3: the first round's candidates requested before the histogram, which they do not depend on — one round trip less on the
      // block's chain —: 0.152 -> 0.165 ms per 256 pairs; not kept)
      // 8 blocks per (image, level).  1 / 2 / 4 / 8 / 16 / 32 blocks: 0.210 / 0.169 / 0.153 / 0.150 / 0.204 / 0.375 ms per 256 pairs:
      // a level's 2 x quota survivors are a few hundred, so a block is one latency chain (histogram -> candidates -> 27 loads per
      // response -> store) and the kernel lives on how many of them are in flight
      // Round 5: the blocks dealt to the levels in proportion to their area (candidates follow the area: level 0 holds 12.8 x level 7's, and
      // with 8 blocks each a level-7 block held ~10 candidates for its 256 threads), at least one each.  Same box, ms per 256 pairs: 8 per
      // level 0.153; 64 / 40 / 24 / 16 blocks per image by area: 0.146 / 0.141 / 0.144 / 0.149 (profiles/r05_harris_blocks_by_area_ab.txt).
      // ORBX_HARRIS_BLOCKS=<total per image> overrides, ORBX_HARRIS_BLOCKS=0 keeps 8 per level (A/B runs).
      HarrisPlan hp{};
      // (the 40 is per 2000 features: 4000 features on 1920x1080 with 40 blocks ran 0.148 against 0.128 ms per 64 pairs with round 4's 64)
      static const int hb_env = [] { const char* e = getenv("ORBX_HARRIS_BLOCKS"); return e ? atoi(e) : -1; }();
      const int hb_auto = std::min(512, std::max(16, (40 * h->orb.n_features + 1000) / 2000));
      const int hb_total = hb_env < 0 ? hb_auto : (hb_env >= 8 && hb_env <= 512 ? hb_env : 0);
      if (hb_total) {
        double area = 0;
        for (int l = 0; l < nl; ++l) area += (double)g.lv[l].w * g.lv[l].h;
        for (int l = 0; l < nl; ++l) hp.start[l + 1] = hp.start[l] + std::max(1, (int)lrint(hb_total * ((double)g.lv[l].w * g.lv[l].h) / area));
      } else
        for (int l = 0; l < nl; ++l) hp.start[l + 1] = hp.start[l] + 8;
      for (int l = nl; l < ORBX_MAX_LEVELS; ++l) hp.start[l + 1] = hp.start[nl];
      const int hblocks = hp.start[nl];
      hipLaunchKernelGGL(harris_select_kernel, xcd_grid(hblocks, n), dim3(256), 0, st, s, g, n, xcd_map(hblocks),
                         (const unsigned*)cand, cc, hs, sel, sc, hp);
    }
    {
      ProfScope ps(h, "rank_select_kernel", st, true);
      hipLaunchKernelGGL(rank_select_kernel, xcd_grid(nl, n), dim3(RK_NT), 0, st, g, n, xcd_map(nl), sel, sc,
                         sel2, kp, dtile);
    }
    if (fork) ORBX_HIP(h, hipStreamWaitEvent(st, h->ev_join, 0));
    // Round 5: one workgroup per describe tile, the patch blur shared by the tile's keypoints (describe_tile_kernel); ORBX_DESC_TILE=0 (read when
    // the geometry is prepared) or a level too small for a 64 x 64 window keeps the per-keypoint form
    // Small calls keep the per-keypoint form: a pair is 104 tiles — 104 workgroups that each walk ~4 window rounds and ~3 keypoint rounds one after
    // the other — against 126 workgroups of 32 keypoints; one pair per call 0.319 against 0.272 ms, 8 pairs 50.9 against 52.7 k frames/s, 64 pairs
    // 133 against 126 k (profiles/r05_bench_b512_mid.json against round 4's line).  ORBX_DESC_TILE=1 forces the tile form (tests).
    static const bool dt_force = [] { const char* e = getenv("ORBX_DESC_TILE"); return e && atoi(e) == 1; }();
    if (g.dt_total > 0 && !unfused && (dt_force || (size_t)g.dt_total * (size_t)n >= (size_t)8 * (size_t)h->n_cu)) {
      ProfScope ps(h, "describe_tile_kernel", st, true);
      // the level's bytes of a window are read where the level lives: level 0 may be the caller's image (row pitch s.l0_pitch >= w >= 64)
      hipLaunchKernelGGL(describe_tile_kernel, xcd_grid(g.dt_total, n), dim3(256), 0, st, s, g, n, xcd_map(g.dt_total), tab + h->dtile_tab_off,
                         (const uint2*)dtile, (const unsigned long long*)sel2, (const unsigned long long*)sel, kp, d_kp + (size_t)img0 * cap_kp,
                         d_desc + (size_t)img0 * cap_kp * 32, d_nkp + img0, cap_kp, (float)h->orb.patch_size, h->d_status);
    } else {
      ProfScope ps(h, unfused ? "describe_kernel" : "describe_fused_kernel", st, true);
      const int blocks_x16 = (h->orb.n_features + 64 + 15) / 16;   // 16 keypoints per block and round
      const int blocks_x = unfused ? blocks_x16 : (blocks_x16 + ORBX_DF_ITERS - 1) / ORBX_DF_ITERS;
      if (unfused)
        hipLaunchKernelGGL(describe_kernel, xcd_grid(blocks_x, n), dim3(256), 0, st, s, g, n, xcd_map(blocks_x), blocks_x,
                           (const unsigned long long*)sel2, (const unsigned long long*)sel, kp, d_kp + (size_t)img0 * cap_kp, d_desc + (size_t)img0 * cap_kp * 32,
                           d_nkp + img0, cap_kp, (float)h->orb.patch_size, h->d_status);
      else
        hipLaunchKernelGGL(describe_fused_kernel, xcd_grid(blocks_x, n), dim3(256), 0, st, s, g, n, xcd_map(blocks_x), blocks_x,
                           (const unsigned long long*)sel2, (const unsigned long long*)sel, kp, d_kp + (size_t)img0 * cap_kp, d_desc + (size_t)img0 * cap_kp * 32,
                           d_nkp + img0, cap_kp, (float)h->orb.patch_size, h->d_status);
    }
  }
  ORBX_HIP(h, hipGetLastError());
  return ORBX_OK;
}

int launch_orb_extract(orbx_handle* h, const uint8_t* d_images, int n_images, int w, int h_px, size_t stride,
                       orbx_keypoint* d_kp, uint8_t* d_desc, int* d_nkp, int cap_kp) {
  if (n_images <= 0) return ORBX_OK;
  if (int rc = orb_extract_prepare(h, n_images, w, h_px)) return rc;
  // (Round 3: the previous batch's stereo matcher on a stream of its own, so that its latency-bound launches run under this batch's resize
  // launches — describe waiting for it, since it rewrites what the matcher reads —: both stretch by what the other takes (resize 0.264 ->
  // 0.360, matcher 0.091 -> 0.155 ms per 256 pairs), frames/s unchanged; withdrawn.)
  // (Two half-batches on two streams were measured: +0.3 % in round 1; again at the end of round 2, when FAST's 7 blocks per CU leave wave
  // slots free: 2 / 3 / 4 / 6 / 8 chunks alternating over two streams: +0.5 ... +2 % / -3 % / 0 / -3 % / -5 % without per-kernel events — so one stream.
  // Round 5: the same with the second stream STAGGERED by a phase, orbx_process_stereo_batch_device.)
  return orb_extract_range(h, h->stream, d_images, n_images, 0, n_images, w, h_px, stride, d_kp, d_desc, d_nkp, cap_kp, nullptr);
}

// ---- stage inspection --------------------------------------------------------------------------------------
extern "C" int orbx_debug_read_level(orbx_handle* h, int image_index, int level, int which, uint8_t* out,
                                     int* w_l, int* h_l) {
  if (!h || !w_l || !h_l) return ORBX_ERR_INVALID;
  const OrbGeom& g = h->geom;
  if (h->geom_w == 0 || level < 0 || level >= g.n_levels || image_index < 0 || (which != 0 && which != 1))
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_debug_read_level: nothing extracted yet or bad index");
  *w_l = g.lv[level].w; *h_l = g.lv[level].h;
  if (!out) return ORBX_OK;
  if (which == 0 && level == 0) return orbx_fail(h, ORBX_ERR_INVALID, "level 0 is the caller's image");
  ORBX_HIP(h, hipSetDevice(h->device));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  if (which == 1) {
    // the blurred level as the CPU specification of A.8 defines it, made here by the whole-level blur kernel (the product path blurs each
    // keypoint's patch inside describe_fused_kernel and keeps no blurred pyramid): same integers, so the descriptors' inputs can be inspected
    if (image_index >= h->last_n_images) return orbx_fail(h, ORBX_ERR_INVALID, "image index out of range");
    if (int rc = orbx_reserve(h, h->ws_blur, (size_t)g.slot_bytes * h->last_n_images)) return rc;
    OrbSrc sb = h->last_src;
    sb.blur = (uint8_t*)h->ws_blur.p;
    hipLaunchKernelGGL(blur_kernel, xcd_grid(g.btiles_total, h->last_n_images), dim3(256), 0, h->stream, sb, g, h->last_n_images, xcd_map(g.btiles_total),
                       (const unsigned*)h->resize_tab.p + h->btile_tab_off);
    ORBX_HIP(h, hipStreamSynchronize(h->stream));
    ORBX_HIP(h, hipGetLastError());
  }
  const DevBuf& b = which ? h->ws_blur : h->ws_pyr;
  if ((size_t)(image_index + 1) * g.slot_bytes > b.bytes) return orbx_fail(h, ORBX_ERR_INVALID, "image index out of range");
  const uint8_t* src = (const uint8_t*)b.p + (size_t)image_index * g.slot_bytes + g.lv[level].off;
  ORBX_HIP(h, hipMemcpy2D(out, g.lv[level].w, src, g.lv[level].pitch, g.lv[level].w, g.lv[level].h, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

extern "C" int orbx_debug_read_candidates(orbx_handle* h, int image_index, int level, uint32_t* out, int cap, int* n) {
  if (!h || !n) return ORBX_ERR_INVALID;
  const OrbGeom& g = h->geom;
  if (h->geom_w == 0 || level < 0 || level >= g.n_levels || image_index < 0)
    return orbx_fail(h, ORBX_ERR_INVALID, "orbx_debug_read_candidates: nothing extracted yet or bad index");
  if ((size_t)(image_index + 1) * g.cand_total * sizeof(unsigned) > h->ws_cand.bytes)
    return orbx_fail(h, ORBX_ERR_INVALID, "image index out of range");
  ORBX_HIP(h, hipSetDevice(h->device));
  ORBX_HIP(h, hipStreamSynchronize(h->stream));
  unsigned cnt = 0;
  ORBX_HIP(h, hipMemcpy(&cnt, (const unsigned*)h->ws_counters.p + (size_t)image_index * g.n_levels + level, sizeof(cnt), hipMemcpyDeviceToHost));
  *n = (int)cnt;
  if (!out) return ORBX_OK;
  if ((int)cnt > cap) return orbx_fail(h, ORBX_ERR_CAPACITY, "candidate buffer too small: need %u", cnt);
  ORBX_HIP(h, hipMemcpy(out, (const unsigned*)h->ws_cand.p + (size_t)image_index * g.cand_total + g.lv[level].cand_off,
                        sizeof(unsigned) * cnt, hipMemcpyDeviceToHost));
  return ORBX_OK;
}
