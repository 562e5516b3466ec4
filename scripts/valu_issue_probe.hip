// valu_issue_probe — measures the wave64 vector-instruction issue rate of gfx950 for the instruction mix the
// extractor kernels are made of, at 1 / 2 / 4 / 8 resident waves per SIMD.  The numbers price `roofline.valu_issue`
// in bench.py (the guide gives 2 cycles per wave64 VALU on a SIMD-32 with more than one wave resident, 4 for a wave
// alone — MI355X_MICROARCH.md "Wave scheduling" and the cycle-constants table; this program checks which of the two
// holds for packed-i16 / v_perm / dot4 / 24-bit multiplies, which the guide does not list).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_issue_probe.hip -o scripts/valu_issue_probe && scripts/valu_issue_probe
//
// Method: every wave runs ITERS iterations of 64 instructions of ONE opcode on 8 independent accumulators (inline
// asm, so nothing is folded or reordered); each wave stamps s_memtime around its loop.  Occupancy is pinned by the
// dynamic LDS size (160 KiB / blocks per CU) and a grid that fills every CU exactly.  Reported per opcode and
// occupancy: cycles per instruction as ONE wave sees it, the SIMD's issue interval (that / waves per SIMD), and the
// chip-wide rate in G wave-instructions/s from the HIP-event wall time.  Output: a text table on stdout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 instructions, one per accumulator; S is the asm template with %0 = accumulator, %8/%9 = loop-invariant operands
#define REP8(S)                                                                                        \
  asm volatile(S : "+v"(a0) : "v"(c0), "v"(c1)); asm volatile(S : "+v"(a1) : "v"(c0), "v"(c1));        \
  asm volatile(S : "+v"(a2) : "v"(c0), "v"(c1)); asm volatile(S : "+v"(a3) : "v"(c0), "v"(c1));        \
  asm volatile(S : "+v"(a4) : "v"(c0), "v"(c1)); asm volatile(S : "+v"(a5) : "v"(c0), "v"(c1));        \
  asm volatile(S : "+v"(a6) : "v"(c0), "v"(c1)); asm volatile(S : "+v"(a7) : "v"(c0), "v"(c1));
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

enum Op {
  OP_ADD_U32, OP_AND_B32, OP_LSHRREV, OP_PK_MIN_I16, OP_PK_MAX_I16, OP_PK_SUB_I16, OP_PK_MAD_I16, OP_PK_MIN_OPSEL,
  OP_PERM, OP_ALIGNBYTE, OP_DOT4_U8, OP_DOT2_U16, OP_MUL_U24, OP_MAD_U24, OP_MUL_LO_U32, OP_MAD_U64_U32, OP_MIN_I32, OP_MIN3_I32,
  OP_CNDMASK, OP_CMP_GT, OP_MBCNT, OP_DPP_SHR, OP_CVT_F32_U32, OP_FMA_F32, OP_PK_FMA_F32, OP_FMA_F64, OP_BFE, OP_SAD_U8,
  OP_OR_B32, OP_XOR_B32, OP_SUB_U32, OP_LSHLREV, OP_MOV_B32, OP_MIN_F32, OP_MAX_F32, OP_MAX3_F32, OP_MIN3_F32, OP_ADD_F32, OP_MUL_F32,
  OP_PK_MIN_F16, OP_PK_MAX_F16, OP_PK_ADD_F16, OP_PK_MIN3_F16, OP_PK_MAX3_F16, OP_PK_MAX3_F16_OPSEL, OP_BITOP3, OP_AND_OR, OP_LSHL_OR, OP_ADD3, OP_LSHL_ADD,
  OP_BCNT, OP_ADD_SDWA, OP_CNDMASK_SGPR, OP_CVT_F32_UBYTE0, OP_MIN_U16, OP_MAX3_I16, OP_PK_SUB_U16_CLAMP, OP_MSAD_U8, OP_MAD_I32_I24, OP_CVT_PKRTZ,
  OP_MIX_SALU, OP_MIX_LDS, OP_MIX_LDS_ATOMIC, OP_COUNT
};
static const char* kOpName[OP_COUNT] = {
  "v_add_u32", "v_and_b32", "v_lshrrev_b32", "v_pk_min_i16", "v_pk_max_i16", "v_pk_sub_i16", "v_pk_mad_i16", "v_pk_min_i16 op_sel",
  "v_perm_b32", "v_alignbyte_b32", "v_dot4_u32_u8", "v_dot2_u32_u16", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo_u32", "v_mad_u64_u32",
  "v_min_i32", "v_min3_i32", "v_cndmask_b32", "v_cmp_gt_u32", "v_mbcnt_lo_u32_b32", "v_mov_b32 dpp wave_shr:1", "v_cvt_f32_u32", "v_fma_f32",
  "v_pk_fma_f32", "v_fma_f64", "v_bfe_u32", "v_sad_u8",
  "v_or_b32", "v_xor_b32", "v_sub_u32", "v_lshlrev_b32", "v_mov_b32", "v_min_f32", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_add_f32", "v_mul_f32",
  "v_pk_min_f16", "v_pk_max_f16", "v_pk_add_f16", "v_pk_minimum3_f16", "v_pk_maximum3_f16", "v_pk_maximum3_f16 op_sel", "v_bitop3_b32", "v_and_or_b32",
  "v_lshl_or_b32", "v_add3_u32", "v_lshl_add_u32", "v_bcnt_u32_b32", "v_add_u32_sdwa (byte selects)", "v_cndmask_b32 (sgpr-pair mask)",
  "v_cvt_f32_ubyte0", "v_min_u16", "v_max3_i16", "v_pk_sub_u16 clamp", "v_msad_u8", "v_mad_i32_i24", "v_cvt_pkrtz_f16_f32",
  "2 v_add_u32 : 1 s_add_u32 (VALU counted)", "4 v_add_u32 : 1 ds_read_b32 (VALU counted)", "4 v_add_u32 : 1 ds_add_rtn_u32, 16 lanes/wave one address"};

template <int OP>
__global__ __launch_bounds__(1024) void probe(unsigned* __restrict__ sink, unsigned long long* __restrict__ cyc, int iters) {
  extern __shared__ unsigned lds[];
  unsigned a0 = threadIdx.x * 2654435761u, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 17u, a3 = a1 + 29u, a4 = a0 * 3u, a5 = a1 * 5u, a6 = a0 >> 3, a7 = a1 >> 5;
  const unsigned c0 = 0x01030507u ^ (threadIdx.x & 3u), c1 = 0x00020103u;
  if (OP == OP_MIX_LDS) { lds[threadIdx.x] = a0; __syncthreads(); }
  unsigned long long d0 = a0, d1 = a1;       // 64-bit accumulators for the f64 / u64 rows
  unsigned s = 0;
  if (OP == OP_CNDMASK_SGPR) asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == OP_ADD_U32) { REP64("v_add_u32 %0, %0, %1") }
    else if (OP == OP_AND_B32) { REP64("v_and_b32 %0, %0, %1") }
    else if (OP == OP_LSHRREV) { REP64("v_lshrrev_b32 %0, 1, %0") }
    else if (OP == OP_PK_MIN_I16) { REP64("v_pk_min_i16 %0, %0, %1") }
    else if (OP == OP_PK_MAX_I16) { REP64("v_pk_max_i16 %0, %0, %1") }
    else if (OP == OP_PK_SUB_I16) { REP64("v_pk_sub_i16 %0, %0, %1") }
    else if (OP == OP_PK_MAD_I16) { REP64("v_pk_mad_i16 %0, %0, %1, %2") }
    else if (OP == OP_PK_MIN_OPSEL) { REP64("v_pk_min_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]") }
    else if (OP == OP_PERM) { REP64("v_perm_b32 %0, %0, %1, %2") }
    else if (OP == OP_ALIGNBYTE) { REP64("v_alignbyte_b32 %0, %0, %1, 1") }
    else if (OP == OP_DOT4_U8) { REP64("v_dot4_u32_u8 %0, %1, %2, %0") }
    else if (OP == OP_DOT2_U16) { REP64("v_dot2_u32_u16 %0, %1, %2, %0") }
    else if (OP == OP_MUL_U24) { REP64("v_mul_u32_u24 %0, %0, %1") }
    else if (OP == OP_MAD_U24) { REP64("v_mad_u32_u24 %0, %0, %1, %2") }
    else if (OP == OP_MUL_LO_U32) { REP64("v_mul_lo_u32 %0, %0, %1") }
    else if (OP == OP_MIN_I32) { REP64("v_min_i32 %0, %0, %1") }
    else if (OP == OP_MIN3_I32) { REP64("v_min3_i32 %0, %0, %1, %2") }
    else if (OP == OP_CNDMASK) { REP64("v_cndmask_b32 %0, %0, %1, vcc") }
    else if (OP == OP_CMP_GT) { REP64("v_cmp_gt_u32 vcc, %0, %1") }
    else if (OP == OP_MBCNT) { REP64("v_mbcnt_lo_u32_b32 %0, %1, %0") }
    else if (OP == OP_DPP_SHR) { REP64("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf") }
    else if (OP == OP_CVT_F32_U32) { REP64("v_cvt_f32_u32 %0, %0") }
    else if (OP == OP_FMA_F32) { REP64("v_fma_f32 %0, %0, %1, %2") }
    else if (OP == OP_BFE) { REP64("v_bfe_u32 %0, %0, 3, 8") }
    else if (OP == OP_SAD_U8) { REP64("v_sad_u8 %0, %1, %2, %0") }
    else if (OP == OP_OR_B32) { REP64("v_or_b32 %0, %0, %1") }
    else if (OP == OP_XOR_B32) { REP64("v_xor_b32 %0, %0, %1") }
    else if (OP == OP_SUB_U32) { REP64("v_sub_u32 %0, %0, %1") }
    else if (OP == OP_LSHLREV) { REP64("v_lshlrev_b32 %0, 1, %0") }
    else if (OP == OP_MOV_B32) { REP64("v_mov_b32 %0, %1") }
    else if (OP == OP_MIN_F32) { REP64("v_min_f32 %0, %0, %1") }
    else if (OP == OP_MAX_F32) { REP64("v_max_f32 %0, %0, %1") }
    else if (OP == OP_MAX3_F32) { REP64("v_max3_f32 %0, %0, %1, %2") }
    else if (OP == OP_MIN3_F32) { REP64("v_min3_f32 %0, %0, %1, %2") }
    else if (OP == OP_ADD_F32) { REP64("v_add_f32 %0, %0, %1") }
    else if (OP == OP_MUL_F32) { REP64("v_mul_f32 %0, %0, %1") }
    else if (OP == OP_PK_MIN_F16) { REP64("v_pk_min_f16 %0, %0, %1") }
    else if (OP == OP_PK_MAX_F16) { REP64("v_pk_max_f16 %0, %0, %1") }
    else if (OP == OP_PK_ADD_F16) { REP64("v_pk_add_f16 %0, %0, %1") }
    else if (OP == OP_PK_MIN3_F16) { REP64("v_pk_minimum3_f16 %0, %0, %1, %2") }
    else if (OP == OP_PK_MAX3_F16) { REP64("v_pk_maximum3_f16 %0, %0, %1, %2") }
    else if (OP == OP_PK_MAX3_F16_OPSEL) { REP64("v_pk_maximum3_f16 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]") }
    else if (OP == OP_BITOP3) { REP64("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96") }
    else if (OP == OP_AND_OR) { REP64("v_and_or_b32 %0, %0, %1, %2") }
    else if (OP == OP_LSHL_OR) { REP64("v_lshl_or_b32 %0, %0, 3, %2") }
    else if (OP == OP_ADD3) { REP64("v_add3_u32 %0, %0, %1, %2") }
    else if (OP == OP_LSHL_ADD) { REP64("v_lshl_add_u32 %0, %0, 2, %1") }
    else if (OP == OP_BCNT) { REP64("v_bcnt_u32_b32 %0, %1, %0") }
    else if (OP == OP_ADD_SDWA) { REP64("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2") }
    else if (OP == OP_CNDMASK_SGPR) { REP64("v_cndmask_b32 %0, %0, %1, s[20:21]") }
    else if (OP == OP_CVT_F32_UBYTE0) { REP64("v_cvt_f32_ubyte0 %0, %0") }
    else if (OP == OP_MIN_U16) { REP64("v_min_u16 %0, %0, %1") }
    else if (OP == OP_MAX3_I16) { REP64("v_max3_i16 %0, %0, %1, %2") }
    else if (OP == OP_PK_SUB_U16_CLAMP) { REP64("v_pk_sub_u16 %0, %0, %1 clamp") }
    else if (OP == OP_MSAD_U8) { REP64("v_msad_u8 %0, %1, %2, %0") }
    else if (OP == OP_MAD_I32_I24) { REP64("v_mad_i32_i24 %0, %0, %1, %2") }
    else if (OP == OP_CVT_PKRTZ) { REP64("v_cvt_pkrtz_f16_f32 %0, %0, %1") }
    else if (OP == OP_MIX_LDS_ATOMIC) {
      const unsigned addr = 4096u;                 // one address; the lanes 0, 4, 8, ... of the wave add to it
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        unsigned r = 0;
        if ((threadIdx.x & 3u) == 0u) asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(c1));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(c0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(c0));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(c0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(c0));
        asm volatile("s_waitcnt lgkmcnt(0)");
        a4 ^= r;
      }
    }
    else if (OP == OP_PK_FMA_F32 || OP == OP_FMA_F64 || OP == OP_MAD_U64_U32) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        if (OP == OP_PK_FMA_F32) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d0)); asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d1)); }
        else if (OP == OP_FMA_F64) { asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d0)); asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d1)); }
        else { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d0) : "v"(c0), "v"(c1) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d1) : "v"(c0), "v"(c1) : "vcc"); }
      }
    } else if (OP == OP_MIX_SALU) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#define MIX3(x, y) asm volatile("v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %3\n s_add_u32 %2, %2, 1" : "+v"(x), "+v"(y), "+s"(s) : "v"(c0) : "scc");
        MIX3(a0, a1) MIX3(a2, a3) MIX3(a4, a5) MIX3(a6, a7)
#undef MIX3
      }                                                             // 64 VALU + 32 SALU per iteration
    } else if (OP == OP_MIX_LDS) {
      unsigned addr = (threadIdx.x & 1023u) * 4u;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        unsigned r;
        asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(addr));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(c0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a1) : "v"(c0));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(c0)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a3) : "v"(c0));
        asm volatile("s_waitcnt lgkmcnt(0)");
        a4 ^= r;
      }                                                             // 64 VALU (+16 xor) + 16 LDS per iteration
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)d0 ^ (unsigned)(d0 >> 32) ^ (unsigned)d1 ^ s;
  if (r == 0x12345u) sink[0] = r;                                   // keeps the accumulators alive
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// effective shader clock while the chip is busy: s_memtime (shader cycles) against s_memrealtime (100 MHz)
__global__ void clock_probe(unsigned long long* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  const unsigned c0 = 3, c1 = 5;
  const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) { REP64("v_add_u32 %0, %0, %1") }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[2] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = m1 - m0; out[1] = r1 - r0; }
}

template <int OP>
static void run(int n_cu, unsigned* d_sink, unsigned long long* d_cyc, std::vector<std::string>& rows, double clock_ghz) {
  const int iters = 2000;
  char line[640];
  int off = snprintf(line, sizeof line, "%-44s", kOpName[OP]);
  const int per_iter = 64;        // VALU instructions counted per iteration in every variant
  for (int wps : {1, 2, 4, 8}) {
    const int threads = wps == 8 ? 512 : std::min(1024, 256 * wps);
    const int blocks_per_cu = wps * 256 / threads;
    const size_t lds = (160 * 1024) / blocks_per_cu - (blocks_per_cu > 1 ? 2048 : 0);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(threads), lds, 0, d_sink, d_cyc, 200);       // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(threads), lds, 0, d_sink, d_cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const size_t n_waves = (size_t)grid * threads / 64;
    std::vector<unsigned long long> c(n_waves);
    CK(hipMemcpy(c.data(), d_cyc, n_waves * 8, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double med = (double)c[n_waves / 2], n_instr = (double)iters * per_iter;
    const double cyc_per_instr_wave = med / n_instr;
    const double rate = (double)n_waves * n_instr / (ms * 1e-3) / 1e9;
    // resident = how much of the launch ran concurrently: (sum of the waves' own durations) / (waves x wall time), from the
    // clock measured under load; well under 1 means the blocks did not all fit at once and the column is not that occupancy
    const double resident = med / (clock_ghz * 1e9) / (ms * 1e-3);
    off += snprintf(line + off, sizeof line - off, " | %5.2f %5.2f %6.1f %4.2f", cyc_per_instr_wave, cyc_per_instr_wave / wps, rate, resident);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  rows.push_back(line);
}

template <int OP>
static void run_all(int n_cu, unsigned* s, unsigned long long* c, std::vector<std::string>& rows, double ghz) {
  run<OP>(n_cu, s, c, rows, ghz);
  if constexpr (OP + 1 < OP_COUNT) run_all<OP + 1>(n_cu, s, c, rows, ghz);
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  unsigned* d_sink; unsigned long long* d_cyc;
  CK(hipMalloc(&d_sink, 64));
  CK(hipMalloc(&d_cyc, (size_t)n_cu * 32 * 8 + 64));
  // clock under load: every CU busy for ~a few ms
  unsigned long long* d_clk; CK(hipMalloc(&d_clk, 32)); CK(hipMemset(d_clk, 0, 32));
  hipLaunchKernelGGL(clock_probe, dim3(n_cu * 8), dim3(256), 0, 0, d_clk, 20000);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(clock_probe, dim3(n_cu * 8), dim3(256), 0, 0, d_clk, 20000);
  CK(hipDeviceSynchronize());
  unsigned long long clk[2];
  CK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
  printf("# valu_issue_probe on %s (%s), %d CUs, clockRate %.0f MHz; shader clock under load (s_memtime / s_memrealtime): %.3f GHz\n",
         p.name, p.gcnArchName, n_cu, p.clockRate / 1e3, ghz);
  printf("# per occupancy (waves per SIMD = 1, 2, 4, 8): cycles per instruction seen by ONE wave | SIMD issue interval in cycles (= that / waves) |\n"
         "# chip-wide G wave-instructions/s from HIP-event wall time.  4 SIMDs x %d CUs = %d SIMDs; at I cycles per instruction and f GHz the\n"
         "# chip issues %d * f / I G wave-instr/s (2 cycles at 2.4 GHz: %.1f; 4 cycles: %.1f).\n",
         n_cu, 4 * n_cu, 4 * n_cu, 4 * n_cu * 2.4 / 2, 4 * n_cu * 2.4 / 4);
  printf("%-44s | %-23s | %-23s | %-23s | %-23s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
  printf("%-44s | %-23s | %-23s | %-23s | %-23s\n", "", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res");
  std::vector<std::string> rows;
  run_all<0>(n_cu, d_sink, d_cyc, rows, ghz);
  for (auto& r : rows) printf("%s\n", r.c_str());
  return 0;
}
