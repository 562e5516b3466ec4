// valu_issue_probe2 — the opcodes the kernels use that profiles/r02_valu_issue_probe.txt did not measure (the list comes from
// scripts/valu_class_mix.py: "unmeasured" opcodes of the extractor, matcher and BA kernels).  Same method and the same output
// table as scripts/valu_issue_probe.hip: every wave runs ITERS x 64 instructions of ONE opcode on 8 independent accumulators,
// at 1 / 2 / 4 / 8 waves per SIMD on every CU; per occupancy: cycles per instruction as one wave sees it, the SIMD's issue interval,
// the chip-wide rate in G wave-instructions/s from the HIP-event wall time, and the resident fraction.
//   hipcc -O2 --offload-arch=gfx950 scripts/valu_issue_probe2.hip -o scripts/valu_issue_probe2 && scripts/valu_issue_probe2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
#define CKM(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 32-bit accumulators a0..a7 (%0), loop-invariant c0 / c1 (%1 / %2)
#define R8(S)                                                                                                 \
  asm volatile(S : "+v"(a0) : "v"(c0), "v"(c1) : "vcc", "s20"); asm volatile(S : "+v"(a1) : "v"(c0), "v"(c1) : "vcc", "s20"); \
  asm volatile(S : "+v"(a2) : "v"(c0), "v"(c1) : "vcc", "s20"); asm volatile(S : "+v"(a3) : "v"(c0), "v"(c1) : "vcc", "s20"); \
  asm volatile(S : "+v"(a4) : "v"(c0), "v"(c1) : "vcc", "s20"); asm volatile(S : "+v"(a5) : "v"(c0), "v"(c1) : "vcc", "s20"); \
  asm volatile(S : "+v"(a6) : "v"(c0), "v"(c1) : "vcc", "s20"); asm volatile(S : "+v"(a7) : "v"(c0), "v"(c1) : "vcc", "s20");
#define R64(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S)
// 64-bit accumulators q0..q7 (%0 = VGPR pair), loop-invariant k0 (%1, pair) and c0 / c1 (%2 / %3)
#define Q8(S)                                                                                                              \
  asm volatile(S : "+v"(q0) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); asm volatile(S : "+v"(q1) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); \
  asm volatile(S : "+v"(q2) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); asm volatile(S : "+v"(q3) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); \
  asm volatile(S : "+v"(q4) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); asm volatile(S : "+v"(q5) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); \
  asm volatile(S : "+v"(q6) : "v"(k0), "v"(c0), "v"(c1) : "vcc"); asm volatile(S : "+v"(q7) : "v"(k0), "v"(c0), "v"(c1) : "vcc");
#define Q64(S) Q8(S) Q8(S) Q8(S) Q8(S) Q8(S) Q8(S) Q8(S) Q8(S)
// dependent chains: ONE accumulator, every instruction waits for the one before it (latency, not issue rate)
#define D8Q(S) asm volatile(S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S : "+v"(q0) : "v"(k0), "v"(c0), "v"(c1) : "vcc");
#define D64(S) D8Q(S) D8Q(S) D8Q(S) D8Q(S) D8Q(S) D8Q(S) D8Q(S) D8Q(S)
#define D8R(S) asm volatile(S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S : "+v"(a0) : "v"(c0), "v"(c1) : "vcc", "s20");
#define D32(S) D8R(S) D8R(S) D8R(S) D8R(S) D8R(S) D8R(S) D8R(S) D8R(S)

// X(index, printed name, R64 | Q64, asm)
#define OPS(X)                                                                         \
  X(0, "v_subrev_u32", R64, "v_subrev_u32 %0, %1, %0")                                 \
  X(1, "v_ashrrev_i32", R64, "v_ashrrev_i32 %0, 1, %0")                                \
  X(2, "v_or3_b32", R64, "v_or3_b32 %0, %0, %1, %2")                                   \
  X(3, "v_max_i32", R64, "v_max_i32 %0, %0, %1")                                       \
  X(4, "v_max_u32", R64, "v_max_u32 %0, %0, %1")                                       \
  X(5, "v_min_u32", R64, "v_min_u32 %0, %0, %1")                                       \
  X(6, "v_min3_u32", R64, "v_min3_u32 %0, %0, %1, %2")                                 \
  X(7, "v_med3_i32", R64, "v_med3_i32 %0, %0, %1, %2")                                 \
  X(8, "v_mul_i32_i24", R64, "v_mul_i32_i24 %0, %0, %1")                               \
  X(9, "v_mbcnt_hi_u32_b32", R64, "v_mbcnt_hi_u32_b32 %0, %1, %0")                     \
  X(10, "v_cmp_gt_i32", R64, "v_cmp_gt_i32 vcc, %0, %1")                               \
  X(11, "v_cmp_lt_i32", R64, "v_cmp_lt_i32 vcc, %0, %1")                               \
  X(12, "v_cmp_eq_u32", R64, "v_cmp_eq_u32 vcc, %0, %1")                               \
  X(13, "v_cmp_ne_u32", R64, "v_cmp_ne_u32 vcc, %0, %1")                               \
  X(14, "v_cmp_lt_u32", R64, "v_cmp_lt_u32 vcc, %0, %1")                               \
  X(15, "v_cmp_lt_u16", R64, "v_cmp_lt_u16 vcc, %0, %1")                               \
  X(16, "v_cmp_gt_f32", R64, "v_cmp_gt_f32 vcc, %0, %1")                               \
  X(17, "v_cvt_f32_i32", R64, "v_cvt_f32_i32 %0, %0")                                  \
  X(18, "v_cvt_u32_f32", R64, "v_cvt_u32_f32 %0, %0")                                  \
  X(19, "v_add_co_u32", R64, "v_add_co_u32 %0, vcc, %0, %1")                           \
  X(20, "v_addc_co_u32", R64, "v_addc_co_u32 %0, vcc, %0, %1, vcc")                    \
  X(21, "v_readfirstlane_b32", R64, "v_readfirstlane_b32 s20, %0")                     \
  X(22, "v_readlane_b32", R64, "v_readlane_b32 s20, %0, 3")                            \
  X(23, "v_writelane_b32", R64, "v_writelane_b32 %0, 5, 3")                            \
  X(24, "v_add_lshl_u32", R64, "v_add_lshl_u32 %0, %0, %1, 2")                         \
  X(25, "v_bfrev_b32", R64, "v_bfrev_b32 %0, %0")                                      \
  X(26, "v_max_u16", R64, "v_max_u16 %0, %0, %1")                                      \
  X(27, "v_sub_f32", R64, "v_sub_f32 %0, %0, %1")                                      \
  X(28, "v_fmac_f32", R64, "v_fmac_f32 %0, %1, %2")                                    \
  X(29, "v_rcp_f32", R64, "v_rcp_f32 %0, %0")                                          \
  X(30, "v_sqrt_f32", R64, "v_sqrt_f32 %0, %0")                                        \
  X(31, "v_ffbh_u32", R64, "v_ffbh_u32 %0, %0")                                        \
  X(32, "v_xad_u32", R64, "v_xad_u32 %0, %0, %1, %2")                                  \
  X(33, "v_lshl_add_u64", Q64, "v_lshl_add_u64 %0, %0, 2, %1")                         \
  X(34, "v_lshrrev_b64", Q64, "v_lshrrev_b64 %0, 1, %0")                               \
  X(35, "v_lshlrev_b64", Q64, "v_lshlrev_b64 %0, 1, %0")                               \
  X(36, "v_mov_b64", Q64, "v_mov_b64 %0, %1")                                          \
  X(37, "v_pk_mul_f32", Q64, "v_pk_mul_f32 %0, %0, %1")                                \
  X(38, "v_pk_add_f32", Q64, "v_pk_add_f32 %0, %0, %1")                                \
  X(39, "v_add_f64", Q64, "v_add_f64 %0, %0, %1")                                      \
  X(40, "v_mul_f64", Q64, "v_mul_f64 %0, %0, %1")                                      \
  X(41, "v_mad_i64_i32", Q64, "v_mad_i64_i32 %0, vcc, %2, %3, %0")                     \
  X(42, "v_cmp_lt_u64", Q64, "v_cmp_lt_u64 vcc, %0, %1")                               \
  X(43, "v_rcp_f64", Q64, "v_rcp_f64 %0, %0")                                          \
  X(44, "v_cvt_f64_i32 (pair <- lo dword)", Q64, "v_cvt_f64_i32 %0, %2")               \
  X(45, "v_add_u32 (control row)", R64, "v_add_u32 %0, %0, %1")                        \
  X(46, "v_pk_min_i16 (control row)", R64, "v_pk_min_i16 %0, %0, %1")                  \
  X(47, "v_fmac_f64_dpp row_newbcast:3 (8 accumulators)", Q64, "s_nop 1\n v_fmac_f64_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
  X(48, "v_mov_b64_dpp row_newbcast:3", Q64, "s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
  X(49, "v_fma_f64 (8 accumulators)", Q64, "v_fma_f64 %0, %0, %1, %1")                 \
  X(50, "v_fma_f64 DEPENDENT chain (1 accumulator)", D64, "v_fma_f64 %0, %0, %1, %1")  \
  X(51, "v_fmac_f64_dpp DEPENDENT chain", D64, "s_nop 1\n v_fmac_f64_dpp %0, %1, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf") \
  X(52, "v_rsq_f64", Q64, "v_rsq_f64 %0, %0")                                          \
  X(53, "v_rsq_f64 DEPENDENT chain", D64, "v_rsq_f64 %0, %0")                          \
  X(54, "v_add_u32 DEPENDENT chain", D32, "v_add_u32 %0, %0, %1")                      \
  X(55, "v_fma_f32 DEPENDENT chain", D32, "v_fma_f32 %0, %0, %1, %2")
constexpr int OP_COUNT = 56;
#define NAME_ROW(i, n, k, s) n,
static const char* kOpName[OP_COUNT] = {OPS(NAME_ROW)};

template <int OP>
__global__ __launch_bounds__(1024) void probe(unsigned* __restrict__ sink, unsigned long long* __restrict__ cyc, int iters) {
  unsigned a0 = threadIdx.x * 2654435761u, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 17u, a3 = a1 + 29u, a4 = a0 * 3u, a5 = a1 * 5u, a6 = a0 >> 3, a7 = a1 >> 5;
  const unsigned c0 = 0x01030507u ^ (threadIdx.x & 3u), c1 = 0x00020103u;
  unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
  const unsigned long long k0 = 0x3ff0000000000003ull + threadIdx.x;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define BODY_ROW(i, n, k, s) if constexpr (OP == i) { k(s) }
    OPS(BODY_ROW)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long qq = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
  const unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)qq ^ (unsigned)(qq >> 32);
  if (r == 0x12345u) sink[0] = r;                                   // keeps the accumulators alive
  if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void clock_probe(unsigned long long* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  const unsigned c0 = 3, c1 = 5;
  const unsigned long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) { R64("v_add_u32 %0, %0, %1") }
  const unsigned long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[2] = 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = m1 - m0; out[1] = r1 - r0; }
}

template <int OP>
static void run(int n_cu, unsigned* d_sink, unsigned long long* d_cyc, std::vector<std::string>& rows, double clock_ghz) {
  const int iters = 2000;
  char line[640];
  int off = snprintf(line, sizeof line, "%-44s", kOpName[OP]);
  for (int wps : {1, 2, 4, 8}) {
    const int threads = wps == 8 ? 512 : std::min(1024, 256 * wps);
    const int blocks_per_cu = wps * 256 / threads;
    const size_t lds = (160 * 1024) / blocks_per_cu - (blocks_per_cu > 1 ? 2048 : 0);   // LDS pins the blocks per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = n_cu * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(threads), lds, 0, d_sink, d_cyc, 200);
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
    const double med = (double)c[n_waves / 2], n_instr = (double)iters * 64;
    const double cpi = med / n_instr;
    const double rate = (double)n_waves * n_instr / (ms * 1e-3) / 1e9;
    const double resident = med / (clock_ghz * 1e9) / (ms * 1e-3);
    off += snprintf(line + off, sizeof line - off, " | %5.2f %5.2f %6.1f %4.2f", cpi, cpi / wps, rate, resident);
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
  CKM(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  unsigned* d_sink; unsigned long long* d_cyc;
  CKM(hipMalloc(&d_sink, 64));
  CKM(hipMalloc(&d_cyc, (size_t)n_cu * 32 * 8 + 64));
  unsigned long long* d_clk; CKM(hipMalloc(&d_clk, 32)); CKM(hipMemset(d_clk, 0, 32));
  hipLaunchKernelGGL(clock_probe, dim3(n_cu * 8), dim3(256), 0, 0, d_clk, 20000);
  CKM(hipDeviceSynchronize());
  hipLaunchKernelGGL(clock_probe, dim3(n_cu * 8), dim3(256), 0, 0, d_clk, 20000);
  CKM(hipDeviceSynchronize());
  unsigned long long clk[2];
  CKM(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)clk[0] / ((double)clk[1] / 100e6) / 1e9;
  printf("# valu_issue_probe2 on %s (%s), %d CUs, clockRate %.0f MHz; shader clock under load (s_memtime / s_memrealtime): %.3f GHz\n",
         p.name, p.gcnArchName, n_cu, p.clockRate / 1e3, ghz);
  printf("# the opcodes of the product kernels that profiles/r02_valu_issue_probe.txt did not cover; same columns: per occupancy (waves per SIMD =\n"
         "# 1, 2, 4, 8): cycles per instruction seen by ONE wave | SIMD issue interval | chip-wide G wave-instructions/s | resident fraction\n");
  printf("%-44s | %-23s | %-23s | %-23s | %-23s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
  printf("%-44s | %-23s | %-23s | %-23s | %-23s\n", "", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res", "cyc/w  simd   G/s  res");
  std::vector<std::string> rows;
  run_all<0>(n_cu, d_sink, d_cyc, rows, ghz);
  for (auto& r : rows) printf("%s\n", r.c_str());
  return 0;
}
