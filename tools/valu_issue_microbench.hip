// Microbenchmark (gfx950): issue cost of the instructions on K2's factorisation chain, one wave per SIMD, eight
// independent registers per instruction kind (throughput) and one register (dependent latency).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue_microbench tools/valu_issue_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND, int DEP>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, const double* in, int reps) {
  double r[8], a = in[threadIdx.x], b = in[64 + threadIdx.x] * 1e-3;
  float f[8];
  for (int i = 0; i < 8; i++) { r[i] = in[128 + i] + threadIdx.x * 1e-3; f[i] = (float)r[i]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int q = 0; q < reps; q++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#define IDX(i) (DEP ? 0 : i)
#define OP(i)                                                                                                              \
  if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[IDX(i)]) : "v"(a), "v"(b));                               \
  if (KIND == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[IDX(i)]));                                                        \
  if (KIND == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(r[IDX(i)]));                                                        \
  if (KIND == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r[IDX(i)]) : "v"(a)); \
  if (KIND == 4) asm volatile("s_nop 1");                                                                                   \
  if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[IDX(i)]));                                                        \
  if (KIND == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[IDX(i)]) : "v"(r[IDX(i)]));                                   \
  if (KIND == 7) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[IDX(i)]) : "v"(f[IDX(i)]));                                   \
  if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[IDX(i)]) : "v"(a));                                           \
  if (KIND == 9) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(r[IDX(i)]) : "v"(a), "v"(b)); \
  if (KIND == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[IDX(i)]) : "v"(f[7 - IDX(i)]));                    \
  if (KIND == 11) asm volatile("v_mov_b64 %0, %1" : "=v"(r[IDX(i)]) : "v"(a));                                              \
  if (KIND == 12) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a0" : "+v"(f[IDX(i)]) : : "a0");
      REP8(OP)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; i++) s += r[i] + f[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
static double *g_in, *g_out; static unsigned long long* g_cyc;
template <int KIND, int DEP> void run(const char* name) {
  const int reps = 200, blocks = 1024;
  hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps);
  hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, g_cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("%-22s %s: %.2f cycles per instruction\n", name, DEP ? "dependent  " : "independent", (double)h[3] / (reps * 32.0) / (KIND == 12 ? 2 : 1));
}
// the same instruction stream with W waves resident per SIMD (1024 W workgroups of one wave): cycles one wave takes per
// instruction, and per SIMD (that / W) -- what a second resident wave buys an issue-bound kernel
template <int KIND, int DEP> void run_occ(const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int W = 1; W <= 8; W *= 2) {
    const int reps = 4000, blocks = 1024 * W;
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[64];
    (void)hipMemcpy(h, g_cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < 64; i++) m += (double)h[i] / 64;
    printf("%-18s %s, %d wave(s) per SIMD: %.2f cycles per instruction and wave (s_memtime), kernel %.3f ms = %.2f ns per instruction and SIMD\n", name,
           DEP ? "dependent  " : "independent", W, m / (reps * 32.0), ms, ms * 1e6 / (reps * 32.0) / W);
  }
}
int main() {
  (void)hipMalloc(&g_in, 256 * 8); (void)hipMalloc(&g_out, 8192 * 64 * 8); (void)hipMalloc(&g_cyc, 8192 * 8);
  double h[256]; for (int i = 0; i < 256; i++) h[i] = 1.0 + (i % 7) * 0.01;
  (void)hipMemcpy(g_in, h, sizeof h, hipMemcpyHostToDevice);
  run<0, 0>("v_fma_f64"); run<0, 1>("v_fma_f64");
  run<8, 0>("v_mul_f64"); run<8, 1>("v_mul_f64");
  run<9, 0>("v_fmac_f64_dpp"); run<9, 1>("v_fmac_f64_dpp");
  run<1, 0>("v_rcp_f64"); run<1, 1>("v_rcp_f64");
  run<2, 0>("v_rsq_f64"); run<2, 1>("v_rsq_f64");
  run<3, 0>("v_mov_b64_dpp");
  run<11, 0>("v_mov_b64");
  run<4, 0>("s_nop 1");
  run<5, 0>("v_rcp_f32"); run<5, 1>("v_rcp_f32");
  run<6, 0>("v_cvt_f32_f64"); run<7, 0>("v_cvt_f64_f32");
  run<10, 0>("v_cndmask_b32");
  run<12, 0>("v_accvgpr write+read");
  run_occ<0, 0>("v_fma_f64"); run_occ<0, 1>("v_fma_f64"); run_occ<9, 0>("v_fmac_f64_dpp"); run_occ<9, 1>("v_fmac_f64_dpp");
  return 0;
}
