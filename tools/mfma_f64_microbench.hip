// Microbenchmark (gfx950): what the fp64 matrix pipe offers a 12x12 Riccati step.
//   1. issue interval / dependent latency of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64, one wave per SIMD
//   2. whether independent v_fma_f64 (plain and DPP row_newbcast) of the SAME wave issue beside fp64 MFMAs
//   3. v_permlane16_swap / v_permlane32_swap issue cost
//   4. the A / B / C-D lane maps of v_mfma_f64_16x16x4_f64, checked with exact integer data (asymmetric operands)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_microbench tools/mfma_f64_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: NACC independent accumulators of 16x16x4, K MFMAs each per rep
template <int NACC, int NFMA, int DPP>
__global__ __launch_bounds__(64) void k_mix16(double* out, unsigned long long* cyc, const double* in, int reps) {
  double a = in[threadIdx.x], b = in[64 + threadIdx.x];
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{in[128 + i], 0, 0, 0};
  double f[8];
  for (int i = 0; i < 8; i++) f[i] = in[140 + i];
  double x = in[150] * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFMA; j++) {
          if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[j % 8]) : "v"(a), "v"(x));
          else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[j % 8]) : "v"(a), "v"(x));
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 8; i++) s += f[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int NFMA>
__global__ __launch_bounds__(64) void k_mix4(double* out, unsigned long long* cyc, const double* in, int reps) {
  double a = in[threadIdx.x], b = in[64 + threadIdx.x];
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = in[128 + i];
  double f[8];
  for (int i = 0; i < 8; i++) f[i] = in[140 + i];
  double x = in[150] * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NFMA; j++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[j % 8]) : "v"(a), "v"(x));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  for (int i = 0; i < 8; i++) s += f[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// permlane swaps: 8 independent register pairs
template <int W>
__global__ __launch_bounds__(64) void k_swap(unsigned* out, unsigned long long* cyc, int reps) {
  unsigned v[8], w[8];
  for (int i = 0; i < 8; i++) { v[i] = threadIdx.x * 8 + i; w[i] = 1000 + threadIdx.x * 8 + i; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (W == 16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w[i]));
        else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w[i]));
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += v[i] ^ w[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// semantics of one swap: out[0..63] = v after, out[64..127] = w after (v = lane, w = 100 + lane before)
template <int W>
__global__ void k_swap_sem(unsigned* out) {
  unsigned v = threadIdx.x, w = 100 + threadIdx.x;
  if (W == 16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(w));
  out[threadIdx.x] = v; out[64 + threadIdx.x] = w;
}
// lane maps: D = A B + C with the claimed maps A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[row = (l>>4) + 4 r][col = l&15]
__global__ void k_map(const double* A, const double* B, const double* C, double* D) {  // all 16x16 row-major, K = 16 (4 steps)
  int l = threadIdx.x;
  d4 acc;
  for (int r = 0; r < 4; r++) acc[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
  for (int s = 0; s < 4; s++) {
    double a = A[(l & 15) * 16 + 4 * s + (l >> 4)], b = B[(4 * s + (l >> 4)) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
// chained use: Y = X^T W with X = (previous D registers) as the A operand, and Y2 = W X with X as the B operand
__global__ void k_chain(const double* X, const double* W, double* Yt, double* Yb) {
  int l = threadIdx.x;
  d4 x, yt = {0, 0, 0, 0}, yb = {0, 0, 0, 0};
  for (int r = 0; r < 4; r++) x[r] = X[((l >> 4) + 4 * r) * 16 + (l & 15)];  // D layout
  for (int s = 0; s < 4; s++) {
    double wb = W[(4 * s + (l >> 4)) * 16 + (l & 15)];   // W as B operand
    double wa = W[(l & 15) * 16 + 4 * s + (l >> 4)];     // W as A operand
    yt = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], wb, yt, 0, 0, 0);  // claims X^T W
    yb = __builtin_amdgcn_mfma_f64_16x16x4f64(wa, x[s], yb, 0, 0, 0);  // claims W X
  }
  for (int r = 0; r < 4; r++) {
    Yt[((l >> 4) + 4 * r) * 16 + (l & 15)] = yt[r];
    Yb[((l >> 4) + 4 * r) * 16 + (l & 15)] = yb[r];
  }
}

static double *g_in, *g_out; static unsigned long long* g_cyc;
template <class F> void timeit(const char* name, F launch, int blocks, int reps, double per) {
  launch(); launch();
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, g_cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("%-44s blocks %5d: %.2f cycles per unit\n", name, blocks, (double)h[3] / (reps * per));
}
#define T16(NACC, NFMA, DPP, blocks) timeit("mfma16x16x4 nacc=" #NACC " +fma=" #NFMA " dpp=" #DPP, [&] { hipLaunchKernelGGL((k_mix16<NACC, NFMA, DPP>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps); }, blocks, reps, 8.0 * NACC)
#define T4(NACC, NFMA, blocks) timeit("mfma4x4x4_4b nacc=" #NACC " +fma=" #NFMA, [&] { hipLaunchKernelGGL((k_mix4<NACC, NFMA>), dim3(blocks), dim3(64), 0, 0, g_out, g_cyc, g_in, reps); }, blocks, reps, 8.0 * NACC)
int main() {
  hipMalloc(&g_in, 256 * 8); hipMalloc(&g_out, 8192 * 64 * 8); hipMalloc(&g_cyc, 8192 * 8);
  double h[256]; for (int i = 0; i < 256; i++) h[i] = 1.0 + (i % 7) * 0.01;
  hipMemcpy(g_in, h, sizeof h, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int blocks : {1024, 2048}) {
    T16(1, 0, 0, blocks); T16(2, 0, 0, blocks); T16(4, 0, 0, blocks);
    T16(4, 4, 0, blocks); T16(4, 8, 0, blocks); T16(4, 12, 0, blocks); T16(4, 16, 0, blocks);
    T16(4, 8, 1, blocks); T16(4, 12, 1, blocks);
    T16(1, 4, 0, blocks); T16(1, 8, 0, blocks);
    T4(1, 0, blocks); T4(4, 0, blocks); T4(8, 0, blocks); T4(8, 2, blocks); T4(8, 4, blocks);
  }
  for (int blocks : {1024}) {
    timeit("permlane16_swap", [&] { hipLaunchKernelGGL(k_swap<16>, dim3(blocks), dim3(64), 0, 0, (unsigned*)g_out, g_cyc, reps); }, blocks, reps, 32.0);
    timeit("permlane32_swap", [&] { hipLaunchKernelGGL(k_swap<32>, dim3(blocks), dim3(64), 0, 0, (unsigned*)g_out, g_cyc, reps); }, blocks, reps, 32.0);
  }
  {
    unsigned hs[128];
    for (int W : {16, 32}) {
      if (W == 16) hipLaunchKernelGGL(k_swap_sem<16>, dim3(1), dim3(64), 0, 0, (unsigned*)g_out);
      else hipLaunchKernelGGL(k_swap_sem<32>, dim3(1), dim3(64), 0, 0, (unsigned*)g_out);
      hipMemcpy(hs, g_out, sizeof hs, hipMemcpyDeviceToHost);
      printf("permlane%d_swap v(after):", W); for (int i = 0; i < 64; i += 4) printf(" %u", hs[i]); printf("\n");
      printf("permlane%d_swap w(after):", W); for (int i = 0; i < 64; i += 4) printf(" %u", hs[64 + i]); printf("\n");
    }
  }
  {
    double A[256], B[256], C[256], D[256], R[256];
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { A[i * 16 + j] = (i * 3 + j * 7) % 11 - 5; B[i * 16 + j] = (i * 5 + j * 2 + (i > j)) % 13 - 6; C[i * 16 + j] = i - 2 * j; }
    double *dA, *dB, *dC, *dD, *dE;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048); hipMalloc(&dE, 2048);
    hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B, 2048, hipMemcpyHostToDevice); hipMemcpy(dC, C, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D, dD, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = C[i * 16 + j]; for (int k = 0; k < 16; k++) s += A[i * 16 + k] * B[k * 16 + j]; R[i * 16 + j] = s; bad += (s != D[i * 16 + j]); }
    printf("lane map check D = A B + C: %d mismatches of 256\n", bad);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, dE);
    double Yt[256], Yb[256];
    hipMemcpy(Yt, dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(Yb, dE, 2048, hipMemcpyDeviceToHost);
    int bt = 0, bb = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
      double st = 0, sb = 0;
      for (int k = 0; k < 16; k++) { st += A[k * 16 + i] * B[k * 16 + j]; sb += B[i * 16 + k] * A[k * 16 + j]; }
      bt += (st != Yt[i * 16 + j]); bb += (sb != Yb[i * 16 + j]);
    }
    printf("D-layout registers as A operand give X^T W: %d mismatches; as B operand give W X: %d mismatches\n", bt, bb);
  }
  return 0;
}
