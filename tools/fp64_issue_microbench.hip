// Microbenchmark: fp64 FMA issue rate of ONE wavefront per SIMD as a function of instruction-level parallelism
// (1, 2, 3, 4, 8 independent dependent-chains), measured with s_memtime around a long unrolled loop (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/fp64_issue_microbench tools/fp64_issue_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64) void kern(double* out, unsigned long long* cyc, const double* in, int reps) {
  double acc[ILP], a = in[threadIdx.x], x = in[64 + threadIdx.x] * 1e-3;
  for (int i = 0; i < ILP; i++) acc[i] = in[128 + i];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 32; k++)
#pragma unroll
      for (int i = 0; i < ILP; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(x));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < ILP; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int ILP>
void run(double* out, unsigned long long* cyc, const double* in, int blocks) {
  const int reps = 500;
  hipLaunchKernelGGL(kern<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, in, reps);
  hipLaunchKernelGGL(kern<ILP>, dim3(blocks), dim3(64), 0, 0, out, cyc, in, reps);
  hipDeviceSynchronize();
  unsigned long long h[16];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("blocks %5d (%.2f waves/SIMD) ILP %d: %.2f s_memtime ticks per v_fma_f64 (%.2f per chain step)\n", blocks, blocks / 1024.0, ILP,
         (double)h[3] / (reps * 32.0 * ILP), (double)h[3] / (reps * 32.0));
}
int main() {
  double *in, *out; unsigned long long* cyc;
  hipMalloc(&in, 256 * 8); hipMalloc(&out, 8192 * 64 * 8); hipMalloc(&cyc, 8192 * 8);
  double h[256]; for (int i = 0; i < 256; i++) h[i] = 1.0 + (i % 7) * 0.01;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  for (int blocks : {256, 1024, 2048}) {
    run<1>(out, cyc, in, blocks); run<2>(out, cyc, in, blocks); run<3>(out, cyc, in, blocks);
    run<4>(out, cyc, in, blocks); run<8>(out, cyc, in, blocks);
  }
  return 0;
}
