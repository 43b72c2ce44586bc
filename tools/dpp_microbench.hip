// Microbenchmark: v_fmac_f64 with a DPP row_newbcast operand vs plain v_fma_f64 (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/dpp_microbench tools/dpp_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define FM(a, p, q, L) "v_fmac_f64_dpp " a ", " p ", " q " row_newbcast:" #L " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void rank1_bi(double (&acc)[12], double p, double q) {
  asm volatile("s_nop 1\n\t" FM("%0", "%12", "%13", 0) FM("%1", "%12", "%13", 1) FM("%2", "%12", "%13", 2)
                   FM("%3", "%12", "%13", 3) FM("%4", "%12", "%13", 4) FM("%5", "%12", "%13", 5)
                       FM("%6", "%12", "%13", 6) FM("%7", "%12", "%13", 7) FM("%8", "%12", "%13", 8)
                           FM("%9", "%12", "%13", 9) FM("%10", "%12", "%13", 10) FM("%11", "%12", "%13", 11)
               : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                 "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11])
               : "v"(p), "v"(q));
}
__device__ __forceinline__ void rank1_plain(double (&acc)[12], double p, double q) {
#pragma unroll
  for (int i = 0; i < 12; i++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(p), "v"(q));
}
template <int MODE>
__global__ __launch_bounds__(64) void kern(double* out, const double* in, int reps) {
  double acc[12], a[12], x[12];
  for (int i = 0; i < 12; i++) { acc[i] = 0; a[i] = in[i * 64 + threadIdx.x]; x[i] = in[(12 + i) * 64 + threadIdx.x] * 1e-3; }
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
      if (MODE == 0) rank1_bi(acc, a[k], x[k]); else rank1_plain(acc, a[k], x[k]);
    }
  }
  double s = 0;
  for (int i = 0; i < 12; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  double *in, *out;
  hipMalloc(&in, 24 * 64 * 8); hipMalloc(&out, 8192 * 64 * 8);
  double h[24 * 64]; for (int i = 0; i < 24 * 64; i++) h[i] = 1.0 + (i % 7) * 0.01;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  for (int blocks : {256, 1024, 2048, 4096}) {
    for (int mode = 0; mode < 2; mode++) {
      for (int w = 0; w < 2; w++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(64), 0, 0, out, in, reps);
        else hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(64), 0, 0, out, in, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fma = (double)blocks * 64 * reps * 144;
      printf("blocks %5d mode %s: %.3f ms, %.2f TFLOP/s fp64, %.2f cycles/instr/wave at 2.4GHz (waves/SIMD=%.2f)\n", blocks,
             mode == 0 ? "fmac_dpp" : "fma     ", ms, 2 * fma / ms / 1e9, ms * 1e-3 * 2.4e9 / (reps * 144.0) / (blocks / 1024.0 > 1 ? blocks / 1024.0 : 1), blocks / 1024.0);
    }
  }
  return 0;
}
