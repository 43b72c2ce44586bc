// Microbenchmark: issue cost of global_load_lds_dwordx4 (LDS-DMA) for ONE wavefront, s_memtime ticks per instruction:
//   mode 0  M0 saved / set / restored around every DMA (rl_dma16 of tolg_kernels.hip)
//   mode 1  M0 set once, destinations through the instruction offset (<= 4 KB apart)
//   mode 2  plain global_load_dwordx4 into registers (no LDS), for comparison
// Build: hipcc --offload-arch=gfx950 -O3 -o build_ab/lds_dma_microbench tools/lds_dma_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void kern(const char* g, unsigned long long* cyc, double* out, int reps) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned voff = threadIdx.x * 16;
  const char* base = g + (size_t)blockIdx.x * 65536;
  double acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; r++) {
    const char* b = base + (r & 7) * 8192;
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff + c * 1024), "s"(b), "s"(lds0 + c * 1024) : "memory");
      }
    } else if (MODE == 1) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                   "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %1, %5 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %5 offset:2048\n\tglobal_load_lds_dwordx4 %1, %5 offset:3072\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(b), "s"(lds0), "s"(lds0 + 4096), "s"(b + 4096) : "memory");
    } else {
      typedef double __attribute__((ext_vector_type(2))) f64x2;
      f64x2 v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = *reinterpret_cast<const f64x2*>(b + voff + c * 1024);
#pragma unroll
      for (int c = 0; c < 8; c++) acc += v[c].x;
    }
    if (MODE != 2 && (r & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + threadIdx.x] = acc + lds[threadIdx.x];
}
int main() {
  char* g; unsigned long long* cyc; double* out;
  hipMalloc(&g, 1024 * 65536); hipMalloc(&cyc, 8192 * 8); hipMalloc(&out, 8192 * 64 * 8);
  hipMemset(g, 0, 1024 * 65536);
  const int reps = 256;
  for (int blocks : {1, 256, 1024}) {
    for (int mode = 0; mode < 3; mode++) {
      for (int w = 0; w < 2; w++) {
        if (mode == 0) hipLaunchKernelGGL(kern<0>, dim3(blocks), dim3(64), 0, 0, g, cyc, out, reps);
        if (mode == 1) hipLaunchKernelGGL(kern<1>, dim3(blocks), dim3(64), 0, 0, g, cyc, out, reps);
        if (mode == 2) hipLaunchKernelGGL(kern<2>, dim3(blocks), dim3(64), 0, 0, g, cyc, out, reps);
      }
      hipDeviceSynchronize();
      unsigned long long h[4];
      hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
      printf("blocks %5d mode %d: %.1f ticks per 1 KB instruction (incl. the data wait every 4th repetition)\n", blocks, mode,
             (double)h[0] / (reps * 8.0));
    }
  }
  return 0;
}
