// Which lane does DPP row_shl / row_shr read from?  (prints the source lane seen by lanes 0..15)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
  double x = (double)threadIdx.x;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1.0, x, 0x106, 0xf, 0xf, false);        // row_shl:6
  out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1.0, x, 0x116, 0xf, 0xf, false);   // row_shr:6
}
int main() {
  double* d; (void)hipMalloc(&d, 128 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[128]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("row_shl:6 :"); for (int i = 0; i < 16; i++) printf(" %g", h[i]); printf("\nrow_shr:6 :"); for (int i = 0; i < 16; i++) printf(" %g", h[64 + i]); printf("\n");
  return 0;
}
