// Accuracy of v_rcp_f64 / v_rsq_f64 seeds and of one / two Newton steps on gfx950 (how many refinement steps the
// pivots of K2's L D L^T and the Cholesky-style rsqrt really need).  Prints max relative errors over 2^20 samples.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = exp2(-20.0 + 40.0 * (double)i / n) * (1.0 + 0.37 * (double)(i % 977) / 977.0);
  double r0 = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r0, 1.0), r1 = fma(r0, e, r0);
  e = fma(-d, r1, 1.0);
  double r2 = fma(r1, e, r1);
  double ex = 1.0 / d;
  double y0 = __builtin_amdgcn_rsq(d);
  double g = d * y0, h = 0.5 * y0, rr = fma(-h, g, 0.5);
  double g1 = fma(g, rr, g), h1 = fma(h, rr, h);
  double y1 = 2.0 * h1;
  rr = fma(-h1, g1, 0.5);
  double y2 = 2.0 * fma(h1, rr, h1);
  double ey = 1.0 / sqrt(d);
  out[6 * i + 0] = fabs(r0 - ex) / ex; out[6 * i + 1] = fabs(r1 - ex) / ex; out[6 * i + 2] = fabs(r2 - ex) / ex;
  out[6 * i + 3] = fabs(y0 - ey) / ey; out[6 * i + 4] = fabs(y1 - ey) / ey; out[6 * i + 5] = fabs(y2 - ey) / ey;
}
int main() {
  const int n = 1 << 20;
  double *d, *h = (double*)malloc(sizeof(double) * 6 * n);
  hipMalloc(&d, sizeof(double) * 6 * n);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
  hipMemcpy(h, d, sizeof(double) * 6 * n, hipMemcpyDeviceToHost);
  double mx[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) for (int c = 0; c < 6; c++) if (h[6 * i + c] > mx[c]) mx[c] = h[6 * i + c];
  printf("max relative error: v_rcp_f64 %.3g, +1 Newton %.3g, +2 Newton %.3g | v_rsq_f64 %.3g, +1 step %.3g, +2 steps %.3g\n",
         mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]);
  return 0;
}
