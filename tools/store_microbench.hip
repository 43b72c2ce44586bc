// Microbenchmark: how fast can one-thread-per-(trajectory,knot) write a 145-double record?
//  mode 0: [knot][b/4][field][b%4]        8-byte stores (the current record layout)
//  mode 1: [knot][b/4][field/2][b%4][2]   16-byte stores (field pairs)
//  mode 2: [knot][field][b]               8-byte stores, fully coalesced rows (the v1 layout)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_microbench tools/store_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int F = 146;
typedef double d2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void kern(double* out, const double* in, int Bp, int nk) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)nk * Bp) return;
  int b = (int)(t % Bp), i = (int)(t / Bp);
  double x = in[t % 4096];
  double v[F];
#pragma unroll
  for (int f = 0; f < F; f++) { x = x * 1.0000001 + 0.5; v[f] = x; }  // a little dependent work per field
  double* base = out + (size_t)i * F * Bp;
  if (MODE == 0) {
#pragma unroll
    for (int f = 0; f < F; f++) base[((size_t)(b >> 2) * F + f) * 4 + (b & 3)] = v[f];
  } else if (MODE == 1) {
#pragma unroll
    for (int p = 0; p < F / 2; p++) {
      d2 w = {v[2 * p], v[2 * p + 1]};
      *reinterpret_cast<d2*>(base + ((size_t)(b >> 2) * (F / 2) + p) * 8 + (b & 3) * 2) = w;
    }
  } else {
#pragma unroll
    for (int f = 0; f < F; f++) base[(size_t)f * Bp + b] = v[f];
  }
}
int main() {
  const int Bp = 4096, nk = 201;
  double *in, *out;
  hipMalloc(&in, 4096 * 8); hipMalloc(&out, (size_t)Bp * nk * F * 8);
  hipMemset(in, 0, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  size_t n = (size_t)Bp * nk;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9;
    for (int w = 0; w < 5; w++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(kern<0>, dim3((n + 255) / 256), dim3(256), 0, 0, out, in, Bp, nk);
      if (mode == 1) hipLaunchKernelGGL(kern<1>, dim3((n + 255) / 256), dim3(256), 0, 0, out, in, Bp, nk);
      if (mode == 2) hipLaunchKernelGGL(kern<2>, dim3((n + 255) / 256), dim3(256), 0, 0, out, in, Bp, nk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double gb = (double)n * F * 8 / 1e9;
    printf("mode %d: %.3f ms  %.2f GB  %.2f TB/s\n", mode, best, gb, gb / best);
  }
  return 0;
}
