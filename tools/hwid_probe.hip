// Where do the waves of a workgroup land?  512 workgroups of 4 waves (70 KB of LDS each: two fit a CU), every wave
// records HW_ID / XCC_ID; the host tabulates SIMD placement inside a workgroup and which workgroups share a CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/hwid_probe tools/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  __shared__ char pad[70 * 1024];
  pad[threadIdx.x] = 0;
  unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // HW_REG_XCC_ID
  // stay resident long enough for the whole grid to be placed
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  if (pad[threadIdx.x] == 77) out[0] = 0;
}
int main() {
  const int WG = 512;
  unsigned* d; hipMalloc(&d, WG * 4 * 2 * 4);
  std::vector<unsigned> h(WG * 4 * 2);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(probe, dim3(WG), dim3(256), 0, 0, d, 20000);  // 200 us
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int distinct4 = 0;
    std::map<unsigned, std::vector<int>> cu_wgs;
    std::map<std::vector<int>, int> patterns;
    for (int b = 0; b < WG; b++) {
      std::set<int> simds; std::vector<int> pat;
      unsigned cuid = 0;
      for (int w = 0; w < 4; w++) {
        unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
        int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        simds.insert(simd); pat.push_back(simd);
        cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      }
      if (simds.size() == 4) distinct4++;
      patterns[pat]++;
      cu_wgs[cuid].push_back(b);
    }
    printf("launch %d: %d of %d workgroups have their 4 waves on 4 different SIMDs; %zu distinct CUs used\n", rep, distinct4, WG, cu_wgs.size());
    for (auto& p : patterns) printf("  SIMD pattern %d %d %d %d : %d workgroups\n", p.first[0], p.first[1], p.first[2], p.first[3], p.second);
    std::map<size_t, int> per; std::map<int, int> diff;
    for (auto& c : cu_wgs) { per[c.second.size()]++; if (c.second.size() == 2) diff[c.second[1] - c.second[0]]++; }
    for (auto& p : per) printf("  CUs hosting %zu workgroups: %d\n", p.first, p.second);
    int shown = 0;
    for (auto& p : diff) if (shown++ < 8) printf("  blockIdx distance of the two workgroups of a CU %d : %d CUs\n", p.first, p.second);
  }
  return 0;
}
