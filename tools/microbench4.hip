// What does a divergent dword gather cost per wave instruction on MI355X?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int DP = 47237;
// MODE 0: 64 lanes, random addresses.  1: every 8th lane random, others masked off (exec).
// 2: 64 lanes same address.  3: every 8th lane random, others read w[0] (what the product kernel did)
template <int MODE>
__global__ void __launch_bounds__(1024) k(const float* __restrict__ w, const int* __restrict__ idx, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  const unsigned base = (blockIdx.x * 1024 + threadIdx.x);
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    int c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = idx[(base + (unsigned)(i * 8 + j) * 4099u) & 0xFFFFFu];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) acc += w[c[j]];
      if (MODE == 1) { if ((lane & 7) == j) acc += w[c[j]]; }
      if (MODE == 2) acc += w[c[j] & 0];
      if (MODE == 3) acc += w[(lane & 7) == j ? c[j] : 0];
    }
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int MODE> void run(const char* name, const float* w, const int* idx, float* out) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 64;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, w, idx, out, iters); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, w, idx, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  const double winstr = 256.0 * 16 * iters * 8;   // gather wave-instructions
  printf("%-46s %7.3f ms  %6.1f clk per gather wave-instr per CU (2.25 GHz, 16 waves/CU)\n", name, ms, ms * 1e-3 * 2.25e9 / (winstr / 256));
}
int main() {
  std::vector<float> w(DP, 1.0f); std::vector<int> idx(1 << 20);
  unsigned s = 12345; for (auto& x : idx) { s = s * 1664525u + 1013904223u; x = (s >> 8) % DP; }
  float *dw, *dout; int* didx;
  CK(hipMalloc(&dw, DP * 4)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&didx, idx.size() * 4));
  CK(hipMemcpy(dw, w.data(), DP * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  run<0>("64 lanes, random addresses", dw, didx, dout);
  run<1>("8 of 64 lanes active (exec mask), random", dw, didx, dout);
  run<2>("64 lanes, one address", dw, didx, dout);
  run<3>("8 lanes random + 56 lanes w[0]", dw, didx, dout);
  return 0;
}
