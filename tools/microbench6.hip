// Microbenchmark (round 6): 64-bit integer atomic adds into a 230 KB table per XCD -- the cold gradient columns of a
// row-chunk step (28.8 K columns, 1.8 M contributions per step at N = 804,414) -- by scope:
//   agent      one table for the whole device (what the WIDE path of cg_tile does)
//   workgroup  one table per XCD, executed in that XCD's L2 (what bt_add<1> does for a private strip)
// and with the table in LDS for comparison (what cg_tile does today, per workgroup).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench6.hip -o tools/microbench6 && tools/microbench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NC = 28841;
__device__ __forceinline__ unsigned int mix(unsigned int z) {
  z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
  return z;
}
template <int SCOPE>   // 0 agent, 1 workgroup (per-XCD table)
__global__ void __launch_bounds__(1024) k_atomics(unsigned long long* tab, int per_lane, int stream_words, const float4* __restrict__ stream, float* sink) {
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long* t = tab + (SCOPE == 1 ? (size_t)(xcc & 7u) * NC : 0);
  unsigned int h = blockIdx.x * 1024u + threadIdx.x;
  float acc = 0.f;
  // an HBM stream beside the atomics (stream_words float4 per lane), as the kernel's tiles are
  const float4* s = stream + (size_t)blockIdx.x * 1024 * stream_words + threadIdx.x;
  for (int i = 0; i < per_lane; ++i) {
    h = mix(h + 0x9e3779b9u * (i + 1));
    const unsigned int c = h % NC;
    if (SCOPE == 1) __hip_atomic_fetch_add(&t[c], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(&t[c], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i < stream_words) { const float4 v = s[(size_t)i * 1024]; acc += v.x + v.y + v.z + v.w; }
  }
  if (acc == 12345.f) sink[0] = acc;
}
int main() {
  unsigned long long* tab; float4* stream; float* sink;
  const int per_lane = 8, sw = 8;
  const size_t stream_bytes = (size_t)256 * 1024 * sw * 16 * 8;
  CHECK(hipMalloc(&tab, sizeof(unsigned long long) * 8 * NC));
  CHECK(hipMalloc(&stream, stream_bytes));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(tab, 0, sizeof(unsigned long long) * 8 * NC));
  CHECK(hipMemset(stream, 0, stream_bytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int scope = 0; scope < 2; ++scope)
    for (int with_stream = 0; with_stream < 2; ++with_stream)
      for (int grid : {256, 2048}) {
        const int swv = with_stream ? sw : 0;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CHECK(hipEventRecord(e0));
          if (scope) hipLaunchKernelGGL(k_atomics<1>, dim3(grid), dim3(1024), 0, 0, tab, per_lane, swv, stream, sink);
          else hipLaunchKernelGGL(k_atomics<0>, dim3(grid), dim3(1024), 0, 0, tab, per_lane, swv, stream, sink);
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double n = (double)grid * 1024 * per_lane;
        printf("scope %-9s stream %d grid %4d: %.1f us, %.1f G atomics/s%s\n", scope ? "workgroup" : "agent", with_stream, grid, best * 1e3, n / best / 1e6,
               with_stream ? "" : "");
      }
  std::vector<unsigned long long> h(8 * NC);
  CHECK(hipMemcpy(h.data(), tab, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  unsigned long long tot = 0; for (auto v : h) tot += v;
  printf("sum of all tables %llu (every atomic counted once: expected %llu)\n", tot, (unsigned long long)(2 * 2 * 5) * (256ull + 2048ull) * 1024ull * per_lane);
  return 0;
}
