// Microbenchmark (round 6): the lock-free engine's update of the DENSE HEAD of w -- every worker (workgroup of 512 lanes)
// adds to the same 2,048 consecutive fp32 words once per iteration and waits for the acknowledgement, as
// dsgd_hogwild_kernel does (csrc/dsgd_batch.hpp).  Question: what does one such round cost as the number of workers
// grows, and which layout of the hot words removes the serialisation?
//   shared     one vector, agent-scope atomics (the engine)
//   spread     one vector, one word per 64-byte line (stride 16)
//   replicas R worker k adds to replica k mod R (a reader would add R vectors up)
//   xcd        one replica per XCD, workgroup-scope atomics (executed in that XCD's L2)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench7.hip -o tools/microbench7 && tools/microbench7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int HEAD = 2048, THREADS = 512, SLOTS = HEAD / THREADS;
// MODE 0 shared, 1 spread (stride 16), 2 replicas (rep = block % R), 3 per-XCD replica with workgroup scope
template <int MODE>
__global__ void __launch_bounds__(THREADS) k_head(float* w, int iters, int R, unsigned int density_q16, unsigned long long* t_out) {
  unsigned int xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int tid = threadIdx.x;
  float* base = w;
  if (MODE == 2) base = w + (size_t)(blockIdx.x % R) * HEAD;
  if (MODE == 3) base = w + (size_t)(xcc & 7u) * HEAD;
  unsigned int h = blockIdx.x * 7919u + tid * 104729u + 1u;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < SLOTS; ++e) {
      h = h * 1664525u + 1013904223u;
      if ((h >> 16) < density_q16) {
        const int j = e * THREADS + tid;
        float* p = MODE == 1 ? base + (size_t)j * 16 : base + j;
        if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0) t_out[blockIdx.x] = t1 - t0;
}
int main() {
  float* w; unsigned long long* t;
  CHECK(hipMalloc(&w, sizeof(float) * HEAD * 16 * 2));
  CHECK(hipMalloc(&t, sizeof(unsigned long long) * 1024));
  CHECK(hipMemset(w, 0, sizeof(float) * HEAD * 16 * 2));
  const int iters = 400;
  std::vector<unsigned long long> ht(1024);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct V { const char* name; int mode, R; };
  const V vs[] = {{"shared", 0, 1}, {"spread x16", 1, 1}, {"replicas 2", 2, 2}, {"replicas 4", 2, 4}, {"replicas 8", 2, 8}, {"replicas 16", 2, 16}, {"per-XCD wg-scope", 3, 1}};
  for (double dens : {1.0, 0.3})
    for (const V& v : vs)
      for (int grid : {32, 64, 128, 256}) {
        const unsigned int dq = (unsigned int)(dens * 65536.0);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipEventRecord(e0));
          switch (v.mode) {
            case 0: hipLaunchKernelGGL(k_head<0>, dim3(grid), dim3(THREADS), 0, 0, w, iters, v.R, dq, t); break;
            case 1: hipLaunchKernelGGL(k_head<1>, dim3(grid), dim3(THREADS), 0, 0, w, iters, v.R, dq, t); break;
            case 2: hipLaunchKernelGGL(k_head<2>, dim3(grid), dim3(THREADS), 0, 0, w, iters, v.R, dq, t); break;
            default: hipLaunchKernelGGL(k_head<3>, dim3(grid), dim3(THREADS), 0, 0, w, iters, v.R, dq, t); break;
          }
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("density %.1f  %-18s workers %3d: %7.2f us per round   (%.1f G word-atomics/s)\n", dens, v.name, grid, best * 1e3 / iters,
               dens * HEAD * grid * iters / best / 1e6);
      }
  return 0;
}
