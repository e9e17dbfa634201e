// What do the lock-free engine's weight updates cost, and what would XCD-local replicas buy?  (not part of the product)
//   A: every workgroup sweeps the 20,480 hottest ranks and adds to ONE shared vector with agent-scope atomics (today)
//   B: the same sweep into the replica of the workgroup's own XCD with workgroup-scope atomics (executed in that XCD's L2)
//   C: B + a second L2 atomic per update into an outbox of the XCD (the pending part a merge pass would forward)
//   D: C + per iteration a slice of the outbox is taken (exchange) and forwarded to the shared vector with agent-scope
//      atomics, the view refreshed from it (the merge traffic itself)
// Every variant is CHECKED: the sum over the vector(s) must be the number of adds times the addend (no lost update), and
// in D the shared vector alone must hold everything after a final merge.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench5 tools/microbench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int DP = 47237, HL = 20480, NT = 512, NX = 8, NSL = 32;
__device__ __forceinline__ unsigned int mix(unsigned int x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); }
__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}
struct Args {
  float* w;            // shared vector
  float* view;         // [NX][DP]
  float* outbox;       // [NX][DP]
  unsigned int* tick;  // [NX]
  unsigned long long* count;   // [NX] adds issued by the workgroups of the XCD, [NX + x] workgroups seen there
  int iters, density;  // one column in `density` gets an update
};
template <int MODE>
__global__ void __launch_bounds__(NT) k(Args a) {
  const int tid = threadIdx.x, x = xcc_id();
  float* v = a.view + (long long)x * DP;
  float* o = a.outbox + (long long)x * DP;
  unsigned int n = 0;
  if (tid == 0) atomicAdd(&a.count[NX + x], 1ull);
  for (int it = 0; it < a.iters; ++it) {
    for (int j = tid; j < HL; j += NT) {
      if (mix((unsigned)j * 2654435761u + (unsigned)it * 97u + blockIdx.x * 7919u) % (unsigned)a.density == 0u) {
        if (MODE == 0) atomicAdd(&a.w[j], -1.0f);
        else {
          __hip_atomic_fetch_add(&v[j], -1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (MODE >= 2) __hip_atomic_fetch_add(&o[j], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        ++n;
      }
    }
    if (MODE == 3) {   // the merge: one slice of the XCD's outbox per iteration, whoever comes by takes the next
      __shared__ unsigned int sl;
      __syncthreads();
      if (tid == 0) sl = __hip_atomic_fetch_add(&a.tick[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) % NSL;
      __syncthreads();
      const int per = (DP + NSL - 1) / NSL, j0 = (int)sl * per;
      for (int j = j0 + tid; j < j0 + per && j < DP; j += NT) {
        const float p = __hip_atomic_exchange(&o[j], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        float wn;
        if (p != 0.0f) wn = atomicAdd(&a.w[j], -p) - p;
        else wn = __hip_atomic_load(&a.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&v[j], wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off, 64);
  if ((tid & 63) == 0) atomicAdd(&a.count[x], (unsigned long long)n);
}
template <int MODE> void run(const char* name, Args a, int n_wg) {
  CK(hipMemset(a.w, 0, DP * 4)); CK(hipMemset(a.view, 0, NX * DP * 4)); CK(hipMemset(a.outbox, 0, NX * DP * 4));
  CK(hipMemset(a.tick, 0, NX * 4)); CK(hipMemset(a.count, 0, 2 * NX * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(n_wg), dim3(NT), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> w(DP), view((size_t)NX * DP), ob((size_t)NX * DP); std::vector<unsigned long long> cnt(2 * NX);
  CK(hipMemcpy(w.data(), a.w, DP * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(view.data(), a.view, NX * DP * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ob.data(), a.outbox, NX * DP * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(cnt.data(), a.count, 2 * NX * 8, hipMemcpyDeviceToHost));
  double total = 0, sw = 0, sob = 0; for (int x = 0; x < NX; ++x) total += (double)cnt[x];
  for (float q : w) sw += q;
  for (float q : ob) sob += q;
  bool ok = true;
  if (MODE == 0) ok = sw == -total;
  if (MODE == 1 || MODE == 2) for (int x = 0; x < NX; ++x) { double s = 0; for (int j = 0; j < DP; ++j) s += view[(size_t)x * DP + j]; ok = ok && s == -(double)cnt[x]; }
  if (MODE == 2) ok = ok && sob == total;
  if (MODE == 3) ok = (sw - sob) == -total;   // what was forwarded plus what is still pending
  printf("%-64s %8.3f ms  %7.2f G atomics-updates/s  %s   workgroups per XCD:", name, ms, total / ms / 1e6, ok ? "sums exact" : "SUMS WRONG");
  for (int x = 0; x < NX; ++x) printf(" %llu", cnt[NX + x]);
  printf("\n");
}
int main(int argc, char** argv) {
  Args a; a.iters = argc > 1 ? atoi(argv[1]) : 200; a.density = 16;
  CK(hipMalloc(&a.w, DP * 4)); CK(hipMalloc(&a.view, NX * DP * 4)); CK(hipMalloc(&a.outbox, NX * DP * 4));
  CK(hipMalloc(&a.tick, NX * 4)); CK(hipMalloc(&a.count, 2 * NX * 8));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("A shared vector, agent-scope atomics (today)", a, 256);
    run<1>("B replica of the XCD, workgroup-scope atomics", a, 256);
    run<2>("C replica + outbox", a, 256);
    run<3>("D replica + outbox + slice merge per iteration", a, 256);
  }
  a.density = 4;
  run<0>("A, one column in 4", a, 256);
  run<2>("C, one column in 4", a, 256);
  run<3>("D, one column in 4", a, 256);
  return 0;
}
