// LDS accumulate-op throughput on MI355X with the real column stream (not part of the product).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
extern "C" {
void* dsgd_synth_create(uint64_t seed, int32_t dim);
void dsgd_synth_destroy(void*);
int64_t dsgd_synth_row_ptr(const void*, int64_t row0, int64_t n_rows, int64_t* row_ptr);
void dsgd_synth_fill(const void*, int64_t row0, int64_t n_rows, const int64_t* row_ptr, int32_t* col, float* val, int8_t* label);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int D = 47236, DP = D + 1;

template <int MODE, int H>
__global__ void __launch_bounds__(1024) k_lds(const int4* col, const float4* val, long long n4, float* g) {
  extern __shared__ __attribute__((aligned(16))) float gl[];
  unsigned int* gu = reinterpret_cast<unsigned int*>(gl);
  unsigned long long* gq = reinterpret_cast<unsigned long long*>(gl);
  for (int j = threadIdx.x; j < H; j += 1024) gl[j] = 0.f;
  __syncthreads();
  float sink = 0.f;
  auto add = [&](int c, float v) {
    const int j = c % H;
    if (MODE == 0) atomicAdd(&gl[j], v);                                   // ds_add_f32
    if (MODE == 1) atomicAdd(&gu[j], (unsigned int)(int)(v * 1048576.f));  // ds_add_u32 (fixed point)
    if (MODE == 2) atomicAdd(&gq[j % (H / 2)], (unsigned long long)(long long)(v * 4294967296.f));  // ds_add_u64
    if (MODE == 3) gl[j] += v;                                             // non-atomic RMW (wrong, rate only)
    if (MODE == 4) gl[j] = v;                                              // ds_write_b32
    if (MODE == 5) sink += atomicAdd(&gl[j], v);                           // ds_add_rtn_f32
    if (MODE == 6) sink += gl[j];                                          // ds_read_b32
    if (MODE == 7) atomicMax(&gu[j], (unsigned int)__float_as_int(v));     // ds_max_u32
  };
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
    int4 c = col[i]; float4 v = val[i];
    add(c.x, v.x); add(c.y, v.y); add(c.z, v.z); add(c.w, v.w);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += 1024) { float v = gl[j]; if (v != 0.f) atomicAdd(&g[j], v); }
  if (sink == 123.456f) g[0] = sink;
}
template <class F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char** argv) {
  long long rows = argc > 1 ? atoll(argv[1]) : 1200000;
  void* gen = dsgd_synth_create(0, D);
  std::vector<int64_t> rp(rows + 1);
  long long nnz = dsgd_synth_row_ptr(gen, 0, rows, rp.data());
  std::vector<int32_t> col(nnz + 4); std::vector<float> val(nnz + 4); std::vector<int8_t> lab(rows);
  dsgd_synth_fill(gen, 0, rows, rp.data(), col.data(), val.data(), lab.data());
  dsgd_synth_destroy(gen);
  std::vector<long long> cnt(DP, 0); for (long long p = 0; p < nnz; ++p) cnt[col[p]]++;
  std::vector<int> order(DP); std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  std::vector<int> rank(DP); for (int r = 0; r < DP; ++r) rank[order[r]] = r;
  std::vector<int32_t> colr(nnz + 4); for (long long p = 0; p < nnz; ++p) colr[p] = rank[col[p]];
  long long n4 = nnz / 4; double gb = 8.0 * 4 * n4 / 1e9;
  int* d_colr; float *d_val, *d_g;
  CK(hipMalloc(&d_colr, 4 * (nnz + 4))); CK(hipMalloc(&d_val, 4 * (nnz + 4))); CK(hipMalloc(&d_g, 4 * DP));
  CK(hipMemcpy(d_colr, colr.data(), 4 * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(d_val, val.data(), 4 * nnz, hipMemcpyHostToDevice)); CK(hipMemset(d_g, 0, 4 * DP));
  auto rep = [&](const char* name, float ms) { printf("%-34s %8.3f ms  %7.1f GB/s(alg)  %6.1f Gnnz/s  %5.2f lanes/clk/CU@2.4GHz\n", name, ms, gb / ms * 1e3, 4.0 * n4 / ms / 1e6, 4.0 * n4 / (ms * 1e-3) / 256 / 2.4e9); fflush(stdout); };
  constexpr int H = 32768; size_t lds = H * 4;
#define RUN(MODE, NAME) do { CK(hipFuncSetAttribute((const void*)k_lds<MODE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_lds<MODE, H>), dim3(256), dim3(1024), lds, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_g); }, 3)); } while (0)
  RUN(0, "ds_add_f32");
  RUN(1, "ds_add_u32 (fixed point)");
  RUN(2, "ds_add_u64 (fixed point)");
  RUN(3, "non-atomic read-add-write");
  RUN(4, "ds_write_b32");
  RUN(5, "ds_add_rtn_f32");
  RUN(6, "ds_read_b32");
  RUN(7, "ds_max_u32");
  return 0;
}
