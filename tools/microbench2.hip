// Where does the scatter time go?  (not part of the product)
//  lds_zipf    : every non-zero -> ds_add_f32 into an LDS tile, frequency-ranked Zipf columns folded with c % H
//  lds_uniform : same with uniformly random columns (bank conflicts only, no hot addresses)
//  lds_hotpriv : columns < 64 go to a per-wave private copy (16 x 64 floats), the rest as lds_zipf
//  cold_only   : only columns >= H go to global atomics, nothing else
//  none        : stream only
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
__global__ void __launch_bounds__(1024) k_scatter(const int4* col, const float4* val, long long n4, float* g) {
  extern __shared__ float gl[];
  float* priv = gl + H;  // 16 waves x 64
  for (int j = threadIdx.x; j < H + 1024; j += 1024) gl[j] = 0.f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  auto add = [&](int c, float v) {
    if (MODE == 0) return;                                                   // none
    if (MODE == 1) { atomicAdd(&gl[c % H], v); return; }                      // lds (zipf or uniform by input)
    if (MODE == 2) { if (c < 64) atomicAdd(&priv[wave * 64 + c], v); else atomicAdd(&gl[c % H], v); return; }
    if (MODE == 3) { if (c >= H) atomicAdd(&g[c], v); return; }               // cold only
    if (MODE == 4) { if (c < H) atomicAdd(&gl[c], v); return; }               // hot only (no cold)
  };
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
    int4 c = col[i]; float4 v = val[i];
    acc += v.x + v.y + v.z + v.w;
    add(c.x, v.x); add(c.y, v.y); add(c.z, v.z); add(c.w, v.w);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += 1024) { float v = gl[j]; if (v != 0.f) atomicAdd(&g[j], v); }
  if (acc == 123.456f) g[0] = acc;
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
  std::vector<int32_t> colr(nnz + 4), colu(nnz + 4);
  uint64_t s = 88172645463325252ull;
  for (long long p = 0; p < nnz; ++p) { colr[p] = rank[col[p]]; s ^= s << 13; s ^= s >> 7; s ^= s << 17; colu[p] = (int)(s % DP); }
  long long n4 = nnz / 4; double gb = 8.0 * 4 * n4 / 1e9;
  printf("rows %lld nnz %lld\n", rows, nnz);
  int *d_colr, *d_colu; float *d_val, *d_g;
  CK(hipMalloc(&d_colr, 4 * (nnz + 4))); CK(hipMalloc(&d_colu, 4 * (nnz + 4))); CK(hipMalloc(&d_val, 4 * (nnz + 4))); CK(hipMalloc(&d_g, 4 * DP));
  CK(hipMemcpy(d_colr, colr.data(), 4 * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(d_colu, colu.data(), 4 * nnz, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), 4 * nnz, hipMemcpyHostToDevice)); CK(hipMemset(d_g, 0, 4 * DP));
  auto rep = [&](const char* name, float ms) { printf("%-40s %8.3f ms  %7.1f GB/s(alg)  %6.1f Gnnz/s\n", name, ms, gb / ms * 1e3, 4.0 * n4 / ms / 1e6); fflush(stdout); };
  constexpr int H = 32768; size_t lds = (H + 1024) * 4;
#define RUN(MODE, COLS, NAME) do { CK(hipFuncSetAttribute((const void*)k_scatter<MODE, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_scatter<MODE, H>), dim3(256), dim3(1024), lds, 0, (const int4*)COLS, (const float4*)d_val, n4, d_g); }, 3)); } while (0)
  RUN(0, d_colr, "none (stream only)");
  RUN(1, d_colr, "lds atomics, zipf ranked cols %H");
  RUN(1, d_colu, "lds atomics, uniform random cols %H");
  RUN(2, d_colr, "lds + per-wave private top-64");
  RUN(4, d_colr, "hot only (c<H lds), cold dropped");
  RUN(3, d_colr, "cold only (c>=H global atomics)");
  long long cold = 0; for (long long p = 0; p < nnz; ++p) cold += colr[p] >= H; printf("cold nnz (rank >= %d): %lld = %.2f%%\n", H, cold, 100.0 * cold / nnz);
  long long top64 = 0; for (long long p = 0; p < nnz; ++p) top64 += colr[p] < 64; printf("top-64 nnz: %.2f%%\n", 100.0 * top64 / nnz);
  return 0;
}
