// Design-space microbenchmarks for the gradient kernel on MI355X (not part of the product):
// how fast can one GPU (a) stream CSR col/val, (b) gather w[col] from L2, (c) scatter-add into g
// with device-scope atomics, (d) with workgroup-scope atomics into XCD-private copies selected by
// the hardware XCC id, (e) with an LDS-privatised hot-column tile.  Uses the same synthetic data.
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

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID

__global__ void __launch_bounds__(256) k_stream(const int4* col, const float4* val, long long n4, float* out) {
  float acc = 0.f; int iacc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int4 c = col[i]; float4 v = val[i];
    acc += v.x + v.y + v.z + v.w; iacc += c.x ^ c.y ^ c.z ^ c.w;
  }
  if (acc == 123.456f && iacc == 77) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather(const int4* col, const float4* val, long long n4, const float* __restrict__ w, float* out) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int4 c = col[i]; float4 v = val[i];
    acc += v.x * w[c.x] + v.y * w[c.y] + v.z * w[c.z] + v.w * w[c.w];
  }
  if (acc == 123.456f) out[0] = acc;
}
// LDS-staged hot tile of w: columns (frequency-ranked ids) < H read from LDS
template <int H>
__global__ void __launch_bounds__(1024) k_gather_lds(const int4* col, const float4* val, long long n4, const float* __restrict__ w, float* out) {
  extern __shared__ float wl[];
  for (int j = threadIdx.x; j < H; j += 1024) wl[j] = w[j];
  __syncthreads();
  float acc = 0.f;
  auto get = [&](int c) { return c < H ? wl[c] : w[c]; };
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
    int4 c = col[i]; float4 v = val[i];
    acc += v.x * get(c.x) + v.y * get(c.y) + v.z * get(c.z) + v.w * get(c.w);
  }
  if (acc == 123.456f) out[0] = acc;
}
__device__ __forceinline__ bool act(long long i, int every) { return every <= 1 || ((unsigned)((i >> 4) * 2654435761u) >> 8) % every == 0; }
__global__ void __launch_bounds__(256) k_scatter_agent(const int4* col, const float4* val, long long n4, float* g, int every) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int4 c = col[i]; float4 v = val[i];
    if (!act(i, every)) continue;
    atomicAdd(&g[c.x], v.x); atomicAdd(&g[c.y], v.y); atomicAdd(&g[c.z], v.z); atomicAdd(&g[c.w], v.w);
  }
}
__device__ __forceinline__ void wg_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__global__ void __launch_bounds__(256) k_scatter_xcd(const int4* col, const float4* val, long long n4, float* gp /* 8 x DP */, int every) {
  float* g = gp + (long long)xcc_id() * DP;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int4 c = col[i]; float4 v = val[i];
    if (!act(i, every)) continue;
    wg_add(&g[c.x], v.x); wg_add(&g[c.y], v.y); wg_add(&g[c.z], v.z); wg_add(&g[c.w], v.w);
  }
}
// agent-scope atomics but into XCD-private copies (separates "scope" from "privatisation")
__global__ void __launch_bounds__(256) k_scatter_xcd_agent(const int4* col, const float4* val, long long n4, float* gp, int every) {
  float* g = gp + (long long)xcc_id() * DP;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    int4 c = col[i]; float4 v = val[i];
    if (!act(i, every)) continue;
    atomicAdd(&g[c.x], v.x); atomicAdd(&g[c.y], v.y); atomicAdd(&g[c.z], v.z); atomicAdd(&g[c.w], v.w);
  }
}
template <int H>
__global__ void __launch_bounds__(1024) k_scatter_lds(const int4* col, const float4* val, long long n4, float* g, int every) {
  extern __shared__ float gl[];
  for (int j = threadIdx.x; j < H; j += 1024) gl[j] = 0.f;
  __syncthreads();
  auto add = [&](int c, float v) { if (c < H) atomicAdd(&gl[c], v); else atomicAdd(&g[c], v); };
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (long long)gridDim.x * 1024) {
    int4 c = col[i]; float4 v = val[i];
    if (!act(i, every)) continue;
    add(c.x, v.x); add(c.y, v.y); add(c.z, v.z); add(c.w, v.w);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += 1024) { float v = gl[j]; if (v != 0.f) atomicAdd(&g[j], v); }
}
__global__ void k_sum8(const float* gp, float* out) {
  int j = blockIdx.x * 256 + threadIdx.x; if (j >= DP) return;
  float a = 0; for (int x = 0; x < 8; ++x) a += gp[(long long)x * DP + j]; out[j] = a;
}
__global__ void k_census(int* cnt) { if (threadIdx.x == 0) atomicAdd(&cnt[xcc_id()], 1); }

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
  // frequency-rank remap: new id = rank of the column by descending count (hot columns get small ids)
  std::vector<long long> cnt(DP, 0); for (long long p = 0; p < nnz; ++p) cnt[col[p]]++;
  std::vector<int> order(DP); std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });
  std::vector<int> rank(DP); for (int r = 0; r < DP; ++r) rank[order[r]] = r;
  std::vector<int32_t> colr(nnz + 4); for (long long p = 0; p < nnz; ++p) colr[p] = rank[col[p]];
  long long n4 = nnz / 4; double gb = 8.0 * 4 * n4 / 1e9;
  printf("rows %lld nnz %lld (%.2f GB col+val)\n", rows, nnz, gb);
  int *d_col, *d_colr; float *d_val, *d_w, *d_g, *d_gp, *d_out, *d_g2;
  CK(hipMalloc(&d_col, 4 * (nnz + 4))); CK(hipMalloc(&d_colr, 4 * (nnz + 4))); CK(hipMalloc(&d_val, 4 * (nnz + 4)));
  CK(hipMalloc(&d_w, 4 * DP)); CK(hipMalloc(&d_g, 4 * DP)); CK(hipMalloc(&d_g2, 4 * DP)); CK(hipMalloc(&d_gp, 4ll * 8 * DP)); CK(hipMalloc(&d_out, 64));
  CK(hipMemcpy(d_col, col.data(), 4 * nnz, hipMemcpyHostToDevice)); CK(hipMemcpy(d_colr, colr.data(), 4 * nnz, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), 4 * nnz, hipMemcpyHostToDevice));
  std::vector<float> w(DP); for (int j = 0; j < DP; ++j) w[j] = 0.001f * (j % 97); CK(hipMemcpy(d_w, w.data(), 4 * DP, hipMemcpyHostToDevice));
  int* d_cnt; CK(hipMalloc(&d_cnt, 64)); CK(hipMemset(d_cnt, 0, 64));
  hipLaunchKernelGGL(k_census, dim3(2048), dim3(256), 0, 0, d_cnt); int hc[16]; CK(hipMemcpy(hc, d_cnt, 64, hipMemcpyDeviceToHost));
  printf("xcc census of 2048 blocks:"); for (int i = 0; i < 8; ++i) printf(" %d", hc[i]); printf("\n");
  auto rep = [&](const char* name, float ms) { printf("%-34s %8.3f ms  %7.1f GB/s(alg 8B/nnz)  %6.1f Gnnz/s\n", name, ms, gb / ms * 1e3, 4.0 * n4 / ms / 1e6); fflush(stdout); };
  for (int blocks : {1024, 2048, 4096}) {
    char nm[64]; snprintf(nm, 64, "stream col+val  grid=%d", blocks);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_out); }));
  }
  rep("gather w[col] (orig ids)", timeit([&] { hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_w, d_out); }));
  rep("gather w[col] (freq-ranked ids)", timeit([&] { hipLaunchKernelGGL(k_gather, dim3(2048), dim3(256), 0, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_w, d_out); }));
  rep("gather LDS hot tile H=16384", timeit([&] { hipLaunchKernelGGL(k_gather_lds<16384>, dim3(512), dim3(1024), 16384 * 4, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_w, d_out); }));
  rep("gather LDS hot tile H=32768", timeit([&] { hipLaunchKernelGGL(k_gather_lds<32768>, dim3(256), dim3(1024), 32768 * 4, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_w, d_out); }));
  for (int every : {1, 4, 10}) {
    char nm[96];
    CK(hipMemset(d_g, 0, 4 * DP));
    snprintf(nm, 96, "scatter agent atomics  1/%d active", every);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_scatter_agent, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_g, every); }, 3));
    CK(hipMemset(d_gp, 0, 4ll * 8 * DP));
    snprintf(nm, 96, "scatter XCD-private wg-scope 1/%d", every);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_scatter_xcd, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_gp, every); }, 3));
    CK(hipMemset(d_gp, 0, 4ll * 8 * DP));
    snprintf(nm, 96, "scatter XCD-private agent    1/%d", every);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_scatter_xcd_agent, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_gp, every); }, 3));
    snprintf(nm, 96, "scatter LDS hot H=16384      1/%d", every);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_scatter_lds<16384>, dim3(512), dim3(1024), 16384 * 4, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_g2, every); }, 3));
    snprintf(nm, 96, "scatter LDS hot H=32768      1/%d", every);
    rep(nm, timeit([&] { hipLaunchKernelGGL(k_scatter_lds<32768>, dim3(256), dim3(1024), 32768 * 4, 0, (const int4*)d_colr, (const float4*)d_val, n4, d_g2, every); }, 3));
  }
  // correctness of the XCD-private workgroup-scope variant: one pass each, compare sums
  CK(hipMemset(d_g, 0, 4 * DP)); CK(hipMemset(d_gp, 0, 4ll * 8 * DP));
  hipLaunchKernelGGL(k_scatter_agent, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_g, 1);
  hipLaunchKernelGGL(k_scatter_xcd, dim3(2048), dim3(256), 0, 0, (const int4*)d_col, (const float4*)d_val, n4, d_gp, 1);
  hipLaunchKernelGGL(k_sum8, dim3((DP + 255) / 256), dim3(256), 0, 0, d_gp, d_g2);
  std::vector<float> ga(DP), gx(DP); CK(hipMemcpy(ga.data(), d_g, 4 * DP, hipMemcpyDeviceToHost)); CK(hipMemcpy(gx.data(), d_g2, 4 * DP, hipMemcpyDeviceToHost));
  double maxrel = 0; for (int j = 0; j < DP; ++j) { double d = fabs((double)ga[j] - gx[j]) / std::max(1.0, fabs((double)ga[j])); maxrel = std::max(maxrel, d); }
  printf("XCD-private wg-scope vs agent: max rel diff %.3e (%s)\n", maxrel, maxrel < 1e-3 ? "OK" : "MISMATCH");
  return 0;
}
