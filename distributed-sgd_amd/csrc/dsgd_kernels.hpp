// Device code of libdsgd_hip (gfx950 / CDNA4 only: wave64, 160 KiB LDS per CU, 8 XCDs x 32 CUs).
//
// Citations "ref:" are relative to /root/reference/src/main/scala/epfl/distributed/.
//
// Column space: inside the library every dense vector (w, g, ds) and the CSR column array use
// FREQUENCY-RANKED column ids (rank 0 = the most frequent feature).  RCV1-like data is Zipfian,
// so the first H ranks cover most non-zeros; a workgroup stages those H weights in LDS and
// accumulates those H gradient coordinates in LDS instead of going to L2 for every non-zero.
// Measured on MI355X (tools/microbench*.hip, 90 M non-zeros): scattered device-scope fp32 atomics top
// out at ~5-14 G/s, an LDS-staged gather runs at 667 Gnnz/s vs 216-349 Gnnz/s through L1/L2, and
// ds_add_u32 at the streaming rate where ds_add_f32 manages 188 Gnnz/s (hence fixed point).  The
// API (include/dsgd.h) speaks original keys; dsgd_hip.hip permutes at the boundary.
//
// Kernel families, in file order: finish kernels (regularise, sum, apply), prediction / evaluation by row, layout
// kernels, fixed-point helpers and the fused reduce + update, the wave-tile streaming kernel of the hot columns, the
// split-matrix layout kernels and the two cold-stream kernels.  The mini-batch engine (index lists, Hogwild) lives in
// dsgd_batch.hpp.  DESIGN.md section 3 describes each with its roofline.
#pragma once

#include <hip/hip_runtime.h>

#define DSGD_EPS 1e-20f  // ref: math/Sparse.scala:104 (Sparse.epsilon); representable in fp32
#define DSGD_LDS_FLOATS 40960  // 160 KiB / 4

// Sparse(...) constructor filter: entries with abs(v) <= 1e-20 vanish (ref: math/Sparse.scala:108-118)
__device__ __forceinline__ float filt(float v) { return fabsf(v) > DSGD_EPS ? v : 0.0f; }

// Workgroup-wide bulk moves between global memory and LDS tiles in 16-byte pieces (the 4-byte loops they replace were
// 18-28 dependent iterations per thread in the prologue / epilogue of every persistent workgroup: the cold-stream
// kernels stage or clear 115 KB each and run for only ~70 us).  Both pointers 16-byte aligned when n >= 4 is used
// with aligned bases; the tail (n % 4) goes element-wise.
__device__ __forceinline__ void wg_copy_in(float* lds_dst, const float* __restrict__ src, int n, int tid, int nthreads,
                                           bool aligned) {
  // eight 16-byte requests in flight per lane (clamped, unconditional loads; predicated LDS writes): the one-load-per-
  // iteration loop this replaces waited out a full memory round trip seven times in a row for the 115 KB of the cold
  // dot kernel -- a fifth of a 50 us kernel
  const int n4 = aligned ? n >> 2 : 0;
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(lds_dst);
  for (int j0 = 0; j0 < n4; j0 += 8 * nthreads) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * nthreads + tid;
      t[u] = s4[j < n4 ? j : n4 - 1];
    }
    // (the values are "used" here: otherwise the compiler sinks every load into the predicated block of its store and
    //  the eight round trips run one after the other again)
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(t[u].x), "+v"(t[u].y), "+v"(t[u].z), "+v"(t[u].w));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u * nthreads + tid;
      if (j < n4) d4[j] = t[u];
    }
  }
  for (int j = 4 * n4 + tid; j < n; j += nthreads) lds_dst[j] = src[j];
}
// The same copy in two halves (round 6, the row-chunk kernel): the requests go out, the caller issues OTHER requests behind
// them (the first tiles of the phase: registers only), then the values are written to LDS -- vmcnt retires in order, so
// the wait in front of the LDS writes covers the copy's own requests and nothing issued behind them.  With lds_barrier()
// (below) instead of __syncthreads() the phase's first tiles stay in flight across the whole set-up.
struct WgCopy8 {
  float4 t[8];
};
__device__ __forceinline__ void wg_copy_in_issue(const float* __restrict__ src, int n, int tid, int nthreads, bool aligned,
                                                 WgCopy8& r) {
  const int n4 = aligned ? n >> 2 : 0;
  const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = u * nthreads + tid;
    r.t[u] = s4[j < n4 ? j : (n4 > 0 ? n4 - 1 : 0)];
  }
}
__device__ __forceinline__ void wg_copy_in_store(float* lds_dst, const float* __restrict__ src, int n, int tid, int nthreads,
                                                 bool aligned, WgCopy8& r) {
  const int n4 = aligned ? n >> 2 : 0;
  float4* d4 = reinterpret_cast<float4*>(lds_dst);
#pragma unroll
  for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(r.t[u].x), "+v"(r.t[u].y), "+v"(r.t[u].z), "+v"(r.t[u].w));
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = u * nthreads + tid;
    if (j < n4) d4[j] = r.t[u];
  }
  // (what eight pieces per lane do not cover -- models wider than RCV1 -- and the unaligned tail: plain loops)
  const float4* s4 = reinterpret_cast<const float4*>(src);
  for (int j = 8 * nthreads + tid; j < n4; j += nthreads) d4[j] = s4[j];
  for (int j = 4 * n4 + tid; j < n; j += nthreads) lds_dst[j] = src[j];
}
// a workgroup barrier that orders LDS only.  (Neither this nor __syncthreads() waits for vector memory operations with
// this compiler -- both are `s_waitcnt lgkmcnt(0); s_barrier` in the ISA, a workgroup lives on one CU -- but
// __syncthreads() is also a compiler fence for global accesses; requests issued in front of either stay in flight.)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void wg_zero(int* lds_dst, int n, int tid, int nthreads) {   // lds_dst 16-byte aligned
  const int n4 = n >> 2;
  for (int j = tid; j < n4; j += nthreads) reinterpret_cast<int4*>(lds_dst)[j] = make_int4(0, 0, 0, 0);
  for (int j = 4 * n4 + tid; j < n; j += nthreads) lds_dst[j] = 0;
}
__device__ __forceinline__ void wg_copy_out(int* __restrict__ dst, const int* lds_src, int n, int tid, int nthreads,
                                            bool aligned) {
  const int n4 = aligned ? n >> 2 : 0;
  for (int j = tid; j < n4; j += nthreads) reinterpret_cast<int4*>(dst)[j] = reinterpret_cast<const int4*>(lds_src)[j];
  for (int j = 4 * n4 + tid; j < n; j += nthreads) dst[j] = lds_src[j];
}
__device__ __forceinline__ bool is_aligned16(const void* p) { return (reinterpret_cast<unsigned long long>(p) & 15ull) == 0; }

// device scalars shared by the kernels of one context
struct DevScalars {
  float s_reg;   // 2 * lambda * (w . ds)            (ref: core/ml/SparseSVM.scala:31)
  float wnorm2;  // |w|^2                            (ref: math/Vec.scala:55)
  int err;       // != 0: a sample index / key was out of range
  int pad;
  unsigned long long n_active;   // rows with y*(x.w) >= 0
  unsigned long long n_samples;  // rows processed
  unsigned long long counts[4];  // eval tallies {pred==y, pred==0, pred==-y, rows}
  // multi-workgroup apply: per-workgroup partial sums of w.ds and |w|^2, combined in a fixed order by the
  // last workgroup to arrive (ticket)
  float part_dot[64];
  float part_nsq[64];
  unsigned int ticket;
  unsigned int pad2;
};

// one unit of gradient work: worker k processes items [begin, end) -- positions in the resident
// index list (idx != nullptr) or CSR row numbers themselves (contiguous range)
struct WorkSeg {
  long long begin;
  long long end;
};

struct CsrView {
  long long n_rows;
  const long long* __restrict__ row_ptr;
  const int* __restrict__ col;  // frequency-ranked ids
  const float* __restrict__ val;
  const signed char* __restrict__ label;
};

// v + (v of the lane selected by a DPP control), no LDS crossbar round trip
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sum over the G lanes of a group, result on every lane.  Every step pairs lanes through an
// involution (quad xor 1, quad xor 2, mirror within 8, mirror within 16, xor 16, xor 32), so all
// lanes of the group hold the bitwise-identical sum (they must agree on the gate) and the order
// is fixed: x.w is reproducible run to run.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group width");
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  if (G >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// weight lookup: ranks < hw come from the LDS tile, the tail from L1/L2
// (two separate loads and a select of VALUES: a select of pointers would become one flat_load)
__device__ __forceinline__ float w_at(const float* wl, const float* __restrict__ w, int c, int hw) {
  // the volatile qualifier keeps the LDS read a ds_read_b32 (LLVM otherwise folds the two loads
  // into select(ptr) + flat_load_dword, which is slower than either address space's own path)
  const bool hot = c < hw;
  typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
  float a = ((lds_cvfloat*)wl)[hot ? c : 0];
  if (!hot) a = w[c];
  return a;
}

// A row held by a group of G lanes: the first UNR*G non-zeros stay in registers between the
// dot product and the scatter (no second trip to memory for ~half of the rows).
template <int G, int UNR>
struct RowRegs {
  int c[UNR];
  float v[UNR];
};

// x_row . w ; ref: math/Vec.scala:58 -> math/Sparse.scala:46,20-31 (products filtered at 1e-20)
template <int G, int UNR, bool LDSW>
__device__ __forceinline__ float row_dot(const CsrView& m, long long start, long long end, const float* wl,
                                         const float* __restrict__ w, int hw, int sub, RowRegs<G, UNR>& r) {
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    const long long p = start + sub + k * G;
    const bool in = p < end;
    r.c[k] = in ? m.col[p] : -1;
    r.v[k] = in ? m.val[p] : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < UNR; ++k) {
    if (r.c[k] >= 0) {
      const float wv = LDSW ? w_at(wl, w, r.c[k], hw) : w[r.c[k]];
      acc += filt(r.v[k] * wv);
    }
  }
  for (long long p = start + sub + UNR * G; p < end; p += G) {
    const int c = m.col[p];
    const float wv = LDSW ? w_at(wl, w, c, hw) : w[c];
    acc += filt(m.val[p] * wv);
  }
  return group_sum<G>(acc);
}

// ---- K2: support-only scalar regulariser ------------------------------------------------------------------
// g_k[j] += s for j in supp(g_k), s = 2*lambda*(w.ds)  (ref: SparseSVM.scala:31, math/Vec.scala:65-75)
__global__ void __launch_bounds__(1024) dsgd_regularize_kernel(float* g_base, long long g_stride, int dp,
                                                              const DevScalars* sc) {
  float* g = g_base + (long long)blockIdx.y * g_stride;
  const float s = sc->s_reg;
  const bool add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < dp; j += gridDim.x * blockDim.x) {
    float v = filt(g[j]);
    if (add && v != 0.0f) v = filt(v + s);
    g[j] = v;
  }
}

__device__ __forceinline__ float block_sum_1024(float v, float* red /* 16 floats of LDS */) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float t = 0.0f;
  const int n_waves = blockDim.x >> 6;
  for (int i = 0; i < n_waves; ++i) t += red[i];
  return t;
}

// ---- K3: mean over workers + update + next regulariser scalar ---------------------------------------------
// w <- w - lr * (g_sum / K); g <- 0; s <- 2*lambda*(w.ds); |w|^2
// ref: core/Master.scala:194-197 (Vec.mean then batchWeights - learningRate * grad).
// dsgd_apply_cols_kernel (below, next to the fused reduction whose tail it shares).


// ---- async iteration (host-driven form of Slave.asyncTask) -------------------------------------------------
// grad = g_sum / n; delta = lr * regularize(grad, w); w -= delta  (ref: core/Slave.scala:93-101)
__global__ void __launch_bounds__(1024) dsgd_async_finish_kernel(float* __restrict__ w, float* __restrict__ g, int dp,
                                                                const float* __restrict__ ds, float n_samples, float lr,
                                                                float lambda, float* delta_out, DevScalars* sc) {
  __shared__ float red[16];
  const float s = sc->s_reg;
  const bool add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
  float dot = 0.0f, nsq = 0.0f;
  for (int j = threadIdx.x; j < dp; j += blockDim.x) {
    float v = filt(filt(g[j]) / n_samples);  // Vec.mean over samples
    if (add && v != 0.0f) v = filt(v + s);   // regularize on the support
    const float upd = filt(v * lr);
    if (delta_out) delta_out[j] = upd;
    const float wn = filt(w[j] - upd);
    w[j] = wn;
    g[j] = 0.0f;
    dot += filt(wn * ds[j]);
    nsq += wn * wn;
  }
  const float dsum = block_sum_1024(dot, red);
  const float nsum = block_sum_1024(nsq, red);
  if (threadIdx.x == 0) {
    sc->s_reg = lambda * 2.0f * dsum;
    sc->wnorm2 = nsum;
  }
}

// w[perm[key]] -= dv (ref: core/Slave.scala:177-185, core/MasterAsync.scala:164-177, core/ml/GradState.scala:8).  Keys are
// validated on the host.  Atomic adds: the update may arrive WHILE the lock-free engine is adding to the same weights
// (the reference calls updateGrad concurrently with asyncTask by design).  s_reg != nullptr (engine running): the
// change of the engine's incrementally kept regulariser scalar, -2 lambda sum(dv_j ds_j), is folded in with one
// atomic per block.  256-lane blocks of a few registers: they become resident beside the engine's workgroups.
__global__ void __launch_bounds__(256) dsgd_update_grad_kernel(float* w, const int* __restrict__ perm,
                                                              const float* __restrict__ ds, const int* __restrict__ key,
                                                              const float* __restrict__ dv, long long nnz, float lambda,
                                                              float* s_reg) {
  __shared__ float red[4];
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
    const float d = dv[i];
    if (d == 0.0f) continue;
    const int r = perm[key[i]];
    atomicAdd(&w[r], -d);
    acc += d * ds[r];
  }
  if (!s_reg) return;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    if (tot != 0.0f) atomicAdd(s_reg, -2.0f * lambda * tot);
  }
}
__global__ void dsgd_filter_kernel(float* w, int dp) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < dp; j += gridDim.x * blockDim.x) w[j] = filt(w[j]);
}

// ---- K4: prediction  p = -signum(x.w)  (ref: core/ml/SparseSVM.scala:14, core/Slave.scala:129-140) ---------
template <int G>
__global__ void __launch_bounds__(256) dsgd_forward_kernel(CsrView m, const float* __restrict__ w,
                                                          const int* __restrict__ idx, long long n, float* pred,
                                                          DevScalars* sc) {
  constexpr int UNR = 4;
  const int sub = threadIdx.x % G;
  const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long long n_groups = (long long)gridDim.x * blockDim.x / G;
  for (long long t = group; t < n; t += n_groups) {
    const long long row = idx[t];
    if (row < 0 || row >= m.n_rows) {
      if (sub == 0) atomicOr(&sc->err, 1);
      continue;
    }
    RowRegs<G, UNR> r;
    const float d = row_dot<G, UNR, false>(m, m.row_ptr[row], m.row_ptr[row + 1], nullptr, w, 0, sub, r);
    if (sub == 0) pred[t] = d > 0.0f ? -1.0f : (d < 0.0f ? 1.0f : 0.0f);
  }
}

// the three evaluation tallies of a workgroup: ONE global atomic per counter and workgroup (one per wave puts thousands
// of atomics on three addresses when the grid finishes; they are served one after the other)
// (tally: three words of LDS the caller owns -- a static __shared__ next to a 160 KiB dynamic allocation is refused)
__device__ __forceinline__ void block_tally3(unsigned int c0, unsigned int c1, unsigned int c2, DevScalars* sc,
                                             unsigned int* tally) {
  __syncthreads();   // (everybody is done with whatever the words held)
  if (threadIdx.x < 3) tally[threadIdx.x] = 0u;
  __syncthreads();
  c0 = wave_sum_u32(c0);
  c1 = wave_sum_u32(c1);
  c2 = wave_sum_u32(c2);
  if ((threadIdx.x & 63) == 0) {
    if (c0) atomicAdd(&tally[0], c0);
    if (c1) atomicAdd(&tally[1], c1);
    if (c2) atomicAdd(&tally[2], c2);
  }
  __syncthreads();
  if (threadIdx.x < 3 && tally[threadIdx.x]) atomicAdd(&sc->counts[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// ---- K5: loss / accuracy tallies over a row range -------------------------------------------------------------
// ref: core/Master.scala:100-107, core/ml/SparseSVM.scala:16-23: with p = -signum(x.w),
//   y*p = +1 (loss 0, correct) iff y*(x.w) < 0;  p = 0 (loss 1) iff x.w == 0;  y*p = -1 (loss 2) otherwise
// Persistent 1024-lane workgroups with the hw hottest weights staged in LDS (hw up to 40960).
template <int G>
__global__ void __launch_bounds__(1024) dsgd_eval_kernel(CsrView m, const float* __restrict__ w, long long row_begin,
                                                        long long row_end, DevScalars* sc, int hw) {
  constexpr int UNR = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;
  for (int j = threadIdx.x; j < hw; j += blockDim.x) wl[j] = w[j];
  __syncthreads();
  const int sub = threadIdx.x % G;
  const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long long n_groups = (long long)gridDim.x * blockDim.x / G;   // (256-lane blocks beside the Hogwild engine, 1024 otherwise)
  unsigned int c0 = 0, c1 = 0, c2 = 0;
  for (long long row = row_begin + group; row < row_end; row += n_groups) {
    RowRegs<G, UNR> r;
    const float d = row_dot<G, UNR, true>(m, m.row_ptr[row], m.row_ptr[row + 1], wl, w, hw, sub, r);
    const float yd = (float)m.label[row] * d;
    if (sub == 0) {
      if (yd < 0.0f) c0++;
      else if (yd > 0.0f) c2++;
      else c1++;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&sc->counts[3], (unsigned long long)(row_end - row_begin));
  block_tally3(c0, c1, c2, sc, reinterpret_cast<unsigned int*>(wl + hw));   // (the launch allocates hw + 4 floats)
}

// ---- layout: column frequencies, ranking, permutations ---------------------------------------------------------
// histogram of column ids with an LDS-privatised counter tile for ids < hcnt (same reasoning as K1b)
// Entries with abs(value) <= 1e-20 are not counted: the reference's Sparse constructor drops them
// before any key set is taken (math/Sparse.scala:108-118), and the padding element of an empty row
// (value 0) must stay invisible.
__global__ void __launch_bounds__(1024) dsgd_colcount_kernel(const int* __restrict__ col, const float* __restrict__ val,
                                                            long long nnz, unsigned int* cnt, int dp, int hcnt,
                                                            DevScalars* sc) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lcnt[];
  for (int j = threadIdx.x; j < hcnt; j += 1024) lcnt[j] = 0u;
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 1024 + threadIdx.x; p < nnz; p += (long long)gridDim.x * 1024) {
    const int c = col[p];
    if (!(fabsf(val[p]) > DSGD_EPS)) continue;
    if (c < 0 || c >= dp) {
      atomicOr(&sc->err, 1);
      continue;
    }
    if (c < hcnt) atomicAdd(&lcnt[c], 1u);
    else atomicAdd(&cnt[c], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < hcnt; j += 1024) {
    const unsigned int v = lcnt[j];
    if (v) atomicAdd(&cnt[j], v);
  }
}

// col[p] <- perm[col[p]]
__global__ void dsgd_remap_cols_kernel(int* col, long long nnz, const int* __restrict__ perm) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (long long)gridDim.x * blockDim.x)
    col[p] = perm[col[p]];
}

// out[perm[j]] = filt(in[j])  (external key order -> internal ranked order)
__global__ void dsgd_permute_in_kernel(const float* __restrict__ in, float* out, const int* __restrict__ perm, int dp) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < dp; j += gridDim.x * blockDim.x) out[perm[j]] = filt(in[j]);
}
// out[j] = in[perm[j]]  (internal ranked order -> external key order)
__global__ void dsgd_permute_out_kernel(const float* __restrict__ in, float* out, const int* __restrict__ perm, int dp) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < dp; j += gridDim.x * blockDim.x) out[j] = in[perm[j]];
}

// dimSparsity (ref: Main.scala:54-65): buff(idx - 1) += 1 over the train rows, then
// ds[i] = 1 / (buff(i) + 1) for 0-based key i where buff(i) != 0 -- i.e. key i carries the count of
// feature id i+1.  cnt is indexed by ranked id; ds is written in ranked order of KEY i.
__global__ void dsgd_ds_kernel(const unsigned int* __restrict__ cnt, const int* __restrict__ perm, float* ds, int dp) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dp; i += gridDim.x * blockDim.x) {
    const unsigned int c = (i < dp - 1) ? cnt[perm[i + 1]] : 0u;  // buff has D entries: keys 0..D-1
    ds[perm[i]] = c ? filt(1.0f / ((float)c + 1.0f)) : 0.0f;
  }
}

// ======================================================================================================
// K1c / K5b: nnz-STREAMING kernels for contiguous row ranges -- shared pieces
// ======================================================================================================
// The row-per-group kernels above chain two dependent memory round trips per row (row_ptr -> col/val) and top
// out near 1.2 TB/s on MI355X.  The streaming kernels decouple the HBM stream from the row structure: rows are
// cut at load time into tiles whose addresses are known without row_ptr, so tiles are prefetched while earlier
// ones are processed.  (Two earlier generations -- LDS-staged products, then workgroup tiles with a register
// segmented scan -- are in the git history; profiles/README.md keeps their measurements.)

// Fixed-point gradient accumulation.  Measured on MI355X (tools/microbench3.hip): ds_add_f32 retires
// 0.31 lanes/clk/CU (188 Gnnz/s chip-wide) while ds_add_u32 runs at the HBM streaming rate
// (671 Gnnz/s).  The scatter therefore accumulates round(y*x * 2^21 / vmax2) as 32-bit integers in
// LDS (vmax2 = max|x| rounded up to a power of two, so the scaling is exact; overflow control of the
// 32-bit accumulators: see w_scatter).  Integer addition is
// associative: the gradient of a whole-shard batch is bit-reproducible run to run, and exact up to
// the 2^-22 * vmax2 rounding of each contribution (fp32 atomics round at the ulp of the RUNNING sum).
constexpr int FIX_SHIFT = 21;

struct StreamSeg {
  long long row_begin, row_end;    // rows of this worker's batch
  long long tile_begin, tile_end;  // tiles intersecting [row_begin, row_end)
  long long long_begin, long_end;  // wave-tile mode: range of the long-row list (rows that fit no tile)
  long long ctile_begin, ctile_end;  // cold-stream tiles intersecting [row_begin, row_end)
};

// the same with per-workgroup partial sums: worker k owns workgroups [k * n_wg, (k + 1) * n_wg) of `part` (main
// kernel, columns [0, hg)) and [k * n_wgc, (k + 1) * n_wgc) of `partc` (dsgd_cgrad_kernel, columns [hc, hc + nc));
// g[j] += (float)((g64[j] + sum of the partials of column j) * inv_scale), g64[j] = 0.  Fixed order, no atomics.
// Columns >= hc carry the cold scale (inv_scale_cold); pass hc = dp when there is no such split.
// Block = 64 columns x 16 workgroup phases.
__global__ void __launch_bounds__(1024) dsgd_fix_reduce_kernel(long long* g64_base, float* g_base, long long g_stride,
                                                              int dp, int hg, const int* __restrict__ part,
                                                              int part_stride, int n_wg, int hc, int nc,
                                                              const int* __restrict__ partc, int partc_stride,
                                                              int n_wgc, double inv_scale, double inv_scale_cold) {
  __shared__ long long red[16][64];
  long long* g64 = g64_base + (long long)blockIdx.y * g_stride;
  float* g = g_base + (long long)blockIdx.y * g_stride;
  const int cx = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cx;
  long long q = 0;
  if (j < hg) {
    const int* p = part + (long long)blockIdx.y * n_wg * part_stride + j;
    for (int b = ph; b < n_wg; b += 16) q += (long long)p[(long long)b * part_stride];
  } else if (j >= hc && j < hc + nc) {
    const int* p = partc + (long long)blockIdx.y * n_wgc * partc_stride + (j - hc);
    for (int b = ph; b < n_wgc; b += 16) q += (long long)p[(long long)b * partc_stride];
  }
  red[ph][cx] = q;
  __syncthreads();
  if (ph == 0 && j < dp) {
    long long tot = g64[j];
    if (tot != 0) g64[j] = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += red[k][cx];
    if (tot != 0) g[j] += (float)((double)tot * (j >= hc ? inv_scale_cold : inv_scale));
  }
}

constexpr int FRA_COLS = 192;
__device__ __forceinline__ void fra_scalars(float dot, float nsq, float lambda, DevScalars* sc,
                                            float* __restrict__ redpart, float* fred, int* is_last, bool finalize);
// the per-block pairs (w . ds, |w|^2) added in block order by the first wave of a workgroup: lanes take blocks
// lane, lane + 64, ..., then a butterfly -- THE summation order of s and |w|^2, whoever runs it
__device__ __forceinline__ void fra_sum_pairs(const float* __restrict__ redpart, unsigned int n_blocks, int lane, float& d,
                                              float& qq) {
  d = 0.0f;
  qq = 0.0f;
  for (unsigned int b0 = 0; b0 < n_blocks; b0 += 256) {   // (D + 1 = 47,237: 247 blocks, one round)
    // eight requests in flight, then the adds in block order (a loop of load + add pays one round trip per iteration)
    float x[4], y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int b = b0 + 64u * i + lane;
      const unsigned int bc = b < n_blocks ? b : n_blocks - 1;
      // (agent-scope loads: served past this CU's L1)
      x[i] = __hip_atomic_load(&redpart[2 * bc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      y[i] = __hip_atomic_load(&redpart[2 * bc + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int b = b0 + 64u * i + lane;
      if (b < n_blocks) {
        d += x[i];
        qq += y[i];
      }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    d += __shfl_xor(d, off, 64);
    qq += __shfl_xor(qq, off, 64);
  }
}
// Tail shared by dsgd_fix_reduce_apply_kernel<true> and dsgd_apply_cols_kernel: lane tid < FRA_COLS of a block owns
// column j = block * FRA_COLS + tid and holds the sum of the regularised gradients over the workers; mean, update, and
// the block's share of w . ds and |w|^2 (combined by the last block to arrive, in block order: reproducible).
// (wj, dsj: w[j] and ds[j], requested by the caller when the kernel starts -- behind the last workgroup barrier their
//  round trip would be paid once more; the step's tail is a chain of dependent round trips as it is)
__device__ __forceinline__ void fra_update_and_scalars(float gsum, float k_total, int j, int dp, float* __restrict__ w,
                                                       float wj, float dsj, float lr, float lambda,
                                                       DevScalars* sc, float* __restrict__ redpart, float* fred,
                                                       int* is_last, bool finalize) {
  const int tid = threadIdx.x;
  float dot = 0.0f, nsq = 0.0f;
  if (tid < FRA_COLS && j < dp) {
    const float upd = filt(filt(gsum / k_total) * lr);   // Vec.mean over the workers, then learningRate * grad
    const float wn = filt(wj - upd);
    w[j] = wn;
    dot = filt(wn * dsj);
    nsq = wn * wn;
  }
  fra_scalars(dot, nsq, lambda, sc, redpart, fred, is_last, finalize);
}

// s = 2 * lambda * (w . ds) and |w|^2 from the per-column terms of a block (lanes tid < FRA_COLS; zeros elsewhere):
// wave sums, the block's pair published, and -- `finalize` -- the last block to arrive adds the pairs in block order
// (fra_sum_pairs).  ONE summation order for every kernel that leaves these scalars (the fused step, the update behind an
// all-reduce, dsgd_wstats_cols_kernel after dsgd_set_weights): equal weights give bit-equal scalars whichever path
// wrote them.
// finalize = false (round 3, the steps of the synchronous path): the pairs are left in `redpart` and whoever needs
// the scalars next adds them itself -- the next step's fused kernel in every block's first wave, anything else through
// dsgd_scalars_finalize_kernel.  The finalising tail is a chain of dependent round trips at the end of every step:
// store acknowledgement -> 247 returning atomics on ONE address (served one after the other) -> the last block's
// loads -> its store; 3-4 of the kernel's 9-10 us on a small batch.
__device__ __forceinline__ void fra_scalars(float dot, float nsq, float lambda, DevScalars* sc,
                                            float* __restrict__ redpart, float* fred, int* is_last, bool finalize) {
  const int tid = threadIdx.x;
  if (tid < 256) {   // the finishing waves (FRA_COLS = 192 -> three of them carry data)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      dot += __shfl_xor(dot, off, 64);
      nsq += __shfl_xor(nsq, off, 64);
    }
    if ((tid & 63) == 0) {
      fred[tid >> 6] = dot;
      fred[4 + (tid >> 6)] = nsq;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const float d = (fred[0] + fred[1]) + (fred[2] + fred[3]);
    const float qq = (fred[4] + fred[5]) + (fred[6] + fred[7]);
    __hip_atomic_store(&redpart[2 * blockIdx.x], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&redpart[2 * blockIdx.x + 1], qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (finalize) {
      // the two partials are write-through stores: once they are acknowledged the ticket may be taken (a full
      // __threadfence() here writes back the L2's dirty lines -- this block's weights -- ~3.5 us per block)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int t = atomicAdd(&sc->ticket, 1u);
      *is_last = (t == gridDim.x - 1);
    }
  }
  if (!finalize) return;
  __syncthreads();
  if (*is_last && tid < 64) {
    float d, qq;
    fra_sum_pairs(redpart, gridDim.x, tid, d, qq);
    if (tid == 0) {
      sc->s_reg = lambda * 2.0f * d;
      sc->wnorm2 = qq;
      sc->ticket = 0;
    }
  }
}

// the scalars from the pairs a finalize = false kernel left behind (one wave; dsgd_hip.hip: ensure_s)
__global__ void __launch_bounds__(64) dsgd_scalars_finalize_kernel(const float* __restrict__ redpart, unsigned int n_blocks,
                                                                  float lambda, DevScalars* sc) {
  float d, qq;
  fra_sum_pairs(redpart, n_blocks, threadIdx.x, d, qq);
  if (threadIdx.x == 0) {
    sc->s_reg = lambda * 2.0f * d;
    sc->wnorm2 = qq;
  }
}


// The same reduction fused with K2 + K3 for the workers hosted by ONE context (the benchmark's whole-shard step, every
// small / mid-size batch): per column the exact sum of every worker's partials becomes g_k[j] in a register, gets the
// support-only regulariser (ref: core/ml/SparseSVM.scala:31), the sums are folded over the workers (Vec.sum), divided
// by their number (Vec.mean) and applied (ref: core/Master.scala:194-197) -- g is never written.  The two dot products
// of the new weights are combined by the last block to arrive, in block order (reproducible).  Every block reads the
// old s before it takes its ticket, so the last block's write of the new s cannot be seen by any of them.
// APPLY = false (a communicator is attached): the regularised sum over the hosted workers is written to `gsum_out` for
// the all-reduce across ranks, dsgd_apply_cols_kernel<false> finishes (two launches around the collective instead of
// four).
// Geometry: a block owns FRA_COLS = 192 columns (247 blocks for D + 1 = 47,237: one round over 256 CUs); a thread adds
// FOUR adjacent columns of every 21st workgroup's partials with 16-byte loads (768-byte pieces per partial row; the
// first form -- 64 columns per block, 4-byte loads, 256-byte pieces -- moved the 48 MB of partials of a whole-shard
// step at 1.7 TB/s).
constexpr int FRA_GROUPS = FRA_COLS / 4;          // 48 column groups
constexpr int FRA_PHASES = 1024 / FRA_GROUPS;     // 21 phases (1008 of the 1024 lanes add partials)
template <bool APPLY>
__global__ void __launch_bounds__(1024) dsgd_fix_reduce_apply_kernel(long long* __restrict__ g64_base, long long g_stride,
                                                                    int n_workers, float* __restrict__ w,
                                                                    const float* __restrict__ ds, int dp, int hg,
                                                                    const int* __restrict__ part, int part_stride,
                                                                    int n_wg, int hc, int nc,
                                                                    const int* __restrict__ partc, int partc_stride,
                                                                    int n_wgc, double inv_scale, double inv_scale_cold,
                                                                    float lr, float lambda, DevScalars* sc,
                                                                    float* __restrict__ redpart,
                                                                    float* __restrict__ gsum_out,
                                                                    const float* __restrict__ redpart_in, int s_lazy,
                                                                    unsigned long long* __restrict__ mail) {
  __shared__ __attribute__((aligned(16))) long long red[FRA_PHASES][FRA_COLS];   // 32 KB
  __shared__ float fred[8];
  __shared__ int is_last;
  __shared__ float s_sh;
  const int tid = threadIdx.x;
  // per-request steps (dsgd_sync_step): the gradient kernel's active-row count and the error flags go to a host-mapped
  // mailbox from HERE -- the last launch of the request -- instead of through a device-to-host copy behind it
  if (mail && blockIdx.x == 0 && tid == 0) {
    __hip_atomic_store(&mail[0], sc->n_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mail[1], (unsigned long long)(unsigned int)sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // s of the weights this step starts from: final in DevScalars, or (s_lazy) still the previous step's per-block pairs
  // (redpart_in: the OTHER half of the pair buffer -- a fast block of this launch publishes its new pair while a slow
  // one still reads the old ones)
  if (s_lazy) {
    if (tid < 64) {
      float d, qq;
      fra_sum_pairs(redpart_in, gridDim.x, tid, d, qq);
      if (tid == 0) s_sh = lambda * 2.0f * d;
    }
  } else if (tid == 0) {
    s_sh = sc->s_reg;
  }
  const int cg = tid % FRA_GROUPS, ph = tid / FRA_GROUPS;   // (ph == FRA_PHASES: the 16 spare lanes)
  float s = 0.0f;
  bool add = false;
  const int j0 = blockIdx.x * FRA_COLS;
  const int jg = j0 + 4 * cg;
  // where the thread's four columns live: 1 = all in the hot partials, 2 = all in the cold partials (16-byte loads),
  // 3 = a group across a boundary or a misaligned layout (element by element), 0 = nothing to add
  const bool al = ((part_stride | partc_stride | hc) & 3) == 0;
  int mode = 3;
  if (ph >= FRA_PHASES || jg >= dp) mode = 0;
  else if (al && jg + 3 < hg) mode = 1;
  else if (al && jg >= hc && jg + 3 < hc + nc) mode = 2;
  const int j = j0 + tid;   // the column a lane of the first three waves finishes
  // everything the finishing lanes need that does not depend on the partials is requested NOW, next to the partials
  // (their round trips overlap): the column's weight and dimSparsity value, the first worker's 64-bit accumulator
  const bool fin = tid < FRA_COLS && j < dp;
  // (relaxed ATOMIC loads: a plain load is sunk by the compiler behind the barriers, to where its result is consumed,
  //  and its round trip is paid a second time there; an atomic one stays where it is written)
  float wj = 0.0f, dsj = 0.0f;
  long long tot_next = 0;
  if (fin) {
    if (APPLY) {
      wj = __hip_atomic_load(&w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dsj = __hip_atomic_load(&ds[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tot_next = __hip_atomic_load(&g64_base[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  float gsum = 0.0f;        // Vec.sum over the workers, folded left with the Sparse filter after every add
  // Small steps (the reference's 3 x 100 and 4 x 200: a few workgroups per worker, no cold partials): every
  // (worker, workgroup) pair is ONE phase of a single pass -- one round trip and one pair of barriers for all workers
  // instead of one per worker; the finishing lanes then walk the workers exactly as the loop below does (each worker's
  // exact sum rounded once, regularised, folded in worker order).
  const bool flat = n_workers > 1 && n_workers <= 4 && n_workers * n_wg <= FRA_PHASES && nc == 0;
  if (flat) {
    long long q[4] = {0, 0, 0, 0};
    if (ph < n_workers * n_wg) {
      if (mode == 1) {
        const int4 v = *reinterpret_cast<const int4*>(part + (long long)ph * part_stride + jg);   // ph = k * n_wg + b
        q[0] = v.x;
        q[1] = v.y;
        q[2] = v.z;
        q[3] = v.w;
      } else if (mode == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (jg + e < hg) q[e] = part[(long long)ph * part_stride + jg + e];
      }
    }
    long long tot_k[4] = {tot_next, 0, 0, 0};
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (fin && k < n_workers)
        tot_k[k] = __hip_atomic_load(&(g64_base + (long long)k * g_stride)[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ph < FRA_PHASES) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[ph][4 * cg + e] = q[e];
    }
    __syncthreads();
    s = s_sh;
    add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
    if (fin) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < n_workers) {
          long long tot = tot_k[k];
          if (tot != 0) (g64_base + (long long)k * g_stride)[j] = 0;
          for (int b = 0; b < n_wg; ++b) tot += red[k * n_wg + b][tid];
          float gv = filt((float)((double)tot * (j >= hc ? inv_scale_cold : inv_scale)));   // one rounding of the exact sum
          if (add && gv != 0.0f) gv = filt(gv + s);
          gsum = filt(gsum + gv);
        }
      }
    }
  }
  for (int k = 0; k < (flat ? 0 : n_workers); ++k) {
    long long q[4] = {0, 0, 0, 0};
    if (mode == 1 || mode == 2) {
      const int n = mode == 1 ? n_wg : n_wgc;
      const long long st = mode == 1 ? part_stride : partc_stride;
      const int* p = (mode == 1 ? part + jg : partc + (jg - hc)) + (long long)k * n * st;
#pragma unroll 4
      for (int b = ph; b < n; b += FRA_PHASES) {
        const int4 v = *reinterpret_cast<const int4*>(p + (long long)b * st);
        q[0] += v.x;
        q[1] += v.y;
        q[2] += v.z;
        q[3] += v.w;
      }
    } else if (mode == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int je = jg + e;
        if (je < hg) {
          const int* p = part + (long long)k * n_wg * part_stride + je;
          for (int b = ph; b < n_wg; b += FRA_PHASES) q[e] += (long long)p[(long long)b * part_stride];
        } else if (je >= hc && je < hc + nc) {
          const int* p = partc + (long long)k * n_wgc * partc_stride + (je - hc);
          for (int b = ph; b < n_wgc; b += FRA_PHASES) q[e] += (long long)p[(long long)b * partc_stride];
        }
      }
    }
    if (k) __syncthreads();   // the previous worker's sums have been read
    if (ph < FRA_PHASES) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[ph][4 * cg + e] = q[e];
    }
    __syncthreads();
    if (k == 0) {
      s = s_sh;
      add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
    }
    long long tot = tot_next;
    if (fin && k + 1 < n_workers)   // (under the barrier below)
      tot_next = __hip_atomic_load(&(g64_base + (long long)(k + 1) * g_stride)[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (fin) {
      long long* g64 = g64_base + (long long)k * g_stride;
      if (tot != 0) g64[j] = 0;
#pragma unroll
      for (int i = 0; i < FRA_PHASES; ++i) tot += red[i][tid];
      float gv = filt((float)((double)tot * (j >= hc ? inv_scale_cold : inv_scale)));   // one rounding of the exact sum
      if (add && gv != 0.0f) gv = filt(gv + s);
      gsum = filt(gsum + gv);
    }
  }
  if (!APPLY) {
    if (tid < FRA_COLS && j < dp) gsum_out[j] = gsum;
    return;
  }
  fra_update_and_scalars(gsum, (float)n_workers, j, dp, w, wj, dsj, lr, lambda, sc, redpart, fred, &is_last, false);
}

// The update outside the fused kernel: after the all-reduce across ranks (a communicator is attached) and for
// dsgd_apply.  The same columns per block, the same arithmetic and the same summation order of the two dot products as
// the fused kernel above -- a communicator of size one leaves exactly the weights and the regulariser scalar of the
// engine without one.
__global__ void __launch_bounds__(256) dsgd_apply_cols_kernel(float* __restrict__ w, const float* __restrict__ gsum,
                                                             const float* __restrict__ ds, int dp, float k_total, float lr,
                                                             float lambda, DevScalars* sc, float* __restrict__ redpart,
                                                             int finalize) {
  __shared__ float fred[8];
  __shared__ int is_last;
  const int j = blockIdx.x * FRA_COLS + threadIdx.x;
  const bool mine = threadIdx.x < FRA_COLS && j < dp;
  const float g = mine ? gsum[j] : 0.0f;
  const float wj = mine ? w[j] : 0.0f, dsj = mine ? ds[j] : 0.0f;
  fra_update_and_scalars(g, k_total, j, dp, w, wj, dsj, lr, lambda, sc, redpart, fred, &is_last, finalize != 0);
}

// s and |w|^2 for weights that were set from outside (dsgd_set_weights, the lock-free engine's weights at a loss
// check): same blocks, same order as above; 247 blocks instead of one workgroup walking all D + 1 columns (17 us).
__global__ void __launch_bounds__(256) dsgd_wstats_cols_kernel(const float* __restrict__ w, const float* __restrict__ ds,
                                                              int dp, float lambda, DevScalars* sc,
                                                              float* __restrict__ redpart) {
  __shared__ float fred[8];
  __shared__ int is_last;
  const int j = blockIdx.x * FRA_COLS + threadIdx.x;
  float dot = 0.0f, nsq = 0.0f;
  if (threadIdx.x < FRA_COLS && j < dp) {
    const float wn = w[j];
    dot = filt(wn * ds[j]);
    nsq = wn * wn;
  }
  fra_scalars(dot, nsq, lambda, sc, redpart, fred, &is_last, true);
}

// segmented scan over the 64 lanes of a wave, DPP only (no LDS round trip)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get_f(float src) {  // lanes without a source (or masked rows) read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_get_i(int src) {
  return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, 0xf, false);
}
// one step of the inclusive segmented scan: add the partner's running sum unless a segment head
// lies between the partner and this lane
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(float& v, int& f) {
  const float pv = dpp_get_f<CTRL, ROW_MASK>(v);
  const int pf = dpp_get_i<CTRL, ROW_MASK>(f);
  v = f ? v : v + pv;
  f |= pf;
}
// inclusive segmented scan over the 64 lanes of a wave (f = 1 on lanes that start a new segment)
__device__ __forceinline__ void wave_seg_scan(float& v, int& f) {
  seg_scan_step<0x111, 0xf>(v, f);  // row_shr:1
  seg_scan_step<0x112, 0xf>(v, f);  // row_shr:2
  seg_scan_step<0x114, 0xf>(v, f);  // row_shr:4
  seg_scan_step<0x118, 0xf>(v, f);  // row_shr:8
  seg_scan_step<0x142, 0xa>(v, f);  // row_bcast:15 -> rows 1 and 3
  seg_scan_step<0x143, 0xc>(v, f);  // row_bcast:31 -> rows 2 and 3
}

// ======================================================================================================
// K1e / K5d: WAVE-independent streaming kernels ("wseg"): no workgroup barrier inside the stream loop
// ======================================================================================================
// Measurements on MI355X that shaped this kernel (profiles/README.md):
//  * with one 1024-lane workgroup per CU (needed for the 160 KiB of LDS tiles) two barriers per tile put all 16
//    waves in lockstep -- HBM, VALU and the LDS atomics were used one after the other (51-70 % SQ_WAIT_ANY);
//  * the first barrier-free version executed ~750 instructions per 512-slot tile and was bound by VALU issue;
//  * trimming it to ~250 changed nothing once the memory pipeline was the limiter: what helped next was taking
//    the cold columns out of the stream (split layout), exact-size windows, and atomics that never overflow.
// So: every WAVE owns its own tiles of 512 slots made of WHOLE rows (no data ever crosses waves; the 16 waves of the
// workgroup drift apart and overlap each other's memory waits, DPP scans and LDS atomics while sharing the LDS
// weight tile and the LDS gradient tile):
//   * lane l owns the 8 CONTIGUOUS slots [8l, 8l+8): one 16-byte buffer load of eight 16-bit ranks and two of eight
//     values through resources that cover exactly the tile's own bytes, four register sets rotated by unrolling
//     (three tiles in flight); every request of a tile -- ranks, values, lane descriptors, the cold parts of its
//     rows' x.w -- needs the tile record only (no load depends on a load), all indices are 32-bit (scalar unit);
//   * one 16-bit descriptor per lane: row-start bits (8), label sign of the row ENDING at each start (8); the
//     local row of the lane's first slot (rows 1-based, 0 / nrows+1 = padding) is a DPP prefix sum over the start
//     bits: no per-row loads;
//   * ONE DPP segmented scan per tile; a lane with at most one row start (the common case: rows >= 8
//     non-zeros) finalises branch-free, the general loop runs only when some lane of the wave holds two;
//   * the cold parts of the rows' x.w come in and the gate coefficients go out through a per-wave LDS strip (slot r:
//     row r's cold part until the lane that closes row r has read it, then its coefficient, pre-multiplied by the
//     fixed-point scale; same wave writes and reads: LDS executes a wave's accesses in order, no barrier);
//   * rows longer than WS_MAXNNZ non-zeros are processed after the tiles, one wave per row.
// The tiles hold the HOT columns only (split layout: 16-bit ranks, cold part of x.w from dcold, plain ds_add_u32 with
// a per-launch scale that rules out overflow).  The first generation (all columns in one stream, cold weights
// gathered, returning atomics with a spill rule) is in the git history; profiles/README.md keeps its numbers.
// Rounding to the fixed-point grid also absorbs the reference's 1e-20 filter on y*x.
constexpr int WS_SLOTS = 512;
constexpr int WS_MAXNNZ = WS_SLOTS - 8;
constexpr int WS_MAXROWS = 254;
constexpr int WS_PAD = WS_SLOTS + 8;   // padding elements behind col/val
constexpr int WS_SPILL_AT = 1 << 28;   // cold-stream accumulators (shift 21, returning atomics): spill / panic bands
constexpr int WS_PANIC_AT = 1 << 30;
constexpr int WS_COEF_STRIDE = 256;

struct WTile {       // 16 bytes, read with one scalar load
  long long pos0;    // first slot of the window (multiple of 4)
  int r0;            // global row of local row 1
  int info;          // rows in the tile (low 16 bits, signed; -1: unused entry) | slots holding its non-zeros << 16
};

struct WTables {
  const WTile* __restrict__ tiles;
  const unsigned short* __restrict__ meta;  // n_tiles x 64 lane descriptors: row-start bits | label signs << 8
};

struct WRegs {
  int4 c0;             // eight 16-bit column ranks
  float4 v0, v1;
  float dcv;           // cold part of x.w of the tile's local row lane + 1 (0 beyond the tile's rows)
  unsigned int meta;
  long long pos0;      // wave-uniform (scalar registers): window start
  int tc;              // ... clamped tile index
  int r0, nrows, nb;   // wave-uniform: first row, rows, bytes of the window that belong to the tile
};

// the tile record of tile t (clamped), fetched one iteration before w_issue_cols needs it: a load whose
// result feeds the address computation would otherwise expose its latency once per tile
// (a scalar s_load -- counted by lgkmcnt, not queued behind the stream loads in vmcnt -- as long as the kernel
// passes the table as a __restrict__ parameter of its own)
// (tile indices are 32-bit: a 64-bit `t < t_end` is a VALU compare whose operands the register allocator parks in
//  whatever vector registers are dead -- e.g. half of a register set with a stream load still in flight, and the
//  write-after-write hazard made the loop head wait for EVERY outstanding load (s_waitcnt vmcnt(0)) once per round)
__device__ __forceinline__ WTile w_fetch(const WTables& tt, int t, int t_end) {
  return tt.tiles[t < t_end ? t : t_end - 1];
}

// the stream of a tile: eight 16-bit column ranks (one 16-byte load), eight values (two), the lane descriptor
__device__ __forceinline__ void w_issue_cols(const CsrView& m, int t, int t_end, int lane, const WTile& wt,
                                             WRegs& r) {
  const bool live = t < t_end;
  r.tc = live ? t : t_end - 1;     // wave-uniform
  r.r0 = wt.r0;
  r.nrows = live ? (int)(short)(wt.info & 0xffff) : -1;
  r.pos0 = wt.pos0;
  // Raw buffer over exactly the tile's own slots (rounded up to 16 bytes): lanes past the end get zeros WITHOUT a
  // memory access.  Reading the whole 512-slot window would fetch the first ~50 slots of the next tile a second
  // time -- with three tiles in flight per wave those lines have left the caches again (rocprofv3 FETCH_SIZE was
  // 1.2 x the algorithmic bytes).
  r.nb = (int)(((unsigned int)wt.info >> 16) + 3u & ~3u) * 4;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  // The hot stream carries 16-BIT column ranks (hot ranks < hsplit <= 65536 always: LDS caps hsplit near 18 K), eight
  // per lane in ONE 16-byte load -- 6 bytes per non-zero instead of 8.  The kernel was moving 5.6 TB/s of physical
  // traffic (89 % of the 6.29 TB/s copy ceiling) at 0.62 of the ALGORITHMIC roofline: bytes, not latency, were the
  // lever.  Windows start at a multiple of 8 slots (16-byte aligned in this array).
  const unsigned short* col16 = reinterpret_cast<const unsigned short*>(m.col);
  const int nb16 = (int)(((unsigned int)wt.info >> 16) + 7u & ~7u) * 2;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(col16 + wt.pos0), 0, nb16, 0x00020000);
  const i32x4 a = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, 0, 0));
  r.c0 = make_int4(a.x, a.y, a.z, a.w);
}
__device__ __forceinline__ void w_issue_vals(const CsrView& m, const WTables& tt, const float* __restrict__ dcold, int lane,
                                             WRegs& r) {
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.val + r.pos0), 0, r.nb, 0x00020000);
  // (whole-vector bit casts: an element-wise __builtin_bit_cast(float, a.x) of the returned vector is folded to
  //  component 0 for all four elements by this compiler)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane, 0, 0));
  const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane + 16, 0, 0));
  r.v0 = make_float4(a.x, a.y, a.z, a.w);
  r.v1 = make_float4(b.x, b.y, b.z, b.w);
  // lane descriptor, 16 bits (row-start bits | label signs << 8); the local row of the lane's first slot is a wave
  // prefix sum over the start bits (w_rows_below) instead of a stored field
  r.meta = (tt.meta + (long long)r.tc * 64)[(unsigned int)lane];
  // The cold part of x.w of the tile's rows, by local row: ONE coalesced load whose address needs the tile record
  // only.  Before round 3 every lane fetched dcold[row ending in this lane] one tile ahead -- an address that needs the
  // lane descriptors, i.e. a load that depends on a load: issued behind the stream requests of the two tiles after it,
  // and vmcnt retires in order, so waiting for it also waited for those (one tile in flight during the arithmetic
  // instead of three).  Tiles of more than 64 rows (rows shorter than 8 non-zeros) fetch the rest when they are
  // processed (w_tile).
  const __amdgpu_buffer_rsrc_t rd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dcold + r.r0), 0, r.nrows > 0 ? 4 * r.nrows : 0, 0x00020000);
  r.dcv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, 4 * lane, 0, 0));
}

struct WCtx {
  signed char* coef8;
  float* coefw;   // this wave's strip of WS_COEF_STRIDE floats
  int* gl;
  const float* wl;
  long long* g64;
  DevScalars* sc;
  const float* dcold;   // per-row cold part of x.w, written by the cold-stream kernel
  int row_begin, row_end;   // (rows are 32-bit throughout the tile records)
  int hw, hg;
  float fix_scale;    // fixed-point scale of the LDS gradient tile (chosen per launch)
  float cold_scale;   // scale of the cold columns' 64-bit accumulators (2^FIX_SHIFT / vmax2)
};

// row starts in the lanes below this one = the local row that ENDS at this lane's first start (rows are 1-based, the
// end mark of the tile's last row counts as a start): inclusive DPP scan of the popcounts minus the lane's own
__device__ __forceinline__ int w_rows_below(unsigned int bits) {
  const int pc = __popc(bits);
  int v = pc;
  v += dpp_get_i<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get_i<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get_i<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get_i<0x118, 0xf>(v);   // row_shr:8
  v += dpp_get_i<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_get_i<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v - pc;
}


// fixed-point scatter of a lane's eight contributions into the LDS gradient tile.  The host picks the fixed-point
// scale of the launch so that NO sum of one workgroup can leave 32 bits (at most one contribution per row and column,
// |contribution| <= 2^shift, rows per workgroup known; dsgd_wseg_bound_kernel refines it): plain ds_add_u32 under the
// exec mask -- no return value, no spill path, nothing to wait for.  Measured on MI355X against the returning form
// with dummy slots and 2^21 scaling: -18 % kernel time in the trained state (one-sided sums kept crossing the 2^28
// spill threshold), identical loss to 6 digits.  cc[] are LDS byte offsets.
__device__ __forceinline__ void w_scatter(const WCtx& x, const int (&cc)[8], const int (&q)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (q[k] != 0) atomicAdd(reinterpret_cast<int*>(reinterpret_cast<char*>(x.gl) + cc[k]), q[k]);
}

// One tile of one wave.  `cur` = tile t (everything landed), `far` = the register set that receives tile t+3.  The
// requests of a tile -- column ranks, values, lane descriptors, the cold parts of its rows' x.w -- need the tile record
// only and are issued FIRST, so that three tiles are on their way while the wave works on tile t.  The tiles hold the
// HOT part of the matrix only (every column rank < hw = hg): no gathers, no clamps, no cold checks.
// (w_tile_q: the tile whose requests go out, t_issue, and the one whose record is fetched, t_fetch, are named by the
//  caller -- the row-chunk kernel hands a workgroup's tiles to its waves as they come free, csrc/dsgd_fstep.hpp;
//  w_tile is the strided walk of the streaming kernel)
template <bool SCATTER>
__device__ __forceinline__ void w_tile_q(const CsrView& m, const WTables& tt, const WCtx& x, int t_issue, int t_fetch,
                                         int t_end, WRegs& cur, WRegs& far, WTile& wt_far, unsigned int& n_all,
                                         unsigned int& n_neg, unsigned int& n_pos);
template <bool SCATTER>
__device__ __forceinline__ void w_tile(const CsrView& m, const WTables& tt, const WCtx& x, int tile,
                                       int stride, int t_end, WRegs& cur, WRegs& far,
                                       WTile& wt_far, unsigned int& n_all, unsigned int& n_neg, unsigned int& n_pos) {
  w_tile_q<SCATTER>(m, tt, x, tile + 3 * stride, tile + 4 * stride, t_end, cur, far, wt_far, n_all, n_neg, n_pos);
}
template <bool SCATTER>
__device__ __forceinline__ void w_tile_q(const CsrView& m, const WTables& tt, const WCtx& x, int t_issue, int t_fetch,
                                         int t_end, WRegs& cur, WRegs& far, WTile& wt_far, unsigned int& n_all,
                                         unsigned int& n_neg, unsigned int& n_pos) {
  const int lane = threadIdx.x & 63;
  // the record load goes out BEFORE this iteration's stream loads: vmcnt retires in order, so next iteration's
  // wait for it does not drain the stream loads issued behind it
  const WTile wt_now = wt_far;                                              // record fetched last iteration
  wt_far = w_fetch(tt, t_fetch, t_end);                                     // record used next iteration
  w_issue_cols(m, t_issue, t_end, lane, wt_now, far);
  w_issue_vals(m, tt, x.dcold, lane, far);   // (values with the column ids: the window descriptor stays in scalar registers)

  const int nrows = cur.nrows;                          // wave-uniform; -1: the whole tile is padding
  const unsigned int desc = nrows < 0 ? 0u : cur.meta;
  const unsigned int bits = desc & 255u;
  const int re_n = w_rows_below(bits);          // local row that ends at the lane's first start
  const int rf = re_n + (int)(bits & 1u);       // local row of the lane's first slot (a start at slot 0 opens the NEW row)
  // the cold parts of the tile's rows go through the wave's coefficient strip: slot r holds dcold of local row r until the
  // lane that closes row r has read it -- the same lane then overwrites it with the row's gate coefficient (LDS executes a
  // wave's accesses in order; every other lane reads coefficients only behind the wave barrier further down)
  x.coefw[lane + 1] = cur.dcv;
  if (nrows > 64) {   // wave-uniform, rare: rows shorter than 8 non-zeros
    for (int i = 64 + lane; i < nrows; i += 64) x.coefw[i + 1] = (x.dcold + __builtin_amdgcn_readfirstlane(cur.r0))[i];
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned int ys = (desc >> 8) & 255u;
  int cc[8];
  {
    // 16-bit ranks -> LDS byte offsets 4 * rank (two VALU per id; the weights sit at LDS address 0)
    const unsigned int cw[4] = {(unsigned int)cur.c0.x, (unsigned int)cur.c0.y, (unsigned int)cur.c0.z, (unsigned int)cur.c0.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // low half: one SDWA shift of the selected 16-bit word (the compiler finds that form for the high half only and
      // spends a shift and a mask on this one)
      unsigned int lo4;
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
          : "=v"(lo4)
          : "v"(2u), "v"(cw[k]));
      cc[2 * k] = (int)lo4;
      cc[2 * k + 1] = (int)((cw[k] >> 14) & 0x3fffcu);
    }
  }
  const float vv[8] = {cur.v0.x, cur.v0.y, cur.v0.z, cur.v0.w, cur.v1.x, cur.v1.y, cur.v1.z, cur.v1.w};
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  // hot weights: eight LDS reads back to back; products filtered as the reference's product map is.
  // ref: math/Sparse.scala:46
  float a[8], pk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // byte offset = LDS address: the weights sit at LDS address 0 (checked when the kernel starts) -- adding the
    // link-time base cost a v_add_u32 with literal 0 per slot
    a[k] = *(lds_cfloat*)(unsigned int)cc[k];
  }
  {
    // products two at a time (v_pk_mul_f32), then the reference's 1e-20 filter
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const f32x2 pr = f32x2{vv[k], vv[k + 1]} * f32x2{a[k], a[k + 1]};
      pk[k] = filt(pr.x);
      pk[k + 1] = filt(pr.y);
    }
  }

  // rows of the worker's batch, as local rows of this tile (wave-uniform: readfirstlane keeps the 64-bit clamps and
  // the row bases on the scalar unit -- the compiler had them in vector registers, ~20 VALU instructions per tile)
  const int r0s = __builtin_amdgcn_readfirstlane(cur.r0);
  const int lo = x.row_begin - r0s + 1, hi = x.row_end - r0s + 1;   // (both operands in [0, 2^31): no overflow)
  const int r_lo = lo < 1 ? 1 : (lo > 1024 ? 1024 : lo);
  const int r_hi = hi > nrows + 1 ? nrows + 1 : (hi < 0 ? 0 : hi);   // exclusive
  const float ps = x.fix_scale, ns = -x.fix_scale;
  signed char* const coef8_tile = x.coef8 + ((long long)r0s - 1);   // [local row] -> global row's gate

  const int nb = __popc(bits);
  int q[8];
  if (__builtin_amdgcn_ballot_w64(nb > 1) == 0) {
    // ---- common case: at most one row start per lane -> at most one row ENDS in this lane ----
    // T = slots from the lane's row start on (all eight when no row starts here): they continue into the next
    // lane (`trail`); the slots before the start close the row entering the lane (`head`)
    const unsigned int kstar = bits ? (unsigned int)__builtin_ctz(bits) : 0u;   // first slot of T
    bool in_t[8];
    float mk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      in_t[k] = (unsigned int)k >= kstar;
      mk[k] = in_t[k] ? pk[k] : 0.0f;
    }
    // pairwise trees (packed adds); head = total - trail is exact when the lane holds no start or starts at slot 0
    const float total = ((pk[0] + pk[1]) + (pk[2] + pk[3])) + ((pk[4] + pk[5]) + (pk[6] + pk[7]));
    const float trail = ((mk[0] + mk[1]) + (mk[2] + mk[3])) + ((mk[4] + mk[5]) + (mk[6] + mk[7]));
    const float head = total - trail;
    float s = trail;
    int f = bits != 0u;
    wave_seg_scan(s, f);
    const float incoming = dpp_get_f<0x138, 0xf>(s);      // wave_shr:1: running sum of the row entering this lane
    const int r_end = rf - (int)(bits & 1u);              // local row that ends at the lane's row start
    {
      const bool fin = nb == 1 && r_end >= r_lo && r_end < r_hi;
      const float dc = x.coefw[(nb == 1 && r_end >= 1 && r_end <= nrows) ? r_end : 0];
      const float d = (incoming + head) + dc;              // x . w of that row
      const bool ypos = (ys & bits) != 0u;
      const float yd = ypos ? d : -d;
      if (SCATTER) {
        const bool active = fin && !(yd < 0.0f);           // ref: core/ml/SparseSVM.scala:27-28
        if (nb == 1 && r_end >= 1 && r_end <= nrows) {
          x.coefw[r_end] = active ? (ypos ? ps : ns) : 0.0f;
          if (fin) coef8_tile[(unsigned int)r_end] = (signed char)(active ? (ypos ? 1 : -1) : 0);
        }
        n_all += active;
      } else {
        n_all += fin;                                       // ref: core/ml/SparseSVM.scala:14,16
        n_neg += fin && (yd < 0.0f);
        n_pos += fin && (yd > 0.0f);
      }
    }
    if (SCATTER) {
      if (lane == 0) {
        x.coefw[0] = 0.0f;                          // padding rows carry a zero coefficient
        x.coefw[nrows < 0 ? 1 : nrows + 1] = 0.0f;
      }
      __builtin_amdgcn_wave_barrier();  // same wave wrote the strip; LDS executes a wave's accesses in order
      // slots before the lane's row start belong to local row rf, the slots from it on to row rf + 1 -- unless
      // the start sits at slot 0 (then rf already names the new row) or there is no start at all
      const float cA = x.coefw[rf];
      const float cB = x.coefw[rf + 1 <= WS_MAXROWS + 1 ? rf + 1 : rf];
      const float cT = (bits != 0u && !(bits & 1u)) ? cB : cA;
      // y * x on the fixed-point grid: |v * coef| <= 2^21, so v * coef + 1.5 * 2^23 rounds (once, to nearest even)
      // to an fp32 whose low mantissa bits are the integer (two slots per v_pk_fma_f32)
      typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const f32x2 coef = {in_t[k] ? cT : cA, in_t[k + 1] ? cT : cA};
        const f32x2 r = __builtin_elementwise_fma(f32x2{vv[k], vv[k + 1]}, coef, f32x2{12582912.0f, 12582912.0f});
        q[k] = __float_as_int(r.x) - 0x4B400000;
        q[k + 1] = __float_as_int(r.y) - 0x4B400000;
      }
      w_scatter(x, cc, q);
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    // ---- general case (some lane holds two or more row starts: rows shorter than 8 non-zeros) ----
    float trail = 0.0f;   // the fragment after the lane's LAST row start continues into the next lane
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool st = (bits >> k) & 1u;
      trail = st ? 0.0f : trail;
      trail += pk[k];
    }
    float s = trail;
    int f = bits != 0u;
    wave_seg_scan(s, f);
    const float incoming = dpp_get_f<0x138, 0xf>(s);
    {
      float run = incoming;
      int r = rf - (int)(bits & 1u);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if ((bits >> k) & 1u) {
          if (r >= 1 && r <= nrows) {
            const bool in_range = r >= r_lo && r < r_hi;
            const bool ypos = (ys >> k) & 1u;
            const float dfull = run + x.coefw[r];
            const float yd = ypos ? dfull : -dfull;
            if (SCATTER) {
              const bool active = in_range && !(yd < 0.0f);
              x.coefw[r] = active ? (ypos ? ps : ns) : 0.0f;
              if (in_range) coef8_tile[(unsigned int)r] = (signed char)(active ? (ypos ? 1 : -1) : 0);
              n_all += active;
            } else {
              n_all += in_range;
              n_neg += in_range && (yd < 0.0f);
              n_pos += in_range && (yd > 0.0f);
            }
          }
          run = 0.0f;
          ++r;
        }
        run += pk[k];
      }
    }
    if (SCATTER) {
      if (lane == 0) {
        x.coefw[0] = 0.0f;
        x.coefw[nrows < 0 ? 1 : nrows + 1] = 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      int r = rf;
      float coef = x.coefw[r];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k > 0 && ((bits >> k) & 1u)) {
          ++r;
          coef = x.coefw[r];
        }
        q[k] = __float2int_rn(vv[k] * coef);
      }
      w_scatter(x, cc, q);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- rows longer than a wave tile: one wave per row, four 64-element chunks in flight -------------------------
// Run by the waves of dsgd_wseg_kernel after their tiles, from the WHOLE ranked CSR (same LDS weight tile and
// fixed-point accumulators; cold columns straight to the 64-bit accumulators at the cold scale); rows of 500+
// non-zeros are 0.5 % of the RCV1-like rows but 3 % of the non-zeros.
template <bool SCATTER>
__device__ __forceinline__ void w_long_row(const CsrView& m, const float* __restrict__ w, const WCtx& x, long long row,
                                           unsigned int& n_all, unsigned int& n_neg, unsigned int& n_pos) {
  typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
  const int lane = threadIdx.x & 63;
  const long long start = m.row_ptr[row], end = m.row_ptr[row + 1];
  const float y = (float)m.label[row];
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long p = start + lane; p < end; p += 256) {
    int c[4];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = p + 64 * k < end;
      c[k] = in ? m.col[p + 64 * k] : 0;
      v[k] = in ? m.val[p + 64 * k] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float wv = c[k] < x.hw ? ((lds_cvfloat*)x.wl)[c[k]] : w[c[k]];
      a[k] += filt(v[k] * wv);   // ref: math/Sparse.scala:46
    }
  }
  const float d = group_sum<64>((a[0] + a[1]) + (a[2] + a[3]));
  const float yd = y * d;
  if (SCATTER) {
    const bool active = !(yd < 0.0f);     // ref: core/ml/SparseSVM.scala:27-28
    if (lane == 0) {
      x.coef8[row] = (signed char)(active ? (int)y : 0);
      n_all += active;
    }
    if (active) {
      const float cs = y * x.fix_scale;
      for (long long p = start + lane; p < end; p += 64) {
        const int c = m.col[p];
        if (c < x.hg) {
          const int q = __float2int_rn(m.val[p] * cs);
          if (q != 0) atomicAdd(&x.gl[c], q);   // (these rows are part of the launch's row bound: no overflow)
        } else {
          const int q = __float2int_rn(m.val[p] * (y * x.cold_scale));
          if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&x.g64[c]), (unsigned long long)(long long)q);
        }
      }
    }
  } else if (lane == 0) {
    n_all += 1;
    n_neg += yd < 0.0f;
    n_pos += yd > 0.0f;
  }
}

// The same row with up to LR_IT x 256 of its non-zeros held in REGISTERS (the row-chunk kernel, csrc/dsgd_fstep.hpp): every
// stream request goes out before the first is used, every weight gather before the first product, and the scatter runs
// from the registers -- four round trips in a row (row id, bounds, stream, cold weights) instead of one per 256 non-zeros
// for x.w plus one per 64 for the scatter.  Same products, same order of additions as w_long_row (a[k] over the
// 256-element pieces in turn, then (a0 + a1) + (a2 + a3), then the wave sum), the same q per entry into the same
// accumulators: the same bits.
constexpr int LR_IT = 5;   // 1,280 non-zeros; longer rows take w_long_row
__device__ __forceinline__ void w_long_row_regs(const CsrView& m, const float* __restrict__ w, const WCtx& x, long long row,
                                                unsigned int& n_all, unsigned int& n_neg, unsigned int& n_pos) {
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  const int lane = threadIdx.x & 63;
  const long long start = m.row_ptr[row], end = m.row_ptr[row + 1];
  if (end - start > 256LL * LR_IT) {
    w_long_row<true>(m, w, x, row, n_all, n_neg, n_pos);
    return;
  }
  const float y = (float)m.label[row];
  int c[LR_IT][4];
  float v[LR_IT][4], wv[LR_IT][4];
#pragma unroll
  for (int i = 0; i < LR_IT; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long p = start + lane + 256 * i + 64 * k;
      const bool in = p < end;
      c[i][k] = in ? m.col[p] : -1;
      v[i][k] = in ? m.val[p] : 0.0f;
    }
  }
#pragma unroll
  for (int i = 0; i < LR_IT; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[i][k] = c[i][k] >= x.hw ? w[c[i][k]] : 0.0f;   // the few cold ranks: global, together
  }
#pragma unroll
  for (int i = 0; i < LR_IT; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = c[i][k];
      const float hot = ((lds_cfloat*)x.wl)[(cc >= 0 && cc < x.hw) ? cc : x.hw];   // (wl[hw] is the zero slot)
      wv[i][k] = (cc >= 0 && cc < x.hw) ? hot : wv[i][k];
    }
  }
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < LR_IT; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += filt(v[i][k] * wv[i][k]);       // ref: math/Sparse.scala:46
  }
  const float d = group_sum<64>((a[0] + a[1]) + (a[2] + a[3]));
  const bool active = !(y * d < 0.0f);     // ref: core/ml/SparseSVM.scala:27-28
  if (lane == 0) {
    x.coef8[row] = (signed char)(active ? (int)y : 0);
    n_all += active;
  }
  if (active) {
    const float cs = y * x.fix_scale, cq = y * x.cold_scale;
#pragma unroll
    for (int i = 0; i < LR_IT; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = c[i][k];
        if (cc >= 0 && cc < x.hg) {
          const int q = __float2int_rn(v[i][k] * cs);
          if (q != 0) atomicAdd(&x.gl[cc], q);   // (these rows are part of the launch's row bound: no overflow)
        } else if (cc >= 0) {
          const int q = __float2int_rn(v[i][k] * cq);
          if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&x.g64[cc]), (unsigned long long)(long long)q);
        }
      }
    }
  }
}

// m: the hot stream (row_ptr = hot row offsets, col = 16-bit ranks, val); mfull: the whole ranked CSR (long rows)
template <bool SCATTER>
__global__ void __launch_bounds__(1024) dsgd_wseg_kernel(CsrView m, CsrView mfull, const WTile* __restrict__ tiles,
                                                        const unsigned short* __restrict__ meta,
                                                        const float* __restrict__ w, long long* __restrict__ g64_base,
                                                        long long g_stride, const StreamSeg* __restrict__ segs,
                                                        DevScalars* __restrict__ sc, int hw, int hg, float fix_scale,
                                                        signed char* __restrict__ coef8,
                                                        const int* __restrict__ long_rows, int* __restrict__ part,
                                                        int part_stride, const float* __restrict__ dcold,
                                                        float cold_scale) {
  // (the tables are direct __restrict__ parameters: only then can the compiler prove that the stores of this kernel
  // do not clobber them and select scalar loads for the wave-uniform tile records)
  WTables tt;
  tt.tiles = tiles;
  tt.meta = meta;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile records, bases and row ranges stay in SGPRs
  WCtx x;
  x.coef8 = coef8;
  // LDS: hw + 1 weights FIRST (the column ranks become byte offsets from LDS address 0; zero slot at wl[hw]), 16
  // coefficient strips, hg + 64 gradient words (SCATTER only).
  float* wl = lds;
  float* strips = lds + ((hw + 4) & ~3);
  x.coefw = strips + wave * WS_COEF_STRIDE;
  x.gl = reinterpret_cast<int*>(strips + 16 * WS_COEF_STRIDE);
  x.wl = wl;
  const StreamSeg seg = segs[blockIdx.y];
  x.g64 = g64_base + (long long)blockIdx.y * g_stride;
  x.sc = sc;
  x.dcold = dcold;
  x.row_begin = (int)seg.row_begin;
  x.row_end = (int)seg.row_end;
  x.hw = hw;
  x.hg = hg;
  x.fix_scale = fix_scale;
  x.cold_scale = cold_scale;
  if (SCATTER) {
    if (is_aligned16(x.gl)) wg_zero(x.gl, hg + 64, tid, 1024);
    else
      for (int j = tid; j < hg + 64; j += 1024) x.gl[j] = 0;
  }
  if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)lds != 0u) {
    // (w_tile turns column ranks into LDS addresses without adding a base: all LDS of this kernel is dynamic)
    if (tid == 0) atomicOr(&sc->err, 2);
    return;
  }
  wg_copy_in(wl, w, hw, tid, 1024, is_aligned16(wl) && is_aligned16(w));
  if (tid == 0) wl[hw] = 0.0f;
  __syncthreads();

  unsigned int n_all = 0, n_neg = 0, n_pos = 0;
  const int stride = (int)gridDim.x * 16;          // waves of this worker's grid row
  const int t_end = (int)seg.tile_end;
  int tile = (int)seg.tile_begin + (int)blockIdx.x * 16 + wave;
  if (tile < t_end) {
    // four register sets rotated by unrolling: three tiles in flight per wave
    WRegs A, B, C, D;
    WTile wt = w_fetch(tt, tile, t_end);
    w_issue_cols(m, tile, t_end, lane, wt, A);
    w_issue_vals(m, tt, x.dcold, lane, A);
    __builtin_amdgcn_sched_barrier(0);   // (a tile's requests stay together and in tile order: vmcnt retires in order,
                                         //  and the loop's first wait merges this path with the back edge)
    wt = w_fetch(tt, tile + stride, t_end);
    w_issue_cols(m, tile + stride, t_end, lane, wt, B);
    w_issue_vals(m, tt, x.dcold, lane, B);
    __builtin_amdgcn_sched_barrier(0);
    wt = w_fetch(tt, tile + 2 * stride, t_end);
    w_issue_cols(m, tile + 2 * stride, t_end, lane, wt, C);
    w_issue_vals(m, tt, x.dcold, lane, C);
    __builtin_amdgcn_sched_barrier(0);
    wt = w_fetch(tt, tile + 3 * stride, t_end);
#define DSGD_WT(CUR, FAR) w_tile<SCATTER>(m, tt, x, tile, stride, t_end, CUR, FAR, wt, n_all, n_neg, n_pos)
    for (;;) {
      DSGD_WT(A, D); tile += stride; if (tile >= t_end) break;
      DSGD_WT(B, A); tile += stride; if (tile >= t_end) break;
      DSGD_WT(C, B); tile += stride; if (tile >= t_end) break;
      DSGD_WT(D, C); tile += stride; if (tile >= t_end) break;
    }
#undef DSGD_WT
  }
  // rows that fit no tile: one wave per row
  for (long long t = seg.long_begin + (long long)blockIdx.x * 16 + wave; t < seg.long_end; t += (long long)stride)
    w_long_row<SCATTER>(mfull, w, x, (long long)long_rows[t], n_all, n_neg, n_pos);

  // Tallies: ONE global atomic per workgroup and counter.  (One per wave put 4,096 atomics on one address at the end
  // of the launch, when every wave arrives together: they are served one after the other -- measured on the
  // index-list kernel, where that queue was 40 of 67 us.)  The strips are free now: every wave's tiles are done.
  n_all = wave_sum_u32(n_all);
  n_neg = wave_sum_u32(n_neg);
  n_pos = wave_sum_u32(n_pos);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    unsigned int* mine3 = reinterpret_cast<unsigned int*>(x.coefw);
    mine3[0] = n_all;
    mine3[1] = n_neg;
    mine3[2] = n_pos;
  }
  __syncthreads();
  unsigned int t_all = 0, t_neg = 0, t_pos = 0;
  if (tid == 0) {
    for (int i = 0; i < 16; ++i) {
      const unsigned int* s3 = reinterpret_cast<const unsigned int*>(strips + i * WS_COEF_STRIDE);
      t_all += s3[0];
      t_neg += s3[1];
      t_pos += s3[2];
    }
  }
  if (SCATTER) {
    // this workgroup's exact partial sums, written whole (zeros included): the reduce kernels add the partials of a
    // worker in a fixed order -- no atomics, and 256 workgroups do not meet on one address
    int* mine = part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * part_stride;
    wg_copy_out(mine, x.gl, hg, tid, 1024, is_aligned16(mine) && is_aligned16(x.gl));
    if (tid == 0 && t_all) atomicAdd(&sc->n_active, (unsigned long long)t_all);
  } else {
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&sc->counts[3], (unsigned long long)(seg.row_end - seg.row_begin));
    if (tid == 0) {
      if (t_neg) atomicAdd(&sc->counts[0], (unsigned long long)t_neg);                          // pred == y
      if (t_all - t_neg - t_pos) atomicAdd(&sc->counts[1], (unsigned long long)(t_all - t_neg - t_pos));  // pred == 0
      if (t_pos) atomicAdd(&sc->counts[2], (unsigned long long)t_pos);                          // pred == -y
    }
  }
}

// ---- split layout: a data-dependent bound for the fixed-point scale of the hot gradient tile -------------------
// The LDS accumulators of dsgd_wseg_kernel are 32-bit; a column's sum over the rows ONE workgroup sees must stay
// below 2^30.  "rows per workgroup x largest value" (26 K rows -> shift 15) is a crude bound for L2-normalised rows:
// what actually limits the sum is A = max_column sum_rows |x|.  This kernel runs the launch's own tile -> workgroup
// assignment once per (ranges, grid) configuration, accumulates ceil(|x| * 2^s0 / vmax2) per column in LDS at the
// crude-but-safe shift s0 and returns the largest column sum; the host then picks the finest shift s <= 21 with
// 2^(s - s0) * A + rows <= 2^30 (the `+ rows` covers the half-unit rounding of every contribution).  Whatever
// subset of rows is active and whatever their signs, no accumulator can pass that bound.
__global__ void __launch_bounds__(1024) dsgd_wseg_bound_kernel(const long long* __restrict__ hrow_ptr,
                                                              const unsigned short* __restrict__ hcol16,
                                                              const float* __restrict__ hval,
                                                              const WTile* __restrict__ tiles, long long n_tiles,
                                                              long long n_rows, CsrView mfull,
                                                              const int* __restrict__ long_rows,
                                                              const StreamSeg* __restrict__ segs, int hg, float scale0,
                                                              unsigned int* __restrict__ out_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned int bl[];   // hg column sums + 16 words
  unsigned int* red = bl + hg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < hg; j += 1024) bl[j] = 0u;
  __syncthreads();
  const StreamSeg seg = segs[blockIdx.y];
  for (long long t = seg.tile_begin + (long long)blockIdx.x * 16; t < seg.tile_end; t += (long long)gridDim.x * 16) {
    const long long t1 = t + 16 < seg.tile_end ? t + 16 : seg.tile_end;
    const long long ra = tiles[t].r0, rb = t1 < n_tiles ? (long long)tiles[t1].r0 : n_rows;
    const long long e0 = hrow_ptr[ra], e1 = hrow_ptr[rb];
    for (long long e = e0 + tid; e < e1; e += 1024) {
      const unsigned int q = (unsigned int)ceilf(fabsf(hval[e]) * scale0);
      if (q) atomicAdd(&bl[hcol16[e]], q);
    }
  }
  // the rows that fit no tile are shared out one wave per row, exactly as the gradient kernel does
  for (long long t = seg.long_begin + (long long)blockIdx.x * 16 + wave; t < seg.long_end; t += (long long)gridDim.x * 16) {
    const long long row = long_rows[t];
    for (long long p = mfull.row_ptr[row] + lane; p < mfull.row_ptr[row + 1]; p += 64) {
      const int c = mfull.col[p];
      const unsigned int q = (unsigned int)ceilf(fabsf(mfull.val[p]) * scale0);
      if (c < hg && q) atomicAdd(&bl[c], q);
    }
  }
  __syncthreads();
  unsigned int m = 0u;
  for (int j = tid; j < hg; j += 1024) m = max(m, bl[j]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off, 64));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 16; ++i) m = max(m, red[i]);
    atomicMax(out_max, m);
  }
}

// ======================================================================================================
// the matrix split by column rank -- layout kernels and the two cold-stream kernels
// ======================================================================================================
// Measured on MI355X (profiles/README.md): with 9 % of the non-zeros outside the LDS weight tile the eight
// per-lane gathers of a tile keep a CU's texture addresser busy 60 % of the time (20 % without them), and the
// transposed cold lists pay one scattered coefficient lookup per cold entry.  Both disappear when the cold
// entries leave the main stream: they form their own row-ordered stream that two small kernels read linearly --
// dsgd_cdot_kernel with the cold WEIGHTS in LDS (cold part of x.w per row, before the main kernel),
// dsgd_cgrad_kernel with the cold GRADIENT in LDS (after it, gate coefficients read by row).
//
// Third generation (round 3): the cold stream has the SAME form as the hot stream -- (rank - hsplit) as 16-bit words
// (32-bit when there are more than 65536 cold columns), fp32 values, wave tiles of WHOLE rows with 16-bit lane
// descriptors (row-start bits) -- 6 bytes per entry instead of 8 (the row of an entry used to ride in the upper half
// of a 32-bit key), no row-start search, no carry between tiles, and the tile walk of the main kernel: exact-size
// buffer windows, four register sets rotated by unrolling, three tiles in flight per wave.  The second generation
// (eight contiguous entries per lane of a row-agnostic 512-entry tile, two tiles in flight; git history) measured
// 77-83 us (dot) and 66-70 us (gradient) for 0.30 GB each.

// cold entries per row (ranked column ids; G lanes per row)
template <int G>
__global__ void __launch_bounds__(256) dsgd_split_count_kernel(CsrView m, int hsplit, int* __restrict__ cnt_cold) {
  const int sub = threadIdx.x % G;
  const long long group = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const long long n_groups = (long long)gridDim.x * blockDim.x / G;
  for (long long row = group; row < m.n_rows; row += n_groups) {
    const long long start = m.row_ptr[row], end = m.row_ptr[row + 1];
    int n = 0;
    for (long long p = start + sub; p < end; p += G) n += m.col[p] >= hsplit;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) n += __shfl_xor(n, o, G);
    if (sub == 0) cnt_cold[row] = n;
  }
}

// one wave per row: stable partition of the row into the hot and the cold stream.  A row whose hot range is empty
// in hrow_ptr belongs to the long-row list and is left out of both streams; a row without any hot (cold) entry gets
// one explicit zero on the first hot (cold) rank so that every tiled row owns a slot in both streams.
template <bool COL16>
__global__ void __launch_bounds__(256) dsgd_split_fill_kernel(CsrView m, int hsplit,
                                                             const long long* __restrict__ hrow_ptr,
                                                             const long long* __restrict__ crow_ptr,
                                                             unsigned short* __restrict__ hcol, float* __restrict__ hval,
                                                             void* __restrict__ ccol, float* __restrict__ cval) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  unsigned short* ccol16 = reinterpret_cast<unsigned short*>(ccol);
  unsigned int* ccol32 = reinterpret_cast<unsigned int*>(ccol);
  for (long long row = wave; row < m.n_rows; row += n_waves) {
    long long hp = hrow_ptr[row];
    if (hrow_ptr[row + 1] == hp) continue;   // long row
    long long cp = crow_ptr[row];
    const long long start = m.row_ptr[row], end = m.row_ptr[row + 1];
    bool any_hot = false, any_cold = false;
    for (long long p0 = start; p0 < end; p0 += 64) {
      const long long p = p0 + lane;
      const bool in = p < end;
      const int c = in ? m.col[p] : 0;
      const float v = in ? m.val[p] : 0.0f;
      const bool hot = in && c < hsplit, cold = in && c >= hsplit;
      const unsigned long long mh = __builtin_amdgcn_ballot_w64(hot), mc = __builtin_amdgcn_ballot_w64(cold);
      const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
      if (hot) {
        const long long o = hp + __popcll(mh & below);
        hcol[o] = (unsigned short)c;   // the hot stream carries 16-bit ranks (6 bytes per non-zero)
        hval[o] = v;
      }
      if (cold) {
        const long long o = cp + __popcll(mc & below);
        if (COL16) ccol16[o] = (unsigned short)(c - hsplit);
        else ccol32[o] = (unsigned int)(c - hsplit);
        cval[o] = v;
      }
      hp += __popcll(mh);
      cp += __popcll(mc);
      any_hot = any_hot || mh != 0ull;
      any_cold = any_cold || mc != 0ull;
    }
    if (lane == 0) {
      if (!any_hot) {
        hcol[hp] = 0;
        hval[hp] = 0.0f;
      }
      if (!any_cold) {
        if (COL16) ccol16[cp] = 0;
        else ccol32[cp] = 0u;
        cval[cp] = 0.0f;
      }
    }
  }
}

// ---- the cold tile walk ------------------------------------------------------------------------------------
// Cold tiles: WTile records and 16-bit lane descriptors exactly as the hot stream's (build_wave_tiles on the cold row
// offsets), at most CT_MAXROWS rows per tile: cold rows hold ~6 entries, a 512-slot tile ~90 rows, and the gradient
// kernel keeps a tile's gate coefficients as two bytes per lane.
constexpr int CT_MAXROWS = 128;
constexpr int CT_STRIP = 256;        // floats per wave: [0, CT_MAXROWS + 2) by local row, [192, 256) per-lane dummy slots
constexpr int CT_DUMMY = 192;

template <bool COL16>
struct CRegs {
  int4 c0, c1;         // COL16: eight 16-bit ids in c0; otherwise eight 32-bit ids
  float4 v0, v1;
  unsigned int meta;   // row-start bits of the lane's eight slots
  int cf0, cf1;        // gradient kernel: gate coefficients of local rows lane + 1 and lane + 65 (0 outside the range)
  long long pos0;      // wave-uniform from here on
  int r0, nrows;
};

struct CTabs {
  const WTile* __restrict__ tiles;
  const unsigned short* __restrict__ meta;
  const void* __restrict__ col;
  const float* __restrict__ val;
  const signed char* __restrict__ coef8;   // gradient kernel only
  int row_begin, row_end;                  // the worker's rows (gradient kernel: rows outside contribute nothing)
  int t_lo, t_hi;                          // its tiles (32-bit: see w_fetch)
};

__device__ __forceinline__ int c_map(const CTabs& tt, int t) {   // logical tile -> tile record (clamped)
  return t < tt.t_hi ? t : tt.t_hi - 1;
}

template <bool COL16, bool GRAD>
__device__ __forceinline__ void c_issue(const CTabs& tt, int t, int lane, const WTile& wt, CRegs<COL16>& r) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const bool live = t < tt.t_hi;
  const int tc = c_map(tt, t);
  r.r0 = wt.r0;
  r.nrows = live ? (int)(short)(wt.info & 0xffff) : -1;
  r.pos0 = wt.pos0;
  const unsigned int slots = live ? (unsigned int)wt.info >> 16 : 0u;
  // raw buffers over exactly the tile's own slots: lanes past the end get zeros without a memory access
  const int nb = (int)((slots + 3u) & ~3u) * 4;
  if (COL16) {
    const unsigned short* col16 = reinterpret_cast<const unsigned short*>(tt.col);
    const int nb16 = (int)((slots + 7u) & ~7u) * 2;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(col16 + wt.pos0), 0, nb16, 0x00020000);
    const i32x4 a = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane, 0, 0));
    r.c0 = make_int4(a.x, a.y, a.z, a.w);
  } else {
    const int* col32 = reinterpret_cast<const int*>(tt.col);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(col32 + wt.pos0), 0, nb, 0x00020000);
    const i32x4 a = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane, 0, 0));
    const i32x4 b = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane + 16, 0, 0));
    r.c0 = make_int4(a.x, a.y, a.z, a.w);
    r.c1 = make_int4(b.x, b.y, b.z, b.w);
  }
  {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tt.val + wt.pos0), 0, nb, 0x00020000);
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 32 * lane + 16, 0, 0));
    r.v0 = make_float4(a.x, a.y, a.z, a.w);
    r.v1 = make_float4(b.x, b.y, b.z, b.w);
  }
  r.meta = (tt.meta + (long long)tc * 64)[(unsigned int)lane];
  if (GRAD) {
    // gate coefficients of the tile's rows, one byte per row (written by the main kernel): local row l + 1 is global
    // row r0 + l.  The window is clamped to the worker's rows -- whatever coef8 holds outside them is stale --
    // and lanes outside the window read 0 (an offset below the window wraps to a huge unsigned one).
    const int lo = tt.row_begin > wt.r0 ? tt.row_begin : wt.r0;
    int hi = wt.r0 + (r.nrows > 0 ? r.nrows : 0);
    if (hi > tt.row_end) hi = tt.row_end;
    const int n = hi > lo ? hi - lo : 0;
    const int shift = lo - wt.r0;   // 0 .. nrows
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char*>(tt.coef8 + lo), 0, n, 0x00020000);
    r.cf0 = (int)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, lane - shift, 0, 0);
    r.cf1 = (int)(signed char)__builtin_amdgcn_raw_buffer_load_b8(rs, lane + 64 - shift, 0, 0);
  }
}

// byte offsets 4 * id of a lane's eight entries (the LDS tile sits at LDS address 0)
template <bool COL16>
__device__ __forceinline__ void c_offsets(const CRegs<COL16>& r, int (&cc)[8]) {
  if (COL16) {
    const unsigned int cw[4] = {(unsigned int)r.c0.x, (unsigned int)r.c0.y, (unsigned int)r.c0.z, (unsigned int)r.c0.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int lo4;
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
          : "=v"(lo4)
          : "v"(2u), "v"(cw[k]));
      cc[2 * k] = (int)lo4;
      cc[2 * k + 1] = (int)((cw[k] >> 14) & 0x3fffcu);
    }
  } else {
    const int c[8] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w, r.c1.x, r.c1.y, r.c1.z, r.c1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) cc[k] = c[k] << 2;
  }
}

// local row closed by the lane's first row start = row starts in the lanes below (rows are 1-based; row 0 is the
// padding in front of the tile's first row): inclusive DPP scan of the popcounts minus the lane's own
__device__ __forceinline__ int c_rows_below(unsigned int bits) {
  const int pc = __popc(bits);
  int v = pc;
  v += dpp_get_i<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get_i<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get_i<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get_i<0x118, 0xf>(v);   // row_shr:8
  v += dpp_get_i<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_get_i<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v - pc;
}

// dcold[row] = sum over the cold entries of the row of filt(value * w[hsplit + col]): eight LDS reads, a sequential
// pass over the lane's slots (rows that begin and end inside the lane are complete there), ONE segmented scan for the
// fragments that cross lanes, the sums of the tile's rows meet in the wave's LDS strip and leave as coalesced stores.
// Every row of a tile is written, fixed order, no atomics.  ref: math/Vec.scala:58, math/Sparse.scala:46.
template <bool COL16, bool WIDE>
__device__ __forceinline__ void cd_tile(const CRegs<COL16>& cur, float* strip, float* __restrict__ dcold,
                                        const float* __restrict__ wcold, int nc_lds) {
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  const int lane = threadIdx.x & 63;
  const int nrows = cur.nrows;                                  // wave-uniform; -1: padding tile
  const unsigned int bits = nrows < 0 ? 0u : (cur.meta & 255u);
  const int re_n = c_rows_below(bits);
  int cc[8];
  c_offsets<COL16>(cur, cc);
  const float vv[8] = {cur.v0.x, cur.v0.y, cur.v0.z, cur.v0.w, cur.v1.x, cur.v1.y, cur.v1.z, cur.v1.w};
  float a[8], pk[8];
  if (!WIDE) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = *(lds_cfloat*)(unsigned int)cc[k];
  } else {   // cold columns beyond the LDS tile exist (very wide models only; a kernel of its own: a conditional load
             // in the tile loop costs every tile its counted waits)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lim = (nc_lds - 1) << 2;
      a[k] = *(lds_cfloat*)(unsigned int)(cc[k] < lim ? cc[k] : lim);
      if ((cc[k] >> 2) >= nc_lds) a[k] = wcold[cc[k] >> 2];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) pk[k] = filt(vv[k] * a[k]);
  float run = 0.0f, head = 0.0f;
  bool seen = false;
  int r = re_n;                                                 // the row the lane's next start closes
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const bool st = (bits >> k) & 1u;
    // a start that is not the lane's first closes a row that began in this lane: complete (per-lane dummy slot otherwise)
    strip[(st && seen) ? r : CT_DUMMY + lane] = run;
    head = (st && !seen) ? run : head;
    r += st ? 1 : 0;
    seen = seen || st;
    run = st ? 0.0f : run;
    run += pk[k];
  }
  float S = run;
  int f = bits != 0u;
  wave_seg_scan(S, f);
  const float incoming = dpp_get_f<0x138, 0xf>(S);              // wave_shr:1 (lane 0 reads 0)
  strip[bits != 0u ? re_n : CT_DUMMY + lane] = incoming + head;   // the row entering the lane ends at its first start
  __builtin_amdgcn_wave_barrier();                              // same wave writes and reads: LDS keeps a wave's order
  if (nrows > 0) {
    // A buffer over exactly the tile's rows: lanes past the last row are dropped by the bounds check.  The
    // wave-uniform branch is deliberate: with a straight-line tile body the compiler sinks the NEXT tiles' stream
    // requests (issued in front of this function) down the chain of exit checks to the iteration that consumes them
    // -- a block with two predecessors stops that.
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(dcold + cur.r0, 0, 4 * nrows, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(strip[lane + 1]), rs, 4 * lane, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(strip[lane + 65]), rs, 4 * lane + 256, 0, 0);
  }
  __builtin_amdgcn_wave_barrier();
}

// cold gradient columns: q = round(value * coef[row] * scale) accumulated in the LDS tile of 32-bit integers at the
// cold scale (shift 21) with returning atomics: whoever SEES |old| >= 2^28 moves the word into the 64-bit global
// accumulator, |old| >= 2^30 raises DevScalars::err (profiles/DESIGN_notes_r01-r05.md 3.5).  ref: core/Slave.scala:147-153 restricted to the
// cold columns.
template <bool COL16, bool WIDE>
__device__ __forceinline__ void cg_tile(const CRegs<COL16>& cur, float* strip, int* gc, long long* __restrict__ g64cold,
                                        DevScalars* __restrict__ sc, float fix_scale, int nc_lds) {
  const int lane = threadIdx.x & 63;
  const int nrows = cur.nrows;
  const unsigned int bits = nrows < 0 ? 0u : (cur.meta & 255u);
  const int re_n = c_rows_below(bits);
  // coefficients of the tile's rows by local row (0 for padding row 0, for the row behind the last and outside the range)
  strip[lane + 1] = (float)cur.cf0 * fix_scale;
  strip[lane + 65] = (float)cur.cf1 * fix_scale;
  if (lane == 0) strip[0] = 0.0f;
  if (lane == 1) strip[CT_MAXROWS + 1] = 0.0f;
  __builtin_amdgcn_wave_barrier();
  int cc[8];
  c_offsets<COL16>(cur, cc);
  const float vv[8] = {cur.v0.x, cur.v0.y, cur.v0.z, cur.v0.w, cur.v1.x, cur.v1.y, cur.v1.z, cur.v1.w};
  float coef[8];
  int r = re_n;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    r += (bits >> k) & 1u;                 // a start at slot k opens the next row
    coef[k] = strip[r <= CT_MAXROWS + 1 ? r : CT_MAXROWS + 1];
  }
  int q[8], old[8];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    // |v * coef| <= 2^21: v * coef + 1.5 * 2^23 rounds once (to nearest even) to an fp32 whose low mantissa bits are q
    const f32x2 rr = __builtin_elementwise_fma(f32x2{vv[k], vv[k + 1]}, f32x2{coef[k], coef[k + 1]},
                                               f32x2{12582912.0f, 12582912.0f});
    q[k] = __float_as_int(rr.x) - 0x4B400000;
    q[k + 1] = __float_as_int(rr.y) - 0x4B400000;
  }
  if (WIDE) {   // columns beyond the LDS tile go to the 64-bit global accumulator
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if ((cc[k] >> 2) >= nc_lds) {
        if (q[k] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g64cold[cc[k] >> 2]), (unsigned long long)(long long)q[k]);
        q[k] = 0;
      }
    }
  }
  char* gcb = reinterpret_cast<char*>(gc);
  const int dummy = (nc_lds + lane) << 2;
#pragma unroll
  for (int k = 0; k < 8; ++k) old[k] = atomicAdd(reinterpret_cast<int*>(gcb + (q[k] != 0 ? cc[k] : dummy)), q[k]);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (old[k] >= WS_SPILL_AT || old[k] <= -WS_SPILL_AT) {
      if (old[k] >= WS_PANIC_AT || old[k] <= -WS_PANIC_AT) atomicOr(&sc->err, 2);
      const int x = atomicExch(reinterpret_cast<int*>(gcb + cc[k]), 0);
      if (x != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g64cold[cc[k] >> 2]), (unsigned long long)(long long)x);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// GRAD = false: dsgd_cdot_kernel  (LDS: nc_lds cold weights at address 0, 16 strips)
// GRAD = true:  dsgd_cgrad_kernel (LDS: nc_lds + 64 gradient words at address 0, 16 strips; partials out)
template <bool COL16, bool GRAD, bool WIDE>
__global__ void __launch_bounds__(1024) dsgd_cold_kernel(const WTile* __restrict__ tiles,
                                                        const unsigned short* __restrict__ meta,
                                                        const void* __restrict__ col, const float* __restrict__ val,
                                                        const StreamSeg* __restrict__ segs, const float* __restrict__ w,
                                                        float* __restrict__ dcold, const signed char* __restrict__ coef8,
                                                        long long* __restrict__ g64_base, long long g_stride,
                                                        DevScalars* __restrict__ sc, int hsplit, int nc_lds,
                                                        float fix_scale, int* __restrict__ partc, int partc_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tile = GRAD ? nc_lds + 64 : nc_lds;
  float* strip = lds + ((n_tile + 3) & ~3) + wave * CT_STRIP;
  if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)lds != 0u) {
    if (tid == 0) atomicOr(&sc->err, 2);   // (ids become LDS addresses without a base: all LDS of this kernel is dynamic)
    return;
  }
  if (GRAD) wg_zero(reinterpret_cast<int*>(lds), n_tile, tid, 1024);
  else wg_copy_in(lds, w + hsplit, nc_lds, tid, 1024, is_aligned16(w + hsplit));
  __syncthreads();
  const StreamSeg seg = segs[blockIdx.y];
  CTabs tt;
  tt.tiles = tiles;
  tt.meta = meta;
  tt.col = col;
  tt.val = val;
  tt.coef8 = coef8;
  tt.row_begin = (int)seg.row_begin;
  tt.row_end = (int)seg.row_end;
  tt.t_lo = (int)seg.ctile_begin;
  tt.t_hi = (int)seg.ctile_end;
  long long* g64cold = GRAD ? g64_base + (long long)blockIdx.y * g_stride + hsplit : nullptr;
  const int stride = (int)gridDim.x * 16;
  int tile = tt.t_lo + (int)blockIdx.x * 16 + wave;
  if (tile < tt.t_hi) {
    CRegs<COL16> A, B, C, D;
    WTile wt = tiles[c_map(tt, tile)];
    // (scheduling barriers: the requests of a tile stay together and in tile order -- vmcnt retires in order, and the
    //  loop's first wait is the merge of this path and the back edge)
    c_issue<COL16, GRAD>(tt, tile, lane, wt, A);
    __builtin_amdgcn_sched_barrier(0);
    wt = tiles[c_map(tt, tile + stride)];
    c_issue<COL16, GRAD>(tt, tile + stride, lane, wt, B);
    __builtin_amdgcn_sched_barrier(0);
    wt = tiles[c_map(tt, tile + 2 * stride)];
    c_issue<COL16, GRAD>(tt, tile + 2 * stride, lane, wt, C);
    __builtin_amdgcn_sched_barrier(0);
    wt = tiles[c_map(tt, tile + 3 * stride)];
#define DSGD_CT(CUR, FAR)                                                                          \
  {                                                                                                \
    const WTile wt_now = wt;                                                                       \
    wt = tiles[c_map(tt, tile + 4 * stride)];                                                      \
    c_issue<COL16, GRAD>(tt, tile + 3 * stride, lane, wt_now, FAR);                                \
    if (GRAD) cg_tile<COL16, WIDE>(CUR, strip, reinterpret_cast<int*>(lds), g64cold, sc, fix_scale, nc_lds); \
    else cd_tile<COL16, WIDE>(CUR, strip, dcold, w + hsplit, nc_lds);                              \
  }
    for (;;) {
      DSGD_CT(A, D); tile += stride; if (tile >= tt.t_hi) break;
      DSGD_CT(B, A); tile += stride; if (tile >= tt.t_hi) break;
      DSGD_CT(C, B); tile += stride; if (tile >= tt.t_hi) break;
      DSGD_CT(D, C); tile += stride; if (tile >= tt.t_hi) break;
    }
#undef DSGD_CT
  }
  if (GRAD) {
    __syncthreads();
    int* mine = partc + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * partc_stride;
    wg_copy_out(mine, reinterpret_cast<int*>(lds), nc_lds, tid, 1024, is_aligned16(mine));
  }
}
