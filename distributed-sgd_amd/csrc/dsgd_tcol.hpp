// Device code of libdsgd_hip, part 6 (gfx950 only): "column lists" -- whole-split steps of 10^3 .. 10^5 rows.
// Included by dsgd_hip.hip after dsgd_fstep.hpp.
//
// ref: core/Master.scala:179-199 with batch-size >= the split (a batch is a worker's whole split; a sum does not depend on
// the order), core/Slave.scala:142-157 (the worker's regularised sum); the sizes are the reference's own: N = 23,149
// (application.conf:24, `full = false`) and what one GPU of eight holds of RCV1's 804,414 rows (core/ml/SplitStrategy.scala:13-14).
//
// Every ROW-parallel form of such a step pays "workgroups x D": each workgroup sets up tens of KB of weights in LDS and
// writes D fixed-point words of partials that a second launch adds up again -- 18,519 rows are 11 MB of matrix, their
// row-wise step moved 25 MB of partials (37 us, 0.04 of the HBM roofline), their chunked step (dsgd_fstep.hpp) 7 us of
// set-ups per workgroup for 10 us of tiles.  A whole-split step repeats the SAME rows every time, so the transposed
// matrix can be laid out once per (ranges) configuration and the gradient taken COLUMN by column, where a sum is local:
//
//   dsgd_tc_dot_kernel    one 16-lane group per row: x.w with the weights gathered from L2 (189 KB: resident), the
//                         gate (core/ml/SparseSVM.scala:27-28), ONE byte per row out: active or not.  No LDS, no set-up.
//   dsgd_tc_grad_kernel   the entries of the ranges' rows SORTED BY (worker, column) -- {row, y * value, column id} --
//                         cut into equal shares, one per workgroup.  A share covers a contiguous run of columns: its
//                         fixed-point sums live in an LDS table indexed by (column - first column of the share), exact
//                         64-bit integers (no bound to derive), a wave whose 64 entries fall into ONE column (the hot
//                         head) adds its sum once.  A column wholly inside a share is written to the worker's 64-bit
//                         global accumulator by that share alone; a column cut by a share boundary gets one atomic per
//                         share.  No partials: D words leave the launch, not workgroups x D.
//   dsgd_fix_reduce_apply_kernel (as behind every other gradient kernel, with zero partials): one rounding of each exact
//                         sum, the support-only regulariser, the fold over the workers, mean, update, the next scalar.
//
// Same fixed-point grid as every other path (round(y x 2^shift / vmax2), shift = the context's cap: 21): the integer
// sums are those the row-parallel kernels would have formed -- same gate decisions => the same bits.
// The layout is built by the device (count -> scan -> fill; the order inside a column is whatever the fill's atomics
// produce: integer sums do not care) once per configuration and cached by the host (dsgd_hip.hip: tcol_layout).
#pragma once

#include "dsgd_kernels.hpp"

constexpr int TC_G = 16;                 // lanes per row
constexpr int TC_UNR = 5;                // register-held rounds of a row: 80 non-zeros (RCV1's mean row: 75)
constexpr int TC_THREADS = 1024;
constexpr int TC_ROWS_PER_WG = TC_THREADS / TC_G;
constexpr int TC_MAX_SHARE = 8192;       // entries per workgroup of the gradient kernel: 64 KB of LDS for its 64-bit table

// ---- the layout: count, scan, fill ---------------------------------------------------------------------------------
// grid (x, workers): a 16-lane group per row of worker blockIdx.y's range
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_count_kernel(CsrView m, const WorkSeg* __restrict__ segs, int dp,
                                                                  unsigned int* __restrict__ cnt) {
  const WorkSeg sg = segs[blockIdx.y];
  const int sub = threadIdx.x & (TC_G - 1);
  const long long group = ((long long)blockIdx.x * TC_THREADS + threadIdx.x) / TC_G;
  const long long n_groups = (long long)gridDim.x * TC_ROWS_PER_WG;
  unsigned int* mine = cnt + (long long)blockIdx.y * dp;
  for (long long row = sg.begin + group; row < sg.end; row += n_groups)
    for (long long p = m.row_ptr[row] + sub; p < m.row_ptr[row + 1]; p += TC_G) atomicAdd(&mine[m.col[p]], 1u);
}

// ONE workgroup: cursor[key] = entries in front of key's first; cid[key] = non-empty keys in front of it; totals = {entries, non-empty keys}
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_scan_kernel(const unsigned int* __restrict__ cnt, int n_keys,
                                                                 unsigned int* __restrict__ cursor, int* __restrict__ cid,
                                                                 unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long se[TC_THREADS];
  __shared__ unsigned int sn[TC_THREADS];
  const int tid = threadIdx.x;
  const int chunk = (n_keys + TC_THREADS - 1) / TC_THREADS;
  const int k0 = tid * chunk, k1 = min(n_keys, k0 + chunk);
  unsigned long long e = 0;
  unsigned int n = 0;
  for (int k = k0; k < k1; ++k) {
    const unsigned int c = cnt[k];
    e += c;
    n += c != 0u;
  }
  se[tid] = e;
  sn[tid] = n;
  __syncthreads();
  // exclusive prefix over the 1024 per-thread totals (Hillis-Steele in LDS: once per configuration)
  for (int off = 1; off < TC_THREADS; off <<= 1) {
    unsigned long long ae = 0;
    unsigned int an = 0;
    if (tid >= off) {
      ae = se[tid - off];
      an = sn[tid - off];
    }
    __syncthreads();
    se[tid] += ae;
    sn[tid] += an;
    __syncthreads();
  }
  unsigned long long be = se[tid] - e;
  unsigned int bn = sn[tid] - n;
  for (int k = k0; k < k1; ++k) {
    const unsigned int c = cnt[k];
    cursor[k] = (unsigned int)be;
    cid[k] = (int)bn;
    be += c;
    bn += c != 0u;
  }
  if (tid == TC_THREADS - 1) {
    totals[0] = se[tid];
    totals[1] = sn[tid];
  }
}

// the entries to their places: {row, y * value, compact column id}; key_of_cid[id] = worker * dp + column
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_fill_kernel(CsrView m, const WorkSeg* __restrict__ segs, int dp,
                                                                 unsigned int* __restrict__ cursor, const int* __restrict__ cid,
                                                                 int* __restrict__ ent_row, float* __restrict__ ent_val,
                                                                 unsigned int* __restrict__ ent_cid, int* __restrict__ key_of_cid) {
  const WorkSeg sg = segs[blockIdx.y];
  const int sub = threadIdx.x & (TC_G - 1);
  const long long group = ((long long)blockIdx.x * TC_THREADS + threadIdx.x) / TC_G;
  const long long n_groups = (long long)gridDim.x * TC_ROWS_PER_WG;
  const long long kbase = (long long)blockIdx.y * dp;
  for (long long row = sg.begin + group; row < sg.end; row += n_groups) {
    const float y = (float)m.label[row];
    for (long long p = m.row_ptr[row] + sub; p < m.row_ptr[row + 1]; p += TC_G) {
      const long long key = kbase + m.col[p];
      const unsigned int at = atomicAdd(&cursor[key], 1u);
      const int id = cid[key];
      ent_row[at] = (int)row;
      ent_val[at] = y * m.val[p];          // (+-1 times a value: exact)
      ent_cid[at] = (unsigned int)id;
      key_of_cid[id] = (int)key;           // (every entry of the column stores the same word)
    }
  }
}

// ---- pass 1: x.w and the gate, one byte per row ---------------------------------------------------------------------
// ref: math/Vec.scala:58 -> math/Sparse.scala:46 (the products filtered at 1e-20), core/ml/SparseSVM.scala:27-28 (the gate)
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_dot_kernel(CsrView m, const float* __restrict__ w,
                                                                const WorkSeg* __restrict__ segs, signed char* __restrict__ act,
                                                                DevScalars* __restrict__ sc) {
  __shared__ unsigned int wg_active;
  const WorkSeg sg = segs[blockIdx.y];
  const int sub = threadIdx.x & (TC_G - 1);
  const long long group = ((long long)blockIdx.x * TC_THREADS + threadIdx.x) / TC_G;
  const long long n_groups = (long long)gridDim.x * TC_ROWS_PER_WG;
  if (threadIdx.x == 0) wg_active = 0u;
  __syncthreads();
  unsigned int n_act = 0;
  for (long long row = sg.begin + group; row < sg.end; row += n_groups) {
    // every request unconditional, positions clamped into the row (the internal CSR holds no empty row), a lane's surplus
    // values zeroed: behind a conditional load the compiler waits for each gather on its own -- five dependent L2 round
    // trips per row instead of one
    const long long st = m.row_ptr[row], en = m.row_ptr[row + 1];
    const float y = (float)m.label[row];
    float acc = 0.0f;
    for (long long p0 = st + sub; p0 - sub < en; p0 += TC_UNR * TC_G) {   // (one round for rows of up to 80 non-zeros)
      int cc[TC_UNR];
      float vv[TC_UNR], ww[TC_UNR];
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) {
        const long long p = p0 + k * TC_G;
        const long long pc = p < en ? p : en - 1;
        cc[k] = m.col[pc];
        vv[k] = m.val[pc];
        vv[k] = p < en ? vv[k] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) ww[k] = w[cc[k]];
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) acc += filt(vv[k] * ww[k]);   // ref: math/Sparse.scala:46 (products filtered at 1e-20)
    }
    const float d = group_sum<TC_G>(acc);
    const float yd = y * d;
    const bool active = !(yd < 0.0f);
    if (sub == 0) {
      act[row] = active ? 1 : 0;
      n_act += active ? 1u : 0u;
    }
  }
  n_act = wave_sum_u32(n_act);
  if ((threadIdx.x & 63) == 0 && n_act) atomicAdd(&wg_active, n_act);
  __syncthreads();
  if (threadIdx.x == 0 && wg_active) atomicAdd(&sc->n_active, (unsigned long long)wg_active);   // one atomic per workgroup
}

// ---- pass 2: the gradient, column by column -------------------------------------------------------------------------
// ref: core/Slave.scala:147-153 (Vec.sum of the gated sub-gradients), column-major
struct TcGradArgs {
  const int* ent_row;
  const float* ent_val;
  const unsigned int* ent_cid;
  const int* key_of_cid;
  const signed char* act;
  long long* g64;            // [workers][dp] 64-bit fixed-point accumulators, zero between steps: index = key
  long long n_ent;
  int share;                 // entries per workgroup (<= TC_MAX_SHARE)
  float scale;               // 2^shift / vmax2
};

// NR: rounds of 1024 entries a share holds at most.  No loop: ALL of a lane's entries are requested at once, then all of
// their rows' gate bytes -- two dependent round trips per workgroup, whatever the share.
template <int NR>
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_grad_kernel(TcGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) long long tc_tab[];   // [share]
  const int tid = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * a.share;
  const long long e1 = e0 + a.share < a.n_ent ? e0 + a.share : a.n_ent;
  if (e0 >= e1) return;
  int row[NR];
  float val[NR];
  unsigned int cid[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {   // (clamped: a lane beyond the share re-reads its last entry and contributes nothing)
    const long long e = e0 + tid + (long long)u * TC_THREADS;
    const long long ec = e < e1 ? e : e1 - 1;
    row[u] = a.ent_row[ec];
    val[u] = a.ent_val[ec];
    cid[u] = a.ent_cid[ec];
  }
  const unsigned int cid0 = a.ent_cid[e0], cid1 = a.ent_cid[e1 - 1];   // (sorted by column id: the share's first and last)
  int on[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) on[u] = (int)a.act[row[u]];
  const int n_local = (int)(cid1 - cid0) + 1;
  for (int j = tid; j < n_local; j += TC_THREADS) tc_tab[j] = 0;
#pragma unroll
  for (int u = 0; u < NR; ++u) asm volatile("" : "+v"(on[u]));   // (the gate bytes stay requested together, up there)
  __syncthreads();
  // where the table's sums go: requested now, used behind the adds (in the write-out loop each would be a round trip of
  // its own, behind the atomics of the round before)
  int key[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int j = tid + u * TC_THREADS;
    key[u] = a.key_of_cid[cid0 + (unsigned int)(j < n_local ? j : n_local - 1)];
  }
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const long long e = e0 + tid + (long long)u * TC_THREADS;
    const int dc = (int)(cid[u] - cid0);
    const int q = (e < e1 && on[u]) ? __float2int_rn(val[u] * a.scale) : 0;
    // a wave inside ONE column (the head of the ranking: thousands of entries per column): one add for its 64 entries
    const int dcf = __builtin_amdgcn_readfirstlane(dc);
    if (__all(dc == dcf || q == 0)) {
      int s = q;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);   // (64 x 2^21 < 2^31)
      if ((tid & 63) == 0 && s != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&tc_tab[dcf]), (unsigned long long)(long long)s);
    } else if (q != 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&tc_tab[dc]), (unsigned long long)(long long)q);
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int j = tid + u * TC_THREADS;
    const long long v = j < n_local ? tc_tab[j] : 0;
    // first / last column of a share may continue in the neighbouring shares: atomics (exact: integers); all others are
    // this share's alone -- the same instruction, uncontended
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&a.g64[key[u]]), (unsigned long long)v);
  }
}
