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
//   dsgd_tc_dot_kernel    one 16-lane group per row, 64 consecutive rows per workgroup: x.w with the 4,096 hottest
//                         weights (79 % of RCV1-like non-zeros) from a 16 KB LDS copy and the rest gathered from L2, the
//                         gate (core/ml/SparseSVM.scala:27-28), ONE BIT per row out -- the workgroup's 64 rows are two
//                         whole words of the step's bitmap: plain stores, nothing to clear between steps.
//   dsgd_tc_grad_kernel   the entries of the ranges' rows SORTED BY (worker, column) -- 8 bytes each: {bit of the row,
//                         column relative to the share's first} packed, y * value -- cut into equal shares, one per
//                         workgroup.  The bitmap (rows / 8 bytes) is copied into LDS; a share covers a contiguous run of
//                         columns: its fixed-point sums live in an LDS table indexed by the relative column, exact
//                         64-bit integers (no bound to derive), a wave whose 64 entries fall into ONE column (the hot
//                         head) adds its sum once.  A column wholly inside a share is written to the worker's 64-bit
//                         global accumulator by that share alone; a column cut by a share boundary gets one atomic per
//                         share.  No partials: D words leave the launch, not workgroups x D.  Workgroup 0 counts the
//                         bitmap's bits: the step's active rows, one atomic.
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
constexpr int TC_ROWS_PER_WG = TC_THREADS / TC_G;   // 64: two words of the bitmap
constexpr int TC_MAX_SHARE = 8192;       // entries per workgroup of the gradient kernel: 64 KB of LDS for its 64-bit table
constexpr int TC_DC_BITS = 13;           // packed entry: relative column (< TC_MAX_SHARE) below, bit of the row above
constexpr int TC_MAX_BITS = 1 << (32 - TC_DC_BITS);   // 524,288 bits of bitmap (every worker's range padded to 64 rows)
constexpr int TC_WL = 4096;              // hottest ranks with an LDS copy of their weight in the dot kernel (one 16-byte piece per lane)

struct TcShare {     // per share of the gradient kernel
  unsigned int cid0; // compact id of its first column
  int n_local;       // columns it touches
};

// ---- the layout: count, scan, shares, fill ---------------------------------------------------------------------------
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

// ONE workgroup: ptr[key] = cursor[key] = entries in front of key's first (ptr[n_keys] = all); cid[key] = non-empty keys in
// front of it; totals = {entries, non-empty keys}
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_scan_kernel(const unsigned int* __restrict__ cnt, int n_keys,
                                                                 unsigned int* __restrict__ ptr, unsigned int* __restrict__ cursor,
                                                                 int* __restrict__ cid, unsigned long long* __restrict__ totals) {
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
  // inclusive prefix over the 1024 per-thread totals (Hillis-Steele in LDS: once per configuration)
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
    ptr[k] = (unsigned int)be;
    cursor[k] = (unsigned int)be;
    cid[k] = (int)bn;
    be += c;
    bn += c != 0u;
  }
  if (tid == TC_THREADS - 1) {
    ptr[n_keys] = (unsigned int)se[tid];
    totals[0] = se[tid];
    totals[1] = sn[tid];
  }
}

// the key that holds entry position `pos`: the LAST key with ptr[key] <= pos (empty keys share their ptr with the next)
__device__ __forceinline__ int tc_key_at(const unsigned int* __restrict__ ptr, int n_keys, unsigned int pos) {
  int lo = 0, hi = n_keys;   // first key with ptr[key] > pos
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] > pos) hi = mid;
    else lo = mid + 1;
  }
  return lo - 1;
}
// one lane per share: the compact ids of its first and last column
__global__ void __launch_bounds__(256) dsgd_tc_shares_kernel(const unsigned int* __restrict__ ptr, const int* __restrict__ cid,
                                                            int n_keys, long long n_ent, int share, int n_shares,
                                                            TcShare* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_shares) return;
  const long long e0 = (long long)b * share;
  const long long e1 = e0 + share < n_ent ? e0 + share : n_ent;
  const int c0 = cid[tc_key_at(ptr, n_keys, (unsigned int)e0)], c1 = cid[tc_key_at(ptr, n_keys, (unsigned int)(e1 - 1))];
  TcShare r;
  r.cid0 = (unsigned int)c0;
  r.n_local = c1 - c0 + 1;
  out[b] = r;
}

// the entries to their places: {bit of the row << 13 | column relative to the share's first, y * value};
// key_of_cid[id] = worker * dp + column.  bit_base[k]: first bit of worker k's rows in the step's bitmap.
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_fill_kernel(CsrView m, const WorkSeg* __restrict__ segs, int dp,
                                                                 unsigned int* __restrict__ cursor, const int* __restrict__ cid,
                                                                 const TcShare* __restrict__ shares, int share,
                                                                 const int* __restrict__ bit_base,
                                                                 unsigned int* __restrict__ ent_pk, float* __restrict__ ent_val,
                                                                 int* __restrict__ key_of_cid) {
  const WorkSeg sg = segs[blockIdx.y];
  const int sub = threadIdx.x & (TC_G - 1);
  const long long group = ((long long)blockIdx.x * TC_THREADS + threadIdx.x) / TC_G;
  const long long n_groups = (long long)gridDim.x * TC_ROWS_PER_WG;
  const long long kbase = (long long)blockIdx.y * dp;
  const unsigned int bb = (unsigned int)bit_base[blockIdx.y];
  for (long long row = sg.begin + group; row < sg.end; row += n_groups) {
    const float y = (float)m.label[row];
    const unsigned int bit = bb + (unsigned int)(row - sg.begin);
    for (long long p = m.row_ptr[row] + sub; p < m.row_ptr[row + 1]; p += TC_G) {
      const long long key = kbase + m.col[p];
      const unsigned int at = atomicAdd(&cursor[key], 1u);
      const int id = cid[key];
      const unsigned int dc = (unsigned int)id - shares[at / (unsigned int)share].cid0;
      ent_pk[at] = (bit << TC_DC_BITS) | dc;
      ent_val[at] = y * m.val[p];          // (+-1 times a value: exact)
      key_of_cid[id] = (int)key;           // (every entry of the column stores the same word)
    }
  }
}

// ---- pass 1: x.w and the gate, one bit per row ----------------------------------------------------------------------
// ref: math/Vec.scala:58 -> math/Sparse.scala:46 (the products filtered at 1e-20), core/ml/SparseSVM.scala:27-28 (the gate)
// grid (x, workers); workgroup (x, k) owns the blocks x, x + gridDim.x, ... of ROWS = 32 or 64 rows of worker k: block b =
// the rows sg.begin + ROWS b .. = whole words of the bitmap from bit_base[k] / 32 + (ROWS / 32) b (bits beyond the range's last
// row: zero; every worker's rows are padded to 64).
// (A persistent form -- two workgroups per CU walking their blocks with the next block's row records and non-zeros
//  requested ahead, LDS-only barriers -- was measured SLOWER: 10.0 -> 20.4 us at 18,519 rows, 22 -> 52 at 80,441.  The
//  -- not understood; the evidence is in profiles/r05_tcol_probe_v9_*.json.)
// What the 9-10 us of this kernel at 18,519 rows are (profiles/r05_tcol_dot_decomposition.txt): NOT its requests -- with
// four of a row's five rounds of non-zeros left out it takes 9.1 us, with every gather from L2 left out 9.3 -- but the launch
// and ONE chain of dependent round trips per workgroup (row records -> non-zeros -> weights outside the LDS copy -> the
// bitmap word); at 80,441 rows (1,257 workgroups, two resident per CU) 2.5 such lifetimes one after the other: 22 us.
// Workgroups of 512 lanes (32 rows = one word) were measured three times SLOWER (33 us at 18,519 rows); so was a
// persistent form with the next block's requests in flight (below).
template <int NT>   // NT lanes of a workgroup: 1024 (64 rows = two words of the bitmap)
__global__ void __launch_bounds__(NT) dsgd_tc_dot_kernel(CsrView m, const float* __restrict__ w,
                                                        const WorkSeg* __restrict__ segs, const int* __restrict__ bit_base,
                                                        unsigned int* __restrict__ bitmap, int wl) {
  constexpr int ROWS = NT / TC_G, WORDS = ROWS / 32;
  __shared__ __attribute__((aligned(16))) float wlds[TC_WL];
  __shared__ unsigned int mask[2];
  const WorkSeg sg = segs[blockIdx.y];
  const int tid = threadIdx.x, sub = tid & (TC_G - 1), grp = tid >> 4;
  unsigned int* words = bitmap + (bit_base[blockIdx.y] >> 5);
  // the hottest weights: 16-byte pieces, requested first, stored behind the row's own requests
  float4 wpiece[TC_WL / 4 / NT];
#pragma unroll
  for (int i = 0; i < TC_WL / 4 / NT; ++i) wpiece[i] = reinterpret_cast<const float4*>(w)[4 * (tid + i * NT) < wl ? tid + i * NT : 0];
  bool staged = false;
  typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
  for (long long blk = blockIdx.x; blk * ROWS < sg.end - sg.begin; blk += gridDim.x) {
    const long long row = sg.begin + blk * ROWS + grp;
    const bool in_range = row < sg.end;
    const long long rowc = in_range ? row : sg.end - 1;
    // every request unconditional, positions clamped into the row (the internal CSR holds no empty row), a lane's surplus
    // values zeroed: behind a conditional load the compiler waits for each gather on its own -- five dependent L2 round
    // trips per row instead of one
    const long long st = m.row_ptr[rowc], en = m.row_ptr[rowc + 1];
    const float y = (float)m.label[rowc];
    if (tid < WORDS) mask[tid] = 0u;
    if (!staged) {
#pragma unroll
      for (int i = 0; i < TC_WL / 4 / NT; ++i)
        if (4 * (tid + i * NT) < wl) reinterpret_cast<float4*>(wlds)[tid + i * NT] = wpiece[i];
      staged = true;
    }
    __syncthreads();
    float acc = 0.0f;
    for (long long p0 = st + sub; p0 - sub < en; p0 += TC_UNR * TC_G) {   // (one round for rows of up to 80 non-zeros)
      int cc[TC_UNR];
      float vv[TC_UNR], wg[TC_UNR], wh[TC_UNR];
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) {
        const long long p = p0 + k * TC_G;
        const long long pc = p < en ? p : en - 1;
        cc[k] = m.col[pc];
        vv[k] = m.val[pc];
        vv[k] = p < en ? vv[k] : 0.0f;
      }
      // hot ranks from the LDS copy, the rest from L2 -- with UNCONDITIONAL global loads (the hot lanes all read w[0]: one
      // line): a fully divergent 4-byte gather costs the texture path 64 cycles per wave instruction
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) wg[k] = w[cc[k] < wl ? 0 : cc[k]];
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) wh[k] = ((lds_cvfloat*)wlds)[cc[k] < wl ? cc[k] : 0];
#pragma unroll
      for (int k = 0; k < TC_UNR; ++k) acc += filt(vv[k] * (cc[k] < wl ? wh[k] : wg[k]));   // ref: math/Sparse.scala:46
    }
    const float d = group_sum<TC_G>(acc);
    const float yd = y * d;
    const bool active = in_range && !(yd < 0.0f);
    if (sub == 0 && active) atomicOr(&mask[grp >> 5], 1u << (grp & 31));
    __syncthreads();
    if (tid < WORDS) words[WORDS * blk + tid] = mask[tid];
  }
}

// ---- pass 2: the gradient, column by column -------------------------------------------------------------------------
// ref: core/Slave.scala:147-153 (Vec.sum of the gated sub-gradients), column-major
struct TcGradArgs {
  const unsigned int* ent_pk;
  const float* ent_val;
  const TcShare* shares;
  const int* key_of_cid;
  const unsigned int* bitmap;
  long long* g64;            // [workers][dp] 64-bit fixed-point accumulators, zero between steps: index = key
  DevScalars* sc;
  long long n_ent;
  int share;                 // entries per workgroup (<= TC_MAX_SHARE)
  int bm_words;              // words of the bitmap (a multiple of 4)
  float scale;               // 2^shift / vmax2
};

// NP: 4-entry pieces per lane (a share holds at most 4096 NP entries).  A lane owns 4 NP CONSECUTIVE entries of the sorted
// list: two 16-byte requests per piece (a quarter of the load instructions of one entry per lane and request -- the
// kernel sits at the texture addresser's instruction rate, not at bytes), and a lane adds a run of equal columns up in a
// register before it touches the table: one LDS add per run and lane (in the head of the ranking: per lane; per WAVE when
// all of its lanes end in the same column).
template <int NP>
__global__ void __launch_bounds__(TC_THREADS) dsgd_tc_grad_kernel(TcGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) long long tc_tab[];   // [share], then the bitmap, then 16 words
  unsigned int* bm = reinterpret_cast<unsigned int*>(tc_tab + a.share);
  unsigned int* red = bm + a.bm_words;   // [16] (in the dynamic allocation: the launch sizes it)
  const int tid = threadIdx.x;
  const long long e0 = (long long)blockIdx.x * a.share;
  const long long e1 = e0 + a.share < a.n_ent ? e0 + a.share : a.n_ent;
  if (e0 >= e1) return;
  constexpr int NR = 4 * NP;
  uint4 pk4[NP];
  float4 val4[NP];
  // (the arrays are padded to whole pieces: a piece beyond the share's end is requested -- clamped to the list's last
  //  piece -- and its entries masked by their positions)
  const long long last_piece = ((a.n_ent + 3) >> 2) - 1;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    long long piece = (e0 >> 2) + (long long)tid * NP + p;
    piece = piece < last_piece ? piece : last_piece;
    pk4[p] = reinterpret_cast<const uint4*>(a.ent_pk)[piece];
    val4[p] = reinterpret_cast<const float4*>(a.ent_val)[piece];
  }
  const TcShare sh = a.shares[blockIdx.x];
  const int n_local = sh.n_local;
  // where the table's sums go: requested now, used behind the adds
  int key[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int j = tid + u * TC_THREADS;
    key[u] = a.key_of_cid[sh.cid0 + (unsigned int)(j < n_local ? j : n_local - 1)];
  }
  // the step's gate decisions: rows / 8 bytes, every workgroup its own copy
  unsigned int n_act = 0;
  {
    const uint4* b4 = reinterpret_cast<const uint4*>(a.bitmap);
    const int n4 = a.bm_words >> 2;
    for (int i = tid; i < n4; i += TC_THREADS) {
      const uint4 v = b4[i];
      reinterpret_cast<uint4*>(bm)[i] = v;
      n_act += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
  }
  for (int j = tid; j < n_local; j += TC_THREADS) tc_tab[j] = 0;
  if (blockIdx.x == 0) {   // (workgroup-uniform) the step's active rows: one atomic
    n_act = wave_sum_u32(n_act);
    if ((tid & 63) == 0) red[tid >> 6] = n_act;
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid == 0) {
    unsigned int t = 0;
    for (int i = 0; i < TC_THREADS / 64; ++i) t += red[i];
    if (t) atomicAdd(&a.sc->n_active, (unsigned long long)t);
  }
  const long long ebase = e0 + (long long)tid * NR;
  int run_dc = -1, run = 0;
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const unsigned int pk = (&pk4[u >> 2].x)[u & 3];
    const float val = (&val4[u >> 2].x)[u & 3];
    const int dc = (int)(pk & ((1u << TC_DC_BITS) - 1u));
    const unsigned int bit = pk >> TC_DC_BITS;
    const bool on = ((bm[bit >> 5] >> (bit & 31u)) & 1u) != 0u;
    const int q = (ebase + u < e1 && on) ? __float2int_rn(val * a.scale) : 0;   // (4 NP x 2^21 < 2^31)
    if (u > 0 && dc != run_dc && ebase + u < e1) {   // the lane's run of one column ends: its sum into the table
      if (run != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&tc_tab[run_dc]), (unsigned long long)(long long)run);
      run = 0;
    }
    if (u == 0 || ebase + u < e1) run_dc = dc;
    run += q;
  }
  {
    // the lane's last run.  A wave inside ONE column (the head of the ranking: thousands of entries per column): one add for
    // all of its lanes.  (Lanes beyond the share's end carry run = 0.)
    const bool mine = ebase < e1;
    const int dcf = __builtin_amdgcn_readfirstlane(run_dc);
    if (__all(!mine || run == 0 || run_dc == dcf)) {
      long long s = mine ? (long long)run : 0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
      // (lane 0 may itself be beyond the end or hold a different, empty run: the common column is any contributing lane's)
      const unsigned long long has = __ballot(mine && run != 0);
      if (has) {
        const int src = __builtin_ctzll(has);
        const int dcc = __shfl(run_dc, src, 64);
        if ((tid & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&tc_tab[dcc]), (unsigned long long)s);
      }
    } else if (mine && run != 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&tc_tab[run_dc]), (unsigned long long)(long long)run);
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
