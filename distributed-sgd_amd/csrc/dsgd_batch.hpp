// Device code of libdsgd_hip, part 2 (gfx950 only): the MINI-BATCH engine and the two persistent kernels built on
// it -- the lock-free "Hogwild" engine (BASELINE.json configs[3]) and the small-batch plan kernel (the reference's
// batch-size 100-200).  Included by dsgd_hip.hip after dsgd_kernels.hpp.
//
// Citations "ref:" are relative to /root/reference/src/main/scala/epfl/distributed/.
#pragma once

// ======================================================================================================
// Mini-batch engine: ONE workgroup computes a whole mini-batch (gather-dot, gate, batch sum)
// ======================================================================================================
// Both users are LATENCY problems (a batch of 100 rows is 60 KB of CSR scattered over gigabytes), so the work is a
// fixed number of dependent memory round trips per BATCH, whatever the row lengths, and every stage is a separate
// function so that the callers can put the NEXT batch's round trips under the current batch's update sweep:
//   bt_rows_issue   one thread per row: row_ptr / label requested (registers; nothing waits)
//   bt_build        workgroup scan over the rows' chunk counts (128 non-zeros per chunk) -> a table of WORK ITEMS
//                   (row, chunk) in LDS; as many rows as fit the NG x R item slots form a sub-batch (100 RCV1-like
//                   rows are ~118 items)
//   bt_items_issue  a group of 16 lanes per item, R items per group: 8 + 8 loads per lane (col, val) requested
//   bt_items_dot    the 8 weights per lane and item, products, DPP butterfly -> partial x.w of the item in LDS
//   bt_gate         one thread per row adds the row's partials in chunk order (fixed order: x.w is reproducible),
//                   gates (core/ml/SparseSVM.scala:27-28) and leaves y or 0 as the row's coefficient
//   bt_scatter      every item -- its non-zeros are STILL IN REGISTERS -- adds coefficient * x to a fixed-point LDS
//                   accumulator (ranks < hl: ds_add_u32, exact, order-independent) or to the workgroup's private
//                   global strip plus an LDS bitmap (the few ranks >= hl)
// The caller's sweep over the accumulators turns the batch sum into the update.  No second pass over the CSR.
// Fixed point: q = round(y*x * 2^shift / vmax2), shift = 30 - ceil(log2 batch): a column receives at most one
// contribution per row, so no 32-bit word can pass 2^30; contributions below half a grid unit vanish (this
// absorbs the reference's 1e-20 filter on y*x, math/Vec.scala:42 -> math/Sparse.scala:108-118).
constexpr int HBIT_WORDS = 1024;   // COLD = 3: words of the bitmap of touched LDS accumulators (ranks hh + word + 1024 * bit)
constexpr int BT_G = 16;     // lanes per work item
constexpr int BT_K = 8;      // non-zeros per lane and item
constexpr int BT_CH = BT_G * BT_K;   // 128 non-zeros per item

struct BtLds {
  int* acc;              // hl fixed-point accumulators, zero between batches
  unsigned int* cbits;   // one bit per rank >= hl: the strip entry was touched by this batch
  long long* g64;        // COLD = 2: 64-bit fixed-point global accumulators of the ranks >= hl (indexed by rank)
  int hl;
  unsigned int* hbits;   // COLD = 3: one bit per rank in [hh, hl): the LDS accumulator was touched by this batch
  int hh;
  // sub-batch tables: cap = item slots (NG x R) = most rows of a sub-batch
  long long* rst;        // [cap] first non-zero of the row
  int* rlen;             // [cap] its length
  float* rcoef;          // [cap] label, then (after the gate) label or 0
  int* ifirst;           // [cap] first work item of the row
  int* item_row;         // [cap]
  float* pdot;           // [cap] partial x.w per item
  int* misc;             // [56]: 16 wave sums, 16 wave counts of fitting rows, 16 wave item totals, scratch
};
__host__ __device__ constexpr int bt_lds_words(int cap) { return 7 * cap + 56; }
__device__ __forceinline__ void bt_carve(BtLds& L, int* base, int cap) {
  L.rst = reinterpret_cast<long long*>(base);   // (base is 8-byte aligned)
  L.rlen = base + 2 * cap;
  L.rcoef = reinterpret_cast<float*>(base + 3 * cap);
  L.ifirst = base + 4 * cap;
  L.item_row = base + 5 * cap;
  L.pdot = reinterpret_cast<float*>(base + 6 * cap);
  L.misc = base + 7 * cap;
}

// Requested data is kept RAW until its consumer runs: any arithmetic on a loaded value (a length = end - start, a
// masked select) makes the compiler wait for the load on the spot, which is exactly what staging a batch ahead must
// not do (the phase counters showed the "request" phase of the plan kernel waiting out a full HBM round trip until the
// masks and differences moved into the consumers).
template <int R>
struct BtItems {
  int c[R][BT_K];     // as loaded: slots k >= cnt[r] hold whatever follows the row
  float v[R][BT_K];
  int cnt[R];         // valid slots of this lane (<= 0: none)
  int irow[R];        // row of the sub-batch (-1: unused slot)
};
struct BtRow {
  long long st, en;   // row_ptr[row], row_ptr[row + 1] as loaded
  signed char lab;    // label as loaded
  bool ok;            // the row id was valid (known without any load)
  __device__ __forceinline__ int len() const { return ok ? (int)(en - st) : 0; }   // (a skipped row has no items)
};

// contribution of one non-zero of an active row.  COLD says where the few ranks beyond the LDS accumulators go:
//   0  the workgroup's private fp32 strip + LDS bitmap, device-scope atomics
//   1  the same with workgroup-scope atomics (plan kernel, Hogwild: only this workgroup touches its strip -- the
//      atomics stay in this XCD's L2 instead of crossing the fabric)
//   2  64-bit fixed-point global accumulators on the same grid as the LDS ones (mid-size batches spread over many
//      workgroups: dsgd_fix_reduce_* adds them to the partial sums exactly)
//   3  as 1, and the LDS accumulators beyond the dense head [0, hh) are marked in a second bitmap (the lock-free engine:
//      its update walks the touched ranks instead of sweeping all of them)
template <int COLD>
__device__ __forceinline__ void bt_add(const BtLds& L, float* __restrict__ gcold, int c, float xv, float qscale) {
  if (c < L.hl) {
    const int q = __float2int_rn(xv * qscale);
    if (q != 0) {
      atomicAdd(&L.acc[c], q);   // ds_add_u32
      // (interleaved: word = low bits of the rank, so that every word -- every lane of the update -- holds its share of
      //  the dense ranks near the head and of the sparse ones far from it)
      if (COLD == 3 && c >= L.hh) atomicOr(&L.hbits[(unsigned int)(c - L.hh) & (HBIT_WORDS - 1)], 1u << ((unsigned int)(c - L.hh) / HBIT_WORDS));
    }
  } else if (COLD == 2) {
    const int q = __float2int_rn(xv * qscale);
    if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&L.g64[c]), (unsigned long long)(long long)q);
  } else {
    const float f = filt(xv);
    if (f != 0.0f) {
      if (COLD == 1 || COLD == 3) __hip_atomic_fetch_add(&gcold[c - L.hl], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else atomicAdd(&gcold[c - L.hl], f);
      atomicOr(&L.cbits[(unsigned int)(c - L.hl) >> 5], 1u << ((c - L.hl) & 31));
    }
  }
}

__device__ __forceinline__ int wave_incl_scan_i32(int v) {
  v += dpp_get_i<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get_i<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get_i<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get_i<0x118, 0xf>(v);   // row_shr:8
  v += dpp_get_i<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_get_i<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v;
}

// row records of the rows row_of(b0 .. b0 + min(CAP, B - b0) - 1), one per thread; loads only
template <int CAP, class RowOf>
__device__ __forceinline__ BtRow bt_rows_issue(const CsrView& m, int B, int b0, RowOf row_of, int* bad) {
  BtRow r;
  r.st = 0;
  r.en = 0;
  r.lab = 0;
  r.ok = false;
  const int tid = threadIdx.x;
  if (tid < min(CAP, B - b0)) {
    long long row = row_of(b0 + tid);
    r.ok = row >= 0 && row < m.n_rows;
    if (!r.ok) {
      atomicOr(bad, 1);
      row = 0;
    }
    r.st = m.row_ptr[row];
    r.en = m.row_ptr[row + 1];
    r.lab = m.label[row];
  }
  return r;
}

// tables of the sub-batch in three parts around two workgroup barriers (a caller with barriers of its own in the
// right places shares them): p1 ... barrier ... p2 ... barrier ... p3 returns {rows that fit the item slots (a
// prefix), work items}.  p1/p2 write L.misc and (p2) the tables of L; nothing else.
struct BtScan {
  int nch, incl;
};
template <int THREADS, int CAP>
__device__ __forceinline__ BtScan bt_build_p1(const BtLds& L, int B, int b0, const BtRow& row) {
  static_assert(CAP <= THREADS, "one thread per row of a sub-batch");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = min(CAP, B - b0);
  BtScan sc;
  sc.nch = tid < nb ? (row.len() + BT_CH - 1) / BT_CH : 0;
  sc.incl = wave_incl_scan_i32(sc.nch);
  if (lane == 63) L.misc[wave] = sc.incl;
  return sc;
}
template <int THREADS, int CAP>
__device__ __forceinline__ void bt_build_p2(const BtLds& L, int B, int b0, const BtRow& row, const BtScan& sc) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = min(CAP, B - b0);
  int first = sc.incl - sc.nch;
  for (int i = 0; i < wave; ++i) first += L.misc[i];
  const bool fits = tid < nb && first + sc.nch <= CAP;   // monotone in tid: the fitting rows are a prefix
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(fits);
  // items of the sub-batch = end of the LAST fitting row (the prefix sums are monotone).  (A shared-word atomicMax by
  // every row's thread was a 64-way LDS conflict: 4,500 of 33,500 cycles per batch in the phase counters.)
  const int last = bal ? 63 - __builtin_clzll(bal) : 0;
  const int wave_items = bal ? __builtin_amdgcn_readlane(first + sc.nch, last) : 0;
  if (lane == 0) {
    L.misc[16 + wave] = __popcll(bal);
    L.misc[32 + wave] = wave_items;
  }
  if (fits) {
    L.rst[tid] = row.st;
    L.rlen[tid] = row.len();
    L.rcoef[tid] = (float)row.lab;
    L.ifirst[tid] = first;
    for (int c = 0; c < sc.nch; ++c) L.item_row[first + c] = tid;
  } else if (tid == 0) {   // the first row alone exceeds the item slots: bt_giant_row
    L.rst[0] = row.st;
    L.rlen[0] = row.len();
    L.rcoef[0] = (float)row.lab;
  }
}
template <int THREADS>
__device__ __forceinline__ int2 bt_build_p3(const BtLds& L) {
  int nbf = 0, items = 0;
  for (int i = 0; i < THREADS / 64; ++i) {
    nbf += L.misc[16 + i];
    items = max(items, L.misc[32 + i]);
  }
  return make_int2(nbf, items);
}
template <int THREADS, int CAP>
__device__ __forceinline__ int2 bt_build(const BtLds& L, int B, int b0, const BtRow& row) {
  const BtScan sc = bt_build_p1<THREADS, CAP>(L, B, b0, row);
  __syncthreads();
  bt_build_p2<THREADS, CAP>(L, B, b0, row, sc);
  __syncthreads();
  return bt_build_p3<THREADS>(L);
}

template <int THREADS, int R>
__device__ __forceinline__ void bt_items_issue(const CsrView& m, const BtLds& L, int n_items, BtItems<R>& it) {
  constexpr int NG = THREADS / BT_G;
  const int sub = threadIdx.x % BT_G, gidx = threadIdx.x / BT_G;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = r * NG + gidx;
    const bool valid = i < n_items;
    const int row = valid ? L.item_row[i] : 0;
    it.irow[r] = valid ? row : -1;
    const int ch = valid ? i - L.ifirst[row] : 0;
    const long long p0 = L.rst[row] + (long long)ch * BT_CH + sub * BT_K;
    const int cnt = (valid ? min(BT_CH, L.rlen[row] - ch * BT_CH) : 0) - sub * BT_K;   // of this lane's eight slots
    // lane `sub` owns EIGHT CONTIGUOUS non-zeros: two 16-byte loads per array (dword-aligned: rows start anywhere)
    // instead of eight 4-byte ones -- the request phase was bound by the number of load instructions (512 per batch
    // through one CU's texture addresser).  Slots past the row's end read the next row (or the arrays' padding) and
    // are masked.
    typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const i32x4u ca = *reinterpret_cast<const i32x4u*>(m.col + p0), cb = *reinterpret_cast<const i32x4u*>(m.col + p0 + 4);
    const f32x4u va = *reinterpret_cast<const f32x4u*>(m.val + p0), vb = *reinterpret_cast<const f32x4u*>(m.val + p0 + 4);
    it.cnt[r] = cnt;
    it.c[r][0] = ca.x; it.c[r][1] = ca.y; it.c[r][2] = ca.z; it.c[r][3] = ca.w;
    it.c[r][4] = cb.x; it.c[r][5] = cb.y; it.c[r][6] = cb.z; it.c[r][7] = cb.w;
    it.v[r][0] = va.x; it.v[r][1] = va.y; it.v[r][2] = va.z; it.v[r][3] = va.w;
    it.v[r][4] = vb.x; it.v[r][5] = vb.y; it.v[r][6] = vb.z; it.v[r][7] = vb.w;
  }
}

// partial x.w of every item -> L.pdot; `wload(c)` returns the weight of rank c
template <int THREADS, int R, class WLoad>
__device__ __forceinline__ void bt_items_dot(const BtLds& L, const BtItems<R>& it, WLoad wload) {
  constexpr int NG = THREADS / BT_G;
  const int sub = threadIdx.x % BT_G, gidx = threadIdx.x / BT_G;
  float wv[R][BT_K];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int k = 0; k < BT_K; ++k) wv[r][k] = wload(k < it.cnt[r] ? it.c[r][k] : 0);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < BT_K; ++k) acc += k < it.cnt[r] ? filt(it.v[r][k] * wv[r][k]) : 0.0f;   // ref: math/Sparse.scala:46
    acc = group_sum<BT_G>(acc);
    if (sub == 0 && it.irow[r] >= 0) L.pdot[r * NG + gidx] = acc;
  }
}

// x.w per row in chunk order, gate; returns this thread's count of active rows (0 or 1).
// gmask (LDS, may be null; traced runs of the lock-free engine): bit b0 + row of the mini-batch set for an active row;
// gdot (global, null unless gmask is set): the x.w the row was gated on, entry b0 + row.
__device__ __forceinline__ unsigned int bt_gate(const BtLds& L, int nbf, unsigned int* gmask = nullptr, int b0 = 0,
                                                float* gdot = nullptr) {
  const int tid = threadIdx.x;
  if (tid >= nbf) return 0u;
  const int f0 = L.ifirst[tid], n = (L.rlen[tid] + BT_CH - 1) / BT_CH;
  float d = 0.0f;
  for (int i = 0; i < n; ++i) d += L.pdot[f0 + i];
  const float yy = L.rcoef[tid];
  const bool active = L.rlen[tid] > 0 && !(yy * d < 0.0f);   // ref: core/ml/SparseSVM.scala:27-28
  L.rcoef[tid] = active ? yy : 0.0f;
  if (gmask && active) atomicOr(&gmask[(unsigned int)(b0 + tid) >> 5], 1u << ((b0 + tid) & 31));
  if (gdot) gdot[b0 + tid] = d;
  return active ? 1u : 0u;
}

template <int R, int COLD>
__device__ __forceinline__ void bt_scatter(const BtLds& L, float* __restrict__ gcold, const BtItems<R>& it, float qscale) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float coef = it.irow[r] >= 0 ? L.rcoef[it.irow[r]] : 0.0f;
    if (coef != 0.0f) {
#pragma unroll
      for (int k = 0; k < BT_K; ++k)
        if (k < it.cnt[r]) bt_add<COLD>(L, gcold, it.c[r][k], it.v[r][k] * coef, qscale);
    }
  }
}

// a single row longer than CAP x 128 non-zeros (never the case for RCV1): all threads share it.  The row record was
// left in slot 0 by bt_build.  Three workgroup barriers.
template <int THREADS, int COLD, class WLoad>
__device__ __forceinline__ unsigned int bt_giant_row(const CsrView& m, const BtLds& L, float* __restrict__ gcold,
                                                     WLoad wload, float qscale, unsigned int* gmask = nullptr, int b0 = 0,
                                                     float* gdot = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long s0 = L.rst[0];
  const int ln = L.rlen[0];
  const float yy = L.rcoef[0];
  float part = 0.0f;
  for (int p = tid; p < ln; p += THREADS) part += filt(m.val[s0 + p] * wload(m.col[s0 + p]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
  __syncthreads();
  if (lane == 0) L.pdot[wave] = part;
  __syncthreads();
  float d = 0.0f;
  for (int i = 0; i < THREADS / 64; ++i) d += L.pdot[i];
  unsigned int n_act = 0;
  if (gdot && tid == 0) gdot[b0] = d;
  if (!(yy * d < 0.0f)) {
    n_act = tid == 0 ? 1u : 0u;
    if (gmask && tid == 0) atomicOr(&gmask[(unsigned int)b0 >> 5], 1u << (b0 & 31));
    for (int p = tid; p < ln; p += THREADS) bt_add<COLD>(L, gcold, m.col[s0 + p], m.val[s0 + p] * yy, qscale);
  }
  __syncthreads();
  return n_act;
}

// The gated batch sum of the rows row_of(b0 .. B-1), stage after stage (nothing overlapped): the general path for
// whatever a pipelined caller could not stage ahead.  Returns this thread's share of the active-row count.
template <int THREADS, int R, int COLD, class RowOf, class WLoad>
__device__ __forceinline__ unsigned int bt_batch(const CsrView& m, const BtLds& L, float* __restrict__ gcold, int B, int b0,
                                                 RowOf row_of, WLoad wload, float qscale, int* bad,
                                                 unsigned int* gmask = nullptr, float* gdot = nullptr) {
  constexpr int CAP = THREADS / BT_G * R;
  unsigned int n_act = 0;
  while (b0 < B) {   // workgroup-uniform
    const BtRow row = bt_rows_issue<CAP>(m, B, b0, row_of, bad);
    const int2 bd = bt_build<THREADS, CAP>(L, B, b0, row);
    if (bd.x == 0) {
      n_act += bt_giant_row<THREADS, COLD>(m, L, gcold, wload, qscale, gmask, b0, gdot);
      b0 += 1;
      continue;
    }
    BtItems<R> it;
    bt_items_issue<THREADS, R>(m, L, bd.y, it);
    bt_items_dot<THREADS, R>(L, it, wload);
    __syncthreads();
    n_act += bt_gate(L, bd.x, gmask, b0, gdot);
    __syncthreads();
    bt_scatter<R, COLD>(L, gcold, it, qscale);
    b0 += bd.x;
    // (bt_build writes only misc[] before its first barrier; nothing above reads misc[] after the last barrier)
  }
  return n_act;
}

// ======================================================================================================
// K1m: index-list batches spread over workgroups (several hosted workers, hundreds to 10^5 rows per list)
// ======================================================================================================
// ref: core/Slave.scala:147-153 (Vec.sum of the gated sub-gradients of a batch).  Every workgroup takes `rows_per_wg`
// rows of one worker's list into its own fixed-point LDS accumulators and writes them out as one partial
// (part[worker][workgroup][rank]); the ranks beyond LDS go to 64-bit fixed-point global accumulators on the same
// grid.  dsgd_fix_reduce_kernel / dsgd_fix_reduce_apply_kernel -- the finish of the streaming path -- add the
// partials exactly, in a fixed order, with ONE rounding: the gradient is bit-reproducible.
//
// Round 3: WAVE-AUTONOMOUS.  Round 2's form ran the workgroup-wide staged engine above over the slice -- a table of
// work items built by a workgroup scan, three workgroup barriers and four dependent memory round trips per 128-item
// sub-batch (B = 65,536: three sub-batches per workgroup, 69 us per step = 0.07 of the HBM roofline; B = 4,096:
// 26 us).  Here every WAVE owns a contiguous piece of the slice and needs no barrier and no shared table:
//   * one lane per row requests the row record (index -> row_ptr, label): two dependent round trips for up to 64 rows;
//   * rows are classed by length: SHORT (<= 128 non-zeros, 86 % of RCV1-like rows) take one 16-lane group each, eight
//     rows per pass; MEDIUM (129..512) take a whole wave slot -- four groups, one 128-chunk each -- two rows per pass;
//     WIDE (513..1024, 0.3 %) take both slots of a pass.  All chunks of a row sit in ONE pass, so x.w is a DPP
//     butterfly over 16 or 64 lanes, the gate follows at once and the non-zeros -- still in registers -- go straight
//     into the accumulators.  (The first form of this kernel walked rows beyond 512 non-zeros chunk by chunk: one such
//     row per ~370 left one wave six dependent round trips behind, and a launch is as slow as its slowest wave --
//     B = 65,536 took 74 us instead of 47.)  Rows up to 2048 non-zeros are held by both register sets at once;
//   * the row records of a class are compacted into a per-wave LDS strip (short rows from the front, medium rows
//     from the back), passes are software-pipelined two deep (the loads of pass q+1 are issued before pass q is
//     processed), every lane reads EIGHT CONTIGUOUS non-zeros with 16-byte loads;
//   * the workgroup's only shared state: the fixed-point accumulators (ds_add_u32) and a copy of the MB_WL hottest
//     weights fetched straight into LDS (global_load_lds) while the row records are in flight -- ~85 % of the weight
//     gathers never reach the texture path.
struct MbArgs {
  CsrView m;
  const float* w;
  const int* idx;           // null: the list positions are the row numbers themselves
  const WorkSeg* segs;      // one list per worker (blockIdx.y)
  int* part;
  long long* g64_base;      // per worker: stride g_stride
  long long g_stride;
  DevScalars* sc;
  float qscale;
  int part_stride, rows_per_wg, hl, wl, dp;   // hl: ranks with an LDS accumulator; wl: ranks with an LDS copy of w (multiple of 256)
  unsigned long long* tprof;   // optional (tuning runs, DSGD_PLAN_PROF=1): cycles of wave 0 of workgroup 0 by phase, [15] = launches
};
constexpr int MB_THREADS = 512;    // 8 waves of up to 256 VGPRs: three register sets of a pass pipeline per wave
constexpr int MB_R = 2;          // item slots per 16-lane group and pass: 8 short rows or 2 medium rows per pass
constexpr int MB_HL = 24576;     // ranks with an LDS accumulator per workgroup (96 KiB)
constexpr int MB_WL = 11264;     // ranks with an LDS copy of their weight (44 KiB)
constexpr int MB_SHORT = BT_CH;          // 128
constexpr int MB_MEDIUM = 4 * BT_CH;     // 512
constexpr int MB_WIDE = 4 * BT_CH * MB_R;  // 1024: one row over both slots of a pass
struct __attribute__((aligned(16))) MbRec {   // one row of a wave's batch, as the passes need it
  long long st;    // first non-zero
  int len;
  float y;         // label (+1 / -1)
};
__host__ __device__ constexpr int mb_lds_words(int hl, int wl) { return ((hl + 3) & ~3) + wl + (MB_THREADS / 64) * 64 * 4 + 4; }

template <int R>
struct MbPass {   // the non-zeros of a pass, as loaded (which slots are valid is re-derived from the strip: registers are scarce)
  int c[R][BT_K];
  float v[R][BT_K];
};
// The passes of a wave's batch, in order: n_sp SHORT passes (8 rows each: slot (r, g) = short row q*4R + r*4 + g,
// chunk 0), n_mp MEDIUM passes (slot r = medium row (q - n_sp)*R + r from the BACK of the strip, group g = its chunk g),
// then one WIDE pass per wide row (513 .. 1024 non-zeros: both slots, slot r group g = chunk 4r + g).
struct MbPlan {
  int n_short, n_medium, n_wide, n_sp, n_mp;
  __device__ __forceinline__ int kind(int q) const { return q < n_sp ? 0 : (q < n_sp + n_mp ? 1 : 2); }
  __device__ __forceinline__ int passes() const { return n_sp + n_mp + n_wide; }
};
// slot (r, g) of pass q: its row record (len 0: empty slot) and this lane's offset into the row
struct MbSlot {
  MbRec rec;
  int off;
};
template <int R>
__device__ __forceinline__ MbSlot mb_slot(const MbRec* recs, const MbPlan& pl, int q, int r, int g, int sub) {
  const int kind = pl.kind(q);
  int i, at, chunk;
  bool valid;
  if (q >= pl.passes()) {   // a prefetch beyond the last pass: an empty slot
    valid = false;
    at = 0;
    chunk = 0;
  } else if (kind == 0) {
    i = q * 4 * R + r * 4 + g;
    valid = i < pl.n_short;
    at = i;
    chunk = 0;
  } else if (kind == 1) {
    i = (q - pl.n_sp) * R + r;
    valid = i < pl.n_medium;
    at = 63 - i;
    chunk = g;
  } else {
    i = q - pl.n_sp - pl.n_mp;
    valid = true;
    at = pl.n_short + i;
    chunk = 4 * r + g;
  }
  MbSlot s;
  s.rec = recs[valid ? at : 0];
  s.off = chunk * BT_CH + sub * BT_K;
  if (!valid) {
    s.rec.st = 0;
    s.rec.len = 0;
    s.rec.y = 0.0f;
  }
  return s;
}

// 16-byte loads of eight contiguous non-zeros from position p0 (dword-aligned: rows start anywhere; slots past the
// row's end read what follows -- the arrays are padded -- and are masked by the row length)
__device__ __forceinline__ void mb_load8(const CsrView& m, long long p0, int (&c)[BT_K], float (&v)[BT_K]) {
  typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const i32x4u ca = *reinterpret_cast<const i32x4u*>(m.col + p0), cb = *reinterpret_cast<const i32x4u*>(m.col + p0 + 4);
  const f32x4u va = *reinterpret_cast<const f32x4u*>(m.val + p0), vb = *reinterpret_cast<const f32x4u*>(m.val + p0 + 4);
  c[0] = ca.x; c[1] = ca.y; c[2] = ca.z; c[3] = ca.w; c[4] = cb.x; c[5] = cb.y; c[6] = cb.z; c[7] = cb.w;
  v[0] = va.x; v[1] = va.y; v[2] = va.z; v[3] = va.w; v[4] = vb.x; v[5] = vb.y; v[6] = vb.z; v[7] = vb.w;
}

// loads of pass q (nothing waits here)
template <int R>
__device__ __forceinline__ void mb_issue(const CsrView& m, const MbRec* recs, const MbPlan& pl, int q, int g, int sub,
                                         MbPass<R>& P) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const MbSlot s = mb_slot<R>(recs, pl, q, r, g, sub);
    mb_load8(m, s.off < s.rec.len ? s.rec.st + s.off : 0, P.c[r], P.v[r]);
  }
}

// weights of the ranks c: the LDS copy for c < wl_n, global memory beyond
struct MbWeights {
  const float* wl;   // LDS
  const float* __restrict__ w;
  int wl_n;
  // the global half: requests only (the results stay untouched until `finish`: any arithmetic on them here would make
  // the compiler wait on the spot)
  template <int R>
  __device__ __forceinline__ void request(const int (&c)[R][BT_K], const int (&cnt)[R], float (&u)[R][BT_K]) const {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < BT_K; ++k) {
        const int cc = k < cnt[r] ? c[r][k] : 0;
        u[r][k] = w[cc < wl_n ? 0 : cc];
      }
  }
  // the LDS half and the select of values
  template <int R>
  __device__ __forceinline__ void finish(const int (&c)[R][BT_K], const int (&cnt)[R], float (&u)[R][BT_K]) const {
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    float v[R][BT_K];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < BT_K; ++k) {
        const int cc = k < cnt[r] ? c[r][k] : 0;
        v[r][k] = ((lds_cfloat*)wl)[cc < wl_n ? cc : 0];
      }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < BT_K; ++k) {
        const int cc = k < cnt[r] ? c[r][k] : 0;
        u[r][k] = cc < wl_n ? v[r][k] : u[r][k];
      }
  }
  __device__ __forceinline__ float operator()(int c) const {   // (the rare long-row paths)
    typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
    const bool hot = c < wl_n;
    const float v = ((lds_cvfloat*)wl)[hot ? c : 0];
    const float u = w[hot ? 0 : c];
    return hot ? v : u;
  }
};

// the weight gathers of pass q (its column ids have landed): requested BEFORE the loads of pass q+1 go out, so that
// the counted wait for them (vmcnt retires in order) does not also wait for that younger, slower HBM request
template <int R, class WLoad>
__device__ __forceinline__ void mb_gather(const MbRec* recs, const MbPlan& pl, int q, int g, const MbPass<R>& P, int sub,
                                          WLoad wload, float (&wv)[R][BT_K]) {
  int cnt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const MbSlot s = mb_slot<R>(recs, pl, q, r, g, sub);
    cnt[r] = s.rec.len - s.off;
  }
  wload.request(P.c, cnt, wv);
}

// x.w of every slot's row, the gate, the scatter of the active rows' non-zeros (still in registers)
template <int R, class WLoad, class Stamp>
__device__ __forceinline__ unsigned int mb_process(const BtLds& L, const MbRec* recs, const MbPlan& pl, int q, int g,
                                                   const MbPass<R>& P, float (&wv)[R][BT_K], int sub, int lane,
                                                   WLoad wload, float qscale, Stamp stamp) {
  const int kind = pl.kind(q);
  unsigned int n_act = 0;
  float acc[R], y[R];
  int cnt[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const MbSlot s = mb_slot<R>(recs, pl, q, r, g, sub);
    cnt[r] = s.rec.len - s.off;   // valid slots of this lane (<= 0: none)
    y[r] = s.rec.y;
  }
  wload.finish(P.c, cnt, wv);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < BT_K; ++k) a += k < cnt[r] ? filt(P.v[r][k] * wv[r][k]) : 0.0f;   // ref: math/Sparse.scala:46
    acc[r] = a;
  }
  // fixed reduction trees: every lane of a row holds the bitwise-identical sum (they must agree on the gate)
  if (kind == 2) {   // one row over all the slots
    float t = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) t += acc[r];
    t = group_sum<64>(t);
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = t;
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = kind == 0 ? group_sum<BT_G>(acc[r]) : group_sum<64>(acc[r]);
  }
  stamp(5);   // (tuning runs) weights landed, products, reduction
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool active = y[r] != 0.0f && !(y[r] * acc[r] < 0.0f);   // ref: core/ml/SparseSVM.scala:27-28
    const bool counts = kind == 0 ? sub == 0 : (kind == 1 ? lane == 0 : (lane == 0 && r == 0));
    n_act += (active && counts) ? 1u : 0u;
    if (active) {
#pragma unroll
      for (int k = 0; k < BT_K; ++k)
        if (k < cnt[r]) bt_add<2>(L, nullptr, P.c[r][k], P.v[r][k] * y[r], qscale);
    }
  }
  return n_act;
}

// a HUGE row (1025 .. 2048 non-zeros, 0.01 % of RCV1-like rows): both register sets of the pass pipeline hold it at
// once -- one round trip for the whole row instead of a chunk-by-chunk walk that would leave one wave (and with it
// the whole launch) several dependent round trips behind the others
template <int R, class WLoad>
__device__ __forceinline__ unsigned int mb_huge(const CsrView& m, const BtLds& L, const MbRec& rec, int g, int sub, int lane,
                                                MbPass<R>& A, MbPass<R>& B, WLoad wload, float qscale) {
  float t = 0.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    MbPass<R>& P = h ? B : A;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int off = ((h * R + r) * 4 + g) * BT_CH + sub * BT_K;
      mb_load8(m, off < rec.len ? rec.st + off : 0, P.c[r], P.v[r]);
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const MbPass<R>& P = h ? B : A;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int cnt = rec.len - (((h * R + r) * 4 + g) * BT_CH + sub * BT_K);
#pragma unroll
      for (int k = 0; k < BT_K; ++k)
        if (k < cnt) t += filt(P.v[r][k] * wload(P.c[r][k]));
    }
  }
  const float d = group_sum<64>(t);
  if (rec.y * d < 0.0f) return 0u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const MbPass<R>& P = h ? B : A;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int cnt = rec.len - (((h * R + r) * 4 + g) * BT_CH + sub * BT_K);
#pragma unroll
      for (int k = 0; k < BT_K; ++k)
        if (k < cnt) bt_add<2>(L, nullptr, P.c[r][k], P.v[r][k] * rec.y, qscale);
    }
  }
  return lane == 0 ? 1u : 0u;
}

// a row beyond that (never the case for RCV1): the whole wave walks it twice (x.w, then the scatter)
template <class WLoad>
__device__ __forceinline__ unsigned int mb_giant(const CsrView& m, const BtLds& L, long long st, int len, float y, int lane,
                                                 WLoad wload, float qscale) {
  float acc = 0.0f;
  for (int off = lane * BT_K; off < len; off += 64 * BT_K) {
    int c[BT_K];
    float v[BT_K];
    mb_load8(m, st + off, c, v);
#pragma unroll
    for (int k = 0; k < BT_K; ++k)
      if (off + k < len) acc += filt(v[k] * wload(c[k]));
  }
  const float d = group_sum<64>(acc);
  if (y * d < 0.0f) return 0u;
  for (int off = lane * BT_K; off < len; off += 64 * BT_K) {
    int c[BT_K];
    float v[BT_K];
    mb_load8(m, st + off, c, v);
#pragma unroll
    for (int k = 0; k < BT_K; ++k)
      if (off + k < len) bt_add<2>(L, nullptr, c[k], v[k] * y, qscale);
  }
  return lane == 0 ? 1u : 0u;
}

// copy of w[0, wl) into LDS, 1 KiB pieces straight from L2 (no staging registers); wl a multiple of 256
__device__ __forceinline__ void mb_wcache_issue(const float* __restrict__ w, float* wl_lds, int wl) {
  const int lane = threadIdx.x & 63;
  const unsigned int lds_base = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)wl_lds;
  for (int piece = threadIdx.x >> 6; piece < (wl >> 8); piece += MB_THREADS / 64) {
    const float* src = w + piece * 256 + lane * 4;
    const unsigned int dst = (unsigned int)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned int)piece * 1024u));
    unsigned int keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
  }
}

__global__ void __launch_bounds__(MB_THREADS) dsgd_mb_grad_kernel(MbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, sub = tid & (BT_G - 1), g = (tid >> 4) & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  BtLds L;
  L.hl = a.hl;
  L.acc = reinterpret_cast<int*>(lds);
  L.cbits = nullptr;
  L.g64 = a.g64_base + (long long)blockIdx.y * a.g_stride;
  L.hbits = nullptr;
  L.hh = 0;
  float* wl = lds + ((a.hl + 3) & ~3);
  MbRec* recs = reinterpret_cast<MbRec*>(wl + a.wl) + wave * 64;   // this wave's strip
  // active rows of the workgroup: ONE global atomic per workgroup at the end (one per wave -- 3,072 same-address atomics
  // for B = 65,536 -- queued up behind each other: the phase counters showed the last phase growing with the wave count).
  // (In the dynamic allocation: a static __shared__ on top of 160 KiB of dynamic LDS is refused at launch set-up.)
  unsigned int& wg_active = *reinterpret_cast<unsigned int*>(wl + a.wl + (MB_THREADS / 64) * 64 * 4);
  const WorkSeg seg = a.segs[blockIdx.y];
  const long long b = seg.begin + (long long)blockIdx.x * a.rows_per_wg;
  const long long e = b + a.rows_per_wg < seg.end ? b + a.rows_per_wg : seg.end;
  const long long n = e > b ? e - b : 0;
  const long long rpw = (n + MB_THREADS / 64 - 1) / (MB_THREADS / 64);   // rows per wave: a contiguous piece each
  const long long wb = b + wave * rpw, we = wb + rpw < e ? wb + rpw : e;
  const int* __restrict__ idx = a.idx;
  const int wl_n = a.wl;
  // hot ranks from the LDS copy, the tail from L1/L2 -- with UNCONDITIONAL global loads (the hot lanes all read w[0],
  // one cache line), all of a pass issued back to back before the first is used: a branch around each load made the
  // compiler wait for every one of a pass's 16 loads inside its own branch, i.e. 16 dependent L2 round trips per pass
  // (measured: B = 65,536 took 86 us that way, more than round 2's kernel)
  const MbWeights wload{wl, a.w, wl_n};
  // the row ids of the wave's first batch go out first; the weight copy and the clearing of the accumulators run
  // under that round trip
  auto row_id = [&](long long t) -> long long { return t < we ? (idx ? (long long)idx[t] : t) : -1; };
  const bool prof = a.tprof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0;
  unsigned long long tl = prof ? __builtin_readcyclecounter() : 0ull, tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i) {   // cycles since the previous stamp -> tp[i]
    if (prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      tp[i] += now - tl;
      tl = now;
    }
  };
  long long row = row_id(wb + lane);
  mb_wcache_issue(a.w, wl, a.wl);
  wg_zero(L.acc, a.hl, tid, MB_THREADS);
  if (tid == 0) wg_active = 0u;
  stamp(0);   // issue of the row ids, the weight copy, clearing
  unsigned int n_act = 0;
  bool first = true;
  for (long long t0 = wb; first || t0 < we; t0 += 64) {   // (every wave runs the first round: it holds the barrier)
    if (!first) row = row_id(t0 + lane);
    const bool ok = row >= 0 && row < a.m.n_rows;
    if (t0 + lane < we && !ok) atomicOr(&a.sc->err, 1);
    long long st = 0, en = 0;
    float y = 0.0f;
    if (ok) {
      st = a.m.row_ptr[row];
      en = a.m.row_ptr[row + 1];
      y = (float)a.m.label[row];
    }
    const int len = (int)(en - st);
    if (prof) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(1);   // row ids + row records landed (two dependent round trips)
    }
    const bool is_s = ok && len > 0 && len <= MB_SHORT, is_m = ok && len > MB_SHORT && len <= MB_MEDIUM;
    const bool is_w = ok && len > MB_MEDIUM && len <= MB_WIDE, is_g = ok && len > MB_WIDE;
    const unsigned long long ms = __builtin_amdgcn_ballot_w64(is_s), mm = __builtin_amdgcn_ballot_w64(is_m);
    const unsigned long long mw = __builtin_amdgcn_ballot_w64(is_w), mg = __builtin_amdgcn_ballot_w64(is_g);
    MbPlan pl;
    pl.n_short = __popcll(ms);
    pl.n_medium = __popcll(mm);
    pl.n_wide = __popcll(mw);
    pl.n_sp = (pl.n_short + 4 * MB_R - 1) / (4 * MB_R);
    pl.n_mp = (pl.n_medium + MB_R - 1) / MB_R;
    const int n_giant = __popcll(mg);
    __builtin_amdgcn_wave_barrier();   // (the previous round's reads of the strip are behind us: same wave, in order)
    if (ok && len > 0) {
      // short rows from the front, wide rows behind them, then the giants; medium rows from the back: never meeting
      const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
      MbRec rec;
      rec.st = st;
      rec.len = len;
      rec.y = y;
      const int at = is_s ? __popcll(ms & below)
                          : (is_m ? 63 - __popcll(mm & below)
                                  : (is_w ? pl.n_short + __popcll(mw & below) : pl.n_short + pl.n_wide + __popcll(mg & below)));
      recs[at] = rec;
    }
    __builtin_amdgcn_wave_barrier();   // same wave writes and reads the strip; LDS executes a wave's accesses in order
    const int n_q = pl.passes();
    // Three register sets, rotated by unrolling: while pass q is processed the weights of pass q+1 and the non-zeros of
    // pass q+2 are in flight.  EVERY request below is unconditional (a pass beyond the last one reads address 0 and is
    // masked): behind a conditional load the compiler no longer knows how many younger loads are outstanding and falls
    // back to s_waitcnt vmcnt(0) -- which waits for the prefetches too and puts both round trips back on every pass
    // (the first form of this loop: 3,400 + 2,000 cycles per pass in the phase counters).
    MbPass<MB_R> A, B, C;
    float ua[MB_R][BT_K], ub[MB_R][BT_K], uc[MB_R][BT_K];
    mb_issue<MB_R>(a.m, recs, pl, 0, g, sub, A);
    mb_issue<MB_R>(a.m, recs, pl, 1, g, sub, B);
    if (first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the weight copy have landed ...
      stamp(2);                                           // (first pass's non-zeros landed)
      __syncthreads();                                    // ... and everybody's; the accumulators are clear
      stamp(3);                                           // (waiting for the other waves)
      first = false;
    }
    int q = 0;
    mb_gather<MB_R>(recs, pl, 0, g, A, sub, wload, ua);
#define MB_STEP(PC, UC, PN, UN, PF)                                                                      \
  mb_issue<MB_R>(a.m, recs, pl, q + 2, g, sub, PF);       /* non-zeros of pass q+2 */                      \
  mb_gather<MB_R>(recs, pl, q + 1, g, PN, sub, wload, UN); /* weights of pass q+1 (its column ids landed) */ \
  stamp(4);                                                                                              \
  n_act += mb_process<MB_R>(L, recs, pl, q, g, PC, UC, sub, lane, wload, a.qscale, stamp);                \
  stamp(6);   /* gate + scatter (LDS atomics, 64-bit global atomics of the tail) */                       \
  if (++q >= n_q) break;
    while (q < n_q) {   // wave-uniform
      MB_STEP(A, ua, B, ub, C)
      MB_STEP(B, ub, C, uc, A)
      MB_STEP(C, uc, A, ua, B)
    }
#undef MB_STEP
    for (int j = 0; j < n_giant; ++j) {   // wave-uniform
      const MbRec rec = recs[pl.n_short + pl.n_wide + j];
      if (rec.len <= 2 * MB_WIDE) n_act += mb_huge<MB_R>(a.m, L, rec, g, sub, lane, A, B, wload, a.qscale);
      else n_act += mb_giant(a.m, L, rec.st, rec.len, rec.y, lane, wload, a.qscale);
    }
  }
  n_act = wave_sum_u32(n_act);
  if (lane == 0 && n_act) atomicAdd(&wg_active, n_act);
  __syncthreads();
  stamp(7);   // waiting for the other waves
  int* mine = a.part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * a.part_stride;
  wg_copy_out(mine, L.acc, a.hl, tid, MB_THREADS, is_aligned16(mine));
  if (tid == 0 && wg_active) atomicAdd(&a.sc->n_active, (unsigned long long)wg_active);
  if (prof && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long now = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) a.tprof[i] += tp[i];
    a.tprof[8] += now - tl;   // writing the partial out
    a.tprof[15] += 1;
  }
}

// ======================================================================================================
// K1v: index-list batches as VIRTUAL TILES over the split streams (round 3)
// ======================================================================================================
// ref: core/Slave.scala:147-153 (Vec.sum of the gated sub-gradients of a batch), the same contract as
// dsgd_mb_grad_kernel: per-workgroup fixed-point partials of the hot ranks, 64-bit fixed-point global accumulators
// for the cold ones, the exact reduce + regularise + update behind it.
//
// dsgd_mb_grad_kernel walks the whole ranked CSR row by row: 32-bit ids, half of the weights gathered from global
// memory, every slot masked in four places -- ~1,500 instructions per pass of 1,024 slots, 40 us for B = 65,536
// (0.1 of the HBM roofline), the worst fraction in the repository.  The streaming kernel next door spends ~180 per
// 512-slot tile because its tiles are laid out for it.  Here the HOST lays a batch out the same way when a plan is
// created: every row of a list gets ceil(hot entries / 8) consecutive lanes of a 64-lane "virtual tile", each lane a
// 16-byte descriptor {position of its 8 slots in the hot stream, valid slots, first / last lane of the row, label
// sign, local row, position AND rank of ONE cold entry}.  All hot weights and the hot gradient sit in LDS exactly as
// in dsgd_wseg_kernel (no gathers, no rank checks), one segmented scan per tile (a lane belongs to ONE row: no head /
// trail fragments), the gate on the row's last lane, coefficients through the wave's strip.
//
// A step of this size is latency, not bytes (the waves of the first form waited 73 % of their cycles with 650 VALU
// instructions each: profiles/r03_vt_pmc_and_ablation.txt), so a wave requests the descriptors of up to FOUR tiles at
// once, then everything they point to at once -- the cold entry's weight included: the host knows its rank and stores
// it in the descriptor -- and only then computes: two round trips per group of four tiles (the grid is sized for one
// tile per wave; two to three at B = 65,536, where the 256 CUs cap it).  The rows outside the tiled streams get
// workgroups of their own (vt_long_row: record -> whole row
// in registers -> cold weights), running beside the tile workgroups instead of behind them.  What an ablation then
// found under all of it: 55 of the 67 us went to 4,096 waves finishing together and adding their active-row counts
// to ONE address -- one atomic per workgroup: 24.6 us for B = 65,536 (dsgd_mb_grad_kernel: 40), 15.7 for 4,096 (18.4),
// 12.3 for 3 x 100 (15.4).
struct VtLane {          // 16 bytes, one per lane of a virtual tile (written by the host: vt_build)
  unsigned int hp;       // first of the lane's <= 8 slots in the hot stream
  unsigned int info;     // valid slots (bits 0-3) | first lane of its row (4) | last lane (5) | label > 0 (6) |
                         // has a cold entry (8) | local row of the tile (bits 16-21)
  unsigned int cp;       // the lane's cold entry in the cold stream
  unsigned int crank;    // ... and its rank - hsplit (the host keeps a copy of the cold ranks: no load depends on a load)
};
constexpr unsigned int VT_START = 1u << 4, VT_LAST = 1u << 5, VT_YPOS = 1u << 6, VT_COLD = 1u << 8;

struct VtArgs {
  const unsigned short* hcol;   // hot stream: 16-bit ranks, values
  const float* hval;
  const unsigned short* ccol;   // cold stream (16-bit form): rank - hsplit, values
  const float* cval;
  const float* w;
  const VtLane* lanes;          // 64 per tile
  const uint4* packed;          // PACKED plans: 4 KiB per tile -- the tile's own copy of what its descriptors point to
  const WorkSeg* tsegs;         // tile range of every worker's list of this step (blockIdx.y)
  const WorkSeg* lsegs;         // ... and its range of `long_recs`: rows outside the tiled streams (more than 504 hot or
  const MbRec* long_recs;       //     64 cold entries), one wave per row from the whole ranked CSR (vt_long_row)
  CsrView mfull;
  int* part;
  long long* g64_base;
  long long g_stride;
  DevScalars* sc;
  float qscale, cold_scale;     // fixed-point scales of the LDS tile (per launch) and of the 64-bit accumulators
  int part_stride, hsplit, gx_tiles;   // gx_tiles: workgroups per worker that walk tiles (the rest take the long rows)
};

struct VtRegs {
  uint4 d;                // the lane's descriptor (raw: a select on it at request time would wait for the load on the spot)
  int live;               // wave-uniform: the tile exists (a tile beyond the wave's last one is requested and masked)
  unsigned int c[5];      // 20 bytes from the dword below the lane's first rank (ranks sit at any 2-byte offset)
  float4 v0, v1;
  float cv, cw;           // cold value, cold weight
};

// stage D: the descriptors of tile t (a tile beyond the wave's last one reads the list's first tile and is masked)
__device__ __forceinline__ void vt_issue_desc(const VtArgs& a, int t, int t_end, int lane, VtRegs& r) {
  r.live = t < t_end;
  r.d = make_uint4(0u, 0u, 0u, 0u);
  // (wave-uniform branch: thousands of waves re-reading the list's last tile -- what a clamped index does -- meet on
  //  one L2 channel)
  if (r.live) r.d = reinterpret_cast<const uint4*>(a.lanes)[(long long)t * 64 + lane];
}
// stage S: the lane's slots of the hot stream and its cold entry
__device__ __forceinline__ void vt_issue_stream(const VtArgs& a, VtRegs& r) {
  typedef unsigned int u32x4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  if (!r.live) return;   // (wave-uniform; vt_process masks the set)
  const unsigned int hp = r.d.x;
  const unsigned int* cb = reinterpret_cast<const unsigned int*>(a.hcol) + (hp >> 1);   // the dword holding rank hp
  const u32x4u ca = *reinterpret_cast<const u32x4u*>(cb);
  r.c[0] = ca.x; r.c[1] = ca.y; r.c[2] = ca.z; r.c[3] = ca.w;
  r.c[4] = cb[4];
  const f32x4u va = *reinterpret_cast<const f32x4u*>(a.hval + hp), vb = *reinterpret_cast<const f32x4u*>(a.hval + hp + 4);
  r.v0 = make_float4(va.x, va.y, va.z, va.w);
  r.v1 = make_float4(vb.x, vb.y, vb.z, vb.w);
  const bool hasc = (r.d.y & VT_COLD) != 0u;
  r.cv = a.cval[hasc ? r.d.z : 0u];
  r.cw = a.w[a.hsplit + (hasc ? (int)r.d.w : 0)];
}

// PACKED plans (small and mid-size plans: vt_build): the rows a tile touches are copied into the plan in tile order
// when it is built -- four 1 KiB blocks per tile {descriptor-like words, eight ranks, values 0-3, values 4-7}, each one
// coalesced 16-byte load per lane.  A step then makes ONE round trip into HBM instead of two dependent ones
// (descriptors, then wherever they point) -- its data also sits where the translation caches see it again next step --
// and the cold weights' gather (L2) goes out as soon as the first block has landed, under the other three.
__device__ __forceinline__ void vt_issue_packed_misc(const VtArgs& a, int t, int t_end, int lane, VtRegs& r) {
  r.live = t < t_end;
  r.d = make_uint4(0u, 0u, 0u, 0u);
  if (r.live) r.d = a.packed[(long long)t * 256 + lane];   // {0, info, cold value bits, cold rank}
}
__device__ __forceinline__ void vt_issue_packed_data(const VtArgs& a, int t, int lane, VtRegs& r) {
  if (!r.live) return;
  const uint4* b = a.packed + (long long)t * 256 + lane;
  const uint4 c = b[64], x = b[128], y = b[192];
  r.c[0] = c.x; r.c[1] = c.y; r.c[2] = c.z; r.c[3] = c.w;
  r.c[4] = 0u;
  r.v0 = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
  r.v1 = make_float4(__uint_as_float(y.x), __uint_as_float(y.y), __uint_as_float(y.z), __uint_as_float(y.w));
}
__device__ __forceinline__ void vt_issue_packed_cw(const VtArgs& a, VtRegs& r) {   // (the first block has landed)
  if (!r.live) return;
  r.cv = __uint_as_float(r.d.z);
  r.cw = a.w[a.hsplit + ((r.d.y & VT_COLD) ? (int)r.d.w : 0)];
}
// one wave per tile: the plan's packed copy from the descriptors and the streams (runs once, when the plan is built)
__global__ void __launch_bounds__(256) dsgd_vt_pack_kernel(VtArgs a, long long n_tiles, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= n_tiles) return;
  VtRegs r;
  vt_issue_desc(a, (int)t, (int)n_tiles, lane, r);
  vt_issue_stream(a, r);
  const unsigned int info = r.d.y, cnt = info & 15u;
  const unsigned int sh = (r.d.x & 1u) << 4;
  uint4 c;
  c.x = __builtin_amdgcn_alignbit(r.c[1], r.c[0], sh);
  c.y = __builtin_amdgcn_alignbit(r.c[2], r.c[1], sh);
  c.z = __builtin_amdgcn_alignbit(r.c[3], r.c[2], sh);
  c.w = __builtin_amdgcn_alignbit(r.c[4], r.c[3], sh);
  // slots past the row's end hold the next row's entries: rank 0 / value 0 in the copy
  const unsigned int keep[4] = {cnt >= 2u ? 0xffffffffu : (cnt == 1u ? 0xffffu : 0u), cnt >= 4u ? 0xffffffffu : (cnt == 3u ? 0xffffu : 0u),
                                cnt >= 6u ? 0xffffffffu : (cnt == 5u ? 0xffffu : 0u), cnt >= 8u ? 0xffffffffu : (cnt == 7u ? 0xffffu : 0u)};
  c.x &= keep[0]; c.y &= keep[1]; c.z &= keep[2]; c.w &= keep[3];
  const float v[8] = {r.v0.x, r.v0.y, r.v0.z, r.v0.w, r.v1.x, r.v1.y, r.v1.z, r.v1.w};
  unsigned int vb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) vb[k] = (unsigned int)k < cnt ? __float_as_uint(v[k]) : 0u;
  const bool hasc = (info & VT_COLD) != 0u;
  uint4* o = out + t * 256 + lane;
  o[0] = make_uint4(0u, info, hasc ? __float_as_uint(r.cv) : 0u, hasc ? r.d.w : 0u);
  o[64] = c;
  o[128] = make_uint4(vb[0], vb[1], vb[2], vb[3]);
  o[192] = make_uint4(vb[4], vb[5], vb[6], vb[7]);
}

// stage P
__device__ __forceinline__ unsigned int vt_process(const VtArgs& a, const VtRegs& r, float* strip, int* gl,
                                                   long long* __restrict__ g64) {
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  const unsigned int info = r.live ? r.d.y : 0u;   // (masked tile: no valid slot, no row end, no cold entry)
  const unsigned int cnt = info & 15u;
  const unsigned int lrow = (info >> 16) & 63u;
  // eight 16-bit ranks from the five dwords: funnel shift by 0 or 16 bits
  const unsigned int sh = (r.d.x & 1u) << 4;
  unsigned int cw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) cw[k] = __builtin_amdgcn_alignbit(r.c[k + 1], r.c[k], sh);
  int cc[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cc[2 * k] = (int)((cw[k] & 0xffffu) << 2);
    cc[2 * k + 1] = (int)((cw[k] >> 14) & 0x3fffcu);
  }
  // slots past the row's end (the row's last lane) hold the NEXT row's entries: their values become 0 -- the products
  // and the fixed-point contributions follow; the ranks stay valid LDS addresses (< hsplit) whatever they are
  float vv[8] = {r.v0.x, r.v0.y, r.v0.z, r.v0.w, r.v1.x, r.v1.y, r.v1.z, r.v1.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) vv[k] = (unsigned int)k < cnt ? vv[k] : 0.0f;
  float p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = filt(vv[k] * *(lds_cfloat*)(unsigned int)cc[k]);   // ref: math/Sparse.scala:46
  const bool hasc = (info & VT_COLD) != 0u;
  const float cv = hasc ? r.cv : 0.0f;
  float t = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  t += filt(cv * r.cw);
  // a lane belongs to ONE row: an inclusive segmented scan over the lanes leaves x.w on the row's last lane
  int f = (info & VT_START) != 0u;
  wave_seg_scan(t, f);
  const bool last = (info & VT_LAST) != 0u;
  const bool ypos = (info & VT_YPOS) != 0u;
  const float yd = ypos ? t : -t;
  const bool active = last && !(yd < 0.0f);                   // ref: core/ml/SparseSVM.scala:27-28
  if (last) strip[lrow] = active ? (ypos ? a.qscale : -a.qscale) : 0.0f;
  __builtin_amdgcn_wave_barrier();   // same wave writes and reads the strip: LDS keeps a wave's order
  const float coef = cnt != 0u || hasc ? strip[lrow] : 0.0f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  int q[8];
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    // |v * coef| <= 2^shift <= 2^22: v * coef + 1.5 * 2^23 rounds once to an fp32 whose low mantissa bits are q
    const f32x2 rr = __builtin_elementwise_fma(f32x2{vv[k], vv[k + 1]}, f32x2{coef, coef}, f32x2{12582912.0f, 12582912.0f});
    q[k] = __float_as_int(rr.x) - 0x4B400000;
    q[k + 1] = __float_as_int(rr.y) - 0x4B400000;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (q[k] != 0) atomicAdd(reinterpret_cast<int*>(reinterpret_cast<char*>(gl) + cc[k]), q[k]);
  if (hasc && coef != 0.0f) {   // the cold entry of an active row: 64-bit global accumulator at the cold scale
    const int qc = __float2int_rn(cv * (coef > 0.0f ? a.cold_scale : -a.cold_scale));
    if (qc != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g64[a.hsplit + (int)r.d.w]), (unsigned long long)(long long)qc);
  }
  __builtin_amdgcn_wave_barrier();
  return active ? 1u : 0u;
}

// A row outside the tiled streams (0.5 % of RCV1-like rows, 3 % of the non-zeros), one wave per row from the whole
// ranked CSR.  The launch is as slow as its slowest wave and a batch of 65,536 rows holds ~330 of them, so the row is
// NOT walked chunk by chunk (dsgd_wseg_kernel's w_long_row chains two dependent loads per 64 entries -- invisible
// inside a 600 us kernel, 40 us on the critical path here): up to 2,048 non-zeros sit in registers at once (four sets
// of eight contiguous entries per lane, sixteen 16-byte requests in flight), the cold weights are gathered in one more
// round trip, the non-zeros stay in registers for the scatter.  The row record {first non-zero, length, label} comes
// from the plan (no row_ptr round trip).  Longer rows (never the case for RCV1) take the chunked walk.
struct VtLong {
  int c[4][BT_K];
  float v[4][BT_K];
};
__device__ __forceinline__ void vt_long_issue(const VtArgs& a, const MbRec& rec, int lane, VtLong& R) {
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int off = (h * 64 + lane) * BT_K;
    mb_load8(a.mfull, off < rec.len ? rec.st + off : rec.st, R.c[h], R.v[h]);
  }
}
__device__ __forceinline__ unsigned int vt_long_finish(const VtArgs& a, const MbRec& rec, VtLong& R, const float* wl, int* gl,
                                                       long long* __restrict__ g64, int lane) {
  typedef __attribute__((address_space(3))) const float lds_cfloat;
  const int H = a.hsplit;
  int (&c)[4][BT_K] = R.c;
  float (&v)[4][BT_K] = R.v;
  float u[4][BT_K];
  // cold weights: unconditional requests (hot lanes read w[0]), all of them before the first is used
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int k = 0; k < BT_K; ++k) {
      const bool in = (h * 64 + lane) * BT_K + k < rec.len;
      c[h][k] = in ? c[h][k] : 0;
      v[h][k] = in ? v[h][k] : 0.0f;
      u[h][k] = a.w[c[h][k] < H ? 0 : c[h][k]];
    }
  float t = 0.0f;
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int k = 0; k < BT_K; ++k) {
      const float wh = *(lds_cfloat*)(unsigned int)((c[h][k] < H ? c[h][k] : 0) << 2);
      t += filt(v[h][k] * (c[h][k] < H ? wh : u[h][k]));   // ref: math/Sparse.scala:46
    }
  const float d = group_sum<64>(t);
  if (rec.y * d < 0.0f) return 0u;                          // ref: core/ml/SparseSVM.scala:27-28
  const float ch = rec.y * a.qscale, cc = rec.y * a.cold_scale;
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int k = 0; k < BT_K; ++k) {
      if (c[h][k] < H) {
        const int q = __float2int_rn(v[h][k] * ch);
        if (q != 0) atomicAdd(&gl[c[h][k]], q);
      } else {
        const int q = __float2int_rn(v[h][k] * cc);
        if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&g64[c[h][k]]), (unsigned long long)(long long)q);
      }
    }
  return lane == 0 ? 1u : 0u;
}

template <bool PACKED>
__global__ void __launch_bounds__(1024) dsgd_vt_grad_kernel(VtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.hsplit;
  // LDS as in dsgd_wseg_kernel: H + 1 weights at address 0 (ranks become byte offsets), 16 strips, H + 64 gradient words
  float* wl = lds;
  float* strips = lds + ((H + 4) & ~3);
  float* strip = strips + wave * 64;
  int* gl = reinterpret_cast<int*>(strips + 16 * 64);
  const WorkSeg seg = a.tsegs[blockIdx.y];
  const WorkSeg ls = a.lsegs[blockIdx.y];
  long long* g64 = a.g64_base + (long long)blockIdx.y * a.g_stride;
  // the first gx_t workgroups of a worker walk its tiles, the others (gridDim.x - gx_t, when the list holds rows outside
  // the tiled streams) take those rows, one wave per row
  const int gx_t = a.gx_tiles;
  const bool tile_wg = (int)blockIdx.x < gx_t;
  const int stride = gx_t * 16, t_end = (int)seg.end;
  int tile = (int)seg.begin + (int)blockIdx.x * 16 + wave;
  const long long l_stride = (long long)((int)gridDim.x - gx_t) * 16;
  long long lt = ls.begin + (long long)((int)blockIdx.x - gx_t) * 16 + wave;
  if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)lds != 0u) {
    if (tid == 0) atomicOr(&a.sc->err, 2);
    return;
  }
  unsigned int n_act = 0;
  // (the two kinds of workgroup share no code between their requests and their arithmetic: with the LDS set-up in a
  //  common stretch both register files -- four tile sets, a whole long row -- were live across it and spilled)
  if (tile_wg) {
    VtRegs A, B, C, D;
    if (PACKED) {
      // ONE round trip: the tiles' first blocks, then their other three; the cold weights go out under those
      vt_issue_packed_misc(a, tile, t_end, lane, A);
      vt_issue_packed_misc(a, tile + stride, t_end, lane, B);
      vt_issue_packed_misc(a, tile + 2 * stride, t_end, lane, C);
      vt_issue_packed_misc(a, tile + 3 * stride, t_end, lane, D);
      vt_issue_packed_data(a, tile, lane, A);
      vt_issue_packed_data(a, tile + stride, lane, B);
      vt_issue_packed_data(a, tile + 2 * stride, lane, C);
      vt_issue_packed_data(a, tile + 3 * stride, lane, D);
      wg_zero(gl, H + 64, tid, 1024);
      vt_issue_packed_cw(a, A);
      vt_issue_packed_cw(a, B);
      vt_issue_packed_cw(a, C);
      vt_issue_packed_cw(a, D);
    } else {
      // round trip 1: the descriptors of the wave's first four tiles; the accumulators are cleared under it
      vt_issue_desc(a, tile, t_end, lane, A);
      vt_issue_desc(a, tile + stride, t_end, lane, B);
      vt_issue_desc(a, tile + 2 * stride, t_end, lane, C);
      vt_issue_desc(a, tile + 3 * stride, t_end, lane, D);
      wg_zero(gl, H + 64, tid, 1024);
      // round trip 2: everything the descriptors point to, and the weight tile
      vt_issue_stream(a, A);
      vt_issue_stream(a, B);
      vt_issue_stream(a, C);
      vt_issue_stream(a, D);
    }
    wg_copy_in(wl, a.w, H, tid, 1024, is_aligned16(a.w));
    if (tid == 0) wl[H] = 0.0f;
    __syncthreads();
    for (;;) {   // (wave-uniform) groups of four tiles: two round trips each
      n_act += vt_process(a, A, strip, gl, g64);
      n_act += vt_process(a, B, strip, gl, g64);
      n_act += vt_process(a, C, strip, gl, g64);
      n_act += vt_process(a, D, strip, gl, g64);
      tile += 4 * stride;
      if (tile >= t_end) break;
      if (PACKED) {
        vt_issue_packed_misc(a, tile, t_end, lane, A);
        vt_issue_packed_misc(a, tile + stride, t_end, lane, B);
        vt_issue_packed_misc(a, tile + 2 * stride, t_end, lane, C);
        vt_issue_packed_misc(a, tile + 3 * stride, t_end, lane, D);
        vt_issue_packed_data(a, tile, lane, A);
        vt_issue_packed_data(a, tile + stride, lane, B);
        vt_issue_packed_data(a, tile + 2 * stride, lane, C);
        vt_issue_packed_data(a, tile + 3 * stride, lane, D);
        vt_issue_packed_cw(a, A);
        vt_issue_packed_cw(a, B);
        vt_issue_packed_cw(a, C);
        vt_issue_packed_cw(a, D);
      } else {
        vt_issue_desc(a, tile, t_end, lane, A);
        vt_issue_desc(a, tile + stride, t_end, lane, B);
        vt_issue_desc(a, tile + 2 * stride, t_end, lane, C);
        vt_issue_desc(a, tile + 3 * stride, t_end, lane, D);
        vt_issue_stream(a, A);
        vt_issue_stream(a, B);
        vt_issue_stream(a, C);
        vt_issue_stream(a, D);
      }
    }
  } else {
    VtLong LR;
    MbRec lrec;
    lrec.st = 0;
    lrec.len = 0;
    lrec.y = 0.0f;
    if (lt < ls.end) lrec = a.long_recs[lt];                                        // round trip 1
    wg_zero(gl, H + 64, tid, 1024);
    if (lrec.len != 0 && lrec.len <= 4 * 64 * BT_K) vt_long_issue(a, lrec, lane, LR);   // round trip 2
    wg_copy_in(wl, a.w, H, tid, 1024, is_aligned16(a.w));
    if (tid == 0) wl[H] = 0.0f;
    __syncthreads();
    for (; lt < ls.end; lt += l_stride) {   // (wave-uniform)
      const bool first = lrec.len != 0;   // (wave-uniform) the row requested before the barrier
      const MbRec rec = first ? lrec : a.long_recs[lt];
      lrec.len = 0;
      if (rec.len <= 4 * 64 * BT_K) {
        if (!first) vt_long_issue(a, rec, lane, LR);
        n_act += vt_long_finish(a, rec, LR, wl, gl, g64, lane);
      } else {   // beyond the register sets (never the case for RCV1): dsgd_mb_grad_kernel's chunked walk
        BtLds L;
        L.hl = H;
        L.acc = gl;
        L.cbits = nullptr;
        L.g64 = g64;
        L.hbits = nullptr;
        L.hh = 0;
        const MbWeights wload{wl, a.w, H};
        float acc = 0.0f;
        for (int off = lane * BT_K; off < rec.len; off += 64 * BT_K) {
          int c[BT_K];
          float v[BT_K];
          mb_load8(a.mfull, rec.st + off, c, v);
#pragma unroll
          for (int k = 0; k < BT_K; ++k)
            if (off + k < rec.len) acc += filt(v[k] * wload(c[k]));
        }
        const float d = group_sum<64>(acc);
        if (!(rec.y * d < 0.0f)) {
          for (int off = lane * BT_K; off < rec.len; off += 64 * BT_K) {
            int c[BT_K];
            float v[BT_K];
            mb_load8(a.mfull, rec.st + off, c, v);
#pragma unroll
            for (int k = 0; k < BT_K; ++k)   // (hot ranks at the launch's scale, cold ranks at the cold one)
              if (off + k < rec.len) bt_add<2>(L, nullptr, c[k], v[k] * rec.y, c[k] < H ? a.qscale : a.cold_scale);
          }
          n_act += lane == 0 ? 1u : 0u;
        }
      }
    }
  }
  // active rows: ONE global atomic per workgroup.  (One per wave -- 4,096 atomics on one address when the waves of a
  // short launch all finish together -- queued up for 40 us: an ablation with every request and all the arithmetic
  // switched off still took 55 of this kernel's first 67 us.)
  n_act = wave_sum_u32(n_act);
  if (lane == 0) reinterpret_cast<unsigned int*>(strip)[0] = n_act;   // (the strips are free: the tiles are done)
  __syncthreads();
  int* mine = a.part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * a.part_stride;
  wg_copy_out(mine, gl, H, tid, 1024, is_aligned16(mine) && is_aligned16(gl));
  if (tid == 0) {
    unsigned int tot = 0;
    for (int i = 0; i < 16; ++i) tot += reinterpret_cast<const unsigned int*>(strips + i * 64)[0];
    if (tot) atomicAdd(&a.sc->n_active, (unsigned long long)tot);
  }
}

// ======================================================================================================
// K7: persistent lock-free ("Hogwild") engine -- Slave.asyncTask for many workers sharing ONE w
// ======================================================================================================
// ref: core/Slave.scala:79-111 (the loop), :177-185 / core/MasterAsync.scala:164-177 (applying updates),
//      README.md:35 (Recht et al. 2011).
// The reference gives every slave its own replica of w and gossips each update to every peer, who
// subtracts it; with all workers on one GPU the replicas collapse into a single device-resident w that
// every worker (= workgroup) reads without locks and updates with atomicAdd(w[j], -delta_j).
// One iteration of a worker:
//   draw `batch` rows of its assigned range -> gated sub-gradients on whatever w holds right now -> batch sum
//   (mini-batch engine above) -> MEAN over the batch -> support-only regulariser with s = 2*lambda*(w.ds) -> scale
//   by lr -> atomicAdd into w.  The scalar s is kept up to date incrementally (s -= 2*lambda*sum(delta_j*ds_j))
//   instead of re-reducing 47 K products per mini-batch as SparseSVM.regularize does.
// The sample of iteration i+1 does not depend on w: its row records and its non-zeros are requested while
// iteration i's update sweep runs (only the weight gather and everything after it wait for the sweep).
// SAMPLING (deliberate deviation): Slave.scala:87 draws `Random.shuffle(indices) take B`; a
// device-side Fisher-Yates of a 20,000-row range per mini-batch would cost more than the mini-batch.  The engine
// draws the affine progression rows (mul * t + off) mod n, t = 0..B-1 with gcd(mul, n) = 1 from a counter-based
// generator keyed by (seed, worker, iteration): B DISTINCT rows of the range, every row equally likely, replayable
// on the host (tests/test_gpu_parity.py hog_rows) -- but not the JVM's stream and not a uniform B-subset.
// batch == 1 is a single uniform draw, as Slave.scala:84.  The wire-level worker (wire.SlaveWorker) replays the JVM
// generator exactly for hosts that need it.
struct HogState {
  unsigned long long updates;   // mini-batch updates applied (MasterAsync counts these: MasterAsync.scala:83,171): ONE returning atomic
                                //   per update -- alone on its 128-byte line with words that are written rarely
  unsigned long long samples;   // rows whose gradient was computed    } flushed by a worker every HOG_STATS_EVERY of its iterations
  unsigned long long active;    // ... of which the gate let through    } and when it leaves (exact once the engine is joined; up to
  unsigned long long atomics;   // lane-level atomicAdd(w[j], -delta_j)  } 16 iterations per worker behind while it runs)
  int done_blocks;
  int err;                      // a sampled row fell outside the data
  // 2 * lambda * (w . ds), maintained incrementally: the OTHER returning atomic of an update, on a line of its own.  (Five
  // atomics per update on one line -- 1,280 per round of 256 workers, served one after the other -- were 8 of an
  // iteration's 45 us: profiles/r04_hogwild_phase_cycles.txt)
  alignas(128) float s_reg;
  // raised by the host (a 4-byte copy on a side stream): workers exit after their mini-batch.  Read with a RETURNING
  // atomic (fetch_or 0): an sc1 load is served by the XCD's L2, which a copy engine's write does not reach -- the flag was
  // only ever seen fresh because the atomics on the same line had just dropped it, and stopped being seen (256 workers)
  // the day the load was issued next to them instead of behind them.  On its own line: the third of three returning
  // atomics that go out together.
  alignas(128) int stop;
};

// Traced runs (dsgd_async_set_trace; parity evidence for many workers, tests/test_gpu_hogwild_trace.py): the update
// whose returning atomic on HogState::updates saw `commit - 1` leaves record [commit - 1] of HOG_TRACE_HDR + ceil(B / 32)
// 32-bit words:
//   [0] its worker   [1] that worker's iteration number (the sampler's key)
//   [2..3] read_at: the update count its weights were read at -- what thread 0's returning atomic of the worker's PREVIOUS
//          commit returned (the launch's starting count for a first iteration); the LDS copy of the hot weights is
//          requested right next to that atomic
//   [4] the regulariser scalar s = 2 lambda (w . ds) the iteration used (fp32 bits)   [5] its active rows
//   [6] seen_from (low 32 bits): the update count thread 0 read with a returning atomic BEFORE the barrier behind which the
//       workgroup requests its LDS copy of the hot weights -- every update committed up to there had landed when any
//       weight of this iteration was read (an update's atomics are drained before its commit number is drawn), so the
//       weights the iteration saw contain ALL of the updates 1..seen_from; seen_from <= read_at (the copy is requested
//       before the worker's own commit returns)   [7] flags: bit 0 = first iteration of this launch (its copy was
//       requested at start-up: only the run's starting weights are known to be contained, seen_from := 0)
//   [8..] the GATE DECISIONS of its mini-batch: bit t = row t of the sample was active (core/ml/SparseSVM.scala:27-28)
//   [8 + ceil(B / 32) ..] B floats: the x . w every sampled row was GATED ON (what bt_gate / bt_giant_row compared with
//       zero), row t of the sample at word t -- oracle/hogwild_replay.gate_check_recorded_dots holds each of them to the
//       reference's rule and to the range of the replayed weights it can have been computed from.
// A constant-step run from w = 0 is chaotic (a 1e-7 perturbation of the initial weights moves the test loss by 0.1 after 400
// updates: every margin starts AT the gate), so no replay that re-decides the gates can follow the engine.  With the
// engine's own decisions and scalar on record the oracle recomputes every update EXACTLY (oracle/hogwild_replay.py):
// the final weights must then agree to rounding -- every update applied once, with the reference's rule -- and the
// recorded decisions are checked against the margins of the replayed weights at `read_at`.
constexpr int HOG_TRACE_HDR = 8;
__host__ __device__ constexpr long long hog_trace_words(int batch) { return HOG_TRACE_HDR + (batch + 31) / 32 + batch; }

struct HogArgs {
  CsrView m;
  float* w;
  const float* ds;
  float* gcold;                 // n_workers x (dp - hl) private strips, zero between iterations
  const long long* asg_begin;
  const long long* asg_end;
  unsigned long long* it;       // per worker: iterations done so far (continues across exchange rounds)
  HogState* st;
  long long max_updates;
  unsigned long long seed;
  float lr, lambda;
  float qscale, inv_qscale;     // 2^shift / vmax2 and its inverse
  int batch, positional_bug, hl, wl, dp;   // hl: ranks with an LDS accumulator; wl: ranks with an LDS copy of w (whole 1 KiB pieces: a multiple of 256)
  int hh;                       // hog_hh(hl): the dense head of the update
  int direct;                   // few workers (<= HOG_DIRECT_MAX): a lane issues its touched accumulators' updates itself, in its own
                                //   order (one request per word, no second walk: 16.2 instead of 18.4 us per iteration at 4 workers);
                                //   many workers: by bit position (fewer requests: 36.5 instead of 40-43 us at 256)
  unsigned long long* tprof;    // PROF: [0..6] cycles by phase (batch, update head, update rounds + updates, drain, scalars + weight copy,
                                //   next tables, next requests), [12] update rounds, [15] iterations (tools/hog_prof.py)
  unsigned int* trace;          // optional (dsgd_async_set_trace): one record per mini-batch update, indexed by its commit number
  long long trace_cap;          // ... records of hog_trace_words(batch) words
  float* tdot;                  // traced runs: n_workers x batch, the x . w of the mini-batch in flight (copied into its record at the commit)
};

__device__ __forceinline__ unsigned long long hog_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned int hog_gcd32(unsigned int a, unsigned int b) {
  while (b) {
    const unsigned int t = a % b;
    a = b;
    b = t;
  }
  return a;
}

constexpr int HOG_THREADS = 512;   // 2 waves per SIMD: leaves registers and LDS for the master's concurrent loss check
constexpr int HOG_R = 4;           // work items in flight per group: 32 groups x 4 = 128 item slots per sub-batch
constexpr int HOG_CAP = HOG_THREADS / BT_G * HOG_R;
constexpr int HOG_MAX_BATCH = 4096;
constexpr int HOG_HL = 20480;      // ranks with an LDS accumulator (80 KiB)
constexpr int HOG_WL = 12288;      // ranks whose weight is gathered from an LDS copy refreshed every iteration (48 KiB;
                                   // with the tables 136 KiB: 16 KiB of dsgd_eval_kernel still fit the CU)
constexpr int HOG_SW = 4;          // accumulator slots per thread and sweep pass (and as many dimSparsity values of the NEXT pass in flight)
constexpr unsigned int HOG_ATOMIC_ONE = 1u << 13;   // active rows of a mini-batch (<= 4096) in the low 13 bits, weight atomics above
constexpr int HOG_STATS_EVERY = 16;   // iterations between a worker's flushes of its sample / active / atomic counts
constexpr int HOG_REDERIVE = 4096; // worker 0 re-derives s = 2 lambda (w . ds) from the weights every so many of its iterations

struct HogCtl {   // per iteration parity
  unsigned int mul, off;
  float s;
  int stop;
};

constexpr int HOG_DIRECT_MAX = 128;    // workers up to which the updates go out word by word (HogArgs::direct)
constexpr int HOG_TS = 8, HOG_CS = 2;   // touched accumulators / strip entries a lane takes per round of the update
constexpr int HOG_HH = HOG_THREADS * HOG_SW;   // 2,048: the dense head of the update (every accumulator read); beyond it a bitmap of touched ranks
static_assert(HOG_HL - HOG_HH <= 32 * HBIT_WORDS && HBIT_WORDS <= 2 * HOG_THREADS, "one trip of the update takes every word of the accumulators' bitmap");
__host__ __device__ constexpr int hog_hh(int hl) { return hl < HOG_HH ? hl : HOG_HH; }
__host__ __device__ constexpr int hog_hbit_words(int hl) { return hl > hog_hh(hl) ? HBIT_WORDS : 0; }
__host__ __device__ constexpr int hog_lds_words(int hl, int wl, int dp) {
  return wl + ((hl + (dp - hl + 31) / 32 + hog_hbit_words(hl) + 1) & ~1) + bt_lds_words(HOG_CAP) + 16 + 12 + HOG_MAX_BATCH / 32 + 16 + 8 + 2;
}

// Copy of w[0, wl) into LDS: each wave moves 1 KiB pieces straight from the fabric into LDS (global_load_lds_dwordx4:
// no staging registers -- the kernel has none to spare).  sc1: past this XCD's L2, which other XCDs' updates never
// reach.  The caller waits (hog_wcache_wait) and crosses a workgroup barrier before gathering.
__device__ __forceinline__ void hog_wcache_issue(const float* __restrict__ w, float* wl_lds, int wl) {
  const int lane = threadIdx.x & 63;
  const unsigned int lds_base = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)wl_lds;
  for (int piece = threadIdx.x >> 6; piece < (wl >> 8); piece += HOG_THREADS / 64) {
    const float* src = w + piece * 256 + lane * 4;
    // wave-uniform (M0); the lanes land at dst + lane * 16
    const unsigned int dst = (unsigned int)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned int)piece * 1024u));
    unsigned int keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off sc1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
  }
}
__device__ __forceinline__ void hog_wcache_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The stop flag, read where the host's copy engine wrote it: a RETURNING atomic executes at the memory side.  (Written as
// fetch_or(p, 0) the compiler turns it back into an sc1 load -- an idempotent read-modify-write -- hence the asm; the wait
// covers the two returning atomics issued just before it as well.)
__device__ __forceinline__ int hog_read_stop(int* p) {
  int v;
  const int zero = 0;
  asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}
// the update counter, read where the commits' returning atomics execute (traced runs only: seen_from)
__device__ __forceinline__ unsigned long long hog_read_u64(unsigned long long* p) {
  unsigned long long v;
  const unsigned long long zero = 0ull;
  asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p), "v"(zero) : "memory");
  return v;
}

// the sampler of iteration `it`: rows base + (mul * t + off) mod n_k, t = 0..B-1 (distinct rows)
__device__ __forceinline__ void hog_sampler(const HogArgs& a, int worker, unsigned long long it, unsigned int n_k,
                                            HogCtl* out) {
  const unsigned long long key = hog_mix(a.seed ^ hog_mix((unsigned long long)worker * 0x100000001B3ull + it));
  unsigned int mul = 1u + (unsigned int)(hog_mix(key) % (unsigned long long)n_k);
  while (hog_gcd32(mul, n_k) != 1u) mul = mul % n_k + 1u;
  out->mul = mul;
  out->off = (unsigned int)(hog_mix(key ^ 0xABCDEF12345ull) % (unsigned long long)n_k);
}

// Registers: the master's loss check (dsgd_eval_kernel, launched with 256-lane blocks while this engine runs: one wave
// of 40 VGPRs per SIMD) must stay co-resident with the two waves per SIMD of this kernel (225 VGPRs -> 232 allocated):
// 2 x 232 + 40 <= 512.
// TRACE: the traced form (dsgd_async_set_trace) is its own instantiation -- the recording costs 6 registers the untraced
// engine does not have (236 -> 240 allocated: 2 x 240 + 40 > 512).
template <bool PROF, bool TRACE>   // PROF (DSGD_PLAN_PROF=1, tuning runs): thread 0 of worker 0 counts the cycles of an iteration by phase
__global__ void __launch_bounds__(HOG_THREADS) dsgd_hogwild_kernel(HogArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  BtLds L;
  L.hl = a.hl;
  L.g64 = nullptr;
  L.hbits = nullptr;
  L.hh = 0;
  float* wl = lds;                                           // copy of w[0, a.wl), 16-byte aligned
  L.acc = reinterpret_cast<int*>(lds + a.wl);
  const int n_cw = (a.dp - a.hl + 31) / 32;                  // bitmap words of the cold strip (0 when dp <= hl)
  L.cbits = reinterpret_cast<unsigned int*>(lds + a.wl + a.hl);
  const int n_hw = hog_hbit_words(a.hl);                     // bitmap words of the accumulators beyond the dense head
  L.hh = a.hh;
  L.hbits = L.cbits + n_cw;
  int* tables = reinterpret_cast<int*>(lds) + a.wl + ((a.hl + n_cw + n_hw + 1) & ~1);
  bt_carve(L, tables, HOG_CAP);
  float* red = reinterpret_cast<float*>(tables + bt_lds_words(HOG_CAP));   // 8 floats + 8 counters
  unsigned int* redn = reinterpret_cast<unsigned int*>(red + 8);
  HogCtl* ctl = reinterpret_cast<HogCtl*>(red + 16);                       // 3 slots (iteration mod 3: this one, the next, the one being sampled)
  unsigned int* gmask = reinterpret_cast<unsigned int*>(red + 28);         // traced runs: gate decisions of the mini-batch
  unsigned int* tp = gmask + HOG_MAX_BATCH / 32;                           // PROF: phase sums [0..13], [14] the last stamp (32-bit), [15] iterations
  unsigned int* stl = tp + 16;                                             // thread 0: samples / active rows / weight atomics not yet flushed;
                                                                           //   [4..5] the update count its weights were read at (a register
                                                                           //   pair in every lane otherwise: the kernel has none to spare);
                                                                           //   traced runs: [3] / [6] seen_from of the next / this iteration, [7] flags
  unsigned int* ctl_rec = stl + 8;                                         // traced runs: the record number of the update just committed (2 words)
  const int tid = threadIdx.x;
  const int worker = blockIdx.x;
  const bool prof = PROF && worker == 0 && tid == 0;
  auto stamp = [&](int i) {
    if (PROF && prof) {
      const unsigned int now = (unsigned int)__builtin_readcyclecounter();
      tp[i] += now - tp[14];
      tp[14] = now;
    }
  };
  if (PROF && prof)
    for (int i = 0; i < 16; ++i) tp[i] = 0u;
  if (tid < HOG_MAX_BATCH / 32) gmask[tid] = 0u;
  if (tid < 4) stl[tid] = 0u;
  unsigned int* const gm = TRACE ? gmask : nullptr;
  float* const gdot = TRACE ? a.tdot + (long long)worker * a.batch : nullptr;
  if (tid == 0) {   // traced runs: seen_from / flags of the record in the making (bit 0: the first iteration of this launch)
    stl[6] = 0u;
    stl[7] = 1u;
  }
  const long long begin = a.asg_begin[worker];
  const unsigned int n_k = (unsigned int)(a.asg_end[worker] - begin);   // < 2^31 rows per context
  const long long base = a.positional_bug ? 0 : begin;   // ref: core/Slave.scala:87 indexes `data` by POSITION
  const double inv_n = 1.0 / (double)n_k;
  float* gc = a.gcold + (long long)worker * (a.dp > a.hl ? a.dp - a.hl : 1);
  for (int j = tid; j < a.hl + n_cw + n_hw; j += HOG_THREADS) L.acc[j] = 0;   // accumulators and both bitmaps

  const int B = a.batch;
  const float fB = (float)B;
  unsigned long long it = a.it[worker];
  int sl = 0;   // control slot of iteration `it` (rotates 0, 1, 2)
  hog_wcache_issue(a.w, wl, a.wl);   // (lands under the start-up loads below; waited for in front of the first barrier)
  // thread 0 carries the shared scalars between iterations: what its own returning atomics saw
  float s = 0.0f;
  if (tid == 0) {
    const unsigned long long u = __hip_atomic_load(&a.st->updates, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stl[4] = (unsigned int)u;   // (between commits: the update count this iteration's weights were read at)
    stl[5] = (unsigned int)(u >> 32);
    s = __hip_atomic_load(&a.st->s_reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int stop = hog_read_stop(&a.st->stop);
    HogCtl* c0 = &ctl[sl];
    hog_sampler(a, worker, it, n_k, c0);
    c0->s = s;
    c0->stop = stop != 0 || (long long)u >= a.max_updates;
    hog_sampler(a, worker, it + 1, n_k, &ctl[sl == 2 ? 0 : sl + 1]);
  }
  hog_wcache_wait();
  __syncthreads();
  // The weight gather: ~89 % of a batch's non-zeros have a rank below wl and read this iteration's LDS copy -- 7,600
  // scattered coherent 4-byte loads per batch of 100 became 768 coalesced 64-byte lines + ~850 scattered ones (with
  // 256 workers the scattered form alone held an iteration at 100 us; served from L2 -- stale, not an option -- it
  // was 59 us: profiles/README.md).  The copy is as fresh as the gather was: taken after this workgroup's own
  // previous update has been performed, other workers' updates as they happen to have landed.
  typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
  const int wl_n = a.wl;
  auto wload = [&](int c) -> float {
    const bool hot = c < wl_n;
    float v = ((lds_cvfloat*)wl)[hot ? c : 0];
    // agent scope: other workgroups update w concurrently -- a plain load could be served by a stale L1 / L2 line forever
    if (!hot) v = __hip_atomic_load(&a.w[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  auto row_at = [&](unsigned long long mul, unsigned long long off, int t) -> long long {
    const unsigned long long x = mul * (unsigned long long)t + off;          // < 2^44: exact in a double
    long long r = (long long)x - (long long)((unsigned long long)((double)x * inv_n)) * (long long)n_k;
    if (r < 0) r += n_k;
    if (r >= (long long)n_k) r -= n_k;
    return base + r;
  };
  if (ctl[sl].stop) {
    if (tid == 0) atomicAdd(&a.st->done_blocks, 1);
    return;
  }
  // prologue: stage the first iteration's sub-batch
  BtItems<HOG_R> items;
  int2 bd;
  {
    const unsigned long long mul = ctl[sl].mul, off = ctl[sl].off;
    const BtRow row = bt_rows_issue<HOG_CAP>(a.m, B, 0, [&](int t) { return row_at(mul, off, t); }, &a.st->err);
    bd = bt_build<HOG_THREADS, HOG_CAP>(L, B, 0, row);
    if (bd.x > 0) bt_items_issue<HOG_THREADS, HOG_R>(a.m, L, bd.y, items);
  }
  if (PROF && prof) tp[14] = (unsigned int)__builtin_readcyclecounter();
  for (;;) {
    const HogCtl cur = ctl[sl];
    const int sl1 = sl == 2 ? 0 : sl + 1, sl2 = sl == 0 ? 2 : sl - 1;   // the slots of iterations it + 1 and it + 2 (= it - 1: free)
    // The sampler of iteration + 2 -- two 64-bit remainders and a gcd loop, ~3,000 cycles of one lane; in front of the
    // commit's atomics until round 6: 1.4 us on every worker's critical path -- by a lane of its own, HERE: its wave is
    // about to wait for the sub-batch's non-zeros anyway.
    if (tid == 64) hog_sampler(a, worker, it + 2, n_k, &ctl[sl2]);
    const unsigned long long mul = cur.mul, off = cur.off;
    const float s_it = cur.s;
    const bool add_s = (s_it != 0.0f) && (fabsf(s_it) > DSGD_EPS);
    auto row_of = [&](int t) { return row_at(mul, off, t); };
    // phase 1: gated sub-gradient sum of the batch (ref: core/Slave.scala:93-98): the staged sub-batch ...
    unsigned int n_act = 0;
    int done = 0;
    if (bd.x > 0) {
      bt_items_dot<HOG_THREADS, HOG_R>(L, items, wload);
      __syncthreads();
      n_act += bt_gate(L, bd.x, gm, 0, gdot);
      __syncthreads();
      bt_scatter<HOG_R, 3>(L, gc, items, a.qscale);
      done = bd.x;
    }
    // ... and whatever did not fit its item slots (long rows, batches beyond 128 rows)
    if (done < B) n_act += bt_batch<HOG_THREADS, HOG_R, 3>(a.m, L, gc, B, done, row_of, wload, a.qscale, &a.st->err, gm, gdot);
    // the next iteration's sample does not depend on w: request its row records now
    const HogCtl nxt = ctl[sl1];
    const BtRow row_n = bt_rows_issue<HOG_CAP>(a.m, B, 0, [&](int t) { return row_at(nxt.mul, nxt.off, t); }, &a.st->err);
    // dimSparsity of this lane's slots of the dense head (one buffer resource over the head: beyond it a load returns zero);
    // requested with the row records, used behind the barrier
    const __amdgpu_buffer_rsrc_t ds_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ds), 0, a.hh * 4, 0x00020000);
    float dsh[HOG_SW];
#pragma unroll
    for (int e = 0; e < HOG_SW; ++e)
      dsh[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ds_rs, tid * 4, e * HOG_THREADS * 4, 0));
    __syncthreads();
    stamp(0);
    // phase 2: mean, regularise on the support, scale, subtract from the shared w (ref: Slave.scala:98-101).
    // Round 6: the update WALKS THE TOUCHED RANKS.  Rounds 3-5 swept every LDS accumulator (40 slots per lane, twice; the
    // dimSparsity value of every slot requested a pass ahead: ten dependent round trips per iteration, 14.5 of its 38 us,
    // profiles/r05_hogwild_phase_cycles.txt) and then the cold strip's bitmap one bit -- one round trip -- at a time
    // (5.5 us).  Now (profiles/r06_hogwild_notes.txt):
    //   1. a dense HEAD of hh = 2,048 ranks, 4 slots per lane, their dimSparsity values requested with the row records;
    //   2. the other accumulators through a bitmap set by the scatter (bt_add<3>), the cold strip through its own: a lane
    //      takes up to eight touched accumulators and two strip entries at a time and requests everything they need
    //      TOGETHER (dimSparsity values; the strip entries with their take-and-clear exchange) -- one round trip for
    //      nearly every lane; the accumulators' deltas go back into their slots;
    //   3. the updates, with no load behind them: the head, then the accumulators BY BIT POSITION (consecutive lanes =
    //      consecutive ranks: the memory side serves ~21 G requests/s whether a request carries one word of a line or
    //      sixteen, tools/microbench7.hip -- with 256 workers an iteration is bound by that rate: without any update
    //      it takes 24 us, with them 37), the strip's as they are computed.
    // The same arithmetic per rank as before; the terms of the regulariser's increment are added in another order.
    float ds_acc = 0.0f;
    auto hot_delta = [&](int q) -> float {
      float g = filt(((float)q * a.inv_qscale) / fB);   // Vec.mean divides (ref: math/Vec.scala:139)
      if (add_s && g != 0.0f) g = filt(g + s_it);
      return filt(g * a.lr);
    };
    float dh[HOG_SW];
#pragma unroll
    for (int e = 0; e < HOG_SW; ++e) {
      const int j = e * HOG_THREADS + tid;
      const int q = j < a.hh ? L.acc[j] : 0;
      dh[e] = 0.0f;
      if (q != 0) {
        L.acc[j] = 0;
        dh[e] = hot_delta(q);
        if (dh[e] != 0.0f) {
          ds_acc += dh[e] * dsh[e];
          n_act += HOG_ATOMIC_ONE;      // (counted in the upper bits of the active-row counter)
        }
      }
    }
    stamp(1);
    bool walked = false;
    unsigned long long hb_all = 0ull;   // this lane's two words of the accumulators' bitmap (the first trip takes them all)
    for (int wb = 0; wb < n_cw || !walked; wb += 2 * HOG_THREADS) {   // (RCV1: 837 words of cold bitmap -- ONE trip)
      // two bitmap words of each kind per lane, taken as one 64-bit mask
      unsigned long long hb = 0ull, cb = 0ull;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int wd = wb + h * HOG_THREADS + tid;
        if (wd < n_hw) {   // (n_hw <= 2 x HOG_THREADS: the first trip takes them all)
          const unsigned int v = L.hbits[wd];
          if (v) L.hbits[wd] = 0u;
          hb |= (unsigned long long)v << (32 * h);
        }
        if (wd < n_cw) {
          const unsigned int v = L.cbits[wd];
          if (v) L.cbits[wd] = 0u;
          cb |= (unsigned long long)v << (32 * h);
        }
      }
      if (wb == 0) hb_all = hb;
      do {
        int jt[HOG_TS], jc[HOG_CS];
#pragma unroll
        for (int e = 0; e < HOG_TS; ++e) {
          jt[e] = -1;
          if (hb) {
            const int b = __builtin_ctzll(hb);
            hb &= hb - 1ull;
            jt[e] = a.hh + (b & 31) * HBIT_WORDS + (b >> 5) * HOG_THREADS + tid;
          }
        }
#pragma unroll
        for (int e = 0; e < HOG_CS; ++e) {
          jc[e] = -1;
          if (cb) {
            const int b = __builtin_ctzll(cb);
            cb &= cb - 1ull;
            jc[e] = (wb + (b >> 5) * HOG_THREADS + tid) * 32 + (b & 31);
          }
        }
        float dt[HOG_TS], dst[HOG_TS], gs[HOG_CS], dsc[HOG_CS];
#pragma unroll
        for (int e = 0; e < HOG_CS; ++e) {
          gs[e] = dsc[e] = 0.0f;
          if (jc[e] >= 0) {
            // take-and-clear the private strip entry (written with L2 atomics of this workgroup: read it there)
            gs[e] = __hip_atomic_exchange(&gc[jc[e]], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            dsc[e] = a.ds[a.hl + jc[e]];
          }
        }
#pragma unroll
        for (int e = 0; e < HOG_TS; ++e) {
          dt[e] = dst[e] = 0.0f;
          if (jt[e] >= 0) {
            dst[e] = a.ds[jt[e]];
            const int q = L.acc[jt[e]];
            dt[e] = q != 0 ? hot_delta(q) : 0.0f;
            L.acc[jt[e]] = a.direct ? 0 : __float_as_int(dt[e]);   // (0.0f = all bits clear: nothing to apply)
          }
        }
        // ---- everything requested.  EVERY loaded value is consumed before the first update goes out: an update between
        // two uses made the compiler wait for the update in front of it (conditional updates: it cannot count what is
        // in flight behind a load) -- fourteen acknowledgements in a row, 11.7 us (ISA + phase counters).
#pragma unroll
        for (int e = 0; e < HOG_TS; ++e) {
          if (dt[e] != 0.0f) {
            ds_acc += dt[e] * dst[e];
            n_act += HOG_ATOMIC_ONE;
          }
        }
        float dc[HOG_CS];
#pragma unroll
        for (int e = 0; e < HOG_CS; ++e) {
          dc[e] = 0.0f;
          if (jc[e] < 0) continue;
          float g = filt(gs[e] / fB);
          if (g == 0.0f) continue;
          if (add_s) g = filt(g + s_it);
          dc[e] = filt(g * a.lr);
          if (dc[e] != 0.0f) {
            ds_acc += dc[e] * dsc[e];
            n_act += HOG_ATOMIC_ONE;
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing is pending: said aloud, so that no wait lands between the updates)
        // ---- the strip's updates (few workers: the accumulators' too), back to back ----
        walked = true;
        if (a.direct) {
#pragma unroll
          for (int e = 0; e < HOG_TS; ++e)
            if (dt[e] != 0.0f) atomicAdd(&a.w[jt[e]], -dt[e]);
        }
#pragma unroll
        for (int e = 0; e < HOG_CS; ++e)
          if (dc[e] != 0.0f) atomicAdd(&a.w[a.hl + jc[e]], -dc[e]);
        if (PROF && prof) ++tp[12];   // rounds
      } while (hb | cb);
    }
    // The accumulators' updates by BIT POSITION: bit b of the words of 512 consecutive lanes = 512 consecutive ranks, so a
    // wave's instruction covers whole lines where neighbours were touched (the memory side serves ~21 G REQUESTS/s,
    // a line with sixteen words as fast as one with a single word: tools/microbench7.hip).  The deltas wait in the
    // accumulators' own slots.
    {
#pragma unroll
      for (int e = 0; e < HOG_SW; ++e)
        if (dh[e] != 0.0f) atomicAdd(&a.w[e * HOG_THREADS + tid], -dh[e]);   // lock-free update of the ONE weight vector
      const int nb = a.direct ? 0 : (a.hl - a.hh + HBIT_WORDS - 1) / HBIT_WORDS;   // bit positions in use (RCV1: 18)
      for (int b0 = 0; b0 < nb; b0 += 6) {   // (twelve slots read together, then their updates: no LDS round trip per update)
        int d[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int b = b0 + (i >> 1), h = i & 1;
          const int j = a.hh + b * HBIT_WORDS + h * HOG_THREADS + tid;
          d[i] = 0;
          if (b < nb && ((hb_all >> (32 * h + b)) & 1ull)) {
            d[i] = L.acc[j];
            L.acc[j] = 0;
          }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          const int j = a.hh + (b0 + (i >> 1)) * HBIT_WORDS + (i & 1) * HOG_THREADS + tid;
          if (d[i] != 0) atomicAdd(&a.w[j], -__int_as_float(d[i]));
        }
      }
    }
    stamp(2);
    // one returning atomic per workgroup for the incremental regulariser scalar and the update counter: thread 0
    // continues with what they saw (no separate loads of the shared scalars in the next iteration)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ds_acc += __shfl_xor(ds_acc, o, 64);
    n_act = wave_sum_u32(n_act);
    if ((tid & 63) == 0) {
      red[tid >> 6] = ds_acc;
      redn[tid >> 6] = n_act;
    }
    // traced runs: the count of updates that have LANDED before any weight of the next iteration is requested (the copy of
    // the hot weights goes out behind the barrier below; the value is back before this thread reaches it)
    if (TRACE && tid == 0) stl[3] = (unsigned int)hog_read_u64(&a.st->updates);
    // Every wave waits for the acknowledgement of ITS updates of w, then the barrier: the commit number below is drawn when
    // the whole update has been performed -- what `seen_from` and the next weight copy rely on.  (Said in asm: this
    // compiler's __syncthreads() waits for LDS operations only -- `s_waitcnt lgkmcnt(0); s_barrier` in the ISA -- and
    // without the wait a commit could overtake the updates of the other waves: tests/test_gpu_hogwild_trace.py's
    // small-lag statement found decisions taken on weights that missed part of a COMMITTED update, 38 of 141,700.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(3);
    // The scalar s is kept by fp32 atomic increments (one per mini-batch, plus dsgd_update_grad's foreign updates): over
    // 10^6+ updates the rounding of every add accumulates like a random walk.  Every HOG_REDERIVE iterations worker 0
    // recomputes w . ds from the weights as they are now (fp64 partial sums) and adds the difference to the shared
    // scalar: the accumulated drift is removed; what remains is the fuzz of the <= n_workers updates in flight
    // around the recomputation, which does not accumulate.  Block-uniform condition.
    const bool rederive = worker == 0 && ((it + 1) & (unsigned long long)(HOG_REDERIVE - 1)) == 0;
    double* dred = reinterpret_cast<double*>(L.pdot);   // (free between the gate and the next iteration's products)
    if (rederive) {
      double part = 0.0;
      for (int j = tid; j < a.dp; j += HOG_THREADS)
        part += (double)__hip_atomic_load(&a.w[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (double)a.ds[j];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
      if ((tid & 63) == 0) dred[tid >> 6] = part;
      __syncthreads();
    }
    // next iteration's copy of the weights, requested while thread 0 exchanges the shared scalars
    hog_wcache_issue(a.w, wl, a.wl);
    if (tid == 0) {
      float tot = 0.0f;
      unsigned int na = 0;
      for (int i = 0; i < HOG_THREADS / 64; ++i) {
        tot += red[i];
        na += redn[i];
      }
      const float ds_term = -2.0f * a.lambda * tot;
      // the two returning atomics and the stop flag go out TOGETHER: one round trip
      const float s_seen = atomicAdd(&a.st->s_reg, ds_term);
      const unsigned long long u_seen = atomicAdd(&a.st->updates, 1ull);
      const int stop = hog_read_stop(&a.st->stop);
      s = s_seen + ds_term;
      if (rederive) {
        double dot = 0.0;
        for (int i = 0; i < HOG_THREADS / 64; ++i) dot += dred[i];
        const float corr = (float)(2.0 * (double)a.lambda * dot) - s;
        s = atomicAdd(&a.st->s_reg, corr) + corr;
      }
      const unsigned long long read_at = ((unsigned long long)stl[5] << 32) | stl[4];
      const unsigned long long u = u_seen + 1ull;
      stl[4] = (unsigned int)u;
      stl[5] = (unsigned int)(u >> 32);
      if (TRACE) {   // (one lane, once per mini-batch; the decisions were taken several barriers ago)
        const int mw = (B + 31) >> 5;
        const bool kept = (long long)u <= a.trace_cap;
        if (kept) {
          unsigned int* rec = a.trace + (u - 1) * (unsigned long long)hog_trace_words(B);
          rec[0] = (unsigned int)worker;
          rec[1] = (unsigned int)it;
          rec[2] = (unsigned int)read_at;
          rec[3] = (unsigned int)(read_at >> 32);
          rec[4] = __float_as_uint(s_it);
          rec[5] = na & (HOG_ATOMIC_ONE - 1u);
          rec[6] = stl[6];   // seen_from of THIS iteration (the count read in front of the barrier its weight copy went out behind)
          rec[7] = stl[7];
          for (int i = 0; i < mw; ++i) rec[HOG_TRACE_HDR + i] = gmask[i];
        }
        for (int i = 0; i < mw; ++i) gmask[i] = 0u;   // (the next gate is behind the barriers below)
        stl[6] = stl[3];   // ... of the next iteration: read in front of the drain barrier above
        stl[7] = 0u;
        ctl_rec[0] = kept ? (unsigned int)(u - 1) : 0xFFFFFFFFu;   // the record every thread adds its row's x . w to, below
        ctl_rec[1] = kept ? (unsigned int)((u - 1) >> 32) : 0xFFFFFFFFu;
      }
      // statistics: kept in LDS, flushed every HOG_STATS_EVERY iterations and when the worker leaves
      const bool leaving = stop != 0 || (long long)u >= a.max_updates;
      stl[0] += (unsigned int)B;
      stl[1] += na & (HOG_ATOMIC_ONE - 1u);
      stl[2] += na >> 13;
      if (leaving || ((it + 1) & (unsigned long long)(HOG_STATS_EVERY - 1)) == 0) {
        atomicAdd(&a.st->samples, (unsigned long long)stl[0]);
        atomicAdd(&a.st->active, (unsigned long long)stl[1]);
        atomicAdd(&a.st->atomics, (unsigned long long)stl[2]);
        stl[0] = stl[1] = stl[2] = 0u;
      }
      HogCtl* cn = &ctl[sl1];
      cn->s = s;
      cn->stop = leaving;
    }
    stamp(4);
    // The tables of the next sub-batch (its row records were requested in front of the update and waited for with the
    // update's own loads; this iteration's tables are no longer needed): LDS only.  Its two barriers publish the next iteration's control words and the
    // weight copy.  Then the request for its non-zeros -- BEHIND the update: vmcnt retires in order, and in front of the
    // update's requests these (random rows of a matrix of gigabytes) held every one of them back for ~7 us.
    hog_wcache_wait();
    const int2 bd_n = bt_build<HOG_THREADS, HOG_CAP>(L, B, 0, row_n);
    stamp(5);
    if (TRACE) {   // the x . w of this mini-batch (written by bt_gate several barriers ago; read past L1) into its record
      const unsigned long long r = ((unsigned long long)ctl_rec[1] << 32) | ctl_rec[0];
      if (r != ~0ull) {
        float* dst = reinterpret_cast<float*>(a.trace + r * (unsigned long long)hog_trace_words(B) + HOG_TRACE_HDR + ((B + 31) >> 5));
        for (int t = tid; t < B; t += HOG_THREADS) dst[t] = __hip_atomic_load(&gdot[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (bd_n.x > 0) bt_items_issue<HOG_THREADS, HOG_R>(a.m, L, bd_n.y, items);
    stamp(6);
    ++it;
    bd = bd_n;
    if (PROF && prof) ++tp[15];
    sl = sl1;
    if (ctl[sl].stop) break;
  }
  if (PROF && prof && a.tprof) {
    for (int i = 0; i < 14; ++i) a.tprof[i] += tp[i];
    a.tprof[15] += tp[15];
  }
  if (tid == 0) {
    a.it[worker] = it;
    atomicAdd(&a.st->done_blocks, 1);
  }
}

// ---- cross-GPU asynchronous mode: replicas + periodic exchange of the summed updates ---------------------------
// ref: core/Slave.scala:103-105 (every update is gossiped to every peer), :177-185 (a peer subtracts it),
// core/MasterAsync.scala:164-177.  One single-w engine per GPU; every `exchange_every` local updates the replicas
// all-reduce what each subtracted since the last exchange and subtract the PEERS' part on top of their own.
__global__ void __launch_bounds__(1024) dsgd_exchange_delta_kernel(const float* __restrict__ w,
                                                                  const float* __restrict__ wprev,
                                                                  float* __restrict__ dsum, float* __restrict__ dlocal,
                                                                  int dp) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < dp; j += gridDim.x * blockDim.x) {
    const float d = wprev[j] - w[j];
    dsum[j] = d;     // all-reduced in place by the caller
    dlocal[j] = d;
  }
}
// The peers' part dsum - dlocal is subtracted from this replica's weights; s follows; wprev <- w.  One workgroup
// (fixed-order sum).  With several ranks the new weights are formed as wprev - dsum: the same number as
// w - (dsum - dlocal) up to rounding (w = wprev - dlocal, the engine is quiescent between the delta kernel and this
// one), but built from two vectors that are bit-identical on every rank (wprev by induction, dsum from the
// all-reduce) -- the replicas leave every exchange BIT-IDENTICAL instead of drifting apart by an ulp per round
// (found by executing world = 2: tests/test_gpu_world2.py).  With a single rank the peers' part is exactly zero and
// the replica keeps its own weights bit for bit.
__global__ void __launch_bounds__(1024) dsgd_exchange_apply_kernel(float* __restrict__ w, float* __restrict__ wprev,
                                                                  const float* __restrict__ dsum,
                                                                  const float* __restrict__ dlocal,
                                                                  const float* __restrict__ ds, int dp, float lambda,
                                                                  HogState* st, int world) {
  __shared__ float red[16];
  float acc = 0.0f;
  for (int j = threadIdx.x; j < dp; j += blockDim.x) {
    float wn = w[j];
    if (world > 1) {
      const float wo = wn;
      wn = filt(wprev[j] - dsum[j]);
      if (wn != wo) {
        w[j] = wn;
        acc += (wo - wn) * ds[j];
      }
    } else {
      const float o = dsum[j] - dlocal[j];   // exactly 0 with a single rank
      if (o != 0.0f) {
        wn = filt(wn - o);
        w[j] = wn;
        acc += o * ds[j];
      }
    }
    wprev[j] = wn;
  }
  const float tot = block_sum_1024(acc, red);
  if (threadIdx.x == 0 && tot != 0.0f) st->s_reg += -2.0f * lambda * tot;
}

// ======================================================================================================
// K1p: small-batch synchronous steps (the reference's batch-size 100-200) as ONE persistent workgroup
// ======================================================================================================
// ref: core/Master.scala:179-199 (the batch closure), core/Slave.scala:142-157, application.conf:15 (batch-size 100).
// A B = 100 step is 60 KB of CSR: as separate launches (gradient rows -> regularise -> sum -> ticketed apply) it
// took 31-41 us, all of it dependent-launch and cross-workgroup latency (profiles/README.md: a hipGraph of the same
// chain changed nothing).  Here ONE 1024-lane workgroup owns the weights for the steps [step_begin, step_end) of a
// resident plan and runs them back to back with nothing but workgroup barriers in between:
//   * the hl hottest weights and their dimSparsity values live in LDS next to the fixed-point accumulators: the
//     weight gather of ~88 % of the non-zeros and the whole update sweep of those ranks never leave the CU (global w
//     is written through, never read back for them);
//   * per step: mini-batch engine on the worker's index list (snapshot of w), then the sweep turns the batch sum into
//     g = regularize(sum, w) on its support (SparseSVM.scala:31) and applies w <- w - lr * g there (the mean over one
//     worker, Master.scala:194-197).  Steps with several hosted workers are not run here: their batches would queue
//     up in the one workgroup (3 x 100 rows: 44 us) while dsgd_mb_grad_kernel gives every worker its own workgroups;
//   * the NEXT batch's index list, row records and non-zeros are requested while the current batch is swept -- only
//     the weight gather and the gate wait for the update (the reference's synchronous semantics are kept exactly:
//     every gradient of a step sees the weights of the previous step);
//   * the regulariser scalar s = 2*lambda*(w . ds) is computed exactly (fp64, all D+1 products) when the launch
//     starts and then carried in fp64 through the updates of the touched coordinates -- closer to the fp64 reference
//     than the fp32 re-reduction of the multi-launch path, and no 47 K-element pass per step.
// The gradient is deterministic: integer accumulation, fixed sweep order (the multi-launch path used fp32 L2
// atomics in arrival order).
struct PlanArgs {
  CsrView m;
  float* w;
  const float* ds;
  float* gcold;              // dp - hl floats, zero between batches
  const int* idx;
  const WorkSeg* segs;       // one per step
  DevScalars* sc;
  unsigned long long* tprof; // optional (tuning runs): 16 words -- shader-clock cycles of thread 0 in nine phases of a batch, [15] = steps
  unsigned long long* mail;  // optional (per-request steps): host-mapped {n_active, err, mail_seq} written when the launch ends
  unsigned long long mail_seq;
  long long step_begin, step_end;
  float lr, lambda;
  int vexp, dp;
};

constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_R = 3;        // 64 groups x 3 = 192 item slots per sub-batch (100 RCV1-like rows: 118 +- 6 items)
constexpr int PLAN_CAP = PLAN_THREADS / BT_G * PLAN_R;
constexpr int PLAN_HL = 11264;   // LDS-resident ranks (accumulator + weight + dimSparsity: 12 bytes each)
__host__ __device__ constexpr int plan_hl(int dp) { return PLAN_HL < dp ? PLAN_HL : dp; }
__host__ __device__ constexpr int plan_lds_words(int dp) {
  const int hl = plan_hl(dp);
  const int n_cw = (dp - hl + 31) / 32;
  return ((3 * hl + n_cw + 1) & ~1) + 2 * bt_lds_words(PLAN_CAP) + 32 + 8;
}

// sum over the 64 lanes of a wave, valid in lane 63; DPP moves of the two halves (a __shfl_xor of a double is two
// ds_bpermute round trips per step: 1,500 cycles per reduction, measured as "reduce" in the phase counters)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, ROW_MASK, 0xf, false);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);   // (lanes without a source read +0.0)
}
__device__ __forceinline__ double wave_sum_f64_lane63(double v) {
  v += dpp_get_f64<0x111, 0xf>(v);   // row_shr:1
  v += dpp_get_f64<0x112, 0xf>(v);   // row_shr:2
  v += dpp_get_f64<0x114, 0xf>(v);   // row_shr:4
  v += dpp_get_f64<0x118, 0xf>(v);   // row_shr:8
  v += dpp_get_f64<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_get_f64<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v;
}
// workgroup sum in two halves so that the barriers of other code in between can be shared:
// publish (every wave leaves its partial in LDS) ... any workgroup barrier ... collect (fixed order: reproducible)
__device__ __forceinline__ void block_sum_f64_publish(double v, double* red /* 16 doubles of LDS */) {
  v = wave_sum_f64_lane63(v);
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
}
__device__ __forceinline__ double block_sum_f64_collect(const double* red) {
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  __syncthreads();
  block_sum_f64_publish(v, red);
  __syncthreads();
  return block_sum_f64_collect(red);
}

// Every list of the launch fits the staged sub-batch (at most PLAN_CAP rows and PLAN_CAP work items): the host knows
// the row lengths and sends anything else down the multi-workgroup path (measured: the stage-by-stage general path
// inlined here cost 80+ spilled registers in the main loop and ran B = 200..1000 slower than the multi-launch kernels).
__global__ void __launch_bounds__(PLAN_THREADS) dsgd_plan_kernel(PlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hl = plan_hl(a.dp);
  const int n_cw = (a.dp - hl + 31) / 32;
  BtLds L;
  L.hl = hl;
  L.g64 = nullptr;
  L.hbits = nullptr;
  L.hh = 0;
  L.acc = reinterpret_cast<int*>(lds);
  float* wl = lds + hl;                       // hot weights
  float* dsl = lds + 2 * hl;                  // hot dimSparsity
  L.cbits = reinterpret_cast<unsigned int*>(lds + 3 * hl);
  int* tables = reinterpret_cast<int*>(lds) + ((3 * hl + n_cw + 1) & ~1);
  // two sets of sub-batch tables (batch parity): the next batch's tables are built while the current one's are in use
  BtLds L2 = L;
  bt_carve(L, tables, PLAN_CAP);
  bt_carve(L2, tables + bt_lds_words(PLAN_CAP), PLAN_CAP);
  L2.misc = L.misc;   // (one scratch area: a build is over before the next one starts)
  double* red = reinterpret_cast<double*>(tables + 2 * bt_lds_words(PLAN_CAP));   // 16 doubles
  const int tid = threadIdx.x;
  for (int j = tid; j < hl; j += PLAN_THREADS) {
    L.acc[j] = 0;
    wl[j] = a.w[j];
    dsl[j] = a.ds[j];
  }
  for (int j = tid; j < n_cw; j += PLAN_THREADS) L.cbits[j] = 0u;
  // exact w . ds of the weights this launch starts from
  double dot_part = 0.0;
  for (int j = tid; j < a.dp; j += PLAN_THREADS) dot_part += (double)a.w[j] * (double)a.ds[j];
  double dot = block_sum_f64(dot_part, red);   // (also the barrier behind the LDS initialisation)
  unsigned long long n_act_total = 0;
  unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int i, unsigned long long& last) {   // tuning runs only: cycles since the previous stamp -> tp[i]
    if (a.tprof) {
      const unsigned long long now = __builtin_readcyclecounter();
      tp[i] += now - last;
      last = now;
    }
  };

  typedef __attribute__((address_space(3))) const volatile float lds_cvfloat;
  auto wload = [&](int c) -> float {   // hot ranks from LDS, the tail from L1/L2 (two loads + a select of VALUES: w_at)
    const bool hot = c < hl;
    float v = ((lds_cvfloat*)wl)[hot ? c : 0];
    if (!hot) v = a.w[c];
    return v;
  };
  // new weight of coordinate j given the summed gradient; returns the change of w[j] * ds[j] (fp32 per thread -- a
  // thread adds at most a dozen such terms per step -- and fp64 from the workgroup reduction on: the fp64 form cost
  // five half-rate instructions per slot of the sweep, which is issue-bound)
  auto step_w = [&](float g, float wo, float dsj, float& wn) -> float {
    const float updv = filt(g * a.lr);           // Vec.mean over one worker is the identity (ref: math/Vec.scala:139); learningRate * grad (ref: core/Master.scala:197)
    wn = filt(wo - updv);
    return (wn - wo) * dsj;
  };

  const long long n_batches = a.step_end - a.step_begin;
  const WorkSeg* segs = a.segs + a.step_begin;
  // The list descriptors of the batches n .. n+3 live in registers (a sliding window); the descriptor of batch n+4 is
  // requested at the top of iteration n with a VECTOR load: a scalar load would share lgkmcnt with the LDS traffic
  // of the whole iteration and stall the first LDS wait behind it for a memory round trip -- four such loads per
  // batch were ~3 us of the 15 us step (phase counters, profiles/README.md).
  auto seg_load = [&](long long n, long long& beg, long long& end) {   // raw: no arithmetic on the loaded words here
    beg = 0;
    end = 0;
    if (n < n_batches) {
      const WorkSeg* p = segs + n;
      asm volatile("" : "+v"(p));   // keep the address in vector registers: global_load, counted by vmcnt
      const WorkSeg sg = *p;
      beg = sg.begin;
      end = sg.end;
    }
  };
  // ... and back to scalar registers once the load has landed (the values are workgroup-uniform)
  auto to_scalar = [&](long long vbeg, long long vend, long long& beg, int& len) {
    auto rfl = [](long long v) -> long long {
      const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)v);
      const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
      return (long long)(((unsigned long long)hi << 32) | lo);
    };
    beg = rfl(vbeg);
    len = (int)(rfl(vend) - beg);
  };
  long long sb[4];
  int sl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long vb, ve;
    seg_load(i, vb, ve);
    to_scalar(vb, ve, sb[i], sl[i]);
  }
  // row ids of a batch whose descriptor is (beg, len)
  auto load_rid = [&](long long beg, int len) -> int { return tid < min(PLAN_CAP, len) ? a.idx[beg + tid] : -1; };
  // Software pipeline over the batches n = 0, 1, ... (a batch = the worker's list of one step):
  //   iteration n:  dot(n) | build part 1 (n+1)            -- the non-zeros of batch n were requested an iteration ago
  //                 barrier
  //                 gate(n) | build part 2 (n+1)            -- into the OTHER table set
  //                 barrier
  //                 scatter(n); leftovers(n); request the non-zeros of batch n+1 and the row records of batch n+2
  //                 barrier
  //                 sweep(n) (the update); publish the change of w . ds
  //                 barrier
  // Four barriers per batch; every global round trip of batch n+1 / n+2 runs under batch n's arithmetic.  Only the
  // weight gather (LDS for the hot ranks) and what follows it wait for the update -- the reference's synchronous
  // semantics are untouched.
  auto rows_of = [&](int len, int rid) -> BtRow {
    return bt_rows_issue<PLAN_CAP>(a.m, len, 0, [&](int) { return (long long)rid; }, &a.sc->err);
  };
  BtItems<PLAN_R> items;
  int2 bd = make_int2(0, 0);
  BtRow row_next;          // row records of batch n+1 (requested during iteration n-1)
  int rid_next2;           // row ids of batch n+2
  {
    const int rid0 = load_rid(sb[0], sl[0]);
    const int rid1 = load_rid(sb[1], sl[1]);
    rid_next2 = load_rid(sb[2], sl[2]);
    const BtRow row0 = rows_of(sl[0], rid0);
    row_next = rows_of(sl[1], rid1);
    if (n_batches > 0) {
      bd = bt_build<PLAN_THREADS, PLAN_CAP>(L, sl[0], 0, row0);
      if (bd.x > 0) bt_items_issue<PLAN_THREADS, PLAN_R>(a.m, L, bd.y, items);
    }
  }
  for (long long n = 0; n < n_batches; ++n) {
    const BtLds& Lc = (n & 1) ? L2 : L;     // tables of batch n
    const BtLds& Ln = (n & 1) ? L : L2;     // tables of batch n+1
    const float s = (float)(2.0 * (double)a.lambda * dot);   // thread-uniform: every thread carries the same dot
    const bool add_s = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
    float ddot = 0.0f;
    unsigned int n_act = 0;
    unsigned long long tl = a.tprof ? __builtin_readcyclecounter() : 0ull;
    const int B = sl[0], Bn = sl[1];
    int bits = 0;
    while ((1 << bits) < B) ++bits;
    const int shift = 30 - bits;   // at most one contribution per row and column: sums stay below 2^30
    const float qscale = ldexpf(1.0f, shift - a.vexp), inv_qscale = ldexpf(1.0f, a.vexp - shift);
    // ---- gradient on the weights of the previous step, interleaved with the tables of batch n+1 ----
    if (bd.x > 0) bt_items_dot<PLAN_THREADS, PLAN_R>(Lc, items, wload);
    const BtScan scn = bt_build_p1<PLAN_THREADS, PLAN_CAP>(Ln, Bn, 0, row_next);
    stamp(0, tl);
    __syncthreads();
    stamp(1, tl);
    if (bd.x > 0) n_act += bt_gate(Lc, bd.x);
    bt_build_p2<PLAN_THREADS, PLAN_CAP>(Ln, Bn, 0, row_next, scn);
    stamp(2, tl);
    __syncthreads();
    stamp(3, tl);
    if (bd.x > 0) bt_scatter<PLAN_R, 1>(Lc, a.gcold, items, qscale);
    const int2 bd_n = bt_build_p3<PLAN_THREADS>(Ln);
    if (bd.x < B && tid == 0) atomicOr(&a.sc->err, 4);   // the host's fit check and the device disagree: the step is invalid
    stamp(4, tl);
    __syncthreads();   // every contribution of this batch is in the accumulators (and the strip's atomics are performed)
    stamp(5, tl);
    // ---- nothing of batch n+1 / n+2 depends on w until the weight gather: request it all now, UNDER the sweep.
    // (Round 3 measured 9,000 cycles at the barrier above with these requests in front of it and concluded that
    // __syncthreads() waits for vector memory operations.  It does not with this compiler: the ISA is `s_waitcnt
    // lgkmcnt(0); s_barrier` -- a workgroup-scope fence needs no vmcnt wait when a workgroup lives on one CU; what
    // must be PERFORMED device-wide before a flag goes out waits in asm, e.g. dsgd_hogwild_kernel in front of its
    // commit.)  What consumes last iteration's loads (the row ids) goes first, the fresh requests last.
    row_next = rows_of(sl[2], rid_next2);
    rid_next2 = load_rid(sb[3], sl[3]);
    long long vb4, ve4;
    seg_load(n + 4, vb4, ve4);   // used when the window shifts at the end of the iteration
    if (Bn > 0 && bd_n.x > 0) bt_items_issue<PLAN_THREADS, PLAN_R>(a.m, Ln, bd_n.y, items);
    stamp(6, tl);
    // ---- sweep: the regularised sum on its support and the update; the hot ranks never leave LDS ----
    for (int j = tid; j < hl; j += PLAN_THREADS) {
      const int q = L.acc[j];
      if (q == 0) continue;
      L.acc[j] = 0;
      float g = filt((float)q * inv_qscale);            // Vec.sum of the batch (ref: core/Slave.scala:153)
      if (g == 0.0f) continue;
      if (add_s) g = filt(g + s);                       // ref: core/ml/SparseSVM.scala:31, math/Vec.scala:65-75
      float wn;
      ddot += step_w(g, wl[j], dsl[j], wn);
      wl[j] = wn;
      a.w[j] = wn;                                      // written through; never read back for a hot rank
    }
    for (int wd = tid; wd < n_cw; wd += PLAN_THREADS) {
      unsigned int cb = L.cbits[wd];
      if (!cb) continue;
      L.cbits[wd] = 0u;
      while (cb) {
        // up to four touched ranks of the word per round: their loads are in flight together (a batch of 100 rows
        // touches ~900 cold ranks over ~1,100 words: one round for nearly every thread)
        int jc[4];
        float old[4], dsj[4], gs[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          jc[e] = cb ? wd * 32 + __builtin_ctz(cb) : -1;
          cb &= cb - 1u;   // (0 stays 0)
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          old[e] = dsj[e] = gs[e] = 0.0f;
          if (jc[e] >= 0) {
            old[e] = a.w[hl + jc[e]];
            dsj[e] = a.ds[hl + jc[e]];
            // (written with L2 atomics of this workgroup: read it there, not through L1)
            gs[e] = __hip_atomic_exchange(&a.gcold[jc[e]], 0.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (jc[e] < 0) continue;
          const int j = hl + jc[e];
          float g = filt(gs[e]);
          if (g == 0.0f) continue;
          if (add_s) g = filt(g + s);
          float wn;
          ddot += step_w(g, old[e], dsj[e], wn);
          a.w[j] = wn;
        }
      }
    }
    // (`red` is free: the previous collect is behind three barriers)
    block_sum_f64_publish((double)ddot, red);
    stamp(7, tl);
    __syncthreads();   // the weights of the next gather are written; the wave partials are visible
    dot += block_sum_f64_collect(red);   // every thread adds the same total: `dot` stays thread-uniform
    n_act_total += n_act;
    bd = bd_n;
    if (Bn == 0) bd = make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      sb[i] = sb[i + 1];
      sl[i] = sl[i + 1];
    }
    to_scalar(vb4, ve4, sb[3], sl[3]);
    stamp(8, tl);
  }
  n_act_total = (unsigned long long)wave_sum_u32((unsigned int)n_act_total);
  // one atomic for the workgroup (`red` is free: the last collect is behind a barrier); its return value is what the
  // host-mapped mailbox of a per-request step gets -- no copy back behind the launch
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = (double)n_act_total;
  __threadfence();   // (this workgroup's error flags, if any, are out before thread 0 reads them)
  __syncthreads();
  if (tid == 0) {
    unsigned long long tot = 0;
    for (int i = 0; i < PLAN_THREADS / 64; ++i) tot += (unsigned long long)red[i];
    const unsigned long long old = atomicAdd(&a.sc->n_active, tot);
    if (a.mail) {
      __hip_atomic_store(&a.mail[0], old + tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const int err = __hip_atomic_load(&a.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.mail[1], (unsigned long long)(unsigned int)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    a.sc->s_reg = (float)(2.0 * (double)a.lambda * dot);
    // the request's sequence number LAST, with release order: the host may take the two words above once it sees it
    if (a.mail) __hip_atomic_store(&a.mail[2], a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.tprof) {
      for (int i = 0; i < 9; ++i) a.tprof[i] += tp[i];
      a.tprof[15] += (unsigned long long)(a.step_end - a.step_begin);
    }
  }
}
