// The whole gradient of a row range in ONE launch ("row chunks"): the shape for shards of 10^4 .. 2 * 10^6 rows -- RCV1's
// own size (804,414 rows; ref: src/test/scala/epfl/distributed/utils/DatasetTests.scala:18) and what one GPU of eight
// holds of it (core/ml/SplitStrategy.scala:13-14) -- where the three launches of the split streams (cold x.w, hot tiles,
// cold gradient; dsgd_kernels.hpp) spend a third of the step on launch boundaries, on the LDS tiles every workgroup of
// every launch sets up, and on each launch waiting for its slowest workgroup.
//
// What makes one launch possible WITHOUT a grid-wide wait: the only things that cross the three passes are per ROW --
// the cold part of x.w (dcold) into the gate, the gate's coefficient (coef8) into the cold gradient.  So workgroup b owns
// a CHUNK of consecutive rows, with wave tiles of both streams cut at the chunk's boundaries (the host lays these tiles
// out per (ranges, grid) configuration: dsgd_hip.hip, fstep_layout), and walks its chunk three times:
//
//   phase A   cold weights -> LDS; the chunk's cold tiles: dcold[row]                              (cd_tile)
//   phase B   hot weights -> LDS, hot gradient words cleared; the chunk's hot tiles: x.w, gate, coef8[row], hot
//             gradient; the chunk's long rows; the workgroup's exact hot partial sums out          (w_tile, w_long_row)
//   phase C   cold gradient words cleared; the chunk's cold tiles again (they were read a phase ago: the die cache
//             still holds them), coefficients by row; the cold partial sums out                    (cg_tile)
//
// dcold and coef8 go through global memory (the LDS is full in every phase) but never leave the CU's side of the
// machine: the workgroup that wrote a row's value is the one that reads it, behind a workgroup barrier (the vector
// cache is write-through and serves the CU's own stores back; the barrier's fence drains them).  Workgroups are in
// different phases at any one time -- one's LDS set-up runs under the others' streams -- and the launch ends when the
// slowest CHUNK is done, not three times when the slowest of three passes is.  The chunks are balanced by stream bytes.
// The exact reduce + regulariser + update (dsgd_fix_reduce_apply_kernel) follows as before: two launches per step
// instead of four, bit-identical sums (integers) whatever the chunking.
//
// The tile bodies are the streaming kernels' own (dsgd_kernels.hpp): same arithmetic, same order inside a row, same
// fixed-point grid -- a step through this kernel and one through the three launches differ only in which workgroup's
// partial a contribution lands in, and integer partial sums do not care.
#pragma once

#include "dsgd_kernels.hpp"

struct FChunk {          // 32 bytes, read with scalar loads
  int row_begin, row_end;      // the chunk's rows
  int tile_begin, tile_end;    // its hot tiles (of the chunked tile table)
  int ctile_begin, ctile_end;  // its cold tiles
  int long_begin, long_end;    // its range of the long-row list
};

// m: the hot stream (col = 16-bit ranks); mfull: the whole ranked CSR (long rows).  16-bit cold ids, every cold column in
// the LDS tile (the host takes the three-launch path otherwise).
__global__ void __launch_bounds__(1024) dsgd_fstep_kernel(CsrView m, CsrView mfull, const WTile* __restrict__ tiles,
                                                         const unsigned short* __restrict__ meta,
                                                         const WTile* __restrict__ ctiles,
                                                         const unsigned short* __restrict__ cmeta,
                                                         const void* __restrict__ ccol, const float* __restrict__ cval,
                                                         const FChunk* __restrict__ chunks, const float* __restrict__ w,
                                                         long long* __restrict__ g64_base, long long g_stride,
                                                         DevScalars* __restrict__ sc, int hw, int nc_lds, float fix_scale,
                                                         float cold_scale, signed char* coef8, float* dcold,
                                                         const int* __restrict__ long_rows, int* __restrict__ part,
                                                         int part_stride, int* __restrict__ partc, int partc_stride,
                                                         unsigned long long* __restrict__ tprof,
                                                         unsigned long long* __restrict__ wg_time) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((unsigned int)(unsigned long long)(__attribute__((address_space(3))) float*)lds != 0u) {
    if (tid == 0) atomicOr(&sc->err, 2);   // (ids become LDS addresses without a base: all LDS of this kernel is dynamic)
    return;
  }
  const FChunk ch = chunks[blockIdx.y * gridDim.x + blockIdx.x];
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  // how long this workgroup takes (device-wide 100 MHz clock), summed over launches: the host cuts the chunks again by it
  // (dsgd_hip.hip: fstep_rebalance -- a CU's speed differs by a few per cent by where it sits, stably over a run)
  const unsigned long long wg_t0 = (wg_time != nullptr && tid == 0) ? wall_clock64() : 0ull;
  constexpr int stride = 16;   // the waves of THIS workgroup share the chunk's tiles
  // tuning runs (DSGD_PLAN_PROF=1): shader-clock cycles of thread 0 by phase, summed over the workgroups
  //   [0] A set-up  [1] A tiles  [2] B set-up  [3] B tiles + long rows  [4] B partial out  [5] C set-up  [6] C tiles
  //   [7] C partial out  [8] sum of the workgroups' totals  [9] the slowest workgroup's total  [15] workgroups
  const bool prof = tprof != nullptr && tid == 0;
  unsigned long long t_prev = prof ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long t_first = t_prev;
  const unsigned long long rt_first = prof ? wall_clock64() : 0ull;   // (100 MHz, one clock for the whole device: start offsets)
#define DSGD_FPROF(I)                                              \
  if (prof) {                                                      \
    const unsigned long long t_now = __builtin_readcyclecounter(); \
    atomicAdd(&tprof[I], t_now - t_prev);                          \
    t_prev = t_now;                                                \
  }

  // ---- phase A: cold part of x.w of the chunk's rows ------------------------------------------------------------
  {
    float* strip = lds + ((nc_lds + 3) & ~3) + wave * CT_STRIP;
    CTabs tt;
    tt.tiles = ctiles;
    tt.meta = cmeta;
    tt.col = ccol;
    tt.val = cval;
    tt.coef8 = coef8;
    tt.row_begin = ch.row_begin;
    tt.row_end = ch.row_end;
    tt.t_lo = ch.ctile_begin;
    tt.t_hi = ch.ctile_end;
    int tile = tt.t_lo + wave;
    const bool any = tile < tt.t_hi;   // (wave-uniform)
    // The first three tiles' streams are requested BEFORE the weight tile is set up and stay in flight ACROSS the set-up
    // (round 6: every phase's first requests used to go out behind its set-up -- one exposed round trip through this
    // kernel's own queues, 2-3 us, per phase): the copy's requests first, the tiles' behind them (vmcnt retires in
    // order: the wait in front of the LDS writes covers the copy alone), a barrier that orders LDS only.
    CRegs<true> A, B, C, D;
    // (the four tile records in ONE round trip: fetched one after the other, each in front of the requests it describes,
    //  they were three dependent scalar round trips in a row at the head of every phase)
    const WTile wt0 = ctiles[c_map(tt, tile)], wt1 = ctiles[c_map(tt, tile + stride)], wt2 = ctiles[c_map(tt, tile + 2 * stride)];
    WTile wt = ctiles[c_map(tt, tile + 3 * stride)];
    WgCopy8 cp;
    wg_copy_in_issue(w + hw, nc_lds, tid, 1024, is_aligned16(w + hw), cp);
    __builtin_amdgcn_sched_barrier(0);
    if (any) {
      c_issue<true, false>(tt, tile, lane, wt0, A);
      __builtin_amdgcn_sched_barrier(0);
      c_issue<true, false>(tt, tile + stride, lane, wt1, B);
      __builtin_amdgcn_sched_barrier(0);
      c_issue<true, false>(tt, tile + 2 * stride, lane, wt2, C);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    wg_copy_in_store(lds, w + hw, nc_lds, tid, 1024, is_aligned16(w + hw), cp);
    lds_barrier();
    DSGD_FPROF(0)
    if (any) {
#define DSGD_FA(CUR, FAR)                                            \
  {                                                                  \
    const WTile wt_now = wt;                                         \
    wt = ctiles[c_map(tt, tile + 4 * stride)];                       \
    c_issue<true, false>(tt, tile + 3 * stride, lane, wt_now, FAR);  \
    cd_tile<true, false>(CUR, strip, dcold, w + hw, nc_lds);         \
  }
      for (;;) {
        DSGD_FA(A, D); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FA(B, A); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FA(C, B); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FA(D, C); tile += stride; if (tile >= tt.t_hi) break;
      }
#undef DSGD_FA
    }
    __syncthreads();   // every wave's dcold stores are out (the barrier's fence drains them) and the LDS tile is free
    DSGD_FPROF(1)
  }

  // ---- phase B: the hot tiles -- x.w, gate, hot gradient ---------------------------------------------------------
  {
    WTables tt;
    tt.tiles = tiles;
    tt.meta = meta;
    WCtx x;
    x.coef8 = coef8;
    float* wl = lds;
    float* strips = lds + ((hw + 4) & ~3);
    x.coefw = strips + wave * WS_COEF_STRIDE;
    x.gl = reinterpret_cast<int*>(strips + 16 * WS_COEF_STRIDE);
    x.wl = wl;
    x.g64 = g64_base + (long long)blockIdx.y * g_stride;
    x.sc = sc;
    x.dcold = dcold;
    x.row_begin = ch.row_begin;
    x.row_end = ch.row_end;
    x.hw = hw;
    x.hg = hw;
    x.fix_scale = fix_scale;
    x.cold_scale = cold_scale;
    // The chunk's hot tiles go to the waves AS THEY COME FREE (round 6): a wave's first four tiles are fixed (tile_begin +
    // wave + 16 i: their requests go out before the set-up below, registers only), every further one is drawn from a
    // counter in LDS one iteration before its record is fetched.  A strided walk made the workgroup wait for its slowest
    // wave -- the one that also had one of the chunk's LONG ROWS to do behind its tiles (rows that fit no tile; 0.75 us of
    // the workgroup's time each, up to ten per chunk: the workgroups with the most of them were the launch's last,
    // profiles/r06_fstep_wg_times.txt).  Now a wave with a long row does it FIRST, from registers (w_long_row_regs: four
    // round trips in a row -- row id, bounds, stream, cold weights -- instead of two dozen; ~10 us of the wave's time all
    // the same: a round trip through this kernel's own queues is 2-3 us), requests its first tiles behind it, and draws
    // fewer tiles than the others; the chunks are cut with that cost in their weight (dsgd_hip.hip: fstep_layout).
    // Which wave adds a tile's rows to the workgroup's integer accumulators does not change any sum: the same bits.
    int* const tile_ctr = x.gl + hw + 64;   // (the four words LDS had left: dsgd_hip.hip launch_fstep)
    unsigned int n_all = 0, n_neg = 0, n_pos = 0;
    const int t_end = ch.tile_end;
    int qa = ch.tile_begin + wave, qb = qa + stride, qc = qb + stride, qd = qc + stride;
    const bool any = qa < t_end;                                 // (wave-uniform)
    const bool has_long = ch.long_begin + wave < ch.long_end;    // (wave-uniform)
    WRegs A, B, C, D;
    const WTile wt0 = w_fetch(tt, qa, t_end), wt1 = w_fetch(tt, qb, t_end), wt2 = w_fetch(tt, qc, t_end);   // (one round trip)
    WTile wt = w_fetch(tt, qd, t_end);
#define DSGD_FB_PROLOGUE                                 \
  {                                                      \
    w_issue_cols(m, qa, t_end, lane, wt0, A);            \
    w_issue_vals(m, tt, x.dcold, lane, A);               \
    __builtin_amdgcn_sched_barrier(0);                   \
    w_issue_cols(m, qb, t_end, lane, wt1, B);            \
    w_issue_vals(m, tt, x.dcold, lane, B);               \
    __builtin_amdgcn_sched_barrier(0);                   \
    w_issue_cols(m, qc, t_end, lane, wt2, C);            \
    w_issue_vals(m, tt, x.dcold, lane, C);               \
    __builtin_amdgcn_sched_barrier(0);                   \
  }
    WgCopy8 cp;
    wg_copy_in_issue(w, hw, tid, 1024, is_aligned16(wl) && is_aligned16(w), cp);
    __builtin_amdgcn_sched_barrier(0);
    if (!has_long) {
      if (any) DSGD_FB_PROLOGUE
    }
    __builtin_amdgcn_sched_barrier(0);
    if (is_aligned16(x.gl)) wg_zero(x.gl, hw + 64, tid, 1024);
    else
      for (int j = tid; j < hw + 64; j += 1024) x.gl[j] = 0;
    wg_copy_in_store(wl, w, hw, tid, 1024, is_aligned16(wl) && is_aligned16(w), cp);
    if (tid == 0) {
      wl[hw] = 0.0f;
      tile_ctr[0] = ch.tile_begin + 4 * stride;
    }
    lds_barrier();   // (LDS only: the first tiles' requests stay in flight)
    DSGD_FPROF(2)

    if (has_long) {   // (the registers of the first tiles' requests are not live yet on this path: the row sits in them)
      const unsigned long long t_l0 = prof ? __builtin_readcyclecounter() : 0ull;
      for (int t = ch.long_begin + wave; t < ch.long_end; t += stride)
        w_long_row_regs(mfull, w, x, (long long)long_rows[t], n_all, n_neg, n_pos);
      if (prof) {   // [10] cycles wave 0 spent on its long rows, [11] how many workgroups had one
        atomicAdd(&tprof[10], __builtin_readcyclecounter() - t_l0);
        atomicAdd(&tprof[11], 1ull);
      }
      if (any) DSGD_FB_PROLOGUE
    }
#undef DSGD_FB_PROLOGUE
    if (any) {
      // lane 0 draws; the value is picked up (readfirstlane) an iteration later, when it names the record to fetch
      auto draw = [&]() -> int {
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
      };
      int qe = __builtin_amdgcn_readfirstlane(draw());
      int pend = draw();
#define DSGD_FB(CUR, FAR)                                                                \
  {                                                                                      \
    w_tile_q<true>(m, tt, x, qd, qe, t_end, CUR, FAR, wt, n_all, n_neg, n_pos);           \
    qa = qb; qb = qc; qc = qd; qd = qe;                                                  \
    qe = __builtin_amdgcn_readfirstlane(pend);                                           \
    pend = draw();                                                                       \
  }
      for (;;) {
        DSGD_FB(A, D); if (qa >= t_end) break;
        DSGD_FB(B, A); if (qa >= t_end) break;
        DSGD_FB(C, B); if (qa >= t_end) break;
        DSGD_FB(D, C); if (qa >= t_end) break;
      }
#undef DSGD_FB
    }

    n_all = wave_sum_u32(n_all);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) reinterpret_cast<unsigned int*>(x.coefw)[0] = n_all;
    __syncthreads();   // (also: every wave's coef8 stores are out)
    const unsigned long long t_b0 = t_prev;
    DSGD_FPROF(3)
    if (prof && wg < 1024) tprof[16 + 4 * wg + 2] = t_prev - t_b0;
  }
  // ---- phase C: the cold gradient columns of the chunk's rows ----------------------------------------------------
  {
    // (its first requests -- the cold stream again, the rows' coefficients: registers only -- go out before phase B's
    //  partial sums leave and the tile is cleared)
    CTabs tt;
    tt.tiles = ctiles;
    tt.meta = cmeta;
    tt.col = ccol;
    tt.val = cval;
    tt.coef8 = coef8;
    tt.row_begin = ch.row_begin;
    tt.row_end = ch.row_end;
    tt.t_lo = ch.ctile_begin;
    tt.t_hi = ch.ctile_end;
    int tile = tt.t_lo + wave;
    const bool any = tile < tt.t_hi;   // (wave-uniform)
    CRegs<true> A, B, C, D;
    const WTile wt0 = ctiles[c_map(tt, tile)], wt1 = ctiles[c_map(tt, tile + stride)], wt2 = ctiles[c_map(tt, tile + 2 * stride)];
    WTile wt = ctiles[c_map(tt, tile + 3 * stride)];
    if (any) {
      c_issue<true, true>(tt, tile, lane, wt0, A);
      __builtin_amdgcn_sched_barrier(0);
      c_issue<true, true>(tt, tile + stride, lane, wt1, B);
      __builtin_amdgcn_sched_barrier(0);
      c_issue<true, true>(tt, tile + 2 * stride, lane, wt2, C);
      __builtin_amdgcn_sched_barrier(0);
    }
    {   // phase B's finish: the active-row count, the workgroup's exact hot partial sums out
      float* strips = lds + ((hw + 4) & ~3);
      int* gl = reinterpret_cast<int*>(strips + 16 * WS_COEF_STRIDE);
      if (tid == 0) {
        unsigned int t_all = 0;
        for (int i = 0; i < 16; ++i) t_all += reinterpret_cast<const unsigned int*>(strips + i * WS_COEF_STRIDE)[0];
        if (t_all) atomicAdd(&sc->n_active, (unsigned long long)t_all);   // one atomic per workgroup
      }
      int* mine = part + (long long)wg * part_stride;
      wg_copy_out(mine, gl, hw, tid, 1024, is_aligned16(mine) && is_aligned16(gl));
      lds_barrier();   // the gradient words have been READ (LDS only: their stores and the tiles' requests stay in flight)
      DSGD_FPROF(4)
    }
    const int n_tile = nc_lds + 64;
    float* strip = lds + ((n_tile + 3) & ~3) + wave * CT_STRIP;
    wg_zero(reinterpret_cast<int*>(lds), n_tile, tid, 1024);
    lds_barrier();
    DSGD_FPROF(5)
    long long* g64cold = g64_base + (long long)blockIdx.y * g_stride + hw;
    if (any) {
#define DSGD_FC(CUR, FAR)                                                                              \
  {                                                                                                    \
    const WTile wt_now = wt;                                                                           \
    wt = ctiles[c_map(tt, tile + 4 * stride)];                                                         \
    c_issue<true, true>(tt, tile + 3 * stride, lane, wt_now, FAR);                                     \
    cg_tile<true, false>(CUR, strip, reinterpret_cast<int*>(lds), g64cold, sc, cold_scale, nc_lds);    \
  }
      for (;;) {
        DSGD_FC(A, D); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FC(B, A); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FC(C, B); tile += stride; if (tile >= tt.t_hi) break;
        DSGD_FC(D, C); tile += stride; if (tile >= tt.t_hi) break;
      }
#undef DSGD_FC
    }
    __syncthreads();
    DSGD_FPROF(6)
    if (wg_time != nullptr && tid == 0) wg_time[wg] += wall_clock64() - wg_t0;   // (this workgroup is the word's only writer)
    int* minec = partc + (long long)wg * partc_stride;
    wg_copy_out(minec, reinterpret_cast<int*>(lds), nc_lds, tid, 1024, is_aligned16(minec));
    if (tprof) {
      __syncthreads();
      DSGD_FPROF(7)
      if (prof) {
        atomicAdd(&tprof[8], t_prev - t_first);
        atomicMax(&tprof[9], t_prev - t_first);
        atomicAdd(&tprof[15], 1ull);
        if (wg < 1024) {
          unsigned int xcc;
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
          tprof[16 + 4 * wg] = rt_first;
          tprof[16 + 4 * wg + 1] = wall_clock64();
          // the XCD and what the chunk held: hot tiles << 8, cold tiles << 24, rows << 40, long rows << 56
          tprof[16 + 4 * wg + 3] = (unsigned long long)(xcc & 0xffu) | ((unsigned long long)(ch.tile_end - ch.tile_begin) << 8) |
                                   ((unsigned long long)(ch.ctile_end - ch.ctile_begin) << 24) |
                                   ((unsigned long long)(ch.row_end - ch.row_begin) << 40) |
                                   ((unsigned long long)(ch.long_end - ch.long_begin) << 56);
        }
      }
    }
  }
#undef DSGD_FPROF
}

// The data-dependent bound of the hot fixed-point scale (dsgd_wseg_bound_kernel's job) for the chunked assignment:
// the largest column sum of ceil(|x| * scale0) over the rows ONE chunk holds.
__global__ void __launch_bounds__(1024) dsgd_fstep_bound_kernel(const long long* __restrict__ hrow_ptr,
                                                               const unsigned short* __restrict__ hcol16,
                                                               const float* __restrict__ hval,
                                                               const FChunk* __restrict__ chunks, CsrView mfull,
                                                               const int* __restrict__ long_rows, int hg, float scale0,
                                                               unsigned int* __restrict__ out_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned int fbl[];   // hg column sums + 16 words
  unsigned int* red = fbl + hg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < hg; j += 1024) fbl[j] = 0u;
  __syncthreads();
  const FChunk ch = chunks[blockIdx.y * gridDim.x + blockIdx.x];
  // (rows of the long list own no slots in the hot stream: [hrow_ptr[row_begin], hrow_ptr[row_end]) are the tiled rows')
  const long long e0 = hrow_ptr[ch.row_begin], e1 = hrow_ptr[ch.row_end];
  for (long long e = e0 + tid; e < e1; e += 1024) {
    const unsigned int q = (unsigned int)ceilf(fabsf(hval[e]) * scale0);
    if (q) atomicAdd(&fbl[hcol16[e]], q);
  }
  for (int t = ch.long_begin + wave; t < ch.long_end; t += 16) {
    const long long row = long_rows[t];
    for (long long p = mfull.row_ptr[row] + lane; p < mfull.row_ptr[row + 1]; p += 64) {
      const int c = mfull.col[p];
      const unsigned int q = (unsigned int)ceilf(fabsf(mfull.val[p]) * scale0);
      if (c < hg && q) atomicAdd(&fbl[c], q);
    }
  }
  __syncthreads();
  unsigned int mx = 0u;
  for (int j = tid; j < hg; j += 1024) mx = max(mx, fbl[j]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, off, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    for (int i = 1; i < 16; ++i) mx = max(mx, red[i]);
    atomicMax(out_max, mx);
  }
}
