// Device code of libdsgd_hip, part 4 (gfx950 only): K1c -- the reference's own batch sizes (3 workers x batch 100,
// application.conf:15,27; 4 x 200, kube/config-sync.yaml) as a FEATURE-PARALLEL persistent kernel ("column slices").
// Included by dsgd_hip.hip after dsgd_batch.hpp.
//
// ref: core/Master.scala:179-199 (the batch closure), core/Slave.scala:142-157 (a worker's regularised sum).
//
// A 3 x 100 step is 22 K non-zeros -- 180 KB -- scattered over gigabytes: nothing but latency.  Rounds 1-3 spread the
// ROWS of such a step over workgroups (one per worker), which forces every workgroup to own all 18 K hot weights and
// gradient words (147 KB of LDS set up, 73 KB of partials written and read back, a second launch for the exact reduce,
// regulariser and update): 10 + 5 us of fixed cost for 1 us of arithmetic.  Here the COLUMNS are spread instead:
//   * G workgroups (8, or 16 beyond four workers), one per CU; workgroup b owns the ranks r = b (mod G) -- frequency
//     ranks are Zipfian, so a stride-G interleave balances the slices -- and keeps ITS weights, its dimSparsity values
//     and one fixed-point accumulator per hosted worker in LDS for the whole launch (5,905 columns x (2 + K) words).  The
//     weights never leave the CU between steps: no gather, no partials, no reduce kernel, no update kernel;
//   * the host lays a resident plan out per (step, slice): every row's entries inside the slice, in chunks ("slots") of
//     <= 16 with slice-local 16-bit column indices, one slot per lane, all of a step's slots in two 16-byte requests per
//     array and lane.  The slots of step n + 1 are requested while the workgroup waits in step n's exchange;
//   * per step ONE exchange between the workgroups: every slice publishes its partial x.w of the step's rows (300
//     floats) and its share of w . ds as 8-byte GRANULES {value, step tag} -- one write-through (sc1) store each, no
//     flag, no drain: a granule is valid when its tag is the step's (MI355X_MICROARCH.md: data-tagged granules need no
//     ordering; one hand-off ~1 us).  Every slice polls the G granules of each of its rows (sc1 loads), adds the
//     partials in slice order -- bitwise the same sum in every workgroup, so all take the same gate decisions
//     (core/ml/SparseSVM.scala:27-28) -- scatters ITS entries of the active rows into ITS accumulators (ds_add_u32,
//     exact integer sums), and finishes ITS columns: one rounding per worker's sum, the support-only regulariser
//     (SparseSVM.scala:31), the fold over the workers, the mean, the update (Master.scala:194-197) -- the arithmetic of
//     dsgd_fix_reduce_apply_kernel, column for column.  The first form of this exchange (sc1 payload, drain, arrival on
//     a device-scope counter, poll, sc1 loads of the payload: three dependent trips over the fabric) cost ~6.5 of a
//     step's 11 us (3 x 100: 17.1 us, slower than the row-parallel kernels' 14.9).
// The reference's synchronous semantics are untouched: every gradient of a step sees the weights of the step before.
// Two buffers alternate between steps: a slice can publish step t + 2 only after it has gathered step t + 1, which
// needs every peer's step t + 1 granules, which a peer stores only after it has read everything of step t.  A bounded
// poll raises an abort word that every workgroup honours (a launch can end with DevScalars::err = 8, never hang).
#pragma once

constexpr int CS_THREADS = 512;     // 8 waves, two per SIMD (<= 256 VGPRs): one or two slots per lane and register set, two sets
constexpr int CS_THREADS_NARROW = 256;   // (tuning runs: DSGD_CS_NT=256)
constexpr int CS_L = 16;            // entries per slot
constexpr int CS_MAX_G = 16;        // slices = workgroups
constexpr int CS_MAX_K = 8;         // hosted workers
constexpr int CS_MAX_SLOTS = 1024;   // per (step, slice); also the most rows of a step
constexpr int CS_XSTRIDE = CS_MAX_SLOTS + 64;           // granules per (parity, slice) of the exchange buffer: [row] partial x.w, [CS_MAX_SLOTS] share of w . ds
constexpr unsigned int CS_POLL_LIMIT = 1u << 18;        // polls of one granule set before the launch is given up
constexpr int CS_MAX_CLT = 8;                           // listed columns per lane: a step may touch 4,096 columns of a slice

struct CsHdr {                // per (slice, step)
  unsigned int counts;        // slots of the step inside the slice (low 16 bits) | rows of the step << 16
  int shift;                  // fixed-point shift of the step: 30 - ceil(log2(largest list)) (the same in every slice), low 16 bits
                              //   | listed columns of the step inside the slice << 16
};

struct CsArgs {
  const CsHdr* hdr;                 // [G][n_steps]
  const unsigned int* slot_meta;    // [G][n_steps][slot_stride]: row of the step (bits 0-15) | worker (bits 16-19)
  const unsigned short* row_first;  // [G][n_steps][row_stride]: first slot of row r (bits 0-10; entry n_rows = n_slots) | label > 0 (bit 15)
  const uint4* col;                 // [G][n_steps][2][slot_stride]: 16-byte PIECES of the slots' 16 slice-local columns (rank / G, 16 bits
  const float4* val;                //   each) and [G][n_steps][4][slot_stride] of their values: lane = slot, so every request of a wave is
                                    //   one contiguous KiB (slot-major records -- 32- and 64-byte strides between lanes -- took 7 us to land
                                    //   at 4 x 200); padding: column 0, value 0
  const unsigned short* clist;      // [G][n_steps][cl_stride]: the distinct columns of the step inside the slice, ascending; 0xffff: none
  float* w;                         // the weights SLICE-MAJOR: [G][Sp], slice b's local column i = rank b + G * i (read when the launch
                                    //   starts, written when it ends; dsgd_cs_slice_kernel / dsgd_cs_unslice_kernel convert)
  const float* ds;                  // dimSparsity, the same layout
  unsigned long long* xbuf;         // [2][G][CS_XSTRIDE] granules {value bits, step tag << 32}; zero when a launch starts
  unsigned int* sync;               // [1] abort word (zero when a launch starts)
  DevScalars* sc;
  unsigned long long* tprof;        // optional (tuning runs, DSGD_PLAN_PROF=1): cycles of thread 0 of slice 0 by phase, [15] = steps
  long long n_steps_plan, step_begin, step_end;
  int slot_stride, row_stride, cl_stride;
  unsigned int tag0;                // steps of the context's earlier launches
  float lr, lambda;
  int vexp, dp, G, K;
  // per-request steps (dsgd_cs_request_kernel): host-mapped {n_active, err, sequence number} written when slice 0 leaves
  unsigned long long* mail;
  unsigned long long mail_seq;
  // optional record of a plan's run (dsgd_plan_record): the gate decision of every row of every step (bit r of the step's
  // words = row r was active, core/ml/SparseSVM.scala:27-28) and the regulariser scalar the step used -- what an oracle
  // needs to REPLAY a trajectory with the engine's own decisions (oracle/sync_replay.py)
  unsigned int* gate_rec;           // [n_steps_plan][gate_words], zero before the run
  float* s_rec;                     // [n_steps_plan]
  int gate_words;
  int test_skip_publish;            // TEST BUILDS ONLY (DSGD_TEST_COLLECTIVE_SEAM): slice 1 dies at this step of the launch (1-based; 0: never)
};

__host__ __device__ constexpr int cs_lds_words(int dp, int G, int K) {
  return (2 + K) * ((((dp + G - 1) / G) + 4) & ~3) + 2 * CS_MAX_SLOTS + 32;   // (a slice is padded by 1 .. 4 columns)
}

// a per-request launch: the builder's scratch sits where the accumulators, partial dots and coefficients go afterwards
__host__ __device__ constexpr int cs_req_lds_words(int dp, int G, int K) {
  const int sp = (((dp + G - 1) / G) + 4) & ~3;
  const int rest = K * sp + 2 * CS_MAX_SLOTS;
  return 2 * sp + (rest > 7000 ? rest : 7000) + 32;   // 7000 >= CS_BUILD_WORDS (static_assert below)
}

template <int SPL, int CLT>
struct CsSet {              // the slots of one step, as loaded (nothing is computed on them before their step runs)
  uint4 c[SPL][2];          // 16 slice-local columns, 16 bits each
  float4 v[SPL][4];
  unsigned int meta[SPL];
  unsigned int rf[SPL];     // row_first[r] | row_first[r + 1] << 16 of row r = tid + CS_THREADS * i
  unsigned short cl[CLT];   // listed columns tid + CS_THREADS * i of the step
  uint2 h;                  // the step's header
};

template <int NT, int SPL, int CLT>
__device__ __forceinline__ void cs_issue(const CsArgs& a, int b, long long step, CsSet<SPL, CLT>& R) {
  // every request unconditional, indices clamped (a step beyond the launch's last one re-reads the last and is never used)
  const long long sc = step < a.step_end ? step : a.step_end - 1;
  const long long sidx = (long long)b * a.n_steps_plan + sc;
  const uint2* hp = reinterpret_cast<const uint2*>(a.hdr + sidx);
  asm volatile("" : "+v"(hp));   // a vector load (vmcnt): a scalar one would share lgkmcnt with the LDS traffic of the whole step
  R.h = *hp;
  const long long sbase = sidx * a.slot_stride, rbase = sidx * a.row_stride, cbase = sidx * a.cl_stride;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    int slot = (int)threadIdx.x + NT * i;
    slot = slot < a.slot_stride ? slot : a.slot_stride - 1;
    R.c[i][0] = a.col[2 * sbase + slot];
    R.c[i][1] = a.col[2 * sbase + a.slot_stride + slot];
    R.v[i][0] = a.val[4 * sbase + slot];
    R.v[i][1] = a.val[4 * sbase + a.slot_stride + slot];
    R.v[i][2] = a.val[4 * sbase + 2 * a.slot_stride + slot];
    R.v[i][3] = a.val[4 * sbase + 3 * a.slot_stride + slot];
    R.meta[i] = a.slot_meta[sbase + slot];
    int r = (int)threadIdx.x + NT * i;
    r = r < a.row_stride - 1 ? r : a.row_stride - 2;
    R.rf[i] = (unsigned int)a.row_first[rbase + r] | ((unsigned int)a.row_first[rbase + r + 1] << 16);
  }
#pragma unroll
  for (int i = 0; i < CLT; ++i) {
    const int e = (int)threadIdx.x + NT * i;
    const unsigned short cl = a.clist[cbase + (e < a.cl_stride ? e : a.cl_stride - 1)];
    R.cl[i] = e < a.cl_stride ? cl : (unsigned short)0xffffu;
  }
}

// granule `p` of slice g behind a buffer resource: an agent-scope (sc1) 8-byte load
typedef unsigned int cs_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long cs_granule(__amdgpu_buffer_rsrc_t rs, unsigned int p, int g) {
  const cs_u32x2 x = __builtin_bit_cast(cs_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(p * 8u), g * CS_XSTRIDE * 8, 16 /* sc1 */));
  return ((unsigned long long)x.y << 32) | x.x;
}

// the G granules of each of N exchange slots at[] (a row, or CS_MAX_SLOTS for the shares of w . ds; at < 0: none), polled
// TOGETHER until all carry `tag` (one slot after the other was one round trip per slot: 2.6 us of a 4 x 200 step); the
// values added in slice order.  false = given up (the abort word is raised for everybody).
template <int N>
__device__ __forceinline__ bool cs_gather(const unsigned long long* xall, int G, const int (&at)[N], unsigned int tag,
                                          unsigned int* abort_word, float (&sum)[N]) {
  unsigned int v[N][CS_MAX_G];   // (the values; the tags are checked as they arrive)
  // ONE scalar base and a 32-bit lane offset per slot (global_load ... v_off, s[base]): a 64-bit lane address per granule
  // was 64 registers of addresses alone
  // buffer loads: ONE resource, the slice's stride in a scalar, a 32-bit lane offset per slot -- a 64-bit lane address per
  // granule was 64 registers of addresses alone
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(xall), 0, G * CS_XSTRIDE * 8, 0x00020000);
  bool any = false;
#pragma unroll
  for (int n = 0; n < N; ++n) any = any || at[n] >= 0;
  if (!any) return true;
  for (unsigned int spin = 0;; ++spin) {
    asm volatile("" ::: "memory");   // (every poll reads memory again)
    bool all = true;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const unsigned int p = at[n] < 0 ? 0u : (unsigned int)at[n];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const unsigned long long x = cs_granule(rs, p, g);
        v[n][g] = (unsigned int)x;
        all = all && ((unsigned int)(x >> 32) == tag || at[n] < 0);
      }
    }
    if (G > 8) {   // (workgroup-uniform: G is 8 or 16)
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const unsigned int p = at[n] < 0 ? 0u : (unsigned int)at[n];
#pragma unroll
        for (int g = 8; g < CS_MAX_G; ++g) {
          const unsigned long long x = cs_granule(rs, p, g);
          v[n][g] = (unsigned int)x;
          all = all && ((unsigned int)(x >> 32) == tag || at[n] < 0);
        }
      }
    } else {
#pragma unroll
      for (int n = 0; n < N; ++n)
#pragma unroll
        for (int g = 8; g < CS_MAX_G; ++g) v[n][g] = 0u;
    }
    if (all) break;
    if ((spin & 63u) == 63u) {
      if (spin > CS_POLL_LIMIT || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) {
    float d = 0.0f;
#pragma unroll
    for (int g = 0; g < CS_MAX_G; ++g) d += __uint_as_float(v[n][g]);   // (the zeros of an 8-slice run change nothing)
    sum[n] = d;
  }
  return true;
}

// Workgroup barrier that orders LDS only.  __syncthreads() is a workgroup-scope FENCE: it waits for every outstanding
// memory operation of the wave (s_waitcnt vmcnt(0)) -- here that is the next step's slots, requested at the top of the
// step precisely so that they stay in flight under it (measured with __syncthreads(): the first barrier of every step
// waited out the whole HBM round trip, 1.9 us at 3 x 100 and 8 us at 4 x 200).  Everything the barriers of a step
// order lives in LDS; what crosses workgroups goes through the tagged granules, which need no ordering.
__device__ __forceinline__ void cs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the workgroup, the same bits on every thread (wave butterflies, then the wave sums pairwise in order)
template <int NT>
__device__ __forceinline__ float cs_block_sum(float v, float* red16) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  cs_barrier();
  if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
  cs_barrier();
  float t[NT / 64];
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t[i] = red16[i];
#pragma unroll
  for (int n = NT / 64; n > 1; n >>= 1)
#pragma unroll
    for (int i = 0; i < n / 2; ++i) t[i] = t[2 * i] + t[2 * i + 1];
  return t[0];
}

// the LDS carve and the launch-long state of a workgroup
struct CsState {
  float* w_l;
  float* ds_l;
  int* acc;       // [K][Sp], zero between steps
  float* ps;      // partial x.w per slot
  float* coef;    // per row of the step: +-2^shift / vmax2 (active) or 0
  float* red;     // [0..15] wave sums, [16] s of the step, [17] abort flag, [18..31] the tuning counters
  int b, Sb, Sp;
  float sp;       // this slice's share of w . ds of the current weights (the same bits on every thread)
  unsigned int n_act, n_rel;   // active rows counted (slice 0 only); steps of this launch behind us
  unsigned long long* tp;      // tuning runs (LDS, thread 0 of slice 0): cycles by phase (dot, publish, exchange, scatter, sweep, reduce), [6] last stamp
};

// this lane's part of the slice's share of w . ds: all columns, four per lane and round (the padding holds zeros)
template <int NT>
__device__ __forceinline__ float cs_wds_share(const CsState& z) {
  const float4* w4 = reinterpret_cast<const float4*>(z.w_l);
  const float4* d4 = reinterpret_cast<const float4*>(z.ds_l);
  float sp = 0.0f;
  for (int i4 = threadIdx.x; i4 < (z.Sp >> 2); i4 += NT) {
    const float4 wv = w4[i4], dv = d4[i4];
    sp += (filt(wv.x * dv.x) + filt(wv.y * dv.y)) + (filt(wv.z * dv.z) + filt(wv.w * dv.w));
  }
  return sp;
}

// The listed columns of a step, KK hosted workers: per worker ONE rounding of the exact sum, the support-only
// regulariser, the fold over the workers, the mean, the update -- dsgd_fix_reduce_apply_kernel's arithmetic
// (fra_update_and_scalars), here without a branch: a lane without a column works on the slice's padding column (always
// there: zero weight, zero accumulators), an untouched worker contributes the zero it would have been skipped for.
template <int NT, int KK, int CLT>
__device__ __forceinline__ void cs_sweep(CsState& z, const unsigned short (&cl)[CLT], int n_cols, int Sp, float inv_scale, float s, bool add,
                                         float lr) {
  constexpr int U = 2;   // columns per lane and round: every LDS request of a round before the first is used
#pragma unroll
  for (int i0 = 0; i0 < CLT; i0 += U) {
    if (i0 * NT >= n_cols) break;   // (workgroup-uniform: the list is dense from entry 0)
    int c[U], t[U][KK];
    float wo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      c[u] = cl[i0 + u] == 0xffffu ? Sp - 1 : (int)cl[i0 + u];
#pragma unroll
      for (int k = 0; k < KK; ++k) t[u][k] = z.acc[k * Sp + c[u]];
      wo[u] = z.w_l[c[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float gsum = 0.0f;
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        z.acc[k * Sp + c[u]] = 0;
        // one rounding of the worker's exact sum (int -> fp32 rounds it to 24 bits; the power of two is exact: the same
        // bits as rounding the fp64 product, as dsgd_fix_reduce_apply_kernel does)
        const float g0 = filt((float)t[u][k] * inv_scale);
        const float g1 = filt(g0 + s);                                 // ref: core/ml/SparseSVM.scala:31, math/Vec.scala:65-75
        gsum = filt(gsum + ((add && g0 != 0.0f) ? g1 : g0));           // Vec.sum over the workers
      }
      const float upd = filt(filt(gsum / (float)KK) * lr);             // Vec.mean, learningRate * grad (ref: core/Master.scala:194-197)
      z.w_l[c[u]] = gsum != 0.0f ? filt(wo[u] - upd) : wo[u];
    }
  }
}

// the 16 columns / values of slot i of a register set (compile-time indices only: the set stays in registers)
#define CS_COL(R, i, j) ((int)(((j) & 1) ? ((&(R).c[i][(j) >> 3].x)[((j) >> 1) & 3] >> 16) : ((&(R).c[i][(j) >> 3].x)[((j) >> 1) & 3] & 0xffffu)))
#define CS_VAL(R, i, j) ((&(R).v[i][(j) >> 2].x)[(j) & 3])

// One step.  `cur`: the step's slots (landed); `nxt` receives the next step's.  false = the launch was aborted.
template <int NT, int SPL, int CLT>
__device__ __forceinline__ bool cs_step(const CsArgs& a, CsState& z, CsSet<SPL, CLT>& cur, CsSet<SPL, CLT>& nxt, long long step) {
  int tid = threadIdx.x, G = __builtin_amdgcn_readfirstlane(a.G), K = __builtin_amdgcn_readfirstlane(a.K), b = z.b,
      Sp = __builtin_amdgcn_readfirstlane(z.Sp);
  // what a step derives from these is recomputed in every step: hoisted out of the step loop, the lanes' addresses and
  // scale factors were ~150 registers too many, and their reloads from scratch are vector-memory operations -- they
  // retire in order BEHIND the next step's slots, so every one of them waited the prefetch out
  asm volatile("" : "+v"(tid));
  asm volatile("" : "+s"(G), "+s"(K), "+s"(Sp));
  float* const ps = z.ps;
  float* const coef = z.coef;
  float* const red = z.red;
  const int n_slots = __builtin_amdgcn_readfirstlane((int)(cur.h.x & 0xffffu));
  const int n_rows = __builtin_amdgcn_readfirstlane((int)(cur.h.x >> 16));
  const int shift = __builtin_amdgcn_readfirstlane((int)(cur.h.y & 0xffffu));
  const int n_cols = __builtin_amdgcn_readfirstlane((int)(cur.h.y >> 16));
  const float qscale = ldexpf(1.0f, shift - a.vexp);
  const float inv_scale = ldexpf(1.0f, a.vexp - shift);
  const bool prof = a.tprof != nullptr && b == 0 && tid == 0;
  auto stamp = [&](int i) {
    if (prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      z.tp[i] += now - z.tp[6];
      z.tp[6] = now;
    }
  };
  // ---- 1: partial x.w of every slot from this slice's weights (ref: math/Vec.scala:58, math/Sparse.scala:46) ----
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    float p = 0.0f;
#pragma unroll
    for (int j = 0; j < CS_L; ++j) p += filt(CS_VAL(cur, i, j) * z.w_l[CS_COL(cur, i, j)]);
    const int slot = tid + NT * i;
    if (slot < n_slots) ps[slot] = p;
  }
  if (tid == 0) reinterpret_cast<int*>(red)[17] = 0;   // (this step's "given up" flag: raised in phase 3, read behind its barrier)
  cs_barrier();
  stamp(0);
  // ---- 2: this slice's partial of every row (its slots in order), published as granules {value, tag}: write-through,
  //         nothing waits for them ----
  const unsigned int tag = a.tag0 + z.n_rel + 1u;   // (tags run on across the launches of a context: nothing is cleared between them)
  unsigned long long* xb = a.xbuf + ((long long)(z.n_rel & 1u) * G + b) * CS_XSTRIDE;
#ifdef DSGD_TEST_COLLECTIVE_SEAM
  // (tests of the failure path: one slice DIES here -- publishes nothing more, writes nothing back; its peers must give
  //  the launch up at their poll limit, not hang)
  if (a.test_skip_publish > 0 && b == 1 && (long long)z.n_rel + 1 >= a.test_skip_publish) return false;
#endif
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int r = tid + NT * i;
    if (r < n_rows) {
      const int f0 = (int)(cur.rf[i] & 0x7ffu), f1 = (int)((cur.rf[i] >> 16) & 0x7ffu);
      float t = 0.0f;
      for (int f = f0; f < f1; ++f) t += ps[f];
      __hip_atomic_store(&xb[r], ((unsigned long long)tag << 32) | __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0)
    __hip_atomic_store(&xb[CS_MAX_SLOTS], ((unsigned long long)tag << 32) | __float_as_uint(z.sp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- the NEXT step's slots are requested HERE: they land while this workgroup waits for its peers' granules (vmcnt
  //      retires in order, so the polls below return no earlier than these -- the exchange is a round trip anyway).
  //      Requested at the top of the step they were waited for on the spot: beyond 256 registers the allocator parks
  //      freshly loaded values in accumulation registers, and the move needs the value. ----
  cs_issue<NT, SPL, CLT>(a, b, step + 1, nxt);
  stamp(1);
  // ---- 3: every slice's granules of this thread's rows: x.w in slice order, the gate, the row's coefficient ----
  const unsigned long long* xall = a.xbuf + (long long)(z.n_rel & 1u) * G * CS_XSTRIDE;
  // (the shares of w . ds ride as "row" n_rows -- s = 2 lambda (w . ds) of the weights this step's gradients see -- unless
  //  every lane position holds a row)
  int at[SPL];
  float dd[SPL];
  bool act[SPL];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int r = tid + NT * i;
    at[i] = r < n_rows ? r : (r == n_rows ? CS_MAX_SLOTS : -1);
    act[i] = false;
  }
  bool got = cs_gather<SPL>(xall, G, at, tag, &a.sync[1], dd);
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int r = tid + NT * i;
    if (r < n_rows) {
      const bool ypos = (cur.rf[i] & 0x8000u) != 0u;
      const float yd = ypos ? dd[i] : -dd[i];
      const bool active = got && !(yd < 0.0f);                 // ref: core/ml/SparseSVM.scala:27-28
      coef[r] = active ? (ypos ? qscale : -qscale) : 0.0f;
      z.n_act += (active && b == 0) ? 1u : 0u;
      act[i] = active;
    } else if (r == n_rows) {
      red[16] = a.lambda * 2.0f * dd[i];
    }
  }
  if (a.gate_rec != nullptr && b == 0) {   // (workgroup-uniform) the decisions on record: a wave's 64 rows are two words
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
      const unsigned long long m = __ballot(act[i]);
      const int r0 = (tid & ~63) + NT * i;
      if ((tid & 63) == 0 && r0 < n_rows) {
        unsigned int* g = a.gate_rec + step * (long long)a.gate_words + (r0 >> 5);
        g[0] = (unsigned int)m;
        if (r0 + 32 < n_rows) g[1] = (unsigned int)(m >> 32);
      }
    }
  }
  if (n_rows == NT * SPL && tid == NT - 1) {
    const int at1[1] = {CS_MAX_SLOTS};
    float d1[1];
    got = cs_gather<1>(xall, G, at1, tag, &a.sync[1], d1) && got;
    red[16] = a.lambda * 2.0f * d1[0];
  }
  if (!got) reinterpret_cast<int*>(red)[17] = 1;
  cs_barrier();
  if (reinterpret_cast<int*>(red)[17]) return false;
  stamp(2);
  const float s = red[16];
  const bool add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
  if (a.s_rec != nullptr && b == 0 && tid == 0) a.s_rec[step] = s;
  // ---- 4: y * x of the active rows into the accumulator of the row's worker (exact integer sums; the non-zeros are
  //         still in registers).  ref: core/Slave.scala:147-153 restricted to this slice's columns ----
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int slot = tid + NT * i;
    if (slot < n_slots) {
      const float cf = coef[cur.meta[i] & 0xffffu];
      if (cf != 0.0f) {
        int* ak = z.acc + (int)((cur.meta[i] >> 16) & 15u) * Sp;
#pragma unroll
        for (int j = 0; j < CS_L; ++j) {
          const int q = __float2int_rn(CS_VAL(cur, i, j) * cf);
          if (q != 0) atomicAdd(&ak[CS_COL(cur, i, j)], q);
        }
      }
    }
  }
  cs_barrier();
  stamp(3);
  // ---- 5: the columns this step can have touched (the plan lists them: dense lanes -- a sweep over all 5,905 columns of
  //         the slice spent 9.8 of a 3 x 100 step's 15.8 us on the ~80 % it does not touch): per worker ONE rounding
  //         of the exact sum, the support-only regulariser, the fold over the workers, the mean, the update --
  //         dsgd_fix_reduce_apply_kernel's arithmetic (fra_update_and_scalars) ----
  switch (K) {   // (straight-line code per worker count: the loop to CS_MAX_K under "k < K" was a branch per worker and column)
    case 1: cs_sweep<NT, 1, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 2: cs_sweep<NT, 2, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 3: cs_sweep<NT, 3, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 4: cs_sweep<NT, 4, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 5: cs_sweep<NT, 5, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 6: cs_sweep<NT, 6, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    case 7: cs_sweep<NT, 7, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
    default: cs_sweep<NT, 8, CLT>(z, cur.cl, n_cols, Sp, inv_scale, s, add, a.lr); break;
  }
  cs_barrier();
  // ... and this slice's share of w . ds of the new weights: all columns, four per lane and round (padding holds zeros)
  const float spn = cs_wds_share<NT>(z);
  stamp(4);
  z.sp = cs_block_sum<NT>(spn, red);
  stamp(5);
  ++z.n_rel;
  return true;
}

// the slice's end of a launch: statistics, and for a per-request step the host-mapped mailbox (slice 0)
template <int NT>
__device__ __forceinline__ void cs_finish_stats(const CsArgs& a, CsState& z, bool ok) {
  const int tid = threadIdx.x;
  if (z.b != 0) return;
  const unsigned int n_act = wave_sum_u32(ok ? z.n_act : 0u);
  __syncthreads();
  if ((tid & 63) == 0) reinterpret_cast<unsigned int*>(z.red)[tid >> 6] = n_act;
  __threadfence();   // (this workgroup's error flags, if any, are out before thread 0 reads them)
  __syncthreads();
  if (tid == 0) {
    const unsigned int* r4 = reinterpret_cast<const unsigned int*>(z.red);
    unsigned int tot = 0u;
    for (int i = 0; i < NT / 64; ++i) tot += r4[i];
    unsigned long long now = 0ull;
    if (tot || a.mail) now = atomicAdd(&a.sc->n_active, (unsigned long long)tot) + tot;
    if (a.mail) {
      __hip_atomic_store(&a.mail[0], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const int err = __hip_atomic_load(&a.sc->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&a.mail[1], (unsigned long long)(unsigned int)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // the request's sequence number LAST, with release order: the host may take the two words above once it sees it
      __hip_atomic_store(&a.mail[2], a.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The launch: set-up (slice-major weights and dimSparsity into LDS, accumulators cleared), the steps, the write-back.
// `pre`: called between the set-up's requests and their use (the per-request kernel lays the step's slots out there, with
// the launch's LDS still free beyond the two vectors); returns false to give the launch up before its first step.
template <int NT, int SPL, int CLT, bool REQ, class Pre>
__device__ __forceinline__ void cs_launch_body(const CsArgs& a, float* lds, Pre pre) {
  const unsigned long long t_launch = a.tprof ? __builtin_readcyclecounter() : 0ull;
  const int tid = threadIdx.x;
  const int G = a.G, K = a.K;
  CsState z;
  z.b = blockIdx.x;
  const int S = (a.dp + G - 1) / G;                    // local columns of the widest slice
  z.Sb = (a.dp - z.b + G - 1) / G;                     // ... of this one (ranks b, b + G, ...)
  z.Sp = (S + 4) & ~3;                                 // (at least one padding column: cs_sweep parks idle lanes on it)
  z.w_l = lds;
  z.ds_l = lds + z.Sp;
  z.acc = reinterpret_cast<int*>(lds + 2 * z.Sp);
  z.ps = lds + (2 + K) * z.Sp;
  z.coef = z.ps + CS_MAX_SLOTS;
  z.red = z.coef + CS_MAX_SLOTS;
  z.n_act = 0u;
  z.n_rel = 0u;
  z.tp = reinterpret_cast<unsigned long long*>(z.red + 18);
  // the first step's slots are requested before anything else: they land with the weights.  (The weights used to be picked
  // out of the rank-ordered vector here -- every slice touched every 128-byte line of w and ds, 378 KB through ONE CU:
  // 8.6 of the 18 us of a one-step launch.)
  CsSet<SPL, CLT> A, B;
  if (!REQ) cs_issue<NT, SPL, CLT>(a, z.b, a.step_begin, A);
  {   // the slice's weights and dimSparsity values: two contiguous pieces (the padding holds zeros), requested at once; the
      // accumulators are cleared while the first requests are on their way (clearing first cost 1.3 us of a one-step launch)
    const float4* ws4 = reinterpret_cast<const float4*>(a.w + (long long)z.b * z.Sp);
    const float4* ds4 = reinterpret_cast<const float4*>(a.ds + (long long)z.b * z.Sp);
    float4* wl4 = reinterpret_cast<float4*>(z.w_l);
    float4* dl4 = reinterpret_cast<float4*>(z.ds_l);
    constexpr int UB = 4;
    const int n4 = z.Sp >> 2;
    for (int i0 = tid, round = 0; round == 0 || i0 < n4; i0 += NT * UB, ++round) {   // (every lane runs the first round)
      float4 wv[UB], dv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i4 = i0 + NT * u;
        wv[u] = ws4[i4 < n4 ? i4 : n4 - 1];
        dv[u] = ds4[i4 < n4 ? i4 : n4 - 1];
      }
      if (round == 0 && !REQ)
        for (int i = tid; i < K * z.Sp; i += NT) z.acc[i] = 0;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i4 = i0 + NT * u;
        if (i4 < n4) {
          wl4[i4] = wv[u];
          dl4[i4] = dv[u];
        }
      }
    }
  }
  if (REQ) {
    // a per-request step: ITS slots are laid out now, by this workgroup for its own slice (the LDS beyond the two vectors
    // is the builder's scratch), written to the context's one-step layout and requested back like a plan's
    const bool built = pre(reinterpret_cast<unsigned int*>(z.acc));
    __threadfence();      // the layout this workgroup stored is what it loads next: out of the CU, the L1 lines dropped
    __syncthreads();
    if (!built) {         // (workgroup-uniform) the step does not fit the one-step layout: every peer is told, nobody publishes
      if (tid == 0) {
        // the soft-fallback flag FIRST, the abort word behind it with release order: a peer that sees the abort (and raises
        // 8, "timed out") can only publish a report in which 16 already stands -- finish_mail then reads "fall back", not a
        // hard error
        __hip_atomic_fetch_or(&a.sc->err, 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&a.sync[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      cs_finish_stats<NT>(a, z, false);
      return;
    }
    cs_issue<NT, SPL, CLT>(a, z.b, a.step_begin, A);
    for (int i = tid; i < K * z.Sp; i += NT) z.acc[i] = 0;
  }
  if (tid == 0)
    for (int i = 0; i < 6; ++i) z.tp[i] = 0ull;
  cs_barrier();
  // this slice's share of w . ds of the weights the launch starts from: the SAME pass, lane assignment and order as behind
  // every step -- a plan run step by step and in one launch see the same bits
  z.sp = cs_block_sum<NT>(cs_wds_share<NT>(z), z.red);
  if (tid == 0 && a.tprof) {
    z.tp[6] = __builtin_readcyclecounter();
    if (z.b == 0) a.tprof[6] += z.tp[6] - t_launch;   // the set-up of the launch
  }
  bool ok = true;
  for (long long step = a.step_begin; step < a.step_end; step += 2) {   // two register sets, rotated by unrolling
    ok = cs_step<NT, SPL, CLT>(a, z, A, B, step);
    if (!ok || step + 1 >= a.step_end) break;
    ok = cs_step<NT, SPL, CLT>(a, z, B, A, step + 1);
    if (!ok) break;
  }
  if (!ok) {
    // Given up.  NO slice has written its weights back: a slice writes back only behind the LAST step's gather, which
    // needs the last step's granules of all G slices -- published, by each, behind its gather of the step before, and so
    // on down: whoever passes the last gather has seen every peer reach the last step, and a peer that has published its
    // last granules polls nothing any more that could make it give up, except those same G granules, which are there.
    // So either every slice writes back or none does; here none: global w is what the launch found, the host rejects it.
    if (tid == 0) atomicOr(&a.sc->err, 8);
    cs_finish_stats<NT>(a, z, false);
    return;
  }
  {
    float4* ws4 = reinterpret_cast<float4*>(a.w + (long long)z.b * z.Sp);
    const float4* wl4 = reinterpret_cast<const float4*>(z.w_l);
    for (int i4 = tid; i4 < (z.Sp >> 2); i4 += NT) ws4[i4] = wl4[i4];
  }
  if (a.tprof && z.b == 0 && tid == 0) {
    a.tprof[7] += __builtin_readcyclecounter() - z.tp[6];   // the write-back
    for (int i = 0; i < 6; ++i) a.tprof[i] += z.tp[i];
    a.tprof[15] += (unsigned long long)(a.step_end - a.step_begin);
  }
  cs_finish_stats<NT>(a, z, true);
}

template <int NT, int SPL, int CLT>
__global__ void __launch_bounds__(NT) dsgd_cs_step_kernel(CsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  cs_launch_body<NT, SPL, CLT, false>(a, lds, [](unsigned int*) { return true; });
}

// ---- the layout of a step's slots, built ON THE DEVICE ------------------------------------------------------------
// Rounds 1-4 laid a plan's column slices out on the host: the listed rows' (rank, value) pairs gathered on the device,
// copied back (8 bytes per non-zero), bucketed per slice by one host thread, copied up again -- 0.6 ms per 3 x 100 step
// (1.3 s for the 2,146 steps of an epoch over 804,414 rows: a hundred times the 11 ms the epoch then RUNS), and nothing a
// per-request step (core/Slave.scala:142-157 hands over fresh index lists every call) could use at all.  Here one
// workgroup builds one (step, slice) cell from the resident ranked CSR: the step's rows, the entries whose rank is the
// slice's (mod G) in CSR order, cut into slots of <= CS_L -- byte for byte the host's layout (the partial x.w of a slot and
// of a row are sums in entry order: the same bits), so both builders can be checked against each other.
//   pass 1 (FILL = false): slots and distinct columns of the cell -> device-wide maxima (the strides of the layout);
//   pass 2 (FILL = true):  everything the step kernel may read of the cell: header, row_first, slot_meta, the column
//                          and value pieces (zeros beyond a row's last entry and in the slots beyond the step's), the
//                          sorted list of the step's columns inside the slice (0xffff beyond it).
// The per-request kernel (dsgd_cs_request_kernel) calls pass 2 for ITS OWN slice in front of the step.
constexpr int CS_BUILD_BITMAP_WORDS = 2048;   // a slice holds at most 65,536 columns
constexpr int CS_BUILD_WORDS = (CS_MAX_SLOTS + 8) + CS_MAX_SLOTS + 2 * CS_MAX_SLOTS + CS_BUILD_BITMAP_WORDS + 64 + CS_THREADS;   // LDS words of scratch

static_assert(CS_BUILD_WORDS <= 7000, "cs_req_lds_words reserves 7000 words for the builder");
struct CsBuildArgs {
  CsrView m;                 // the resident CSR, columns as frequency ranks
  const int* idx;            // the lists' rows
  const WorkSeg* segs;       // [n_steps][K]: positions [begin, end) in idx; a step's lists are contiguous
  CsHdr* hdr;                // outputs: the arrays of CsArgs
  unsigned int* slot_meta;
  unsigned short* row_first;
  unsigned short* col;
  float* val;
  unsigned short* clist;
  unsigned int* maxima;      // [0] most slots of a cell, [1] most listed columns, [2] flags: 1 a row index outside the data,
                             //   2 a cell that does not fit the strides, [3] most rows of a step
  long long n_steps_plan;
  int slot_stride, row_stride, cl_stride;
  int dp, G, K;
};

// exclusive prefix sums over n <= 2 * NT values in LDS (in place), total returned on every thread; `tmp`: NT / 64 + 1 words
template <int NT>
__device__ __forceinline__ unsigned int cs_excl_scan(unsigned int* v, int n, unsigned int* tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i0 = 2 * tid, i1 = 2 * tid + 1;
  const unsigned int a0 = i0 < n ? v[i0] : 0u, a1 = i1 < n ? v[i1] : 0u;
  unsigned int x = a0 + a1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned int y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) tmp[wv] = x;
  __syncthreads();
  unsigned int base = 0u, total = 0u;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) {
    const unsigned int t = tmp[i];
    if (i < wv) base += t;
    total += t;
  }
  const unsigned int excl = base + x - (a0 + a1);
  if (i0 < n) v[i0] = excl;
  if (i1 < n) v[i1] = excl + a0;
  __syncthreads();
  return total;
}

// One (step, slice) cell.  Returns true when the cell fits the strides (always in pass 1).  Workgroup-uniform.
template <int NT, bool FILL>
__device__ __forceinline__ bool cs_build_cell(const CsBuildArgs& a, int b, long long s, unsigned int* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NW = NT / 64;
  const int G = a.G, K = a.K;
  const bool pow2 = (G & (G - 1)) == 0;
  const int gsh = 31 - __clz(G);
  // scratch: per-row slot counts (then their prefix), row lengths, row starts, the column bitmap, the scan's wave totals
  unsigned int* nslot = scratch;                                   // [CS_MAX_SLOTS + 8]
  unsigned int* rlen = nslot + CS_MAX_SLOTS + 8;                   // [CS_MAX_SLOTS]
  unsigned long long* rst = reinterpret_cast<unsigned long long*>(rlen + CS_MAX_SLOTS);   // [CS_MAX_SLOTS] (8-byte aligned: even offset)
  unsigned int* bitmap = reinterpret_cast<unsigned int*>(rst + CS_MAX_SLOTS);   // [CS_BUILD_BITMAP_WORDS]
  unsigned int* tmp = bitmap + CS_BUILD_BITMAP_WORDS;              // [64], then the scan's [NT] per-thread totals
  const WorkSeg* sg = a.segs + s * K;
  const long long t0 = sg[0].begin;
  const long long Rl = sg[K - 1].end - t0;
  const int R = (int)(Rl < (long long)CS_MAX_SLOTS ? Rl : (long long)CS_MAX_SLOTS);
  long long worst = 1;
  for (int k = 0; k < K; ++k) worst = worst > sg[k].end - sg[k].begin ? worst : sg[k].end - sg[k].begin;
  const int Sloc = (a.dp + G - 1) / G;                             // slice-local columns
  const int bm_words = (Sloc + 31) >> 5;
  for (int i = tid; i < bm_words; i += NT) bitmap[i] = 0u;
  unsigned int bad = 0u;
  // the rows of the step: where they start, how long they are (every lane its own row: two dependent loads, all in flight)
  for (int r = tid; r < R; r += NT) {
    long long row = a.idx[t0 + r];
    if (row < 0 || row >= a.m.n_rows) {
      bad = 1u;
      row = 0;
    }
    const long long st = a.m.row_ptr[row], en = a.m.row_ptr[row + 1];
    rst[r] = (unsigned long long)st;
    rlen[r] = bad ? 0u : (unsigned int)(en - st);
  }
  __syncthreads();
  // count: one wave per row, the row's entries 64 at a time
  for (int r = wv; r < R; r += NW) {
    const long long st = (long long)rst[r];
    const int len = (int)rlen[r];
    int cnt = 0;
    for (int j0 = 0; j0 < len; j0 += 64) {
      const int j = j0 + lane;
      const int rank = j < len ? a.m.col[st + j] : -1;
      const bool mine = rank >= 0 && (pow2 ? (rank & (G - 1)) == b : rank % G == b);
      if (mine) {
        const int lc = pow2 ? rank >> gsh : rank / G;
        atomicOr(&bitmap[lc >> 5], 1u << (lc & 31));
      }
      cnt += __popcll(__ballot(mine));
    }
    if (lane == 0) nslot[r] = (unsigned int)((cnt + CS_L - 1) / CS_L);
  }
  __syncthreads();
  const unsigned int n_slots = cs_excl_scan<NT>(nslot, R, tmp);   // nslot[r] = first slot of row r
  // the distinct columns: per thread a run of bitmap words, exclusive prefix of their popcounts
  const int wpt = (bm_words + NT - 1) / NT;                        // words per thread (<= 4)
  unsigned int mypop = 0u;
  for (int i = 0; i < wpt; ++i) {
    const int wd = tid * wpt + i;
    mypop += wd < bm_words ? (unsigned int)__popc(bitmap[wd]) : 0u;
  }
  unsigned int* cs_pop = tmp + 64;   // [NT]
  cs_pop[tid] = mypop;
  __syncthreads();
  const unsigned int n_cols = cs_excl_scan<NT>(cs_pop, NT, tmp);
  const unsigned int col_base = cs_pop[tid];
  if (tid == 0) {
    atomicMax(&a.maxima[0], n_slots);
    atomicMax(&a.maxima[1], n_cols);
    atomicMax(&a.maxima[3], (unsigned int)(Rl > 0x7fffffffLL ? 0x7fffffffLL : Rl));
  }
  if (__syncthreads_or(bad != 0u) && tid == 0) atomicOr(&a.maxima[2], 1u);
  if (!FILL) return true;
  const bool fits = Rl <= (long long)CS_MAX_SLOTS && (int)n_slots <= a.slot_stride && R + 1 <= a.row_stride &&
                    (int)n_cols <= a.cl_stride && n_slots <= (unsigned int)CS_MAX_SLOTS;
  if (!fits) {
    if (tid == 0) atomicOr(&a.maxima[2], 2u);
    return false;
  }
  const long long cell = (long long)b * a.n_steps_plan + s;
  const long long sbase = cell * a.slot_stride, rbase = cell * a.row_stride, cbase = cell * a.cl_stride;
  // everything the step kernel may read of the cell, zeroed first (coalesced 16-byte stores) ...
  {
    uint4* c4 = reinterpret_cast<uint4*>(a.col + 2 * sbase * 8);      // [2][slot_stride] pieces of 8 columns
    uint4* v4 = reinterpret_cast<uint4*>(a.val + 4 * sbase * 4);      // [4][slot_stride] pieces of 4 values
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * a.slot_stride; i += NT) c4[i] = zero;
    for (int i = tid; i < 4 * a.slot_stride; i += NT) v4[i] = zero;
    for (int i = tid; i < a.slot_stride; i += NT) a.slot_meta[sbase + i] = 0u;
  }
  // ... the listed columns, ascending (0xffff beyond them) ...
  {
    unsigned int at = col_base;
    for (int i = 0; i < wpt; ++i) {
      const int wd = tid * wpt + i;
      unsigned int bits = wd < bm_words ? bitmap[wd] : 0u;
      while (bits) {
        const int bit = __ffs((int)bits) - 1;
        bits &= bits - 1u;
        a.clist[cbase + at++] = (unsigned short)(wd * 32 + bit);
      }
    }
    for (int i = (int)n_cols + tid; i < a.cl_stride; i += NT) a.clist[cbase + i] = (unsigned short)0xffffu;
  }
  // ... the first slot of every row with its label, the sentinel, the header
  for (int r = tid; r <= R; r += NT) {
    unsigned short v;
    if (r < R) {
      const long long row = a.idx[t0 + r];
      const bool in = row >= 0 && row < a.m.n_rows;
      v = (unsigned short)(nslot[r] | ((in && a.m.label[row] > 0) ? 0x8000u : 0u));
    } else {
      v = (unsigned short)n_slots;
    }
    a.row_first[rbase + r] = v;
  }
  if (tid == 0) {
    int bits = 0;
    while ((1LL << bits) < worst) ++bits;
    CsHdr h;
    h.counts = n_slots | ((unsigned int)R << 16);
    h.shift = (30 - bits) | (int)(n_cols << 16);   // at most one contribution per row and column: a worker's sums stay below 2^30
    a.hdr[cell] = h;
  }
  __syncthreads();   // (the zeros above are out -- vmcnt(0) -- before another lane stores an entry over them)
  // fill: one wave per row again (the row's lines are in the caches), entry q of the row inside the slice -> slot q / 16
  for (int r = wv; r < R; r += NW) {
    const long long st = (long long)rst[r];
    const int len = (int)rlen[r];
    const unsigned int first = nslot[r];
    int k = 0;
    while (k + 1 < K && t0 + r >= sg[k].end) ++k;                // the worker whose list holds the row
    int base = 0;
    for (int j0 = 0; j0 < len; j0 += 64) {
      const int j = j0 + lane;
      const int rank = j < len ? a.m.col[st + j] : -1;
      const float v = j < len ? a.m.val[st + j] : 0.0f;
      const bool mine = rank >= 0 && (pow2 ? (rank & (G - 1)) == b : rank % G == b);
      const unsigned long long mask = __ballot(mine);
      if (mine) {
        const int q = base + __popcll(mask & ((1ull << lane) - 1ull));
        const long long sl = (long long)first + (q >> 4);
        const int e = q & 15;
        const int lc = pow2 ? rank >> gsh : rank / G;
        a.col[((2 * sbase + (long long)(e >> 3) * a.slot_stride + sl) << 3) + (e & 7)] = (unsigned short)lc;
        a.val[((4 * sbase + (long long)(e >> 2) * a.slot_stride + sl) << 2) + (e & 3)] = v;
        if (e == 0) a.slot_meta[sbase + sl] = (unsigned int)r | ((unsigned int)k << 16);
      }
      base += __popcll(mask);
    }
  }
  return true;
}

// a plan's cells: grid = n_steps x G workgroups (a step's G cells next to each other: they read the same rows)
template <bool FILL>
__global__ void __launch_bounds__(CS_THREADS) dsgd_cs_layout_kernel(CsBuildArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned int scratch[CS_BUILD_WORDS];
  const long long cell = blockIdx.x;
  const long long s = cell / a.G;
  const int b = (int)(cell - s * a.G);
  cs_build_cell<CS_THREADS, FILL>(a, b, s, scratch);
}

// A per-request step (dsgd_sync_step with the reference's batch sizes): ONE launch -- every slice's workgroup lays its own
// cell of the step out (into the context's one-step layout, strides at their maxima), then runs the step.
// ref: core/Slave.scala:142-157 + core/Master.scala:184-197 for the workers hosted here.
__global__ void __launch_bounds__(CS_THREADS) dsgd_cs_request_kernel(CsArgs a, CsBuildArgs ba) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  cs_launch_body<CS_THREADS, 2, 8, true>(a, lds, [&](unsigned int* scratch) { return cs_build_cell<CS_THREADS, true>(ba, (int)blockIdx.x, 0, scratch); });
}

// rank-ordered vector -> slice-major [G][Sp] (the padding zero) and back; one lane per rank
__global__ void __launch_bounds__(256) dsgd_cs_slice_kernel(const float* __restrict__ v, float* __restrict__ out, int dp, int G, int Sp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < G * Sp) {
    const int b = j / Sp, i = j - b * Sp;
    const long long r = (long long)b + (long long)G * i;
    out[j] = r < dp ? v[r] : 0.0f;
  }
}
__global__ void __launch_bounds__(256) dsgd_cs_unslice_kernel(const float* __restrict__ sl, float* __restrict__ v, int dp, int G, int Sp) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < dp) v[r] = sl[(long long)(r % G) * Sp + r / G];
}

// one wave per listed row: its (ranked column, value) pairs copied to out[out_ptr[t] ...] (the host lays a plan's column
// slices out from them: cs_build in dsgd_hip.hip; the ranked CSR itself lives on the device only)
__global__ void __launch_bounds__(256) dsgd_rows_gather_kernel(CsrView m, const int* __restrict__ idx, long long n,
                                                              const long long* __restrict__ out_ptr, int* __restrict__ out_col,
                                                              float* __restrict__ out_val) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long t = wave; t < n; t += n_waves) {
    const long long row = idx[t];
    const long long st = m.row_ptr[row], o = out_ptr[t];
    const long long len = out_ptr[t + 1] - o;
    for (long long j = lane; j < len; j += 64) {
      out_col[o + j] = m.col[st + j];
      out_val[o + j] = m.val[st + j];
    }
  }
}
