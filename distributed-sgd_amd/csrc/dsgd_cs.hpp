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

constexpr int CS_THREADS = 256;     // 4 waves: one per SIMD, 512 VGPRs each -- two register sets of 4 slots per lane
constexpr int CS_L = 16;            // entries per slot
constexpr int CS_MAX_G = 16;        // slices = workgroups
constexpr int CS_MAX_K = 8;         // hosted workers
constexpr int CS_MAX_SPL = 4;       // slots (and rows) per lane
constexpr int CS_MAX_SLOTS = CS_THREADS * CS_MAX_SPL;   // per (step, slice); also the most rows of a step
constexpr int CS_XSTRIDE = CS_MAX_SLOTS + 64;           // granules per (parity, slice) of the exchange buffer: [row] partial x.w, [CS_MAX_SLOTS] share of w . ds
constexpr unsigned int CS_POLL_LIMIT = 1u << 18;        // polls of one granule set before the launch is given up

struct CsHdr {                // per (slice, step)
  unsigned int counts;        // slots of the step inside the slice (low 16 bits) | rows of the step << 16
  int shift;                  // fixed-point shift of the step: 30 - ceil(log2(largest list)) (the same in every slice)
};

struct CsArgs {
  const CsHdr* hdr;                 // [G][n_steps]
  const unsigned int* slot_meta;    // [G][n_steps][slot_stride]: row of the step (bits 0-15) | worker (bits 16-19)
  const unsigned short* row_first;  // [G][n_steps][row_stride]: first slot of row r (bits 0-10; entry n_rows = n_slots) | label > 0 (bit 15)
  const unsigned short* col;        // [G][n_steps][slot_stride][CS_L]: slice-local column (rank / G); padding: column 0, value 0
  const float* val;
  float* w;                         // ranked weights (read when the launch starts, written when it ends)
  const float* ds;
  unsigned long long* xbuf;         // [2][G][CS_XSTRIDE] granules {value bits, step tag << 32}; zero when a launch starts
  unsigned int* sync;               // [1] abort word (zero when a launch starts)
  DevScalars* sc;
  unsigned long long* tprof;        // optional (tuning runs, DSGD_PLAN_PROF=1): cycles of thread 0 of slice 0 by phase, [15] = steps
  long long n_steps_plan, step_begin, step_end;
  int slot_stride, row_stride;
  float lr, lambda;
  int vexp, dp, G, K;
};

__host__ __device__ constexpr int cs_lds_words(int dp, int G, int K) {
  return (2 + K) * ((((dp + G - 1) / G) + 3) & ~3) + 2 * CS_MAX_SLOTS + 32;
}

template <int SPL>
struct CsSet {              // the slots of one step, as loaded (nothing is computed on them before their step runs)
  uint4 c[SPL][2];          // 16 slice-local columns, 16 bits each
  float4 v[SPL][4];
  unsigned int meta[SPL];
  unsigned int rf[SPL];     // row_first[r] | row_first[r + 1] << 16 of row r = tid + CS_THREADS * i
  uint2 h;                  // the step's header
};

template <int SPL>
__device__ __forceinline__ void cs_issue(const CsArgs& a, int b, long long step, CsSet<SPL>& R) {
  // every request unconditional, indices clamped (a step beyond the launch's last one re-reads the last and is never used)
  const long long sc = step < a.step_end ? step : a.step_end - 1;
  const long long sidx = (long long)b * a.n_steps_plan + sc;
  const uint2* hp = reinterpret_cast<const uint2*>(a.hdr + sidx);
  asm volatile("" : "+v"(hp));   // a vector load (vmcnt): a scalar one would share lgkmcnt with the LDS traffic of the whole step
  R.h = *hp;
  const long long sbase = sidx * a.slot_stride, rbase = sidx * a.row_stride;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    int slot = (int)threadIdx.x + CS_THREADS * i;
    slot = slot < a.slot_stride ? slot : a.slot_stride - 1;
    const uint4* cp = reinterpret_cast<const uint4*>(a.col + (sbase + slot) * CS_L);
    const float4* vp = reinterpret_cast<const float4*>(a.val + (sbase + slot) * CS_L);
    R.c[i][0] = cp[0];
    R.c[i][1] = cp[1];
    R.v[i][0] = vp[0];
    R.v[i][1] = vp[1];
    R.v[i][2] = vp[2];
    R.v[i][3] = vp[3];
    R.meta[i] = a.slot_meta[sbase + slot];
    int r = (int)threadIdx.x + CS_THREADS * i;
    r = r < a.row_stride - 1 ? r : a.row_stride - 2;
    R.rf[i] = (unsigned int)a.row_first[rbase + r] | ((unsigned int)a.row_first[rbase + r + 1] << 16);
  }
}

// the G granules of exchange slot `at` (a row, or CS_MAX_SLOTS for the shares of w . ds), polled until all carry `tag`;
// the values added in slice order.  false = given up (the abort word is raised for everybody).
__device__ __forceinline__ bool cs_gather(const unsigned long long* xall, int G, int at, unsigned int tag, unsigned int* abort_word,
                                          float& sum) {
  unsigned long long v[CS_MAX_G];
  for (unsigned int spin = 0;; ++spin) {
    bool all = true;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      v[g] = __hip_atomic_load(&xall[(long long)g * CS_XSTRIDE + at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      all = all && (unsigned int)(v[g] >> 32) == tag;
    }
    if (G > 8) {   // (workgroup-uniform: G is 8 or 16)
#pragma unroll
      for (int g = 8; g < CS_MAX_G; ++g) {
        v[g] = __hip_atomic_load(&xall[(long long)g * CS_XSTRIDE + at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        all = all && (unsigned int)(v[g] >> 32) == tag;
      }
    } else {
#pragma unroll
      for (int g = 8; g < CS_MAX_G; ++g) v[g] = 0ull;
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
  float d = 0.0f;
#pragma unroll
  for (int g = 0; g < CS_MAX_G; ++g) d += __uint_as_float((unsigned int)v[g]);   // (the zeros of an 8-slice run change nothing)
  sum = d;
  return true;
}

// Workgroup barrier that orders LDS only.  __syncthreads() is a workgroup-scope FENCE: it waits for every outstanding
// memory operation of the wave (s_waitcnt vmcnt(0)) -- here that is the next step's slots, requested at the top of the
// step precisely so that they stay in flight under it (measured with __syncthreads(): the first barrier of every step
// waited out the whole HBM round trip, 1.9 us at 3 x 100 and 8 us at 4 x 200).  Everything the barriers of a step
// order lives in LDS; what crosses workgroups goes through the tagged granules, which need no ordering.
__device__ __forceinline__ void cs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the workgroup, the same bits on every thread (wave butterflies, then the four wave sums in order)
__device__ __forceinline__ float cs_block_sum(float v, float* red4) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  cs_barrier();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  cs_barrier();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// the LDS carve and the launch-long state of a workgroup
struct CsState {
  float* w_l;
  float* ds_l;
  int* acc;       // [K][Sp], zero between steps
  float* ps;      // partial x.w per slot
  float* coef;    // per row of the step: +-2^shift / vmax2 (active) or 0
  float* red;     // [0..3] wave sums, [8] s of the step, [9] abort flag
  int b, Sb, Sp;
  float sp;       // this slice's share of w . ds of the current weights (the same bits on every thread)
  unsigned int n_act, n_rel;   // active rows counted (slice 0 only); steps of this launch behind us
  unsigned long long tp[6], tl; // tuning runs: cycles by phase (dot, publish, exchange, scatter, sweep, reduce), last stamp
};

// One step.  `cur`: the step's slots (landed); `nxt` receives the next step's.  false = the launch was aborted.
template <int SPL>
__device__ __forceinline__ bool cs_step(const CsArgs& a, CsState& z, CsSet<SPL>& cur, CsSet<SPL>& nxt, long long step) {
  const int tid = threadIdx.x, G = a.G, K = a.K, b = z.b, Sp = z.Sp;
  float* const ps = z.ps;
  float* const coef = z.coef;
  float* const red = z.red;
  const int n_slots = __builtin_amdgcn_readfirstlane((int)(cur.h.x & 0xffffu));
  const int n_rows = __builtin_amdgcn_readfirstlane((int)(cur.h.x >> 16));
  const int shift = __builtin_amdgcn_readfirstlane((int)cur.h.y);
  const float qscale = ldexpf(1.0f, shift - a.vexp);
  const double inv_scale = (double)ldexpf(1.0f, a.vexp - shift);
  const bool prof = a.tprof != nullptr && b == 0 && tid == 0;
  auto stamp = [&](int i) {
    if (prof) {
      const unsigned long long now = __builtin_readcyclecounter();
      z.tp[i] += now - z.tl;
      z.tl = now;
    }
  };
  // ---- 0: the NEXT step's slots are requested first (a whole step to land; vmcnt retires in order, so the polls of this
  //         step's exchange return no earlier than these -- requested later they would be waited for at the next step's top) ----
  cs_issue<SPL>(a, b, step + 1, nxt);
  // ---- 1: partial x.w of every slot from this slice's weights (ref: math/Vec.scala:58, math/Sparse.scala:46) ----
  int cc[SPL][CS_L];
  float vv[SPL][CS_L];
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const unsigned int cw[8] = {cur.c[i][0].x, cur.c[i][0].y, cur.c[i][0].z, cur.c[i][0].w,
                                cur.c[i][1].x, cur.c[i][1].y, cur.c[i][1].z, cur.c[i][1].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cc[i][2 * j] = (int)(cw[j] & 0xffffu);
      cc[i][2 * j + 1] = (int)(cw[j] >> 16);
    }
    const float4 v0 = cur.v[i][0], v1 = cur.v[i][1], v2 = cur.v[i][2], v3 = cur.v[i][3];
    const float t[CS_L] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    float p = 0.0f;
#pragma unroll
    for (int j = 0; j < CS_L; ++j) {
      vv[i][j] = t[j];
      p += filt(t[j] * z.w_l[cc[i][j]]);
    }
    const int slot = tid + CS_THREADS * i;
    if (slot < n_slots) ps[slot] = p;
  }
  if (tid == 0) reinterpret_cast<int*>(red)[9] = 0;   // (this step's "given up" flag: raised in phase 3, read behind its barrier)
  cs_barrier();
  stamp(0);
  // ---- 2: this slice's partial of every row (its slots in order), published as granules {value, tag}: write-through,
  //         nothing waits for them ----
  const unsigned int tag = z.n_rel + 1u;
  unsigned long long* xb = a.xbuf + ((long long)(z.n_rel & 1u) * G + b) * CS_XSTRIDE;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int r = tid + CS_THREADS * i;
    if (r < n_rows) {
      const int f0 = (int)(cur.rf[i] & 0x7ffu), f1 = (int)((cur.rf[i] >> 16) & 0x7ffu);
      float t = 0.0f;
      for (int f = f0; f < f1; ++f) t += ps[f];
      __hip_atomic_store(&xb[r], ((unsigned long long)tag << 32) | __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (tid == 0) {
    __hip_atomic_store(&xb[CS_MAX_SLOTS], ((unsigned long long)tag << 32) | __float_as_uint(z.sp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamp(1);
  // ---- 3: every slice's granules of this thread's rows: x.w in slice order, the gate, the row's coefficient ----
  const unsigned long long* xall = a.xbuf + (long long)(z.n_rel & 1u) * G * CS_XSTRIDE;
  bool got = true;
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int r = tid + CS_THREADS * i;
    if (r < n_rows && got) {
      float d = 0.0f;
      got = cs_gather(xall, G, r, tag, &a.sync[1], d);
      const bool ypos = (cur.rf[i] & 0x8000u) != 0u;
      const float yd = ypos ? d : -d;
      const bool active = got && !(yd < 0.0f);                 // ref: core/ml/SparseSVM.scala:27-28
      coef[r] = active ? (ypos ? qscale : -qscale) : 0.0f;
      z.n_act += (active && b == 0) ? 1u : 0u;
    }
  }
  if (tid == CS_THREADS - 1) {   // s = 2 lambda (w . ds) of the weights this step's gradients see: the slices' shares in slice order
    float dsum = 0.0f;
    got = got && cs_gather(xall, G, CS_MAX_SLOTS, tag, &a.sync[1], dsum);
    red[8] = a.lambda * 2.0f * dsum;
  }
  if (!got) reinterpret_cast<int*>(red)[9] = 1;
  cs_barrier();
  if (reinterpret_cast<int*>(red)[9]) return false;
  stamp(2);
  const float s = red[8];
  const bool add = (s != 0.0f) && (fabsf(s) > DSGD_EPS);
  // ---- 5: y * x of the active rows into the accumulator of the row's worker (exact integer sums; the non-zeros are
  //         still in registers).  ref: core/Slave.scala:147-153 restricted to this slice's columns ----
#pragma unroll
  for (int i = 0; i < SPL; ++i) {
    const int slot = tid + CS_THREADS * i;
    if (slot < n_slots) {
      const float cf = coef[cur.meta[i] & 0xffffu];
      if (cf != 0.0f) {
        int* ak = z.acc + (int)((cur.meta[i] >> 16) & 15u) * Sp;
#pragma unroll
        for (int j = 0; j < CS_L; ++j) {
          const int q = __float2int_rn(vv[i][j] * cf);
          if (q != 0) atomicAdd(&ak[cc[i][j]], q);
        }
      }
    }
  }
  cs_barrier();
  stamp(3);
  // ---- 6: this slice's columns: per worker ONE rounding of the exact sum, the support-only regulariser, the fold over
  //         the workers, the mean, the update -- dsgd_fix_reduce_apply_kernel's arithmetic (fra_update_and_scalars) ----
  // Four adjacent columns per thread and round, every LDS request of a round issued before the first is used (the first
  // form walked one column at a time with a dependent LDS read per worker: 1,000 cycles per column, 9.8 of a 3 x 100
  // step's 15.8 us).  Columns beyond this slice's last one hold zeros in all arrays.
  float spn = 0.0f;
  {
    const int n4 = Sp >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(z.w_l);
    const float4* d4 = reinterpret_cast<const float4*>(z.ds_l);
    int4* a4 = reinterpret_cast<int4*>(z.acc);
    for (int i4 = tid; i4 < n4; i4 += CS_THREADS) {
      int4 t[CS_MAX_K];
#pragma unroll
      for (int k = 0; k < CS_MAX_K; ++k) t[k] = k < K ? a4[k * n4 + i4] : make_int4(0, 0, 0, 0);
      const float4 wo = w4[i4], dv = d4[i4];
      int any = 0;
#pragma unroll
      for (int k = 0; k < CS_MAX_K; ++k) any |= t[k].x | t[k].y | t[k].z | t[k].w;
      float wn[4] = {wo.x, wo.y, wo.z, wo.w};
      if (any != 0) {
        float gsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < CS_MAX_K; ++k) {
          if (k < K) {
            const int tk[4] = {t[k].x, t[k].y, t[k].z, t[k].w};
            if ((tk[0] | tk[1] | tk[2] | tk[3]) != 0) a4[k * n4 + i4] = make_int4(0, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (tk[e] != 0) {
                float gv = filt((float)((double)tk[e] * inv_scale));   // one rounding of the worker's exact sum
                if (add && gv != 0.0f) gv = filt(gv + s);              // ref: core/ml/SparseSVM.scala:31, math/Vec.scala:65-75
                gsum[e] = filt(gsum[e] + gv);                          // Vec.sum over the workers
              }
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gsum[e] != 0.0f) {
            const float upd = filt(filt(gsum[e] / (float)K) * a.lr);   // Vec.mean, learningRate * grad (ref: core/Master.scala:194-197)
            wn[e] = filt(wn[e] - upd);
          }
        }
        reinterpret_cast<float4*>(z.w_l)[i4] = make_float4(wn[0], wn[1], wn[2], wn[3]);
      }
      spn += (filt(wn[0] * dv.x) + filt(wn[1] * dv.y)) + (filt(wn[2] * dv.z) + filt(wn[3] * dv.w));
    }
  }
  stamp(4);
  z.sp = cs_block_sum(spn, red);
  stamp(5);
  ++z.n_rel;
  return true;
}

template <int SPL>
__global__ void __launch_bounds__(CS_THREADS) dsgd_cs_step_kernel(CsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int G = a.G, K = a.K;
  CsState z;
  z.b = blockIdx.x;
  const int S = (a.dp + G - 1) / G;                    // local columns of the widest slice
  z.Sb = (a.dp - z.b + G - 1) / G;                     // ... of this one (ranks b, b + G, ...)
  z.Sp = (S + 3) & ~3;
  z.w_l = lds;
  z.ds_l = lds + z.Sp;
  z.acc = reinterpret_cast<int*>(lds + 2 * z.Sp);
  z.ps = lds + (2 + K) * z.Sp;
  z.coef = z.ps + CS_MAX_SLOTS;
  z.red = z.coef + CS_MAX_SLOTS;
  z.n_act = 0u;
  z.n_rel = 0u;
  for (int i = 0; i < 6; ++i) z.tp[i] = 0ull;
  float sp = 0.0f;
  for (int i = tid; i < z.Sp; i += CS_THREADS) {
    const float wv = i < z.Sb ? a.w[z.b + G * i] : 0.0f, dv = i < z.Sb ? a.ds[z.b + G * i] : 0.0f;
    z.w_l[i] = wv;
    z.ds_l[i] = dv;
    sp += filt(wv * dv);
  }
  for (int i = tid; i < K * z.Sp; i += CS_THREADS) z.acc[i] = 0;
  z.sp = cs_block_sum(sp, z.red);   // this slice's share of w . ds of the weights the launch starts from (also the barrier behind the set-up)
  CsSet<SPL> A, B;
  cs_issue<SPL>(a, z.b, a.step_begin, A);
  z.tl = a.tprof ? __builtin_readcyclecounter() : 0ull;
  bool ok = true;
  for (long long step = a.step_begin; step < a.step_end; step += 2) {   // two register sets, rotated by unrolling
    ok = cs_step<SPL>(a, z, A, B, step);
    if (!ok || step + 1 >= a.step_end) break;
    ok = cs_step<SPL>(a, z, B, A, step + 1);
    if (!ok) break;
  }
  if (!ok) {
    if (tid == 0) atomicOr(&a.sc->err, 8);
    return;   // (global w stays as the launch found it: the host rejects the run)
  }
  for (int i = tid; i < z.Sb; i += CS_THREADS) a.w[z.b + G * i] = z.w_l[i];
  if (a.tprof && z.b == 0 && tid == 0) {
    for (int i = 0; i < 6; ++i) a.tprof[i] += z.tp[i];
    a.tprof[15] += (unsigned long long)(a.step_end - a.step_begin);
  }
  if (z.b == 0) {
    const unsigned int n_act = wave_sum_u32(z.n_act);
    __syncthreads();
    if ((tid & 63) == 0) reinterpret_cast<unsigned int*>(z.red)[tid >> 6] = n_act;
    __syncthreads();
    if (tid == 0) {
      const unsigned int* r4 = reinterpret_cast<const unsigned int*>(z.red);
      const unsigned int tot = r4[0] + r4[1] + r4[2] + r4[3];
      if (tot) atomicAdd(&a.sc->n_active, (unsigned long long)tot);
    }
  }
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
