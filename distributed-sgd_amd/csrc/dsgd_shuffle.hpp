// Device code of libdsgd_hip, part 7 (gfx950 only): the index lists of an EPOCH of Master.fit, drawn on the device --
// draw for draw the reference's own random stream.  Included by dsgd_hip.hip after dsgd_tcol.hpp.
//
// ref: core/Master.scala:184 -- for EVERY batch of an epoch every worker's whole split is reshuffled and sliced,
//          workers.zip(split.map(Random.shuffle(_))).map { case (worker, idx) => idx.slice(batch, batch + batchSize) }
//      scala.util.Random.shuffle (2.12): for (n <- len to 2 by -1) swap(n - 1, nextInt(n)) on a copy;
//      java.util.Random: seed' = seed * 0x5DEECE66D + 0xB (mod 2^48), next(31) = seed' >> 17, nextInt(bound): a power of
//      two -> (bound * next(31)) >> 31, else r = next(31) mod bound, drawn again while `u - r + (bound - 1) < 0` (int).
//
// That is len - 1 draws per worker and BATCH: 1.38 G draws per epoch of RCV1 (full = true: 2,146 batches x 3 splits of
// 214,510 rows) for 644 K list entries.  csrc/jrand.c reproduces the stream on 32 host threads in 0.19 s per epoch -- a
// resident plan runs the epoch's 2,146 steps in 10 ms: `fit` at the reference's configuration was a host-RNG benchmark
// (batch_loop 198 us per step of which 192 the shuffle).  Here, per epoch:
//
//   dsgd_jr_scan_kernel    every lane jumps to its own piece of the raw stream (an LCG jumps ahead in O(log n)) and lists
//                          the CANDIDATES for a rejection: raw values >= 2^31 - longest split (one in ~10^4); a
//                          workgroup's candidates leave in raw order.  (jrand.c's pass A)
//   host                   walks the ~10^5 candidates once, in order: with the rejections so far known each one's bound is
//                          known -- which fixes the raw index every shuffle starts at and the few (~10) raw values
//                          rejected INSIDE each shuffle.  Sequential by nature (a rejection shifts every later draw), 0.3 ms.
//   dsgd_jr_slice_kernel   one workgroup per (batch, worker).  Only the SLICE [b, b + B) of the shuffled split is ever
//                          used, and a Fisher-Yates from the top finalises position p at step n = p + 1: traced
//                          BACKWARDS, entry p of the result is whatever sat at position k_{p+1} just before that step --
//                          i.e. the identity's value there unless a LATER-numbered (earlier-run) step n'' targeted that
//                          position (k_{n''} = q), in which case it is what sat on top (position n'' - 1) before step n'',
//                          and so on upwards.  So the kernel keeps, per entry of the slice, ONE wanted position and sweeps
//                          the steps n'' = b + 1 .. len in increasing order; every lane evaluates its own step's draw
//                          straight from the LCG (no shuffle is ever materialised: no memory traffic but a bitmap of the
//                          wanted positions in LDS) and the rare hits (~8 per entry over the whole sweep) are applied in
//                          step order.  (len - b) draw evaluations per list instead of len sequential swaps.
//
// The lists land in the plan's own index buffer (dsgd_plan_create_from_seed, include/dsgd.h): nothing crosses the host but
// the candidates (1.6 MB per RCV1 epoch) and the shuffles' start records.
#pragma once

#include <hip/hip_runtime.h>

constexpr unsigned long long JR_MULT = 0x5DEECE66DULL;
constexpr unsigned long long JR_ADD = 0xBULL;
constexpr unsigned long long JR_MASK = (1ULL << 48) - 1;

struct JrAffine {   // x -> a x + c (mod 2^48)
  unsigned long long a, c;
};
__host__ __device__ inline JrAffine jr_compose(const JrAffine& f, const JrAffine& g) {   // g after f
  JrAffine r;
  r.a = (f.a * g.a) & JR_MASK;
  r.c = (f.c * g.a + g.c) & JR_MASK;
  return r;
}
__host__ __device__ inline JrAffine jr_power(JrAffine f, unsigned long long n) {   // f applied n times
  JrAffine r{1ULL, 0ULL};
  while (n) {
    if (n & 1ULL) r = jr_compose(r, f);
    f = jr_compose(f, f);
    n >>= 1;
  }
  return r;
}
__host__ __device__ inline unsigned long long jr_apply(const JrAffine& f, unsigned long long s) { return (s * f.a + f.c) & JR_MASK; }
__host__ __device__ inline unsigned long long jr_jump_dev(unsigned long long s, unsigned long long n) {
  return jr_apply(jr_power(JrAffine{JR_MULT, JR_ADD}, n), s);
}
// the inverse step: x_{i-1} = A^-1 (x_i - C)
inline JrAffine jr_inverse_step() {
  unsigned long long inv = 1;   // Newton: A odd, inv <- inv (2 - A inv) doubles the correct low bits
  for (int i = 0; i < 7; ++i) inv = (inv * (2ULL - JR_MULT * inv)) & JR_MASK;
  JrAffine r;
  r.a = inv;
  r.c = (0ULL - inv * JR_ADD) & JR_MASK;
  return r;
}

// ---- pass A: candidates for a rejection -------------------------------------------------------------------------------
constexpr int JR_SCAN_THREADS = 1024;
constexpr int JR_LANE_CAND = 8;      // candidates a lane keeps in registers (its piece is sized for ~0.5 expected)
struct JrScanArgs {
  unsigned long long s0;             // state in front of raw value 0
  long long scan;                    // raw values to look at
  int per_lane;                      // raw values per lane: a workgroup covers 1024 * per_lane
  unsigned int cand_min;             // raw values >= this can be rejected by SOME bound <= the longest split
  long long cap;                     // entries of cand_i / cand_u
  long long* cand_i;                 // raw index of every candidate; a workgroup's block is contiguous and in raw order
  unsigned int* cand_u;
  unsigned long long* wg_base;       // per workgroup: first entry of its block << 16 | its count (blocks land in arrival order)
  unsigned long long* total;         // [0] entries taken, [1] != 0: a lane or the arrays overflowed (the caller falls back)
};
__global__ void __launch_bounds__(JR_SCAN_THREADS) dsgd_jr_scan_kernel(JrScanArgs a) {
  __shared__ unsigned int cnt[JR_SCAN_THREADS];
  __shared__ unsigned long long base_sh;
  const int tid = threadIdx.x;
  const long long lo = ((long long)blockIdx.x * JR_SCAN_THREADS + tid) * (long long)a.per_lane;
  const long long hi = lo + a.per_lane < a.scan ? lo + a.per_lane : a.scan;
  long long ci[JR_LANE_CAND];
  unsigned int cu[JR_LANE_CAND];
  unsigned int n = 0;
  if (lo < hi) {
    unsigned long long s = jr_jump_dev(a.s0, (unsigned long long)lo);
    for (long long i = lo; i < hi; ++i) {
      s = (s * JR_MULT + JR_ADD) & JR_MASK;
      const unsigned int u = (unsigned int)(s >> 17);
      if (u >= a.cand_min) {
#pragma unroll
        for (int e = 0; e < JR_LANE_CAND; ++e)
          if ((int)n == e) {
            ci[e] = i;
            cu[e] = u;
          }
        ++n;
      }
    }
  }
  if (n > (unsigned int)JR_LANE_CAND) {
    atomicAdd(&a.total[1], 1ULL);
    n = JR_LANE_CAND;
  }
  cnt[tid] = n;
  __syncthreads();
  // exclusive prefix over the workgroup (Hillis-Steele in LDS: 1,024 small counts)
  unsigned int incl = n;
  for (int off = 1; off < JR_SCAN_THREADS; off <<= 1) {
    const unsigned int add = tid >= off ? cnt[tid - off] : 0u;
    __syncthreads();
    incl += add;
    cnt[tid] = incl;
    __syncthreads();
  }
  if (tid == JR_SCAN_THREADS - 1) {
    const unsigned long long b = atomicAdd(&a.total[0], (unsigned long long)incl);
    base_sh = b;
    a.wg_base[blockIdx.x] = (b << 16) | (unsigned long long)incl;   // (<= 8,192 per workgroup)
  }
  __syncthreads();
  const unsigned long long at = base_sh + (incl - n);
  if (at + n > (unsigned long long)a.cap) {
    if (n) atomicAdd(&a.total[1], 1ULL);
    return;
  }
#pragma unroll
  for (int e = 0; e < JR_LANE_CAND; ++e)
    if ((unsigned int)e < n) {
      a.cand_i[at + e] = ci[e];
      a.cand_u[at + e] = cu[e];
    }
}

// ---- pass B: the slice [b, b + take) of one shuffle, traced backwards ---------------------------------------------------
constexpr int JR_SLICE_THREADS = 1024;
constexpr int JR_MAX_TAKE = 1024;            // list entries per (batch, worker): the reference's batch sizes are 100-200
constexpr int JR_MAX_LEN = 1 << 20;          // rows of a split: the bitmap of wanted positions is len / 8 bytes of LDS
constexpr int JR_MAX_REJ = 64;               // rejected raw values inside ONE shuffle (expected len^2 / 2^32: 10 at 214 K rows)
constexpr int JR_MAX_HITS = 2048;
struct JrShuf {            // per (batch, worker), from the host's walk over the candidates
  long long raw0;          // raw index of the shuffle's first raw value
  int rej_begin, rej_end;  // its rejected raw values (offsets from raw0, ascending) in JrSliceArgs::rej
};
struct JrSliceArgs {
  unsigned long long s0;               // state in front of raw value 0
  JrAffine back_block;                 // JR_SLICE_THREADS inverse steps at once
  JrAffine back_one;                   // one inverse step
  const JrShuf* shuf;                  // n_steps * n_splits records, step-major, worker-minor
  const int* rej;
  const long long* split_begin;        // per worker
  const long long* split_end;
  const long long* offsets;            // n_steps * n_splits + 1 prefix offsets of the lists
  int* idx_out;
  int n_splits, batch_size;
  int* err;                            // != 0: a shuffle exceeded a limit above (the caller falls back)
};

// java.util.Random.nextInt(bound) of the ACCEPTED raw value u (31 bits)
__device__ __forceinline__ int jr_next_int_of(unsigned int u, unsigned int bound) {
  if ((bound & (bound - 1u)) == 0u) return (int)(((unsigned long long)bound * (unsigned long long)u) >> 31);
  return (int)(u % bound);
}
// raw offset (from the shuffle's first raw value) of the raw value that serves draw m: the (m + 1)-th accepted one
__device__ __forceinline__ long long jr_raw_of_draw(long long m, const int* rej, int n_rej) {
  long long x = m;
  for (int i = 0; i < n_rej; ++i) {
    if ((long long)rej[i] <= x) ++x;
    else break;
  }
  return x;
}

__global__ void __launch_bounds__(JR_SLICE_THREADS) dsgd_jr_slice_kernel(JrSliceArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned int jr_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long q = blockIdx.x;
  const int k = (int)(q % a.n_splits);
  const long long st = q / a.n_splits;
  const long long len = a.split_end[k] - a.split_begin[k];
  const long long b = st * (long long)a.batch_size;
  const long long o0 = a.offsets[q];
  const int take = (int)(a.offsets[q + 1] - o0);
  if (take <= 0) return;
  const JrShuf sh = a.shuf[q];
  const int n_rej = sh.rej_end - sh.rej_begin;
  if (take > JR_MAX_TAKE || len > JR_MAX_LEN || n_rej > JR_MAX_REJ || b + take > len) {
    if (tid == 0) atomicOr(a.err, 1);
    return;
  }
  // LDS: bitmap of the wanted positions | wanted[take] | karr[1024] (the block's draws) | hits | rejections | counters
  const int bm_words = (int)((len + 31) >> 5);
  unsigned int* bitmap = jr_lds;
  int* wanted = reinterpret_cast<int*>(bitmap + bm_words);
  int* karr = wanted + JR_MAX_TAKE;
  int* hits = karr + JR_SLICE_THREADS;
  int* rej = hits + JR_MAX_HITS;
  int* ctl = rej + JR_MAX_REJ;   // [0] hits of the block
  for (int i = tid; i < bm_words; i += JR_SLICE_THREADS) bitmap[i] = 0u;
  if (tid < n_rej) rej[tid] = a.rej[sh.rej_begin + tid];
  if (tid == 0) ctl[0] = 0;
  __syncthreads();
  const unsigned long long s_first = jr_jump_dev(a.s0, (unsigned long long)sh.raw0);   // state in front of the shuffle's first raw value
  // the draw of step n (n = len .. 2) is draw m = len - n of the shuffle
  auto draw_of_step = [&](long long n) -> int {
    const long long raw = jr_raw_of_draw(len - n, rej, n_rej);
    const unsigned long long s = jr_jump_dev(s_first, (unsigned long long)(raw + 1));
    return jr_next_int_of((unsigned int)(s >> 17), (unsigned int)n);
  };

  // ---- phase 1: the steps whose TOP is an entry of the slice, n = b + 1 .. b + take (position p = n - 1), in order ----
  // step n swaps (p, k_n): the entry that wants position k_n now wants p's old content, entry p wants k_n's old content
  for (int t = tid; t < take; t += JR_SLICE_THREADS) {
    const long long p = b + t;
    karr[t] = p >= 1 ? draw_of_step(p + 1) : 0;
  }
  __syncthreads();
  if (wave == 0) {
    // wave 0 holds the entries in registers: entry t in lane t % 64, slot t / 64
    int w[JR_MAX_TAKE / 64];
#pragma unroll
    for (int e = 0; e < JR_MAX_TAKE / 64; ++e) w[e] = -1;
    for (int t = 0; t < take; ++t) {
      const long long p = b + t;
      if (p == 0) {   // position 0 is never on top: it wants itself from the start
        if (lane == 0) w[0] = 0;
        continue;
      }
      const int kk = karr[t];   // (uniform)
#pragma unroll
      for (int e = 0; e < JR_MAX_TAKE / 64; ++e)
        if (w[e] == kk) w[e] = (int)p;
      const int e_t = t >> 6;
#pragma unroll
      for (int e = 0; e < JR_MAX_TAKE / 64; ++e)
        if (e == e_t && lane == (t & 63)) w[e] = kk;
    }
#pragma unroll
    for (int e = 0; e < JR_MAX_TAKE / 64; ++e) {
      const int t = e * 64 + lane;
      if (t < take) {
        wanted[t] = w[e];
        atomicOr(&bitmap[(unsigned int)w[e] >> 5], 1u << (w[e] & 31));
      }
    }
  }
  __syncthreads();

  // ---- phase 2: the steps above the slice, n = b + take + 1 .. len, 1,024 at a time in increasing order ----
  const long long n_first = b + take + 1;
  if (n_first <= len) {
    long long n_mine = n_first + tid;                          // this lane's step of the current block
    long long raw_mine = 0;
    unsigned long long s_mine = 0;
    bool have = false;
    // Rejection i happened while draw T_i = rej[i] - i was being served (T ascending): draw m is served by raw value
    // m + #{T_i <= m}.  The blocks walk the draws DOWNWARDS, so the count for the block's highest draw only ever falls
    // (a uniform pointer), and a lane differs from it only if a rejection lies INSIDE the block (10 per 214 K draws).
    int cnt_hi = n_rej;
    for (long long n0 = n_first; n0 <= len; n0 += JR_SLICE_THREADS, n_mine += JR_SLICE_THREADS) {
      int kk = -1;
      const long long m_hi = len - n0;
      while (cnt_hi > 0 && (long long)(rej[cnt_hi - 1] - (cnt_hi - 1)) > m_hi) --cnt_hi;   // (uniform)
      if (n_mine <= len) {
        const long long m = len - n_mine;
        int cnt = cnt_hi;
        while (cnt > 0 && (long long)(rej[cnt - 1] - (cnt - 1)) > m) --cnt;                // (rarely one iteration)
        const long long raw = m + cnt;
        if (!have) {   // (first block of this lane -- or a lane that was beyond the end never comes back)
          s_mine = jr_jump_dev(s_first, (unsigned long long)(raw + 1));
          have = true;
        } else {       // the next block's step is 1,024 draws EARLIER in the stream, plus whatever rejections lie between
          s_mine = jr_apply(a.back_block, s_mine);
          for (long long d = (raw_mine - JR_SLICE_THREADS) - raw; d > 0; --d) s_mine = jr_apply(a.back_one, s_mine);
        }
        raw_mine = raw;
        kk = jr_next_int_of((unsigned int)(s_mine >> 17), (unsigned int)n_mine);
        if ((bitmap[(unsigned int)kk >> 5] >> (kk & 31)) & 1u) {
          const int h = atomicAdd(&ctl[0], 1);
          if (h < JR_MAX_HITS) hits[h] = tid;
        }
      }
      karr[tid] = kk;
      __syncthreads();
      const int n_hits = ctl[0];
      if (n_hits > 0) {
        if (n_hits > JR_MAX_HITS) {
          if (tid == 0) atomicOr(a.err, 1);
          return;   // (uniform)
        }
        if (wave == 0) {
          // the hits in step order: repeatedly the smallest unprocessed lane index (a hit may create later ones)
          int done_upto = -1;
          for (;;) {
            int best = 0x7fffffff;
            const int nh = ctl[0] < JR_MAX_HITS ? ctl[0] : JR_MAX_HITS;
            for (int i = lane; i < nh; i += 64) {
              const int h = hits[i];
              if (h > done_upto && h < best) best = h;
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
            if (best == 0x7fffffff) break;
            done_upto = best;
            const int kq = karr[best];
            // which entry wants position kq?  (none: a stale hit -- the entry moved on inside this block)
            int found = -1;
            for (int t = lane; t < take; t += 64)
              if (wanted[t] == kq) found = t;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) found = max(found, __shfl_xor(found, o, 64));
            if (found < 0) continue;
            const int top = (int)(n0 + best - 1);   // step n0 + best swaps (n - 1, kq)
            if (lane == 0) {
              wanted[found] = top;
              atomicAnd(&bitmap[(unsigned int)kq >> 5], ~(1u << (kq & 31)));
              atomicOr(&bitmap[(unsigned int)top >> 5], 1u << (top & 31));
            }
            // later steps of THIS block that target the new position were tested against the old bitmap
            for (int j = best + 1 + lane; j < JR_SLICE_THREADS; j += 64)
              if (karr[j] == top) {
                const int h = atomicAdd(&ctl[0], 1);
                if (h < JR_MAX_HITS) hits[h] = j;
              }
            __builtin_amdgcn_wave_barrier();
          }
          if (lane == 0) ctl[0] = 0;
        }
      }
      __syncthreads();
    }
  }
  // ---- what the wanted positions held in the identity: the split's own row numbers ----
  for (int t = tid; t < take; t += JR_SLICE_THREADS) a.idx_out[o0 + t] = (int)(a.split_begin[k] + wanted[t]);
}
__host__ __device__ constexpr size_t jr_slice_lds_bytes(long long len) {
  return sizeof(unsigned int) * (size_t)(((len + 31) >> 5) + JR_MAX_TAKE + JR_SLICE_THREADS + JR_MAX_HITS + JR_MAX_REJ + 4);
}
