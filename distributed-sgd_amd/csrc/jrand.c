/* libdsgd_host: the reference's random stream, natively -- HOST-side helper of the mirrors (distributed-sgd_amd/host.py,
 * include/dsgd.hpp); the JVM side of a patched reference draws from its own scala.util.Random.
 *
 * ref: core/Master.scala:184 -- for EVERY batch of an epoch every worker's whole split is reshuffled,
 *          workers.zip(split.map(Random.shuffle(_))).map { case (worker, idx) => idx.slice(batch, batch + batchSize) }
 *      scala.util.Random.shuffle (2.12): an ArrayBuffer copy, then for (n <- len to 2 by -1) swap(n - 1, nextInt(n));
 *      scala.util.Random delegates to java.util.Random, seeded 0 at Main.scala:32;
 *      java.util.Random: 48-bit LCG  seed' = seed * 0x5DEECE66D + 0xB (mod 2^48), next(31) = seed' >> 17,
 *      nextInt(bound): power of two -> (bound * next(31)) >> 31, else r = next(31) mod bound with the rejection loop
 *      `u - r + (bound - 1) < 0` in 32-bit arithmetic.
 *
 * That is len - 1 draws per worker and batch -- 1.15 M draws per epoch of the full=false configuration (62 batches x 3
 * splits of 6,173 rows), 1.38 G per epoch at full=true: O(N) master work per BATCH in the reference.  A Python loop does
 * 2 M draws per second; a resident plan runs the epoch's 62 steps in 0.3 ms.  Here: one thread draws 0.5-1 G per second,
 * and the shuffles of an epoch are drawn IN PARALLEL, exactly:
 *   pass A  the raw stream is cut into chunks (an LCG jumps ahead in O(log n)); every chunk lists its CANDIDATES for a
 *           rejection -- raw values so large that SOME bound up to the longest split would reject them (one in ~10^4-10^6);
 *   resolve the candidates are walked in order, sequentially: with the draws consumed so far known, each one's bound is
 *           known, so whether it really is rejected -- this fixes the raw index at which every shuffle starts;
 *   pass B  every shuffle runs from its own start state, independently (its own rejections handled as java does).
 * The result is the reference's stream draw for draw (tests/test_host_mirror.py: against the pure-Python restatement).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <stdatomic.h>
#include <unistd.h>

#define JR_MULT 0x5DEECE66DULL
#define JR_ADD 0xBULL
#define JR_MASK ((1ULL << 48) - 1)

static inline uint64_t jr_step(uint64_t s) { return (s * JR_MULT + JR_ADD) & JR_MASK; }

/* state after n steps: s -> a^n s + c (a^n - 1) / (a - 1), by doubling on the affine map */
static uint64_t jr_jump(uint64_t s, uint64_t n) {
  uint64_t am = JR_MULT, ac = JR_ADD; /* the map applied 2^i times */
  uint64_t rm = 1, rc = 0;            /* the accumulated map */
  while (n) {
    if (n & 1) {
      rm = (rm * am) & JR_MASK;
      rc = (rc * am + ac) & JR_MASK;
    }
    ac = (ac * am + ac) & JR_MASK;
    am = (am * am) & JR_MASK;
    n >>= 1;
  }
  return (s * rm + rc) & JR_MASK;
}

/* java.util.Random.nextInt(bound) on *state; *extra counts the rejected raw values (beyond the first) */
static inline int32_t jr_next_int(uint64_t* state, int32_t bound, int64_t* extra) {
  uint64_t s = jr_step(*state);
  int32_t r = (int32_t)(s >> 17);
  const int32_t m = bound - 1;
  if ((bound & m) == 0) {
    *state = s;
    return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
  }
  for (;;) {
    const int32_t u = r;
    r = u % bound;
    /* `u - r + m < 0` in Java's wrapping int arithmetic */
    if ((int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m) >= 0) break;
    s = jr_step(s);
    r = (int32_t)(s >> 17);
    ++*extra;
  }
  *state = s;
  return r;
}

/* ---- a small persistent thread pool (mutex + condition variable: the threads SLEEP between calls).  OpenMP's default
 * is to spin at the end of a parallel region; inside a CPU-quota'd container the spinning threads burn the quota and the
 * whole process is throttled for the rest of the scheduler period -- measured: 100 ms for a 0.8 ms job. ---- */
#define JR_MAX_THREADS 64
typedef void (*jr_task_fn)(void* arg, int64_t task);
static struct {
  pthread_mutex_t mu;
  pthread_cond_t wake, done;
  pthread_t th[JR_MAX_THREADS];
  int n_threads;          /* helper threads started */
  jr_task_fn fn;
  void* arg;
  int64_t n_tasks;
  atomic_llong next;
  int active;             /* helpers still inside the current job */
  int want;               /* helpers the current job wants */
  uint64_t gen;           /* job number */
  pthread_mutex_t call_mu;   /* one job at a time */
} jr_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0, 0, 0, 0, 0, 0, PTHREAD_MUTEX_INITIALIZER};

static void* jr_worker(void* p) {
  const int me = (int)(intptr_t)p;
  uint64_t seen = 0;
  pthread_mutex_lock(&jr_pool.mu);
  for (;;) {
    while (jr_pool.gen == seen || me >= jr_pool.want) {
      if (jr_pool.gen != seen) seen = jr_pool.gen;   /* a job that does not want this helper */
      pthread_cond_wait(&jr_pool.wake, &jr_pool.mu);
    }
    seen = jr_pool.gen;
    jr_task_fn fn = jr_pool.fn;
    void* arg = jr_pool.arg;
    const int64_t n = jr_pool.n_tasks;
    pthread_mutex_unlock(&jr_pool.mu);
    for (;;) {
      const int64_t t = atomic_fetch_add(&jr_pool.next, 1);
      if (t >= n) break;
      fn(arg, t);
    }
    pthread_mutex_lock(&jr_pool.mu);
    if (--jr_pool.active == 0) pthread_cond_signal(&jr_pool.done);
  }
  return 0;
}

/* run tasks 0 .. n_tasks-1 of fn on up to `threads` threads (the caller is one of them) */
static void jr_run(jr_task_fn fn, void* arg, int64_t n_tasks, int threads) {
  if (threads > JR_MAX_THREADS) threads = JR_MAX_THREADS;
  if (threads > n_tasks) threads = (int)n_tasks;
  if (threads <= 1) {
    for (int64_t t = 0; t < n_tasks; ++t) fn(arg, t);
    return;
  }
  pthread_mutex_lock(&jr_pool.call_mu);
  pthread_mutex_lock(&jr_pool.mu);
  while (jr_pool.n_threads < threads - 1) {
    if (pthread_create(&jr_pool.th[jr_pool.n_threads], 0, jr_worker, (void*)(intptr_t)jr_pool.n_threads) != 0) break;
    pthread_detach(jr_pool.th[jr_pool.n_threads]);
    ++jr_pool.n_threads;
  }
  const int helpers = jr_pool.n_threads < threads - 1 ? jr_pool.n_threads : threads - 1;
  jr_pool.fn = fn;
  jr_pool.arg = arg;
  jr_pool.n_tasks = n_tasks;
  atomic_store(&jr_pool.next, 0);
  jr_pool.active = helpers;
  jr_pool.want = helpers;
  ++jr_pool.gen;
  pthread_cond_broadcast(&jr_pool.wake);
  pthread_mutex_unlock(&jr_pool.mu);
  for (;;) {
    const int64_t t = atomic_fetch_add(&jr_pool.next, 1);
    if (t >= n_tasks) break;
    fn(arg, t);
  }
  pthread_mutex_lock(&jr_pool.mu);
  while (jr_pool.active > 0) pthread_cond_wait(&jr_pool.done, &jr_pool.mu);
  jr_pool.want = 0;
  pthread_mutex_unlock(&jr_pool.mu);
  pthread_mutex_unlock(&jr_pool.call_mu);
}

/* threads of the parallel passes: DSGD_HOST_THREADS, else the online CPUs capped at 32; fewer for short streams (one
 * thread per 32 K draws: waking a thread costs more than it then draws) */
static int jr_cap(void) {
  long cap = sysconf(_SC_NPROCESSORS_ONLN);
  if (cap < 1) cap = 1;
  if (cap > 32) cap = 32;
  const char* e = getenv("DSGD_HOST_THREADS");
  if (e && atoi(e) > 0) cap = atoi(e);
  if (cap > JR_MAX_THREADS) cap = JR_MAX_THREADS;
  return (int)cap;
}
static int jr_threads(int64_t draws) {
  const int cap = jr_cap();
  int64_t want = draws / 32768;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

int dsgd_host_abi_version(void) { return 1; }
int dsgd_host_threads(void) { return jr_cap(); }

/* scala.util.Random.setSeed / new java.util.Random(seed): the scrambled initial state */
uint64_t dsgd_jrand_seed(int64_t seed) { return ((uint64_t)seed ^ JR_MULT) & JR_MASK; }

/* n draws of nextInt(bound) (tests) */
void dsgd_jrand_next_ints(uint64_t* state, int32_t bound, int64_t n, int32_t* out) {
  int64_t extra = 0;
  for (int64_t i = 0; i < n; ++i) out[i] = jr_next_int(state, bound, &extra);
}

/* One shuffle (scala.util.Random.shuffle of xs[0 .. len)), in place on buf; returns the raw values consumed. */
static int64_t shuffle_one(uint64_t* state, int32_t* buf, int64_t len) {
  int64_t extra = 0;
  for (int64_t n = len; n >= 2; --n) {
    const int32_t k = jr_next_int(state, (int32_t)n, &extra);
    const int32_t t = buf[n - 1];
    buf[n - 1] = buf[k];
    buf[k] = t;
  }
  return (len >= 2 ? len - 1 : 0) + extra;
}

struct jr_scan_job {
  uint64_t s0;
  int64_t scan, chunk;
  uint32_t cand_min;
  int64_t** loc_i;
  uint32_t** loc_u;
  int64_t* loc_n;
  atomic_int* bad;
};
static void jr_scan_task(void* p, int64_t t) {
  struct jr_scan_job* J = (struct jr_scan_job*)p;
  const int64_t lo = t * J->chunk, hi = lo + J->chunk < J->scan ? lo + J->chunk : J->scan;
  if (lo >= hi) return;
  int64_t cap = 256, n = 0;
  int64_t* ci = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
  uint32_t* cu = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
  uint64_t s = jr_jump(J->s0, (uint64_t)lo);
  for (int64_t i = lo; i < hi && ci && cu; ++i) {
    s = jr_step(s);
    const uint32_t u = (uint32_t)(s >> 17);
    if (u >= J->cand_min) {
      if (n == cap) {
        cap *= 2;
        int64_t* ci2 = (int64_t*)realloc(ci, sizeof(int64_t) * (size_t)cap);
        uint32_t* cu2 = (uint32_t*)realloc(cu, sizeof(uint32_t) * (size_t)cap);
        if (!ci2 || !cu2) {
          free(ci2 ? ci2 : ci);
          free(cu2 ? cu2 : cu);
          ci = NULL;
          cu = NULL;
          break;
        }
        ci = ci2;
        cu = cu2;
      }
      ci[n] = i;
      cu[n] = u;
      ++n;
    }
  }
  if (!ci || !cu) {
    atomic_store(J->bad, 1);
    free(ci);
    free(cu);
    ci = NULL;
    cu = NULL;
    n = 0;
  }
  J->loc_i[t] = ci;
  J->loc_u[t] = cu;
  J->loc_n[t] = n;
}

struct jr_shuf_job {
  uint64_t s0;
  const int64_t *split_begin, *split_end;
  int32_t n_splits, batch_size;
  const int64_t *start, *shift, *offsets;
  int32_t* idx_out;
  atomic_int* err;
  atomic_llong* extra_sum;
};
static int64_t shuffle_one(uint64_t* state, int32_t* buf, int64_t len);
static void jr_shuf_task(void* p, int64_t q) {
  struct jr_shuf_job* J = (struct jr_shuf_job*)p;
  const int k = (int)(q % J->n_splits);
  const int64_t st = q / J->n_splits, len = J->split_end[k] - J->split_begin[k], b = st * J->batch_size;
  int32_t* buf = (int32_t*)malloc(sizeof(int32_t) * (size_t)len);
  if (!buf) {
    atomic_store(J->err, 1);
    return;
  }
  for (int64_t i = 0; i < len; ++i) buf[i] = (int32_t)(J->split_begin[k] + i);
  uint64_t s = jr_jump(J->s0, (uint64_t)(J->start[q] + J->shift[q]));
  const int64_t used = shuffle_one(&s, buf, len);
  atomic_fetch_add(J->extra_sum, (long long)(used - (len >= 2 ? len - 1 : 0)));
  const int64_t take = J->offsets[q + 1] - J->offsets[q];
  memcpy(J->idx_out + J->offsets[q], buf + b, sizeof(int32_t) * (size_t)take);
  free(buf);
}

/* The index lists of ONE epoch of Master.fit (core/Master.scala:179-199):
 *   split k = rows [split_begin[k], split_end[k])          (SplitStrategy.vanilla, the caller's)
 *   for batch b = 0, batch_size, ... < max_samples:  for k: idx = shuffle(split k).slice(b, b + batch_size)
 * idx_out receives the lists step-major, worker-minor (what dsgd_plan_create takes); offsets_out the n_steps * n_splits + 1
 * prefix offsets.  A slice past the end of a short split is EMPTY (the reference's Vec.sum would then throw in the slave):
 * steps are emitted up to the first one with an empty list, *n_steps_out says how many.  *state advances exactly as the
 * JVM's generator would over the steps emitted.  Returns 0, or -1 on bad arguments / out of memory.
 * Capacity: idx_out n_steps_max * sum(min(batch, len_k)), offsets_out n_steps_max * n_splits + 1. */
int dsgd_jrand_epoch_lists(uint64_t* state, const int64_t* split_begin, const int64_t* split_end, int32_t n_splits,
                           int64_t max_samples, int32_t batch_size, int32_t* idx_out, int64_t* offsets_out,
                           int64_t* n_steps_out, int64_t* draws_out) {
  if (!state || !split_begin || !split_end || n_splits < 1 || batch_size < 1 || !idx_out || !offsets_out || !n_steps_out) return -1;
  int64_t max_len = 0;
  for (int k = 0; k < n_splits; ++k) {
    const int64_t len = split_end[k] - split_begin[k];
    if (len < 1 || split_end[k] > 0x7fffffffLL) return -1;
    if (len > max_len) max_len = len;
  }
  /* steps the reference runs before an empty slice: batch b needs b < len_k for every k */
  int64_t n_steps = 0;
  for (int64_t b = 0; b < max_samples; b += batch_size) {
    int ok = 1;
    for (int k = 0; k < n_splits; ++k)
      if (b >= split_end[k] - split_begin[k]) ok = 0;
    if (!ok) break;
    ++n_steps;
  }
  *n_steps_out = n_steps;
  offsets_out[0] = 0;
  {
    int64_t off = 0, i = 0;
    for (int64_t s = 0; s < n_steps; ++s)
      for (int k = 0; k < n_splits; ++k) {
        const int64_t len = split_end[k] - split_begin[k], b = s * batch_size;
        const int64_t take = len - b < batch_size ? len - b : batch_size;
        off += take;
        offsets_out[++i] = off;
      }
  }
  const int64_t n_shuf = n_steps * n_splits;
  if (n_shuf == 0) {
    if (draws_out) *draws_out = 0;
    return 0;
  }
  /* nominal raw index of every shuffle's start (no rejections) */
  int64_t* start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_shuf + 1));
  if (!start) return -1;
  start[0] = 0;
  for (int64_t j = 0; j < n_shuf; ++j) {
    const int64_t len = split_end[j % n_splits] - split_begin[j % n_splits];
    start[j + 1] = start[j] + (len >= 2 ? len - 1 : 0);
  }
  const int64_t nominal = start[n_shuf];
  const uint64_t s0 = *state;
  {
    /* Short epochs: a rejection is expected once in a thousand epochs (each draw rejects with probability bound / 2^31).
     * Every shuffle is drawn at once from its NOMINAL start; if none of them saw a rejection the nominal starts were the
     * true ones and the stream is exact -- one parallel pass instead of two.  Otherwise: the exact two-pass form below. */
    double expect = 0.0;
    for (int k = 0; k < n_splits; ++k) {
      const double len = (double)(split_end[k] - split_begin[k]);
      expect += (double)n_steps * len * len / 4294967296.0;
    }
    const char* force = getenv("DSGD_HOST_SPECULATE");   /* (tests: 1 = try the one-pass form whatever the odds, 0 = never) */
    if (force ? atoi(force) != 0 : expect < 0.05) {
      int64_t* zero = (int64_t*)calloc((size_t)(n_shuf + 1), sizeof(int64_t));
      if (zero) {
        atomic_int err0 = 0;
        atomic_llong extra0 = 0;
        struct jr_shuf_job sj0 = {s0, split_begin, split_end, n_splits, batch_size, start, zero, offsets_out, idx_out, &err0, &extra0};
        jr_run(jr_shuf_task, &sj0, n_shuf, jr_threads(nominal));
        free(zero);
        if (!atomic_load(&err0) && atomic_load(&extra0) == 0) {
          *state = jr_jump(s0, (uint64_t)nominal);
          if (draws_out) *draws_out = nominal;
          free(start);
          return 0;
        }
      }
    }
  }
  /* ---- pass A: candidates for a rejection.  u is rejected for bound n iff u >= floor(2^31 / n) * n, never for a power of
   * two; floor(2^31 / n) * n > 2^31 - n >= 2^31 - max_len: only raw values above that can be rejected at all.  The stream
   * is scanned a little past its nominal end (every true rejection lengthens it by one). ---- */
  const uint32_t cand_min = (uint32_t)(0x80000000ULL - (uint64_t)max_len);
  int64_t scan = nominal + 64 + nominal / 4096;
  int64_t n_cand = 0, cap_cand = 1024;
  int64_t* cand_i = NULL;
  uint32_t* cand_u = NULL;
  int rc = 0;
  for (;;) {
    const int nt = jr_threads(scan);
    const int64_t chunk = (scan + nt - 1) / nt;
    int64_t** loc_i = (int64_t**)calloc((size_t)nt, sizeof(int64_t*));
    uint32_t** loc_u = (uint32_t**)calloc((size_t)nt, sizeof(uint32_t*));
    int64_t* loc_n = (int64_t*)calloc((size_t)nt, sizeof(int64_t));
    if (!loc_i || !loc_u || !loc_n) {
      rc = -1;
      free(loc_i); free(loc_u); free(loc_n);
      break;
    }
    atomic_int bad = 0;
    struct jr_scan_job job = {s0, scan, chunk, cand_min, loc_i, loc_u, loc_n, &bad};
    jr_run(jr_scan_task, &job, nt, nt);
    n_cand = 0;
    for (int t = 0; t < nt; ++t) n_cand += loc_n[t];
    cap_cand = n_cand + 1;
    cand_i = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap_cand);
    cand_u = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap_cand);
    if (atomic_load(&bad) || !cand_i || !cand_u) rc = -1;
    int64_t at = 0;
    for (int t = 0; t < nt; ++t) {
      if (rc == 0 && loc_n[t]) {
        memcpy(cand_i + at, loc_i[t], sizeof(int64_t) * (size_t)loc_n[t]);
        memcpy(cand_u + at, loc_u[t], sizeof(uint32_t) * (size_t)loc_n[t]);
        at += loc_n[t];
      }
      free(loc_i[t]);
      free(loc_u[t]);
    }
    free(loc_i); free(loc_u); free(loc_n);
    if (rc) break;
    /* ---- resolve: walk the candidates in raw order; `rej` = rejections so far.  Raw index i serves draw d = i - rej of
     * the epoch; draw d belongs to shuffle j (start[j] <= d < start[j + 1]) with bound len_j - (d - start[j]). ---- */
    int64_t rej = 0, j = 0;
    int64_t* shift = (int64_t*)calloc((size_t)(n_shuf + 1), sizeof(int64_t)); /* rejections BEFORE shuffle j starts */
    if (!shift) {
      rc = -1;
      break;
    }
    int64_t last_j = 0;
    for (int64_t c = 0; c < n_cand; ++c) {
      const int64_t d = cand_i[c] - rej;
      if (d >= nominal) break;
      while (start[j + 1] <= d) ++j;
      for (; last_j < j; ++last_j) shift[last_j + 1] = rej; /* shuffles that begin before this candidate's draw */
      const int64_t len = split_end[j % n_splits] - split_begin[j % n_splits];
      const uint32_t n = (uint32_t)(len - (d - start[j]));
      if ((n & (n - 1)) != 0 && cand_u[c] >= (0x80000000u / n) * n) ++rej;
    }
    for (; last_j < n_shuf; ++last_j) shift[last_j + 1] = rej;
    free(cand_i);
    free(cand_u);
    cand_i = NULL;
    cand_u = NULL;
    if (nominal + rej > scan) { /* more rejections than the margin scanned: scan further and resolve again */
      free(shift);
      scan = nominal + rej + 64 + rej / 8;
      continue;
    }
    /* ---- pass B: every shuffle from its own start state ---- */
    atomic_int err = 0;
    atomic_llong extra_sum = 0;
    struct jr_shuf_job sj = {s0, split_begin, split_end, n_splits, batch_size, start, shift, offsets_out, idx_out, &err, &extra_sum};
    jr_run(jr_shuf_task, &sj, n_shuf, jr_threads(nominal));
    const int64_t total_extra = (int64_t)atomic_load(&extra_sum);
    if (atomic_load(&err)) rc = -1;
    else if (total_extra != rej) rc = -2; /* (cannot happen: the two passes count the same rejections) */
    else {
      *state = jr_jump(s0, (uint64_t)(nominal + rej));
      if (draws_out) *draws_out = nominal + rej;
    }
    free(shift);
    break;
  }
  free(cand_i);
  free(cand_u);
  free(start);
  return rc;
}

/* scala.util.Random.shuffle of buf[0 .. len) in place, sequentially (the reference form the parallel one is tested against);
 * returns the raw values consumed */
int64_t dsgd_jrand_shuffle(uint64_t* state, int32_t* buf, int64_t len) { return shuffle_one(state, buf, len); }
