/*
 * ORACLE -- test infrastructure, NOT product code.
 *
 * CPU (fp64) restatement of the reference's algorithm for the hot path:
 *   per-worker sparse-SVM gradient step, the synchronous master's mean+update, the
 *   asynchronous ("Hogwild") update, prediction and loss/accuracy evaluation.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (distributed-sgd_amd/csrc) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" -- the JVM reference cannot run here (no java/scala/sbt)
 * and its own tests only pin +, dot, *scalar, norm, sparsity and the /0 error
 * (src/test/scala/epfl/distributed/data/VecTests.scala:14-40), which tests/test_oracle_golden.py
 * checks.  Everything else is restated from the cited lines and cross-checked against the
 * independent dict-based restatement in ref_dict.py.
 *
 * Citations are relative to /root/reference/src/main/scala/epfl/distributed/.
 *
 * Representation: a reference Sparse vector (Map[Int,Number] with default 0, keys may be
 * 0..size -- math/Sparse.scala:61-68) is held as a dense double[dim+1] array indexed by key;
 * "key absent" == 0.0.  The constructor filter abs(v) > 1e-20 (math/Sparse.scala:108-118) is
 * applied wherever the reference constructs a Sparse (after every element-wise op).
 * Summation order: ascending key (the reference uses HashMap trie order; difference is
 * O(1e-16) relative, see ref_dict.py header).
 *
 * Two flavours:
 *   orc_*      "fast"    dense fp64 accumulators over CSR (numerical ground truth; the *_omp
 *                        variants use all host cores and are the "best-effort CPU" baseline)
 *   orc_lit_*  "literal" per-sample sparse-vector ops (sorted key/value arrays rebuilt on
 *                        every +, like Sparse.elementWiseOp) -- mirrors the reference's
 *                        algorithmic cost; this is what is timed as "reference CPU path".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_EPS 1e-20 /* math/Sparse.scala:104 */

typedef struct {
  int64_t n_rows;
  int32_t dim; /* D = 47236 for RCV1 (utils/Dataset.scala:16); keys are 1..D, arrays have D+1 slots */
  const int64_t* row_ptr;
  const int32_t* col; /* 1-based feature ids, ascending within a row */
  const float* val;
  const int8_t* label; /* +1 / -1 */
} orc_csr;

static inline double filt(double v) { return fabs(v) > ORC_EPS ? v : 0.0; }

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* x . w -- math/Vec.scala:58 -> Sparse.* (math/Sparse.scala:46, :20-31) -> Vec.sum (:53).
 * Products with abs <= 1e-20 are dropped by the Sparse constructor before the sum. */
double orc_row_dot(const orc_csr* m, int64_t i, const double* w) {
  double acc = 0.0;
  for (int64_t p = m->row_ptr[i]; p < m->row_ptr[i + 1]; ++p) acc = acc + filt((double)m->val[p] * w[m->col[p]]);
  return acc;
}

/* w . ds  (SparseSVM.scala:31) */
double orc_dense_dot(const double* a, const double* b, int32_t dim) {
  double acc = 0.0;
  for (int32_t j = 0; j <= dim; ++j) acc = acc + filt(a[j] * b[j]);
  return acc;
}

/* Main.scala:54-65: ds[i] = 1/(count(feature i+1)+1) for 0-based i, only where count != 0. */
void orc_dim_sparsity(const orc_csr* m, int64_t n_train, double* ds /* dim+1 */) {
  int32_t D = m->dim;
  double* buff = (double*)calloc((size_t)D + 1, sizeof(double));
  /* v.map.keys: the Sparse constructor has already dropped abs(value) <= 1e-20 (math/Sparse.scala:108-118) */
  for (int64_t i = 0; i < n_train; ++i)
    for (int64_t p = m->row_ptr[i]; p < m->row_ptr[i + 1]; ++p)
      if (fabs((double)m->val[p]) > ORC_EPS) buff[m->col[p] - 1] += 1.0;
  for (int32_t i = 0; i <= D; ++i) ds[i] = 0.0;
  for (int32_t i = 0; i < D; ++i)
    if (buff[i] != 0.0) ds[i] = filt(1.0 / (buff[i] + 1.0));
  free(buff);
}

/* SparseSVM.regularize (SparseSVM.scala:31) in place: g + g.valueLike(2*lambda*(w.ds)).
 * valueLike (Vec.scala:65-75): zero value -> zeros; else the SUPPORT of g filled with value
 * (and a value with abs <= 1e-20 is filtered away by the Sparse constructor). */
static void regularize_inplace(double* g, const double* w, const double* ds, double lambda, int32_t dim) {
  double s = lambda * 2.0 * orc_dense_dot(w, ds, dim);
  if (s == 0.0 || !(fabs(s) > ORC_EPS)) return;
  for (int32_t j = 0; j <= dim; ++j)
    if (g[j] != 0.0) g[j] = filt(g[j] + s);
}

typedef struct {
  int64_t n_active;      /* rows with y*(x.w) >= 0 */
  int64_t n_exact_zero;  /* rows with x.w == 0 exactly */
  double min_abs_margin; /* min |x.w| over rows with x.w != 0 -- gate-flip exposure for fp32 engines */
} orc_gate_stats;

static void stats_init(orc_gate_stats* st) {
  if (!st) return;
  st->n_active = 0;
  st->n_exact_zero = 0;
  st->min_abs_margin = INFINITY;
}

/* Slave.gradient body without regularize: g = sum_i backward(w, x_i, y_i)
 * (core/Slave.scala:147-153, SparseSVM.scala:26-29, Vec.scala:128-131). */
static int grad_sum(const orc_csr* m, const double* w, const int32_t* idx, int64_t n, double* g, orc_gate_stats* st) {
  if (n <= 0) return -1; /* Vec.scala:129 require(vecs.nonEmpty) */
  memset(g, 0, ((size_t)m->dim + 1) * sizeof(double));
  for (int64_t t = 0; t < n; ++t) {
    int64_t i = idx[t];
    if (i < 0 || i >= m->n_rows) return -2;
    double d = orc_row_dot(m, i, w);
    double y = (double)m->label[i];
    double activity = y * d;
    if (st) {
      if (d == 0.0) st->n_exact_zero++;
      else if (fabs(d) < st->min_abs_margin) st->min_abs_margin = fabs(d);
    }
    if (activity < 0) continue; /* zerosLike: adding zeros leaves the accumulator unchanged */
    if (st) st->n_active++;
    for (int64_t p = m->row_ptr[i]; p < m->row_ptr[i + 1]; ++p) {
      int32_t c = m->col[p];
      double xv = filt((double)m->val[p] * y); /* x * y, mapValues + constructor filter */
      g[c] = filt(g[c] + xv);                    /* Sparse.scala:33 union add + filter */
    }
  }
  return 0;
}

/* core/Slave.scala:142-157 */
int orc_gradient(const orc_csr* m, const double* w, const double* ds, double lambda, const int32_t* idx, int64_t n,
                 double* g_out, orc_gate_stats* st) {
  stats_init(st);
  int rc = grad_sum(m, w, idx, n, g_out, st);
  if (rc) return rc;
  regularize_inplace(g_out, w, ds, lambda, m->dim);
  return 0;
}

/* core/Master.scala:186-197: mean over WORKERS of per-worker regularised sums; w - lr*grad.
 * Workers whose slice is empty make Vec.sum throw in the reference -> error here too. */
int orc_sync_step(const orc_csr* m, double* w, const double* ds, double lambda, const int32_t* const* idx_per_worker,
                  const int64_t* n_per_worker, int32_t n_workers, double lr, orc_gate_stats* st) {
  int32_t D = m->dim;
  if (n_workers <= 0) return -1;
  double* acc = (double*)calloc((size_t)D + 1, sizeof(double));
  double* g = (double*)malloc(((size_t)D + 1) * sizeof(double));
  orc_gate_stats total, one;
  stats_init(&total);
  int rc = 0;
  for (int32_t k = 0; k < n_workers && !rc; ++k) {
    rc = orc_gradient(m, w, ds, lambda, idx_per_worker[k], n_per_worker[k], g, &one);
    if (rc) break;
    total.n_active += one.n_active;
    total.n_exact_zero += one.n_exact_zero;
    if (one.min_abs_margin < total.min_abs_margin) total.min_abs_margin = one.min_abs_margin;
    for (int32_t j = 0; j <= D; ++j) acc[j] = filt(acc[j] + g[j]); /* Vec.sum over workers */
  }
  if (!rc) {
    double K = (double)n_workers;
    for (int32_t j = 0; j <= D; ++j) {
      double mean = filt(acc[j] / K);  /* Vec.mean: sum / size (Vec.scala:139) */
      double upd = filt(mean * lr);    /* learningRate * grad (Vec.scala:42) */
      w[j] = filt(w[j] - upd);         /* batchWeights - ... (Master.scala:197) */
    }
  }
  if (st) *st = total;
  free(acc);
  free(g);
  return rc;
}

/* core/Slave.scala:92-101: grad = MEAN over samples; gradUpdate = lr * regularize(grad, w); w -= gradUpdate */
int orc_async_step(const orc_csr* m, double* w, const double* ds, double lambda, const int32_t* idx, int64_t n,
                   double lr, double* delta_out /* may be NULL */, orc_gate_stats* st) {
  int32_t D = m->dim;
  double* g = (double*)malloc(((size_t)D + 1) * sizeof(double));
  stats_init(st);
  int rc = grad_sum(m, w, idx, n, g, st);
  if (!rc) {
    double nn = (double)n;
    for (int32_t j = 0; j <= D; ++j) g[j] = filt(g[j] / nn); /* Vec.mean */
    regularize_inplace(g, w, ds, lambda, D);
    for (int32_t j = 0; j <= D; ++j) {
      double upd = filt(g[j] * lr);
      if (delta_out) delta_out[j] = upd;
      w[j] = filt(w[j] - upd); /* Slave.scala:101 / :180 / GradState.scala:8 */
    }
  }
  free(g);
  return rc;
}

/* core/Slave.scala:129-140 + SparseSVM.scala:14: p = -signum(x.w) */
int orc_forward(const orc_csr* m, const double* w, const int32_t* idx, int64_t n, double* pred) {
  for (int64_t t = 0; t < n; ++t) {
    int64_t i = idx[t];
    if (i < 0 || i >= m->n_rows) return -2;
    double d = orc_row_dot(m, i, w);
    pred[t] = (d > 0) ? -1.0 : (d < 0 ? 1.0 : 0.0);
  }
  return 0;
}

/* core/Master.scala:100-107 + SparseSVM.scala:16-23.
 * counts[0] = #rows with pred == y (loss 0), counts[1] = #rows with pred == 0 (loss 1),
 * counts[2] = #rows with pred == -y (loss 2).  loss = lambda*|w|^2 + (c1 + 2*c2)/n; acc = c0/n. */
int orc_loss_acc(const orc_csr* m, const double* w, double lambda, int64_t row_begin, int64_t row_end, double* loss,
                 double* acc, int64_t* counts /* 3 */, double* min_abs_margin) {
  if (row_end <= row_begin || row_begin < 0 || row_end > m->n_rows) return -1; /* reduce on empty throws */
  int64_t c0 = 0, c1 = 0, c2 = 0;
  double mam = INFINITY;
#pragma omp parallel for reduction(+ : c0, c1, c2) reduction(min : mam) schedule(static)
  for (int64_t i = row_begin; i < row_end; ++i) {
    double d = orc_row_dot(m, i, w);
    double yd = (double)m->label[i] * d;
    if (yd < 0) c0++;
    else if (yd > 0) c2++;
    else c1++;
    if (d != 0.0 && fabs(d) < mam) mam = fabs(d);
  }
  double nsq = 0.0;
  for (int32_t j = 0; j <= m->dim; ++j) nsq = nsq + w[j] * w[j]; /* Vec.scala:55 */
  double n = (double)(row_end - row_begin);
  /* samples.map(loss).reduce(_+_) / samples.size -- the sum of {0,1,2} is exact in fp64 */
  *loss = lambda * nsq + ((double)c1 + 2.0 * (double)c2) / n;
  *acc = (double)c0 / n;
  if (counts) {
    counts[0] = c0;
    counts[1] = c1;
    counts[2] = c2;
  }
  if (min_abs_margin) *min_abs_margin = mam;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Best-effort CPU baseline ("path B" of BASELINE.md): same math, rows spread over all host
 * cores with per-thread dense accumulators.  Summation order differs from orc_gradient (so it
 * is compared with a tolerance, never bit-for-bit).  Contiguous row range = a whole-shard batch.
 * ------------------------------------------------------------------------------------------ */
int orc_gradient_range_omp(const orc_csr* m, const double* w, const double* ds, double lambda, int64_t row_begin,
                           int64_t row_end, double* g_out, int64_t* n_active_out) {
  if (row_end <= row_begin || row_begin < 0 || row_end > m->n_rows) return -1;
  int32_t D = m->dim;
  int nt = orc_num_threads();
  static double* priv = NULL; /* per-thread dense accumulators, kept between calls (bench loop) */
  static size_t priv_len = 0;
  size_t need = (size_t)nt * ((size_t)D + 1);
  if (priv_len < need) {
    free(priv);
    priv = (double*)malloc(need * sizeof(double));
    priv_len = need;
  }
  int64_t n_active = 0;
#pragma omp parallel reduction(+ : n_active)
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    double* g = priv + (size_t)tid * ((size_t)D + 1);
    memset(g, 0, ((size_t)D + 1) * sizeof(double));
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = row_begin; i < row_end; ++i) {
      double d = orc_row_dot(m, i, w);
      double y = (double)m->label[i];
      if (y * d < 0) continue;
      n_active++;
      for (int64_t p = m->row_ptr[i]; p < m->row_ptr[i + 1]; ++p) g[m->col[p]] += (double)m->val[p] * y;
    }
    /* implicit barrier above; every thread reduces a slice of the coordinates over all accumulators */
#pragma omp for schedule(static)
    for (int32_t j = 0; j <= D; ++j) {
      double a = 0.0;
      for (int t = 0; t < nt; ++t) a += priv[(size_t)t * ((size_t)D + 1) + j];
      g_out[j] = filt(a);
    }
  }
  regularize_inplace(g_out, w, ds, lambda, D);
  if (n_active_out) *n_active_out = n_active;
  return 0;
}

/* Gate profile of a row range for the derived parity bound of the whole-shard tests: rows whose fp64 margin is
 * within eps of zero (but not exactly zero: an exact zero -- e.g. w = 0, or no common key -- is exactly zero in
 * fp32 too) may be gated differently by an fp32 implementation.  near_l1[j] = sum over those rows of |x_j|: the
 * most coordinate j of the gated sum (core/Slave.scala:147-153) can move if every one of them flips. */
int orc_range_gate_profile(const orc_csr* m, const double* w, int64_t row_begin, int64_t row_end, double eps,
                           int64_t* n_near_out, double* near_l1 /* dim+1, accumulated into */) {
  if (row_end <= row_begin || row_begin < 0 || row_end > m->n_rows) return -1;
  int64_t n_near = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : n_near)
  for (int64_t i = row_begin; i < row_end; ++i) {
    double d = orc_row_dot(m, i, w);
    if (d != 0.0 && fabs(d) < eps) {
      n_near++;
      for (int64_t p = m->row_ptr[i]; p < m->row_ptr[i + 1]; ++p) {
#pragma omp atomic
        near_l1[m->col[p]] += fabs((double)m->val[p]);
      }
    }
  }
  if (n_near_out) *n_near_out = n_near;
  return 0;
}

/* one whole-shard synchronous step with K=1 worker on all cores (bench cpu_baseline leg) */
int orc_sync_step_range_omp(const orc_csr* m, double* w, const double* ds, double lambda, int64_t row_begin,
                            int64_t row_end, double lr, int64_t* n_active_out) {
  int32_t D = m->dim;
  double* g = (double*)malloc(((size_t)D + 1) * sizeof(double));
  int rc = orc_gradient_range_omp(m, w, ds, lambda, row_begin, row_end, g, n_active_out);
  if (!rc)
    for (int32_t j = 0; j <= D; ++j) w[j] = filt(w[j] - filt(filt(g[j] / 1.0) * lr));
  free(g);
  return rc;
}

/* ------------------------------------------------------------------------------------------
 * "Literal" flavour: sparse vectors as sorted (key,value) arrays, every + rebuilds the whole
 * accumulator (math/Sparse.scala:33: new key Set + new Map per op), dot iterates the smaller
 * operand and looks the other one up (math/Sparse.scala:20-31).  This mirrors the reference's
 * per-sample algorithmic cost and is what bench.py times as the "reference CPU path"
 * (single thread per worker request, as core/Slave.scala:142 runs one Future per request).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t n, cap;
  int32_t* k;
  double* v;
} svec;

static void sv_init(svec* s, int32_t cap) {
  s->n = 0;
  s->cap = cap > 0 ? cap : 1;
  s->k = (int32_t*)malloc((size_t)s->cap * sizeof(int32_t));
  s->v = (double*)malloc((size_t)s->cap * sizeof(double));
}
static void sv_free(svec* s) {
  free(s->k);
  free(s->v);
  s->k = NULL;
  s->v = NULL;
  s->n = s->cap = 0;
}
static void sv_push(svec* s, int32_t k, double v) {
  if (!(fabs(v) > ORC_EPS)) return; /* constructor filter */
  if (s->n == s->cap) {
    s->cap *= 2;
    s->k = (int32_t*)realloc(s->k, (size_t)s->cap * sizeof(int32_t));
    s->v = (double*)realloc(s->v, (size_t)s->cap * sizeof(double));
  }
  s->k[s->n] = k;
  s->v[s->n] = v;
  s->n++;
}
static double sv_lookup(const svec* s, int32_t key) { /* map(idx) with default 0 */
  int32_t lo = 0, hi = s->n - 1;
  while (lo <= hi) {
    int32_t mid = (lo + hi) >> 1;
    if (s->k[mid] == key) return s->v[mid];
    if (s->k[mid] < key) lo = mid + 1;
    else hi = mid - 1;
  }
  return 0.0;
}
/* out = a (+|-) b over the union of keys; always a freshly built vector */
static void sv_addsub(const svec* a, const svec* b, double sign, svec* out) {
  sv_init(out, a->n + b->n);
  int32_t i = 0, j = 0;
  while (i < a->n || j < b->n) {
    if (j >= b->n || (i < a->n && a->k[i] < b->k[j])) {
      sv_push(out, a->k[i], a->v[i]);
      i++;
    } else if (i >= a->n || b->k[j] < a->k[i]) {
      sv_push(out, b->k[j], 0.0 + sign * b->v[j]);
      j++;
    } else {
      sv_push(out, a->k[i], a->v[i] + sign * b->v[j]);
      i++;
      j++;
    }
  }
}
static double sv_dot(const svec* a, const svec* b) {
  const svec* s = a->n < b->n ? a : b;
  const svec* l = a->n < b->n ? b : a;
  /* builds the intermediate product map, then sums it (Vec.scala:58) */
  svec prod;
  sv_init(&prod, s->n);
  for (int32_t i = 0; i < s->n; ++i) sv_push(&prod, s->k[i], s->v[i] * sv_lookup(l, s->k[i]));
  double acc = 0.0;
  for (int32_t i = 0; i < prod.n; ++i) acc = acc + prod.v[i];
  sv_free(&prod);
  return acc;
}
static void sv_from_dense(const double* d, int32_t dim, svec* out) {
  sv_init(out, 1024);
  for (int32_t j = 0; j <= dim; ++j) sv_push(out, j, d[j]);
}
static void sv_to_dense(const svec* s, int32_t dim, double* d) {
  memset(d, 0, ((size_t)dim + 1) * sizeof(double));
  for (int32_t i = 0; i < s->n; ++i) d[s->k[i]] = s->v[i];
}
static void sv_row(const orc_csr* m, int64_t i, double scale, svec* out) {
  int64_t b = m->row_ptr[i], e = m->row_ptr[i + 1];
  sv_init(out, (int32_t)(e - b));
  for (int64_t p = b; p < e; ++p) sv_push(out, m->col[p], (double)m->val[p] * scale);
}

/* literal Slave.gradient: decode w (the protobuf map -> Vec of core/package.scala:12-13),
 * per-sample backward, reduce(_ + _), regularize. */
int orc_lit_gradient(const orc_csr* m, const double* w_dense, const double* ds_dense, double lambda,
                     const int32_t* idx, int64_t n, double* g_out) {
  if (n <= 0) return -1;
  int32_t D = m->dim;
  svec w, ds, acc;
  sv_from_dense(w_dense, D, &w);
  sv_from_dense(ds_dense, D, &ds);
  int have = 0;
  for (int64_t t = 0; t < n; ++t) {
    int64_t i = idx[t];
    if (i < 0 || i >= m->n_rows) {
      sv_free(&w);
      sv_free(&ds);
      if (have) sv_free(&acc);
      return -2;
    }
    svec x, gi;
    sv_row(m, i, 1.0, &x);
    double y = (double)m->label[i];
    double activity = y * sv_dot(&x, &w);
    if (activity < 0) sv_init(&gi, 1);
    else sv_row(m, i, y, &gi);
    sv_free(&x);
    if (!have) {
      acc = gi;
      have = 1;
    } else {
      svec nxt;
      sv_addsub(&acc, &gi, 1.0, &nxt);
      sv_free(&acc);
      sv_free(&gi);
      acc = nxt;
    }
  }
  double s = lambda * 2.0 * sv_dot(&w, &ds);
  if (s != 0.0 && fabs(s) > ORC_EPS) {
    svec like, nxt;
    sv_init(&like, acc.n);
    for (int32_t i = 0; i < acc.n; ++i) sv_push(&like, acc.k[i], s);
    sv_addsub(&acc, &like, 1.0, &nxt);
    sv_free(&like);
    sv_free(&acc);
    acc = nxt;
  }
  sv_to_dense(&acc, D, g_out);
  sv_free(&acc);
  sv_free(&w);
  sv_free(&ds);
  return 0;
}

/* literal sync step for K workers, sequentially (the bench runs K of these on K threads) */
int orc_lit_sync_step(const orc_csr* m, double* w, const double* ds, double lambda,
                      const int32_t* const* idx_per_worker, const int64_t* n_per_worker, int32_t n_workers,
                      double lr) {
  int32_t D = m->dim;
  if (n_workers <= 0) return -1;
  double* gs = (double*)malloc((size_t)n_workers * ((size_t)D + 1) * sizeof(double));
  int rc = 0;
#pragma omp parallel for schedule(static, 1) num_threads(n_workers)
  for (int32_t k = 0; k < n_workers; ++k) {
    int r = orc_lit_gradient(m, w, ds, lambda, idx_per_worker[k], n_per_worker[k], gs + (size_t)k * ((size_t)D + 1));
    if (r) {
#pragma omp atomic write
      rc = r;
    }
  }
  if (!rc) {
    for (int32_t j = 0; j <= D; ++j) {
      double a = 0.0;
      for (int32_t k = 0; k < n_workers; ++k) a = filt(a + gs[(size_t)k * ((size_t)D + 1) + j]);
      w[j] = filt(w[j] - filt(filt(a / (double)n_workers) * lr));
    }
  }
  free(gs);
  return rc;
}
