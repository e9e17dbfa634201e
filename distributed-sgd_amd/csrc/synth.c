/*
 * Synthetic "RCV1-like" CSR generator (host tool used by bench.py and the tests; it feeds the
 * SAME arrays to the HIP engine and to the CPU oracle).
 *
 * Shape of the data (BASELINE.md section 3 / SURVEY.md 8(d)); the reference itself ships no data
 * and its loader needs files that are not in the tree (utils/Dataset.scala:47-51):
 *   - D = 47236 features with 1-based ids (utils/Dataset.scala:16,30)
 *   - row nnz ~ clipped log-normal, mean ~ 75, min 1, max 1200
 *   - column ids drawn WITHOUT replacement per row from Zipf(alpha) over a fixed random
 *     permutation of the features (alias-table sampling), stored ascending
 *   - values |N(0,1)| + 0.1, L2-normalised per row (RCV1 vectors are cosine-normalised)
 *   - labels from a planted separator on the 2000 most frequent features plus noise;
 *     threshold tau is fixed from the first 65536 rows so that ~47 % of rows are +1
 *   - rows are i.i.d. (hence "pre-shuffled"); every row depends only on (seed, row index), so
 *     any shard [row0, row0+n) can be generated independently and in parallel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SYN_MAX_NNZ 1200
#define SYN_PLANTED 2000

typedef struct {
  uint64_t seed;
  int32_t dim;       /* D */
  double zipf_alpha; /* 1.1 */
  double nnz_mu;     /* log-normal mu */
  double nnz_sigma;  /* log-normal sigma */
  double label_noise;
  double pos_fraction; /* 0.47 */
  /* derived */
  int32_t* perm;    /* rank -> 1-based feature id */
  double* alias_p;  /* alias table over ranks */
  int32_t* alias_j;
  double* wstar;    /* planted weights by rank (first SYN_PLANTED ranks) */
  double tau;
} synth_t;

static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
typedef struct {
  uint64_t s;
} rng_t;
static inline void rng_seed(rng_t* r, uint64_t seed, uint64_t stream, uint64_t row) {
  r->s = mix64(seed ^ mix64(stream * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull) ^ mix64(row + 0x2545F4914F6CDD1Dull));
}
static inline uint64_t rng_u64(rng_t* r) {
  r->s += 0x9E3779B97F4A7C15ull;
  uint64_t z = r->s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline double rng_unif(rng_t* r) { return ((rng_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static inline double rng_normal(rng_t* r) {
  double u1 = rng_unif(r), u2 = rng_unif(r);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static int32_t row_nnz(const synth_t* g, uint64_t row) {
  rng_t r;
  rng_seed(&r, g->seed, 1, row);
  double v = exp(g->nnz_mu + g->nnz_sigma * rng_normal(&r));
  int32_t n = (int32_t)floor(v + 0.5);
  if (n < 1) n = 1;
  if (n > SYN_MAX_NNZ) n = SYN_MAX_NNZ;
  if (n > g->dim) n = g->dim;
  return n;
}

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return (x > y) - (x < y);
}

/* generate one row; returns the noisy planted margin (before thresholding) */
static double gen_row(const synth_t* g, uint64_t row, int32_t n, int32_t* col, float* val, uint8_t* seen /* dim bits */,
                      int32_t* ranks /* n */) {
  rng_t r;
  rng_seed(&r, g->seed, 2, row);
  int32_t got = 0;
  while (got < n) {
    double u = rng_unif(&r) * g->dim;
    int32_t k = (int32_t)u;
    if (k >= g->dim) k = g->dim - 1;
    int32_t rank = (u - k) < g->alias_p[k] ? k : g->alias_j[k];
    if (seen[rank >> 3] & (1u << (rank & 7))) continue;
    seen[rank >> 3] |= (uint8_t)(1u << (rank & 7));
    ranks[got++] = rank;
  }
  for (int32_t t = 0; t < n; ++t) seen[ranks[t] >> 3] = 0; /* rows never share a byte state across calls */
  /* sort by feature id so that columns are ascending (text format of utils/Dataset.scala:19-34) */
  for (int32_t t = 0; t < n; ++t) col[t] = g->perm[ranks[t]];
  qsort(col, (size_t)n, sizeof(int32_t), cmp_i32);
  /* values are drawn in column order, margin needs rank -> use an inverse lookup through wstar_by_id */
  double nsq = 0.0;
  for (int32_t t = 0; t < n; ++t) {
    double v = fabs(rng_normal(&r)) + 0.1;
    val[t] = (float)v;
    nsq += (double)val[t] * (double)val[t];
  }
  float inv = (float)(1.0 / sqrt(nsq));
  for (int32_t t = 0; t < n; ++t) val[t] = val[t] * inv;
  double margin = 0.0;
  for (int32_t t = 0; t < n; ++t) margin += (double)val[t] * g->wstar[col[t]];
  margin += g->label_noise * rng_normal(&r);
  return margin;
}

static int cmp_f64(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

void dsgd_synth_destroy(synth_t* g) {
  if (!g) return;
  free(g->perm);
  free(g->alias_p);
  free(g->alias_j);
  free(g->wstar);
  free(g);
}

/* the same generator with another column-frequency exponent / mean row length (the dispatcher's thresholds were measured on
 * Zipf(1.1) rows of ~75 non-zeros: tests/test_gpu_dispatch.py holds the choice to shapes they were NOT tuned on) */
synth_t* dsgd_synth_create_shaped(uint64_t seed, int32_t dim, double zipf_alpha, double nnz_mean);
synth_t* dsgd_synth_create(uint64_t seed, int32_t dim) { return dsgd_synth_create_shaped(seed, dim, 1.1, 75.0); }
synth_t* dsgd_synth_create_shaped(uint64_t seed, int32_t dim, double zipf_alpha, double nnz_mean) {
  if (dim < 1 || !(zipf_alpha > 0.0) || !(nnz_mean >= 1.0)) return NULL;
  synth_t* g = (synth_t*)calloc(1, sizeof(synth_t));
  g->seed = seed;
  g->dim = dim;
  g->zipf_alpha = zipf_alpha;
  g->nnz_sigma = 0.8;
  g->nnz_mu = log(nnz_mean) - 0.5 * 0.8 * 0.8; /* E[lognormal] = nnz_mean (75) before clipping */
  g->label_noise = 0.1;
  g->pos_fraction = 0.47;
  /* fixed random permutation rank -> feature id (Fisher-Yates on a seeded stream) */
  g->perm = (int32_t*)malloc((size_t)dim * sizeof(int32_t));
  for (int32_t i = 0; i < dim; ++i) g->perm[i] = i + 1;
  rng_t r;
  rng_seed(&r, seed, 3, 0);
  for (int32_t i = dim - 1; i > 0; --i) {
    int32_t j = (int32_t)(rng_u64(&r) % (uint64_t)(i + 1));
    int32_t t = g->perm[i];
    g->perm[i] = g->perm[j];
    g->perm[j] = t;
  }
  /* Zipf pmf over ranks and its alias table (Vose) */
  double* p = (double*)malloc((size_t)dim * sizeof(double));
  double z = 0.0;
  for (int32_t i = 0; i < dim; ++i) {
    p[i] = pow((double)(i + 1), -g->zipf_alpha);
    z += p[i];
  }
  g->alias_p = (double*)malloc((size_t)dim * sizeof(double));
  g->alias_j = (int32_t*)malloc((size_t)dim * sizeof(int32_t));
  int32_t* small = (int32_t*)malloc((size_t)dim * sizeof(int32_t));
  int32_t* large = (int32_t*)malloc((size_t)dim * sizeof(int32_t));
  int32_t ns = 0, nl = 0;
  for (int32_t i = 0; i < dim; ++i) {
    p[i] = p[i] / z * dim;
    if (p[i] < 1.0) small[ns++] = i;
    else large[nl++] = i;
  }
  while (ns > 0 && nl > 0) {
    int32_t s = small[--ns], l = large[--nl];
    g->alias_p[s] = p[s];
    g->alias_j[s] = l;
    p[l] = (p[l] + p[s]) - 1.0;
    if (p[l] < 1.0) small[ns++] = l;
    else large[nl++] = l;
  }
  while (nl > 0) {
    int32_t l = large[--nl];
    g->alias_p[l] = 1.0;
    g->alias_j[l] = l;
  }
  while (ns > 0) {
    int32_t s = small[--ns];
    g->alias_p[s] = 1.0;
    g->alias_j[s] = s;
  }
  free(small);
  free(large);
  free(p);
  /* planted separator: N(0,1) on the SYN_PLANTED most frequent features, indexed by feature id */
  g->wstar = (double*)calloc((size_t)dim + 1, sizeof(double));
  rng_seed(&r, seed, 4, 0);
  int32_t np = dim < SYN_PLANTED ? dim : SYN_PLANTED;
  for (int32_t i = 0; i < np; ++i) g->wstar[g->perm[i]] = rng_normal(&r);
  /* tau from the first rows so that ~pos_fraction of rows get +1 */
  int32_t ncal = 65536;
  double* margins = (double*)malloc((size_t)ncal * sizeof(double));
#pragma omp parallel
  {
    uint8_t* seen = (uint8_t*)calloc(((size_t)dim >> 3) + 1, 1);
    int32_t* ranks = (int32_t*)malloc(SYN_MAX_NNZ * sizeof(int32_t));
    int32_t* col = (int32_t*)malloc(SYN_MAX_NNZ * sizeof(int32_t));
    float* val = (float*)malloc(SYN_MAX_NNZ * sizeof(float));
#pragma omp for schedule(static)
    for (int32_t i = 0; i < ncal; ++i) margins[i] = gen_row(g, (uint64_t)i, row_nnz(g, (uint64_t)i), col, val, seen, ranks);
    free(seen);
    free(ranks);
    free(col);
    free(val);
  }
  qsort(margins, (size_t)ncal, sizeof(double), cmp_f64);
  int32_t q = (int32_t)((1.0 - g->pos_fraction) * ncal);
  if (q >= ncal) q = ncal - 1;
  g->tau = margins[q];
  free(margins);
  return g;
}

/* pass 1: row_ptr[0..n_rows] for global rows [row0, row0 + n_rows); returns total nnz */
int64_t dsgd_synth_row_ptr(const synth_t* g, int64_t row0, int64_t n_rows, int64_t* row_ptr) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_rows; ++i) row_ptr[i + 1] = row_nnz(g, (uint64_t)(row0 + i));
  row_ptr[0] = 0;
  for (int64_t i = 0; i < n_rows; ++i) row_ptr[i + 1] += row_ptr[i];
  return row_ptr[n_rows];
}

/* pass 2: fill col / val / label for the same rows */
void dsgd_synth_fill(const synth_t* g, int64_t row0, int64_t n_rows, const int64_t* row_ptr, int32_t* col, float* val,
                     int8_t* label) {
#pragma omp parallel
  {
    uint8_t* seen = (uint8_t*)calloc(((size_t)g->dim >> 3) + 1, 1);
    int32_t* ranks = (int32_t*)malloc(SYN_MAX_NNZ * sizeof(int32_t));
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < n_rows; ++i) {
      int32_t n = (int32_t)(row_ptr[i + 1] - row_ptr[i]);
      double m = gen_row(g, (uint64_t)(row0 + i), n, col + row_ptr[i], val + row_ptr[i], seen, ranks);
      label[i] = m > g->tau ? 1 : -1;
    }
    free(seen);
    free(ranks);
  }
}

int32_t dsgd_synth_dim(const synth_t* g) { return g->dim; }
double dsgd_synth_tau(const synth_t* g) { return g->tau; }
