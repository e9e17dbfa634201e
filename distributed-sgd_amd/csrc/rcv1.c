/* RCV1-v2 text files -> CSR, with the reference loader's exact semantics.
 *
 * ref: src/main/scala/epfl/distributed/utils/Dataset.scala:13-60 (Dataset.rcv1).  What that code does, and what
 * this file therefore does:
 *   * vector files (":19-34"): one document per line, `<id>  <key>:<value> <key>:<value> ...` -- the line is split on
 *     single spaces Java-style (String.split(' '): empty strings between consecutive separators are KEPT, trailing
 *     empty strings are dropped), token 0 is the document id and tokens 2.. are the features: the official files have
 *     TWO spaces after the id, so token 1 is empty; a line with one space would silently lose its first feature --
 *     reproduced, not "fixed".  `key:value` pairs go through `.toMap`: a repeated key keeps its LAST value.
 *   * labels (":36-45,53"): `rcv1-v2.topics.qrels` lines are `<topic> <id> 1`; (id -> +1 if topic == "CCAT" else -1)
 *     goes through `.toMap` too, so the LAST line of a document decides: a document tagged CCAT and, on a later
 *     line, another topic is labelled -1.
 *   * file order (":47-50,55-58"): lyrl2004_vectors_train.dat, then (full = true) lyrl2004_vectors_test_pt0..3.dat;
 *     rows in file order; a document without a qrels line is an error (`labels(id)` throws).
 *   * values stay as written; the Sparse constructor behind vecFactory drops abs(v) <= 1e-20 later
 *     (math/Sparse.scala:108-118) -- the engine's counting kernels do the same, so nothing is dropped here.
 * Keys are the file's own 1-based feature ids (Dataset.scala:30: used directly as map keys).
 *
 * Host-side I/O helper (gcc); not part of the HIP library.  SURVEY.md 8(f) item 2.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t n_rows, nnz, cap_rows, cap_nnz;
  int64_t* row_ptr;
  int32_t* col;
  float* val;
  int8_t* label;
  int32_t* doc_id;
  /* id -> label: open addressing, key 0 = empty (document ids are positive) */
  int64_t lab_cap, lab_n;
  int32_t* lab_key;
  int8_t* lab_val;
} dsgd_rcv1;

static void set_err(char* err, int errlen, const char* fmt, const char* a, long long b) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, fmt, a, b);
}

static int lab_grow(dsgd_rcv1* h) {
  const int64_t ncap = h->lab_cap ? h->lab_cap * 2 : (1 << 16);
  int32_t* nk = (int32_t*)calloc((size_t)ncap, sizeof(int32_t));
  int8_t* nv = (int8_t*)calloc((size_t)ncap, 1);
  if (!nk || !nv) {
    free(nk);
    free(nv);
    return -1;
  }
  for (int64_t i = 0; i < h->lab_cap; ++i) {
    if (!h->lab_key[i]) continue;
    uint64_t p = ((uint64_t)(uint32_t)h->lab_key[i] * 0x9E3779B97F4A7C15ull) >> 20;
    while (nk[p & (uint64_t)(ncap - 1)]) ++p;
    nk[p & (uint64_t)(ncap - 1)] = h->lab_key[i];
    nv[p & (uint64_t)(ncap - 1)] = h->lab_val[i];
  }
  free(h->lab_key);
  free(h->lab_val);
  h->lab_key = nk;
  h->lab_val = nv;
  h->lab_cap = ncap;
  return 0;
}
static int lab_put(dsgd_rcv1* h, int32_t id, int8_t v) {  /* last write wins (Map semantics) */
  if ((h->lab_n + 1) * 2 > h->lab_cap && lab_grow(h)) return -1;
  uint64_t p = ((uint64_t)(uint32_t)id * 0x9E3779B97F4A7C15ull) >> 20;
  for (;; ++p) {
    const uint64_t s = p & (uint64_t)(h->lab_cap - 1);
    if (h->lab_key[s] == id) {
      h->lab_val[s] = v;
      return 0;
    }
    if (!h->lab_key[s]) {
      h->lab_key[s] = id;
      h->lab_val[s] = v;
      h->lab_n++;
      return 0;
    }
  }
}
static int lab_get(const dsgd_rcv1* h, int32_t id, int8_t* v) {
  if (!h->lab_cap) return -1;
  uint64_t p = ((uint64_t)(uint32_t)id * 0x9E3779B97F4A7C15ull) >> 20;
  for (;; ++p) {
    const uint64_t s = p & (uint64_t)(h->lab_cap - 1);
    if (h->lab_key[s] == id) {
      *v = h->lab_val[s];
      return 0;
    }
    if (!h->lab_key[s]) return -1;
  }
}

/* Integer.parseInt: optional sign, at least one decimal digit, nothing else; int32 range */
static int parse_int(const char* s, size_t n, int32_t* out) {
  size_t i = 0;
  int neg = 0;
  if (n && (s[0] == '-' || s[0] == '+')) {
    neg = s[0] == '-';
    i = 1;
  }
  if (i == n) return -1;
  int64_t v = 0;
  for (; i < n; ++i) {
    if (s[i] < '0' || s[i] > '9') return -1;
    v = v * 10 + (s[i] - '0');
    if (v > 2147483648LL) return -1;
  }
  v = neg ? -v : v;
  if (v > 2147483647LL || v < -2147483648LL) return -1;
  *out = (int32_t)v;
  return 0;
}

/* Java String.split(' ') over [s, s+n): calls tok(ctx, index, ptr, len) for every kept token */
typedef int (*tok_fn)(void* ctx, int64_t index, const char* p, size_t len);
static int split_java(const char* s, size_t n, tok_fn tok, void* ctx) {
  /* trailing empty strings are removed: find the end of the last non-empty token */
  size_t end = n;
  while (end > 0 && s[end - 1] == ' ') --end;
  if (end == 0) return n == 0 ? tok(ctx, 0, s, 0) : 0; /* "".split -> [""] ; "   ".split -> [] */
  int64_t index = 0;
  size_t b = 0;
  for (size_t i = 0; i <= end; ++i) {
    if (i == end || s[i] == ' ') {
      const int rc = tok(ctx, index++, s + b, i - b);
      if (rc) return rc;
      b = i + 1;
    }
  }
  return 0;
}

static int reserve_rows(dsgd_rcv1* h, int64_t need) {
  if (need <= h->cap_rows) return 0;
  int64_t nc = h->cap_rows ? h->cap_rows * 2 : 4096;
  while (nc < need) nc *= 2;
  int64_t* rp = (int64_t*)realloc(h->row_ptr, sizeof(int64_t) * (size_t)(nc + 1));
  if (!rp) return -1;
  h->row_ptr = rp;
  int8_t* lb = (int8_t*)realloc(h->label, (size_t)nc);
  if (!lb) return -1;
  h->label = lb;
  int32_t* di = (int32_t*)realloc(h->doc_id, sizeof(int32_t) * (size_t)nc);
  if (!di) return -1;
  h->doc_id = di;
  h->cap_rows = nc;
  return 0;
}
static int reserve_nnz(dsgd_rcv1* h, int64_t need) {
  if (need <= h->cap_nnz) return 0;
  int64_t nc = h->cap_nnz ? h->cap_nnz * 2 : (1 << 20);
  while (nc < need) nc *= 2;
  int32_t* c = (int32_t*)realloc(h->col, sizeof(int32_t) * (size_t)nc);
  if (!c) return -1;
  h->col = c;
  float* v = (float*)realloc(h->val, sizeof(float) * (size_t)nc);
  if (!v) return -1;
  h->val = v;
  h->cap_nnz = nc;
  return 0;
}

typedef struct {
  dsgd_rcv1* h;
  int64_t row_start;
  int32_t id;
  int32_t last_key; /* keys seen so far are strictly increasing up to this one (no duplicate search needed) */
  int sorted;
  int seen_id;
  int bad;      /* 1: malformed token */
  int nomem;
} vec_ctx;
static int vec_tok(void* vctx, int64_t index, const char* p, size_t len) {
  vec_ctx* c = (vec_ctx*)vctx;
  dsgd_rcv1* h = c->h;
  if (index == 0) {
    c->seen_id = 1;
    if (parse_int(p, len, &c->id)) c->bad = 1;
    return c->bad;
  }
  if (index == 1) return 0; /* dropped by `.drop(2)` whatever it holds */
  /* elems = row.split(':'); elems(0).toInt -> elems(1).toDouble  (further ':' parts are ignored) */
  const char* colon = (const char*)memchr(p, ':', len);
  if (!colon) {
    c->bad = 1;
    return 1;
  }
  int32_t key;
  if (parse_int(p, (size_t)(colon - p), &key)) {
    c->bad = 1;
    return 1;
  }
  const char* vs = colon + 1;
  size_t vl = len - (size_t)(vs - p);
  const char* colon2 = (const char*)memchr(vs, ':', vl);
  if (colon2) vl = (size_t)(colon2 - vs);
  if (vl == 0 || vl > 63) {
    c->bad = 1;
    return 1;
  }
  char buf[64];
  memcpy(buf, vs, vl);
  buf[vl] = 0;
  char* endp = NULL;
  errno = 0;
  const double v = strtod(buf, &endp);
  if (endp == buf || *endp != 0) {
    c->bad = 1;
    return 1;
  }
  /* `.toMap`: a repeated key keeps the position of its first occurrence?  A Map has no order; what matters is
   * that the LAST value wins.  Replace in place. */
  if (!(c->sorted && key > c->last_key)) {
    for (int64_t q = c->row_start; q < h->nnz; ++q) {
      if (h->col[q] == key) {
        h->val[q] = (float)v;
        return 0;
      }
    }
    c->sorted = 0;
  }
  c->last_key = key;
  if (reserve_nnz(h, h->nnz + 1)) {
    c->nomem = 1;
    return 1;
  }
  h->col[h->nnz] = key;
  h->val[h->nnz] = (float)v;
  h->nnz++;
  return 0;
}

typedef struct {
  int topic_ccat;
  int32_t id;
  int have_id, bad;
} lab_ctx;
static int lab_tok(void* vctx, int64_t index, const char* p, size_t len) {
  lab_ctx* c = (lab_ctx*)vctx;
  if (index == 0) c->topic_ccat = len == 4 && memcmp(p, "CCAT", 4) == 0;
  else if (index == 1) {
    if (parse_int(p, len, &c->id)) c->bad = 1;
    else c->have_id = 1;
  }
  return 0;
}

static ssize_t read_line(FILE* f, char** buf, size_t* cap) {
  ssize_t n = getline(buf, cap, f);
  if (n < 0) return n;
  /* scala.io.Source.getLines strips "\n", "\r\n" and "\r" */
  while (n > 0 && ((*buf)[n - 1] == '\n' || (*buf)[n - 1] == '\r')) --n;
  return n;
}

void dsgd_rcv1_free(dsgd_rcv1* h) {
  if (!h) return;
  free(h->row_ptr);
  free(h->col);
  free(h->val);
  free(h->label);
  free(h->doc_id);
  free(h->lab_key);
  free(h->lab_val);
  free(h);
}

/* Dataset.rcv1(folder, full).  Returns NULL and fills `err` on failure. */
dsgd_rcv1* dsgd_rcv1_load(const char* folder, int full, char* err, int errlen) {
  dsgd_rcv1* h = (dsgd_rcv1*)calloc(1, sizeof(dsgd_rcv1));
  if (!h) return NULL;
  char path[4096];
  char* line = NULL;
  size_t cap = 0;
  /* labels first (Dataset.scala:53) */
  snprintf(path, sizeof(path), "%s/rcv1-v2.topics.qrels", folder);
  FILE* f = fopen(path, "r");
  if (!f) {
    set_err(err, errlen, "cannot open %s (errno %lld)", path, (long long)errno);
    dsgd_rcv1_free(h);
    return NULL;
  }
  long long ln = 0;
  ssize_t n;
  while ((n = read_line(f, &line, &cap)) >= 0) {
    ++ln;
    lab_ctx c = {0, 0, 0, 0};
    split_java(line, (size_t)n, lab_tok, &c);
    if (c.bad || !c.have_id) { /* parts(1) missing -> ArrayIndexOutOfBounds; not an int -> NumberFormatException */
      set_err(err, errlen, "%s: malformed line %lld", path, ln);
      goto fail;
    }
    if (lab_put(h, c.id, (int8_t)(c.topic_ccat ? 1 : -1))) {
      set_err(err, errlen, "%s: out of memory at line %lld", path, ln);
      goto fail;
    }
  }
  fclose(f);
  f = NULL;
  static const char* names[5] = {"lyrl2004_vectors_train.dat", "lyrl2004_vectors_test_pt0.dat", "lyrl2004_vectors_test_pt1.dat",
                                 "lyrl2004_vectors_test_pt2.dat", "lyrl2004_vectors_test_pt3.dat"};
  for (int k = 0; k < (full ? 5 : 1); ++k) {
    snprintf(path, sizeof(path), "%s/%s", folder, names[k]);
    f = fopen(path, "r");
    if (!f) {
      set_err(err, errlen, "cannot open %s (errno %lld)", path, (long long)errno);
      goto fail;
    }
    ln = 0;
    while ((n = read_line(f, &line, &cap)) >= 0) {
      ++ln;
      if (reserve_rows(h, h->n_rows + 1)) {
        set_err(err, errlen, "%s: out of memory at line %lld", path, ln);
        goto fail;
      }
      vec_ctx c = {h, h->nnz, 0, INT32_MIN, 1, 0, 0, 0};
      h->row_ptr[h->n_rows] = h->nnz;
      const int rc = split_java(line, (size_t)n, vec_tok, &c);
      if (rc || c.bad || !c.seen_id) { /* parts(0) of an all-blank line: ArrayIndexOutOfBoundsException */
        set_err(err, errlen, c.nomem ? "%s: out of memory at line %lld" : "%s: malformed line %lld", path, ln);
        goto fail;
      }
      int8_t y;
      if (lab_get(h, c.id, &y)) { /* labels(id): NoSuchElementException */
        set_err(err, errlen, "%s: no label for the document of line %lld", path, ln);
        goto fail;
      }
      h->label[h->n_rows] = y;
      h->doc_id[h->n_rows] = c.id;
      h->n_rows++;
    }
    fclose(f);
    f = NULL;
  }
  if (reserve_rows(h, h->n_rows + 1) == 0) h->row_ptr[h->n_rows] = h->nnz;
  free(line);
  return h;
fail:
  if (f) fclose(f);
  free(line);
  dsgd_rcv1_free(h);
  return NULL;
}

int64_t dsgd_rcv1_rows(const dsgd_rcv1* h) { return h->n_rows; }
int64_t dsgd_rcv1_nnz(const dsgd_rcv1* h) { return h->nnz; }
void dsgd_rcv1_copy(const dsgd_rcv1* h, int64_t* row_ptr, int32_t* col, float* val, int8_t* label, int32_t* doc_id) {
  if (row_ptr) {
    if (h->n_rows) memcpy(row_ptr, h->row_ptr, sizeof(int64_t) * (size_t)(h->n_rows + 1));
    else row_ptr[0] = 0;
  }
  if (col && h->nnz) memcpy(col, h->col, sizeof(int32_t) * (size_t)h->nnz);
  if (val && h->nnz) memcpy(val, h->val, sizeof(float) * (size_t)h->nnz);
  if (label && h->n_rows) memcpy(label, h->label, (size_t)h->n_rows);
  if (doc_id && h->n_rows) memcpy(doc_id, h->doc_id, sizeof(int32_t) * (size_t)h->n_rows);
}
