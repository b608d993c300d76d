/*
 * TEST INFRASTRUCTURE — plain-C restatement of the reference's retrieval loop and contrastive loss.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 *
 *  t2l_oracle_retrieve : training/coarse.py:119-125 — per query float64 `cell_encodings @ t` (f32 values
 *                        widened to f64, coarse.py:81-98) and a FULL descending sort of all N scores, first K kept.
 *                        Exact ties: lower row first (np.argsort is unstable in the reference; fixtures are tie-free).
 *  t2l_oracle_contrastive : training/losses.py:269-283 (no max-subtraction), float64 accumulation.
 * Pinned by tests/test_oracle_golden.py against vectors generated from the imported reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double s; int64_t i; } pair_t;

static int cmp_desc(const void* a, const void* b) {
  const pair_t* x = (const pair_t*)a; const pair_t* y = (const pair_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

/* cells f32[n,d], queries f32[q,d] -> idx i64[q,k], score f64[q,k]; returns 0, or -1 on allocation failure */
int t2l_oracle_retrieve(const float* cells, int64_t n, const float* queries, int64_t q, int64_t d, int64_t k,
                        int64_t* out_idx, double* out_score) {
  double* c64 = (double*)malloc((size_t)n * d * sizeof(double));
  pair_t* sc = (pair_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(pair_t));
  double* t = (double*)malloc((size_t)d * sizeof(double));
  if (!c64 || !sc || !t) { free(c64); free(sc); free(t); return -1; }
  for (int64_t i = 0; i < n * d; ++i) c64[i] = (double)cells[i];       /* np.zeros(float64)[...] = f32 */
  if (k > n) k = n;
  for (int64_t j = 0; j < q; ++j) {
    for (int64_t e = 0; e < d; ++e) t[e] = (double)queries[j * d + e];
    for (int64_t i = 0; i < n; ++i) {                                  /* scores = cell_encodings[:] @ t */
      const double* row = c64 + i * d; double acc = 0.0;
      for (int64_t e = 0; e < d; ++e) acc += row[e] * t[e];
      sc[i].s = acc; sc[i].i = i;
    }
    qsort(sc, (size_t)n, sizeof(pair_t), cmp_desc);                     /* argsort(-scores): full sort */
    for (int64_t r = 0; r < k; ++r) { out_idx[j * k + r] = sc[r].i; out_score[j * k + r] = sc[r].s; }
  }
  free(c64); free(sc); free(t);
  return 0;
}

/* im, s f32[b,d] -> loss (f64). grad_im / grad_s f64[b,d] may be NULL. */
double t2l_oracle_contrastive(const float* im, const float* s, int64_t b, int64_t d, double temperature,
                              double* grad_im, double* grad_s) {
  double* a = (double*)malloc((size_t)b * d * sizeof(double));
  double* p = (double*)malloc((size_t)b * d * sizeof(double));
  double* e = (double*)malloc((size_t)b * b * sizeof(double));
  double* na = (double*)malloc((size_t)b * sizeof(double));
  double* np_ = (double*)malloc((size_t)b * sizeof(double));
  double* row = (double*)calloc((size_t)b, sizeof(double));
  double* col = (double*)calloc((size_t)b, sizeof(double));
  double* diag = (double*)malloc((size_t)b * sizeof(double));
  for (int64_t i = 0; i < b; ++i) {
    double sa = 0, sp = 0;
    for (int64_t k = 0; k < d; ++k) { sa += (double)im[i*d+k] * im[i*d+k]; sp += (double)s[i*d+k] * s[i*d+k]; }
    na[i] = sqrt(sa); np_[i] = sqrt(sp);
    for (int64_t k = 0; k < d; ++k) { a[i*d+k] = im[i*d+k] / na[i]; p[i*d+k] = s[i*d+k] / np_[i]; }   /* :271-272 */
  }
  for (int64_t i = 0; i < b; ++i)
    for (int64_t j = 0; j < b; ++j) {
      double sim = 0; for (int64_t k = 0; k < d; ++k) sim += a[i*d+k] * p[j*d+k];                  /* :274 */
      if (i == j) diag[i] = sim;
      e[i*b+j] = exp(sim / temperature); row[i] += e[i*b+j]; col[j] += e[i*b+j];                   /* :278 */
    }
  double loss = 0;
  for (int64_t i = 0; i < b; ++i) {
    const double num = exp(diag[i] / temperature);
    loss += -log(num / col[i]) - log(num / row[i]);                                                /* :280 */
  }
  loss /= (double)b;                                                                               /* :281 */
  if (grad_im && grad_s) {
    double* ga = (double*)calloc((size_t)b * d, sizeof(double));
    double* gp = (double*)calloc((size_t)b * d, sizeof(double));
    for (int64_t i = 0; i < b; ++i)
      for (int64_t j = 0; j < b; ++j) {
        const double g = (e[i*b+j] / col[j] + e[i*b+j] / row[i] - (i == j ? 2.0 : 0.0)) / (temperature * (double)b);
        for (int64_t k = 0; k < d; ++k) { ga[i*d+k] += g * p[j*d+k]; gp[j*d+k] += g * a[i*d+k]; }
      }
    for (int64_t i = 0; i < b; ++i) {
      double da = 0, dp = 0;
      for (int64_t k = 0; k < d; ++k) { da += ga[i*d+k] * a[i*d+k]; dp += gp[i*d+k] * p[i*d+k]; }
      for (int64_t k = 0; k < d; ++k) {
        grad_im[i*d+k] = (ga[i*d+k] - a[i*d+k] * da) / na[i];
        grad_s[i*d+k] = (gp[i*d+k] - p[i*d+k] * dp) / np_[i];
      }
    }
    free(ga); free(gp);
  }
  free(a); free(p); free(e); free(na); free(np_); free(row); free(col); free(diag);
  return loss;
}
