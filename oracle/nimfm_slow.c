/*
 * oracle/nimfm_slow.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restatement of the brute-force models the reference's own unit tests use
 * as THEIR oracle (tests/kernels_slow.nim, tests/comb.nim,
 * tests/model/fm_slow.nim, tests/model/ffm_slow.nim,
 * tests/optimizer/{sgd,adagrad}_slow.nim, tests/optimizer/{sgd,adagrad}_ffm_slow.nim).
 * They enumerate index subsets explicitly and update every coordinate densely
 * at every step, so they share no code path with nimfm_oracle.c; agreement of
 * the two (tests/test_oracle_*.py) is how this oracle is pinned.
 * Dense X is row-major [n][d].  Model layouts as in nimfm_oracle.h.
 */
#include "nimfm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- tests/comb.nim:1-24: enumerate subsets of size m of {0..n-1} (optionally
 * skipping index `skip`), ascending, calling back with the index tuple. ---- */
typedef void (*comb_cb)(const int* idx, int m, void* ctx);

static void comb_rec(int n, int m, int k0, int skip, int* buf, int depth, int total, comb_cb cb,
                     void* ctx) {
  if (m == 1) {
    for (int i = k0; i < n; i++) {
      if (i == skip) continue;
      buf[depth] = i;
      cb(buf, total, ctx);
    }
  } else {
    for (int i = k0; i < n - m + 1; i++) {
      if (i == skip) continue;
      buf[depth] = i;
      comb_rec(n, m - 1, i + 1, skip, buf, depth + 1, total, cb, ctx);
    }
  }
}

static void for_each_comb(int n, int m, int skip, comb_cb cb, void* ctx) {
  int buf[16];
  if (m <= 0 || m > 16) return;
  comb_rec(n, m, 0, skip, buf, 0, m, cb, ctx);
}

typedef struct prod_ctx {
  const double* x; /* dense row, length d */
  const double* p; /* P[o][s][:], length d+m */
  int d;
  double acc;
} prod_ctx;

/* the body of tests/kernels_slow.nim:21-27 and tests/model/fm_slow.nim:53-59 */
static void prod_cb(const int* idx, int m, void* vctx) {
  prod_ctx* c = (prod_ctx*)vctx;
  double prod = 1.0;
  for (int t = 0; t < m; t++) {
    const int j = idx[t];
    prod *= c->p[j];
    if (j < c->d) prod *= c->x[j];
  }
  c->acc += prod;
}

/* tests/kernels_slow.nim:18-27 */
double slow_anova(const double* Xrow, const double* Prow, int d, int m, int degree) {
  prod_ctx c = {Xrow, Prow, d, 0.0};
  for_each_comb(d + m, degree, -1, prod_cb, &c);
  return c.acc;
}

/* tests/model/fm_slow.nim:42-60 */
static double slow_fm_predict_row(const double* x, int d, int degree, int k, int n_orders, int n_aug,
                                  const double* P, const double* w, double intercept) {
  const int da = d + n_aug;
  double result = intercept;
  for (int j = 0; j < d; j++) result += w[j] * x[j];
  for (int o = 0; o < n_orders; o++)
    for (int s = 0; s < k; s++) {
      prod_ctx c = {x, P + ((size_t)o * k + s) * da, d, 0.0};
      for_each_comb(da, degree - o, -1, prod_cb, &c);
      result += c.acc;
    }
  return result;
}

int slow_fm_decision_function(const double* Xd, int64_t n, int d, int degree, int k, int n_orders,
                              int n_aug, const double* P, const double* w, double intercept,
                              double* out) {
  for (int64_t i = 0; i < n; i++)
    out[i] = slow_fm_predict_row(Xd + (size_t)i * d, d, degree, k, n_orders, n_aug, P, w, intercept);
  return 0;
}

/* tests/model/fm_slow.nim:111-134: grad[o][s][j] += dL * d/dP */
static void slow_fm_compute_grad(const double* x, int d, int degree, int k, int n_orders, int n_aug,
                                 const double* P, double dL, double* grad) {
  const int da = d + n_aug;
  for (int o = 0; o < n_orders; o++)
    for (int s = 0; s < k; s++)
      for (int j = 0; j < da; j++) {
        double tmp = 0.0;
        const int m = degree - o - 1;
        if (m == 0) {
          /* combNotj(n, 0, j) yields nothing in the reference (m==1 is the
           * base case and m<=0 falls through the else-branch with an empty
           * range); degree-o-1 == 0 only when degree-o == 1, which nOrders
           * never produces for degree >= 2. */
          tmp = 0.0;
        } else {
          prod_ctx c = {x, P + ((size_t)o * k + s) * da, d, 0.0};
          for_each_comb(da, m, j, prod_cb, &c);
          tmp = c.acc;
        }
        if (j < d) tmp *= x[j];
        grad[((size_t)o * k + s) * da + j] += dL * tmp;
      }
}

/* tests/optimizer/sgd_slow.nim:38-91 */
int slow_fm_sgd_fit(const double* Xd, int64_t n, int d, const double* y, int degree, int k,
                    int n_orders, int n_aug, double* P, double* w, double* intercept,
                    const orc_sgd_cfg* cfg, int max_iter, const int64_t* perms, int64_t* it) {
  const int da = d + n_aug;
  const size_t np = (size_t)n_orders * k * da;
  double* grad = (double*)calloc(np ? np : 1, sizeof(double));
  if (!grad) return -1;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      const double* x = Xd + (size_t)i * d;
      const double y_pred = slow_fm_predict_row(x, d, degree, k, n_orders, n_aug, P, w, *intercept);
      const double dL = orc_dloss(cfg->loss, cfg->loss_param, y[i], y_pred);
      memset(grad, 0, sizeof(double) * np);
      slow_fm_compute_grad(x, d, degree, k, n_orders, n_aug, P, dL, grad);
      const double w_eta = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha, *it);
      const double P_eta = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->beta, *it);
      if (cfg->fit_intercept) {
        const double update = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha0, *it) *
                              (dL + cfg->alpha0 * *intercept);
        *intercept -= update;
      }
      if (cfg->fit_linear)
        for (int j = 0; j < d; j++) {
          const double update = w_eta * (dL * x[j] + cfg->alpha * w[j]);
          w[j] -= update;
        }
      for (size_t e = 0; e < np; e++) {
        const double update = P_eta * (grad[e] + cfg->beta * P[e]);
        P[e] -= update;
      }
      (*it)++;
    }
  }
  free(grad);
  return 0;
}

/* tests/optimizer/adagrad_slow.nim:29-102 (dense state kept locally: the
 * reference test never warm-starts the slow optimiser) */
typedef struct slow_ada {
  double *gsum_P, *gnorm_P, *gsum_w, *gnorm_w, gsum_b, gnorm_b;
} slow_ada;

static int slow_ada_alloc(slow_ada* S, size_t np, int d, double eps) {
  S->gsum_P = (double*)calloc(np ? np : 1, sizeof(double));
  S->gnorm_P = (double*)calloc(np ? np : 1, sizeof(double));
  S->gsum_w = (double*)calloc(d ? d : 1, sizeof(double));
  S->gnorm_w = (double*)calloc(d ? d : 1, sizeof(double));
  if (!S->gsum_P || !S->gnorm_P || !S->gsum_w || !S->gnorm_w) return -1;
  for (size_t e = 0; e < np; e++) S->gnorm_P[e] = 0.0 + eps;
  for (int j = 0; j < d; j++) S->gnorm_w[j] = 0.0 + eps;
  S->gsum_b = 0.0;
  S->gnorm_b = eps;
  return 0;
}
static void slow_ada_free(slow_ada* S) {
  free(S->gsum_P); free(S->gnorm_P); free(S->gsum_w); free(S->gnorm_w);
}

/* tests/optimizer/adagrad_slow.nim:44-70 */
static void slow_ada_update(slow_ada* S, const double* x, int d, size_t np, double* P, double* w,
                            double* intercept, const double* grad, double dL,
                            const orc_adagrad_cfg* c, int64_t it_) {
  const double it = (double)it_;
  if (c->fit_intercept) {
    S->gsum_b += dL;
    S->gnorm_b += dL * dL;
    const double denom = sqrt(S->gnorm_b) + c->eta0 * it * c->alpha0;
    *intercept = -c->eta0 * S->gsum_b / denom;
  }
  if (c->fit_linear) {
    const double denom = c->eta0 * it * c->alpha;
    for (int j = 0; j < d; j++) {
      S->gsum_w[j] += dL * x[j];
      const double g = dL * x[j];
      S->gnorm_w[j] += g * g;
      w[j] = -c->eta0 * S->gsum_w[j];
      w[j] /= (denom + sqrt(S->gnorm_w[j]));
    }
  }
  const double denom = c->eta0 * it * c->beta;
  for (size_t e = 0; e < np; e++) {
    S->gsum_P[e] += grad[e];
    S->gnorm_P[e] += grad[e] * grad[e];
    P[e] = -c->eta0 * S->gsum_P[e];
    P[e] /= (denom + sqrt(S->gnorm_P[e]));
  }
}

int slow_fm_adagrad_fit(const double* Xd, int64_t n, int d, const double* y, int degree, int k,
                        int n_orders, int n_aug, double* P, double* w, double* intercept,
                        const orc_adagrad_cfg* cfg, int max_iter, const int64_t* perms,
                        int64_t* it) {
  const int da = d + n_aug;
  const size_t np = (size_t)n_orders * k * da;
  double* grad = (double*)calloc(np ? np : 1, sizeof(double));
  slow_ada S;
  if (!grad || slow_ada_alloc(&S, np, d, cfg->eps)) return -1;
  for (int epoch = 0; epoch < max_iter; epoch++)
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      const double* x = Xd + (size_t)i * d;
      const double y_pred = slow_fm_predict_row(x, d, degree, k, n_orders, n_aug, P, w, *intercept);
      const double dL = orc_dloss(cfg->loss, cfg->loss_param, y[i], y_pred);
      memset(grad, 0, sizeof(double) * np);
      slow_fm_compute_grad(x, d, degree, k, n_orders, n_aug, P, dL, grad);
      slow_ada_update(&S, x, d, np, P, w, intercept, grad, dL, cfg, *it);
      (*it)++;
    }
  free(grad);
  slow_ada_free(&S);
  return 0;
}

/* ---- FFM brute force: tests/model/ffm_slow.nim (nAugments == 0) ---- */
/* :38-56 */
static double slow_ffm_predict_row(const double* x, int d, const int64_t* field_of, int k,
                                   const double* P, const double* w, double intercept) {
  double result = intercept;
  for (int j = 0; j < d; j++) result += w[j] * x[j];
  for (int j1 = 0; j1 < d; j1++) {
    const int64_t f1 = field_of[j1];
    const double val1 = x[j1];
    for (int j2 = j1 + 1; j2 < d; j2++) {
      const int64_t f2 = field_of[j2];
      const double val2 = x[j2];
      const double interaction = val1 * val2;
      for (int s = 0; s < k; s++)
        result += interaction * P[((size_t)f2 * d + j1) * k + s] * P[((size_t)f1 * d + j2) * k + s];
    }
  }
  return result;
}

int slow_ffm_decision_function(const double* Xd, int64_t n, int d, const int64_t* field_of,
                               int n_fields, int k, const double* P, const double* w,
                               double intercept, double* out) {
  (void)n_fields;
  for (int64_t i = 0; i < n; i++)
    out[i] = slow_ffm_predict_row(Xd + (size_t)i * d, d, field_of, k, P, w, intercept);
  return 0;
}

/* :110-127 */
static void slow_ffm_compute_grad(const double* x, int d, const int64_t* field_of, int k,
                                  const double* P, double dL, double* grad) {
  for (int j1 = 0; j1 < d; j1++) {
    const int64_t f1 = field_of[j1];
    const double val1 = x[j1];
    for (int j2 = j1 + 1; j2 < d; j2++) {
      const int64_t f2 = field_of[j2];
      const double val2 = x[j2];
      const double interaction = val1 * val2;
      for (int s = 0; s < k; s++) {
        grad[((size_t)f2 * d + j1) * k + s] += dL * P[((size_t)f1 * d + j2) * k + s] * interaction;
        grad[((size_t)f1 * d + j2) * k + s] += dL * P[((size_t)f2 * d + j1) * k + s] * interaction;
      }
    }
  }
}

/* tests/optimizer/sgd_ffm_slow.nim:8-57 */
int slow_ffm_sgd_fit(const double* Xd, int64_t n, int d, const int64_t* field_of, int n_fields,
                     const double* y, int k, double* P, double* w, double* intercept,
                     const orc_sgd_cfg* cfg, int max_iter, const int64_t* perms, int64_t* it) {
  const size_t np = (size_t)n_fields * d * k;
  double* grad = (double*)calloc(np ? np : 1, sizeof(double));
  if (!grad) return -1;
  for (int epoch = 0; epoch < max_iter; epoch++)
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      const double* x = Xd + (size_t)i * d;
      const double y_pred = slow_ffm_predict_row(x, d, field_of, k, P, w, *intercept);
      const double dL = orc_dloss(cfg->loss, cfg->loss_param, y[i], y_pred);
      memset(grad, 0, sizeof(double) * np);
      slow_ffm_compute_grad(x, d, field_of, k, P, dL, grad);
      const double w_eta = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha, *it);
      const double P_eta = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->beta, *it);
      if (cfg->fit_intercept) {
        const double update = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha0, *it) *
                              (dL + cfg->alpha0 * *intercept);
        *intercept -= update;
      }
      if (cfg->fit_linear)
        for (int j = 0; j < d; j++) {
          const double update = w_eta * (dL * x[j] + cfg->alpha * w[j]);
          w[j] -= update;
        }
      for (size_t e = 0; e < np; e++) {
        const double update = P_eta * (grad[e] + cfg->beta * P[e]);
        P[e] -= update;
      }
      (*it)++;
    }
  free(grad);
  return 0;
}

/* tests/optimizer/adagrad_ffm_slow.nim:8-37 */
int slow_ffm_adagrad_fit(const double* Xd, int64_t n, int d, const int64_t* field_of, int n_fields,
                         const double* y, int k, double* P, double* w, double* intercept,
                         const orc_adagrad_cfg* cfg, int max_iter, const int64_t* perms,
                         int64_t* it) {
  const size_t np = (size_t)n_fields * d * k;
  double* grad = (double*)calloc(np ? np : 1, sizeof(double));
  slow_ada S;
  if (!grad || slow_ada_alloc(&S, np, d, cfg->eps)) return -1;
  for (int epoch = 0; epoch < max_iter; epoch++)
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      const double* x = Xd + (size_t)i * d;
      const double y_pred = slow_ffm_predict_row(x, d, field_of, k, P, w, *intercept);
      const double dL = orc_dloss(cfg->loss, cfg->loss_param, y[i], y_pred);
      memset(grad, 0, sizeof(double) * np);
      slow_ffm_compute_grad(x, d, field_of, k, P, dL, grad);
      slow_ada_update(&S, x, d, np, P, w, intercept, grad, dL, cfg, *it);
      (*it)++;
    }
  free(grad);
  slow_ada_free(&S);
  return 0;
}
