/*
 * oracle/nimfm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * "parity unpinned" against reference-run outputs; see nimfm_oracle.h.
 *
 * Faithful single-thread fp64 restatement of the reference's fast path: same
 * loop nests, same operation order, same lazy-scaling bookkeeping, on flat
 * arrays instead of Nim's jagged seq-of-seq.  Citations: /root/reference/.
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime under -std=c99 */
#include "nimfm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* wall-clock seconds of every epoch of the last *_fit call (bench.py's cpu_baseline leg reads them: the layout
 * transposes around the epoch loop, optimizer/sgd.nim:292,328, are per fit, not per epoch) */
#define ORC_MAX_TIMED_EPOCHS 64
double orc_epoch_seconds[ORC_MAX_TIMED_EPOCHS];
static double orc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------ */
/* losses: loss.nim:15-102                                             */
/* ------------------------------------------------------------------ */
double orc_loss(int loss, double param, double y, double p) {
  switch (loss) {
    case ORC_LOSS_SQUARED: { /* loss.nim:18 */
      double r = y - p;
      return 0.5 * (r * r);
    }
    case ORC_LOSS_SQUARED_HINGE: { /* loss.nim:33 */
      double z = 1 - p * y;
      double m = z > 0 ? z : 0;
      return m * m;
    }
    case ORC_LOSS_LOGISTIC: { /* loss.nim:54-59 */
      double z = p * y;
      if (z > 0) return log(1 + exp(-z));
      return log(exp(z) + 1) - z;
    }
    case ORC_LOSS_HUBER: { /* loss.nim:84-87 */
      double z = fabs(y - p);
      if (z < param) return 0.5 * (z * z);
      return param * (z - 0.5 * param);
    }
  }
  return NAN;
}

double orc_dloss(int loss, double param, double y, double p) {
  switch (loss) {
    case ORC_LOSS_SQUARED: /* loss.nim:21 */
      return p - y;
    case ORC_LOSS_SQUARED_HINGE: { /* loss.nim:36-39 */
      double z = 1 - p * y;
      if (z > 0) return -2 * y * z;
      return 0.0;
    }
    case ORC_LOSS_LOGISTIC: { /* loss.nim:62-67 */
      double z = p * y;
      if (z > 0) return -y * exp(-z) / (1 + exp(-z));
      return -y / (exp(z) + 1);
    }
    case ORC_LOSS_HUBER: { /* loss.nim:90-93 -- sign quirk reproduced as-is */
      double z = fabs(y - p);
      if (z < param) return y - p;
      return param;
    }
  }
  return NAN;
}

/* optimizer/sgd.nim:60-69 */
double orc_get_eta(int scheduling, double eta0, double power, double reg, int64_t it) {
  switch (scheduling) {
    case ORC_SCHED_CONSTANT: return eta0;
    case ORC_SCHED_OPTIMAL: return eta0 / pow(1.0 + eta0 * reg * (double)it, power);
    case ORC_SCHED_INVSCALING: return eta0 / pow((double)it, power);
    case ORC_SCHED_PEGASOS: return 1.0 / (reg * (double)it);
  }
  return NAN;
}

/* model/factorization_machine.nim:81-97 */
int orc_n_augments(int degree, int fit_lower, int fit_linear) {
  if (fit_lower == ORC_LOWER_AUGMENT) return fit_linear ? degree - 2 : degree - 1;
  return 0;
}
int orc_n_orders(int degree, int fit_lower) {
  if (degree == 1) return 0;
  if (fit_lower == ORC_LOWER_EXPLICIT) return degree - 1;
  return 1;
}

/* utils.nim:33 */
double orc_expit(double x) { return exp(fmin(0.0, x)) / (1.0 + exp(-fabs(x))); }

/* metrics.nim:5-13 */
double orc_rmse(const double* y_true, const double* y_score, int64_t n) {
  double r = 0.0;
  for (int64_t i = 0; i < n; i++) r += pow(y_score[i] - y_true[i], 2);
  return sqrt(r / (double)n);
}
static int sgn(double x) { return (x > 0) - (x < 0); }
/* model/fm_base.nim:47-48 + metrics.nim:39-47 */
double orc_accuracy_sign(const double* y_true, const double* y_score, int64_t n) {
  double r = 0.0;
  for (int64_t i = 0; i < n; i++) r += (double)(sgn(y_true[i]) == sgn(y_score[i]));
  return r / (double)n;
}
/* optimizer/utils.nim:56-59; norm(X,2) = sqrt(sum sq) (tensor/tensor.nim:583-616) */
double orc_regularization(const double* P, int64_t nP, const double* w, int64_t nw,
                          double intercept, double alpha0, double alpha, double beta) {
  double sw = 0.0, sp = 0.0;
  for (int64_t i = 0; i < nw; i++) sw += w[i] * w[i];
  for (int64_t i = 0; i < nP; i++) sp += P[i] * P[i];
  double nw2 = sqrt(sw), np2 = sqrt(sp);
  double r = 0.5 * alpha0 * (intercept * intercept) + 0.5 * alpha * (nw2 * nw2);
  r += 0.5 * beta * (np2 * np2);
  return r;
}

/* ------------------------------------------------------------------ */
/* decisionFunction: model/factorization_machine.nim:100-122           */
/* ------------------------------------------------------------------ */
int orc_fm_decision_function(const orc_csr* X, int degree, int k, int n_orders, int n_aug,
                             const double* P, const double* lams, const double* w,
                             double intercept, double* out) {
  const int64_t n = X->n, d = X->d, da = d + n_aug;
  double* A = (double*)malloc(sizeof(double) * (size_t)n * (size_t)(degree + 1));
  if (!A) return -1;
  /* linear(): kernels.nim:14-19 (rows WITHOUT dummies; nAugments is still 0) */
  for (int64_t i = 0; i < n; i++) {
    out[i] = 0.0;
    for (int64_t q = X->indptr[i]; q < X->indptr[i + 1]; q++) out[i] += w[X->indices[q]] * X->data[q];
  }
  for (int64_t i = 0; i < n; i++) out[i] += intercept; /* :109-110 */
  /* addDummyFeature(1.0, nAugments): rows now yield n_aug extra (d+t, 1.0) */
  for (int o = 0; o < n_orders; o++) {
    const int deg = degree - o;
    const double* Po = P + (size_t)o * k * da;
    for (int s = 0; s < k; s++) {
      const double* Ps = Po + (size_t)s * da;
      /* anova(): kernels.nim:46-64 */
      const int W = degree + 1;
      for (int64_t i = 0; i < n; i++) {
        for (int t = 1; t < deg + 1; t++) A[i * W + t] = 0.0;
        A[i * W + 0] = 1.0;
      }
      if (deg != 2) {
        for (int64_t i = 0; i < n; i++) {
          double* Ai = A + i * W;
          for (int64_t q = X->indptr[i]; q < X->indptr[i + 1] + n_aug; q++) {
            int64_t j;
            double val;
            if (q < X->indptr[i + 1]) { j = X->indices[q]; val = X->data[q]; }
            else { j = d + (q - X->indptr[i + 1]); val = 1.0; }
            for (int t = 0; t < deg; t++) Ai[deg - t] += Ai[deg - t - 1] * Ps[j] * val;
          }
        }
      } else {
        for (int64_t i = 0; i < n; i++) {
          double* Ai = A + i * W;
          for (int64_t q = X->indptr[i]; q < X->indptr[i + 1] + n_aug; q++) {
            int64_t j;
            double val;
            if (q < X->indptr[i + 1]) { j = X->indices[q]; val = X->data[q]; }
            else { j = d + (q - X->indptr[i + 1]); val = 1.0; }
            Ai[1] += Ps[j] * val;
            double pv = Ps[j] * val;
            Ai[2] += pv * pv;
          }
          Ai[2] = (Ai[1] * Ai[1] - Ai[2]) / 2.0;
        }
      }
      for (int64_t i = 0; i < n; i++) out[i] += lams[s] * A[i * W + deg]; /* :119-120 */
    }
  }
  free(A);
  return 0;
}

/* ------------------------------------------------------------------ */
/* shared pieces of the per-sample step (training layout P[o][j][s])    */
/* ------------------------------------------------------------------ */
typedef struct row_view {
  const int64_t* idx;
  const double* val;
  int64_t m; /* stored nnz; dummies (value 1.0 at d+t) follow virtually */
} row_view;

static inline row_view get_row(const orc_csr* X, int64_t i) {
  row_view r;
  r.idx = X->indices + X->indptr[i];
  r.val = X->data + X->indptr[i];
  r.m = X->indptr[i + 1] - X->indptr[i];
  return r;
}
#define ROW_J(r, q, d) ((q) < (r).m ? (r).idx[q] : (d) + ((q) - (r).m))
#define ROW_V(r, q) ((q) < (r).m ? (r).val[q] : 1.0)

/* optimizer/sgd.nim:146-173; Po is one order in training layout [j][s];
 * A is [k][W].  Row includes n_aug dummies. */
static double compute_anova(const double* Po, row_view r, int64_t d, int n_aug, int k, int deg,
                            double* A, int W) {
  double result = 0.0;
  if (deg != 2) {
    for (int s = 0; s < k; s++) {
      A[s * W + 0] = 1.0;
      for (int t = 1; t < deg + 1; t++) A[s * W + t] = 0;
    }
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, d);
      const double val = ROW_V(r, q);
      for (int s = 0; s < k; s++)
        for (int t = 0; t < deg; t++)
          A[s * W + deg - t] += A[s * W + deg - t - 1] * Po[j * k + s] * val;
    }
  } else {
    for (int s = 0; s < k; s++) { A[s * W + 0] = 1; A[s * W + 1] = 0; A[s * W + 2] = 0; }
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, d);
      const double val = ROW_V(r, q);
      for (int s = 0; s < k; s++) {
        A[s * W + 1] += val * Po[j * k + s];
        double vp = val * Po[j * k + s];
        A[s * W + 2] += vp * vp;
      }
    }
    for (int s = 0; s < k; s++) A[s * W + 2] = (A[s * W + 1] * A[s * W + 1] - A[s * W + 2]) / 2;
  }
  for (int s = 0; s < k; s++) result += A[s * W + deg];
  return result;
}

/* optimizer/sgd.nim:176-188; dAo is one order of dA, [j][s] */
static void compute_anova_derivative(const double* Po, row_view r, int64_t d, int n_aug, int k,
                                     int deg, const double* A, int W, double* dAo) {
  if (deg != 2) {
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, d);
      const double val = ROW_V(r, q);
      for (int s = 0; s < k; s++) {
        dAo[j * k + s] = val;
        for (int t = 1; t < deg; t++) dAo[j * k + s] = val * (A[s * W + t] - Po[j * k + s] * dAo[j * k + s]);
      }
    }
  } else {
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, d);
      const double val = ROW_V(r, q);
      for (int s = 0; s < k; s++) dAo[j * k + s] = val * (A[s * W + 1] - Po[j * k + s] * val);
    }
  }
}

/* optimizer/sgd.nim:191-202 */
static double predict_with_grad(row_view r, int64_t d, int n_aug, int k, int n_orders, int degree,
                                const double* Pt, const double* w, double intercept, double* A,
                                double* dA) {
  const int64_t da = d + n_aug;
  const int W = degree + 1;
  double result = intercept;
  for (int64_t q = 0; q < r.m; q++) result += w[r.idx[q]] * r.val[q];
  for (int o = 0; o < n_orders; o++) {
    const double* Po = Pt + (size_t)o * da * k;
    result += compute_anova(Po, r, d, n_aug, k, degree - o, A, W);
    compute_anova_derivative(Po, r, d, n_aug, k, degree - o, A, W, dA + (size_t)o * da * k);
  }
  return result;
}

/* optimizer/sgd.nim:92-96: P1[o][s][j] = P2[o][j][s] and the inverse */
static void to_train_layout(double* Pt, const double* P, int n_orders, int k, int64_t da) {
  for (int o = 0; o < n_orders; o++)
    for (int s = 0; s < k; s++)
      for (int64_t j = 0; j < da; j++) Pt[((size_t)o * da + j) * k + s] = P[((size_t)o * k + s) * da + j];
}
static void to_model_layout(double* P, const double* Pt, int n_orders, int k, int64_t da) {
  for (int o = 0; o < n_orders; o++)
    for (int64_t j = 0; j < da; j++)
      for (int s = 0; s < k; s++) P[((size_t)o * k + s) * da + j] = Pt[((size_t)o * da + j) * k + s];
}

/* ------------------------------------------------------------------ */
/* SGD: optimizer/sgd.nim                                               */
/* ------------------------------------------------------------------ */
typedef struct sgd_state {
  /* shared by FM (n_blocks = nOrders) and FFM (n_blocks = nFields) */
  int n_blocks, k;
  int64_t d, da; /* da = P.shape[1] of the training tensor */
  double* P;     /* [n_blocks][da][k] */
  double* w;     /* [d] */
  double* intercept;
  double scaling_P, scaling_w;
  double *scalings_P, *scalings_w;
  orc_sgd_cfg cfg;
  int64_t it;
} sgd_state;

/* optimizer/sgd.nim:99-113 */
static void sgd_finalize(sgd_state* S) {
  if (S->cfg.fit_linear) {
    for (int64_t j = 0; j < S->d; j++) S->w[j] *= S->scaling_w;
    for (int64_t j = 0; j < S->d; j++) S->w[j] /= S->scalings_w[j];
    S->scaling_w = 1.0;
    for (int64_t j = 0; j < S->d; j++) S->scalings_w[j] = 1.0;
  }
  for (int o = 0; o < S->n_blocks; o++)
    for (int64_t j = 0; j < S->da; j++)
      for (int s = 0; s < S->k; s++)
        S->P[((size_t)o * S->da + j) * S->k + s] *= S->scaling_P / S->scalings_P[j];
  for (int64_t j = 0; j < S->da; j++) S->scalings_P[j] = 1.0;
  S->scaling_P = 1.0;
}

/* optimizer/sgd.nim:116-131 */
static void sgd_reset_scaling(sgd_state* S) {
  if (S->cfg.fit_linear && S->scaling_w < 1e-9) {
    for (int64_t j = 0; j < S->d; j++) S->w[j] *= S->scaling_w;
    for (int64_t j = 0; j < S->d; j++) S->w[j] /= S->scalings_w[j];
    for (int64_t j = 0; j < S->d; j++) S->scalings_w[j] = 1.0;
    S->scaling_w = 1.0;
  }
  if (S->scaling_P < 1e-9) {
    for (int o = 0; o < S->n_blocks; o++)
      for (int64_t j = 0; j < S->d; j++) /* len(w): dummies skipped */
        for (int s = 0; s < S->k; s++)
          S->P[((size_t)o * S->da + j) * S->k + s] *= S->scaling_P / S->scalings_P[j];
    for (int64_t j = 0; j < S->da; j++) S->scalings_P[j] = 1.0;
    S->scaling_P = 1.0;
  }
}

/* optimizer/sgd.nim:134-143 (row WITHOUT dummies) */
static void sgd_lazily_update(sgd_state* S, row_view r) {
  for (int o = 0; o < S->n_blocks; o++)
    for (int64_t q = 0; q < r.m; q++) {
      const int64_t j = r.idx[q];
      for (int s = 0; s < S->k; s++)
        S->P[((size_t)o * S->da + j) * S->k + s] *= S->scaling_P / S->scalings_P[j];
    }
  if (S->cfg.fit_linear)
    for (int64_t q = 0; q < r.m; q++) {
      const int64_t j = r.idx[q];
      S->w[j] *= S->scaling_w / S->scalings_w[j];
    }
}

/* optimizer/sgd.nim:205-243 + fit_linear.nim:41-47 */
static double sgd_update(sgd_state* S, row_view r, int n_aug, const double* df, double yi,
                         double y_pred) {
  const orc_sgd_cfg* c = &S->cfg;
  double result = 0.0;
  const double dL = orc_dloss(c->loss, c->loss_param, yi, y_pred);
  const double eta_w = orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha, S->it);
  const double eta_P = orc_get_eta(c->scheduling, c->eta0, c->power, c->beta, S->it);
  for (int o = 0; o < S->n_blocks; o++)
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, S->d);
      for (int s = 0; s < S->k; s++) {
        const size_t e = ((size_t)o * S->da + j) * S->k + s;
        const double update = eta_P * (dL * df[e] + c->beta * S->P[e]);
        result += fabs(update);
        S->P[e] -= update;
      }
    }
  if (c->fit_intercept) {
    const double update =
        orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha0, S->it) * (dL + c->alpha0 * *S->intercept);
    result += fabs(update);
    *S->intercept -= update;
  }
  if (c->fit_linear) { /* fitLinearSGD */
    double res = 0.0;
    for (int64_t q = 0; q < r.m; q++) {
      const int64_t j = r.idx[q];
      const double update = eta_w * (dL * r.val[q] + c->alpha * S->w[j]);
      S->w[j] -= update;
      res += fabs(update);
    }
    result += res;
  }
  S->scaling_P *= (1 - eta_P * c->beta);
  S->scaling_w *= (1 - eta_w * c->alpha);
  for (int64_t q = 0; q < r.m; q++) {
    const int64_t j = r.idx[q];
    S->scalings_P[j] = S->scaling_P;
    S->scalings_w[j] = S->scaling_w;
  }
  for (int t = 0; t < n_aug; t++) S->scalings_P[S->da - n_aug + t] = S->scaling_P;
  sgd_reset_scaling(S);
  return result;
}

/* optimizer/sgd.nim:72-89; returns isContinue */
static int stopping_criterion(double loss_val, double viol, double tol, int* is_converged) {
  int result = 1;
  if (isnan(loss_val)) result = 0;
  if (viol < tol) { *is_converged = 1; result = 0; }
  return result;
}

static int sgd_state_init(sgd_state* S, int n_blocks, int k, int64_t d, int64_t da, double* w,
                          double* intercept, const orc_sgd_cfg* cfg, int64_t it) {
  memset(S, 0, sizeof(*S));
  S->n_blocks = n_blocks; S->k = k; S->d = d; S->da = da;
  S->w = w; S->intercept = intercept; S->cfg = *cfg; S->it = it;
  S->scaling_P = 1.0; S->scaling_w = 1.0;
  S->scalings_P = (double*)malloc(sizeof(double) * (size_t)(da > 0 ? da : 1));
  S->scalings_w = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
  if (!S->scalings_P || !S->scalings_w) return -1;
  for (int64_t j = 0; j < da; j++) S->scalings_P[j] = 1.0;
  for (int64_t j = 0; j < d; j++) S->scalings_w[j] = 1.0;
  return 0;
}

/* optimizer/sgd.nim:246-258 */
static void fm_sgd_step(sgd_state* S, const orc_csr* X, int64_t i, double yi, int degree, int n_aug,
                        double* A, double* dA, double* running_loss, double* viol) {
  row_view r = get_row(X, i);
  sgd_lazily_update(S, r);
  const double y_pred =
      predict_with_grad(r, S->d, n_aug, S->k, S->n_blocks, degree, S->P, S->w, *S->intercept, A, dA);
  *running_loss += orc_loss(S->cfg.loss, S->cfg.loss_param, yi, y_pred);
  *viol += sgd_update(S, r, n_aug, dA, yi, y_pred);
}

int orc_fm_sgd_fit(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                   double* P, double* w, double* intercept, const orc_sgd_cfg* cfg, int max_iter,
                   double tol, const int64_t* perms, int64_t* it, double* epoch_loss,
                   double* epoch_viol, int* n_epochs_run) {
  const int64_t n = X->n, d = X->d, da = d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  double* A = (double*)calloc((size_t)k * (degree + 1), sizeof(double));
  sgd_state S;
  if (!Pt || !dA || !A || sgd_state_init(&S, n_orders, k, d, da, w, intercept, cfg, *it)) return -1;
  S.P = Pt;
  to_train_layout(Pt, P, n_orders, k, da); /* sgd.nim:292 */
  int is_converged = 0, epochs = 0;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    double viol = 0.0, running_loss = 0.0;
    const double t_epoch = orc_now();
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      fm_sgd_step(&S, X, i, y[i], degree, n_aug, A, dA, &running_loss, &viol);
      S.it++; /* sgd.nim:308 */
    }
    if (epoch < ORC_MAX_TIMED_EPOCHS) orc_epoch_seconds[epoch] = orc_now() - t_epoch;
    running_loss /= (double)n;
    if (epoch_loss) epoch_loss[epoch] = running_loss;
    if (epoch_viol) epoch_viol[epoch] = viol;
    epochs = epoch + 1;
    if (!stopping_criterion(running_loss, viol, tol, &is_converged)) break;
  }
  sgd_finalize(&S); /* sgd.nim:327-328 */
  to_model_layout(P, Pt, n_orders, k, da);
  *it = S.it;
  if (n_epochs_run) *n_epochs_run = epochs;
  free(Pt); free(dA); free(A); free(S.scalings_P); free(S.scalings_w);
  return 0;
}

/* ---- Hogwild: optimizer/sgd_multi.nim:13-120 (racy on purpose) ---- */
typedef struct hog_arg {
  sgd_state* S;
  const orc_csr* X;
  const double* y;
  const int64_t* order; /* this epoch's indices or NULL */
  int64_t s, t;
  int degree, n_aug;
  double loss, viol;
} hog_arg;

static void* hog_worker(void* p) {
  hog_arg* a = (hog_arg*)p;
  sgd_state* S = a->S;
  const size_t np = (size_t)S->n_blocks * S->da * S->k;
  /* threadvar A, dA (sgd_multi.nim:8-10,28-31) */
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  double* A = (double*)calloc((size_t)S->k * (a->degree + 1), sizeof(double));
  a->loss = 0.0; a->viol = 0.0;
  for (int64_t ii = a->s; ii < a->t; ii++) {
    const int64_t i = a->order ? a->order[ii] : ii;
    fm_sgd_step(S, a->X, i, a->y[i], a->degree, a->n_aug, A, dA, &a->loss, &a->viol);
    S->it++; /* racy: sgd_multi.nim:37 */
  }
  free(dA); free(A);
  return NULL;
}

int orc_fm_sgd_fit_hogwild(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                           int n_aug, double* P, double* w, double* intercept,
                           const orc_sgd_cfg* cfg, int max_iter, double tol, const int64_t* perms,
                           int64_t* it, int n_threads, double* epoch_loss, double* epoch_viol,
                           int* n_epochs_run) {
  const int64_t n = X->n, d = X->d, da = d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  if (n_threads < 1) n_threads = 1;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  sgd_state S;
  if (!Pt || sgd_state_init(&S, n_orders, k, d, da, w, intercept, cfg, *it)) return -1;
  S.P = Pt;
  to_train_layout(Pt, P, n_orders, k, da);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
  hog_arg* args = (hog_arg*)malloc(sizeof(hog_arg) * n_threads);
  int64_t* borders = (int64_t*)malloc(sizeof(int64_t) * (n_threads + 1));
  int is_converged = 0, epochs = 0;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    double viol = 0.0, running_loss = 0.0;
    const double t_epoch = orc_now();
    /* in-memory CSR: nCached == nSamples, one pass of the while loop (sgd_multi.nim:83-101) */
    borders[0] = 0;
    for (int t = 0; t < n_threads; t++) borders[t + 1] = borders[t] + n / n_threads;
    borders[n_threads] = n;
    for (int t = 0; t < n_threads; t++) {
      args[t].S = &S; args[t].X = X; args[t].y = y;
      args[t].order = perms ? perms + (size_t)epoch * n : NULL;
      args[t].s = borders[t]; args[t].t = borders[t + 1];
      args[t].degree = degree; args[t].n_aug = n_aug;
      pthread_create(&th[t], NULL, hog_worker, &args[t]);
    }
    for (int t = 0; t < n_threads; t++) {
      pthread_join(th[t], NULL);
      running_loss += args[t].loss;
      viol += args[t].viol;
    }
    if (epoch < ORC_MAX_TIMED_EPOCHS) orc_epoch_seconds[epoch] = orc_now() - t_epoch;
    running_loss /= (double)n;
    if (epoch_loss) epoch_loss[epoch] = running_loss;
    if (epoch_viol) epoch_viol[epoch] = viol;
    epochs = epoch + 1;
    if (!stopping_criterion(running_loss, viol, tol, &is_converged)) break;
  }
  sgd_finalize(&S);
  to_model_layout(P, Pt, n_orders, k, da);
  *it = S.it;
  if (n_epochs_run) *n_epochs_run = epochs;
  free(Pt); free(th); free(args); free(borders); free(S.scalings_P); free(S.scalings_w);
  return 0;
}

/* ------------------------------------------------------------------ */
/* AdaGrad: optimizer/adagrad.nim                                       */
/* ------------------------------------------------------------------ */
typedef struct ada_state {
  int n_blocks, k;
  int64_t d, da;
  double* P; /* [n_blocks][da][k] */
  double* w;
  double* intercept;
  double *gsum_P, *gnorm_P, *gsum_w, *gnorm_w, *gsum_b, *gnorm_b;
  orc_adagrad_cfg cfg;
  int64_t it;
} ada_state;

/* optimizer/adagrad.nim:47-59 (state (re)creation when it == 1) */
static void ada_init(ada_state* S) {
  if (S->it == 1) {
    const size_t np = (size_t)S->n_blocks * S->da * S->k;
    for (size_t e = 0; e < np; e++) { S->gsum_P[e] = 0.0; S->gnorm_P[e] = S->cfg.eps; }
    for (int64_t j = 0; j < S->d; j++) { S->gsum_w[j] = 0.0; S->gnorm_w[j] = S->cfg.eps; }
    *S->gsum_b = 0.0;
    *S->gnorm_b = S->cfg.eps;
  }
}

/* optimizer/adagrad.nim:65-84 */
static void ada_finalize(ada_state* S) {
  const orc_adagrad_cfg* c = &S->cfg;
  const double it = (double)(S->it - 1);
  const double denom = c->eta0 * it * c->beta;
  const size_t np = (size_t)S->n_blocks * S->da * S->k;
  for (size_t e = 0; e < np; e++) {
    S->P[e] = -c->eta0 * S->gsum_P[e];
    S->P[e] /= denom + sqrt(S->gnorm_P[e]);
  }
  if (c->fit_intercept) {
    const double den = sqrt(*S->gnorm_b) + c->eta0 * it * c->alpha0;
    *S->intercept = -c->eta0 * *S->gsum_b / den;
  }
  if (c->fit_linear) {
    const double den = c->eta0 * it * c->alpha;
    for (int64_t j = 0; j < S->d; j++) {
      S->w[j] = -c->eta0 * S->gsum_w[j];
      S->w[j] /= den + sqrt(S->gnorm_w[j]);
    }
  }
}

/* optimizer/adagrad.nim:87-110 + fit_linear.nim:50-57 */
static double ada_update(ada_state* S, row_view r, int n_aug) {
  const orc_adagrad_cfg* c = &S->cfg;
  double result = 0.0;
  const double it = (double)(S->it - 1);
  const double tmp = c->eta0 * it * c->beta;
  for (int o = 0; o < S->n_blocks; o++)
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, S->d);
      for (int s = 0; s < S->k; s++) {
        const size_t e = ((size_t)o * S->da + j) * S->k + s;
        const double pjs = S->P[e];
        const double denom = tmp + sqrt(S->gnorm_P[e]);
        S->P[e] = -(c->eta0 * S->gsum_P[e]) / denom;
        result += fabs(pjs - S->P[e]);
      }
    }
  if (c->fit_intercept) {
    const double old = *S->intercept;
    const double denom = sqrt(*S->gnorm_b) + c->eta0 * it * c->alpha0;
    *S->intercept = -c->eta0 * *S->gsum_b / denom;
    result += fabs(old - *S->intercept);
  }
  if (c->fit_linear) { /* fitLinearAdaGrad(w, g_sum_w, g_norms_w, X, i, alpha, eta0, it) */
    double res = 0.0;
    const double denom = it * c->eta0 * c->alpha;
    for (int64_t q = 0; q < r.m; q++) {
      const int64_t j = r.idx[q];
      const double wj = S->w[j];
      S->w[j] = -c->eta0 * S->gsum_w[j] / (denom + sqrt(S->gnorm_w[j]));
      res += fabs(wj - S->w[j]);
    }
    result += res;
  }
  return result;
}

/* optimizer/adagrad.nim:113-134 */
static void ada_update_g(ada_state* S, row_view r, int n_aug, const double* df, double yi,
                         double y_pred) {
  const orc_adagrad_cfg* c = &S->cfg;
  const double dL = orc_dloss(c->loss, c->loss_param, yi, y_pred);
  for (int o = 0; o < S->n_blocks; o++)
    for (int64_t q = 0; q < r.m + n_aug; q++) {
      const int64_t j = ROW_J(r, q, S->d);
      for (int s = 0; s < S->k; s++) {
        const size_t e = ((size_t)o * S->da + j) * S->k + s;
        const double grad = dL * df[e];
        S->gsum_P[e] += grad;
        S->gnorm_P[e] += grad * grad;
      }
    }
  if (c->fit_intercept) {
    *S->gsum_b += dL;
    *S->gnorm_b += dL * dL;
  }
  if (c->fit_linear)
    for (int64_t q = 0; q < r.m; q++) {
      const int64_t j = r.idx[q];
      S->gsum_w[j] += dL * r.val[q];
      const double g = dL * r.val[q];
      S->gnorm_w[j] += g * g;
    }
}

int orc_fm_adagrad_fit(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                       double* P, double* w, double* intercept, const orc_adagrad_cfg* cfg,
                       int max_iter, double tol, const int64_t* perms, int64_t* it,
                       double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                       double* gsum_b, double* gnorm_b, double* epoch_loss, double* epoch_viol,
                       int* n_epochs_run) {
  const int64_t n = X->n, d = X->d, da = d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  double* A = (double*)calloc((size_t)k * (degree + 1), sizeof(double));
  if (!Pt || !dA || !A) return -1;
  ada_state S;
  memset(&S, 0, sizeof(S));
  S.n_blocks = n_orders; S.k = k; S.d = d; S.da = da; S.P = Pt; S.w = w; S.intercept = intercept;
  S.gsum_P = gsum_P; S.gnorm_P = gnorm_P; S.gsum_w = gsum_w; S.gnorm_w = gnorm_w;
  S.gsum_b = gsum_b; S.gnorm_b = gnorm_b; S.cfg = *cfg; S.it = *it;
  ada_init(&S);
  to_train_layout(Pt, P, n_orders, k, da); /* adagrad.nim:162 */
  int is_converged = 0, epochs = 0;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    double viol = 0.0, running_loss = 0.0;
    const double t_epoch = orc_now();
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      row_view r = get_row(X, i);
      if (S.it != 1) viol += ada_update(&S, r, n_aug); /* adagrad.nim:171-173 */
      const double y_pred =
          predict_with_grad(r, d, n_aug, k, n_orders, degree, Pt, w, *intercept, A, dA);
      running_loss += orc_loss(cfg->loss, cfg->loss_param, y[i], y_pred);
      ada_update_g(&S, r, n_aug, dA, y[i], y_pred);
      S.it++;
    }
    if (epoch < ORC_MAX_TIMED_EPOCHS) orc_epoch_seconds[epoch] = orc_now() - t_epoch;
    running_loss /= (double)n;
    if (epoch_loss) epoch_loss[epoch] = running_loss;
    if (epoch_viol) epoch_viol[epoch] = viol;
    epochs = epoch + 1;
    if (!stopping_criterion(running_loss, viol, tol, &is_converged)) break;
  }
  ada_finalize(&S); /* adagrad.nim:202-203 */
  to_model_layout(P, Pt, n_orders, k, da);
  *it = S.it;
  if (n_epochs_run) *n_epochs_run = epochs;
  free(Pt); free(dA); free(A);
  return 0;
}

/* ------------------------------------------------------------------ */
/* FFM: model/field_aware_factorization_machine.nim, optimizer/{sgd,adagrad}_ffm.nim */
/* ------------------------------------------------------------------ */
int orc_ffm_decision_function(const orc_csr* X, int k, const double* P, const double* w,
                              double intercept, double* out) {
  const int64_t n = X->n, d = X->d;
  for (int64_t i = 0; i < n; i++) out[i] = intercept; /* :66 */
  for (int64_t i = 0; i < n; i++)                     /* :68-70 */
    for (int64_t q = X->indptr[i]; q < X->indptr[i + 1]; q++) out[i] += X->data[q] * w[X->indices[q]];
  for (int64_t i = 0; i < n; i++) /* :72-76 */
    for (int64_t q1 = X->indptr[i]; q1 < X->indptr[i + 1]; q1++)
      for (int64_t q2 = X->indptr[i]; q2 < X->indptr[i + 1]; q2++) {
        const int64_t j1 = X->indices[q1], j2 = X->indices[q2];
        if (j1 < j2) {
          const int64_t f1 = X->fields[q1], f2 = X->fields[q2];
          const double* a = P + ((size_t)f2 * d + j1) * k;
          const double* b = P + ((size_t)f1 * d + j2) * k;
          double dot = 0.0; /* tensor/tensor.nim:686-692 */
          for (int s = 0; s < k; s++) dot += a[s] * b[s];
          out[i] += X->data[q1] * X->data[q2] * dot;
        }
      }
  return 0;
}

/* optimizer/sgd_ffm.nim:11-30 */
static double ffm_predict_with_grad(const orc_csr* X, int64_t i, int F, int k, const double* P,
                                    const double* w, double intercept, double* dA) {
  const int64_t d = X->d;
  const int64_t b0 = X->indptr[i], b1 = X->indptr[i + 1];
  double result = intercept;
  for (int64_t q = b0; q < b1; q++) result += w[X->indices[q]] * X->data[q];
  for (int f = 0; f < F; f++)
    for (int64_t q = b0; q < b1; q++)
      for (int s = 0; s < k; s++) dA[((size_t)f * d + X->indices[q]) * k + s] = 0.0;
  for (int64_t q1 = b0; q1 < b1; q1++)
    for (int64_t q2 = b0; q2 < b1; q2++) {
      const int64_t j1 = X->indices[q1], j2 = X->indices[q2];
      if (j1 < j2) {
        const int64_t f1 = X->fields[q1], f2 = X->fields[q2];
        const double val1 = X->data[q1], val2 = X->data[q2];
        const double* a = P + ((size_t)f2 * d + j1) * k;
        const double* b = P + ((size_t)f1 * d + j2) * k;
        double tmp = 0.0;
        for (int s = 0; s < k; s++) tmp += a[s] * b[s];
        result += tmp * val1 * val2;
        for (int s = 0; s < k; s++) {
          dA[((size_t)f2 * d + j1) * k + s] += val1 * val2 * b[s];
          dA[((size_t)f1 * d + j2) * k + s] += val1 * val2 * a[s];
        }
      }
    }
  return result;
}

int orc_ffm_sgd_fit(const orc_csr* X, const double* y, int k, double* P, double* w,
                    double* intercept, const orc_sgd_cfg* cfg, int max_iter, double tol,
                    const int64_t* perms, int64_t* it, double* epoch_loss, double* epoch_viol,
                    int* n_epochs_run) {
  const int64_t n = X->n, d = X->d;
  const int F = (int)X->n_fields;
  const size_t np = (size_t)F * d * k;
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  sgd_state S;
  if (!dA || sgd_state_init(&S, F, k, d, d, w, intercept, cfg, *it)) return -1;
  S.P = P; /* FFM trains in place: P is already [f][j][s] (sgd_ffm.nim:78) */
  int is_converged = 0, epochs = 0;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    const double t_epoch = orc_now();
    double viol = 0.0, running_loss = 0.0;
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      row_view r = get_row(X, i);
      sgd_lazily_update(&S, r); /* sgd_ffm.nim:39-40 */
      const double y_pred = ffm_predict_with_grad(X, i, F, k, P, w, *intercept, dA);
      running_loss += orc_loss(cfg->loss, cfg->loss_param, y[i], y_pred);
      viol += sgd_update(&S, r, 0, dA, y[i], y_pred);
      S.it++;
    }
    if (epoch < ORC_MAX_TIMED_EPOCHS) orc_epoch_seconds[epoch] = orc_now() - t_epoch;
    running_loss /= (double)n;
    if (epoch_loss) epoch_loss[epoch] = running_loss;
    if (epoch_viol) epoch_viol[epoch] = viol;
    epochs = epoch + 1;
    if (!stopping_criterion(running_loss, viol, tol, &is_converged)) break;
  }
  sgd_finalize(&S);
  *it = S.it;
  if (n_epochs_run) *n_epochs_run = epochs;
  free(dA); free(S.scalings_P); free(S.scalings_w);
  return 0;
}

int orc_ffm_adagrad_fit(const orc_csr* X, const double* y, int k, double* P, double* w,
                        double* intercept, const orc_adagrad_cfg* cfg, int max_iter, double tol,
                        const int64_t* perms, int64_t* it, double* gsum_P, double* gnorm_P,
                        double* gsum_w, double* gnorm_w, double* gsum_b, double* gnorm_b,
                        double* epoch_loss, double* epoch_viol, int* n_epochs_run) {
  const int64_t n = X->n, d = X->d;
  const int F = (int)X->n_fields;
  const size_t np = (size_t)F * d * k;
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  if (!dA) return -1;
  ada_state S;
  memset(&S, 0, sizeof(S));
  S.n_blocks = F; S.k = k; S.d = d; S.da = d; S.P = P; S.w = w; S.intercept = intercept;
  S.gsum_P = gsum_P; S.gnorm_P = gnorm_P; S.gsum_w = gsum_w; S.gnorm_w = gnorm_w;
  S.gsum_b = gsum_b; S.gnorm_b = gnorm_b; S.cfg = *cfg; S.it = *it;
  ada_init(&S);
  int is_converged = 0, epochs = 0;
  for (int epoch = 0; epoch < max_iter; epoch++) {
    const double t_epoch = orc_now();
    double viol = 0.0, running_loss = 0.0;
    for (int64_t ii = 0; ii < n; ii++) {
      const int64_t i = perms ? perms[(size_t)epoch * n + ii] : ii;
      row_view r = get_row(X, i);
      if (S.it != 1) viol += ada_update(&S, r, 0);
      const double y_pred = ffm_predict_with_grad(X, i, F, k, P, w, *intercept, dA);
      running_loss += orc_loss(cfg->loss, cfg->loss_param, y[i], y_pred);
      ada_update_g(&S, r, 0, dA, y[i], y_pred);
      S.it++;
    }
    if (epoch < ORC_MAX_TIMED_EPOCHS) orc_epoch_seconds[epoch] = orc_now() - t_epoch;
    running_loss /= (double)n;
    if (epoch_loss) epoch_loss[epoch] = running_loss;
    if (epoch_viol) epoch_viol[epoch] = viol;
    epochs = epoch + 1;
    if (!stopping_criterion(running_loss, viol, tol, &is_converged)) break;
  }
  ada_finalize(&S);
  *it = S.it;
  if (n_epochs_run) *n_epochs_run = epochs;
  free(dA);
  return 0;
}
