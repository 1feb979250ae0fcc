/*
 * oracle/nimfm_jagged.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE ("parity unpinned", see nimfm_oracle.h).
 *
 * The reference keeps its tensors as Nim seq-of-seq-of-seq (tensor/tensor.nim:8-17): in the training layout every
 * row P[order][j] is its OWN heap block behind two pointer loads.  nimfm_oracle.c restates the
 * arithmetic on flat arrays; this file restates the SAME single-order degree-2 SGD epoch (optimizer/sgd.nim:134-258,
 * fit_linear.nim:41-47) on jagged storage -- one malloc per row with a 16-byte seq header in front, rows reached through
 * a pointer table -- so that bench.py's cpu_baseline can quote both layouts.  Single-threaded
 * it must give bit-identical results to orc_fm_sgd_fit (tests/test_oracle_sgd.py); with n_threads > 1 it is the racy
 * Hogwild driver of optimizer/sgd_multi.nim:21-37,83-101 on the same storage.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nimfm_oracle.h"

extern double orc_epoch_seconds[];
double orc_loss(int loss, double param, double y, double p);
double orc_dloss(int loss, double param, double y, double p);
double orc_get_eta(int scheduling, double eta0, double power, double reg, int64_t it);

static double jag_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct jag_state {
  int k;
  int64_t d;
  double** P;  /* [d] -> row of k doubles (training layout, one order) */
  double* w;
  double* intercept;
  double scaling_P, scaling_w;
  double *scalings_P, *scalings_w;
  orc_sgd_cfg cfg;
  int64_t it;
} jag_state;

static double** jag_alloc(int64_t d, int k) {
  double** rows = (double**)malloc(sizeof(double*) * (size_t)(d > 0 ? d : 1));
  if (!rows) return NULL;
  for (int64_t j = 0; j < d; j++) {
    char* blk = (char*)malloc(16 + sizeof(double) * (size_t)k); /* a Nim seq: {len, cap} header, then the payload */
    if (!blk) return NULL;
    rows[j] = (double*)(blk + 16);
    memset(rows[j], 0, sizeof(double) * (size_t)k);
  }
  return rows;
}
static void jag_free(double** rows, int64_t d) {
  for (int64_t j = 0; j < d; j++) free((char*)rows[j] - 16);
  free(rows);
}

/* one step: optimizer/sgd.nim:246-258 (lazilyUpdate, predictWithGrad, update), degree 2, no dummy features */
static void jag_step(jag_state* S, const orc_csr* X, int64_t i, double yi, double* A, double** dA, double* loss, double* viol) {
  const orc_sgd_cfg* c = &S->cfg;
  const int k = S->k;
  const int64_t q0 = X->indptr[i], m = X->indptr[i + 1] - q0;
  const int64_t* idx = X->indices + q0;
  const double* val = X->data + q0;
  /* lazilyUpdate, sgd.nim:134-143 */
  for (int64_t q = 0; q < m; q++) {
    double* row = S->P[idx[q]];
    for (int s = 0; s < k; s++) row[s] *= S->scaling_P / S->scalings_P[idx[q]];
  }
  if (c->fit_linear)
    for (int64_t q = 0; q < m; q++) S->w[idx[q]] *= S->scaling_w / S->scalings_w[idx[q]];
  /* predictWithGrad, sgd.nim:191-202 -> computeAnova (:160-170), computeAnovaDerivative (:185-188) */
  double y_pred = *S->intercept;
  for (int64_t q = 0; q < m; q++) y_pred += S->w[idx[q]] * val[q];
  for (int s = 0; s < k; s++) { A[3 * s] = 1; A[3 * s + 1] = 0; A[3 * s + 2] = 0; }
  for (int64_t q = 0; q < m; q++) {
    const double* row = S->P[idx[q]];
    for (int s = 0; s < k; s++) {
      A[3 * s + 1] += val[q] * row[s];
      const double vp = val[q] * row[s];
      A[3 * s + 2] += vp * vp;
    }
  }
  for (int s = 0; s < k; s++) A[3 * s + 2] = (A[3 * s + 1] * A[3 * s + 1] - A[3 * s + 2]) / 2;
  double ker = 0.0;
  for (int s = 0; s < k; s++) ker += A[3 * s + 2];
  y_pred += ker;
  for (int64_t q = 0; q < m; q++) {
    const double* row = S->P[idx[q]];
    double* drow = dA[idx[q]];
    for (int s = 0; s < k; s++) drow[s] = val[q] * (A[3 * s + 1] - row[s] * val[q]);
  }
  *loss += orc_loss(c->loss, c->loss_param, yi, y_pred);
  /* update, sgd.nim:205-243 */
  double result = 0.0;
  const double dL = orc_dloss(c->loss, c->loss_param, yi, y_pred);
  const double eta_w = orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha, S->it);
  const double eta_P = orc_get_eta(c->scheduling, c->eta0, c->power, c->beta, S->it);
  for (int64_t q = 0; q < m; q++) {
    double* row = S->P[idx[q]];
    const double* drow = dA[idx[q]];
    for (int s = 0; s < k; s++) {
      const double update = eta_P * (dL * drow[s] + c->beta * row[s]);
      result += fabs(update);
      row[s] -= update;
    }
  }
  if (c->fit_intercept) {
    const double update = orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha0, S->it) * (dL + c->alpha0 * *S->intercept);
    result += fabs(update);
    *S->intercept -= update;
  }
  if (c->fit_linear) {
    double res = 0.0;
    for (int64_t q = 0; q < m; q++) {
      const double update = eta_w * (dL * val[q] + c->alpha * S->w[idx[q]]);
      S->w[idx[q]] -= update;
      res += fabs(update);
    }
    result += res;
  }
  S->scaling_P *= (1 - eta_P * c->beta);
  S->scaling_w *= (1 - eta_w * c->alpha);
  for (int64_t q = 0; q < m; q++) {
    S->scalings_P[idx[q]] = S->scaling_P;
    S->scalings_w[idx[q]] = S->scaling_w;
  }
  /* resetScaling, sgd.nim:116-131 */
  if (c->fit_linear && S->scaling_w < 1e-9) {
    for (int64_t j = 0; j < S->d; j++) S->w[j] *= S->scaling_w;
    for (int64_t j = 0; j < S->d; j++) S->w[j] /= S->scalings_w[j];
    for (int64_t j = 0; j < S->d; j++) S->scalings_w[j] = 1.0;
    S->scaling_w = 1.0;
  }
  if (S->scaling_P < 1e-9) {
    for (int64_t j = 0; j < S->d; j++)
      for (int s = 0; s < k; s++) S->P[j][s] *= S->scaling_P / S->scalings_P[j];
    for (int64_t j = 0; j < S->d; j++) S->scalings_P[j] = 1.0;
    S->scaling_P = 1.0;
  }
  *viol += result;
}

typedef struct jag_arg {
  jag_state* S;
  const orc_csr* X;
  const double* y;
  const int64_t* order;
  int64_t s, t;
  double loss, viol;
} jag_arg;

static void* jag_worker(void* p) {
  jag_arg* a = (jag_arg*)p;
  jag_state* S = a->S;
  double* A = (double*)calloc((size_t)S->k * 3, sizeof(double));
  double** dA = jag_alloc(S->d, S->k); /* threadvar dA (sgd_multi.nim:8-10,28-31) */
  a->loss = 0.0;
  a->viol = 0.0;
  for (int64_t ii = a->s; ii < a->t; ii++) {
    const int64_t i = a->order ? a->order[ii] : ii;
    jag_step(S, a->X, i, a->y[i], A, dA, &a->loss, &a->viol);
    S->it++; /* racy with several threads: sgd_multi.nim:37 */
  }
  free(A);
  jag_free(dA, S->d);
  return NULL;
}

/* P: model layout [1][k][d] in/out (sgd.nim:292,328 transposes around the loop, as here) */
int orc_fm_sgd_fit_jagged(const orc_csr* X, const double* y, int k, double* P, double* w, double* intercept,
                          const orc_sgd_cfg* cfg, int max_iter, const int64_t* perms, int64_t* it, int n_threads,
                          double* epoch_loss, double* epoch_viol) {
  const int64_t n = X->n, d = X->d;
  if (n_threads < 1) n_threads = 1;
  jag_state S;
  memset(&S, 0, sizeof(S));
  S.k = k; S.d = d; S.w = w; S.intercept = intercept; S.cfg = *cfg; S.it = *it;
  S.scaling_P = 1.0; S.scaling_w = 1.0;
  S.P = jag_alloc(d, k);
  S.scalings_P = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
  S.scalings_w = (double*)malloc(sizeof(double) * (size_t)(d > 0 ? d : 1));
  if (!S.P || !S.scalings_P || !S.scalings_w) return -1;
  for (int64_t j = 0; j < d; j++) { S.scalings_P[j] = 1.0; S.scalings_w[j] = 1.0; }
  for (int s = 0; s < k; s++)
    for (int64_t j = 0; j < d; j++) S.P[j][s] = P[(size_t)s * d + j];
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
  jag_arg* args = (jag_arg*)malloc(sizeof(jag_arg) * n_threads);
  for (int epoch = 0; epoch < max_iter; epoch++) {
    double viol = 0.0, running_loss = 0.0;
    const double t_epoch = jag_now();
    for (int t = 0; t < n_threads; t++) {
      args[t].S = &S; args[t].X = X; args[t].y = y;
      args[t].order = perms ? perms + (size_t)epoch * n : NULL;
      args[t].s = (int64_t)t * (n / n_threads);
      args[t].t = t == n_threads - 1 ? n : (int64_t)(t + 1) * (n / n_threads);
      if (n_threads > 1) pthread_create(&th[t], NULL, jag_worker, &args[t]);
      else jag_worker(&args[t]);
    }
    for (int t = 0; t < n_threads; t++) {
      if (n_threads > 1) pthread_join(th[t], NULL);
      running_loss += args[t].loss;
      viol += args[t].viol;
    }
    if (epoch < 64) orc_epoch_seconds[epoch] = jag_now() - t_epoch;
    if (epoch_loss) epoch_loss[epoch] = running_loss / (double)n;
    if (epoch_viol) epoch_viol[epoch] = viol;
  }
  /* finalize, sgd.nim:99-113 */
  if (cfg->fit_linear) {
    for (int64_t j = 0; j < d; j++) w[j] *= S.scaling_w;
    for (int64_t j = 0; j < d; j++) w[j] /= S.scalings_w[j];
  }
  for (int64_t j = 0; j < d; j++)
    for (int s = 0; s < k; s++) S.P[j][s] *= S.scaling_P / S.scalings_P[j];
  for (int64_t j = 0; j < d; j++)
    for (int s = 0; s < k; s++) P[(size_t)s * d + j] = S.P[j][s];
  *it = S.it;
  jag_free(S.P, d);
  free(S.scalings_P); free(S.scalings_w); free(th); free(args);
  return 0;
}
