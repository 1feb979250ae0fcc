/*
 * oracle/nimfm_mb.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of THIS REPOSITORY's deterministic mini-batch rule
 *  -- not a reference function.  The reference updates
 * after every sample (optimizer/sgd.nim:298-308); its own parallel mode is a
 * racy Hogwild (optimizer/sgd_multi.nim:83-101, README.md:56-58) with no
 * reproducible result.  The HIP throughput kernels instead implement:
 *
 *   all samples of a batch see the parameters as of the batch start; their
 *   per-sample updates (the reference's per-sample expressions, each with its
 *   own step counter `it`) are combined per coordinate in sample order -- SGD:
 *   averaged over the samples that touch the coordinate, AdaGrad: summed into
 *   the (additive) state; L2 decay is the product of the per-step factors
 *   (1 - eta_t * reg) of the batch (what the reference's lazy scaling amounts
 *   to).  Exact statements precede sgd_epoch_mb and ada_epoch_mb below.
 *
 * With batch == 1 this IS the reference's sequential step in exact
 * arithmetic (tests/test_oracle_mb.py checks that against nimfm_oracle.c).
 * The per-sample forward/gradient code is shared with nimfm_oracle.c by
 * textual inclusion so the two cannot drift.
 */
#include "nimfm_oracle.c"

/* touch cap of the SGD rule (1: the per-coordinate MEAN of the batch's steps; larger: up to that many of them are
 * SUMMED, the rest averaged in) -- see sgd_epoch_mb */
double orc_mb_touch_cap = 1.0;
/* AdaGrad rule (round 5, nfm_opt_set_ada_cross in the library): a coordinate's g_norm grows by the batch's
 *   sum_i g_i^2 + orc_mb_ada_cross * max((sum_i g_i)^2 - sum_i g_i^2, 0)
 * -- the cross products of the samples' gradients, all taken from the batch-start parameters (0: the samples' squares alone,
 * the rule of rounds 1-4; one touch: nothing changes) */
double orc_mb_ada_cross = 0.0;
static double mb_norm_inc(double acc, double accn) {
  const double c = acc * acc - accn;
  return accn + (orc_mb_ada_cross != 0.0 && c > 0.0 ? orc_mb_ada_cross * c : 0.0);
}

/* EXPERIMENT (tests/sgd_agreement_proto.py; no counterpart in the library): instead of the touch cap, a coordinate's summed steps
 * are divided by 1 + (c - 1) rho, rho = the samples' agreement on it, ((sum s_i)^2 - sum s_i^2) / ((c - 1) sum s_i^2) clamped to
 * [0, 1] -- 1: all steps equal (the mean), 0: uncorrelated (the sum).  0: off */
double orc_mb_sgd_agree = 0.0;

typedef double (*mb_predict_fn)(const orc_csr* X, int64_t i, int n_blocks, int k, int degree,
                                int n_aug, const double* Pt, const double* w, double intercept,
                                double* A, double* dA);

static double mb_predict_fm(const orc_csr* X, int64_t i, int n_blocks, int k, int degree, int n_aug,
                            const double* Pt, const double* w, double intercept, double* A,
                            double* dA) {
  return predict_with_grad(get_row(X, i), X->d, n_aug, k, n_blocks, degree, Pt, w, intercept, A, dA);
}
static double mb_predict_ffm(const orc_csr* X, int64_t i, int n_blocks, int k, int degree, int n_aug,
                             const double* Pt, const double* w, double intercept, double* A,
                             double* dA) {
  (void)degree; (void)n_aug; (void)A;
  return ffm_predict_with_grad(X, i, n_blocks, k, Pt, w, intercept, dA);
}

/* ---------------- SGD ----------------
 * Per batch b (steps t = it_b .. it_b+len-1, parameters theta_b at the batch start), for every
 * coordinate j touched by c_j >= 1 samples of the batch:
 *     acc_j      = sum_{i touches j} eta_{t_i} * dL_i * dA_ij(theta_b)       (sample order)
 *     theta_j   <- D_b^(1/c_j) * theta_j - acc_j / c_j
 *     viol      += | (acc_j + (sum_i eta_{t_i}) * reg * theta_j) / c_j |
 * untouched coordinates:  theta_j <- D_b * theta_j,   D_b = prod_t (1 - eta_t * reg).
 * i.e. the per-sample steps of the reference (sgd.nim:217-231) are AVERAGED per coordinate
 * instead of applied one after the other, and the coordinate receives len/c_j decay steps per
 * gradient step, which keeps the loss/regulariser balance of the sequential process for sparse
 * (c_j = 1: exactly the reference step after len-1 lazily applied decays) and dense (c_j = len:
 * plain mini-batch SGD on the batch mean) coordinates alike.  A plain sum (Hogwild without lost
 * updates) multiplies the step of a coordinate by c_j and diverges on dense ones (intercept).
 * The intercept is a coordinate touched by every sample (c = len).  batch == 1 gives the
 * reference's step exactly.
 * Touch cap (orc_mb_touch_cap = C >= 1, nfm_opt_set_touch_cap in the library): c_j above is replaced by
 * max(1, c_j / C) -- up to C of the batch's steps on a coordinate are SUMMED (what the reference's C Hogwild
 * threads do to their shared model), beyond that the sum is scaled by C / c_j; the decay exponent follows.
 * C = 1 is the mean written above. */
static int sgd_epoch_mb(const orc_csr* X, const double* y, mb_predict_fn predict, int n_blocks,
                        int k, int degree, int n_aug, double* Pt /*[n_blocks][da][k]*/, double* w,
                        double* intercept, const orc_sgd_cfg* c, const int64_t* perm, int64_t begin,
                        int64_t end, int64_t batch, int64_t* it, double* loss_sum,
                        double* viol_sum) {
  const int64_t d = X->d, da = d + n_aug;
  const size_t np = (size_t)n_blocks * da * k;
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  double* accP = (double*)calloc(np ? np : 1, sizeof(double));
  double* accw = (double*)calloc(d ? d : 1, sizeof(double));
  double* cnt = (double*)calloc(da ? da : 1, sizeof(double));     /* c_j */
  double* accP2 = (double*)calloc(np ? np : 1, sizeof(double));   /* experiment: sum of squared steps */
  double* setaP = (double*)calloc(da ? da : 1, sizeof(double));   /* sum of eta_P over touching samples */
  double* setaw = (double*)calloc(da ? da : 1, sizeof(double));
  double* A = (double*)calloc((size_t)k * (degree + 2), sizeof(double));
  if (!dA || !accP || !accw || !cnt || !setaP || !setaw || !A) return -1;
  double loss = 0.0, viol = 0.0;
  if (batch < 1) batch = 1;
  for (int64_t p0 = begin; p0 < end; p0 += batch) {
    const int64_t p1 = p0 + batch < end ? p0 + batch : end;
    const double len = (double)(p1 - p0);
    memset(accP, 0, sizeof(double) * np);
    if (orc_mb_sgd_agree != 0.0) memset(accP2, 0, sizeof(double) * np);
    memset(accw, 0, sizeof(double) * (size_t)d);
    memset(cnt, 0, sizeof(double) * (size_t)da);
    memset(setaP, 0, sizeof(double) * (size_t)da);
    memset(setaw, 0, sizeof(double) * (size_t)da);
    double accb = 0.0, seta0 = 0.0, DP = 1.0, Dw = 1.0, D0 = 1.0;
    for (int64_t pos = p0; pos < p1; pos++) {
      const int64_t i = perm ? perm[pos] : pos;
      const int64_t t = *it + (pos - p0);
      const double etaP = orc_get_eta(c->scheduling, c->eta0, c->power, c->beta, t);
      const double etaw = orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha, t);
      const double eta0 = orc_get_eta(c->scheduling, c->eta0, c->power, c->alpha0, t);
      row_view r = get_row(X, i);
      const double y_pred = predict(X, i, n_blocks, k, degree, n_aug, Pt, w, *intercept, A, dA);
      loss += orc_loss(c->loss, c->loss_param, y[i], y_pred);
      const double dL = orc_dloss(c->loss, c->loss_param, y[i], y_pred);
      for (int64_t q = 0; q < r.m + n_aug; q++) {
        const int64_t j = ROW_J(r, q, d);
        cnt[j] += 1.0;
        setaP[j] += etaP;
        setaw[j] += etaw;
        for (int o = 0; o < n_blocks; o++)
          for (int s = 0; s < k; s++) {
            const size_t e = ((size_t)o * da + j) * k + s;
            accP[e] += etaP * (dL * dA[e]);
            if (orc_mb_sgd_agree != 0.0) accP2[e] += (etaP * (dL * dA[e])) * (etaP * (dL * dA[e]));
          }
        if (c->fit_linear && q < r.m) accw[j] += etaw * (dL * r.val[q]);
      }
      accb += eta0 * dL;
      seta0 += eta0;
      DP *= (1 - etaP * c->beta);
      Dw *= (1 - etaw * c->alpha);
      D0 *= (1 - eta0 * c->alpha0);
    }
    for (int64_t j = 0; j < da; j++) {
      const double cj = cnt[j] > orc_mb_touch_cap ? cnt[j] / orc_mb_touch_cap : (cnt[j] > 0 ? 1.0 : 0.0);
      const double fP = cj > 0 ? (cj == 1.0 ? DP : pow(DP, 1.0 / cj)) : DP;
      for (int o = 0; o < n_blocks; o++)
        for (int s = 0; s < k; s++) {
          const size_t e = ((size_t)o * da + j) * k + s;
          if (cj > 0 && orc_mb_sgd_agree != 0.0) {
            double se = 1.0;
            if (cnt[j] > 1.0 && accP2[e] > 0.0) {
              double rho = (accP[e] * accP[e] - accP2[e]) / ((cnt[j] - 1.0) * accP2[e]);
              rho = rho < 0.0 ? 0.0 : (rho > 1.0 ? 1.0 : rho);
              se = 1.0 + (cnt[j] - 1.0) * pow(rho, orc_mb_sgd_agree);
            }
            viol += fabs((accP[e] + setaP[j] * c->beta * Pt[e]) / se);
            Pt[e] = DP * Pt[e] - accP[e] / se;
          } else if (cj > 0) {
            viol += fabs((accP[e] + setaP[j] * c->beta * Pt[e]) / cj);
            Pt[e] = fP * Pt[e] - accP[e] / cj;
          } else {
            Pt[e] = DP * Pt[e];
          }
        }
      if (c->fit_linear && j < d) {
        if (cj > 0) {
          const double fw = cj == 1.0 ? Dw : pow(Dw, 1.0 / cj);
          viol += fabs((accw[j] + setaw[j] * c->alpha * w[j]) / cj);
          w[j] = fw * w[j] - accw[j] / cj;
        } else {
          w[j] = Dw * w[j];
        }
      }
    }
    if (c->fit_intercept) {
      const double lb = len > orc_mb_touch_cap ? len / orc_mb_touch_cap : 1.0;
      const double f0 = lb == 1.0 ? D0 : pow(D0, 1.0 / lb);
      viol += fabs((accb + seta0 * c->alpha0 * *intercept) / lb);
      *intercept = f0 * *intercept - accb / lb;
    }
    *it += p1 - p0;
  }
  *loss_sum = loss;
  *viol_sum = viol;
  free(dA); free(accP); free(accP2); free(accw); free(cnt); free(setaP); free(setaw); free(A);
  return 0;
}

int orc_fm_sgd_epoch_mb(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                        int n_aug, double* P, double* w, double* intercept,
                        const orc_sgd_cfg* cfg, const int64_t* perm, int64_t begin, int64_t end,
                        int64_t batch, int64_t* it, double* loss_sum, double* viol_sum) {
  const int64_t da = X->d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  if (!Pt) return -1;
  to_train_layout(Pt, P, n_orders, k, da);
  int rc = sgd_epoch_mb(X, y, mb_predict_fm, n_orders, k, degree, n_aug, Pt, w, intercept, cfg,
                        perm, begin, end, batch, it, loss_sum, viol_sum);
  to_model_layout(P, Pt, n_orders, k, da);
  free(Pt);
  return rc;
}

int orc_ffm_sgd_epoch_mb(const orc_csr* X, const double* y, int k, double* P, double* w,
                         double* intercept, const orc_sgd_cfg* cfg, const int64_t* perm,
                         int64_t begin, int64_t end, int64_t batch, int64_t* it,
                         double* loss_sum, double* viol_sum) {
  return sgd_epoch_mb(X, y, mb_predict_ffm, (int)X->n_fields, k, 2, 0, P, w, intercept, cfg, perm,
                      begin, end, batch, it, loss_sum, viol_sum);
}

/* ---------------- AdaGrad ----------------
 * State is additive over samples (adagrad.nim:113-134), parameters are a pure
 * function of (state, it) (adagrad.nim:87-110).  Per batch: parameters of the
 * rows the batch touches are re-derived from the batch-start state with
 * it' = it_b - 1 (the reference's `update`, once per unique row), every sample
 * computes its gradient with them, the state receives the sum.  The reference
 * skips `update` for the very first sample ever (it == 1, adagrad.nim:171):
 * that sample is run as a batch of its own on the stored parameters. */
static int ada_epoch_mb(const orc_csr* X, const double* y, mb_predict_fn predict, int n_blocks,
                        int k, int degree, int n_aug, double* Pt, double* w, double* intercept,
                        const orc_adagrad_cfg* c, const int64_t* perm, int64_t begin, int64_t end,
                        int64_t batch, int64_t* it, double* gsum_P, double* gnorm_P, double* gsum_w,
                        double* gnorm_w, double* gsum_b, double* gnorm_b, double* loss_sum,
                        double* viol_sum) {
  const int64_t d = X->d, da = d + n_aug;
  const size_t np = (size_t)n_blocks * da * k;
  double* dA = (double*)calloc(np ? np : 1, sizeof(double));
  double* accG = (double*)calloc(np ? np : 1, sizeof(double));
  double* accN = (double*)calloc(np ? np : 1, sizeof(double));
  double* accGw = (double*)calloc(d ? d : 1, sizeof(double));
  double* accNw = (double*)calloc(d ? d : 1, sizeof(double));
  int64_t* seen = (int64_t*)malloc(sizeof(int64_t) * (size_t)(da ? da : 1));
  double* A = (double*)calloc((size_t)k * (degree + 2), sizeof(double));
  if (!dA || !accG || !accN || !accGw || !accNw || !seen || !A) return -1;
  for (int64_t j = 0; j < da; j++) seen[j] = -1;
  ada_state S;
  memset(&S, 0, sizeof(S));
  S.n_blocks = n_blocks; S.k = k; S.d = d; S.da = da; S.P = Pt; S.w = w; S.intercept = intercept;
  S.gsum_P = gsum_P; S.gnorm_P = gnorm_P; S.gsum_w = gsum_w; S.gnorm_w = gnorm_w;
  S.gsum_b = gsum_b; S.gnorm_b = gnorm_b; S.cfg = *c; S.it = *it;
  ada_init(&S);
  double loss = 0.0, viol = 0.0;
  if (batch < 1) batch = 1;
  int64_t p0 = begin, batch_id = 0;
  while (p0 < end) {
    const int first = (*it == 1);
    const int64_t p1 = first ? p0 + 1 : (p0 + batch < end ? p0 + batch : end);
    const double itp = (double)(*it - 1);
    memset(accG, 0, sizeof(double) * np);
    memset(accN, 0, sizeof(double) * np);
    memset(accGw, 0, sizeof(double) * (size_t)d);
    memset(accNw, 0, sizeof(double) * (size_t)d);
    double accGb = 0.0, accNb = 0.0;
    if (!first) {
      /* the reference's update(), once per unique touched row */
      const double tmp = c->eta0 * itp * c->beta;
      const double denw = itp * c->eta0 * c->alpha;
      for (int64_t pos = p0; pos < p1; pos++) {
        const int64_t i = perm ? perm[pos] : pos;
        row_view r = get_row(X, i);
        for (int64_t q = 0; q < r.m + n_aug; q++) {
          const int64_t j = ROW_J(r, q, d);
          if (seen[j] == batch_id) continue;
          seen[j] = batch_id;
          for (int o = 0; o < n_blocks; o++)
            for (int s = 0; s < k; s++) {
              const size_t e = ((size_t)o * da + j) * k + s;
              const double pjs = Pt[e];
              const double denom = tmp + sqrt(gnorm_P[e]);
              Pt[e] = -(c->eta0 * gsum_P[e]) / denom;
              viol += fabs(pjs - Pt[e]);
            }
          if (c->fit_linear && j < d) {
            const double wj = w[j];
            w[j] = -c->eta0 * gsum_w[j] / (denw + sqrt(gnorm_w[j]));
            viol += fabs(wj - w[j]);
          }
        }
      }
      if (c->fit_intercept) {
        const double old = *intercept;
        const double denom = sqrt(*gnorm_b) + c->eta0 * itp * c->alpha0;
        *intercept = -c->eta0 * *gsum_b / denom;
        viol += fabs(old - *intercept);
      }
    }
    for (int64_t pos = p0; pos < p1; pos++) {
      const int64_t i = perm ? perm[pos] : pos;
      row_view r = get_row(X, i);
      const double y_pred = predict(X, i, n_blocks, k, degree, n_aug, Pt, w, *intercept, A, dA);
      loss += orc_loss(c->loss, c->loss_param, y[i], y_pred);
      const double dL = orc_dloss(c->loss, c->loss_param, y[i], y_pred);
      for (int o = 0; o < n_blocks; o++)
        for (int64_t q = 0; q < r.m + n_aug; q++) {
          const int64_t j = ROW_J(r, q, d);
          for (int s = 0; s < k; s++) {
            const size_t e = ((size_t)o * da + j) * k + s;
            const double grad = dL * dA[e];
            accG[e] += grad;
            accN[e] += grad * grad;
          }
        }
      if (c->fit_intercept) { accGb += dL; accNb += dL * dL; }
      if (c->fit_linear)
        for (int64_t q = 0; q < r.m; q++) {
          const int64_t j = r.idx[q];
          const double g = dL * r.val[q];
          accGw[j] += g;
          accNw[j] += g * g;
        }
    }
    for (size_t e = 0; e < np; e++) { gsum_P[e] += accG[e]; gnorm_P[e] += mb_norm_inc(accG[e], accN[e]); }
    if (c->fit_linear)
      for (int64_t j = 0; j < d; j++) { gsum_w[j] += accGw[j]; gnorm_w[j] += mb_norm_inc(accGw[j], accNw[j]); }
    if (c->fit_intercept) { *gsum_b += accGb; *gnorm_b += mb_norm_inc(accGb, accNb); }
    *it += p1 - p0;
    p0 = p1;
    batch_id++;
  }
  *loss_sum = loss;
  *viol_sum = viol;
  free(dA); free(accG); free(accN); free(accGw); free(accNw); free(seen); free(A);
  return 0;
}

/* P in/out: model layout [O][k][d+a]; it is NOT finalised here (callers use
 * orc_*_adagrad_finalize_mb, the reference's finalize, adagrad.nim:65-84). */
int orc_fm_adagrad_epoch_mb(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                            int n_aug, double* P, double* w, double* intercept,
                            const orc_adagrad_cfg* cfg, const int64_t* perm, int64_t begin,
                            int64_t end, int64_t batch, int64_t* it, double* gsum_P,
                            double* gnorm_P, double* gsum_w, double* gnorm_w, double* gsum_b,
                            double* gnorm_b, double* loss_sum, double* viol_sum) {
  const int64_t da = X->d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  if (!Pt) return -1;
  to_train_layout(Pt, P, n_orders, k, da);
  int rc = ada_epoch_mb(X, y, mb_predict_fm, n_orders, k, degree, n_aug, Pt, w, intercept, cfg,
                        perm, begin, end, batch, it, gsum_P, gnorm_P, gsum_w, gnorm_w, gsum_b,
                        gnorm_b, loss_sum, viol_sum);
  to_model_layout(P, Pt, n_orders, k, da);
  free(Pt);
  return rc;
}

int orc_ffm_adagrad_epoch_mb(const orc_csr* X, const double* y, int k, double* P, double* w,
                             double* intercept, const orc_adagrad_cfg* cfg, const int64_t* perm,
                             int64_t begin, int64_t end, int64_t batch, int64_t* it,
                             double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                             double* gsum_b, double* gnorm_b, double* loss_sum, double* viol_sum) {
  return ada_epoch_mb(X, y, mb_predict_ffm, (int)X->n_fields, k, 2, 0, P, w, intercept, cfg, perm,
                      begin, end, batch, it, gsum_P, gnorm_P, gsum_w, gnorm_w, gsum_b, gnorm_b,
                      loss_sum, viol_sum);
}

/* the reference's finalize (adagrad.nim:65-84) on caller-held state; P in the
 * TRAINING layout [n_blocks][da][k] (for FM convert with the transpose). */
int orc_adagrad_finalize(int n_blocks, int k, int64_t d, int64_t da, double* Pt, double* w,
                         double* intercept, const orc_adagrad_cfg* cfg, int64_t it,
                         double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                         double* gsum_b, double* gnorm_b) {
  ada_state S;
  memset(&S, 0, sizeof(S));
  S.n_blocks = n_blocks; S.k = k; S.d = d; S.da = da; S.P = Pt; S.w = w; S.intercept = intercept;
  S.gsum_P = gsum_P; S.gnorm_P = gnorm_P; S.gsum_w = gsum_w; S.gnorm_w = gnorm_w;
  S.gsum_b = gsum_b; S.gnorm_b = gnorm_b; S.cfg = *cfg; S.it = it;
  ada_finalize(&S);
  return 0;
}

/* FM convenience: P in the model layout [O][k][d+a]. */
int orc_fm_adagrad_finalize(int degree, int k, int n_orders, int n_aug, int64_t d, double* P,
                            double* w, double* intercept, const orc_adagrad_cfg* cfg, int64_t it,
                            double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                            double* gsum_b, double* gnorm_b) {
  (void)degree;
  const int64_t da = d + n_aug;
  const size_t np = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(np ? np : 1, sizeof(double));
  if (!Pt) return -1;
  to_train_layout(Pt, P, n_orders, k, da);
  orc_adagrad_finalize(n_orders, k, d, da, Pt, w, intercept, cfg, it, gsum_P, gnorm_P, gsum_w,
                       gnorm_w, gsum_b, gnorm_b);
  to_model_layout(P, Pt, n_orders, k, da);
  free(Pt);
  return 0;
}

/* mini-batch proximal SGD shares predict_with_grad with the above */
#include "nimfm_psgd.c"
