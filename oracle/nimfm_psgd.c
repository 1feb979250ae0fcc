/*
 * oracle/nimfm_psgd.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's mini-batch proximal SGD:
 *   optimizer/minibatch_psgd.nim:67-122 (updateGradient, epoch), model/params.nim:33-98
 *   (add / scale / step) and the matrix proximal operators the solver calls,
 *   regularizer/l1.nim:35-39, l21.nim:23-34, squaredl12.nim:16-69,147-162,
 *   squaredl21.nim:46-63, regularizer/utils.nim:4-5 (softthreshold).
 * Textually included by nimfm_mb.c (shares predict_with_grad with the SGD restatement).
 *
 * PARITY PIN STATUS: "parity unpinned" against reference-run outputs (no Nim toolchain, see
 * nimfm_oracle.h).  The reference's tests hold one fixture-free check for this row,
 * tests/test_squaredl12.nim:10-27 (proxSquaredL12 == the sort-based proxSquaredL12Slow of
 * tests/regularizer/squaredl12_slow.nim:10-25 on 1000 random vectors x 13 lambdas); both sides are
 * restated here and tests/test_oracle_psgd.py re-runs that grid.  The solver itself has no reference
 * test; the restatement is cross-checked against a dense numpy statement of the update.
 *
 * Two deliberate notes:
 *  - proxSquaredL12 picks its pivots with Nim's global rand() (squaredl12.nim:34).  The operator's
 *    result does not depend on the pivots except through the summation order of S; this file draws
 *    them from a caller-seeded xorshift generator instead (Nim's stdlib RNG is not in the reference
 *    tree).
 *  - Params.add gates the intercept's gradient step on grad.fitLinear (model/params.nim:47): with
 *    fitIntercept and not fitLinear the intercept only shrinks.  Restated as written.
 */

/* regularizer/utils.nim:4-5 */
static double softthreshold(double x, double alpha) {
  return (double)sgn(x) * fmax(fabs(x) - alpha, 0.0);
}

static uint64_t psgd_next(uint64_t* s) {
  uint64_t x = *s ? *s : 0x9E3779B97F4A7C15ull;
  x ^= x << 13; x ^= x >> 7; x ^= x << 17;
  *s = x;
  return x;
}

/* regularizer/squaredl12.nim:16-69 */
void orc_prox_squaredl12(double* p, int64_t n, double lam, uint64_t* rng) {
  double S = 0.0;
  int64_t theta = 0, offset = 0, n_cand = n;
  int64_t* cand = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) cand[i] = i;
  while (n_cand != 0) {
    const int64_t ii = (int64_t)(psgd_next(rng) % (uint64_t)n_cand);
    const int64_t i = cand[offset + ii];
    const double pivot = fabs(p[i]);
    { int64_t t = cand[offset + ii]; cand[offset + ii] = cand[offset + n_cand - 1]; cand[offset + n_cand - 1] = t; }
    int64_t nG = 1, nL = 0;
    double SGi = pivot;
    for (int64_t ii2 = 0; ii2 < n_cand - 1; ii2++) {
      const int64_t i2 = cand[offset + ii2];
      if (pivot > fabs(p[i2])) {
        int64_t t = cand[offset + nL]; cand[offset + nL] = cand[offset + ii2]; cand[offset + ii2] = t;
        nL++;
      } else {
        nG++;
        SGi += fabs(p[i2]);
      }
    }
    if (pivot > 2 * lam * (S + SGi) / (1.0 + 2.0 * lam * (double)(theta + nG))) { /* L */
      n_cand = nL;
      S += SGi;
      theta += nG;
    } else { /* G */
      offset = offset + nL;
      n_cand = 0;
      for (int64_t ii2 = 0; ii2 < nG - 1; ii2++) {
        const int64_t i2 = cand[offset + ii2];
        if (pivot < fabs(p[i2])) {
          int64_t t = cand[offset + ii2]; cand[offset + ii2] = cand[offset + n_cand]; cand[offset + n_cand] = t;
          n_cand++;
        }
      }
    }
  }
  S /= 1.0 + 2.0 * lam * (double)theta;
  for (int64_t i = 0; i < n; i++) p[i] = softthreshold(p[i], 2 * lam * S);
  free(cand);
}

static int cmp_desc(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return (x < y) - (x > y);
}

/* tests/regularizer/squaredl12_slow.nim:10-25 */
void orc_prox_squaredl12_slow(double* p, int64_t n, double lam) {
  double* absp = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  double* S = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) absp[i] = fabs(p[i]);
  qsort(absp, (size_t)n, sizeof(double), cmp_desc);
  double c = 0.0;
  for (int64_t i = 0; i < n; i++) {
    c += absp[i];
    S[i] = 2.0 * lam * c;
  }
  for (int64_t i = 0; i < n; i++) S[i] /= (1.0 + 2.0 * lam * ((double)i + 1.0));
  int64_t theta = 0;
  for (int64_t i = 0; i < n; i++) {
    if (absp[i] - S[i] < 0) break;
    theta++;
  }
  for (int64_t i = 0; i < n; i++) {
    if (theta == 0 || fabs(p[i]) < absp[theta - 1]) p[i] = 0;
    else p[i] = softthreshold(p[i], S[theta - 1]);
  }
  free(absp);
  free(S);
}

/* tensor/tensor.nim:608-616 norm(x, 2) */
static double norm2(const double* x, int64_t n, int64_t stride) {
  double r = 0.0;
  for (int64_t i = 0; i < n; i++) r += fabs(x[i * stride]) * fabs(x[i * stride]);
  return pow(r, 1.0 / 2.0);
}

/* the matrix prox the solver calls: Po is one order in the training layout [da][k] */
void orc_prox(int reg, int transpose, double* Po, int64_t da, int k, double lam, uint64_t* rng) {
  if (reg == ORC_REG_L1) { /* l1.nim:35-39 */
    for (int64_t j = 0; j < da; j++)
      for (int s = 0; s < k; s++) Po[j * k + s] = softthreshold(Po[j * k + s], lam);
  } else if (reg == ORC_REG_L21) { /* l21.nim:23-34 */
    for (int64_t j = 0; j < da; j++) {
      const double nrm = norm2(Po + j * k, k, 1);
      if (nrm > lam) {
        const double f = 1.0 - lam / nrm;
        for (int s = 0; s < k; s++) Po[j * k + s] *= f;
      } else {
        for (int s = 0; s < k; s++) Po[j * k + s] = 0.0;
      }
    }
  } else if (reg == ORC_REG_SQUAREDL12) { /* squaredl12.nim:147-162 */
    if (transpose) {
      double* col = (double*)malloc(sizeof(double) * (size_t)(da > 0 ? da : 1));
      for (int s = 0; s < k; s++) {
        for (int64_t j = 0; j < da; j++) col[j] = Po[j * k + s];
        orc_prox_squaredl12(col, da, lam, rng);
        for (int64_t j = 0; j < da; j++) Po[j * k + s] = col[j];
      }
      free(col);
    } else {
      for (int64_t j = 0; j < da; j++) orc_prox_squaredl12(Po + j * k, k, lam, rng);
    }
  } else { /* squaredl21.nim:46-54 (transpose = false, the default; the transposed branch indexes
              a length-k vector by feature, :56-63, and is not restated) */
    double* norms = (double*)malloc(sizeof(double) * (size_t)(da > 0 ? da : 1));
    for (int64_t j = 0; j < da; j++) norms[j] = norm2(Po + j * k, k, 1);
    for (int64_t j = 0; j < da; j++)
      if (norms[j] != 0)
        for (int s = 0; s < k; s++) Po[j * k + s] /= norms[j];
    orc_prox_squaredl12(norms, da, lam, rng);
    for (int64_t j = 0; j < da; j++)
      for (int s = 0; s < k; s++) Po[j * k + s] *= norms[j];
    free(norms);
  }
}

/* optimizer/minibatch_psgd.nim:87-122 `epoch` (one outer iteration): stream holds the sample indices in
 * the order the inner loops consume them (indices[ii], wrap-arounds and reshuffles included), n_stream =
 * miniBatchSize * maxIterInner.  P in the model layout [O][k][d+a].  *it advances by maxIterInner.
 * loss_sum = sum of loss(y_i, yhat_i) (the caller divides, :122). */
int orc_fm_mbpsgd_epoch(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                        double* P, double* w, double* intercept, const orc_psgd_cfg* cfg,
                        const int64_t* stream, int64_t n_stream, int64_t batch, int64_t* it,
                        uint64_t* rng, double* loss_sum) {
  const int64_t d = X->d, da = d + n_aug;
  const size_t nP = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(nP ? nP : 1, sizeof(double));
  double* gP = (double*)calloc(nP ? nP : 1, sizeof(double));
  double* dA = (double*)calloc(nP ? nP : 1, sizeof(double));
  double* gw = (double*)calloc((size_t)(d ? d : 1), sizeof(double));
  double* A = (double*)calloc((size_t)k * (degree + 1), sizeof(double));
  to_train_layout(Pt, P, n_orders, k, da);
  double b = *intercept, result = 0.0;
  const int64_t inner = batch > 0 ? n_stream / batch : 0;
  int64_t ii = 0;
  for (int64_t t = 0; t < inner; t++) {
    memset(gP, 0, sizeof(double) * nP); /* grads <- 0.0, :95 */
    memset(gw, 0, sizeof(double) * (size_t)d);
    double gb = 0.0;
    for (int64_t bb = 0; bb < batch; bb++) { /* updateGradient, :67-84 */
      const int64_t i = stream[ii++];
      row_view r = get_row(X, i);
      const double yp = predict_with_grad(r, d, n_aug, k, n_orders, degree, Pt, w, b, A, dA);
      result += orc_loss(cfg->loss, cfg->loss_param, y[i], yp);
      const double coef = 1.0 * orc_dloss(cfg->loss, cfg->loss_param, y[i], yp) / (double)batch;
      for (int o = 0; o < n_orders; o++)
        for (int64_t q = 0; q < r.m + n_aug; q++) {
          const int64_t j = ROW_J(r, q, d);
          for (int s = 0; s < k; s++) gP[((size_t)o * da + j) * k + s] += coef * dA[((size_t)o * da + j) * k + s];
        }
      if (cfg->fit_linear)
        for (int64_t q = 0; q < r.m; q++) gw[r.idx[q]] += coef * r.val[q];
      if (cfg->fit_intercept) gb += coef;
    }
    const double eta_P = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->beta, *it);
    const double eta_w = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha, *it);
    const double eta_0 = orc_get_eta(cfg->scheduling, cfg->eta0, cfg->power, cfg->alpha0, *it);
    /* params.step, model/params.nim:90-98 = add (:33-48) then scale (:60-65) */
    const double scale_P = 1.0 + eta_P * cfg->beta, scale_w = 1.0 + eta_w * cfg->alpha,
                 scale_0 = 1.0 + eta_0 * cfg->alpha0;
    for (size_t e = 0; e < nP; e++) Pt[e] += -eta_P * gP[e];
    if (cfg->fit_linear)
      for (int64_t j = 0; j < d; j++) w[j] += -eta_w * gw[j];
    if (cfg->fit_intercept && cfg->fit_linear) b += -eta_0 * gb; /* params.nim:47, as written */
    {
      const double sP = 1.0 / scale_P, sw = 1.0 / scale_w, s0 = 1.0 / scale_0;
      for (size_t e = 0; e < nP; e++) Pt[e] *= sP;
      if (cfg->fit_linear)
        for (int64_t j = 0; j < d; j++) w[j] *= sw;
      if (cfg->fit_intercept) b *= s0;
    }
    for (int o = 0; o < n_orders; o++) /* :118-120 */
      orc_prox(cfg->reg, cfg->reg_transpose, Pt + (size_t)o * da * k, da, k,
               cfg->gamma * eta_P / (1.0 + eta_P * cfg->beta), rng);
    (*it)++;
  }
  to_model_layout(P, Pt, n_orders, k, da);
  *intercept = b;
  if (loss_sum) *loss_sum = result;
  free(Pt); free(gP); free(dA); free(gw); free(A);
  return 0;
}

/* regularizer eval for the verbose line (minibatch_psgd.nim:196-199): l1.nim:19-22, l21.nim:17-20,
 * squaredl12.nim:72-75, squaredl21.nim:21-23; Po in the training layout [da][k] */
double orc_reg_eval(int reg, int transpose, const double* Po, int64_t da, int k) {
  double r = 0.0;
  if (reg == ORC_REG_L1) {
    for (int64_t e = 0; e < da * k; e++) r += fabs(Po[e]);
  } else if (reg == ORC_REG_L21) {
    for (int64_t j = 0; j < da; j++) r += norm2(Po + j * k, k, 1);
  } else if (reg == ORC_REG_SQUAREDL12) {
    if (transpose) { /* norm(norm(P, 1, axis=0), 2)^2 */
      for (int s = 0; s < k; s++) {
        double c = 0.0;
        for (int64_t j = 0; j < da; j++) c += fabs(Po[j * k + s]);
        r += c * c;
      }
    } else {
      for (int64_t j = 0; j < da; j++) {
        double c = 0.0;
        for (int s = 0; s < k; s++) c += fabs(Po[j * k + s]);
        r += c * c;
      }
    }
  } else { /* norm(norm(P, 2, axis=1), 1)^2 */
    double c = 0.0;
    for (int64_t j = 0; j < da; j++) c += norm2(Po + j * k, k, 1);
    r = c * c;
  }
  return r;
}

/* optimizer/pgd.nim:70-103 predictAllWithGrad: yPred, dL and the gradient of the MEAN loss at the given
 * parameters (the other function SURVEY 8(f) rank 3 names).  P in the model layout [O][k][d+a]; gP is returned in
 * the TRAINING layout [O][d+a][k] like the reference's grads.P. */
int orc_fm_predict_all_with_grad(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                                 const double* P, const double* w, double intercept, int loss, double loss_param,
                                 int fit_linear, int fit_intercept, double* y_pred, double* dL, double* gP,
                                 double* gw, double* gb) {
  const int64_t d = X->d, da = d + n_aug, n = X->n;
  const size_t nP = (size_t)n_orders * da * k;
  double* Pt = (double*)calloc(nP ? nP : 1, sizeof(double));
  double* dA = (double*)calloc(nP ? nP : 1, sizeof(double));
  double* A = (double*)calloc((size_t)k * (degree + 1), sizeof(double));
  to_train_layout(Pt, P, n_orders, k, da);
  memset(gP, 0, sizeof(double) * nP); /* grads.P <- 0.0, :76 */
  for (int64_t j = 0; j < d; j++) gw[j] = 0.0;
  *gb = 0.0;
  for (int64_t i = 0; i < n; i++) {
    row_view r = get_row(X, i);
    /* :79-88: mvmul(X, w) + intercept, then per order computeAnova + derivative -- predict_with_grad's sum in
     * the order linear, intercept, orders; the reference adds intercept after the linear term (:80-81) */
    double yp = 0.0;
    for (int64_t q = 0; q < r.m; q++) yp += w[r.idx[q]] * r.val[q];
    yp += intercept;
    for (int o = 0; o < n_orders; o++) {
      const double* Po = Pt + (size_t)o * da * k;
      yp += compute_anova(Po, r, d, n_aug, k, degree - o, A, degree + 1);
      compute_anova_derivative(Po, r, d, n_aug, k, degree - o, A, degree + 1, dA + (size_t)o * da * k);
    }
    y_pred[i] = yp;
    dL[i] = orc_dloss(loss, loss_param, y[i], yp); /* :90 */
    for (int o = 0; o < n_orders; o++)
      for (int64_t q = 0; q < r.m + n_aug; q++) {
        const int64_t j = ROW_J(r, q, d);
        for (int s = 0; s < k; s++) gP[((size_t)o * da + j) * k + s] += dL[i] * dA[((size_t)o * da + j) * k + s];
      }
  }
  if (fit_linear) /* vmmul(dL, X, grads.w), :96-97 */
    for (int64_t i = 0; i < n; i++) {
      row_view r = get_row(X, i);
      for (int64_t q = 0; q < r.m; q++) gw[r.idx[q]] += dL[i] * r.val[q];
    }
  if (fit_intercept) { /* :98-99 */
    double sm = 0.0;
    for (int64_t i = 0; i < n; i++) sm += dL[i];
    *gb = sm;
  }
  { /* grads /= float(nSamples), :102 = scale(1.0 / n): P always, w / intercept when fitted (params.nim:60-65) */
    const double sc = 1.0 / (double)n;
    for (size_t e = 0; e < nP; e++) gP[e] *= sc;
    if (fit_linear)
      for (int64_t j = 0; j < d; j++) gw[j] *= sc;
    if (fit_intercept) *gb *= sc;
  }
  free(Pt); free(dA); free(A);
  return 0;
}
