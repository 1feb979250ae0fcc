/*
 * oracle/nimfm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64, single thread) of the nimfm FM hot path:
 *   decisionFunction, SGD.fit, AdaGrad.fit for FactorizationMachine and
 *   FieldAwareFactorizationMachine, plus the brute-force "slow" models the
 *   reference's own unit tests compare against, plus a CPU restatement of
 *   this repository's deterministic mini-batch rule, plus (nimfm_psgd.c)
 *   mini-batch proximal SGD, the matrix proximal operators and pgd.predictAllWithGrad.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (nimfm_amd/, libnimfm_hip.so) never
 * links, imports or calls it.
 *
 * PARITY PIN STATUS: "parity unpinned" against reference-run outputs.
 *   - The reference is 100 % Nim; no Nim toolchain exists in the build image,
 *     so the reference cannot be compiled or run.
 *   - The reference's tests hold NO literal golden vectors for this path; they
 *     pin it differentially (fast == brute-force slow, rtol 1e-6 atol 1e-9,
 *     tests/utils.nim:82-105).  This oracle restates BOTH sides and
 *     tests/test_oracle_*.py re-runs the reference's own test grids on them
 *     (test_kernels.nim:26-46, test_sgd.nim:16-126, test_adagrad.nim:58-126,
 *     test_sgd_ffm.nim, test_adagrad_ffm.nim).
 *
 * All file:line citations are relative to /root/reference/.
 * Layouts follow the reference: FM  P[nOrders][k][d+nAug]  (model layout,
 * model/factorization_machine.nim:31-34), FFM P[nFields][d][k]
 * (model/field_aware_factorization_machine.nim:16-17), AdaGrad state in the
 * training layout [nOrders][d+nAug][k] (optimizer/adagrad.nim:53-54,155).
 * Nim `int` is int64, `float64` is double.
 */
#ifndef NIMFM_ORACLE_H
#define NIMFM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tensor/sparse.nim:9-12,19-24 */
typedef struct orc_csr {
  int64_t n, d;            /* shape[0], shape[1] (without augments) */
  const int64_t* indptr;   /* n+1 */
  const int64_t* indices;  /* nnz */
  const double* data;      /* nnz */
  const int64_t* fields;   /* nnz or NULL (CSRFieldMatrix) */
  int64_t n_fields;
} orc_csr;

enum { ORC_LOSS_SQUARED = 0, ORC_LOSS_SQUARED_HINGE = 1, ORC_LOSS_LOGISTIC = 2, ORC_LOSS_HUBER = 3 };
/* optimizer/sgd.nim:7-11 */
enum { ORC_SCHED_CONSTANT = 0, ORC_SCHED_OPTIMAL = 1, ORC_SCHED_INVSCALING = 2, ORC_SCHED_PEGASOS = 3 };
/* model/factorization_machine.nim:6-9 */
enum { ORC_LOWER_EXPLICIT = 0, ORC_LOWER_AUGMENT = 1, ORC_LOWER_NONE = 2 };

typedef struct orc_sgd_cfg {
  double eta0, alpha0, alpha, beta, power, loss_param;
  int32_t loss, scheduling, fit_linear, fit_intercept;
} orc_sgd_cfg;

typedef struct orc_adagrad_cfg {
  double eta0, alpha0, alpha, beta, eps, loss_param;
  int32_t loss, fit_linear, fit_intercept, pad_;
} orc_adagrad_cfg;

/* loss.nim:15-102 */
double orc_loss(int loss, double param, double y, double p);
double orc_dloss(int loss, double param, double y, double p);
/* optimizer/sgd.nim:60-69 */
double orc_get_eta(int scheduling, double eta0, double power, double reg, int64_t it);
/* model/factorization_machine.nim:81-97 */
int orc_n_augments(int degree, int fit_lower, int fit_linear);
int orc_n_orders(int degree, int fit_lower);
/* utils.nim:33, metrics.nim:5-13,39-47, optimizer/utils.nim:56-59 */
double orc_expit(double x);
double orc_rmse(const double* y_true, const double* y_score, int64_t n);
double orc_accuracy_sign(const double* y_true, const double* y_score, int64_t n);
double orc_regularization(const double* P, int64_t nP, const double* w, int64_t nw,
                          double intercept, double alpha0, double alpha, double beta);

/* ---- FM, fast path (faithful restatement) ---- */
/* model/factorization_machine.nim:100-122 + kernels.nim:14-19,46-64 */
int orc_fm_decision_function(const orc_csr* X, int degree, int k, int n_orders, int n_aug,
                             const double* P, const double* lams, const double* w,
                             double intercept, double* out);
/* optimizer/sgd.nim:261-328 (fit) built from :92-258.  perms: [max_iter][n] or
 * NULL (= shuffle off).  *it is the optimizer's `it` (in/out).  Returns 0. */
int orc_fm_sgd_fit(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                   double* P, double* w, double* intercept, const orc_sgd_cfg* cfg,
                   int max_iter, double tol, const int64_t* perms, int64_t* it,
                   double* epoch_loss, double* epoch_viol, int* n_epochs_run);
/* optimizer/adagrad.nim:137-203 built from :47-134 + fit_linear.nim:50-57.
 * State arrays are caller-owned; (re)initialised here when *it == 1
 * (adagrad.nim:52-55). */
int orc_fm_adagrad_fit(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                       double* P, double* w, double* intercept, const orc_adagrad_cfg* cfg,
                       int max_iter, double tol, const int64_t* perms, int64_t* it,
                       double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                       double* gsum_b, double* gnorm_b,
                       double* epoch_loss, double* epoch_viol, int* n_epochs_run);
/* optimizer/sgd_multi.nim:13-120: Hogwild, contiguous slices, shared
 * unsynchronised state (intentionally racy, like the reference).  Used only as
 * the multi-thread CPU baseline. */
int orc_fm_sgd_fit_hogwild(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                           int n_aug, double* P, double* w, double* intercept,
                           const orc_sgd_cfg* cfg, int max_iter, double tol, const int64_t* perms,
                           int64_t* it, int n_threads, double* epoch_loss, double* epoch_viol,
                           int* n_epochs_run);

/* ---- FFM, fast path ---- */
/* model/field_aware_factorization_machine.nim:52-76 */
int orc_ffm_decision_function(const orc_csr* X, int k, const double* P, const double* w,
                              double intercept, double* out);
/* optimizer/sgd_ffm.nim:11-106 */
int orc_ffm_sgd_fit(const orc_csr* X, const double* y, int k, double* P, double* w,
                    double* intercept, const orc_sgd_cfg* cfg, int max_iter, double tol,
                    const int64_t* perms, int64_t* it, double* epoch_loss, double* epoch_viol,
                    int* n_epochs_run);
/* optimizer/adagrad_ffm.nim:11-66 */
int orc_ffm_adagrad_fit(const orc_csr* X, const double* y, int k, double* P, double* w,
                        double* intercept, const orc_adagrad_cfg* cfg, int max_iter, double tol,
                        const int64_t* perms, int64_t* it, double* gsum_P, double* gnorm_P,
                        double* gsum_w, double* gnorm_w, double* gsum_b, double* gnorm_b,
                        double* epoch_loss, double* epoch_viol, int* n_epochs_run);

/* ---- brute-force models from the reference's tests (dense X[n][d]) ---- */
/* tests/kernels_slow.nim:18-27 */
double slow_anova(const double* Xrow, const double* Prow, int d, int m, int degree);
/* tests/model/fm_slow.nim:42-73 */
int slow_fm_decision_function(const double* Xd, int64_t n, int d, int degree, int k, int n_orders,
                              int n_aug, const double* P, const double* w, double intercept,
                              double* out);
/* tests/optimizer/sgd_slow.nim:38-91 */
int slow_fm_sgd_fit(const double* Xd, int64_t n, int d, const double* y, int degree, int k,
                    int n_orders, int n_aug, double* P, double* w, double* intercept,
                    const orc_sgd_cfg* cfg, int max_iter, const int64_t* perms, int64_t* it);
/* tests/optimizer/adagrad_slow.nim:29-102 */
int slow_fm_adagrad_fit(const double* Xd, int64_t n, int d, const double* y, int degree, int k,
                        int n_orders, int n_aug, double* P, double* w, double* intercept,
                        const orc_adagrad_cfg* cfg, int max_iter, const int64_t* perms,
                        int64_t* it);
/* tests/model/ffm_slow.nim:38-56,110-127; tests/optimizer/sgd_ffm_slow.nim:8-57;
 * tests/optimizer/adagrad_ffm_slow.nim:8-37.  field_of[d] maps feature->field. */
int slow_ffm_decision_function(const double* Xd, int64_t n, int d, const int64_t* field_of,
                               int n_fields, int k, const double* P, const double* w,
                               double intercept, double* out);
int slow_ffm_sgd_fit(const double* Xd, int64_t n, int d, const int64_t* field_of, int n_fields,
                     const double* y, int k, double* P, double* w, double* intercept,
                     const orc_sgd_cfg* cfg, int max_iter, const int64_t* perms, int64_t* it);
int slow_ffm_adagrad_fit(const double* Xd, int64_t n, int d, const int64_t* field_of, int n_fields,
                         const double* y, int k, double* P, double* w, double* intercept,
                         const orc_adagrad_cfg* cfg, int max_iter, const int64_t* perms,
                         int64_t* it);

/* ---- this repository's deterministic mini-batch rule ----
 * NOT a reference function: a CPU restatement of the rule the HIP throughput
 * kernels implement, so that they can be checked to ~1e-12.  With batch == 1
 * it is mathematically the reference's sequential step.  Works on samples
 * perm[begin..end) (perm NULL = identity), advances *it by end-begin. */
int orc_fm_sgd_epoch_mb(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                        int n_aug, double* P, double* w, double* intercept,
                        const orc_sgd_cfg* cfg, const int64_t* perm, int64_t begin, int64_t end,
                        int64_t batch, int64_t* it, double* loss_sum, double* viol_sum);
int orc_fm_adagrad_epoch_mb(const orc_csr* X, const double* y, int degree, int k, int n_orders,
                            int n_aug, double* P, double* w, double* intercept,
                            const orc_adagrad_cfg* cfg, const int64_t* perm, int64_t begin,
                            int64_t end, int64_t batch, int64_t* it, double* gsum_P,
                            double* gnorm_P, double* gsum_w, double* gnorm_w, double* gsum_b,
                            double* gnorm_b, double* loss_sum, double* viol_sum);
int orc_ffm_sgd_epoch_mb(const orc_csr* X, const double* y, int k, double* P, double* w,
                         double* intercept, const orc_sgd_cfg* cfg, const int64_t* perm,
                         int64_t begin, int64_t end, int64_t batch, int64_t* it,
                         double* loss_sum, double* viol_sum);
int orc_ffm_adagrad_epoch_mb(const orc_csr* X, const double* y, int k, double* P, double* w,
                             double* intercept, const orc_adagrad_cfg* cfg, const int64_t* perm,
                             int64_t begin, int64_t end, int64_t batch, int64_t* it,
                             double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                             double* gsum_b, double* gnorm_b, double* loss_sum, double* viol_sum);

/* the reference's AdaGrad finalize (adagrad.nim:65-84) on caller-held state.
 * orc_adagrad_finalize: P in the training layout [n_blocks][da][k] (FFM's own
 * layout); orc_fm_adagrad_finalize: P in the FM model layout [O][k][d+a]. */
int orc_adagrad_finalize(int n_blocks, int k, int64_t d, int64_t da, double* Pt, double* w,
                         double* intercept, const orc_adagrad_cfg* cfg, int64_t it,
                         double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                         double* gsum_b, double* gnorm_b);
int orc_fm_adagrad_finalize(int degree, int k, int n_orders, int n_aug, int64_t d, double* P,
                            double* w, double* intercept, const orc_adagrad_cfg* cfg, int64_t it,
                            double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w,
                            double* gsum_b, double* gnorm_b);

/* ---- mini-batch proximal SGD, SURVEY.md 8(f) rank 3 (nimfm_psgd.c) ---- */
enum { ORC_REG_L1 = 0, ORC_REG_L21 = 1, ORC_REG_SQUAREDL12 = 2, ORC_REG_SQUAREDL21 = 3 };
typedef struct orc_psgd_cfg {
  double eta0, alpha0, alpha, beta, gamma, power, loss_param;
  int32_t loss, scheduling, fit_linear, fit_intercept, reg, reg_transpose;
} orc_psgd_cfg;
/* regularizer/squaredl12.nim:16-69 and its brute-force twin tests/regularizer/squaredl12_slow.nim:10-25 */
void orc_prox_squaredl12(double* p, int64_t n, double lam, uint64_t* rng);
void orc_prox_squaredl12_slow(double* p, int64_t n, double lam);
/* the matrix prox of l1.nim:35-39, l21.nim:23-34, squaredl12.nim:147-162, squaredl21.nim:46-54 on one
 * order in the training layout [da][k] */
void orc_prox(int reg, int transpose, double* Po, int64_t da, int k, double lam, uint64_t* rng);
double orc_reg_eval(int reg, int transpose, const double* Po, int64_t da, int k);
/* optimizer/minibatch_psgd.nim:87-122 (one outer iteration over an explicit index stream) */
int orc_fm_mbpsgd_epoch(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                        double* P, double* w, double* intercept, const orc_psgd_cfg* cfg,
                        const int64_t* stream, int64_t n_stream, int64_t batch, int64_t* it,
                        uint64_t* rng, double* loss_sum);
/* optimizer/pgd.nim:70-103 predictAllWithGrad (gP in the training layout [O][d+a][k]) */
int orc_fm_predict_all_with_grad(const orc_csr* X, const double* y, int degree, int k, int n_orders, int n_aug,
                                 const double* P, const double* w, double intercept, int loss, double loss_param,
                                 int fit_linear, int fit_intercept, double* y_pred, double* dL, double* gP,
                                 double* gw, double* gb);

#ifdef __cplusplus
}
#endif
#endif
