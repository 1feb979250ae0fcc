/*
 * include/nimfm_hip.h -- C ABI of libnimfm_hip.so, the MI355X (gfx950) hot path
 * for nimfm factorization machines.
 *
 * The reference (neonnnnn/nimfm, /root/reference) is pure Nim with NO FFI or
 * plugin boundary: `optim.fit(X, y, fm)` and `fm.decisionFunction(X)` are Nim
 * procs resolved at compile time.  This header is therefore the
 * FFI a Nim shim binds with {.importc, cdecl, dynlib.} (nim/nimfm_hip.nim,
 * INTEGRATION.md); each entry point cites the reference proc(s) it replaces.
 * All citations are relative to /root/reference/src/nimfm/.
 *
 * Conventions
 *   - every function returns int32 status: 0 = NFM_OK, < 0 = error; nothing
 *     throws; nfm_last_error() returns a thread-local message.
 *   - plain pointers and sizes only.  Host pointers are caller-owned; the
 *     library copies during the call and keeps nothing host-side.  "*_dev"
 *     pointers are device pointers on the context's GPU.
 *   - one nfm_ctx per GPU per process (one process per GPU); all work of a
 *     context is issued on one HIP stream.  Not thread-safe per context.
 *   - widths follow the reference: Nim int = int64_t, float64 = double.
 *     Parameter layouts at this boundary are the reference's:
 *       FM  P[nOrders][nComponents][nFeatures+nAugments] (model/factorization_machine.nim:31-34)
 *       FFM P[nFields][nFeatures][nComponents]           (model/field_aware_factorization_machine.nim:16-17)
 *       AdaGrad state g_sum/g_norm: P part [nOrders|nFields][nFeatures+nAugments][nComponents],
 *       w part [nFeatures], intercept scalar               (optimizer/adagrad.nim:53-55,155; model/params.nim:5-10)
 *   - there is NO CPU fallback: without a usable HIP device every compute entry
 *     point fails with NFM_ERR_HIP.
 */
#ifndef NIMFM_HIP_H
#define NIMFM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFM_OK 0
#define NFM_ERR_INVALID (-1)    /* ValueError in the reference */
#define NFM_ERR_HIP (-2)        /* HIP runtime / no device */
#define NFM_ERR_NOT_FITTED (-3) /* NotFittedError, model/fm_base.nim:13-15 */
#define NFM_ERR_NOMEM (-4)
#define NFM_ERR_UNSUPPORTED (-5)

typedef struct nfm_ctx nfm_ctx;
typedef struct nfm_dataset nfm_dataset;
typedef struct nfm_model nfm_model;
typedef struct nfm_opt nfm_opt;

/* enums mirror the reference's */
enum { NFM_TASK_REGRESSION = 0, NFM_TASK_CLASSIFICATION = 1 };                 /* model/fm_base.nim:6-8 */
enum { NFM_KIND_FM = 0, NFM_KIND_FFM = 1 };
enum { NFM_LOWER_EXPLICIT = 0, NFM_LOWER_AUGMENT = 1, NFM_LOWER_NONE = 2 };     /* model/factorization_machine.nim:6-9 */
enum { NFM_LOSS_SQUARED = 0, NFM_LOSS_SQUARED_HINGE = 1, NFM_LOSS_LOGISTIC = 2, NFM_LOSS_HUBER = 3 }; /* loss.nim */
enum { NFM_SCHED_CONSTANT = 0, NFM_SCHED_OPTIMAL = 1, NFM_SCHED_INVSCALING = 2, NFM_SCHED_PEGASOS = 3 }; /* optimizer/sgd.nim:7-11 */
/* NFM_MODE_SEQUENTIAL: one sample at a time in the given order -- the
 *   reference's single-thread semantics (optimizer/sgd.nim:294-308,
 *   optimizer/adagrad.nim:164-184): results equal to the CPU path's.  Run as
 *   a dependency window over the chip (0.9-3.0e6 samples/s; degree-2 FMs,
 *   several orders / degree <= 6, field-aware models; n_components <= 64,
 *   degree-2 FMs <= 128 -- their 128-double rows are read as two blocks of 64;
 *   calls of >= 2048 samples), else one sample in flight (fitLower = augment,
 *   more factors, rows too long for the window's LDS).  With a fitted
 *   intercept the window forms each prediction as intercept + (sum of the
 *   sample's other terms) -- same order, same dependencies, parameters
 *   within ~1e-11 relative of the one-sample-in-flight kernel; the environment
 *   variable NFM_SEQ_WIN_EXACT=1 selects the term-by-term window, BIT-equal to
 *   it (1.2e6 samples/s), NFM_SEQ_WIN_PAR=0 the bitwise-reproducible chain,
 *   NFM_SEQ_WIN=0 the one-sample-in-flight kernel itself.
 * NFM_MODE_MINIBATCH: this library's deterministic data-parallel rule
 *   (DESIGN.md section 4); replaces the reference's racy Hogwild drivers
 *   (optimizer/sgd_multi.nim:40-120, adagrad_multi.nim:39-115); equals the
 *   sequential rule at batch == 1. */
enum { NFM_MODE_SEQUENTIAL = 0, NFM_MODE_MINIBATCH = 1 };
/* sparsity-inducing regularisers with a matrix proximal operator (regularizer/l1.nim, l21.nim,
 * squaredl12.nim, squaredl21.nim); OmegaTI / OmegaCS have none and cannot drive MBPSGD */
enum { NFM_REG_L1 = 0, NFM_REG_L21 = 1, NFM_REG_SQUAREDL12 = 2, NFM_REG_SQUAREDL21 = 3 };

const char* nfm_last_error(void);
int32_t nfm_version(void);
/* number of visible HIP devices (0 and NFM_OK when none) */
int32_t nfm_device_count(int32_t* n);

/* ---- context: device + stream ---- */
/* hip_stream: a hipStream_t to issue on (e.g. the host framework's current
 * stream), or NULL to let the library create its own. */
int32_t nfm_ctx_create(int32_t device_id, void* hip_stream, nfm_ctx** out);
int32_t nfm_ctx_destroy(nfm_ctx* ctx);
int32_t nfm_ctx_synchronize(nfm_ctx* ctx);
/* per-kernel timing with HIP events on the context's stream (off by default).
 * nfm_ctx_timing_get: accumulated launches / milliseconds of one kernel family
 * ("row_phase", "col_phase", "predict", "sequential", "schedule", ...). */
int32_t nfm_ctx_timing_enable(nfm_ctx* ctx, int32_t on);
int32_t nfm_ctx_timing_reset(nfm_ctx* ctx);
int32_t nfm_ctx_timing_get(nfm_ctx* ctx, const char* family, int64_t* launches, double* total_ms);

/* ---- dataset: CSRDataset / CSRFieldDataset resident in HBM ----
 * replaces tensor/sparse.nim:9-12,19-24 (CSRMatrix, CSRFieldMatrix) +
 * dataset.nim:10-13,182-231 (row iterators incl. dummy features, which the
 * kernels generate on the fly) as the thing `fit`/`decisionFunction` iterate.
 * indices/fields are narrowed to int32 on the device.  y may be NULL for
 * predict-only datasets.  Rows may be stored in any order (dataset.nim:597-612).
 * A column id REPEATED inside a row: decisionFunction / predict / score take
 * it as the reference does (every entry is one more term of the ANOVA
 * recursion, kernels.nim:46-64); TRAINING needs distinct ids per row --
 * nfm_opt_epoch and nfm_opt_predict_all_with_grad return NFM_ERR_UNSUPPORTED
 * naming the first such row.  The reference does not forbid repeats there,
 * but what its step computes for them is an accident of its lazy scaling and
 * scratch layout (the row is rescaled once per ENTRY, optimizer/sgd.nim:134-143;
 * the later entry overwrites the earlier one's derivative dA[j], :176-188, and
 * the row's parameters are then stepped once per entry with it, :217-223);
 * merge repeated entries before training on such a matrix. */
int32_t nfm_dataset_create_csr(nfm_ctx* ctx, int64_t n_samples, int64_t n_features,
                               const int64_t* indptr /*n+1*/, const int64_t* indices /*nnz*/,
                               const double* data /*nnz*/, const int64_t* fields /*nnz or NULL*/,
                               int64_t n_fields, const double* y /*n or NULL*/,
                               nfm_dataset** out);
/* same, adopting arrays that already live on the device (no copy; the caller
 * keeps them alive until nfm_dataset_destroy). */
int32_t nfm_dataset_create_csr_device(nfm_ctx* ctx, int64_t n_samples, int64_t n_features,
                                      int64_t nnz, const int64_t* indptr_dev,
                                      const int32_t* indices_dev, const double* data_dev,
                                      const int32_t* fields_dev, int64_t n_fields,
                                      const double* y_dev, nfm_dataset** out);
/* ---- text ingest, parsed on the GPU ----
 * loadSVMLightFile (dataset.nim:562-632): "target idx:val idx:val ..." per line;
 * loadFFMFile (dataset.nim:696-790): "target field:idx:val ...".  The index
 * base is detected from the file (0-based iff an index 0 occurs, else 1-based,
 * :589/:732-733), nFeatures = max index + 1 - base (nFields likewise); a
 * positive n_features / n_fields smaller than that is NFM_ERR_INVALID (the
 * reference's ValueError, :623-626, :777-785), a larger one wins (:631, :788).
 * A negative index is NFM_ERR_INVALID ("Negative index is included.", :587).
 * Numbers are read as Nim's parseInt / parseFloat read them (correctly rounded
 * binary64).  Targets are stored with the dataset (nfm_dataset_get_targets).
 * nfm_dataset_parse_text: the same from a memory buffer. */
int32_t nfm_dataset_load_svmlight(nfm_ctx* ctx, const char* path, int64_t n_features,
                                  nfm_dataset** out);
int32_t nfm_dataset_load_ffm(nfm_ctx* ctx, const char* path, int64_t n_features, int64_t n_fields,
                             nfm_dataset** out);
int32_t nfm_dataset_parse_text(nfm_ctx* ctx, const char* text, int64_t len, int32_t with_fields,
                               int64_t n_features, int64_t n_fields, nfm_dataset** out);
/* The reference's binary out-of-core format (tensor/sparse_stream.nim:3-33;
 * newStreamCSRDataset, dataset.nim:170-174): "STREAMCSR" | {nRows, nCols, nnz:
 * int64; max, min: float64} | per row nnz: int64 + nnz x {val: float64, id:
 * int64} ("STREAMCSRFIELD": + nFields, elements {field, val, id}).  The whole
 * matrix is made resident in HBM (the reference streams row blocks through a
 * host cache).  y_path: raw float64 labels (loadStreamLabel, dataset.nim:
 * 1007-1014) or NULL.  A STREAMCSC file is NFM_ERR_UNSUPPORTED. */
int32_t nfm_dataset_load_stream(nfm_ctx* ctx, const char* x_path, const char* y_path,
                                nfm_dataset** out);
/* The same files in row blocks, for matrices that do not fit the HBM left over: the reference's out-of-core epoch walks
 * the file through a host-side cache of rows, block by block in file order, without shuffling
 * (readCache, tensor/sparse_stream.nim:232-270; optimizer/sgd_multi.nim:83-97, sgd.nim:297 `if X.nCached ==
 * X.nSamples and self.shuffle`).  nfm_stream_load_rows makes rows [row_begin, row_end) a resident dataset (targets from
 * the label file's rows, zeros without one); the host runs nfm_opt_epoch over block after block -- the optimizer's step
 * counter, scales and state simply continue -- and destroys each block's dataset when it is done with it. */
typedef struct nfm_stream nfm_stream;
int32_t nfm_stream_open(nfm_ctx* ctx, const char* x_path, const char* y_path /*or NULL*/, nfm_stream** out);
int32_t nfm_stream_shape(const nfm_stream* s, int64_t* n_samples, int64_t* n_features, int64_t* nnz, int64_t* n_fields);
int32_t nfm_stream_load_rows(nfm_stream* s, int64_t row_begin, int64_t row_end, nfm_dataset** out);
/* Starts loading rows [row_begin, row_end) in the background (a thread and a HIP stream of the stream object's own: file
 * walk, upload and split run beside the caller's epoch over the current block) and returns at once; the next
 * nfm_stream_load_rows of exactly that range hands the block over (and reports the load's error, if any), any other range
 * drops it.  One block ahead at most.  The reference's readCache (tensor/sparse_stream.nim:232-270) reads the next block
 * only when the epoch loop asks for it (optimizer/sgd_multi.nim:83-97). */
int32_t nfm_stream_prefetch_rows(nfm_stream* s, int64_t row_begin, int64_t row_end);
int32_t nfm_stream_close(nfm_stream* s);
/* convertSVMLightFile (dataset.nim:1017-1097): svmlight text -> STREAMCSR file +
 * raw float64 label file; the text is parsed on the GPU. */
int32_t nfm_convert_svmlight(nfm_ctx* ctx, const char* f_in, const char* f_out_x,
                             const char* f_out_y);
/* nSamples / nFeatures / nnz / nFields (dataset.nim:44-51, tensor/sparse.nim:26-40) */
int32_t nfm_dataset_shape(const nfm_dataset* ds, int64_t* n_samples, int64_t* n_features,
                          int64_t* nnz, int64_t* n_fields);
/* bytes of text, upload and parse time of the loader that made this dataset */
int32_t nfm_dataset_ingest_stats(const nfm_dataset* ds, int64_t* bytes, double* upload_ms,
                                 double* parse_ms);
/* read-back in the reference's widths (CSRMatrix data/indices/indptr, fields:
 * tensor/sparse.nim:9-12,19-24); any pointer may be NULL */
int32_t nfm_dataset_get_targets(nfm_dataset* ds, double* y /*n*/);
int32_t nfm_dataset_get_csr(nfm_dataset* ds, int64_t* indptr /*n+1*/, int64_t* indices /*nnz*/,
                            double* data /*nnz*/, int64_t* fields /*nnz*/);
/* model/fm_base.nim:29-36 checkTarget is applied by the optimizer according to
 * the model's task; this replaces the targets (host array, n). */
int32_t nfm_dataset_set_targets(nfm_dataset* ds, const double* y);
int32_t nfm_dataset_destroy(nfm_dataset* ds);

/* ---- model: FactorizationMachine / FieldAwareFactorizationMachine ---- */
typedef struct nfm_model_cfg {
  int32_t kind;          /* NFM_KIND_* */
  int32_t task;          /* NFM_TASK_* */
  int32_t degree;        /* FM only; FFM is degree 2 */
  int32_t n_components;  /* k -- no cap for FactorizationMachines: above 128 the factors of an order
                          * are kept as blocks of at most 128 on the device (an ANOVA kernel is a sum over
                          * the factors, kernels.nim:46-64), the layouts at this boundary stay the
                          * reference's; decisionFunction, SGD and AdaGrad in both modes take such models
                          * (MBPSGD does not: its matrix prox needs a feature's factors in one row).
                          * Field-aware models: k <= 128 outside NFM_MODE_SEQUENTIAL. */
  int32_t fit_lower;     /* NFM_LOWER_* (FM only) */
  int32_t fit_intercept;
  int32_t fit_linear;
  int32_t reserved;
  int64_t n_features;    /* d (without augments) */
  int64_t n_fields;      /* FFM only */
} nfm_model_cfg;

/* newFactorizationMachine (model/factorization_machine.nim:43-78) /
 * newFieldAwareFactorizationMachine (model/field_aware_factorization_machine.nim:24-46):
 * NFM_ERR_INVALID if degree < 1 or n_components < 1 (:65-69). */
int32_t nfm_model_create(nfm_ctx* ctx, const nfm_model_cfg* cfg, nfm_model** out);
/* nOrders / nAugments (model/factorization_machine.nim:81-97); for FFM
 * n_blocks = nFields and n_aug = 0. */
int32_t nfm_model_shape(const nfm_model* m, int32_t* n_blocks, int32_t* n_aug);
/* fm.init's result / a loaded model (model/factorization_machine.nim:125-139,
 * 183-220): marks the model initialised.  lams may be NULL (= ones, :78). */
int32_t nfm_model_set_params(nfm_model* m, const double* P, const double* w, double intercept,
                             const double* lams);
/* finalised parameters in the reference layout (what fm.P / fm.w /
 * fm.intercept hold after fit; optimizer/sgd.nim:327-328).  For AdaGrad call
 * nfm_opt_finalize first. */
int32_t nfm_model_get_params(nfm_model* m, double* P, double* w, double* intercept);
/* decisionFunction (model/factorization_machine.nim:100-122 -> kernels.nim:14-19,
 * 46-64; FFM: model/field_aware_factorization_machine.nim:52-76).
 * NFM_ERR_NOT_FITTED when not initialised, NFM_ERR_INVALID on nFeatures /
 * nFields mismatch (:114-115, FFM :60-64).  out: n doubles. */
int32_t nfm_decision_function(nfm_model* m, nfm_dataset* ds, double* out);
int32_t nfm_decision_function_device(nfm_model* m, nfm_dataset* ds, double* out_dev);
/* score (model/fm_base.nim:39-48): rmse for regression, accuracy of the signs for
 * classification, of decisionFunction(ds) against the dataset's targets -- computed
 * on the device, only the scalar comes back. */
int32_t nfm_score(nfm_model* m, nfm_dataset* ds, double* out);
/* metrics.nim:5-13 (rmse), :39-47 (accuracy of signs), :76-103 (rocauc, pos = 1) of
 * decisionFunction(ds) against the dataset's targets; any pointer may be NULL. */
int32_t nfm_metrics(nfm_model* m, nfm_dataset* ds, double* rmse, double* accuracy, double* rocauc);
/* ||P||^2 and ||w||^2 for optimizer/utils.nim:56-59 `regularization` (verbose). */
int32_t nfm_model_sqnorms(nfm_model* m, double* P_sq, double* w_sq);
/* device views for the data-parallel exchange: pointers to
 * the device-layout parameter buffers and their lengths in doubles. scalars
 * holds {scale_P, scale_w, intercept, 5 unused}. (Device layout: FM [nOrders][d+nAug][Kp], FFM [d][nFields][Kp]
 * with Kp >= k zero-padded -- opaque to an element-wise exchange.) The three buffers are pieces of
 * ONE allocation in the order [P | w | scalars] (gaps are zero padding), so a
 * single collective over [P_dev, scalars_dev + n_scalars) reconciles a replica. */
int32_t nfm_model_device_buffers(nfm_model* m, double** P_dev, int64_t* n_P, double** w_dev,
                                 int64_t* n_w, double** scalars_dev, int64_t* n_scalars);
int32_t nfm_model_destroy(nfm_model* m);

/* ---- optimizers ---- */
typedef struct nfm_sgd_cfg { /* newSGD, optimizer/sgd.nim:23-52 */
  double eta0, alpha0, alpha, beta, power, loss_param /* Huber threshold */;
  int32_t loss, scheduling, mode, reserved;
  int64_t batch; /* NFM_MODE_MINIBATCH only */
} nfm_sgd_cfg;

typedef struct nfm_adagrad_cfg { /* newAdaGrad, optimizer/adagrad.nim:20-44 */
  double eta0, alpha0, alpha, beta, eps, loss_param;
  int32_t loss, mode, track_viol /* 1 = keep the reference's sum|dP| (adagrad.nim:99) */, reserved;
  int64_t batch;
} nfm_adagrad_cfg;

/* newMBPSGD (optimizer/minibatch_psgd.nim:24-65): the reference's own mini-batch rule --
 * the gradient of `batch` samples averaged (updateGradient, :67-84), one step on ALL parameters
 * (Params.step, model/params.nim:90-98), then the regulariser's prox per order with
 * gamma * eta_P / (1 + eta_P * beta) (:112-120).  FactorizationMachine only.  SquaredL12 / SquaredL21 need
 * degree 2 (their initSGD raises, squaredl12.nim:103-105); reg_transpose: SquaredL12 column-wise (1, the
 * reference default) or row-wise (0); SquaredL21 only 0 (its default).
 * nfm_opt_epoch on such an optimizer runs (end - begin) / batch mini-batches over perm[begin..end) -- the
 * stream of sample indices the reference's inner loops consume (indices[ii], wrap-arounds included), so
 * end may exceed nSamples when perm is given and (end - begin) must be a multiple of batch; `it` advances
 * once per mini-batch (:121); viol_sum is 0 (the solver has none). */
typedef struct nfm_mbpsgd_cfg {
  double eta0, alpha0, alpha, beta, gamma, power, loss_param;
  int32_t loss, scheduling, reg /* NFM_REG_* */, reg_transpose;
  int64_t batch; /* miniBatchSize >= 1 */
} nfm_mbpsgd_cfg;

int32_t nfm_sgd_create(nfm_model* m, const nfm_sgd_cfg* cfg, nfm_opt** out);
int32_t nfm_adagrad_create(nfm_model* m, const nfm_adagrad_cfg* cfg, nfm_opt** out);
int32_t nfm_mbpsgd_create(nfm_model* m, const nfm_mbpsgd_cfg* cfg, nfm_opt** out);
/* predictAllWithGrad (optimizer/pgd.nim:70-103), the full-batch half of the proximal solvers, on an optimizer made
 * by nfm_mbpsgd_create (its loss is used; its state and `it` are untouched): y_pred[n] = the model's output on every
 * sample, dL[n] = dloss(y_i, y_pred_i), and the gradient of the MEAN loss -- grad_P in the training layout
 * [nOrders][d + nAugments][k] (the reference's grads.P), grad_w[d] (zeros unless fitLinear), *grad_b (0 unless
 * fitIntercept), *loss_sum = sum_i loss(y_i, y_pred_i).  Any output pointer may be NULL. */
int32_t nfm_opt_predict_all_with_grad(nfm_opt* o, nfm_dataset* ds, double* y_pred, double* dL,
                                      double* grad_P, double* grad_w, double* grad_b, double* loss_sum);
/* the optimizer's `it` (optimizer/sgd.nim:18,55-56; adagrad.nim:14,50): starts
 * at 1, +1 per sample; set to 1 to mimic a non-warm-start fit. */
int32_t nfm_opt_set_it(nfm_opt* o, int64_t it);
int32_t nfm_opt_get_it(nfm_opt* o, int64_t* it);
/* AdaGrad g_sum / g_norm (optimizer/adagrad.nim:15-16), reference layout. */
int32_t nfm_opt_get_state(nfm_opt* o, double* gsum_P, double* gnorm_P, double* gsum_w,
                          double* gnorm_w, double* gsum_b, double* gnorm_b);
int32_t nfm_opt_set_state(nfm_opt* o, const double* gsum_P, const double* gnorm_P,
                          const double* gsum_w, const double* gnorm_w, double gsum_b,
                          double gnorm_b);
/* The body of one epoch of fit (optimizer/sgd.nim:298-308, adagrad.nim:169-184;
 * FFM: sgd_ffm.nim:77-87, adagrad_ffm.nim:35-48) over samples
 * perm[begin..end) (perm NULL = identity; perm is the host-side shuffle,
 * sgd.nim:297).  Sub-ranges serve nCalls callbacks (sgd.nim:303-307).  Returns
 * the running sums the reference prints/tests: sum of loss(y_i, yhat_i) and
 * `viol`.  Classification targets are sign()-ed (fm_base.nim:32-34).
 * A range of more than 2^31 - 1 entries is walked as consecutive pieces
 * (whole mini-batches): the results are the one call's (not with a
 * data-parallel group attached; a device-drawn order shuffles inside a piece). */
int32_t nfm_opt_epoch(nfm_opt* o, nfm_dataset* ds, const int64_t* perm, int64_t begin,
                      int64_t end, double* loss_sum, double* viol_sum);
/* shuffle = true on the device (NFM_MODE_MINIBATCH): with seed >= 0 every nfm_opt_epoch call WITHOUT an explicit
 * permutation runs over a fresh random order of the samples [begin, end), drawn on the device from (seed, number of
 * such calls so far) -- the reference shuffles `indices` on the host once per epoch (optimizer/sgd.nim:297, Nim's global
 * generator, not reproduced: RNG parity is unpinned either way) -- and the batch plan of the NEXT such call is built on a
 * second stream while the current epoch runs.  seed < 0 switches it off (the default: perm == NULL is the dataset's own
 * order).  nfm_opt_get_perm: the n = end - begin sample ids of the most recent epoch call that ran in a permuted order
 * (device-drawn or passed in), so that any run can be replayed sample for sample. */
int32_t nfm_opt_set_shuffle(nfm_opt* o, int64_t seed);
/* A host that shuffles itself (the Nim host: shuffle(indices), optimizer/sgd.nim:297) can tell the library the
 * permutation of the NEXT epoch before it runs the current one: nfm_opt_announce_perm(o, perm_next, begin, end), then
 * nfm_opt_epoch(o, ds, perm_current, begin, end, ...).  The plan for perm_next is built on a second stream beside the
 * current epoch and used when the next nfm_opt_epoch call passes the SAME array (same pointer, same range); the array
 * must stay alive and unchanged until that call returns (so the host alternates between two index arrays).  An
 * announcement that is not followed up costs only the wasted build.  NFM_MODE_MINIBATCH; ignored otherwise. */
int32_t nfm_opt_announce_perm(nfm_opt* o, const int64_t* perm_next, int64_t begin, int64_t end);
int32_t nfm_opt_get_perm(nfm_opt* o, int64_t* perm /*n*/, int64_t n);
/* finalize (optimizer/sgd.nim:99-113; adagrad.nim:65-84): leaves the model's
 * parameters as the reference's fm.P/w/intercept after fit. Idempotent. */
int32_t nfm_opt_finalize(nfm_opt* o);
/* device views of the AdaGrad state for the data-parallel exchange; pieces of ONE
 * allocation in the order [gsum_P | gnorm_P | gsum_w | gnorm_w | gscalars]
 * (gaps are zero padding): [gsum_P, gscalars + 2) is one collective. */
int32_t nfm_opt_device_state(nfm_opt* o, double** gsum_P, double** gnorm_P, int64_t* n_P,
                             double** gsum_w, double** gnorm_w, int64_t* n_w,
                             double** gscalars /* {gsum_b, gnorm_b} */);
int32_t nfm_opt_destroy(nfm_opt* o);

/* ---- data-parallel groups: sample shards, replicated parameters, periodic exchange ----
 * Replaces the reference's Hogwild drivers (optimizer/sgd_multi.nim:40-120, adagrad_multi.nim:39-115,
 * sgd_ffm_multi.nim, adagrad_ffm_multi.nim: T threads, contiguous sample slices `nSamples div nThreads`,
 * ONE shared unsynchronised model) across GPUs: rank r trains on its own contiguous slice, resident in its GPU's
 * HBM (the host hands each rank its slice as that rank's nfm_dataset), the model is replicated, and nfm_opt_epoch
 * reconciles the replicas every `sync_period` mini-batches and exactly at the end of the call:
 *   SGD      replicas averaged;
 *   AdaGrad  the replicas' g_sum / g_norm increments since the last exchange summed (the state ONE process would hold
 *            after all shards' samples, optimizer/adagrad.nim:113-134).
 * With overlap != 0 a mid-epoch exchange is delayed by one period: the collective runs on the group's own stream beside
 * the next period's mini-batches and its result is folded in at the next sync point (a fixed point: results are
 * reproducible).  All ranks of a group call nfm_opt_epoch together, with equal step counters at the call's start; the
 * call returns the loss / viol sums over ALL ranks (sgd_multi.nim:98-101) and advances the step counter by the samples
 * of all ranks (the reference's threads share one counter, sgd_multi.nim:37).  After the call all replicas hold the
 * same bits.  NFM_MODE_MINIBATCH, SGD and AdaGrad (FM and FFM).
 *
 * nfm_dp_create: one PROCESS per GPU; the collective is ncclAllReduce(ncclDouble) of RCCL over xGMI on the single
 * parameter / state arena.  Rank 0 obtains an id with nfm_dp_unique_id and the host distributes its
 * NFM_DP_ID_BYTES bytes (file, environment, MPI, torch.distributed ...); nfm_dp_create is collective.
 * nfm_dp_create_local: `world` ranks inside ONE process, one nfm_ctx (and one host thread) each -- several GPUs with
 * peer access, or one GPU shared by all ranks; the collective is a rendezvous of the ranks' host threads and a
 * rank-ordered device-side sum.  out receives `world` handles. */
typedef struct nfm_dp nfm_dp;
#define NFM_DP_ID_BYTES 128
int32_t nfm_dp_unique_id(void* id /*[NFM_DP_ID_BYTES]*/);
int32_t nfm_dp_create(nfm_ctx* ctx, const void* id /*[NFM_DP_ID_BYTES]*/, int32_t rank, int32_t world, nfm_dp** out);
int32_t nfm_dp_create_local(nfm_ctx* const* ctxs /*[world]*/, int32_t world, nfm_dp** out /*[world]*/);
/* rank / world of the group, collectives issued and bytes contributed so far (any pointer may be NULL) */
int32_t nfm_dp_info(const nfm_dp* dp, int32_t* rank, int32_t* world, int64_t* n_collectives, int64_t* bytes);
int32_t nfm_dp_destroy(nfm_dp* dp);
/* attach (dp != NULL) or detach (dp == NULL) a group; sync_period: mini-batches between exchanges, 0 = only the exact
 * exchange at the end of every nfm_opt_epoch call */
int32_t nfm_opt_set_dp(nfm_opt* o, nfm_dp* dp, int64_t sync_period, int32_t overlap);  /* detach before nfm_dp_destroy */
/* How SGD combines the ranks' increments at an exchange (AdaGrad always adds its state increments up):
 * NFM_DP_MEAN (default) -- the replicas' mean: local SGD, as stable as one rank, but the model moves as far as ONE rank's
 * steps take it; NFM_DP_SUM -- every rank's steps land in the model, as every Hogwild thread's steps land in the
 * reference's shared one (optimizer/sgd_multi.nim:83-101): the progress of all ranks' steps, at the price of a step size
 * that is effectively multiplied by the number of ranks wherever their features overlap (keep sync_period small).
 * DESIGN.md section 6 has the measurements. */
/* AdaGrad: the state increments are SUMMED by default (with an exchange after every mini-batch that is synchronous
 * data-parallel AdaGrad -- the state ONE process would hold).  With long periods every rank fits its own shard between
 * exchanges and the summed state over-shoots by up to the number of ranks (measured: 4 ranks x 16 mini-batches between
 * exchanges left the held-out RMSE at 2.9 where one rank reaches 1.04, tools/dp_convergence.py); NFM_DP_STATE_MEAN averages
 * the ranks' state increments instead -- the replicas' mean, as stable as one rank at any period.
 * NFM_DP_AUTO (the default of every optimizer, so a host that only calls nfm_opt_set_dp gets it): SGD -- the mean;
 * AdaGrad -- the sum when sync_period == 1, NFM_DP_STATE_CROSS otherwise (sync_period 0 included; rounds 3-4: the averaged state).
 * NFM_DP_STATE_RSQRT: the ranks' increments weighted by 1 / sqrt(world) -- between the sum (weight 1) and the mean (1 / world). */
/* NFM_DP_STATE_CROSS (AdaGrad, round 5): g_sum increments summed; g_norm += sum_r dN_r + gamma ((sum_r dG_r)^2 - sum_r dG_r^2),
 * gamma = 0.1, never less than before -- the cross products of the ranks' increments inflate the norm where the ranks pushed the
 * same way from the same stale point (there the plain sum over-shoots) and vanish where they saw different things.  One
 * all-reduce of the same size (a rank sends dN_r - gamma dG_r^2 for dN_r).  Keeps one rank's progress per epoch at 2 / 4 / 8
 * ranks for every period, a whole epoch included (profiles/r05g_dp_convergence.txt); NFM_DP_AUTO picks it for AdaGrad beyond
 * one mini-batch per exchange. */
enum { NFM_DP_AUTO = -1, NFM_DP_MEAN = 0, NFM_DP_SUM = 1, NFM_DP_STATE_MEAN = 2, NFM_DP_STATE_RSQRT = 3, NFM_DP_STATE_CROSS = 4 };
int32_t nfm_opt_set_dp_combine(nfm_opt* o, int32_t combine);

/* SGD, NFM_MODE_MINIBATCH: how a mini-batch combines the per-sample steps (optimizer/sgd.nim:205-243) of the `c` samples
 * that touch one coordinate.  The reference's Hogwild threads (optimizer/sgd_multi.nim:83-101, maxThreads of them) each
 * apply their step at full strength to a shared model that is at most maxThreads steps stale.  Here: up to `cap` of a
 * batch's steps on a coordinate are SUMMED; a coordinate touched c > cap times receives cap / c of the sum (and cap / c
 * of the batch's decay exponent).  cap = 1 (the default) is the per-coordinate MEAN: always stable, but one epoch then
 * makes about 1 / c_bar of the sequential order's progress, c_bar = mean touch count per touched coordinate.  cap =
 * the reference's thread count gives its staleness; 16 matched one sequential epoch's held-out loss per epoch on the
 * bench workloads and stayed stable where the plain sum (cap = infinity) diverges on dense coordinates and the
 * intercept.  Larger mini-batches want the cap to follow: with cap about twice the batch's mean touch
 * count per coordinate (batch * nnz-per-row / n_features) the epochs to a held-out loss stay those of the sequential order
 * (measured up to 32 at batch 262144 of the headline shape and 64 at 524288, profiles/r05h_touch_cap_sweep.txt); a cap
 * below that rate turns the tail of the batch into the mean.  Deterministic for every value. */
int32_t nfm_opt_set_touch_cap(nfm_opt* o, double cap);
/* AdaGrad, NFM_MODE_MINIBATCH (round 5): what a mini-batch adds to a coordinate's g_norm (optimizer/adagrad.nim:122-124 adds
 * g^2 per sample).  All samples of a batch take their gradients from the batch-start parameters; summing their squares alone
 * does not see whether they agree, and a coordinate touched by many samples of a batch that push the same way is then stepped
 * as far as if the samples had been independent observations -- field-aware AdaGrad at batch 32768 needed 13 / 28 / more than
 * 40 epochs for the held-out loss of 1 / 3 / 10 epochs in the reference's order.  With gamma > 0 the norm grows by
 *   sum_i g_i^2 + gamma * max((sum_i g_i)^2 - sum_i g_i^2, 0)
 * -- the cross products of the batch's gradients (what squaring the batch gradient does in large-batch AdaGrad, and what
 * NFM_DP_STATE_CROSS does for the ranks of a group): 2 / 4 / 11 epochs at batch 32768, as at batch 2048.
 * A coordinate touched once is not affected, so batch == 1 stays the reference's step.  gamma = 0 (default): off.
 * Deterministic for every value. */
int32_t nfm_opt_set_ada_cross(nfm_opt* o, double gamma);

/* ---- host-side random numbers (no device work) ----
 * FactorizationMachine.init draws P with randomNormal (model/factorization_machine.nim:125-139,
 * tensor/tensor.nim:561-580: Box-Muller over rand(1.0); z = sqrt(-2 ln(1-x)) cos(2 pi y) and the sine twin go
 * to CONSECUTIVE elements of the row-major fill [nOrders][nComponents][nFeatures+nAugments] (FFM:
 * [nFields][nFeatures][nComponents]), an odd count leaves the twin unused) after randomize(randomState);
 * fit shuffles the sample order with the same global generator (optimizer/sgd.nim:297).  A Nim host keeps
 * calling Nim's own procs; these entry points give hosts in other languages the same procedures.  state is the
 * generator's two 64-bit words.  The generator (Nim 1.0 lib/pure/random.nim, xoroshiro128+) is outside the
 * reference tree and restated from memory: the PROCEDURE (pairing, fill order, shuffle loop) follows the
 * reference, the bit stream is unverified. */
int32_t nfm_rng_randomize(int64_t seed, uint64_t* state /*[2]*/);
int32_t nfm_rng_random_normal(uint64_t* state, int64_t n, double loc, double scale, double* out /*n*/);
int32_t nfm_rng_shuffle(uint64_t* state, int64_t* x, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
