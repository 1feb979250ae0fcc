// nimfm_amd/csrc/api.hip -- the C ABI of include/nimfm_hip.h over the HIP kernels.
// Host-side bookkeeping only (allocation, layout conversion launches, plan cache, epoch driver);
// every entry point validates shapes on the host before any kernel is launched.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "dp.h"
#include "ingest.h"
#include "mb.h"

using namespace nfm;

struct nfm_dataset {
  nfm_ctx* ctx = nullptr;
  CsrView v{};
  DevBuf indptr, indices, data, fields, y;
  bool has_y = false;
  int max_row = 0;
  int64_t repeats = 0, repeat_row = -1;  // entries that repeat a column id of their row (plan.h: predict-only data)
  int64_t ingest_bytes = 0;  // set by the text loaders (ingest.hip)
  double ingest_upload_ms = 0.0, ingest_parse_ms = 0.0;
  // Cache keys.  uid: process-unique, never reused -- what a Plan is keyed by (a raw nfm_dataset* can be handed out
  // again by `new` after a `delete`, for a dataset of the same shape and different structure).  serial: bumps whenever
  // one of the v.* device pointers changes (nfm_dataset_set_targets) -- a captured hipGraph holds those pointers.
  uint64_t uid = 0, serial = 0;
  CscIndex csc;  // column-major twin, built by the first plan that can use it (plan.hip)
};
static uint64_t next_dataset_uid() {
  static std::atomic<uint64_t> g{0};
  return ++g;
}

struct Span {  // a piece of an arena allocation
  void* p = nullptr;
  size_t bytes = 0;
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};
static size_t pad256(size_t b) { return (b + 255) / 256 * 256; }

struct nfm_model {
  uint64_t uid = 0;  // process-unique; optimizers refer to their model by it (an address can be handed out again)
  nfm_ctx* ctx = nullptr;
  nfm_model_cfg cfg{};
  // k: n_components; FMs with k > 128: the factors of an order are cut into kc device blocks of kb <= 128 factors
  // (ModelView::kc) -- nb counts DEVICE blocks (orders x kc; fields for field-aware models), no the reference's orders
  int nb = 0, no = 0, kc = 1, kb = 0, n_aug = 0, k = 0, Kp = 0, L = 0;
  int64_t d = 0, da = 0;
  // a wide FM of ONE order keeps its kc blocks FEATURE-major: the row of a feature is one contiguous run of kc * Kp doubles -- kc
  // blocks of Kp to the kernels that walk blocks (row(b, j) = j * kc + b), ONE row of kc * Kp factors to the one-sample-in-flight
  // kernel (seq_row_view: its pipelined step and its single ascending factor sum, as before round 5), blocks of 64 to the window
  // (seq_window_view).  Several orders: order-major blocks (the views above need one stride per block).
  bool wide_rows() const { return cfg.kind == NFM_KIND_FM && kc > 1 && no == 1; }
  int64_t bs_() const { return (cfg.kind == NFM_KIND_FFM || wide_rows()) ? 1 : da; }
  int64_t rs_() const { return (cfg.kind == NFM_KIND_FFM || wide_rows()) ? nb : 1; }
  // P, w and the scalars live back to back in ONE allocation ([P | w | scalars], each padded to 256 B,
  // padding zero) so the data-parallel exchange is a single collective over the arena
  DevBuf arena, lams;
  Span P, w, sc;
  bool initialized = false;
  ModelView view() const {
    ModelView m{};
    m.P = P.as<double>(); m.w = w.as<double>(); m.sc = sc.as<double>(); m.lams = lams.as<double>();
    m.d = d; m.da = da; m.nb = nb; m.k = kb; m.Kp = Kp; m.L = L; m.kc = kc;
    m.bs = bs_(); m.rs = rs_();
    m.degree = cfg.kind == NFM_KIND_FFM ? 2 : cfg.degree;
    m.n_aug = n_aug; m.kind = cfg.kind; m.fit_linear = cfg.fit_linear; m.fit_intercept = cfg.fit_intercept;
    m.task = cfg.task;
    return m;
  }
  int64_t nP() const { return (int64_t)nb * da * Kp; }
  int64_t n_ref() const { return (int64_t)no * k * da; }  // elements of the reference's P (and of each AdaGrad state tensor)
};

// ---- reference layouts <-> device blocks (util.hip does one block range at a time) ----
// FM parameters: reference [no][k][da] <-> device [no * kc][da][Kp]; block o * kc + c holds the factors c * kb ... of order o
static int fm_params_to_device(nfm_model* m, const double* src_ref, double* dst_dev) {
  if (m->kc == 1) return launch_fm_to_device(m->ctx, src_ref, dst_dev, m->nb, m->k, m->Kp, m->da);
  for (int o = 0; o < m->no; ++o)
    for (int c = 0; c < m->kc; ++c) {
      const int kk = std::min(m->kb, m->k - c * m->kb);
      NFM_TRY(launch_fm_to_device(m->ctx, src_ref + ((size_t)o * m->k + (size_t)c * m->kb) * m->da, dst_dev, 1, kk, m->Kp, m->da, m->bs_(),
                                  m->rs_(), o * m->kc + c));
    }
  return NFM_OK;
}
static int fm_params_from_device(nfm_model* m, const double* src_dev, double* dst_ref, const double* scale_dev) {
  if (m->kc == 1) return launch_fm_from_device(m->ctx, src_dev, dst_ref, m->nb, m->k, m->Kp, m->da, scale_dev);
  for (int o = 0; o < m->no; ++o)
    for (int c = 0; c < m->kc; ++c) {
      const int kk = std::min(m->kb, m->k - c * m->kb);
      NFM_TRY(launch_fm_from_device(m->ctx, src_dev, dst_ref + ((size_t)o * m->k + (size_t)c * m->kb) * m->da, 1, kk, m->Kp, m->da, scale_dev,
                                    m->bs_(), m->rs_(), o * m->kc + c));
    }
  return NFM_OK;
}
// row tensors in the training layout (AdaGrad state, gradients): reference [no * da][k] <-> device [nb * da][Kp]
static int rows_to_device(nfm_model* m, const double* src_ref, double* dst_dev, double pad) {
  const int major = m->cfg.kind == NFM_KIND_FFM ? m->nb : 0;  // (feature-major field rows, ModelView::row)
  if (m->kc == 1) return launch_rows_to_device(m->ctx, src_ref, dst_dev, (int64_t)m->nb * m->da, m->k, m->Kp, pad, major);
  return launch_rows_split_to_device(m->ctx, src_ref, dst_dev, m->no, m->da, m->k, m->kc, m->kb, m->Kp, pad, m->bs_(), m->rs_());
}
static int rows_from_device(nfm_model* m, const double* src_dev, double* dst_ref, const double* scale_dev) {
  const int major = m->cfg.kind == NFM_KIND_FFM ? m->nb : 0;
  if (m->kc == 1) return launch_rows_from_device(m->ctx, src_dev, dst_ref, (int64_t)m->nb * m->da, m->k, m->Kp, scale_dev, major);
  return launch_rows_split_from_device(m->ctx, src_dev, dst_ref, m->no, m->da, m->k, m->kc, m->kb, m->Kp, scale_dev, m->bs_(), m->rs_());
}

struct nfm_opt {
  nfm_ctx* ctx = nullptr;  // kept separately: the optimizer may outlive its model handle
  nfm_model* m = nullptr;  // valid only while model_of(o) finds m_uid among the live models
  uint64_t m_uid = 0;
  int kind = OPT_SGD, mode = NFM_MODE_SEQUENTIAL;
  int64_t batch = 1, it = 1;
  OptView o{};
  // AdaGrad state, one allocation [G | N | Gw | Nw | gscalars] (each padded to 256 B, padding zero)
  DevBuf state_arena, out2, perm_dev;
  Span G, N, Gw, Nw, gsc;
  bool state_ready = false;
  MbWork W;
  std::unique_ptr<Plan> plan;
  std::unique_ptr<SeqWin> seqwin;  // NFM_MODE_SEQUENTIAL: dependency table and mailboxes of the window kernel (seqwin.hip)
  // predictAllWithGrad: the one-batch plan of a dataset and its scratch, kept between calls (PGD-style solvers ask for
  // the full gradient of the same data once per iteration)
  MbWork Wg;
  std::unique_ptr<Plan> grad_plan;
  // data-parallel group (dp.h): when set, nfm_opt_epoch reconciles the replicas every dp_sync_period mini-batches
  // (delayed by one period when dp_overlap) and exactly at the end of the call
  // device-side shuffle (nfm_opt_set_shuffle): every epoch call without an explicit permutation draws a fresh order on
  // the device; the plan of the NEXT epoch is built on a second stream while the current epoch runs
  int64_t shuffle_seed = -1;
  uint64_t shuffle_epoch = 0;
  DevBuf perm_gen, perm_next;
  std::unique_ptr<Plan> next_plan;
  uint64_t next_plan_epoch = 0;
  bool next_plan_ready = false;
  hipStream_t plan_stream = nullptr;
  double* out2_pinned = nullptr;
  // nfm_opt_announce_perm: the host's permutation for the NEXT epoch call (kept alive by the caller); its plan is
  // built beside the current epoch like the device-drawn one's
  const int64_t* announced = nullptr;
  int64_t announced_begin = 0, announced_end = 0;
  const int64_t* next_plan_perm = nullptr;  // the host array next_plan was built from
  int64_t next_probe[64] = {0};  // entries of that array at 64 evenly spaced places: it must come back unchanged
  nfm_dp* dp = nullptr;
  uint64_t dp_uid = 0;  // the group is checked against the live groups before every use (it may have been destroyed)
  int64_t dp_sync_period = 0;
  bool dp_overlap = true;
  int dp_combine = NFM_DP_AUTO;  // NFM_DP_AUTO: SGD the mean; AdaGrad summed at sync_period 1, its state increments averaged otherwise
  DevBuf dp_sums;
};

// live models by uid: an optimizer whose model was destroyed (and whose address may since belong to a model of
// another shape) is refused instead of writing through a stale pointer
static std::mutex g_models_mu;
static std::map<uint64_t, nfm_model*> g_models;
static uint64_t register_model(nfm_model* m) {
  static uint64_t next = 0;
  std::lock_guard<std::mutex> lk(g_models_mu);
  g_models[++next] = m;
  return next;
}
static void unregister_model(uint64_t uid) {
  std::lock_guard<std::mutex> lk(g_models_mu);
  g_models.erase(uid);
}
static int model_of(nfm_opt* o, nfm_model** out) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  std::lock_guard<std::mutex> lk(g_models_mu);
  auto it = g_models.find(o->m_uid);
  NFM_CHECK(it != g_models.end() && it->second == o->m, NFM_ERR_INVALID,
            "the optimizer's model was destroyed; create the optimizer again for the new model");
  *out = it->second;
  return NFM_OK;
}

static int use_device(nfm_ctx* ctx) {
  NFM_HIP_CHECK(hipSetDevice(ctx->device));
  return NFM_OK;
}

extern "C" {

const char* nfm_last_error(void) { return nfm::last_error(); }
int32_t nfm_version(void) { return 100; }

int32_t nfm_device_count(int32_t* n) {
  NFM_CHECK(n, NFM_ERR_INVALID, "null out pointer");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *n = c;
  return NFM_OK;
}

int32_t nfm_ctx_create(int32_t device_id, void* hip_stream, nfm_ctx** out) {
  NFM_CHECK(out, NFM_ERR_INVALID, "null out pointer");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c == 0)
    return set_error(NFM_ERR_HIP, "no HIP device available (%s); libnimfm_hip has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  NFM_CHECK(device_id >= 0 && device_id < c, NFM_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, c);
  std::unique_ptr<nfm_ctx> ctx(new nfm_ctx());
  ctx->device = device_id;
  NFM_HIP_CHECK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  NFM_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hip_stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  } else {
    NFM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  *out = ctx.release();
  return NFM_OK;
}

int32_t nfm_ctx_destroy(nfm_ctx* ctx) {
  if (!ctx) return NFM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->timing.pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
  for (auto e : ctx->timing.pool) (void)hipEventDestroy(e);
  delete ctx->predict_pf;  // (the stream has drained)
  delete ctx->seq_scratch;
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return NFM_OK;
}

int32_t nfm_ctx_synchronize(nfm_ctx* ctx) {
  NFM_CHECK(ctx, NFM_ERR_INVALID, "null ctx");
  NFM_TRY(use_device(ctx));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return NFM_OK;
}

int32_t nfm_ctx_timing_enable(nfm_ctx* ctx, int32_t on) {
  NFM_CHECK(ctx, NFM_ERR_INVALID, "null ctx");
  NFM_TRY(timing_flush(ctx));
  ctx->timing.enabled = on != 0;
  return NFM_OK;
}
int32_t nfm_ctx_timing_reset(nfm_ctx* ctx) {
  NFM_CHECK(ctx, NFM_ERR_INVALID, "null ctx");
  NFM_TRY(timing_flush(ctx));
  ctx->timing.acc.clear();
  return NFM_OK;
}
int32_t nfm_ctx_timing_get(nfm_ctx* ctx, const char* family, int64_t* launches, double* total_ms) {
  NFM_CHECK(ctx && family, NFM_ERR_INVALID, "null argument");
  NFM_TRY(timing_flush(ctx));
  auto it = ctx->timing.acc.find(family);
  if (launches) *launches = it == ctx->timing.acc.end() ? 0 : it->second.launches;
  if (total_ms) *total_ms = it == ctx->timing.acc.end() ? 0.0 : it->second.ms;
  return NFM_OK;
}

// ------------------------------------------------------------------ dataset
static int validate_csr_host(int64_t n, int64_t d, const int64_t* indptr, const int64_t* indices, const int64_t* fields,
                             int64_t n_fields, int* max_row) {
  NFM_CHECK(n >= 0 && d >= 0, NFM_ERR_INVALID, "negative shape");
  NFM_CHECK(d < (int64_t)2147483647 - 64, NFM_ERR_UNSUPPORTED, "n_features must fit int32");
  NFM_CHECK(indptr, NFM_ERR_INVALID, "null indptr");
  NFM_CHECK(indptr[0] == 0, NFM_ERR_INVALID, "indptr[0] != 0");
  int64_t mr = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t len = indptr[i + 1] - indptr[i];
    NFM_CHECK(len >= 0, NFM_ERR_INVALID, "indptr not non-decreasing at row %lld", (long long)i);
    mr = std::max(mr, len);
  }
  NFM_CHECK(mr < (1 << 30), NFM_ERR_UNSUPPORTED, "row too long");
  const int64_t nnz = indptr[n];
  NFM_CHECK(nnz == 0 || indices, NFM_ERR_INVALID, "null indices");
  for (int64_t q = 0; q < nnz; ++q) {
    NFM_CHECK(indices[q] >= 0 && indices[q] < d, NFM_ERR_INVALID, "column index %lld out of range [0,%lld) at nnz %lld",
              (long long)indices[q], (long long)d, (long long)q);
    if (fields)
      NFM_CHECK(fields[q] >= 0 && fields[q] < n_fields, NFM_ERR_INVALID, "field %lld out of range [0,%lld) at nnz %lld",
                (long long)fields[q], (long long)n_fields, (long long)q);
  }
  *max_row = (int)mr;
  return NFM_OK;
}

int32_t nfm_dataset_create_csr(nfm_ctx* ctx, int64_t n, int64_t d, const int64_t* indptr, const int64_t* indices,
                               const double* data, const int64_t* fields, int64_t n_fields, const double* y,
                               nfm_dataset** out) {
  NFM_CHECK(ctx && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  int max_row = 0;
  NFM_TRY(validate_csr_host(n, d, indptr, indices, fields, n_fields, &max_row));
  const int64_t nnz = indptr[n];
  NFM_CHECK(nnz == 0 || data, NFM_ERR_INVALID, "null data");
  std::unique_ptr<nfm_dataset> ds(new nfm_dataset());
  ds->ctx = ctx;
  ds->uid = next_dataset_uid();
  ds->max_row = max_row;
  hipStream_t st = ctx->stream;
  NFM_TRY(ds->indptr.alloc(sizeof(int64_t) * (n + 1)));
  NFM_TRY(ds->indices.alloc(sizeof(int32_t) * nnz));
  NFM_TRY(ds->data.alloc(sizeof(double) * nnz));
  NFM_HIP_CHECK(hipMemcpyAsync(ds->indptr.p, indptr, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
  if (nnz > 0) {
    DevBuf wide;
    NFM_TRY(wide.alloc(sizeof(int64_t) * nnz));
    NFM_HIP_CHECK(hipMemcpyAsync(wide.p, indices, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, st));
    NFM_TRY(launch_narrow_i64_i32(ctx, wide.as<int64_t>(), ds->indices.as<int32_t>(), nnz));
    if (fields) {
      NFM_TRY(ds->fields.alloc(sizeof(int32_t) * nnz));
      NFM_HIP_CHECK(hipMemcpyAsync(wide.p, fields, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, st));
      NFM_TRY(launch_narrow_i64_i32(ctx, wide.as<int64_t>(), ds->fields.as<int32_t>(), nnz));
    }
    NFM_HIP_CHECK(hipMemcpyAsync(ds->data.p, data, sizeof(double) * nnz, hipMemcpyHostToDevice, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  if (y) {
    NFM_TRY(ds->y.alloc(sizeof(double) * n));
    NFM_HIP_CHECK(hipMemcpyAsync(ds->y.p, y, sizeof(double) * n, hipMemcpyHostToDevice, st));
    ds->has_y = true;
  }
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  ds->v.indptr = ds->indptr.as<int64_t>();
  ds->v.indices = ds->indices.as<int32_t>();
  ds->v.data = ds->data.as<double>();
  ds->v.fields = fields ? ds->fields.as<int32_t>() : nullptr;
  ds->v.y = ds->has_y ? ds->y.as<double>() : nullptr;
  ds->v.n = n; ds->v.d = d; ds->v.nnz = nnz; ds->v.n_fields = fields ? (int32_t)n_fields : 0;
  ds->v.max_row = max_row;
  NFM_TRY(check_rows_distinct(ctx, ds->v, &ds->repeats, &ds->repeat_row));
  *out = ds.release();
  return NFM_OK;
}

int32_t nfm_dataset_create_csr_device(nfm_ctx* ctx, int64_t n, int64_t d, int64_t nnz, const int64_t* indptr_dev,
                                      const int32_t* indices_dev, const double* data_dev, const int32_t* fields_dev,
                                      int64_t n_fields, const double* y_dev, nfm_dataset** out) {
  NFM_CHECK(ctx && out && indptr_dev, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(n >= 0 && d >= 0 && nnz >= 0 && d < (int64_t)2147483647 - 64, NFM_ERR_INVALID, "bad shape");
  NFM_CHECK(nnz == 0 || (indices_dev && data_dev), NFM_ERR_INVALID, "null indices/data");
  NFM_TRY(use_device(ctx));
  // the row-length bound needs indptr on the host once; values are trusted to be in range
  // (device-resident synthetic data); index range is checked by the caller's generator.
  std::vector<int64_t> ip((size_t)n + 1);
  NFM_HIP_CHECK(hipMemcpyAsync(ip.data(), indptr_dev, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  NFM_CHECK(ip[0] == 0 && ip[n] == nnz, NFM_ERR_INVALID, "indptr[0] != 0 or indptr[n] != nnz");
  int64_t mr = 0;
  for (int64_t i = 0; i < n; ++i) {
    NFM_CHECK(ip[i + 1] >= ip[i], NFM_ERR_INVALID, "indptr not non-decreasing at row %lld", (long long)i);
    mr = std::max(mr, ip[i + 1] - ip[i]);
  }
  std::unique_ptr<nfm_dataset> ds(new nfm_dataset());
  ds->ctx = ctx;
  ds->uid = next_dataset_uid();
  ds->max_row = (int)mr;
  ds->has_y = y_dev != nullptr;
  ds->v.indptr = indptr_dev; ds->v.indices = indices_dev; ds->v.data = data_dev; ds->v.fields = fields_dev;
  ds->v.y = y_dev; ds->v.n = n; ds->v.d = d; ds->v.nnz = nnz; ds->v.n_fields = fields_dev ? (int32_t)n_fields : 0;
  ds->v.max_row = (int32_t)mr;
  NFM_TRY(check_rows_distinct(ctx, ds->v, &ds->repeats, &ds->repeat_row));
  *out = ds.release();
  return NFM_OK;
}

int32_t nfm_dataset_set_targets(nfm_dataset* ds, const double* y) {
  NFM_CHECK(ds && y, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ds->ctx));
  NFM_TRY(ds->y.ensure(sizeof(double) * ds->v.n));
  NFM_HIP_CHECK(hipMemcpyAsync(ds->y.p, y, sizeof(double) * ds->v.n, hipMemcpyHostToDevice, ds->ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ds->ctx->stream));
  ds->has_y = true;
  if (ds->v.y != ds->y.as<double>()) ds->serial++;  // a captured epoch graph holds the old pointer
  ds->v.y = ds->y.as<double>();
  return NFM_OK;
}

// ---- text ingest (ingest.hip) ----
static int dataset_from_ingest(nfm_ctx* ctx, IngestResult& r, bool with_fields, int64_t n_features, int64_t n_fields,
                               nfm_dataset** out) {
  // dataset.nim:623-631, 777-790: a given nFeatures / nFields smaller than what the file needs is an error,
  // a larger one wins
  if (n_features > 0 && r.d > n_features)
    return set_error(NFM_ERR_INVALID, "nFeatures is %lld but dataset has at least %lld features.", (long long)n_features,
                     (long long)r.d);
  if (with_fields && n_fields > 0 && r.n_fields > n_fields)
    return set_error(NFM_ERR_INVALID, "nFields is %lld but dataset has at least %lld fields.", (long long)n_fields,
                     (long long)r.n_fields);
  std::unique_ptr<nfm_dataset> ds(new nfm_dataset());
  ds->ctx = ctx;
  ds->uid = next_dataset_uid();
  ds->max_row = r.max_row;
  ds->indptr.take(r.indptr);
  ds->indices.take(r.indices);
  ds->data.take(r.data);
  ds->y.take(r.y);
  if (with_fields) ds->fields.take(r.fields);
  ds->has_y = true;
  ds->v.indptr = ds->indptr.as<int64_t>();
  ds->v.indices = ds->indices.as<int32_t>();
  ds->v.data = ds->data.as<double>();
  ds->v.fields = with_fields ? ds->fields.as<int32_t>() : nullptr;
  ds->v.y = ds->y.as<double>();
  ds->v.n = r.n;
  ds->v.d = std::max<int64_t>(r.d, n_features);
  ds->v.nnz = r.nnz;
  ds->v.n_fields = with_fields ? (int32_t)std::max<int64_t>(r.n_fields, n_fields) : 0;
  ds->v.max_row = r.max_row;
  ds->ingest_bytes = r.bytes;
  ds->ingest_upload_ms = r.upload_ms;
  ds->ingest_parse_ms = r.parse_ms;
  NFM_TRY(check_rows_distinct(ctx, ds->v, &ds->repeats, &ds->repeat_row));
  *out = ds.release();
  return NFM_OK;
}

int32_t nfm_dataset_load_svmlight(nfm_ctx* ctx, const char* path, int64_t n_features, nfm_dataset** out) {
  NFM_CHECK(ctx && path && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  IngestResult r;
  NFM_TRY(ingest_text(ctx, path, nullptr, 0, false, &r));
  return dataset_from_ingest(ctx, r, false, n_features, 0, out);
}

int32_t nfm_dataset_load_ffm(nfm_ctx* ctx, const char* path, int64_t n_features, int64_t n_fields, nfm_dataset** out) {
  NFM_CHECK(ctx && path && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  IngestResult r;
  NFM_TRY(ingest_text(ctx, path, nullptr, 0, true, &r));
  return dataset_from_ingest(ctx, r, true, n_features, n_fields, out);
}

int32_t nfm_dataset_parse_text(nfm_ctx* ctx, const char* text, int64_t len, int32_t with_fields, int64_t n_features,
                               int64_t n_fields, nfm_dataset** out) {
  NFM_CHECK(ctx && out && len >= 0, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  IngestResult r;
  NFM_TRY(ingest_text(ctx, nullptr, text, len, with_fields != 0, &r));
  return dataset_from_ingest(ctx, r, with_fields != 0, n_features, n_fields, out);
}

int32_t nfm_dataset_load_stream(nfm_ctx* ctx, const char* x_path, const char* y_path, nfm_dataset** out) {
  NFM_CHECK(ctx && x_path && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  IngestResult r;
  NFM_TRY(ingest_stream(ctx, x_path, y_path, &r));
  return dataset_from_ingest(ctx, r, r.n_fields > 0, -1, -1, out);
}

struct nfm_stream {
  nfm_ctx* ctx = nullptr;
  StreamFile f;
  // One row block loaded ahead (nfm_stream_prefetch_rows) by a thread of the stream object's own -- ONE thread for all
  // blocks: the device-memory cache hands a thread its own released blocks back without a device-wide wait (util.hip),
  // so the block-sized staging buffer of the previous load is reused while the caller's epoch is still running -- on a
  // HIP stream of its own.
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  bool job = false, busy = false, quit = false;
  hipStream_t pf_stream = nullptr;
  bool pf_active = false;
  int64_t pf_r0 = 0, pf_r1 = 0;
  int pf_rc = NFM_OK;
  std::string pf_err;
  std::unique_ptr<IngestResult> pf_res;
  void run() {
    if (hipSetDevice(ctx->device) != hipSuccess) {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
      busy = false;
      pf_rc = NFM_ERR_HIP;
      pf_err = "the prefetch thread could not select the context's device";
      cv.notify_all();
      return;
    }
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      cv.wait(lk, [&] { return job || quit; });
      if (quit) return;
      job = false;
      lk.unlock();
      const int rc = f.load_rows(ctx, pf_r0, pf_r1, pf_res.get(), pf_stream);
      std::string err = rc != NFM_OK ? nfm_last_error() : "";  // (the message lives in this thread's slot)
      lk.lock();
      pf_rc = rc;
      pf_err = err;
      busy = false;
      cv.notify_all();
    }
  }
  void join() {  // until the block being loaded, if any, is there
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !busy; });
  }
  ~nfm_stream() {
    join();
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    if (worker.joinable()) worker.join();
    pf_res.reset();
    if (pf_stream) (void)hipStreamDestroy(pf_stream);
  }
};

int32_t nfm_stream_open(nfm_ctx* ctx, const char* x_path, const char* y_path, nfm_stream** out) {
  NFM_CHECK(ctx && x_path && out, NFM_ERR_INVALID, "null argument");
  std::unique_ptr<nfm_stream> s(new nfm_stream());
  s->ctx = ctx;
  NFM_TRY(StreamFile::open_file(x_path, y_path, &s->f));
  *out = s.release();
  return NFM_OK;
}

int32_t nfm_stream_shape(const nfm_stream* s, int64_t* n_samples, int64_t* n_features, int64_t* nnz, int64_t* n_fields) {
  NFM_CHECK(s, NFM_ERR_INVALID, "null stream");
  if (n_samples) *n_samples = s->f.n;
  if (n_features) *n_features = s->f.d;
  if (nnz) *nnz = s->f.nnz;
  if (n_fields) *n_fields = s->f.nf;
  return NFM_OK;
}

int32_t nfm_stream_prefetch_rows(nfm_stream* s, int64_t row_begin, int64_t row_end) {
  NFM_CHECK(s, NFM_ERR_INVALID, "null stream");
  NFM_CHECK(row_begin >= 0 && row_begin <= row_end && row_end <= s->f.n, NFM_ERR_INVALID, "rows [%lld,%lld) outside [0,%lld)",
            (long long)row_begin, (long long)row_end, (long long)s->f.n);
  NFM_TRY(use_device(s->ctx));
  s->join();  // at most one block ahead; an unclaimed one is dropped
  s->pf_res.reset();
  if (!s->pf_stream) NFM_HIP_CHECK(hipStreamCreateWithFlags(&s->pf_stream, hipStreamNonBlocking));
  s->pf_active = true;
  s->pf_r0 = row_begin;
  s->pf_r1 = row_end;
  s->pf_rc = NFM_OK;
  s->pf_err.clear();
  s->pf_res.reset(new IngestResult());
  if (!s->worker.joinable()) s->worker = std::thread([s]() { s->run(); });
  {
    std::lock_guard<std::mutex> lk(s->mu);
    NFM_CHECK(!s->quit, NFM_ERR_HIP, "the prefetch thread could not select device %d", s->ctx->device);
    s->job = true;
    s->busy = true;
  }
  s->cv.notify_all();
  return NFM_OK;
}

int32_t nfm_stream_load_rows(nfm_stream* s, int64_t row_begin, int64_t row_end, nfm_dataset** out) {
  NFM_CHECK(s && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(s->ctx));
  s->join();
  if (s->pf_active && s->pf_r0 == row_begin && s->pf_r1 == row_end) {  // the block asked for ahead
    s->pf_active = false;
    std::unique_ptr<IngestResult> r(std::move(s->pf_res));
    if (s->pf_rc != NFM_OK) return set_error(s->pf_rc, "%s", s->pf_err.c_str());
    return dataset_from_ingest(s->ctx, *r, r->n_fields > 0, -1, -1, out);
  }
  s->pf_active = false;
  s->pf_res.reset();
  IngestResult r;
  NFM_TRY(s->f.load_rows(s->ctx, row_begin, row_end, &r));
  return dataset_from_ingest(s->ctx, r, r.n_fields > 0, -1, -1, out);
}

int32_t nfm_stream_close(nfm_stream* s) {
  if (s && use_device(s->ctx) != NFM_OK) return NFM_ERR_HIP;
  delete s;
  return NFM_OK;
}

int32_t nfm_convert_svmlight(nfm_ctx* ctx, const char* f_in, const char* f_out_x, const char* f_out_y) {
  NFM_CHECK(ctx && f_in && f_out_x && f_out_y, NFM_ERR_INVALID, "null argument");
  NFM_TRY(use_device(ctx));
  return convert_svmlight(ctx, f_in, f_out_x, f_out_y);
}

int32_t nfm_dataset_shape(const nfm_dataset* ds, int64_t* n_samples, int64_t* n_features, int64_t* nnz, int64_t* n_fields) {
  NFM_CHECK(ds, NFM_ERR_INVALID, "null dataset");
  if (n_samples) *n_samples = ds->v.n;
  if (n_features) *n_features = ds->v.d;
  if (nnz) *nnz = ds->v.nnz;
  if (n_fields) *n_fields = ds->v.n_fields;
  return NFM_OK;
}

int32_t nfm_dataset_ingest_stats(const nfm_dataset* ds, int64_t* bytes, double* upload_ms, double* parse_ms) {
  NFM_CHECK(ds, NFM_ERR_INVALID, "null dataset");
  if (bytes) *bytes = ds->ingest_bytes;
  if (upload_ms) *upload_ms = ds->ingest_upload_ms;
  if (parse_ms) *parse_ms = ds->ingest_parse_ms;
  return NFM_OK;
}

int32_t nfm_dataset_get_targets(nfm_dataset* ds, double* y) {
  NFM_CHECK(ds && y, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(ds->has_y, NFM_ERR_INVALID, "the dataset has no targets");
  NFM_TRY(use_device(ds->ctx));
  NFM_HIP_CHECK(hipMemcpyAsync(y, ds->v.y, sizeof(double) * ds->v.n, hipMemcpyDeviceToHost, ds->ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ds->ctx->stream));
  return NFM_OK;
}

int32_t nfm_dataset_get_csr(nfm_dataset* ds, int64_t* indptr, int64_t* indices, double* data, int64_t* fields) {
  NFM_CHECK(ds, NFM_ERR_INVALID, "null dataset");
  NFM_TRY(use_device(ds->ctx));
  hipStream_t st = ds->ctx->stream;
  const int64_t n = ds->v.n, nnz = ds->v.nnz;
  if (indptr) NFM_HIP_CHECK(hipMemcpyAsync(indptr, ds->v.indptr, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, st));
  if (data && nnz) NFM_HIP_CHECK(hipMemcpyAsync(data, ds->v.data, sizeof(double) * nnz, hipMemcpyDeviceToHost, st));
  std::vector<int32_t> tmp;
  if ((indices || fields) && nnz) tmp.resize((size_t)nnz);
  if (indices && nnz) {
    NFM_HIP_CHECK(hipMemcpyAsync(tmp.data(), ds->v.indices, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    for (int64_t q = 0; q < nnz; ++q) indices[q] = tmp[q];
  }
  if (fields && nnz) {
    NFM_CHECK(ds->v.fields, NFM_ERR_INVALID, "the dataset has no fields");
    NFM_HIP_CHECK(hipMemcpyAsync(tmp.data(), ds->v.fields, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    for (int64_t q = 0; q < nnz; ++q) fields[q] = tmp[q];
  }
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  return NFM_OK;
}

int32_t nfm_dataset_destroy(nfm_dataset* ds) {
  if (!ds) return NFM_OK;
  (void)hipSetDevice(ds->ctx->device);
  (void)hipStreamSynchronize(ds->ctx->stream);
  delete ds;
  return NFM_OK;
}

// ------------------------------------------------------------------ model
int32_t nfm_model_create(nfm_ctx* ctx, const nfm_model_cfg* cfg, nfm_model** out) {
  NFM_CHECK(ctx && cfg && out, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(cfg->kind == NFM_KIND_FM || cfg->kind == NFM_KIND_FFM, NFM_ERR_INVALID, "bad model kind");
  NFM_CHECK(cfg->n_components >= 1, NFM_ERR_INVALID, "nComponents < 1.");
  NFM_CHECK(cfg->n_features >= 1, NFM_ERR_INVALID, "n_features < 1");
  NFM_TRY(use_device(ctx));
  std::unique_ptr<nfm_model> m(new nfm_model());
  m->ctx = ctx;
  m->cfg = *cfg;
  m->k = cfg->n_components;
  m->d = cfg->n_features;
  if (cfg->kind == NFM_KIND_FM) {
    NFM_CHECK(cfg->degree >= 1, NFM_ERR_INVALID, "degree < 1.");
    NFM_CHECK(cfg->fit_lower >= 0 && cfg->fit_lower <= 2, NFM_ERR_INVALID, "bad fit_lower");
    // model/factorization_machine.nim:81-97
    m->n_aug = cfg->fit_lower == NFM_LOWER_AUGMENT ? (cfg->fit_linear ? cfg->degree - 2 : cfg->degree - 1) : 0;
    m->nb = cfg->degree == 1 ? 0 : (cfg->fit_lower == NFM_LOWER_EXPLICIT ? cfg->degree - 1 : 1);
    NFM_CHECK(m->n_aug >= 0, NFM_ERR_INVALID, "fit_lower=augment needs degree >= 2 (fit_linear) or >= 1");
  } else {
    NFM_CHECK(cfg->n_fields >= 1, NFM_ERR_INVALID, "n_fields < 1");
    m->n_aug = 0;
    m->nb = (int)cfg->n_fields;
  }
  m->da = m->d + m->n_aug;
  m->no = m->nb;
  m->kb = m->k;
  if (cfg->kind == NFM_KIND_FM && m->k > 128) {  // wide FM: kc blocks of kb <= 128 factors per order (ModelView::kc)
    m->kc = (m->k + 127) / 128;
    m->kb = (m->k + m->kc - 1) / m->kc;
    m->nb = m->no * m->kc;
  }
  m->L = lanes_for_k(m->kb);
  m->Kp = m->kb <= 128 ? 2 * m->L : ((m->kb + 63) / 64) * 64;  // (field-aware models keep wide rows: the one-sample-in-flight kernel takes them)
  if (m->kb > 128) m->L = 64;
  {
    const size_t bP = pad256(sizeof(double) * std::max<int64_t>(m->nP(), 2)), bw = pad256(sizeof(double) * m->d);
    NFM_TRY(m->arena.alloc(bP + bw + sizeof(double) * SC_COUNT));
    NFM_HIP_CHECK(hipMemsetAsync(m->arena.p, 0, m->arena.bytes, ctx->stream));
    char* base = m->arena.as<char>();
    m->P = {base, bP};
    m->w = {base + bP, bw};
    m->sc = {base + bP + bw, sizeof(double) * SC_COUNT};
  }
  NFM_TRY(m->lams.alloc(sizeof(double) * m->kc * m->Kp));
  double sc[SC_COUNT] = {1.0, 1.0, 0.0, 0, 0, 0, 0, 0};
  NFM_HIP_CHECK(hipMemcpyAsync(m->sc.p, sc, sizeof(sc), hipMemcpyHostToDevice, ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  m->uid = register_model(m.get());
  *out = m.release();
  return NFM_OK;
}

int32_t nfm_model_shape(const nfm_model* m, int32_t* n_blocks, int32_t* n_aug) {
  NFM_CHECK(m, NFM_ERR_INVALID, "null model");
  if (n_blocks) *n_blocks = m->no;  // the reference's orders (fields): the first extent of its P
  if (n_aug) *n_aug = m->n_aug;
  return NFM_OK;
}

int32_t nfm_model_set_params(nfm_model* m, const double* P, const double* w, double intercept, const double* lams) {
  NFM_CHECK(m && w, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(m->nb == 0 || P, NFM_ERR_INVALID, "null P");
  nfm_ctx* ctx = m->ctx;
  NFM_TRY(use_device(ctx));
  hipStream_t st = ctx->stream;
  const int64_t n_ref = m->n_ref();
  if (n_ref > 0) {
    DevBuf tmp;
    NFM_TRY(tmp.alloc(sizeof(double) * n_ref));
    NFM_HIP_CHECK(hipMemcpyAsync(tmp.p, P, sizeof(double) * n_ref, hipMemcpyHostToDevice, st));
    if (m->cfg.kind == NFM_KIND_FM)
      NFM_TRY(fm_params_to_device(m, tmp.as<double>(), m->P.as<double>()));
    else
      NFM_TRY(rows_to_device(m, tmp.as<double>(), m->P.as<double>(), 0.0));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  NFM_HIP_CHECK(hipMemcpyAsync(m->w.p, w, sizeof(double) * m->d, hipMemcpyHostToDevice, st));
  double sc[SC_COUNT] = {1.0, 1.0, intercept, 0, 0, 0, 0, 0};
  NFM_HIP_CHECK(hipMemcpyAsync(m->sc.p, sc, sizeof(sc), hipMemcpyHostToDevice, st));
  std::vector<double> lp((size_t)m->kc * m->Kp, 0.0);
  for (int s = 0; s < m->k; ++s) lp[(size_t)(s / m->kb) * m->Kp + s % m->kb] = lams ? lams[s] : 1.0;
  NFM_HIP_CHECK(hipMemcpyAsync(m->lams.p, lp.data(), sizeof(double) * lp.size(), hipMemcpyHostToDevice, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  m->initialized = true;
  return NFM_OK;
}

int32_t nfm_model_get_params(nfm_model* m, double* P, double* w, double* intercept) {
  NFM_CHECK(m, NFM_ERR_INVALID, "null model");
  NFM_CHECK(m->initialized, NFM_ERR_NOT_FITTED, "Factorization machines is not fitted.");
  nfm_ctx* ctx = m->ctx;
  NFM_TRY(use_device(ctx));
  hipStream_t st = ctx->stream;
  double sc[SC_COUNT];
  NFM_HIP_CHECK(hipMemcpyAsync(sc, m->sc.p, sizeof(sc), hipMemcpyDeviceToHost, st));
  const int64_t n_ref = m->n_ref();
  if (P && n_ref > 0) {
    DevBuf tmp;
    NFM_TRY(tmp.alloc(sizeof(double) * n_ref));
    if (m->cfg.kind == NFM_KIND_FM)
      NFM_TRY(fm_params_from_device(m, m->P.as<double>(), tmp.as<double>(), m->sc.as<double>() + SC_SCALE_P));
    else
      NFM_TRY(rows_from_device(m, m->P.as<double>(), tmp.as<double>(), m->sc.as<double>() + SC_SCALE_P));
    NFM_HIP_CHECK(hipMemcpyAsync(P, tmp.p, sizeof(double) * n_ref, hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  if (w) NFM_HIP_CHECK(hipMemcpyAsync(w, m->w.p, sizeof(double) * m->d, hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  if (w && sc[SC_SCALE_W] != 1.0)
    for (int64_t j = 0; j < m->d; ++j) w[j] *= sc[SC_SCALE_W];
  if (intercept) *intercept = sc[SC_INTERCEPT];
  return NFM_OK;
}

// training needs distinct column ids inside every row (plan.h); decisionFunction / score / metrics do not
static int check_trainable(const nfm_dataset* ds) {
  NFM_CHECK(ds->repeats == 0, NFM_ERR_UNSUPPORTED,
            "%lld repeated column ids inside rows (first in row %lld): the ids of one row must be distinct for training -- merge "
            "repeated entries (decisionFunction, predict and score take the dataset as it is)",
            (long long)ds->repeats, (long long)ds->repeat_row);
  return NFM_OK;
}

static int check_predict_shapes(nfm_model* m, nfm_dataset* ds) {
  NFM_CHECK(m && ds, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(m->ctx == ds->ctx, NFM_ERR_INVALID, "model and dataset belong to different contexts");
  NFM_CHECK(m->initialized, NFM_ERR_NOT_FITTED, "Factorization machines is not fitted.");
  NFM_CHECK(ds->v.d == m->d, NFM_ERR_INVALID, "Invalid nFeatures.");
  if (m->cfg.kind == NFM_KIND_FFM) {
    NFM_CHECK(ds->v.fields != nullptr, NFM_ERR_INVALID, "FFM needs a CSRFieldDataset (fields == NULL)");
    NFM_CHECK(ds->v.n_fields == m->nb, NFM_ERR_INVALID, "Invalid nFields.");
  }
  return NFM_OK;
}

int32_t nfm_decision_function_device(nfm_model* m, nfm_dataset* ds, double* out_dev) {
  NFM_TRY(check_predict_shapes(m, ds));
  NFM_CHECK(out_dev || ds->v.n == 0, NFM_ERR_INVALID, "null out");
  NFM_CHECK(m->cfg.kind == NFM_KIND_FFM || m->cfg.degree <= 6, NFM_ERR_UNSUPPORTED, "degree > 6 unsupported");
  NFM_TRY(use_device(m->ctx));
  return launch_predict(m->ctx, ds->v, m->view(), out_dev);
}

int32_t nfm_decision_function(nfm_model* m, nfm_dataset* ds, double* out) {
  NFM_TRY(check_predict_shapes(m, ds));
  NFM_CHECK(out || ds->v.n == 0, NFM_ERR_INVALID, "null out");
  NFM_TRY(use_device(m->ctx));
  DevBuf tmp;
  NFM_TRY(tmp.alloc(sizeof(double) * ds->v.n));
  NFM_TRY(nfm_decision_function_device(m, ds, tmp.as<double>()));
  NFM_HIP_CHECK(hipMemcpyAsync(out, tmp.p, sizeof(double) * ds->v.n, hipMemcpyDeviceToHost, m->ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return NFM_OK;
}

// score / metrics on the device: decisionFunction stays in HBM, only scalars come back
int32_t nfm_metrics(nfm_model* m, nfm_dataset* ds, double* rmse, double* accuracy, double* rocauc) {
  NFM_TRY(check_predict_shapes(m, ds));
  NFM_CHECK(ds->has_y, NFM_ERR_INVALID, "the dataset has no targets");
  NFM_TRY(use_device(m->ctx));
  DevBuf scores;
  NFM_TRY(scores.alloc(sizeof(double) * std::max<int64_t>(ds->v.n, 1)));
  NFM_TRY(nfm_decision_function_device(m, ds, scores.as<double>()));
  return launch_metrics(m->ctx, ds->v.n, scores.as<double>(), ds->v.y, rmse, accuracy, rocauc);
}

int32_t nfm_score(nfm_model* m, nfm_dataset* ds, double* out) {
  NFM_CHECK(m && out, NFM_ERR_INVALID, "null argument");
  // model/fm_base.nim:39-48: rmse for regression, accuracy (of signs) for classification
  if (m->cfg.task == NFM_TASK_REGRESSION) return nfm_metrics(m, ds, out, nullptr, nullptr);
  return nfm_metrics(m, ds, nullptr, out, nullptr);
}

int32_t nfm_model_sqnorms(nfm_model* m, double* P_sq, double* w_sq) {
  NFM_CHECK(m, NFM_ERR_INVALID, "null model");
  NFM_TRY(use_device(m->ctx));
  DevBuf out;
  NFM_TRY(out.alloc(sizeof(double) * 2));
  NFM_TRY(launch_sqnorms(m->ctx, m->view(), out.as<double>()));
  double h[2];
  NFM_HIP_CHECK(hipMemcpy(h, out.p, sizeof(h), hipMemcpyDeviceToHost));
  if (P_sq) *P_sq = h[0];
  if (w_sq) *w_sq = h[1];
  return NFM_OK;
}

int32_t nfm_model_device_buffers(nfm_model* m, double** P_dev, int64_t* n_P, double** w_dev, int64_t* n_w,
                                 double** scalars_dev, int64_t* n_scalars) {
  NFM_CHECK(m, NFM_ERR_INVALID, "null model");
  if (P_dev) *P_dev = m->P.as<double>();
  if (n_P) *n_P = m->nP();
  if (w_dev) *w_dev = m->w.as<double>();
  if (n_w) *n_w = m->d;
  if (scalars_dev) *scalars_dev = m->sc.as<double>();
  if (n_scalars) *n_scalars = SC_COUNT;
  return NFM_OK;
}

int32_t nfm_model_destroy(nfm_model* m) {
  if (!m) return NFM_OK;
  unregister_model(m->uid);
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  delete m;
  return NFM_OK;
}

// ------------------------------------------------------------------ optimizers
static int check_common(nfm_model* m, int loss, int mode, int64_t batch) {
  NFM_CHECK(m, NFM_ERR_INVALID, "null model");
  NFM_CHECK(loss >= 0 && loss <= 3, NFM_ERR_INVALID, "bad loss id");
  NFM_CHECK(mode == NFM_MODE_SEQUENTIAL || mode == NFM_MODE_MINIBATCH, NFM_ERR_INVALID, "bad mode");
  NFM_CHECK(mode == NFM_MODE_SEQUENTIAL || batch >= 1, NFM_ERR_INVALID, "batch must be >= 1");
  return NFM_OK;
}

int32_t nfm_sgd_create(nfm_model* m, const nfm_sgd_cfg* c, nfm_opt** out) {
  NFM_CHECK(c && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(check_common(m, c->loss, c->mode, c->batch));
  NFM_CHECK(c->scheduling >= 0 && c->scheduling <= 3, NFM_ERR_INVALID, "bad scheduling id");
  std::unique_ptr<nfm_opt> o(new nfm_opt());
  o->ctx = m->ctx; o->m = m; o->m_uid = m->uid; o->kind = OPT_SGD; o->mode = c->mode; o->batch = c->mode == NFM_MODE_MINIBATCH ? c->batch : 1; o->it = 1;
  o->o.eta0 = c->eta0; o->o.alpha0 = c->alpha0; o->o.alpha = c->alpha; o->o.beta = c->beta; o->o.power = c->power;
  o->o.eps = 0.0; o->o.loss_param = c->loss_param; o->o.loss = c->loss; o->o.sched = c->scheduling; o->o.track_viol = 1;
  o->o.touch_cap = 1.0;
  NFM_TRY(use_device(m->ctx));
  NFM_TRY(o->out2.alloc(sizeof(double) * 2));
  *out = o.release();
  return NFM_OK;
}

int32_t nfm_adagrad_create(nfm_model* m, const nfm_adagrad_cfg* c, nfm_opt** out) {
  NFM_CHECK(c && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(check_common(m, c->loss, c->mode, c->batch));
  std::unique_ptr<nfm_opt> o(new nfm_opt());
  o->ctx = m->ctx; o->m = m; o->m_uid = m->uid; o->kind = OPT_ADAGRAD; o->mode = c->mode; o->batch = c->mode == NFM_MODE_MINIBATCH ? c->batch : 1; o->it = 1;
  o->o.eta0 = c->eta0; o->o.alpha0 = c->alpha0; o->o.alpha = c->alpha; o->o.beta = c->beta; o->o.power = 1.0;
  o->o.eps = c->eps; o->o.loss_param = c->loss_param; o->o.loss = c->loss; o->o.sched = 0; o->o.track_viol = c->track_viol;
  o->o.touch_cap = 1.0;
  o->o.ada_cross = 0.0;
  NFM_TRY(use_device(m->ctx));
  NFM_TRY(o->out2.alloc(sizeof(double) * 2));
  {
    const size_t bP = pad256(sizeof(double) * std::max<int64_t>(m->nP(), 2)), bw = pad256(sizeof(double) * m->d);
    NFM_TRY(o->state_arena.alloc(2 * bP + 2 * bw + sizeof(double) * 2));
    NFM_HIP_CHECK(hipMemsetAsync(o->state_arena.p, 0, o->state_arena.bytes, m->ctx->stream));
    char* base = o->state_arena.as<char>();
    o->G = {base, bP};
    o->N = {base + bP, bP};
    o->Gw = {base + 2 * bP, bw};
    o->Nw = {base + 2 * bP + bw, bw};
    o->gsc = {base + 2 * bP + 2 * bw, sizeof(double) * 2};
  }
  o->o.G = o->G.as<double>(); o->o.N = o->N.as<double>(); o->o.Gw = o->Gw.as<double>(); o->o.Nw = o->Nw.as<double>();
  o->o.gsc = o->gsc.as<double>();
  *out = o.release();
  return NFM_OK;
}

int32_t nfm_mbpsgd_create(nfm_model* m, const nfm_mbpsgd_cfg* c, nfm_opt** out) {
  NFM_CHECK(c && out, NFM_ERR_INVALID, "null argument");
  NFM_TRY(check_common(m, c->loss, NFM_MODE_MINIBATCH, c->batch));
  NFM_CHECK(c->scheduling >= 0 && c->scheduling <= 3, NFM_ERR_INVALID, "bad scheduling id");
  NFM_CHECK(c->reg >= NFM_REG_L1 && c->reg <= NFM_REG_SQUAREDL21, NFM_ERR_INVALID, "bad regularizer id");
  NFM_CHECK(m->cfg.kind == NFM_KIND_FM, NFM_ERR_UNSUPPORTED, "MBPSGD fits a FactorizationMachine (minibatch_psgd.nim:125-126)");
  NFM_CHECK(m->kc == 1, NFM_ERR_UNSUPPORTED, "MBPSGD supports n_components <= 128 (the matrix prox needs a feature's factors in one row)");
  if (c->reg == NFM_REG_SQUAREDL12)  // squaredl12.nim:103-105
    NFM_CHECK(m->cfg.degree == 2, NFM_ERR_INVALID, "SquaredL12 supports only degree=2.");
  if (c->reg == NFM_REG_SQUAREDL21) {  // squaredl21.nim:27-28
    NFM_CHECK(m->cfg.degree == 2, NFM_ERR_INVALID, "SquaredL21 supports only degree=2.");
    NFM_CHECK(!c->reg_transpose, NFM_ERR_UNSUPPORTED, "SquaredL21 with transpose=true is not supported");
  }
  std::unique_ptr<nfm_opt> o(new nfm_opt());
  o->ctx = m->ctx; o->m = m; o->m_uid = m->uid; o->kind = OPT_PSGD; o->mode = NFM_MODE_MINIBATCH; o->batch = c->batch; o->it = 1;
  o->o.eta0 = c->eta0; o->o.alpha0 = c->alpha0; o->o.alpha = c->alpha; o->o.beta = c->beta; o->o.power = c->power;
  o->o.eps = 0.0; o->o.loss_param = c->loss_param; o->o.loss = c->loss; o->o.sched = c->scheduling; o->o.track_viol = 0;
  o->o.touch_cap = 1.0;
  o->o.gamma = c->gamma; o->o.bsize = (double)c->batch; o->o.reg = c->reg; o->o.reg_transpose = c->reg_transpose ? 1 : 0;
  NFM_TRY(use_device(m->ctx));
  NFM_TRY(o->out2.alloc(sizeof(double) * 2));
  *out = o.release();
  return NFM_OK;
}

int32_t nfm_opt_set_it(nfm_opt* o, int64_t it) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  // newMBPSGD starts at it = 0 and a warm-started fit keeps it (minibatch_psgd.nim:63,153-154)
  NFM_CHECK(it >= (o->kind == OPT_PSGD ? 0 : 1), NFM_ERR_INVALID, "it must be >= 1");
  o->it = it;
  // a new fit starts here: a plan prepared for "the next epoch" of an earlier fit (announced or device-drawn order) is
  // not carried into it
  o->next_plan_ready = false;
  o->announced = nullptr;
  return NFM_OK;
}
int32_t nfm_opt_get_it(nfm_opt* o, int64_t* it) {
  NFM_CHECK(o && it, NFM_ERR_INVALID, "null argument");
  *it = o->it;
  return NFM_OK;
}

// adagrad.nim:52-55: g_sum = 0, g_norm = eps when it == 1
static int adagrad_reset_state(nfm_opt* o) {
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  nfm_ctx* ctx = m->ctx;
  NFM_HIP_CHECK(hipMemsetAsync(o->G.p, 0, o->G.bytes, ctx->stream));
  NFM_HIP_CHECK(hipMemsetAsync(o->Gw.p, 0, o->Gw.bytes, ctx->stream));
  NFM_TRY(launch_fill(ctx, o->N.as<double>(), std::max<int64_t>(m->nP(), 2), o->o.eps));
  NFM_TRY(launch_fill(ctx, o->Nw.as<double>(), m->d, o->o.eps));
  const double g[2] = {0.0, o->o.eps};
  NFM_HIP_CHECK(hipMemcpyAsync(o->gsc.p, g, sizeof(g), hipMemcpyHostToDevice, ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  o->state_ready = true;
  return NFM_OK;
}

int32_t nfm_opt_get_state(nfm_opt* o, double* gsum_P, double* gnorm_P, double* gsum_w, double* gnorm_w, double* gsum_b,
                          double* gnorm_b) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(o->kind == OPT_ADAGRAD, NFM_ERR_INVALID, "only AdaGrad carries state");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  nfm_ctx* ctx = m->ctx;
  NFM_TRY(use_device(ctx));
  if (!o->state_ready) NFM_TRY(adagrad_reset_state(o));
  const int64_t n_ref = m->n_ref();  // (the state shares the parameters' device layout)
  DevBuf tmp;
  NFM_TRY(tmp.alloc(sizeof(double) * std::max<int64_t>(n_ref, 1)));
  for (int which = 0; which < 2; ++which) {
    double* dst = which ? gnorm_P : gsum_P;
    if (!dst || n_ref == 0) continue;
    NFM_TRY(rows_from_device(m, which ? o->N.as<double>() : o->G.as<double>(), tmp.as<double>(), nullptr));
    NFM_HIP_CHECK(hipMemcpyAsync(dst, tmp.p, sizeof(double) * n_ref, hipMemcpyDeviceToHost, ctx->stream));
    NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  if (gsum_w) NFM_HIP_CHECK(hipMemcpyAsync(gsum_w, o->Gw.p, sizeof(double) * m->d, hipMemcpyDeviceToHost, ctx->stream));
  if (gnorm_w) NFM_HIP_CHECK(hipMemcpyAsync(gnorm_w, o->Nw.p, sizeof(double) * m->d, hipMemcpyDeviceToHost, ctx->stream));
  double g[2];
  NFM_HIP_CHECK(hipMemcpyAsync(g, o->gsc.p, sizeof(g), hipMemcpyDeviceToHost, ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (gsum_b) *gsum_b = g[0];
  if (gnorm_b) *gnorm_b = g[1];
  return NFM_OK;
}

int32_t nfm_opt_set_state(nfm_opt* o, const double* gsum_P, const double* gnorm_P, const double* gsum_w,
                          const double* gnorm_w, double gsum_b, double gnorm_b) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(o->kind == OPT_ADAGRAD, NFM_ERR_INVALID, "only AdaGrad carries state");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  nfm_ctx* ctx = m->ctx;
  NFM_TRY(use_device(ctx));
  const int64_t n_ref = m->n_ref();  // (the state shares the parameters' device layout)
  NFM_CHECK(n_ref == 0 || (gsum_P && gnorm_P), NFM_ERR_INVALID, "null state");
  NFM_CHECK(gsum_w && gnorm_w, NFM_ERR_INVALID, "null state");
  DevBuf tmp;
  NFM_TRY(tmp.alloc(sizeof(double) * std::max<int64_t>(n_ref, 1)));
  if (n_ref > 0) {
    NFM_HIP_CHECK(hipMemcpyAsync(tmp.p, gsum_P, sizeof(double) * n_ref, hipMemcpyHostToDevice, ctx->stream));
    NFM_TRY(rows_to_device(m, tmp.as<double>(), o->G.as<double>(), 0.0));
    NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    NFM_HIP_CHECK(hipMemcpyAsync(tmp.p, gnorm_P, sizeof(double) * n_ref, hipMemcpyHostToDevice, ctx->stream));
    NFM_TRY(rows_to_device(m, tmp.as<double>(), o->N.as<double>(), o->o.eps));
    NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  NFM_HIP_CHECK(hipMemcpyAsync(o->Gw.p, gsum_w, sizeof(double) * m->d, hipMemcpyHostToDevice, ctx->stream));
  NFM_HIP_CHECK(hipMemcpyAsync(o->Nw.p, gnorm_w, sizeof(double) * m->d, hipMemcpyHostToDevice, ctx->stream));
  const double g[2] = {gsum_b, gnorm_b};
  NFM_HIP_CHECK(hipMemcpyAsync(o->gsc.p, g, sizeof(g), hipMemcpyHostToDevice, ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  o->state_ready = true;
  return NFM_OK;
}

static int ensure_unit_scale(nfm_model* m) {
  double sc[SC_COUNT];
  NFM_HIP_CHECK(hipMemcpyAsync(sc, m->sc.p, sizeof(sc), hipMemcpyDeviceToHost, m->ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  if (sc[SC_SCALE_P] != 1.0 || sc[SC_SCALE_W] != 1.0) {
    ModelView v = m->view();
    v.fit_linear = 1;  // w carries its scale regardless of who trained it
    NFM_TRY(launch_rescale(m->ctx, v));
  }
  return NFM_OK;
}

// what one nfm_opt_epoch call of an optimizer with a group attached exchanges (dp.h)
static int dp_epoch_setup(nfm_opt* o, nfm_model* m, const ModelView& M, DpEpoch* out) {
  NFM_CHECK(o->kind != OPT_PSGD, NFM_ERR_UNSUPPORTED, "MBPSGD has no data-parallel mode");
  DpEpoch& de = *out;
  de.dp = o->dp;
  de.opt_kind = o->kind;
  de.sync_period = o->dp_sync_period;
  de.overlap = o->dp_overlap;
  // SGD: the ranks' increments averaged unless NFM_DP_SUM; AdaGrad: its state increments summed unless NFM_DP_STATE_MEAN
  // (NFM_DP_AUTO, the default no host has to know about: what DESIGN.md section 6 measured as stable at every period --
  // AdaGrad's SUMMED state over-shoots as soon as the ranks run more than one mini-batch between exchanges)
  // AdaGrad, NFM_DP_AUTO (round 5): the summed state when the ranks exchange after every mini-batch (synchronous data-parallel
  // AdaGrad), NFM_DP_STATE_CROSS otherwise -- one rank's progress per epoch at 2 / 4 / 8 ranks for every period, where the averaged
  // state keeps 0.75 of it at 8 ranks and the plain sum diverges (profiles/r05g_dp_convergence.txt)
  const int ada_rule = (o->kind == OPT_ADAGRAD && o->dp_combine == NFM_DP_AUTO) ? (o->dp_sync_period == 1 ? NFM_DP_SUM : NFM_DP_STATE_CROSS) : o->dp_combine;
  const bool averaged = o->kind == OPT_SGD ? o->dp_combine != NFM_DP_SUM : ada_rule == NFM_DP_STATE_MEAN;
  de.combine_w = averaged ? 1.0 / (double)o->dp->t->world : 1.0;
  if (o->kind == OPT_ADAGRAD && o->dp_combine == NFM_DP_STATE_RSQRT) de.combine_w = 1.0 / sqrt((double)o->dp->t->world);  // (SGD: the mean)
  if (o->kind == OPT_SGD) {
    de.arena = m->arena.as<double>();
    de.n = (int64_t)((m->sc.as<char>() - m->arena.as<char>()) / sizeof(double)) + SC_COUNT;
    de.skip_lo = (int64_t)((m->sc.as<char>() - m->arena.as<char>()) / sizeof(double)) + SC_SCALE_P;
    de.skip_hi = de.skip_lo + 2;  // {scale_P, scale_w}
    de.seg_w = (int64_t)((reinterpret_cast<char*>(M.w) - m->arena.as<char>()) / sizeof(double));
    de.seg_sc = (int64_t)((m->sc.as<char>() - m->arena.as<char>()) / sizeof(double));
  } else {
    de.arena = o->state_arena.as<double>();
    de.n = (int64_t)((o->gsc.as<char>() - o->state_arena.as<char>()) / sizeof(double)) + 2;
    if (ada_rule == NFM_DP_STATE_CROSS) {  // (g_sum, g_norm) pairs of P, w and the intercept (dp.h)
      auto off = [&](const Span& sp) { return (int64_t)((sp.as<char>() - o->state_arena.as<char>()) / sizeof(double)); };
      de.combine_w = 1.0;
      static const double gamma = getenv("NFM_DP_CROSS_GAMMA") ? atof(getenv("NFM_DP_CROSS_GAMMA")) : 0.1;
      de.cross_gamma = gamma;
      de.n_pairs = 3;
      de.pair[0][0] = off(o->G); de.pair[0][1] = off(o->N); de.pair[0][2] = std::max<int64_t>(m->nP(), 0);
      de.pair[1][0] = off(o->Gw); de.pair[1][1] = off(o->Nw); de.pair[1][2] = m->d;
      de.pair[2][0] = off(o->gsc); de.pair[2][1] = off(o->gsc) + 1; de.pair[2][2] = 1;
    }
  }
  return NFM_OK;
}

static int32_t opt_epoch_range(nfm_opt* o, nfm_dataset* ds, const int64_t* perm, int64_t begin, int64_t end, double* loss_sum,
                               double* viol_sum);

// the entries one epoch call may hold (the plan's touch tables and the window's dependency table index them with 32 bits)
static int64_t max_epoch_nnz() {
  int64_t cap = (int64_t)2147483647;
#ifdef NFM_TEST_HOOKS  // (libnimfm_hip_testhooks.so only: lets a test cut a small epoch into pieces)
  if (const char* env = getenv("NFM_TEST_MAX_EPOCH_NNZ")) cap = atoll(env);
#endif
  return cap;
}

int32_t nfm_opt_epoch(nfm_opt* o, nfm_dataset* ds, const int64_t* perm, int64_t begin, int64_t end, double* loss_sum,
                      double* viol_sum) {
  NFM_CHECK(o && ds, NFM_ERR_INVALID, "null argument");
  // A range of more than 2^31 - 1 entries (288 GB hold datasets several times that) is walked as consecutive pieces: an epoch
  // call IS the sequence of its sub-range calls (sequential mode: any cut; mini-batch mode: cuts at mini-batch boundaries --
  // tests/test_gpu_fullsize.py holds one call against two), so the results are those of the one call.  The bound on a piece's
  // entries is samples x the longest row.  Not with a data-parallel group (the exchange points are laid out per call), and a
  // device-drawn order (nfm_opt_set_shuffle) then shuffles inside each piece.
  const int64_t row_max = std::max<int64_t>((int64_t)ds->max_row + 8, 1), cap = max_epoch_nnz();
  const int64_t ns_all = end - begin;
  // (an order over distinct samples holds no more entries than the dataset, however long its longest row is: one outlier row
  // must not cut every epoch of a small dataset into pieces.  MBPSGD's index stream may wrap, so only the product bounds it.)
  bool may_exceed = ns_all > cap / row_max;
  if (may_exceed && o->kind != OPT_PSGD && ns_all <= ds->v.n && ds->v.nnz + 8 * ns_all <= cap) may_exceed = false;
  if (begin >= 0 && ns_all > 0 && !o->dp && may_exceed && (perm || end <= ds->v.n)) {
    const int64_t B = o->mode == NFM_MODE_MINIBATCH ? std::max<int64_t>(o->batch, 1) : 1;
    int64_t piece = cap / row_max / B * B;  // whole mini-batches
    NFM_CHECK(piece >= B, NFM_ERR_UNSUPPORTED, "one mini-batch of %lld samples may hold more than 2^31-1 entries (longest row %lld)",
              (long long)B, (long long)row_max);
    double ls = 0.0, vs = 0.0;
    for (int64_t p0 = begin; p0 < end;) {
      // AdaGrad's very first step is a mini-batch of its own (adagrad.nim:171): the piece that holds it is one sample longer
      const int64_t lead = (o->mode == NFM_MODE_MINIBATCH && o->kind == OPT_ADAGRAD && o->it == 1) ? 1 : 0;
      const int64_t p1 = std::min(end, p0 + piece - (lead ? B - 1 : 0));
      double l1 = 0.0, v1 = 0.0;
      NFM_TRY(opt_epoch_range(o, ds, perm, p0, p1, &l1, &v1));
      ls += l1;
      vs += v1;
      p0 = p1;
    }
    if (loss_sum) *loss_sum = ls;
    if (viol_sum) *viol_sum = vs;
    return NFM_OK;
  }
  return opt_epoch_range(o, ds, perm, begin, end, loss_sum, viol_sum);
}

static int32_t opt_epoch_range(nfm_opt* o, nfm_dataset* ds, const int64_t* perm, int64_t begin, int64_t end, double* loss_sum,
                               double* viol_sum) {
  NFM_CHECK(o && ds, NFM_ERR_INVALID, "null argument");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  nfm_ctx* ctx = m->ctx;
  NFM_CHECK(ds->ctx == ctx, NFM_ERR_INVALID, "optimizer and dataset belong to different contexts");
  NFM_CHECK(!o->dp || dp_is_live(o->dp, o->dp_uid), NFM_ERR_INVALID,
            "the optimizer's data-parallel group was destroyed; detach it (nfm_opt_set_dp(o, NULL, 0, 0)) or attach a new one");
  NFM_TRY(check_predict_shapes(m, ds));
  NFM_CHECK(ds->has_y, NFM_ERR_INVALID, "dataset has no targets");
  NFM_TRY(check_trainable(ds));
  // MBPSGD consumes a stream of sample indices that may wrap past the end of the data (minibatch_psgd.nim:104-108)
  NFM_CHECK(begin >= 0 && begin <= end && (end <= ds->v.n || (o->kind == OPT_PSGD && perm)), NFM_ERR_INVALID,
            "bad sample range [%lld,%lld)", (long long)begin, (long long)end);
  if (o->kind == OPT_PSGD)
    NFM_CHECK((end - begin) % o->batch == 0, NFM_ERR_INVALID, "MBPSGD: %lld samples are not a whole number of mini-batches of %lld",
              (long long)(end - begin), (long long)o->batch);
  if (perm && o->mode == NFM_MODE_SEQUENTIAL)  // mini-batch mode checks the ids on the device (plan.hip)
    for (int64_t p = begin; p < end; ++p)
      NFM_CHECK(perm[p] >= 0 && perm[p] < ds->v.n, NFM_ERR_INVALID, "perm[%lld] = %lld out of range", (long long)p, (long long)perm[p]);
  NFM_TRY(use_device(ctx));
  hipStream_t st = ctx->stream;
  if (o->kind == OPT_ADAGRAD) {
    if (o->it == 1 || !o->state_ready) NFM_TRY(adagrad_reset_state(o));
    NFM_TRY(ensure_unit_scale(m));
  } else if (o->kind == OPT_PSGD) {
    NFM_TRY(ensure_unit_scale(m));
  }
  double out2[2] = {0.0, 0.0};
  const int64_t ns = end - begin;
  if (ns > 0) {
    const ModelView M = m->view();
    if (o->mode == NFM_MODE_SEQUENTIAL) {
      NFM_CHECK(!o->dp, NFM_ERR_UNSUPPORTED, "the data-parallel exchange needs NFM_MODE_MINIBATCH");
      const int64_t* perm_dev = nullptr;
      if (perm) {
        NFM_TRY(o->perm_dev.ensure(sizeof(int64_t) * ns));
        NFM_HIP_CHECK(hipMemcpyAsync(o->perm_dev.p, perm + begin, sizeof(int64_t) * ns, hipMemcpyHostToDevice, st));
        perm_dev = o->perm_dev.as<int64_t>() - begin;  // indexed by absolute position
      }
      // The window kernel's workgroups wait for each other; should one of those waits time out (CUs held by another tenant)
      // the call's samples are partly applied.  So the call starts from a snapshot of everything it may write -- the model's
      // arena, AdaGrad's state arena: one device-to-device copy, 0.2 ms for the headline's 520 MB -- and an aborted call is
      // put back and run by the one-workgroup kernel.  The window is used when the snapshot costs less than a quarter of it.
      const size_t snap_bytes = m->arena.bytes + (o->kind == OPT_ADAGRAD ? o->state_arena.bytes : 0);
      const char* win_env = getenv("NFM_SEQ_WIN");
      const bool snap_pays = (win_env && atoi(win_env) == 2) || (double)ns * 0.8e-6 * 0.25 >= (double)snap_bytes / 2.0e12;
      bool windowed = false;
      const bool win_trusted = !o->seqwin || o->seqwin->fallbacks < 2;  // two aborted launches: CUs are being held -- no more 1 s waits
      if (!win_trusted && o->seqwin) o->seqwin->snap.release();  // (no more window launches from this optimizer: its snapshot goes back)
      bool have_snap = false;
      const ModelView Mw = seq_window_view(M);  // (65 ... 128 factors: the table read as two blocks of 64, seqwin.hip)
      if (snap_pays && win_trusted && seq_window_supported(Mw, ds->max_row + m->n_aug, ns, ds->v.nnz, ctx->n_cu, o->kind == OPT_ADAGRAD)) {
        if (!o->seqwin) o->seqwin.reset(new SeqWin());
        // A model (+ AdaGrad state) beyond half of the free memory has no room for its snapshot: that fit runs in the
        // one-workgroup kernel, which needs none -- it must not fail for want of a safety copy.
        have_snap = o->seqwin->snap.ensure(snap_bytes) == NFM_OK;
#ifdef NFM_TEST_HOOKS  // (libnimfm_hip_testhooks.so only)
        if (getenv("NFM_TEST_NO_SNAPSHOT") && atoi(getenv("NFM_TEST_NO_SNAPSHOT")) != 0) {
          o->seqwin->snap.release();
          have_snap = false;
        }
#endif
        if (!have_snap) {
          (void)hipGetLastError();  // (the failed allocation's sticky error)
          ctx->timing.acc["seq_window_no_snapshot"].launches += 1;
        }
      }
      if (have_snap) {
        SeqWin* sw = o->seqwin.get();
        NFM_HIP_CHECK(hipMemcpyAsync(sw->snap.p, m->arena.p, m->arena.bytes, hipMemcpyDeviceToDevice, st));
        if (o->kind == OPT_ADAGRAD)
          NFM_HIP_CHECK(hipMemcpyAsync(sw->snap.as<char>() + m->arena.bytes, o->state_arena.p, o->state_arena.bytes, hipMemcpyDeviceToDevice, st));
        const int rc = launch_sequential_window(ctx, o->kind, ds->v, Mw, o->o, perm_dev, begin, end, o->it, ds->max_row + m->n_aug,
                                                o->out2.as<double>(), sw, (ds->uid << 20) ^ ds->serial, perm != nullptr);
        if (rc == NFM_WIN_FALLBACK) {
          NFM_HIP_CHECK(hipMemcpyAsync(m->arena.p, sw->snap.p, m->arena.bytes, hipMemcpyDeviceToDevice, st));
          if (o->kind == OPT_ADAGRAD)
            NFM_HIP_CHECK(hipMemcpyAsync(o->state_arena.p, sw->snap.as<char>() + m->arena.bytes, o->state_arena.bytes, hipMemcpyDeviceToDevice, st));
          ++sw->fallbacks;
          sw->clean_calls = 0;
          ctx->timing.acc["seq_window_fallback"].launches += 1;  // counted (timing on or off) where tests and bench.py see it: nfm_ctx_timing_get
        } else {
          NFM_TRY(rc);
          windowed = true;
          if (sw->fallbacks > 0 && ++sw->clean_calls >= 16) {  // (a tenant that held CUs once is not held against the optimizer for ever)
            --sw->fallbacks;
            sw->clean_calls = 0;
          }
        }
      }
      if (!windowed)
        NFM_TRY(launch_sequential(ctx, o->kind, ds->v, seq_row_view(M), o->o, perm_dev, begin, end, o->it, ds->max_row + m->n_aug,
                                  o->out2.as<double>()));
      NFM_HIP_CHECK(hipMemcpyAsync(out2, o->out2.p, sizeof(out2), hipMemcpyDeviceToHost, st));
      NFM_HIP_CHECK(hipStreamSynchronize(st));
    } else {
      const bool first_singleton = o->kind == OPT_ADAGRAD && o->it == 1;
      const bool want_tq = m->cfg.kind == NFM_KIND_FFM;
      const bool reuse = o->plan && !perm && !o->plan->has_perm && o->plan->ds_uid == ds->uid && o->plan->ds_nnz == ds->v.nnz && o->plan->begin == begin &&
                         o->plan->end == end && o->plan->batch == o->batch && o->plan->first_singleton == first_singleton &&
                         o->plan->n_aug == m->n_aug;
      // Features touched once per batch are updated by the row phase itself ("singles"); worth the
      // second visit of the row only when they are a sizeable share of the touches.  With a touch
      // rate lambda = batch * nnz_per_row / d per feature that share is about exp(-lambda).
      const double lambda = (double)o->batch * ((double)ds->v.nnz / (double)std::max<int64_t>(ds->v.n, 1)) / (double)m->d;
      bool use_singles = o->kind != OPT_PSGD && m->cfg.kind == NFM_KIND_FM && m->cfg.degree == 2 && m->nb == 1 && (o->batch == 1 || lambda <= 1.4);
      if (const char* env = getenv("NFM_SINGLES")) use_singles = use_singles && atoi(env) != 0;  // tuning override
      bool sort_by_count = m->Kp * (int)sizeof(double) >= 128;  // rows of at least one 128-byte line (plan.hip)
      if (const char* env = getenv("NFM_SORT_BY_COUNT")) sort_by_count = atoi(env) != 0;  // tuning override
      const bool dev_shuffle = !perm && o->shuffle_seed >= 0 && o->kind != OPT_PSGD;
      auto plan_matches = [&](const Plan& PL, bool fs) {
        return PL.ds_uid == ds->uid && PL.ds_nnz == ds->v.nnz && PL.begin == begin && PL.end == end && PL.batch == o->batch &&
               PL.first_singleton == fs && PL.n_aug == m->n_aug && PL.use_singles == use_singles;
      };
      if (dev_shuffle) {
        // the order of this epoch is a function of (seed, shuffle_epoch); its plan may have been built beside the
        // previous epoch
        if (o->next_plan_ready && o->next_plan && !o->next_plan_perm && o->next_plan_epoch == o->shuffle_epoch &&
            plan_matches(*o->next_plan, first_singleton)) {
          std::swap(o->plan, o->next_plan);
          o->perm_gen.take(o->perm_next);
        } else {
          if (!o->plan) o->plan.reset(new Plan());
          TimedLaunch tl(ctx, "plan_build");
          NFM_TRY(gen_permutation(ctx, st, o->shuffle_seed, o->shuffle_epoch, begin, ns, &o->perm_gen));
          const FeistelKey fk = feistel_key(o->shuffle_seed, o->shuffle_epoch, begin, ns);  // (the order as a function: plan.h)
          NFM_TRY(plan_build(ctx, ds->v, m->n_aug, nullptr, begin, end, o->batch, first_singleton, want_tq, use_singles, sort_by_count,
                             o->plan.get(), st, o->perm_gen.as<int64_t>(), &ds->csc, &fk));
          o->plan->ds_uid = ds->uid;
          o->plan->ds_nnz = ds->v.nnz;
        }
        o->next_plan_ready = false;
      } else if (perm && o->next_plan_ready && o->next_plan && o->next_plan_perm == perm && plan_matches(*o->next_plan, first_singleton) &&
                 [&] {  // the array the plan was built from, unchanged as promised (a few probes)
                   for (int q = 0; q < 64; ++q)
                     if (perm[begin + (ns - 1) * q / 63] != o->next_probe[q]) return false;
                   return true;
                 }()) {
        std::swap(o->plan, o->next_plan);
        o->next_plan_ready = false;
      } else if (!reuse) {
        if (!o->plan) o->plan.reset(new Plan());
        TimedLaunch tl(ctx, "plan_build");
        NFM_TRY(plan_build(ctx, ds->v, m->n_aug, perm, begin, end, o->batch, first_singleton, want_tq, use_singles,
                           sort_by_count, o->plan.get(), nullptr, nullptr, perm ? &ds->csc : nullptr));
        o->plan->ds_uid = ds->uid;
        o->plan->ds_nnz = ds->v.nnz;
        o->next_plan_ready = false;
      }
      // an announced permutation for the next call over the same range: its plan is built beside this epoch
      const int64_t* ann = (!dev_shuffle && o->announced && o->announced_begin == begin && o->announced_end == end && o->kind != OPT_PSGD)
                               ? o->announced : nullptr;
      o->announced = nullptr;
      DpEpoch de;
      if (o->dp) {
        // the ranks of the group run this call together on their own shards (equal step counters at its start)
        NFM_TRY(dp_epoch_setup(o, m, M, &de));
        // leading mini-batches of the regular length (everything but a shorter tail) look alike on every rank
        const Plan& PL = *o->plan;
        int64_t regular = PL.n_batches;
        if (PL.n_batches > 0 && PL.bat_pos[PL.n_batches] - PL.bat_pos[PL.n_batches - 1] < o->batch &&
            !(PL.first_singleton && PL.n_batches == 1))
          regular = PL.n_batches - 1;
        NFM_TRY(dp_epoch_begin(de, regular, PL.n_batches));
        o->W.after_batch = [&de](int64_t b) { return dp_after_batch(de, b); };
        o->W.is_sync = nullptr;  // (no graphs for the threads of a local group: dp.h)
        if (!o->dp->t->in_process)
          o->W.is_sync = [&de](int64_t b) { return de.sync_period > 0 && (b + 1) % de.sync_period == 0 && (b + 1) / de.sync_period <= de.n_sync; };
        o->W.seg_key_now = (de.sync_period << 32) + de.n_sync;
      }
      // the hook refers to this frame: whatever way the call leaves, it is taken off again; a call that fails between
      // a sync point and its fold-in leaves no exchange marked pending (dp_epoch_begin would refuse every later epoch)
      struct DpGuard {
        nfm_opt* o;
        bool ok = false;
        ~DpGuard() {
          o->W.after_batch = nullptr;
          o->W.is_sync = nullptr;
          if (!ok && o->dp) {
            // the all-reduce of the last sync point may still be running on the group's stream (it reads dp->snap and
            // writes dp->recv): the next epoch must not overwrite them under it
            if (o->dp->pending && o->dp->comm) (void)hipStreamSynchronize(o->dp->comm);
            o->dp->pending = false;
          }
        }
      } dp_guard{o};
      int rc_epoch;
      // device shuffle: the epoch is only ENQUEUED here; the next epoch's order and plan are then built on a second
      // stream (its host-side waits block on that stream only) while this epoch runs
      static const bool prefetch_on = !(getenv("NFM_PLAN_PREFETCH") && atoi(getenv("NFM_PLAN_PREFETCH")) == 0);  // 0: plans built in line (profiling)
      const bool prefetch = (dev_shuffle || ann != nullptr) && !ctx->timing.enabled && prefetch_on;
      double* out2_dst = out2;
      if (prefetch) {
        if (!o->out2_pinned) NFM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&o->out2_pinned), sizeof(double) * 2, hipHostMallocDefault));
        if (!o->plan_stream) NFM_HIP_CHECK(hipStreamCreateWithFlags(&o->plan_stream, hipStreamNonBlocking));
        out2_dst = o->out2_pinned;
      }
      if (m->cfg.kind == NFM_KIND_FM)
        rc_epoch = mb_fm_epoch(ctx, o->kind, ds->v, M, o->o, *o->plan, o->W, o->it, out2_dst, (ds->uid << 20) ^ ds->serial, prefetch);
      else
        rc_epoch = mb_ffm_epoch(ctx, o->kind, ds->v, M, o->o, *o->plan, o->W, o->it, out2_dst, prefetch);
      o->W.after_batch = nullptr;
      o->W.is_sync = nullptr;
      if (prefetch) {
        int rc_next = NFM_OK;
        if (rc_epoch == NFM_OK) {
          if (!o->next_plan) o->next_plan.reset(new Plan());
          if (ann) {
            rc_next = plan_build(ctx, ds->v, m->n_aug, ann, begin, end, o->batch, /*first_singleton=*/false, want_tq, use_singles,
                                 sort_by_count, o->next_plan.get(), o->plan_stream, nullptr, &ds->csc);
            if (rc_next == NFM_OK) {
              o->next_plan_perm = ann;
              for (int q = 0; q < 64; ++q) o->next_probe[q] = ann[begin + (ns - 1) * q / 63];
            }
          } else {
            rc_next = gen_permutation(ctx, o->plan_stream, o->shuffle_seed, o->shuffle_epoch + 1, begin, ns, &o->perm_next);
            const FeistelKey fk = feistel_key(o->shuffle_seed, o->shuffle_epoch + 1, begin, ns);
            if (rc_next == NFM_OK)
              rc_next = plan_build(ctx, ds->v, m->n_aug, nullptr, begin, end, o->batch, /*first_singleton=*/false, want_tq, use_singles,
                                   sort_by_count, o->next_plan.get(), o->plan_stream, o->perm_next.as<int64_t>(), &ds->csc, &fk);
            if (rc_next == NFM_OK) o->next_plan_perm = nullptr;
          }
          if (rc_next == NFM_OK) {
            o->next_plan->ds_uid = ds->uid;
            o->next_plan->ds_nnz = ds->v.nnz;
            o->next_plan_epoch = o->shuffle_epoch + 1;
            o->next_plan_ready = true;
          }
        }
        NFM_HIP_CHECK(hipStreamSynchronize(st));
        out2[0] = o->out2_pinned[0];
        out2[1] = o->out2_pinned[1];
        NFM_TRY(rc_next);
      }
      if (dev_shuffle) o->shuffle_epoch++;
      NFM_TRY(rc_epoch);
      if (o->dp) {
        NFM_TRY(dp_fold_pending(de));  // (SGD: the closing exchange itself brings the arena to true values, scales 1)
        NFM_TRY(o->dp_sums.ensure(sizeof(double) * 3));
        double sums[3] = {out2[0], out2[1], (double)ns};
        NFM_HIP_CHECK(hipMemcpyAsync(o->dp_sums.p, sums, sizeof(sums), hipMemcpyHostToDevice, st));
        NFM_TRY(dp_epoch_end(de, o->dp_sums.as<double>()));
        NFM_HIP_CHECK(hipMemcpyAsync(sums, o->dp_sums.p, sizeof(sums), hipMemcpyDeviceToHost, st));
        NFM_HIP_CHECK(hipStreamSynchronize(st));
        out2[0] = sums[0];
        out2[1] = sums[1];
        // the reference's threads share ONE step counter (sgd_multi.nim:37): after the call it has advanced by the
        // samples of all ranks
        o->it += (int64_t)(sums[2] + 0.5) - ns;
      }
      dp_guard.ok = true;
    }
    o->it += o->kind == OPT_PSGD ? ns / o->batch : ns;
    if (o->kind == OPT_SGD && !o->dp) {  // resetScaling, sgd.nim:116-131
      double sc[SC_COUNT];
      NFM_HIP_CHECK(hipMemcpyAsync(sc, m->sc.p, sizeof(sc), hipMemcpyDeviceToHost, st));
      NFM_HIP_CHECK(hipStreamSynchronize(st));
      if (sc[SC_SCALE_P] < 1e-9 || (m->cfg.fit_linear && sc[SC_SCALE_W] < 1e-9)) NFM_TRY(launch_rescale(ctx, M));
    }
  }
  else if (o->dp) {
    // An empty shard (dp.shard_bounds hands ranks 0 .. W-2 nothing when there are fewer samples than ranks): this rank
    // still issues the collectives its peers wait in -- the agreement on the sync points (it offers none, so the group
    // has none mid-epoch), then the closing exchange with a zero increment -- and leaves with the group's sums and counter.
    NFM_CHECK(o->mode == NFM_MODE_MINIBATCH, NFM_ERR_UNSUPPORTED, "the data-parallel exchange needs NFM_MODE_MINIBATCH");
    const ModelView M = m->view();
    DpEpoch de;
    NFM_TRY(dp_epoch_setup(o, m, M, &de));
    NFM_TRY(dp_epoch_begin(de, 0, 0));
    NFM_TRY(o->dp_sums.ensure(sizeof(double) * 3));
    double sums[3] = {0.0, 0.0, 0.0};
    NFM_HIP_CHECK(hipMemcpyAsync(o->dp_sums.p, sums, sizeof(sums), hipMemcpyHostToDevice, st));
    NFM_TRY(dp_epoch_end(de, o->dp_sums.as<double>()));
    NFM_HIP_CHECK(hipMemcpyAsync(sums, o->dp_sums.p, sizeof(sums), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    out2[0] = sums[0];
    out2[1] = sums[1];
    o->it += (int64_t)(sums[2] + 0.5);
  }
  if (loss_sum) *loss_sum = out2[0];
  if (viol_sum) *viol_sum = out2[1];
  return NFM_OK;
}

int32_t nfm_opt_predict_all_with_grad(nfm_opt* o, nfm_dataset* ds, double* y_pred, double* dL, double* grad_P, double* grad_w,
                                      double* grad_b, double* loss_sum) {
  NFM_CHECK(o && ds, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(o->kind == OPT_PSGD, NFM_ERR_INVALID, "predictAllWithGrad needs an optimizer made by nfm_mbpsgd_create");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  nfm_ctx* ctx = m->ctx;
  NFM_CHECK(ds->ctx == ctx, NFM_ERR_INVALID, "optimizer and dataset belong to different contexts");
  NFM_TRY(check_predict_shapes(m, ds));
  NFM_CHECK(ds->has_y, NFM_ERR_INVALID, "dataset has no targets");
  NFM_TRY(check_trainable(ds));
  NFM_TRY(use_device(ctx));
  NFM_TRY(ensure_unit_scale(m));
  hipStream_t st = ctx->stream;
  const int64_t n = ds->v.n;
  const ModelView M = m->view();
  const size_t bP = pad256(sizeof(double) * std::max<int64_t>(m->nP(), 2)), bw = pad256(sizeof(double) * std::max<int64_t>(m->d, 1));
  DevBuf g, rec2;
  NFM_TRY(g.alloc(bP + bw + 256));
  NFM_HIP_CHECK(hipMemsetAsync(g.p, 0, g.bytes, st));  // features no sample touches keep a zero gradient
  double out2[2] = {0.0, 0.0};
  if (n > 0) {
    OptView O = o->o;
    O.bsize = (double)n;  // one mini-batch holding every sample: coef = dloss / nSamples (pgd.nim:102)
    O.gradP = g.as<double>();
    O.gradw = reinterpret_cast<double*>(g.as<char>() + bP);
    O.gradb = reinterpret_cast<double*>(g.as<char>() + bP + bw);
    MbWork& W = o->Wg;
    W.use_graph = false;
    if (!o->grad_plan || o->grad_plan->ds_uid != ds->uid || o->grad_plan->ds_nnz != ds->v.nnz || o->grad_plan->end != n ||
        o->grad_plan->n_aug != m->n_aug) {
      if (!o->grad_plan) o->grad_plan.reset(new Plan());
      const bool sort_by_count = m->Kp * (int)sizeof(double) >= 128;
      NFM_TRY(plan_build(ctx, ds->v, m->n_aug, nullptr, 0, n, n, false, false, false, sort_by_count, o->grad_plan.get()));
      o->grad_plan->ds_uid = ds->uid;
      o->grad_plan->ds_nnz = ds->v.nnz;
    }
    const Plan& plan = *o->grad_plan;
    NFM_TRY(mb_fm_epoch(ctx, OPT_PSGD, ds->v, M, O, plan, W, o->it, out2));
    if (y_pred || dL) {
      NFM_TRY(rec2.alloc(sizeof(double) * 2 * (size_t)n));
      NFM_TRY(mb_fm_records(ctx, W, n, rec2.as<double>(), rec2.as<double>() + n));
      if (y_pred) NFM_HIP_CHECK(hipMemcpyAsync(y_pred, rec2.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
      if (dL) NFM_HIP_CHECK(hipMemcpyAsync(dL, rec2.as<double>() + n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    }
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  if (grad_P && m->nP() > 0) {  // device [nb][da][Kp] -> the training layout [nb][da][k]
    DevBuf tmp;
    NFM_TRY(tmp.alloc(sizeof(double) * m->n_ref()));
    NFM_TRY(rows_from_device(m, g.as<double>(), tmp.as<double>(), nullptr));
    NFM_HIP_CHECK(hipMemcpyAsync(grad_P, tmp.p, sizeof(double) * m->n_ref(), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
  }
  if (grad_w) NFM_HIP_CHECK(hipMemcpyAsync(grad_w, g.as<char>() + bP, sizeof(double) * m->d, hipMemcpyDeviceToHost, st));
  if (grad_b) NFM_HIP_CHECK(hipMemcpyAsync(grad_b, g.as<char>() + bP + bw, sizeof(double), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  if (loss_sum) *loss_sum = out2[0];
  return NFM_OK;
}

int32_t nfm_opt_set_shuffle(nfm_opt* o, int64_t seed) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(seed < 0 || o->mode == NFM_MODE_MINIBATCH, NFM_ERR_UNSUPPORTED, "the device-side shuffle needs NFM_MODE_MINIBATCH");
  o->shuffle_seed = seed;
  o->shuffle_epoch = 0;
  o->next_plan_ready = false;
  return NFM_OK;
}

int32_t nfm_opt_announce_perm(nfm_opt* o, const int64_t* perm_next, int64_t begin, int64_t end) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(!perm_next || (begin >= 0 && begin < end), NFM_ERR_INVALID, "bad sample range [%lld,%lld)", (long long)begin, (long long)end);
  o->announced = o->mode == NFM_MODE_MINIBATCH ? perm_next : nullptr;
  o->announced_begin = begin;
  o->announced_end = end;
  return NFM_OK;
}

int32_t nfm_opt_get_perm(nfm_opt* o, int64_t* perm, int64_t n) {
  NFM_CHECK(o && (perm || n == 0), NFM_ERR_INVALID, "null argument");
  NFM_CHECK(o->plan && o->plan->has_perm && o->plan->perm.p, NFM_ERR_INVALID, "the last epoch ran in the dataset's own order");
  NFM_CHECK(n == o->plan->end - o->plan->begin, NFM_ERR_INVALID, "the last epoch covered %lld samples, not %lld",
            (long long)(o->plan->end - o->plan->begin), (long long)n);
  NFM_TRY(use_device(o->ctx));
  NFM_HIP_CHECK(hipMemcpyAsync(perm, o->plan->perm.p, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, o->ctx->stream));
  NFM_HIP_CHECK(hipStreamSynchronize(o->ctx->stream));
  return NFM_OK;
}

int32_t nfm_opt_set_dp(nfm_opt* o, nfm_dp* dp, int64_t sync_period, int32_t overlap) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(sync_period >= 0, NFM_ERR_INVALID, "sync_period must be >= 0");
  NFM_CHECK(!dp || dp->ctx == o->ctx, NFM_ERR_INVALID, "optimizer and group belong to different contexts");
  NFM_CHECK(!dp || o->mode == NFM_MODE_MINIBATCH, NFM_ERR_UNSUPPORTED, "the data-parallel exchange needs NFM_MODE_MINIBATCH");
  NFM_CHECK(!dp || o->kind != OPT_PSGD, NFM_ERR_UNSUPPORTED, "MBPSGD has no data-parallel mode");
  NFM_CHECK(!dp || dp_is_live(dp, dp->uid), NFM_ERR_INVALID, "the group was destroyed");
  o->dp = dp;
  o->dp_uid = dp ? dp->uid : 0;
  o->dp_sync_period = sync_period;
  o->dp_overlap = overlap != 0;
  return NFM_OK;
}

int32_t nfm_opt_set_touch_cap(nfm_opt* o, double cap) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(o->kind == OPT_SGD && o->mode == NFM_MODE_MINIBATCH, NFM_ERR_UNSUPPORTED,
            "the touch cap belongs to SGD in NFM_MODE_MINIBATCH (AdaGrad's state sums every step anyway)");
  NFM_CHECK(cap >= 1.0 && cap == cap, NFM_ERR_INVALID, "touch cap must be >= 1");
  if (o->o.touch_cap != cap) {
    o->o.touch_cap = cap;
    o->W.drop_graph();  // a captured epoch holds the optimizer's parameters by value
  }
  return NFM_OK;
}

int32_t nfm_opt_set_ada_cross(nfm_opt* o, double gamma) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(o->kind == OPT_ADAGRAD && o->mode == NFM_MODE_MINIBATCH, NFM_ERR_UNSUPPORTED,
            "the cross-product weight belongs to AdaGrad in NFM_MODE_MINIBATCH");
  NFM_CHECK(gamma >= 0.0 && gamma == gamma, NFM_ERR_INVALID, "the cross-product weight must be >= 0");
  if (o->o.ada_cross != gamma) {
    o->o.ada_cross = gamma;
    o->W.drop_graph();  // a captured epoch holds the optimizer's parameters by value
  }
  return NFM_OK;
}

int32_t nfm_opt_set_dp_combine(nfm_opt* o, int32_t combine) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(combine == NFM_DP_AUTO || combine == NFM_DP_MEAN || combine == NFM_DP_SUM || combine == NFM_DP_STATE_MEAN || combine == NFM_DP_STATE_RSQRT ||
                combine == NFM_DP_STATE_CROSS,
            NFM_ERR_INVALID, "combine must be NFM_DP_AUTO, NFM_DP_MEAN, NFM_DP_SUM, NFM_DP_STATE_MEAN, NFM_DP_STATE_RSQRT or NFM_DP_STATE_CROSS");
  o->dp_combine = combine;
  return NFM_OK;
}

int32_t nfm_opt_finalize(nfm_opt* o) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  o->announced = nullptr;  // the end of a fit (or a callback): an announced order refers to an array of the caller's loop
  NFM_TRY(use_device(m->ctx));
  if (o->kind == OPT_SGD) {
    NFM_TRY(launch_rescale(m->ctx, m->view()));
  } else if (o->kind == OPT_PSGD) {
    // pgd.finalize (optimizer/pgd.nim:45-51) only copies the parameters back
  } else if (o->state_ready) {
    NFM_TRY(launch_adagrad_finalize(m->ctx, m->view(), o->o, o->it));
  }
  NFM_HIP_CHECK(hipStreamSynchronize(m->ctx->stream));
  return NFM_OK;
}

int32_t nfm_opt_device_state(nfm_opt* o, double** gsum_P, double** gnorm_P, int64_t* n_P, double** gsum_w, double** gnorm_w,
                             int64_t* n_w, double** gscalars) {
  NFM_CHECK(o, NFM_ERR_INVALID, "null optimizer");
  NFM_CHECK(o->kind == OPT_ADAGRAD, NFM_ERR_INVALID, "only AdaGrad carries state");
  nfm_model* m = nullptr;
  NFM_TRY(model_of(o, &m));
  if (!o->state_ready) {
    NFM_TRY(use_device(m->ctx));
    NFM_TRY(adagrad_reset_state(o));
  }
  if (gsum_P) *gsum_P = o->G.as<double>();
  if (gnorm_P) *gnorm_P = o->N.as<double>();
  if (n_P) *n_P = m->nP();
  if (gsum_w) *gsum_w = o->Gw.as<double>();
  if (gnorm_w) *gnorm_w = o->Nw.as<double>();
  if (n_w) *n_w = m->d;
  if (gscalars) *gscalars = o->gsc.as<double>();
  return NFM_OK;
}

// ------------------------------------------------------------------ host-side random numbers
// The reference draws P (tensor/tensor.nim:561-580) and shuffles the sample order (optimizer/sgd.nim:297) on the host
// with Nim's global generator.  A Nim host keeps doing exactly that; hosts in other languages (nimfm_amd/host.py,
// nimfm_amd/host/nimfm.hpp) get the same procedures here.  The generator is Nim 1.0's lib/pure/random.nim
// (xoroshiro128+), which is NOT part of the reference tree: restated from memory, unverified.
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t nim_next(uint64_t* st) {
  const uint64_t s0 = st[0];
  uint64_t s1 = st[1];
  const uint64_t r = s0 + s1;
  s1 ^= s0;
  st[0] = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
  st[1] = rotl64(s1, 36);
  return r;
}
static inline double nim_rand1(uint64_t* st) {  // rand(1.0): 52 random mantissa bits in [1, 2) minus 1
  const uint64_t u = (0x3FFull << 52) | (nim_next(st) >> 12);
  double f;
  memcpy(&f, &u, sizeof(f));
  return f - 1.0;
}

int32_t nfm_rng_randomize(int64_t seed, uint64_t* state) {  // randomize(seed) = initRand(seed)
  NFM_CHECK(state, NFM_ERR_INVALID, "null state");
  state[0] = (uint64_t)seed >> 16;
  state[1] = (uint64_t)seed & 0xffffull;
  (void)nim_next(state);
  return NFM_OK;
}

int32_t nfm_rng_random_normal(uint64_t* state, int64_t n, double loc, double scale, double* out) {
  NFM_CHECK(state && n >= 0 && (out || n == 0), NFM_ERR_INVALID, "null argument");
  // tensor/tensor.nim:561-580: Box-Muller, the two values of one (x, y) draw go to CONSECUTIVE elements of the
  // row-major fill (the pairing runs on across rows and blocks); an odd count leaves the sine half unused
  double x = 0.0, y = 0.0;
  bool has = false;
  const double two_pi = 2 * 3.14159265358979323846;
  for (int64_t t = 0; t < n; ++t) {
    double z;
    if (!has) {
      x = nim_rand1(state);
      y = nim_rand1(state);
      z = sqrt(-2 * log(1.0 - x)) * cos(two_pi * y);
      has = true;
    } else {
      z = sqrt(-2 * log(1.0 - x)) * sin(two_pi * y);
      has = false;
    }
    out[t] = loc + z * scale;
  }
  return NFM_OK;
}

int32_t nfm_rng_shuffle(uint64_t* state, int64_t* x, int64_t n) {
  NFM_CHECK(state && n >= 0 && (x || n == 0), NFM_ERR_INVALID, "null argument");
  // shuffle: for i in countdown(high, 1): swap(x[i], x[rand(i)]); rand(max: int) rejects the top sliver of the
  // 64-bit range and reduces modulo max + 1
  const uint64_t rand_max = ~0ull;
  for (int64_t i = n - 1; i >= 1; --i) {
    uint64_t r;
    do {
      r = nim_next(state);
    } while (r > rand_max - (rand_max % (uint64_t)i));
    const int64_t j = (int64_t)(r % ((uint64_t)i + 1ull));
    const int64_t t = x[i];
    x[i] = x[j];
    x[j] = t;
  }
  return NFM_OK;
}

int32_t nfm_opt_destroy(nfm_opt* o) {
  if (!o) return NFM_OK;
  (void)hipSetDevice(o->ctx->device);
  (void)hipStreamSynchronize(o->ctx->stream);
  if (o->plan_stream) {
    (void)hipStreamSynchronize(o->plan_stream);
    (void)hipStreamDestroy(o->plan_stream);
  }
  if (o->out2_pinned) (void)hipHostFree(o->out2_pinned);
  delete o;
  return NFM_OK;
}

}  // extern "C"
