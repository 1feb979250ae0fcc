// nimfm_amd/csrc/common.h -- internal declarations shared by the HIP translation units of
// libnimfm_hip.so.  gfx950 (MI355X, CDNA4) only: 64-lane wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/nimfm_hip.h"

namespace nfm {

int set_error(int code, const char* fmt, ...);
const char* last_error();

#define NFM_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return nfm::set_error(NFM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                             \
  } while (0)

#define NFM_CHECK(cond, code, ...) \
  do {                             \
    if (!(cond)) return nfm::set_error(code, __VA_ARGS__); \
  } while (0)

#define NFM_TRY(expr)        \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != NFM_OK) return rc_; \
  } while (0)

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

// ---------------------------------------------------------------------------------------------
// Device data layout
//   CSR:    indptr int64[n+1], indices int32[nnz], data f64[nnz], fields int32[nnz]?, y f64[n]?
//   params: P  f64[nb][da][Kp]   FM: nb = nOrders, da = d + nAug.  FFM: f64[d][nFields][Kp] (feature-major,
//                                ModelView::row); the reference's [nFields][d][k] is converted at the C ABI.
//                                Kp = 2*L >= k, L = lanes per row (power of two); row = Kp*8 bytes,
//                                16-byte aligned; padding s >= k is kept at exactly 0.
//           w  f64[d]
//           sc f64[8]: sc[0] = scale_P, sc[1] = scale_w, sc[2] = intercept.
//   The true parameter values are scale_P * P and scale_w * w: the reference's lazy L2 scaling
//   (optimizer/sgd.nim:99-143) kept as ONE global factor, so decay never touches memory.
// ---------------------------------------------------------------------------------------------
struct CsrView {
  const int64_t* indptr;
  const int32_t* indices;
  const double* data;
  const int32_t* fields;
  const double* y;
  int64_t n, d, nnz;
  int32_t n_fields;
  int32_t max_row;  // longest row (stored entries)
};

enum { SC_SCALE_P = 0, SC_SCALE_W = 1, SC_INTERCEPT = 2, SC_COUNT = 8 };

struct ModelView {
  double* P;
  double* w;
  double* sc;
  const double* lams;  // [kc][Kp], zero padded: block b's factors at lams + (b % kc) * Kp
  int64_t d, da;
  int32_t nb, k, Kp, L, degree, n_aug, kind, fit_linear, fit_intercept, task;
  // FM with more than 128 factors: the factors of one order are cut into kc device blocks of at most 128 (k = factors per
  // block, the last one zero padded) -- an ANOVA kernel is a sum over the factors of terms that do not mix them
  // (kernels.nim:46-64), so kc blocks of the same degree ARE the wide model, and every kernel that walks "orders" takes them
  // as they come: block b belongs to order b / kc.  kc = 1 otherwise.
  int32_t kc;
  __host__ __device__ int deg_of(int b) const { return degree - b / kc; }
  // parameter row (block b, feature j) starts at (b * bs + j * rs) * Kp; every kernel goes through row().
  // FM: order-major (bs = da, rs = 1).  FFM: FEATURE-major (bs = 1, rs = nb): the nb field rows of one feature --
  // what a sample reads and, in the reference's update, writes together -- are one contiguous run of nb * Kp doubles
  // instead of nb pieces of Kp * 8 bytes a whole table apart (F = 16, k = 8: 1 KB instead of 16 x 64 B).
  // (Feature-major for FM with several orders was measured too: cfg5, degree 3, k = 8 got 6 % SLOWER -- the kernels
  // walk one order at a time, and interleaving the orders doubles the lines each such pass touches.)
  int64_t bs, rs;
  __host__ __device__ size_t row(int64_t b, int64_t j) const { return (size_t)(b * bs + j * rs); }
};

struct LossCfg {
  int32_t loss;
  double param;
};

struct Timing {
  struct Acc {
    int64_t launches = 0;
    double ms = 0.0;
  };
  struct Pending {
    std::string family;
    hipEvent_t start, stop;
  };
  bool enabled = false;
  std::map<std::string, Acc> acc;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
};

}  // namespace nfm

namespace nfm {
struct DevBuf;
}
struct nfm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  nfm::Timing timing;
  int n_cu = 256;
  // decisionFunction of a model with several orders: the table with the orders interleaved per feature (predict.hip).  It
  // belongs to the context, not to the call: the kernels that read it are still queued on `stream` when the call returns, and
  // a block handed back to the device-memory cache may go to another stream (nfm_decision_function_device never synchronizes)
  nfm::DevBuf* predict_pf = nullptr;
  // NFM_MODE_SEQUENTIAL, the one-sample-in-flight kernel on rows whose per-sample gradient does not fit the LDS: the
  // gradient scratch in global memory (seq.hip); on the context for the same reason as predict_pf
  nfm::DevBuf* seq_scratch = nullptr;
};

namespace nfm {

// RAII-ish helper: brackets the launches issued between begin() and end() with events when
// timing is enabled.
struct TimedLaunch {
  nfm_ctx* ctx;
  const char* family;
  hipEvent_t start = nullptr, stop = nullptr;
  TimedLaunch(nfm_ctx* c, const char* f);
  ~TimedLaunch();
};
int timing_flush(nfm_ctx* ctx);

struct DevBuf {  // owned device allocation (blocks are recycled through a cache, util.hip)
  void* p = nullptr;
  size_t bytes = 0;  // what was asked for
  size_t cap = 0;    // what the block holds
  int device = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  int alloc(size_t nbytes);
  int ensure(size_t nbytes) { return nbytes <= bytes && p ? NFM_OK : alloc(nbytes); }
  void release();
  void take(DevBuf& o) {  // move
    release();
    p = o.p;
    bytes = o.bytes;
    cap = o.cap;
    device = o.device;
    o.p = nullptr;
    o.bytes = 0;
    o.cap = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// SPLIT (row slots per sample, fm_device.h) for a launch over n samples with avg_row nnz per row:
// enough wavefronts to fill the chip (>= 16 per CU) and a serial chain of at most ~8 rounds of
// kUnroll loads per lane (measured on cfg2: fewer slots win as soon as the chip is full); capped by the 64/L slots a wavefront has and by 16.
inline int choose_split(int L, int64_t n, double avg_row, int n_cu) {
  const int R = kWave / L;
  auto p2 = [](double v) {
    int r = 1;
    while (r < v && r < 64) r <<= 1;
    return r;
  };
  if (const char* env = getenv("NFM_SPLIT")) {  // tuning override
    int s = atoi(env);
    if (s >= 1) return s > R ? R : (s > 16 ? 16 : p2(s));
  }
  const double occ = (double)n_cu * 16.0 * kWave / ((double)(n > 0 ? n : 1) * L);
  int s = p2(occ);
  const int lat = p2(avg_row / 32.0);
  if (lat > s) s = lat;
  if (s > R) s = R;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

inline int lanes_for_k(int k) {
  int half = (k + 1) / 2, L = 1;
  while (L < half) L <<= 1;
  return L;
}

// ---- util.hip ----
int launch_fill(nfm_ctx* ctx, double* p, int64_t n, double v);
// reference FM layout [nb][k][da] <-> device [nb][da][Kp]; FFM reference [nb][da][k] <-> device
// (bs, rs, b0): ModelView's block / row strides of the device tensor and the device block the first reference block lands in;
// 0, 0, 0 = order-major from block 0 (bs = da, rs = 1)
int launch_fm_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int nb, int k, int Kp, int64_t da, int64_t bs = 0, int64_t rs = 0,
                        int b0 = 0);
int launch_fm_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int nb, int k, int Kp, int64_t da, const double* scale_dev,
                          int64_t bs = 0, int64_t rs = 0, int b0 = 0);
// nb_major > 0: the reference tensor is [nb_major][rows / nb_major][k], the device tensor feature-major (ModelView::row)
int launch_rows_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int64_t rows, int k, int Kp, double pad_value, int nb_major = 0);
int launch_rows_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int64_t rows, int k, int Kp, const double* scale_dev, int nb_major = 0);
// FMs with more than 128 factors: reference [no * da][k] <-> device [no * kc][da][Kp] (ModelView::kc)
int launch_rows_split_to_device(nfm_ctx* ctx, const double* src_ref, double* dst_dev, int64_t no, int64_t da, int k, int kc, int kb, int Kp,
                                double pad_value, int64_t bs, int64_t rs);
int launch_rows_split_from_device(nfm_ctx* ctx, const double* src_dev, double* dst_ref, int64_t no, int64_t da, int k, int kc, int kb, int Kp,
                                  const double* scale_dev, int64_t bs, int64_t rs);
// P *= sc[SC_SCALE_P], w *= sc[SC_SCALE_W] (if fit_linear), scales := 1  (sgd.nim:99-113)
int launch_rescale(nfm_ctx* ctx, const ModelView& M);
int launch_sqnorms(nfm_ctx* ctx, const ModelView& M, double* out2_dev /*{P_sq,w_sq}*/);
int launch_narrow_i64_i32(nfm_ctx* ctx, const int64_t* src, int32_t* dst, int64_t n);

// ---- predict.hip ----
int launch_predict(nfm_ctx* ctx, const CsrView& X, const ModelView& M, double* out_dev);
int launch_metrics(nfm_ctx* ctx, int64_t n, const double* scores, const double* y, double* rmse, double* accuracy,
                   double* rocauc);

}  // namespace nfm
