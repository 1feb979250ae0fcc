// nimfm_amd/host/nimfm.hpp -- header-only C++17 host mirror of nimfm's FM surface over the C ABI
// (include/nimfm_hip.h).  The reference's host language (Nim) is compiled code and no Nim toolchain
// exists in the build image; this header is the compiled-language counterpart of nim/nimfm_hip.nim:
// same names, argument meaning, defaults and error behaviour as the reference procs
// (citations relative to /root/reference/src/nimfm/):
//   FactorizationMachine            model/factorization_machine.nim:11-139, model/fm_base.nim:13-48
//   SGD<Loss>::fit / AdaGrad::fit   optimizer/sgd.nim:23-52,261-328, optimizer/adagrad.nim:20-44,137-203
// Errors of the library surface as std::invalid_argument (the reference's ValueError),
// nimfm::NotFittedError, or std::runtime_error.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

#include "../../include/nimfm_hip.h"

namespace nimfm {

struct NotFittedError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void check(int32_t rc) {
  if (rc == NFM_OK) return;
  const std::string msg = nfm_last_error();
  if (rc == NFM_ERR_INVALID) throw std::invalid_argument(msg);
  if (rc == NFM_ERR_NOT_FITTED) throw NotFittedError(msg);
  throw std::runtime_error("libnimfm_hip error " + std::to_string(rc) + ": " + msg);
}

inline nfm_ctx* default_context() {
  static nfm_ctx* ctx = nullptr;
  if (!ctx) check(nfm_ctx_create(0, nullptr, &ctx));
  return ctx;
}

enum TaskKind { regression = NFM_TASK_REGRESSION, classification = NFM_TASK_CLASSIFICATION };
enum FitLowerKind { explicit_ = NFM_LOWER_EXPLICIT, augment = NFM_LOWER_AUGMENT, none = NFM_LOWER_NONE };
enum SchedulingKind { constant = NFM_SCHED_CONSTANT, optimal = NFM_SCHED_OPTIMAL, invscaling = NFM_SCHED_INVSCALING, pegasos = NFM_SCHED_PEGASOS };
struct Squared { static constexpr int id = NFM_LOSS_SQUARED; double param = 1.0; };
struct SquaredHinge { static constexpr int id = NFM_LOSS_SQUARED_HINGE; double param = 1.0; };
struct Logistic { static constexpr int id = NFM_LOSS_LOGISTIC; double param = 1.0; };
struct Huber { static constexpr int id = NFM_LOSS_HUBER; double param = 1.0; /* threshold */ };

// dataset.nim:10-16 + tensor/sparse.nim:9-12: CSR rows resident on the device
class CSRDataset {
 public:
  CSRDataset(const std::vector<double>& data, const std::vector<int64_t>& indices, const std::vector<int64_t>& indptr,
             int64_t nSamples, int64_t nFeatures)
      : n_(nSamples), d_(nFeatures) {
    if ((int64_t)indptr.size() != nSamples + 1) throw std::invalid_argument("len(indptr) != nSamples + 1");
    check(nfm_dataset_create_csr(default_context(), nSamples, nFeatures, indptr.data(), indices.data(), data.data(), nullptr,
                                 0, nullptr, &h_));
  }
  // a dataset made by one of the loaders below (the handle is adopted)
  explicit CSRDataset(nfm_dataset* h) : h_(h) {
    int64_t nnz = 0, nf = 0;
    check(nfm_dataset_shape(h_, &n_, &d_, &nnz, &nf));
  }
  CSRDataset(const CSRDataset&) = delete;
  ~CSRDataset() { nfm_dataset_destroy(h_); }
  std::vector<double> targets() const {  // the loaders' y
    std::vector<double> y((size_t)n_);
    check(nfm_dataset_get_targets(h_, y.data()));
    return y;
  }
  int64_t nSamples() const { return n_; }
  int64_t nFeatures() const { return d_; }
  nfm_dataset* handle() const { return h_; }

 private:
  nfm_dataset* h_ = nullptr;
  int64_t n_, d_;
};

// loadSVMLightFile (dataset.nim:616-632): the text is parsed on the GPU, y comes back with the dataset
inline std::unique_ptr<CSRDataset> loadSVMLightFile(const std::string& f, std::vector<double>& y, int64_t nFeatures = -1) {
  nfm_dataset* h = nullptr;
  check(nfm_dataset_load_svmlight(default_context(), f.c_str(), nFeatures, &h));
  std::unique_ptr<CSRDataset> X(new CSRDataset(h));
  y = X->targets();
  return X;
}
// newStreamCSRDataset + loadStreamLabel (dataset.nim:170-174, 1007-1014): the whole file becomes resident
inline std::unique_ptr<CSRDataset> newStreamCSRDataset(const std::string& f, const std::string& fY = "") {
  nfm_dataset* h = nullptr;
  check(nfm_dataset_load_stream(default_context(), f.c_str(), fY.empty() ? nullptr : fY.c_str(), &h));
  return std::unique_ptr<CSRDataset>(new CSRDataset(h));
}

class FactorizationMachine {
 public:
  TaskKind task;
  int degree, nComponents;
  FitLowerKind fitLower;
  bool fitIntercept, fitLinear, warmStart;
  int randomState;
  double scale;
  bool isInitialized = false;
  std::vector<double> P;  // [nOrders][nComponents][nFeatures + nAugments]
  std::vector<double> lams, w;
  double intercept = 0.0;

  // newFactorizationMachine, model/factorization_machine.nim:43-78
  explicit FactorizationMachine(TaskKind task_, int degree_ = 2, int nComponents_ = 30, FitLowerKind fitLower_ = explicit_,
                                bool fitIntercept_ = true, bool fitLinear_ = true, bool warmStart_ = false,
                                int randomState_ = 1, double scale_ = 0.01)
      : task(task_), degree(degree_), nComponents(nComponents_), fitLower(fitLower_), fitIntercept(fitIntercept_),
        fitLinear(fitLinear_), warmStart(warmStart_), randomState(randomState_), scale(scale_) {
    if (degree < 1) throw std::invalid_argument("degree < 1.");
    if (nComponents < 1) throw std::invalid_argument("nComponents < 1.");
    lams.assign(nComponents, 1.0);
  }
  FactorizationMachine(const FactorizationMachine&) = delete;
  ~FactorizationMachine() { if (h_) nfm_model_destroy(h_); }

  int nAugments() const { return fitLower == augment ? (fitLinear ? degree - 2 : degree - 1) : 0; }  // :81-86
  int nOrders() const { return degree == 1 ? 0 : (fitLower == explicit_ ? degree - 1 : 1); }          // :89-97

  // init, :125-139 (std::mt19937_64 + normal_distribution stand in for Nim's RNG)
  void init(const CSRDataset& X, bool force = false) {
    if (force || !(warmStart && isInitialized)) {
      d_ = X.nFeatures();
      rng_.seed((uint64_t)randomState);
      std::normal_distribution<double> g(0.0, scale);
      w.assign(d_, 0.0);
      P.resize((size_t)nOrders() * nComponents * (d_ + nAugments()));
      for (auto& v : P) v = g(rng_);
      intercept = 0.0;
      dirty_ = true;
    }
    isInitialized = true;
  }
  void setParams(std::vector<double> P_, std::vector<double> w_, double b) {
    d_ = (int64_t)w_.size();
    if (P_.size() != (size_t)nOrders() * nComponents * (d_ + nAugments())) throw std::invalid_argument("bad P shape");
    P = std::move(P_); w = std::move(w_); intercept = b; isInitialized = true; dirty_ = true;
  }
  // decisionFunction, :100-122
  std::vector<double> decisionFunction(const CSRDataset& X) {
    if (!isInitialized) throw NotFittedError("Factorization machines is not fitted.");
    if (X.nFeatures() != d_) throw std::invalid_argument("Invalid nFeatures.");
    std::vector<double> out(X.nSamples());
    check(nfm_decision_function(push(), X.handle(), out.data()));
    return out;
  }
  std::vector<int> predict(const CSRDataset& X) {  // fm_base.nim:18-20
    auto y = decisionFunction(X);
    std::vector<int> r(y.size());
    for (size_t i = 0; i < y.size(); ++i) r[i] = (y[i] > 0) - (y[i] < 0);
    return r;
  }
  double score(const CSRDataset& X, const std::vector<double>& y) {  // fm_base.nim:39-48, reduced on the device
    if (!isInitialized) throw NotFittedError("Factorization machines is not fitted.");
    if (X.nFeatures() != d_) throw std::invalid_argument("Invalid nFeatures.");
    if ((int64_t)y.size() != X.nSamples()) throw std::invalid_argument("len(y) != nSamples");
    check(nfm_dataset_set_targets(X.handle(), y.data()));
    double out = 0.0;
    check(nfm_score(push(), X.handle(), &out));
    return out;
  }

  nfm_model* push() {  // device copy of the host parameters
    if (!h_ || hd_ != d_) {
      if (h_) nfm_model_destroy(h_);
      nfm_model_cfg c{NFM_KIND_FM, (int32_t)task, degree, nComponents, (int32_t)fitLower, fitIntercept, fitLinear, 0, d_, 0};
      check(nfm_model_create(default_context(), &c, &h_));
      hd_ = d_;
      dirty_ = true;
    }
    if (dirty_) {
      check(nfm_model_set_params(h_, P.empty() ? nullptr : P.data(), w.data(), intercept, lams.data()));
      dirty_ = false;
    }
    return h_;
  }
  void pull() { check(nfm_model_get_params(h_, P.empty() ? nullptr : P.data(), w.data(), &intercept)); dirty_ = false; }
  std::mt19937_64& rng() { return rng_; }

 private:
  nfm_model* h_ = nullptr;
  int64_t d_ = 0, hd_ = -1;
  bool dirty_ = true;
  std::mt19937_64 rng_;
};

namespace detail {
// the epoch loop shared by SGD and AdaGrad: optimizer/sgd.nim:294-328, adagrad.nim:164-203
template <class Opt>
void run_fit(Opt& self, nfm_opt* o, const CSRDataset& X, FactorizationMachine& fm,
             const std::function<void(Opt&, FactorizationMachine&)>& callback, bool callback_each_epoch) {
  const int64_t n = X.nSamples();
  std::vector<int64_t> indices(n);
  std::iota(indices.begin(), indices.end(), 0);
  bool isConverged = false;
  check(nfm_opt_set_it(o, self.it));
  if (self.verbose > 0) std::printf("Epoch   Violation    Loss         Regularization\n");
  for (int epoch = 0; epoch < self.maxIter; ++epoch) {
    double viol = 0.0, runningLoss = 0.0;
    const int64_t* perm = nullptr;
    if (self.shuffle) {
      std::shuffle(indices.begin(), indices.end(), fm.rng());
      perm = indices.data();
    }
    check(nfm_opt_epoch(o, X.handle(), perm, 0, n, &runningLoss, &viol));
    self.it += n;
    runningLoss /= (double)n;
    if (callback && callback_each_epoch) {
      check(nfm_opt_finalize(o));
      fm.pull();
      callback(self, fm);
    }
    bool isContinue = true;  // stoppingCriterion, sgd.nim:72-89
    if (std::isnan(runningLoss)) { std::printf("Loss is NaN. Use smaller learning rate.\n"); isContinue = false; }
    if (self.verbose > 0) {
      double psq = 0, wsq = 0, b = 0;
      check(nfm_model_sqnorms(fm.push(), &psq, &wsq));
      check(nfm_model_get_params(fm.push(), nullptr, nullptr, &b));
      std::printf("%-5d   %-10.4e   %-10.4e   %-10.4e\n", epoch + 1, viol, runningLoss,
                  0.5 * self.alpha0 * b * b + 0.5 * self.alpha * wsq + 0.5 * self.beta * psq);
    }
    if (viol < self.tol) {
      if (self.verbose > 0) std::printf("Converged at epoch %d.\n", epoch);
      isConverged = true;
      isContinue = false;
    }
    if (!isContinue) break;
  }
  if (!isConverged && self.verbose > 0) std::printf("Objective did not converge. Increase maxIter.\n");
  check(nfm_opt_finalize(o));
  fm.pull();
}
}  // namespace detail

template <class L = Squared>
class SGD {
 public:
  int maxIter; double eta0, alpha0, alpha, beta; L loss; SchedulingKind scheduling; double power;
  int verbose; double tol; bool shuffle; int nCalls; int64_t it = 1;
  int mode = NFM_MODE_SEQUENTIAL; int64_t batch = 8192;
  double touchCap = 1.0;  // mini-batch mode: nfm_opt_set_touch_cap (1 = the per-coordinate mean; about twice the touches per coordinate and batch)
  // newSGD, optimizer/sgd.nim:23-52
  explicit SGD(int maxIter_ = 100, double eta0_ = 0.01, double alpha0_ = 1e-6, double alpha_ = 1e-3, double beta_ = 1e-3,
               L loss_ = L(), SchedulingKind scheduling_ = optimal, double power_ = 1.0, int verbose_ = 1, double tol_ = 1e-3,
               bool shuffle_ = true, int nCalls_ = -1)
      : maxIter(maxIter_), eta0(eta0_), alpha0(alpha0_), alpha(alpha_), beta(beta_), loss(loss_), scheduling(scheduling_),
        power(power_), verbose(verbose_), tol(tol_), shuffle(shuffle_), nCalls(nCalls_) {}
  ~SGD() { if (o_) nfm_opt_destroy(o_); }
  // fit, optimizer/sgd.nim:261-328; maxThreads != 0 = the Hogwild overload (sgd_multi.nim:40-42) -> mini-batch mode
  // maxThreads only SELECTS the mini-batch mode (a thread count is not a batch size); the mode's knobs are explicit:
  // miniBatchSize (0: this->batch), and across GPUs -- one process per GPU, X this rank's slice -- group + syncPeriod
  void fit(const CSRDataset& X, const std::vector<double>& y, FactorizationMachine& fm, int maxThreads = 0,
           std::function<void(SGD&, FactorizationMachine&)> callback = nullptr, int64_t miniBatchSize = 0,
           int64_t syncPeriod = 0, nfm_dp* group = nullptr) {
    fm.init(X);
    if ((int64_t)y.size() != X.nSamples()) throw std::invalid_argument("len(y) != nSamples");
    check(nfm_dataset_set_targets(X.handle(), y.data()));
    if (!fm.warmStart) it = 1;
    nfm_model* m = fm.push();
    const int md = (maxThreads != 0 || group) ? NFM_MODE_MINIBATCH : mode;
    const int64_t bsz = miniBatchSize > 0 ? miniBatchSize : batch;
    if (!o_ || m_ != m || md_ != md || b_ != bsz) {
      if (o_) nfm_opt_destroy(o_);
      nfm_sgd_cfg c{eta0, alpha0, alpha, beta, power, loss.param, L::id, (int32_t)scheduling, md, 0, bsz};
      check(nfm_sgd_create(m, &c, &o_));
      m_ = m; md_ = md; b_ = bsz; cap_ = 1.0;
    }
    if (md == NFM_MODE_MINIBATCH && cap_ != touchCap) { check(nfm_opt_set_touch_cap(o_, touchCap)); cap_ = touchCap; }
    if (md == NFM_MODE_MINIBATCH) check(nfm_opt_set_dp(o_, group, syncPeriod, 1));
    detail::run_fit<SGD>(*this, o_, X, fm, callback, nCalls <= 0 || md == NFM_MODE_MINIBATCH);
  }

 private:
  nfm_opt* o_ = nullptr; nfm_model* m_ = nullptr; int md_ = -1; int64_t b_ = -1; double cap_ = 1.0;
};

template <class L = Squared>
class AdaGrad {
 public:
  int maxIter; double eta0, alpha0, alpha, beta; L loss; double eps; int verbose; double tol; bool shuffle; int nCalls;
  int64_t it = 1; int mode = NFM_MODE_SEQUENTIAL; int64_t batch = 8192;
  double adaCross = 0.0;   // mini-batch mode: nfm_opt_set_ada_cross (weight of the batch's gradient cross products in g_norm; 0.1 at large batches)
  bool trackViol = true;   // adagrad.nim:99's sum |P_old - P_new| (the stopping criterion); false saves the stored-parameter round trip
  // newAdaGrad, optimizer/adagrad.nim:20-44
  explicit AdaGrad(int maxIter_ = 100, double eta0_ = 0.1, double alpha0_ = 1e-6, double alpha_ = 1e-3, double beta_ = 1e-3,
                   L loss_ = L(), double eps_ = 1e-10, int verbose_ = 1, double tol_ = 1e-3, bool shuffle_ = true,
                   int nCalls_ = -1)
      : maxIter(maxIter_), eta0(eta0_), alpha0(alpha0_), alpha(alpha_), beta(beta_), loss(loss_), eps(eps_),
        verbose(verbose_), tol(tol_), shuffle(shuffle_), nCalls(nCalls_) {}
  ~AdaGrad() { if (o_) nfm_opt_destroy(o_); }
  void fit(const CSRDataset& X, const std::vector<double>& y, FactorizationMachine& fm, int maxThreads = 0,
           std::function<void(AdaGrad&, FactorizationMachine&)> callback = nullptr, int64_t miniBatchSize = 0,
           int64_t syncPeriod = 0, nfm_dp* group = nullptr) {
    fm.init(X);
    if ((int64_t)y.size() != X.nSamples()) throw std::invalid_argument("len(y) != nSamples");
    check(nfm_dataset_set_targets(X.handle(), y.data()));
    if (!fm.warmStart) it = 1;
    nfm_model* m = fm.push();
    const int md = (maxThreads != 0 || group) ? NFM_MODE_MINIBATCH : mode;
    const int64_t bsz = miniBatchSize > 0 ? miniBatchSize : batch;
    if (!o_ || m_ != m || md_ != md || b_ != bsz || tv_ != trackViol) {
      if (o_) nfm_opt_destroy(o_);
      nfm_adagrad_cfg c{eta0, alpha0, alpha, beta, eps, loss.param, L::id, md, trackViol ? 1 : 0, 0, bsz};
      check(nfm_adagrad_create(m, &c, &o_));
      m_ = m; md_ = md; b_ = bsz; tv_ = trackViol; cross_ = 0.0;
    }
    if (md == NFM_MODE_MINIBATCH && cross_ != adaCross) { check(nfm_opt_set_ada_cross(o_, adaCross)); cross_ = adaCross; }
    if (md == NFM_MODE_MINIBATCH) check(nfm_opt_set_dp(o_, group, syncPeriod, 1));
    detail::run_fit<AdaGrad>(*this, o_, X, fm, callback, true);
  }

 private:
  nfm_opt* o_ = nullptr; nfm_model* m_ = nullptr; int md_ = -1; int64_t b_ = -1; bool tv_ = true; double cross_ = 0.0;
};

// regularizer/{l1,l21,squaredl12,squaredl21}.nim: the penalties with a matrix proximal operator
struct L1 { static constexpr int id = NFM_REG_L1; bool transpose = false; };
struct L21 { static constexpr int id = NFM_REG_L21; bool transpose = false; };
struct SquaredL12 { static constexpr int id = NFM_REG_SQUAREDL12; bool transpose = true; /* squaredl12.nim:85 */ };
struct SquaredL21 { static constexpr int id = NFM_REG_SQUAREDL21; bool transpose = false; /* squaredl21.nim:15 */ };

// MBPSGD[L, R], optimizer/minibatch_psgd.nim:11-65,125-210
template <class L = Squared, class R = SquaredL12>
class MBPSGD {
 public:
  int maxIter; double eta0, alpha0, alpha, beta, gamma; L loss; R reg; int64_t miniBatchSize, maxIterInner;
  SchedulingKind scheduling; double power; int verbose; double tol; bool shuffle; int64_t it = 0;
  explicit MBPSGD(int maxIter_ = 100, double eta0_ = 0.1, double alpha0_ = 1e-6, double alpha_ = 1e-3, double beta_ = 1e-4,
                  double gamma_ = 1e-4, L loss_ = L(), R reg_ = R(), int64_t miniBatchSize_ = -1, int64_t maxIterInner_ = -1,
                  SchedulingKind scheduling_ = optimal, double power_ = 1.0, int verbose_ = 1, double tol_ = 1e-6,
                  bool shuffle_ = true)
      : maxIter(maxIter_), eta0(eta0_), alpha0(alpha0_), alpha(alpha_), beta(beta_), gamma(gamma_), loss(loss_), reg(reg_),
        miniBatchSize(miniBatchSize_), maxIterInner(maxIterInner_), scheduling(scheduling_), power(power_),
        verbose(verbose_), tol(tol_), shuffle(shuffle_) {}
  ~MBPSGD() { if (o_) nfm_opt_destroy(o_); }
  void fit(const CSRDataset& X, const std::vector<double>& y, FactorizationMachine& sfm,
           std::function<void(MBPSGD&, FactorizationMachine&)> callback = nullptr) {
    sfm.init(X);
    if ((int64_t)y.size() != X.nSamples()) throw std::invalid_argument("len(y) != nSamples");
    check(nfm_dataset_set_targets(X.handle(), y.data()));
    if (!sfm.warmStart) it = 1;  // :153-154
    const int64_t n = X.nSamples();
    int64_t nnz = 0;
    check(nfm_dataset_shape(X.handle(), nullptr, nullptr, &nnz, nullptr));
    int64_t B = miniBatchSize;
    if (B <= 0) B = std::max<int64_t>((X.nFeatures() * n) / std::max<int64_t>(nnz, 1), 1);  // :160-163
    int64_t inner = maxIterInner;
    if (inner <= 0) inner = std::max<int64_t>((n - 1) / B + 1, 1);  // :164-167
    nfm_model* m = sfm.push();
    if (!o_ || m_ != m || B_ != B) {
      if (o_) nfm_opt_destroy(o_);
      o_ = nullptr;
      nfm_mbpsgd_cfg c{eta0, alpha0, alpha, beta, gamma, power, loss.param, L::id, (int32_t)scheduling, R::id,
                       reg.transpose ? 1 : 0, B};
      check(nfm_mbpsgd_create(m, &c, &o_));
      m_ = m; B_ = B;
    }
    check(nfm_opt_set_it(o_, it));
    std::vector<int64_t> indices(n), chunk((size_t)(B * inner));
    std::iota(indices.begin(), indices.end(), 0);
    int64_t ii = 0;
    if (shuffle) std::shuffle(indices.begin(), indices.end(), sfm.rng());  // :169-170
    if (verbose > 0) {
      std::printf("Minibatch size: %lld\nNumber of inner iteration: %lld\n", (long long)B, (long long)inner);
      std::printf("Epoch   Loss         Regularization\n");
    }
    double oldLossVal = INFINITY;
    bool isConverged = false;
    for (int t = 0; t < maxIter; ++t) {
      for (size_t q = 0; q < chunk.size(); ++q) {  // :98-108: indices[ii], ii wraps and reshuffles
        chunk[q] = indices[ii++];
        if (ii >= n) {
          ii = 0;
          if (shuffle) std::shuffle(indices.begin(), indices.end(), sfm.rng());
        }
      }
      double ls = 0.0, viol = 0.0;
      check(nfm_opt_epoch(o_, X.handle(), chunk.data(), 0, (int64_t)chunk.size(), &ls, &viol));
      it += inner;
      const double runningLoss = ls / (double)(B * inner);  // :122
      if (callback) {
        check(nfm_opt_finalize(o_));
        sfm.pull();
        callback(*this, sfm);
      }
      if (std::isnan(runningLoss)) { std::printf("Loss is NaN. Use smaller learning rate.\n"); break; }
      if (verbose > 0) std::printf("%-5d   %-10.4e\n", t + 1, runningLoss);
      if (std::fabs(oldLossVal - runningLoss) < tol) {  // :201-204
        if (verbose > 0) std::printf("Converged at epoch %d.\n", t + 1);
        isConverged = true;
        break;
      }
      oldLossVal = runningLoss;
    }
    if (!isConverged && verbose > 0) std::printf("Objective did not converge. Increase maxIter.\n");
    check(nfm_opt_finalize(o_));
    sfm.pull();
  }

 private:
  nfm_opt* o_ = nullptr; nfm_model* m_ = nullptr; int64_t B_ = -1;
};

// predictAllWithGrad, optimizer/pgd.nim:70-103: yPred, dL and the gradient of the mean loss at sfm's parameters;
// gradP in the reference's training layout [nOrders][d + nAugments][k]
struct Grads { std::vector<double> P, w; double intercept = 0.0, loss = 0.0; };
template <class L = Squared>
inline Grads predictAllWithGrad(const CSRDataset& X, const std::vector<double>& y, FactorizationMachine& sfm,
                                std::vector<double>& yPred, std::vector<double>& dL, L loss = L()) {
  if (!sfm.isInitialized) throw NotFittedError("Factorization machines is not fitted.");
  if ((int64_t)y.size() != X.nSamples()) throw std::invalid_argument("len(y) != nSamples");
  check(nfm_dataset_set_targets(X.handle(), y.data()));
  nfm_model* m = sfm.push();
  nfm_mbpsgd_cfg c{0.1, 1e-6, 1e-3, 1e-4, 1e-4, 1.0, loss.param, L::id, (int32_t)optimal, NFM_REG_L1, 0, 1};
  nfm_opt* o = nullptr;
  check(nfm_mbpsgd_create(m, &c, &o));
  Grads g;
  g.P.assign(sfm.P.size(), 0.0);
  g.w.assign((size_t)X.nFeatures(), 0.0);
  yPred.assign((size_t)X.nSamples(), 0.0);
  dL.assign((size_t)X.nSamples(), 0.0);
  const int32_t rc = nfm_opt_predict_all_with_grad(o, X.handle(), yPred.data(), dL.data(), g.P.data(), g.w.data(), &g.intercept, &g.loss);
  nfm_opt_destroy(o);
  check(rc);
  g.loss /= (double)std::max<int64_t>(X.nSamples(), 1);
  return g;
}

}  // namespace nimfm
