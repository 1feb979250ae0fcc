// nimfm_amd/csrc/dp.hip -- data-parallel groups: transports (RCCL over xGMI between processes; peer-to-peer sums between
// the ranks of one process), the exchange rules of dp.h, and the nfm_dp_* entry points of include/nimfm_hip.h.
#include "dp.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: the library is opened at run time (no link dependency)
#include <string.h>

#include <chrono>
#include <map>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>

#include "opt_views.h"

namespace nfm {

// ------------------------------------------------------------------------------------------------
// RCCL, opened on first use.  torch ships its own librccl.so; whichever is already mapped under that soname is reused,
// otherwise ROCm's.  One communicator per group and rank, one process per GPU.
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};
Rccl& rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    const char* names[] = {getenv("NFM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      x->h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (x->h) break;
      x->err = dlerror();
    }
    if (!x->h) return x;
    x->GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(x->h, "ncclGetUniqueId"));
    x->CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(x->h, "ncclCommInitRank"));
    x->CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(x->h, "ncclCommDestroy"));
    x->AllReduce = reinterpret_cast<decltype(&ncclAllReduce)>(dlsym(x->h, "ncclAllReduce"));
    x->GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(x->h, "ncclGetErrorString"));
    if (!x->GetUniqueId || !x->CommInitRank || !x->CommDestroy || !x->AllReduce) {
      x->err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
      x->h = nullptr;
    }
    return x;
  }();
  return *r;
}
int rccl_ready() {
  Rccl& r = rccl();
  NFM_CHECK(r.h, NFM_ERR_HIP, "RCCL is not available: %s", r.err.c_str());
  return NFM_OK;
}
#define NFM_NCCL_CHECK(expr)                                                                     \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess)                                                                       \
      return set_error(NFM_ERR_HIP, "%s failed: %s", #expr,                                      \
                       rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error");        \
  } while (0)

struct RcclTransport : DpTransport {
  ncclComm_t comm = nullptr;
  ~RcclTransport() override {
    if (comm) (void)rccl().CommDestroy(comm);
  }
  int allreduce(const double* send, double* recv, int64_t n, int op, hipStream_t st) override {
    NFM_NCCL_CHECK(rccl().AllReduce(send, recv, (size_t)n, ncclDouble, op == DP_MAX ? ncclMax : ncclSum, comm, st));
    return NFM_OK;
  }
};

// ------------------------------------------------------------------------------------------------
// the ranks of ONE process (one nfm_ctx each; several GPUs with peer access, or one GPU shared by all ranks -- the
// way the exchange rules are exercised on a one-GPU box).  Every rank's host thread arrives with its buffers; each
// rank then sums all ranks' send buffers in rank order into its own receive buffer: identical bits everywhere.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxLocalWorld = 16;
struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  uint64_t generation = 0;
  const double* send[kMaxLocalWorld] = {};
  int64_t count[kMaxLocalWorld] = {};  // what every rank believes the collective's length and operation to be
  int opcode[kMaxLocalWorld] = {};
  hipEvent_t ready[kMaxLocalWorld] = {}, done[kMaxLocalWorld] = {};
  int device[kMaxLocalWorld] = {};
  bool broken = false;
  ~LocalGroup() {
    for (int r = 0; r < world; ++r) {
      if (ready[r]) (void)hipEventDestroy(ready[r]);
      if (done[r]) (void)hipEventDestroy(done[r]);
    }
  }
  // false: a rank did not arrive within two minutes (it failed or never made the matching call) -- the group is
  // marked broken and every waiting rank returns an error instead of hanging
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t g = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != g || broken; })) {
      broken = true;
      cv.notify_all();
    }
    return !broken;
  }
};
struct PeerPtrs {
  const double* p[kMaxLocalWorld];
};
__global__ void k_sum_peers(PeerPtrs src, int world, double* __restrict__ dst, int64_t n, int op) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = src.p[0][i];
    for (int q = 1; q < world; ++q) {
      const double o = src.p[q][i];
      v = op == DP_MAX ? (o > v ? o : v) : v + o;
    }
    dst[i] = v;
  }
}
struct LocalTransport : DpTransport {
  LocalTransport() { in_process = true; }
  std::shared_ptr<LocalGroup> g;
  int allreduce(const double* send, double* recv, int64_t n, int op, hipStream_t st) override {
    LocalGroup& G = *g;
    NFM_HIP_CHECK(hipEventRecord(G.ready[rank], st));
    G.send[rank] = send;
    G.count[rank] = n;
    G.opcode[rank] = op;
    NFM_CHECK(G.barrier(), NFM_ERR_HIP, "a rank of the local group did not reach the collective");  // buffers published, "ready" recorded
    // the ranks must be in the SAME collective: a rank that left its epoch loop early (or entered another call) would
    // otherwise have its short buffer read at this rank's length
    for (int q = 0; q < world; ++q)
      if (G.count[q] != n || G.opcode[q] != op) {
        {
          std::lock_guard<std::mutex> lk(G.mu);
          G.broken = true;
        }
        G.cv.notify_all();
        return set_error(NFM_ERR_INVALID, "local group: rank %d is in a collective of %lld values (op %d), rank %d in one of %lld (op %d): "
                         "the ranks did not make the same sequence of calls", rank, (long long)n, op, q, (long long)G.count[q], G.opcode[q]);
      }
    PeerPtrs pp{};
    for (int q = 0; q < world; ++q) {
      pp.p[q] = G.send[q];
      if (q != rank) NFM_HIP_CHECK(hipStreamWaitEvent(st, G.ready[q], 0));
    }
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)blocks), dim3(kBlock), 0, st, pp, world, recv, n, op);
    NFM_HIP_CHECK(hipGetLastError());
    NFM_HIP_CHECK(hipEventRecord(G.done[rank], st));
    NFM_CHECK(G.barrier(), NFM_ERR_HIP, "a rank of the local group did not reach the collective");  // every rank has enqueued its sum
    for (int q = 0; q < world; ++q)  // nobody's send buffer is overwritten before all readers are through
      if (q != rank) NFM_HIP_CHECK(hipStreamWaitEvent(st, G.done[q], 0));
    return NFM_OK;
  }
};
}  // namespace

// ------------------------------------------------------------------------------------------------
// exchange kernels (all plain streaming passes over the arena)
// ------------------------------------------------------------------------------------------------
namespace {
inline unsigned grid_stream(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > 256 * 32) b = 256 * 32;
  return (unsigned)(b < 1 ? 1 : b);
}
// Both optimizers exchange INCREMENTS since the last agreed state (`base`).  AdaGrad's g_sum / g_norm are sums over
// samples: the ranks' increments are added up.  SGD combines the ranks' increments by `cw` times their sum:
//   cw = 1 / world  the replicas' MEAN (the default; local SGD).  Always as stable as one rank, but the model moves as far
//                   as ONE rank's steps take it: after the same number of epochs a 4-rank run stands where a single rank
//                   that saw a quarter of the steps stands (tests/test_gpu_dp.py::test_data_parallel_training_..., held-out
//                   RMSE 0.87 against 0.27 for one rank over all samples).
//   cw = 1          the SUM (NFM_DP_SUM): every rank's steps land in the model, as every Hogwild thread's steps land in
//                   the reference's shared one (sgd_multi.nim:83-101) -- the progress of all ranks' steps, but steps that
//                   were computed from the same stale point add up: with features shared by all ranks it acts like a step
//                   size times world (same test: 0.45 with an exchange every 8 mini-batches, divergence with one per
//                   epoch).
// SGD's arena holds STORED values (true / lazy L2 scale); at a mid-epoch sync all ranks have taken the same steps, hence
// have the same scales, and stored increments combine like true ones.  The two scale slots [skip_lo, skip_hi) are never
// exchanged.
// own = x - base
__global__ void k_inc_own(const double* __restrict__ x, const double* __restrict__ base, double* __restrict__ own, int64_t n,
                          int64_t skip_lo, int64_t skip_hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    own[i] = (i < skip_lo || i >= skip_hi) ? x[i] - base[i] : 0.0;
}
// delayed: the other ranks' increments arrive: x += R - own, base += R
__global__ void k_inc_fold(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R,
                           const double* __restrict__ own, int64_t n, int64_t skip_lo, int64_t skip_hi, double cw) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (i < skip_lo || i >= skip_hi) {
      const double r = R[i] * cw;
      x[i] += r - own[i];
      base[i] += r;
    }
}
// a sync point with a delayed exchange pending: the arrival of the previous period's increments and this period's own
// increments in ONE pass over the arena (7 instead of 9 array passes): x += cw R - own, base += cw R, own = x - base
__global__ void k_inc_fold_own(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R,
                               double* __restrict__ own, int64_t n, int64_t skip_lo, int64_t skip_hi, double cw) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < skip_lo || i >= skip_hi) {
      const double r = R[i] * cw;
      const double xn = x[i] + (r - own[i]), bn = base[i] + r;
      x[i] = xn;
      base[i] = bn;
      own[i] = xn - bn;
    } else {
      own[i] = 0.0;
    }
  }
}
// closing: x = base + R (every rank forms the same sum), base = x
__global__ void k_inc_close(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R, int64_t n, double w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double g = base[i] + w * R[i];
    x[i] = g;
    base[i] = g;
  }
}
// SGD, closing.  The ranks' tails may differ in length (shards of unequal size), so their scales do: the increments
// travel as TRUE values, own = scale_rank * (stored - base), and are added to the base under ONE scale all ranks agree
// on (the smallest: the rank that took the most steps).  The arena is left in true values with both scales 1.
// sc = the rank's {scale_P, scale_w} (the arena's slots skip_lo, skip_lo + 1); segments: [0, seg_w) P, [seg_w, seg_sc) w.
__global__ void k_sgd_close_own(const double* __restrict__ x, const double* __restrict__ base, double* __restrict__ own, int64_t n,
                                int64_t seg_w, int64_t seg_sc, int64_t skip_lo, int64_t skip_hi) {
  const double sP = x[skip_lo], sw = x[skip_lo + 1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double sc = i < seg_w ? sP : (i < seg_sc ? sw : 1.0);
    own[i] = (i < skip_lo || i >= skip_hi) ? sc * (x[i] - base[i]) : 0.0;
  }
}
__global__ void k_sgd_close_apply(double* __restrict__ x, const double* __restrict__ base, const double* __restrict__ R,
                                  const double* __restrict__ neg_min_scales, int64_t n, int64_t seg_w, int64_t seg_sc,
                                  int64_t skip_lo, int64_t skip_hi, double cw) {
  const double sP = -neg_min_scales[0], sw = -neg_min_scales[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i >= skip_lo && i < skip_hi) {
      x[i] = 1.0;  // the arena now holds true values
    } else {
      const double sc = i < seg_w ? sP : (i < seg_sc ? sw : 1.0);
      x[i] = sc * base[i] + R[i] * cw;
    }
  }
}
// ---- AdaGrad, NFM_DP_STATE_CROSS (dp.h): the same four steps on (g_sum, g_norm) PAIRS ----
struct CrossPairs {
  int64_t og[3], on[3], len[3];
  int n;
  double gamma;
};
__device__ __forceinline__ bool cross_locate(const CrossPairs& c, int64_t p, int64_t& ig, int64_t& in) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (q < c.n) {
      if (p < c.len[q]) {
        ig = c.og[q] + p;
        in = c.on[q] + p;
        return true;
      }
      p -= c.len[q];
    }
  }
  return false;
}
// own: what this rank SENDS -- dG, and dN - gamma dG^2
__global__ void k_cross_own(const double* __restrict__ x, const double* __restrict__ base, double* __restrict__ own, int64_t np, CrossPairs c) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t ig, in;
    if (!cross_locate(c, p, ig, in)) continue;
    const double dg = x[ig] - base[ig], dn = x[in] - base[in];
    own[ig] = dg;
    own[in] = dn - c.gamma * (dg * dg);
  }
}
// what all ranks agree on from the reduced vector R: g_sum += R_G; g_norm += max(R_N + gamma R_G^2, 0)
__device__ __forceinline__ void cross_combined(const CrossPairs& c, double Rg, double Rn, double& cg, double& cn) {
  cg = Rg;
  const double v = Rn + c.gamma * (Rg * Rg);
  cn = v > 0.0 ? v : 0.0;
}
// delayed arrival: x += combined - (this rank's TRUE increments of that period), base += combined
__global__ void k_cross_fold(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R, const double* __restrict__ own,
                             int64_t np, CrossPairs c) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t ig, in;
    if (!cross_locate(c, p, ig, in)) continue;
    double cg, cn;
    cross_combined(c, R[ig], R[in], cg, cn);
    const double og = own[ig], on_true = own[in] + c.gamma * (og * og);
    x[ig] += cg - og;
    base[ig] += cg;
    x[in] += cn - on_true;
    base[in] += cn;
  }
}
// arrival of the previous period and this period's own increments in one pass
__global__ void k_cross_fold_own(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R, double* __restrict__ own,
                                 int64_t np, CrossPairs c) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t ig, in;
    if (!cross_locate(c, p, ig, in)) continue;
    double cg, cn;
    cross_combined(c, R[ig], R[in], cg, cn);
    const double og = own[ig], on_true = own[in] + c.gamma * (og * og);
    const double xg = x[ig] + (cg - og), bg = base[ig] + cg, xn = x[in] + (cn - on_true), bn = base[in] + cn;
    x[ig] = xg;
    base[ig] = bg;
    x[in] = xn;
    base[in] = bn;
    const double dg = xg - bg;
    own[ig] = dg;
    own[in] = (xn - bn) - c.gamma * (dg * dg);
  }
}
// closing: x = base + combined, base = x
__global__ void k_cross_close(double* __restrict__ x, double* __restrict__ base, const double* __restrict__ R, int64_t np, CrossPairs c) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t ig, in;
    if (!cross_locate(c, p, ig, in)) continue;
    double cg, cn;
    cross_combined(c, R[ig], R[in], cg, cn);
    const double g = base[ig] + cg, nn = base[in] + cn;
    x[ig] = g;
    base[ig] = g;
    x[in] = nn;
    base[in] = nn;
  }
}
static CrossPairs cross_of(const DpEpoch& e, int64_t* np) {
  CrossPairs c{};
  c.n = e.n_pairs;
  c.gamma = e.cross_gamma;
  *np = 0;
  for (int q = 0; q < e.n_pairs; ++q) {
    c.og[q] = e.pair[q][0];
    c.on[q] = e.pair[q][1];
    c.len[q] = e.pair[q][2];
    *np += e.pair[q][2];
  }
  return c;
}
__global__ void k_neg_scales(const double* __restrict__ x, int64_t skip_lo, double* __restrict__ out) {
  out[0] = -x[skip_lo];
  out[1] = -x[skip_lo + 1];
  out[2] = 0.0;
  out[3] = 0.0;
}
}  // namespace

int dp_fold_pending(DpEpoch& e) {
  nfm_dp* dp = e.dp;
  if (!dp->pending) return NFM_OK;
  hipStream_t st = dp->ctx->stream;
  NFM_HIP_CHECK(hipStreamWaitEvent(st, dp->ev_done, 0));
  const int64_t n = dp->pending_n;
  if (e.n_pairs > 0) {
    int64_t np = 0;
    const CrossPairs c = cross_of(e, &np);
    hipLaunchKernelGGL(k_cross_fold, dim3(grid_stream(np)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(),
                       dp->snap.as<double>(), np, c);
  } else
  hipLaunchKernelGGL(k_inc_fold, dim3(grid_stream(n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(),
                     dp->snap.as<double>(), n, e.skip_lo, e.skip_hi, e.combine_w);
  NFM_HIP_CHECK(hipGetLastError());
  dp->pending = false;
  return NFM_OK;
}

int dp_epoch_begin(DpEpoch& e, int64_t n_full_batches, int64_t n_batches) {
  nfm_dp* dp = e.dp;
  hipStream_t st = dp->ctx->stream;
  NFM_CHECK(!dp->pending, NFM_ERR_INVALID, "a data-parallel exchange of an earlier call is still pending");
  NFM_TRY(dp->snap.ensure(sizeof(double) * (size_t)e.n));
  if (e.n_pairs > 0) NFM_HIP_CHECK(hipMemsetAsync(dp->snap.p, 0, sizeof(double) * (size_t)e.n, st));  // (the pair kernels skip the spans' padding)
  NFM_TRY(dp->recv.ensure(sizeof(double) * (size_t)e.n));
  NFM_TRY(dp->scal.ensure(sizeof(double) * 8));
  // the state all ranks agree on (they enter the call with identical replicas; SGD: in true values, both scales 1)
  NFM_TRY(dp->base.ensure(sizeof(double) * (size_t)e.n));
  NFM_HIP_CHECK(hipMemcpyAsync(dp->base.p, e.arena, sizeof(double) * (size_t)e.n, hipMemcpyDeviceToDevice, st));
  // mid-epoch sync points lie after mini-batches S, 2S, ... that are FULL on every rank (identical step counters and
  // L2 scales there) and strictly before this rank's -- hence every rank's -- last one
  int64_t mine = 0;
  if (e.sync_period > 0) {
    mine = n_full_batches / e.sync_period;
    if (mine * e.sync_period >= n_batches) mine = (n_batches - 1) / e.sync_period;
    if (mine < 0) mine = 0;
  }
  double h[4] = {-(double)mine, 0.0, 0.0, 0.0};
  NFM_HIP_CHECK(hipMemcpyAsync(dp->scal.p, h, sizeof(h), hipMemcpyHostToDevice, st));
  NFM_TRY(dp->t->allreduce(dp->scal.as<double>(), dp->scal.as<double>() + 4, 4, DP_MAX, st));
  NFM_HIP_CHECK(hipMemcpyAsync(h, dp->scal.as<double>() + 4, sizeof(h), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  e.n_sync = (int64_t)(-h[0]);
  return NFM_OK;
}

int dp_after_batch(DpEpoch& e, int64_t b) {
  if (e.sync_period <= 0 || (b + 1) % e.sync_period != 0) return NFM_OK;
  const int64_t k = (b + 1) / e.sync_period;  // 1-based sync point
  if (k > e.n_sync) return NFM_OK;
  nfm_dp* dp = e.dp;
  hipStream_t st = dp->ctx->stream;
  int64_t np_x = 0;
  const CrossPairs cx = cross_of(e, &np_x);
  if (dp->pending) {
    // the previous period's collective has had a whole period to finish: its result arrives and this period's increments
    // are formed in one pass
    NFM_HIP_CHECK(hipStreamWaitEvent(st, dp->ev_done, 0));
    if (e.n_pairs > 0)
      hipLaunchKernelGGL(k_cross_fold_own, dim3(grid_stream(np_x)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(),
                         dp->snap.as<double>(), np_x, cx);
    else
    hipLaunchKernelGGL(k_inc_fold_own, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(),
                       dp->snap.as<double>(), e.n, e.skip_lo, e.skip_hi, e.combine_w);
    dp->pending = false;
  } else if (e.n_pairs > 0) {
    hipLaunchKernelGGL(k_cross_own, dim3(grid_stream(np_x)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->snap.as<double>(), np_x, cx);
  } else {
    hipLaunchKernelGGL(k_inc_own, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->snap.as<double>(), e.n,
                       e.skip_lo, e.skip_hi);
  }
  NFM_HIP_CHECK(hipGetLastError());
  hipStream_t cs = e.overlap ? dp->comm : st;
  if (e.overlap) {
    NFM_HIP_CHECK(hipEventRecord(dp->ev_ready, st));
    NFM_HIP_CHECK(hipStreamWaitEvent(cs, dp->ev_ready, 0));
  }
  NFM_TRY(dp->t->allreduce(dp->snap.as<double>(), dp->recv.as<double>(), e.n, DP_SUM, cs));
  NFM_HIP_CHECK(hipEventRecord(dp->ev_done, cs));
  dp->pending = true;
  dp->pending_n = e.n;
  dp->n_collectives++;
  dp->bytes += (int64_t)sizeof(double) * e.n;
  if (!e.overlap) NFM_TRY(dp_fold_pending(e));
  return NFM_OK;
}

int dp_epoch_end(DpEpoch& e, double* sums_dev) {
  nfm_dp* dp = e.dp;
  hipStream_t st = dp->ctx->stream;
  NFM_TRY(dp_fold_pending(e));
  if (e.opt_kind == OPT_SGD) {
    hipLaunchKernelGGL(k_sgd_close_own, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->snap.as<double>(),
                       e.n, e.seg_w, e.seg_sc, e.skip_lo, e.skip_hi);
    hipLaunchKernelGGL(k_neg_scales, dim3(1), dim3(1), 0, st, e.arena, e.skip_lo, dp->scal.as<double>());
    NFM_TRY(dp->t->allreduce(dp->scal.as<double>(), dp->scal.as<double>() + 4, 4, DP_MAX, st));
    NFM_TRY(dp->t->allreduce(dp->snap.as<double>(), dp->recv.as<double>(), e.n, DP_SUM, st));
    hipLaunchKernelGGL(k_sgd_close_apply, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(),
                       dp->scal.as<double>() + 4, e.n, e.seg_w, e.seg_sc, e.skip_lo, e.skip_hi, e.combine_w);
  } else if (e.n_pairs > 0) {
    int64_t np = 0;
    const CrossPairs c = cross_of(e, &np);
    hipLaunchKernelGGL(k_cross_own, dim3(grid_stream(np)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->snap.as<double>(), np, c);
    NFM_TRY(dp->t->allreduce(dp->snap.as<double>(), dp->recv.as<double>(), e.n, DP_SUM, st));
    hipLaunchKernelGGL(k_cross_close, dim3(grid_stream(np)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(), np, c);
  } else {
    hipLaunchKernelGGL(k_inc_own, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->snap.as<double>(), e.n,
                       (int64_t)0, (int64_t)0);
    NFM_TRY(dp->t->allreduce(dp->snap.as<double>(), dp->recv.as<double>(), e.n, DP_SUM, st));
    hipLaunchKernelGGL(k_inc_close, dim3(grid_stream(e.n)), dim3(kBlock), 0, st, e.arena, dp->base.as<double>(), dp->recv.as<double>(), e.n, e.combine_w);
  }
  NFM_HIP_CHECK(hipGetLastError());
  dp->n_collectives++;
  dp->bytes += (int64_t)sizeof(double) * e.n;
  // the running sums the reference adds up over its threads (sgd_multi.nim:98-101) and the samples all ranks saw
  NFM_TRY(dp->t->allreduce(sums_dev, dp->scal.as<double>(), 3, DP_SUM, st));
  NFM_HIP_CHECK(hipMemcpyAsync(sums_dev, dp->scal.p, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  return NFM_OK;
}

}  // namespace nfm

namespace nfm {
// live groups by uid (an optimizer keeps a raw nfm_dp*: it is checked against this table before every use)
static std::mutex g_dp_mu;
static std::map<uint64_t, const nfm_dp*> g_dps;
static uint64_t dp_register(const nfm_dp* dp) {
  static uint64_t next = 0;
  std::lock_guard<std::mutex> lk(g_dp_mu);
  g_dps[++next] = dp;
  return next;
}
bool dp_is_live(const nfm_dp* dp, uint64_t uid) {
  std::lock_guard<std::mutex> lk(g_dp_mu);
  auto it = g_dps.find(uid);
  return it != g_dps.end() && it->second == dp;
}
}  // namespace nfm
nfm_dp::~nfm_dp() {
  {
    std::lock_guard<std::mutex> lk(nfm::g_dp_mu);
    nfm::g_dps.erase(uid);
  }
  if (ctx) (void)hipSetDevice(ctx->device);
  if (comm) {
    (void)hipStreamSynchronize(comm);
    (void)hipStreamDestroy(comm);
  }
  if (ev_ready) (void)hipEventDestroy(ev_ready);
  if (ev_done) (void)hipEventDestroy(ev_done);
  delete t;
}
using namespace nfm;

static int dp_common_init(nfm_ctx* ctx, nfm_dp* dp) {
  dp->ctx = ctx;
  dp->uid = nfm::dp_register(dp);
  NFM_HIP_CHECK(hipSetDevice(ctx->device));
  NFM_HIP_CHECK(hipStreamCreateWithFlags(&dp->comm, hipStreamNonBlocking));
  NFM_HIP_CHECK(hipEventCreateWithFlags(&dp->ev_ready, hipEventDisableTiming));
  NFM_HIP_CHECK(hipEventCreateWithFlags(&dp->ev_done, hipEventDisableTiming));
  return NFM_OK;
}

extern "C" {

int32_t nfm_dp_unique_id(void* id) {
  NFM_CHECK(id, NFM_ERR_INVALID, "null id");
  NFM_TRY(rccl_ready());
  static_assert(sizeof(ncclUniqueId) == NFM_DP_ID_BYTES, "id size");
  ncclUniqueId u;
  NFM_NCCL_CHECK(rccl().GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return NFM_OK;
}

int32_t nfm_dp_create(nfm_ctx* ctx, const void* id, int32_t rank, int32_t world, nfm_dp** out) {
  NFM_CHECK(ctx && id && out, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(world >= 1 && rank >= 0 && rank < world, NFM_ERR_INVALID, "bad rank %d of %d", rank, world);
  NFM_TRY(rccl_ready());
  std::unique_ptr<nfm_dp> dp(new nfm_dp());
  NFM_TRY(dp_common_init(ctx, dp.get()));
  std::unique_ptr<RcclTransport> t(new RcclTransport());
  t->rank = rank;
  t->world = world;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  NFM_NCCL_CHECK(rccl().CommInitRank(&t->comm, world, u, rank));
  dp->t = t.release();
  *out = dp.release();
  return NFM_OK;
}

int32_t nfm_dp_create_local(nfm_ctx* const* ctxs, int32_t world, nfm_dp** out) {
  NFM_CHECK(ctxs && out, NFM_ERR_INVALID, "null argument");
  NFM_CHECK(world >= 1 && world <= kMaxLocalWorld, NFM_ERR_INVALID, "world must be in [1,%d]", kMaxLocalWorld);
  auto g = std::make_shared<LocalGroup>();
  g->world = world;
  for (int r = 0; r < world; ++r) {
    NFM_CHECK(ctxs[r], NFM_ERR_INVALID, "null context %d", r);
    g->device[r] = ctxs[r]->device;
    NFM_HIP_CHECK(hipSetDevice(ctxs[r]->device));
    NFM_HIP_CHECK(hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming));
    NFM_HIP_CHECK(hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming));
    for (int q = 0; q < world; ++q)
      if (ctxs[q]->device != ctxs[r]->device) {
        const hipError_t e = hipDeviceEnablePeerAccess(ctxs[q]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return set_error(NFM_ERR_HIP, "no peer access from device %d to %d: %s", ctxs[r]->device, ctxs[q]->device, hipGetErrorString(e));
        (void)hipGetLastError();
      }
  }
  std::vector<std::unique_ptr<nfm_dp>> made;
  for (int r = 0; r < world; ++r) {
    std::unique_ptr<nfm_dp> dp(new nfm_dp());
    NFM_TRY(dp_common_init(ctxs[r], dp.get()));
    LocalTransport* t = new LocalTransport();
    t->rank = r;
    t->world = world;
    t->g = g;
    dp->t = t;
    made.push_back(std::move(dp));
  }
  for (int r = 0; r < world; ++r) out[r] = made[r].release();
  return NFM_OK;
}

int32_t nfm_dp_info(const nfm_dp* dp, int32_t* rank, int32_t* world, int64_t* n_collectives, int64_t* bytes) {
  NFM_CHECK(dp, NFM_ERR_INVALID, "null group");
  if (rank) *rank = dp->t->rank;
  if (world) *world = dp->t->world;
  if (n_collectives) *n_collectives = dp->n_collectives;
  if (bytes) *bytes = dp->bytes;
  return NFM_OK;
}

int32_t nfm_dp_destroy(nfm_dp* dp) {
  if (!dp) return NFM_OK;
  (void)hipSetDevice(dp->ctx->device);
  (void)hipStreamSynchronize(dp->ctx->stream);
  delete dp;
  return NFM_OK;
}

}  // extern "C"
