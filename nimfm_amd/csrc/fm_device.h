// nimfm_amd/csrc/fm_device.h -- device-side building blocks shared by the predict, row-phase and
// column-phase kernels.  Written for gfx950 wave64.
//
// Work decomposition of CSR rows on a wavefront ("lanes <-> latent factors"):
//   L lanes (power of two, Kp = 2L) cover one parameter row P[j][0..Kp) with one 16-byte load each:
//   a row is one contiguous, aligned segment of Kp*8 bytes.  A sample owns SPLIT such row slots
//   (L*SPLIT lanes), slot q handles the row's nnz q, q+SPLIT, ...; a wavefront therefore carries
//   64/(L*SPLIT) samples.  Per-factor sums are finished with xor-shuffles across the SPLIT slots,
//   sums over the factors with xor-shuffles across the L lanes.  SPLIT trades per-sample latency
//   (more slots = shorter serial chain) against fixed per-sample work (loss, step sizes, reductions)
//   being amortised over more samples per wave-instruction.
#pragma once
#include "common.h"

namespace nfm {
namespace dev {

constexpr int kMaxDeg = 6;

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, kWave); }
__device__ __forceinline__ double shfl_xor_d(double v, int mask) { return __shfl_xor(v, mask, kWave); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 1; s < kWave; s <<= 1) v += shfl_xor_d(v, s);
  return v;
}

// Parameter rows the column phase reads and writes ONCE can be marked non-temporal (-DNFM_NT=1) so
// that they do not evict the samples' A rows and records from L2.  Measured: column phase -0.7 us
// (cfg2), -0.5 us (headline shape), -6 us (AdaGrad), but the NEXT row phase pays for rows that no
// longer sit in the Infinity Cache (AdaGrad: +18 us) -- off by default.
#ifndef NFM_NT
#define NFM_NT 0
#endif
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld_stream(const double* p) {
#if NFM_NT
  const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
  return double2{v.x, v.y};
#else
  return *reinterpret_cast<const double2*>(p);
#endif
}
__device__ __forceinline__ void st_stream(double* p, double2 v) {
#if NFM_NT
  v2d_t w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<v2d_t*>(p));
#else
  *reinterpret_cast<double2*>(p) = v;
#endif
}

// loss.nim:15-102 (Huber's sign quirk at :90-93 kept)
__device__ __forceinline__ double loss_value(int loss, double param, double y, double p) {
  switch (loss) {
    case NFM_LOSS_SQUARED: {
      const double r = y - p;
      return 0.5 * (r * r);
    }
    case NFM_LOSS_SQUARED_HINGE: {
      const double z = 1 - p * y;
      const double m = z > 0 ? z : 0;
      return m * m;
    }
    case NFM_LOSS_LOGISTIC: {
      const double z = p * y;
      if (z > 0) return log(1 + exp(-z));
      return log(exp(z) + 1) - z;
    }
    default: {
      const double z = fabs(y - p);
      if (z < param) return 0.5 * (z * z);
      return param * (z - 0.5 * param);
    }
  }
}

__device__ __forceinline__ double loss_grad(int loss, double param, double y, double p) {
  switch (loss) {
    case NFM_LOSS_SQUARED:
      return p - y;
    case NFM_LOSS_SQUARED_HINGE: {
      const double z = 1 - p * y;
      return z > 0 ? -2 * y * z : 0.0;
    }
    case NFM_LOSS_LOGISTIC: {
      const double z = p * y;
      if (z > 0) return -y * exp(-z) / (1 + exp(-z));
      return -y / (exp(z) + 1);
    }
    default: {
      const double z = fabs(y - p);
      return z < param ? y - p : param;
    }
  }
}

// optimizer/sgd.nim:60-69
__device__ __forceinline__ double get_eta(int sched, double eta0, double power, double reg, double it) {
  switch (sched) {
    case NFM_SCHED_CONSTANT:
      return eta0;
    case NFM_SCHED_OPTIMAL: {
      const double base = 1.0 + eta0 * reg * it;
      return eta0 / (power == 1.0 ? base : pow(base, power));
    }
    case NFM_SCHED_INVSCALING:
      return eta0 / (power == 1.0 ? it : pow(it, power));
    default:
      return 1.0 / (reg * it);
  }
}

// SGD mini-batch rule: a coordinate touched c times in a batch receives (sum of the c per-sample steps) / touch_div --
// the sum itself up to `cap` touches, cap / c of it beyond (cap = 1: the mean)
__device__ __forceinline__ double touch_div(double c, double cap) { return c > cap ? c / cap : 1.0; }

// model/fm_base.nim:32-34: classification targets are sign(y)
__device__ __forceinline__ double target_of(double y, int task) {
  if (task == NFM_TASK_CLASSIFICATION) return (double)((y > 0) - (y < 0));
  return y;
}

// Parameter sources: how a kernel obtains the true value of P[off], P[off+1].
struct PlainParams {  // SGD / predict: scale * stored
  const double* P;
  double scale;
  __device__ __forceinline__ double2 load(size_t off) const {
    double2 v = *reinterpret_cast<const double2*>(P + off);
    v.x *= scale;
    v.y *= scale;
    return v;
  }
};

// AdaGrad, mini-batch rule: what a coordinate's g_norm grows by, from the batch's sum of gradients and sum of squared gradients
// (opt_views.h: ada_cross; one touch: acc^2 == accn, nothing changes)
__device__ __forceinline__ double ada_norm_inc(double acc, double accn, double cross) {
  const double c = acc * acc - accn;
  return accn + (cross != 0.0 && c > 0.0 ? cross * c : 0.0);
}
// optimizer/adagrad.nim:96-98: P = -(eta0 * g_sum) / (eta0*(it-1)*beta + sqrt(g_norm))
__device__ __forceinline__ double adagrad_param(double g, double n, double eta0, double tmp) {
  return -(eta0 * g) / (tmp + sqrt(n));
}
struct AdaParams {
  const double* G;
  const double* N;
  double eta0, tmp;
  __device__ __forceinline__ double2 load(size_t off) const {
    const double2 g = *reinterpret_cast<const double2*>(G + off);
    const double2 n = *reinterpret_cast<const double2*>(N + off);
    double2 v;
    v.x = adagrad_param(g.x, n.x, eta0, tmp);
    v.y = adagrad_param(g.y, n.y, eta0, tmp);
    return v;
  }
};

// nnz q of the row that starts at q0 with m stored entries followed by n_aug dummy features
// (index d+t, value 1.0; dataset.nim:182-189).  Past the end: (0, 0.0), which contributes nothing
// to any sum below and points at a valid parameter row.
__device__ __forceinline__ void row_entry(const CsrView& X, int64_t q0, int m, int m_tot, int q, int& j, double& x) {
  j = 0;
  x = 0.0;
  if (q < m) {
    j = X.indices[q0 + q];
    x = X.data[q0 + q];
  } else if (q < m_tot) {
    j = (int)(X.d + (q - m));
    x = 1.0;
  }
}

constexpr int kUnroll = 4;  // rows requested together per lane in the singles update
#ifndef NFM_FWD_UNROLL
#define NFM_FWD_UNROLL 8
#endif
constexpr int kFwdUnroll = NFM_FWD_UNROLL;  // independent parameter-row loads in flight per lane (forward)

// ---- ANOVA forward, degree 2: the sum-of-squares trick (optimizer/sgd.nim:160-170,
// kernels.nim:59-64).  A1 = sum x p, A2 = sum (x p)^2 for this lane's factor pair over the whole
// row (identical in all SPLIT slots of the sample on return).
template <int L, int SPLIT, class PS>
__device__ __forceinline__ void anova_fwd_deg2(const PS& ps, const CsrView& X, int64_t q0, int m, int m_tot,
                                               size_t blk_off, int Kp, int slot, int l, double2& A1, double2& A2) {
  double2 a1 = {0.0, 0.0}, a2 = {0.0, 0.0};
  for (int q = slot; q < m_tot; q += kFwdUnroll * SPLIT) {
    int j[kFwdUnroll];
    double x[kFwdUnroll];
    double2 p[kFwdUnroll];
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) row_entry(X, q0, m, m_tot, q + u * SPLIT, j[u], x[u]);
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) p[u] = ps.load(blk_off + (size_t)j[u] * Kp + 2 * l);
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) {
      const double tx = x[u] * p[u].x, ty = x[u] * p[u].y;
      a1.x += tx;
      a1.y += ty;
      a2.x += tx * tx;
      a2.y += ty * ty;
    }
  }
#pragma unroll
  for (int s = L; s < L * SPLIT; s <<= 1) {
    a1.x += shfl_xor_d(a1.x, s);
    a1.y += shfl_xor_d(a1.y, s);
    a2.x += shfl_xor_d(a2.x, s);
    a2.y += shfl_xor_d(a2.y, s);
  }
  A1 = a1;
  A2 = a2;
}

// partial results over the SPLIT slots' disjoint nnz subsets combine by truncated polynomial multiplication
template <int L, int SPLIT>
__device__ __forceinline__ void combine_slots_degn(double2 (&E)[kMaxDeg + 1], int deg, int lane) {
#pragma unroll
  for (int s = L; s < L * SPLIT; s <<= 1) {
    double2 lo[kMaxDeg + 1], hi[kMaxDeg + 1];
    const bool upper = (lane & s) != 0;
#pragma unroll
    for (int t = 0; t <= kMaxDeg; ++t) {
      double2 o;
      o.x = shfl_xor_d(E[t].x, s);
      o.y = shfl_xor_d(E[t].y, s);
      lo[t] = upper ? o : E[t];  // both partners form the same (lo, hi) pair
      hi[t] = upper ? E[t] : o;
    }
#pragma unroll
    for (int t = 1; t <= kMaxDeg; ++t) {
      double2 acc = {0.0, 0.0};
#pragma unroll
      for (int u = 0; u <= t; ++u) {
        acc.x += lo[u].x * hi[t - u].x;
        acc.y += lo[u].y * hi[t - u].y;
      }
      if (t <= deg) E[t] = acc;
    }
  }
}

// ---- ANOVA forward, degree 3..kMaxDeg: the DP of optimizer/sgd.nim:152-159 (kernels.nim:54-58),
// A[t] += A[t-1] * p * x for t = deg..1 per nnz.  A[t] is the t-th elementary symmetric polynomial
// of {p_j x_j}; partial results over the SPLIT slots' disjoint nnz subsets combine by truncated
// polynomial multiplication.  E[0] = 1.  deg is a run-time value (guards on unrolled loops keep E
// in registers).
template <int L, int SPLIT, class PS>
__device__ __forceinline__ void anova_fwd_degn(const PS& ps, const CsrView& X, int64_t q0, int m, int m_tot,
                                               size_t blk_off, int Kp, int slot, int l, int lane, int deg,
                                               double2 (&E)[kMaxDeg + 1]) {
#pragma unroll
  for (int t = 0; t <= kMaxDeg; ++t) E[t] = {0.0, 0.0};
  E[0] = {1.0, 1.0};
  for (int q = slot; q < m_tot; q += SPLIT) {
    int j;
    double x;
    row_entry(X, q0, m, m_tot, q, j, x);
    const double2 p = ps.load(blk_off + (size_t)j * Kp + 2 * l);
#pragma unroll
    for (int t = kMaxDeg; t >= 1; --t)
      if (t <= deg) {
        E[t].x += E[t - 1].x * p.x * x;
        E[t].y += E[t - 1].y * p.y * x;
      }
  }
  combine_slots_degn<L, SPLIT>(E, deg, lane);
}

// value of a compile-time-unrolled array at a run-time index
__device__ __forceinline__ double2 pick(const double2 (&E)[kMaxDeg + 1], int t) {
  double2 r = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u <= kMaxDeg; ++u)
    if (u == t) r = E[u];
  return r;
}

// sum over this lane's factor pair, then over the L lanes of the row slot
template <int L>
__device__ __forceinline__ double sum_lanes(double r) {
#pragma unroll
  for (int s = 1; s < L; s <<= 1) r += shfl_xor_d(r, s);
  return r;
}

// Derivative of the degree-deg ANOVA kernel w.r.t. p_js (optimizer/sgd.nim:176-188):
//   deg == 2: x (A1 - p x);  deg >= 3: dA = x; for t in 1..<deg: dA = x (A[t] - p dA)
// A holds A[1..deg-1] at A[0..deg-2].
__device__ __forceinline__ double anova_grad(int deg, double x, double p, const double (&A)[kMaxDeg - 1]) {
  if (deg == 2) return x * (A[0] - p * x);
  double dA = x;
#pragma unroll
  for (int t = 1; t < kMaxDeg; ++t)
    if (t < deg) dA = x * (A[t - 1] - p * dA);
  return dA;
}

}  // namespace dev
}  // namespace nfm
