// nimfm_amd/csrc/psgd.hip -- the step + proximal stage of MBPSGD; the batch gradient itself
// runs through the mini-batch row / column phase (mb_fm.hip, OPT_PSGD).
#include "fm_device.h"
#include "mb.h"

namespace nfm {
// ------------------------------------------------------------------------------------------------
// MBPSGD (optimizer/minibatch_psgd.nim:87-122): after the column phase has added
// -eta * (batch gradient) to the touched rows, EVERY parameter shrinks by 1 / (1 + eta * reg)
// (Params.step = add then scale, model/params.nim:60-65,90-98) and every order goes through the
// regulariser's proximal operator with lam = gamma * eta_P / (1 + eta_P * beta) (:118-120).  The solver is
// dense by construction: one pass over P per mini-batch is its roofline (16 B per parameter).
//   L1 (l1.nim:35-39), L21 (l21.nim:23-34), row-wise SquaredL12 (squaredl12.nim:161-162): fused into the pass.
//   column-wise SquaredL12 (the default, :150-159) and SquaredL21 (squaredl21.nim:46-54) couple a whole
//   column / all row norms through one threshold tau = 2 lam S, S = sum_{|p_i| > tau} |p_i| / (1 + 2 lam theta)
//   (squaredl12.nim:16-69 finds theta by randomised pivoting).  Here tau is the fixed point of
//   tau <- 2 lam sum_{|p_i| > tau} |p_i| / (1 + 2 lam #{|p_i| > tau}) started at 0: the map is the Newton step
//   of a concave increasing piecewise-linear function, so the active set only shrinks and the iteration ends
//   after finitely many passes at the same theta the pivoting finds (no random numbers, fixed summation order).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double softthreshold(double x, double alpha) {  // regularizer/utils.nim:4-5
  const double t = fmax(fabs(x) - alpha, 0.0);
  return x > 0 ? t : (x < 0 ? -t : 0.0 * t);
}

// Sum over the wavefront with the first four levels on DPP (quad permutes and row mirrors: no LDS-pipe traffic) and
// the last two on ds_bpermute.  Every lane adds the same two group sums at every level, so all lanes end with the
// same bits.  (16 wavefronts per workgroup run two of these per threshold pass.)
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_d<0x141>(v);  // row_half_mirror
  v += dpp_d<0x140>(v);  // row_mirror
  v += dev::shfl_xor_d(v, 16);
  v += dev::shfl_xor_d(v, 32);
  return v;
}
constexpr int kProxBlock = 1024;  // threads of the one-workgroup-per-column kernels
// total of lanes 0..NW-1 (NW = 4 or 16 lanes of the first DPP row), handed to every lane of the wavefront
template <int NW = 16>
__device__ __forceinline__ double row16_total(double v) {
  static_assert(NW == 4 || NW == 16, "wavefront sums of a workgroup of 4 or 16 wavefronts");
  v += dpp_d<0xB1>(v);
  v += dpp_d<0x4E>(v);
  if (NW == 16) {
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
  }
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

struct ProxArgs {
  ModelView M;
  OptView O;
  const double* it0p;
  double it_b;
  double* norms;  // [nb][da]   (SquaredL21)
  double* tau;    // [nb][Kp]   thresholds of the coupled operators
};

__device__ __forceinline__ double psgd_lam(const OptView& O, double etaP) { return O.gamma * etaP / (1.0 + etaP * O.beta); }

// sum over the L lanes of one row (fixed xor tree)
template <int L>
__device__ __forceinline__ double row_sum(double v) {
#pragma unroll
  for (int s = 1; s < L; s <<= 1) v += dev::shfl_xor_d(v, s);
  return v;
}

template <int L>
__global__ __launch_bounds__(kBlock) void k_psgd_dense(ProxArgs a) {
  constexpr int R = kWave / L;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t rows = (int64_t)M.nb * M.da;
  const int64_t r = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  const double it = a.it0p[0] + a.it_b;
  const double etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
  const double invP = 1.0 / (1.0 + etaP * O.beta), lam = psgd_lam(O, etaP);
  const bool act = r < rows;  // inactive lanes keep taking part in the shuffles
  double2 p = {0.0, 0.0};
  const size_t e = (size_t)(act ? r : 0) * M.Kp + 2 * l;
  if (act) p = *reinterpret_cast<const double2*>(M.P + e);
  p.x *= invP;
  p.y *= invP;
  if (O.reg == NFM_REG_L1) {
    p.x = softthreshold(p.x, lam);
    p.y = softthreshold(p.y, lam);
  } else if (O.reg == NFM_REG_L21 || O.reg == NFM_REG_SQUAREDL21) {
    const double nrm = sqrt(row_sum<L>(p.x * p.x + p.y * p.y));
    if (O.reg == NFM_REG_L21) {
      const double f = nrm > lam ? 1.0 - lam / nrm : 0.0;
      p.x = nrm > lam ? p.x * f : 0.0;
      p.y = nrm > lam ? p.y * f : 0.0;
    } else if (act && l == 0) {
      a.norms[r] = nrm;
    }
  } else if (O.reg == NFM_REG_SQUAREDL12 && !O.reg_transpose) {
    // the vector operator on the row's k components
    const double ax = fabs(p.x), ay = fabs(p.y);
    double tau = 0.0;
    int cnt_prev = -1;
    for (int pass = 0; pass < 2 * L + 2; ++pass) {
      const double S = row_sum<L>((ax > tau ? ax : 0.0) + (ay > tau ? ay : 0.0));
      const int c = (int)row_sum<L>((double)((ax > tau) + (ay > tau)));
      if (c == cnt_prev || c == 0) break;  // uniform over the row's lanes; rows of one wavefront may differ:
      cnt_prev = c;                        // a finished row keeps its tau (the map is idempotent at the fixed point)
      tau = 2 * lam * (S / (1.0 + 2.0 * lam * (double)c));
    }
    p.x = softthreshold(p.x, tau);
    p.y = softthreshold(p.y, tau);
  }
  if (act) *reinterpret_cast<double2*>(M.P + e) = p;
}

// linear term and intercept: scale only (model/params.nim:60-65)
__global__ __launch_bounds__(kBlock) void k_psgd_linear(ProxArgs a) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const double it = a.it0p[0] + a.it_b;
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (M.fit_linear && j < M.d) M.w[j] *= 1.0 / (1.0 + dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it) * O.alpha);
  if (M.fit_intercept && j == 0)
    M.sc[SC_INTERCEPT] *= 1.0 / (1.0 + dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it) * O.alpha0);
}

// threshold of one coupled vector: v(i) = |x[i * stride]|, i < n.  One workgroup, fixed-order sums.
__device__ __forceinline__ double prox_threshold(const double* __restrict__ x, int64_t n, int64_t stride, double lam,
                                                 double tau0 = 0.0, double cnt0 = -1.0) {
  __shared__ double sS[2][kProxBlock / kWave];
  __shared__ double sC[2][kProxBlock / kWave];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  double tau = tau0, cnt_prev = cnt0;
  for (int pass = 0;; ++pass) {
    double S = 0.0;
    int cw = 0;
    for (int64_t i0 = 0; i0 < n; i0 += kProxBlock) {  // uniform trip count: the ballot needs every lane
      const int64_t i = i0 + threadIdx.x;
      const double v = i < n ? fabs(x[i * stride]) : 0.0;
      const bool on = i < n && v > tau;
      cw += __popcll(__ballot(on));
      if (on) S += v;
    }
    S = wave_sum_dpp(S);
    const double c = (double)cw;
    // one barrier per pass: the per-wavefront sums alternate between two buffers, every thread adds them up in
    // the same fixed order (a writer of pass p + 2 has passed the barrier of pass p + 1, after all reads of pass p)
    double* bS = sS[pass & 1];
    double* bC = sC[pass & 1];
    if (lane == 0) {
      bS[wv] = S;
      bC[wv] = c;
    }
    __syncthreads();
    // lanes 0..15 pick up the 16 wavefronts' sums and add them on DPP (one row of 16 lanes); every wavefront runs
    // the same tree on the same numbers, so all threads hold the same bits
    const double St = row16_total(lane < kProxBlock / kWave ? bS[lane] : 0.0);
    const double ct = row16_total(lane < kProxBlock / kWave ? bC[lane] : 0.0);
    if (ct == cnt_prev || ct == 0.0) break;
    tau = 2 * lam * (St / (1.0 + 2.0 * lam * ct));
    cnt_prev = ct;
  }
  return tau;
}

// column-wise SquaredL12 (squaredl12.nim:150-159): workgroup (s, o) owns component s of order o
__global__ __launch_bounds__(kProxBlock) void k_psgd_prox_columns(ProxArgs a) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int s = blockIdx.x, o = blockIdx.y;
  const double lam = psgd_lam(O, dev::get_eta(O.sched, O.eta0, O.power, O.beta, a.it0p[0] + a.it_b));
  double* col = M.P + (size_t)o * M.da * M.Kp + s;
  const double tau = prox_threshold(col, M.da, M.Kp, lam);
  for (int64_t j = threadIdx.x; j < M.da; j += kProxBlock) col[j * M.Kp] = softthreshold(col[j * M.Kp], tau);
}

// The same for models of at most VPT * 1024 features, and the whole step of such a model in ONE launch: the
// workgroup reads its column once (VPT values per thread, in registers), applies the shrink 1 / (1 + eta_P beta)
// of Params.step itself (the padding components s >= k are zero and need none), runs the threshold passes on
// the registers -- a pass is one workgroup reduction, no memory traffic -- and writes the column once.  The
// workgroups s >= k of order 0 shrink the linear term and the intercept (k_psgd_linear's work).
template <int VPT, int TB>
__global__ __launch_bounds__(TB) void k_psgd_step_columns(ProxArgs a) {
  __shared__ double sS[2][TB / kWave];
  __shared__ double sC[2][TB / kWave];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int s = blockIdx.x, o = blockIdx.y;
  const double it = a.it0p[0] + a.it_b;
  if (s >= M.k) {
    if (o != 0) return;
    const int64_t j = (int64_t)(s - M.k) * TB + threadIdx.x;
    if (M.fit_linear && j < M.d) M.w[j] *= 1.0 / (1.0 + dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it) * O.alpha);
    if (M.fit_intercept && j == 0)
      M.sc[SC_INTERCEPT] *= 1.0 / (1.0 + dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it) * O.alpha0);
    return;
  }
  const double etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
  const double invP = 1.0 / (1.0 + etaP * O.beta), lam = psgd_lam(O, etaP);
  double* col = M.P + (size_t)o * M.da * M.Kp + s;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  double v[VPT];
#pragma unroll
  for (int q = 0; q < VPT; ++q) {
    const int64_t j = (int64_t)q * TB + threadIdx.x;
    v[q] = j < M.da ? col[j * M.Kp] * invP : 0.0;
  }
  double tau = 0.0, cnt_prev = -1.0;
  for (int pass = 0;; ++pass) {
    double S = 0.0;
    int cw = 0;  // the wavefront's active count comes from ballots: no second reduction tree
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const double av = fabs(v[q]);
      const bool on = av > tau;
      cw += __popcll(__ballot(on));
      if (on) S += av;
    }
    S = wave_sum_dpp(S);
    const double c = (double)cw;
    // one barrier per pass: the per-wavefront sums alternate between two buffers, every thread adds them up in
    // the same fixed order (a writer of pass p + 2 has passed the barrier of pass p + 1, after all reads of pass p)
    double* bS = sS[pass & 1];
    double* bC = sC[pass & 1];
    if (lane == 0) {
      bS[wv] = S;
      bC[wv] = c;
    }
    __syncthreads();
    // lanes 0..15 pick up the 16 wavefronts' sums and add them on DPP (one row of 16 lanes); every wavefront runs
    // the same tree on the same numbers, so all threads hold the same bits
    const double St = row16_total<TB / kWave>(lane < TB / kWave ? bS[lane] : 0.0);
    const double ct = row16_total<TB / kWave>(lane < TB / kWave ? bC[lane] : 0.0);
    if (ct == cnt_prev || ct == 0.0) break;
    tau = 2 * lam * (St / (1.0 + 2.0 * lam * ct));
    cnt_prev = ct;
  }
#pragma unroll
  for (int q = 0; q < VPT; ++q) {
    const int64_t j = (int64_t)q * TB + threadIdx.x;
    if (j < M.da) col[j * M.Kp] = softthreshold(v[q], tau);
  }
}

// SquaredL21 (squaredl21.nim:46-54): the vector operator on the row norms of order o ...
__global__ __launch_bounds__(kProxBlock) void k_psgd_prox_norms(ProxArgs a) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int o = blockIdx.x;
  const double lam = psgd_lam(O, dev::get_eta(O.sched, O.eta0, O.power, O.beta, a.it0p[0] + a.it_b));
  const double tau = prox_threshold(a.norms + (size_t)o * M.da, M.da, 1, lam);
  if (threadIdx.x == 0) a.tau[o] = tau;
}

// ... then every row is rescaled from its old norm to the thresholded one: P[i] /= n; P[i] *= n'
template <int L>
__global__ __launch_bounds__(kBlock) void k_psgd_rescale_rows(ProxArgs a) {
  constexpr int R = kWave / L;
  const ModelView& M = a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t r = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  if (r >= (int64_t)M.nb * M.da) return;
  const double n_old = a.norms[r], n_new = softthreshold(n_old, a.tau[r / M.da]);
  const size_t e = (size_t)r * M.Kp + 2 * l;
  double2 p = *reinterpret_cast<const double2*>(M.P + e);
  if (n_old != 0) {
    p.x /= n_old;
    p.y /= n_old;
  }
  p.x *= n_new;
  p.y *= n_new;
  *reinterpret_cast<double2*>(M.P + e) = p;
}


// ---- column-wise SquaredL12 for models too large for k_psgd_step_columns: the threshold passes run ROW-parallel ----
// One workgroup per column walks da values a 256-byte stride apart per pass (637 us per mini-batch at d = 1e5,
// k = 16).  Here a pass is two launches over the whole matrix with coalesced reads (L lanes per row as everywhere):
// k_prox_pass_partial leaves per-workgroup sums {S, count} per component, k_prox_pass_combine adds them in workgroup
// order and advances every component's threshold.  A dependent launch costs ~2 us, so kPasses passes are enqueued
// blindly; finished components are frozen, a finished matrix makes the remaining launches return at once.  Components
// still open after kPasses (never observed: the pass count grows with lam, 3 at 1e-6, 10 at 1) are finished by the
// one-workgroup-per-column loop, so the result never depends on kPasses.
constexpr int kPasses = 10;
constexpr int kPassBlocks = 1024;

struct PassArgs {
  ProxArgs a;
  double* tau;    // [nb][Kp]
  double* cntp;   // [nb][Kp] active count of the previous pass
  int* done;      // [nb][Kp]
  int* ndone;     // components finished (of nb * Kp)
  double* partial;  // [nb][G][4][L]
  int G;
};

__global__ void k_prox_init(PassArgs p) {
  const int C = p.a.M.nb * p.a.M.Kp;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    p.tau[c] = 0.0;
    p.cntp[c] = -1.0;
    p.done[c] = 0;
  }
  if (threadIdx.x == 0) *p.ndone = 0;
}

template <int L>
__global__ __launch_bounds__(kBlock) void k_prox_pass_partial(PassArgs p) {
  constexpr int R = kWave / L;
  __shared__ double red[4][kBlock];
  const ModelView& M = p.a.M;
  if (*p.ndone >= M.nb * M.Kp) return;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L, o = blockIdx.y;
  const double tx = p.tau[o * M.Kp + 2 * l], ty = p.tau[o * M.Kp + 2 * l + 1];
  double Sx = 0.0, Sy = 0.0, cx = 0.0, cy = 0.0;
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * R;
  const double* base = M.P + (size_t)o * M.da * M.Kp + 2 * l;
  for (int64_t r = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g; r < M.da; r += stride) {
    const double2 v = *reinterpret_cast<const double2*>(base + (size_t)r * M.Kp);
    const double ax = fabs(v.x), ay = fabs(v.y);
    if (ax > tx) { Sx += ax; cx += 1.0; }
    if (ay > ty) { Sy += ay; cy += 1.0; }
  }
  red[0][threadIdx.x] = Sx;
  red[1][threadIdx.x] = Sy;
  red[2][threadIdx.x] = cx;
  red[3][threadIdx.x] = cy;
  __syncthreads();
  // thread (c, t), t < L: the workgroup's lane groups hold factor pair t at threads q * L + t, added in q order
  for (int u = threadIdx.x; u < 4 * L; u += kBlock) {
    const int c = u / L, t = u % L;
    double acc = 0.0;
    for (int q = 0; q < kBlock / L; ++q) acc += red[c][q * L + t];
    p.partial[(((size_t)o * p.G + blockIdx.x) * 4 + c) * L + t] = acc;
  }
}

// workgroup (t, o): factor pair t of order o -- adds the G partial sums in workgroup order (fixed tree)
template <int L>
__global__ __launch_bounds__(kBlock) void k_prox_pass_combine(PassArgs p) {
  __shared__ double red[4][kBlock];
  const ModelView& M = p.a.M;
  const OptView& O = p.a.O;
  if (*p.ndone >= M.nb * M.Kp) return;
  const int t = blockIdx.x, o = blockIdx.y;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < p.G; b += kBlock)
    for (int c = 0; c < 4; ++c) acc[c] += p.partial[(((size_t)o * p.G + b) * 4 + c) * L + t];
  for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 2) {  // component 2t (x) and 2t + 1 (y)
    const int comp = o * M.Kp + 2 * t + threadIdx.x;
    if (!p.done[comp]) {
      const double S = red[threadIdx.x][0], ct = red[2 + threadIdx.x][0];
      const double lam = psgd_lam(O, dev::get_eta(O.sched, O.eta0, O.power, O.beta, p.a.it0p[0] + p.a.it_b));
      if (ct == p.cntp[comp] || ct == 0.0) {
        p.done[comp] = 1;
        atomicAdd(p.ndone, 1);
      } else {
        p.tau[comp] = 2 * lam * (S / (1.0 + 2.0 * lam * ct));
        p.cntp[comp] = ct;
      }
    }
  }
}

// components the blind passes left open: the one-workgroup loop, resumed from their state
__global__ __launch_bounds__(kProxBlock) void k_prox_finish(PassArgs p) {
  const ModelView& M = p.a.M;
  const OptView& O = p.a.O;
  const int s = blockIdx.x, o = blockIdx.y, comp = o * M.Kp + s;
  if (p.done[comp]) return;
  const double lam = psgd_lam(O, dev::get_eta(O.sched, O.eta0, O.power, O.beta, p.a.it0p[0] + p.a.it_b));
  const double tau = prox_threshold(M.P + (size_t)o * M.da * M.Kp + s, M.da, M.Kp, lam, p.tau[comp], p.cntp[comp]);
  if (threadIdx.x == 0) p.tau[comp] = tau;
}

template <int L>
__global__ __launch_bounds__(kBlock) void k_prox_apply(PassArgs p) {
  constexpr int R = kWave / L;
  const ModelView& M = p.a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t r = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  if (r >= (int64_t)M.nb * M.da) return;
  const int o = (int)(r / M.da);
  const size_t e = (size_t)r * M.Kp + 2 * l;
  double2 v = *reinterpret_cast<const double2*>(M.P + e);
  v.x = softthreshold(v.x, p.tau[o * M.Kp + 2 * l]);
  v.y = softthreshold(v.y, p.tau[o * M.Kp + 2 * l + 1]);
  *reinterpret_cast<double2*>(M.P + e) = v;
}

template <int L>
static void launch_psgd_step_t(nfm_ctx* ctx, const ModelView& M, const OptView& O, MbWork& W, const double* it0p, double it_b) {
  constexpr int R = kWave / L;
  hipStream_t st = ctx->stream;
  ProxArgs pa{M, O, it0p, it_b, W.prox.as<double>(), W.prox.as<double>() + (size_t)M.nb * M.da};
  const int64_t rows = (int64_t)M.nb * M.da;
  const unsigned row_blocks = (unsigned)((rows + kWavesPerBlock * R - 1) / (kWavesPerBlock * R));
  TimedLaunch tl(ctx, "psgd_step");
  if (rows > 0 && O.reg == NFM_REG_SQUAREDL12 && O.reg_transpose && M.da <= 16 * kProxBlock) {  // the one-launch step
    // 1024 threads per column: 256 threads with four times the values per thread measured 15.2 vs 13.8 us (the
    // strided column loads and stores want the memory parallelism)
    constexpr int tb = kProxBlock;
    const unsigned gx = (unsigned)(M.k + (std::max<int64_t>(M.d, 1) + tb - 1) / tb);
    const int vpt = (int)((M.da + tb - 1) / tb);
    const dim3 grid(gx, (unsigned)M.nb);
#define NFM_STEP(V, T) hipLaunchKernelGGL((k_psgd_step_columns<V, T>), grid, dim3(T), 0, st, pa)
    if (vpt <= 1) NFM_STEP(1, 1024);
    else if (vpt <= 2) NFM_STEP(2, 1024);
    else if (vpt <= 4) NFM_STEP(4, 1024);
    else if (vpt <= 8) NFM_STEP(8, 1024);
    else NFM_STEP(16, 1024);
#undef NFM_STEP
    return;
  }
  if (rows > 0) hipLaunchKernelGGL((k_psgd_dense<L>), dim3(row_blocks), dim3(kBlock), 0, st, pa);
  hipLaunchKernelGGL(k_psgd_linear, dim3((unsigned)((std::max<int64_t>(M.d, 1) + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, pa);
  if (rows > 0 && O.reg == NFM_REG_SQUAREDL12 && O.reg_transpose) {
    static const bool by_column = getenv("NFM_PROX_COLUMNS") && atoi(getenv("NFM_PROX_COLUMNS")) != 0;  // the old path (tuning)
    if (by_column) {
      hipLaunchKernelGGL(k_psgd_prox_columns, dim3((unsigned)M.k, (unsigned)M.nb), dim3(kProxBlock), 0, st, pa);
      return;
    }
    const int C = M.nb * M.Kp;
    double* sbase = W.prox.as<double>() + (size_t)M.nb * M.da;  // scratch behind the norms: state, then partials
    const int G = (int)std::min<int64_t>(kPassBlocks, (M.da + kWavesPerBlock * R - 1) / (kWavesPerBlock * R));
    PassArgs ps{pa, sbase, sbase + C, reinterpret_cast<int*>(sbase + 2 * C), reinterpret_cast<int*>(sbase + 3 * C), sbase + 3 * C + 2, G};
    hipLaunchKernelGGL(k_prox_init, dim3(1), dim3(kBlock), 0, st, ps);
    // passes grow with lam (3 at 1e-6, 4 at 1e-4, 8 at 1e-2, 10 at 1 on N(0, 0.01) columns): enqueue what the largest
    // lam of the schedule needs (eta <= eta0 except for pegasos); k_prox_finish covers any shortfall
    const double lam_max = O.gamma * O.eta0 / (1.0 + O.eta0 * O.beta);
    const int passes = O.sched == NFM_SCHED_PEGASOS ? kPasses : (lam_max < 1e-5 ? 4 : (lam_max < 1e-3 ? 6 : kPasses));
    for (int pass = 0; pass < passes; ++pass) {
      hipLaunchKernelGGL((k_prox_pass_partial<L>), dim3((unsigned)G, (unsigned)M.nb), dim3(kBlock), 0, st, ps);
      hipLaunchKernelGGL((k_prox_pass_combine<L>), dim3((unsigned)L, (unsigned)M.nb), dim3(kBlock), 0, st, ps);
    }
    hipLaunchKernelGGL(k_prox_finish, dim3((unsigned)M.Kp, (unsigned)M.nb), dim3(kProxBlock), 0, st, ps);
    hipLaunchKernelGGL((k_prox_apply<L>), dim3(row_blocks), dim3(kBlock), 0, st, ps);
  } else if (rows > 0 && O.reg == NFM_REG_SQUAREDL21) {
    hipLaunchKernelGGL(k_psgd_prox_norms, dim3((unsigned)M.nb), dim3(kProxBlock), 0, st, pa);
    hipLaunchKernelGGL((k_psgd_rescale_rows<L>), dim3(row_blocks), dim3(kBlock), 0, st, pa);
  }
}

void launch_psgd_step(nfm_ctx* ctx, const ModelView& M, const OptView& O, MbWork& W, const double* it0p, double it_b) {
  switch (M.L) {
    case 1: return launch_psgd_step_t<1>(ctx, M, O, W, it0p, it_b);
    case 2: return launch_psgd_step_t<2>(ctx, M, O, W, it0p, it_b);
    case 4: return launch_psgd_step_t<4>(ctx, M, O, W, it0p, it_b);
    case 8: return launch_psgd_step_t<8>(ctx, M, O, W, it0p, it_b);
    case 16: return launch_psgd_step_t<16>(ctx, M, O, W, it0p, it_b);
    case 32: return launch_psgd_step_t<32>(ctx, M, O, W, it0p, it_b);
    default: return launch_psgd_step_t<64>(ctx, M, O, W, it0p, it_b);
  }
}

}  // namespace nfm
