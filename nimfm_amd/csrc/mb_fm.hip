// nimfm_amd/csrc/mb_fm.hip -- NFM_MODE_MINIBATCH for FactorizationMachine: the throughput path.
//
// Replaces the reference's Hogwild drivers (optimizer/sgd_multi.nim:21-37,83-101,
// adagrad_multi.nim:15-36,78-96) -- T threads racing on shared P/w/intercept/it -- with a
// deterministic rule: all samples of a batch see the batch-start
// parameters; their per-sample updates (the reference's expressions: sgd.nim:205-243,
// fit_linear.nim:41-47; adagrad.nim:87-134) are summed per coordinate in sample order.
//
// Two kernels per batch, both HBM/L2-gather bound, no atomics:
//   row phase     one wavefront per SAMPLE: gathers the row's parameter rows (16 B per lane,
//                 coalesced segments of Kp*8 bytes), forms A = sum x p (and sum (x p)^2) per factor
//                 with shuffles, yhat, loss, dL; writes the per-factor sums A[s] (Kp doubles) and a
//                 32-byte record {dL, eta_P, eta_w} per sample.
//   column phase  L lanes per UNIQUE FEATURE of the batch (plan.hip): reads the parameter row once,
//                 walks the feature's touches (sample, x) in sample order, recomputes
//                 dA = x (A[s] - p x) from the sample's A row (L2-resident), accumulates, writes the
//                 row once.  Rows touched c times in a batch are read and written once, not c times.
//   batch finish  one workgroup: fixed-order reduction of the per-block partial sums (loss, viol,
//                 intercept gradient), intercept update.
// L2 decay is carried by the global scales (common.h): the schedule kernel forms the per-batch
// products of (1 - eta_t * reg) and the prefix kernel the scale at every batch boundary.
#include "fm_device.h"
#include "mb.h"

namespace nfm {

struct SampleRec {
  double dL, etaP, etaw, pad;
};
struct PartA {
  double loss, viol, acc0, acc1;
};

static_assert(sizeof(SampleRec) == 32 && sizeof(PartA) == 32, "record layout");

// ------------------------------------------------------------------------------------------------
// schedule: per-batch decay products and scales (SGD only)
// ------------------------------------------------------------------------------------------------
constexpr int kFtab = 64;  // touch counts 1..kFtab have a tabulated decay correction

__global__ void k_schedule(OptView O, int fit_linear, int fit_intercept, const int64_t* __restrict__ bat_pos, double it0,
                           double* __restrict__ Dtab /*[nb][4]*/, double* __restrict__ Ftab /*[nb][2][kFtab]*/) {
  __shared__ double red[3][kBlock];
  const int b = blockIdx.x;
  const int64_t p0 = bat_pos[b], p1 = bat_pos[b + 1];
  double dP = 1.0, dw = 1.0, d0 = 1.0;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += kBlock) {
    const double it = it0 + (double)p;
    dP *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.beta, it) * O.beta;
    if (fit_linear) dw *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it) * O.alpha;
    if (fit_intercept) d0 *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it) * O.alpha0;
  }
  red[0][threadIdx.x] = dP;
  red[1][threadIdx.x] = dw;
  red[2][threadIdx.x] = d0;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] *= red[0][threadIdx.x + s];
      red[1][threadIdx.x] *= red[1][threadIdx.x + s];
      red[2][threadIdx.x] *= red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Dtab[4 * b + 0] = red[0][0];
    Dtab[4 * b + 1] = red[1][0];
    Dtab[4 * b + 2] = red[2][0];
    Dtab[4 * b + 3] = 0.0;
  }
  // a coordinate touched c times receives D^(1/c) instead of D; relative to the global scale
  // (which advances by D) that is the factor D^(1/c) / D, tabulated for c = 1..kFtab
  if (threadIdx.x < 2 * kFtab) {
    const int which = threadIdx.x / kFtab, c = threadIdx.x % kFtab + 1;
    const double D = red[which][0];
    Ftab[((size_t)b * 2 + which) * kFtab + (c - 1)] = c == 1 ? 1.0 : pow(D, 1.0 / (double)c) / D;
  }
}

__global__ void k_scale_prefix(double* __restrict__ sc, const double* __restrict__ Dtab, double* __restrict__ Stab, int64_t nb) {
  double sP = sc[SC_SCALE_P], sw = sc[SC_SCALE_W];
  for (int64_t b = 0; b < nb; ++b) {
    Stab[2 * b] = sP;
    Stab[2 * b + 1] = sw;
    sP *= Dtab[4 * b];
    sw *= Dtab[4 * b + 1];
  }
  Stab[2 * nb] = sP;
  Stab[2 * nb + 1] = sw;
  sc[SC_SCALE_P] = sP;
  sc[SC_SCALE_W] = sw;
}

// ------------------------------------------------------------------------------------------------
// row phase
// ------------------------------------------------------------------------------------------------
struct RowArgs {
  CsrView X;
  ModelView M;
  OptView O;
  const int64_t* perm;  // relative to begin, or null
  int64_t begin, p0;    // first sample of the batch = begin + p0 (position), identity when perm null
  int32_t len, use_stored, TA, pad_;
  double it_b;
  const double* scales;  // {scale_P, scale_w} at the batch start
  double* Abuf;          // [len][TA][Kp]
  SampleRec* rec;        // [len]
  PartA* parts;          // [gridDim.x]
};

template <int L, class PS>
__device__ __forceinline__ double row_forward(const PS& ps, const CsrView& X, const ModelView& M, int64_t q0, int m,
                                              int m_tot, int lane, double* __restrict__ Arow) {
  const int g = lane / L, l = lane % L;
  double acc = 0.0;
  int slot = 0;
  for (int o = 0; o < M.nb; ++o) {
    const size_t blk = (size_t)o * M.da * M.Kp;
    const int deg = M.degree - o;
    double2 ker;
    switch (deg) {
      case 2: {
        double2 A1, A2;
        dev::anova_fwd_deg2<L>(ps, X, q0, m, m_tot, blk, M.Kp, lane, A1, A2);
        ker.x = (A1.x * A1.x - A2.x) / 2;
        ker.y = (A1.y * A1.y - A2.y) / 2;
        if (g == 0) *reinterpret_cast<double2*>(Arow + (size_t)slot * M.Kp + 2 * l) = A1;
        slot += 1;
        break;
      }
#define NFM_DEG_CASE(DG)                                                                       \
  case DG: {                                                                                   \
    double2 E[DG + 1];                                                                         \
    dev::anova_fwd_degn<L, DG>(ps, X, q0, m, m_tot, blk, M.Kp, lane, E);                       \
    ker = E[DG];                                                                               \
    if (g == 0) {                                                                              \
      _Pragma("unroll") for (int t = 1; t < DG; ++t)                                           \
          *reinterpret_cast<double2*>(Arow + (size_t)(slot + t - 1) * M.Kp + 2 * l) = E[t];    \
    }                                                                                          \
    slot += DG - 1;                                                                            \
    break;                                                                                     \
  }
        NFM_DEG_CASE(3)
        NFM_DEG_CASE(4)
        NFM_DEG_CASE(5)
        NFM_DEG_CASE(6)
#undef NFM_DEG_CASE
      default:
        ker = {0.0, 0.0};
    }
    acc += dev::sum_factors<L>(ker);
  }
  return acc;
}

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_row_phase(RowArgs a) {
  __shared__ double red[kWavesPerBlock][4];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int pib = blockIdx.x * kWavesPerBlock + wv;
  double r_loss = 0.0, r_viol = 0.0, r_acc0 = 0.0, r_acc1 = 0.0;
  if (pib < a.len) {
    const int64_t pos = a.p0 + pib;
    const int64_t i = a.perm ? a.perm[pos] : a.begin + pos;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int m_tot = m + M.n_aug;
    const double y = dev::target_of(X.y[i], M.task);
    double* Arow = a.Abuf + (size_t)pib * a.TA * M.Kp;
    double yh, b0;
    if (OPT == OPT_SGD) {
      const double sP = a.scales[0], sw = a.scales[1];
      b0 = M.sc[SC_INTERCEPT];
      double lin = 0.0;
      for (int c = 0; c * kWave < m; ++c) {
        const int q = c * kWave + lane;
        if (q < m) lin += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
      }
      lin = dev::wave_sum(lin);
      const dev::PlainParams ps{M.P, sP};
      yh = b0 + lin + row_forward<L>(ps, X, M, q0, m, m_tot, lane, Arow);
    } else {
      const double itp = a.it_b - 1.0;
      const bool stored = a.use_stored != 0;
      b0 = M.sc[SC_INTERCEPT];
      if (!stored && M.fit_intercept) b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
      const double denw = itp * O.eta0 * O.alpha;
      double lin = 0.0;
      for (int c = 0; c * kWave < m; ++c) {
        const int q = c * kWave + lane;
        if (q < m) {
          const int j = X.indices[q0 + q];
          double wj = M.w[j];
          if (!stored && M.fit_linear) wj = -O.eta0 * O.Gw[j] / (denw + sqrt(O.Nw[j]));
          lin += wj * X.data[q0 + q];
        }
      }
      lin = dev::wave_sum(lin);
      double acc;
      if (stored) {
        const dev::PlainParams ps{M.P, 1.0};
        acc = row_forward<L>(ps, X, M, q0, m, m_tot, lane, Arow);
      } else {
        const dev::AdaParams ps{O.G, O.N, O.eta0, O.eta0 * itp * O.beta};
        acc = row_forward<L>(ps, X, M, q0, m, m_tot, lane, Arow);
      }
      yh = b0 + lin + acc;
    }
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);
    r_loss = dev::loss_value(O.loss, O.loss_param, y, yh);
    if (OPT == OPT_SGD) {
      const double it = a.it_b + (double)pib;
      const double etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
      const double etaw = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
      if (lane == 0) a.rec[pib] = SampleRec{dL, etaP, etaw, 0.0};
      if (M.fit_intercept) {
        const double eta0 = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it);
        r_acc0 = eta0 * dL;
        r_acc1 = eta0;
      }
    } else {
      if (lane == 0) a.rec[pib] = SampleRec{dL, 0.0, 0.0, 0.0};
      if (M.fit_intercept) {
        r_acc0 = dL;
        r_acc1 = dL * dL;
      }
    }
  }
  if (lane == 0) {
    red[wv][0] = r_loss;
    red[wv][1] = r_viol;
    red[wv][2] = r_acc0;
    red[wv][3] = r_acc1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PartA p{0.0, 0.0, 0.0, 0.0};
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) {
      p.loss += red[w_][0];
      p.viol += red[w_][1];
      p.acc0 += red[w_][2];
      p.acc1 += red[w_][3];
    }
    a.parts[blockIdx.x] = p;
  }
}

// ------------------------------------------------------------------------------------------------
// column phase
// ------------------------------------------------------------------------------------------------
struct ColArgs {
  ModelView M;
  OptView O;
  const int32_t* ucol;
  const int64_t* uptr;
  const int32_t* tpos;
  const double* tx;
  int64_t u0, u1;
  const double* scales_b;  // {scale_P, scale_w} at the batch start
  const double* scales_n;  // ... at the next batch start
  const double* Dtab_b;    // SGD: {D_P, D_w, D_0} of this batch
  const double* Ftab_b;    // SGD: [2][kFtab] decay corrections by touch count
  const double* Abuf;
  const SampleRec* rec;
  double* parts;  // [gridDim.x]
  double it_b;
  int32_t TA, use_stored;
};

// one parameter block (order) of one unique feature: this lane's factor pair at element e
template <int DEG, int OPT>
__device__ __forceinline__ double col_block(const ColArgs& a, size_t e, int slot, int l, int64_t t0, int64_t t1, double sP,
                                            double sPn, double fP) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  double viol = 0.0;
  double2 stored = {0.0, 0.0}, g2 = {0.0, 0.0}, n2 = {0.0, 0.0}, p;
  if (OPT == OPT_SGD) {
    stored = *reinterpret_cast<const double2*>(M.P + e);
    p.x = sP * stored.x;
    p.y = sP * stored.y;
  } else {
    g2 = *reinterpret_cast<const double2*>(O.G + e);
    n2 = *reinterpret_cast<const double2*>(O.N + e);
    if (a.use_stored) {
      p = *reinterpret_cast<const double2*>(M.P + e);
    } else {
      const double tmp = O.eta0 * (a.it_b - 1.0) * O.beta;
      p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmp);
      p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmp);
      if (O.track_viol) {  // adagrad.nim:96-99: sum |old - new| over the touched rows
        stored = *reinterpret_cast<const double2*>(M.P + e);
        viol += fabs(stored.x - p.x) + fabs(stored.y - p.y);
        *reinterpret_cast<double2*>(M.P + e) = p;
      }
    }
  }
  double2 acc = {0.0, 0.0}, accn = {0.0, 0.0};
  double seta = 0.0;
  for (int64_t t = t0; t < t1; ++t) {
    const int pib = a.tpos[t];
    const double x = a.tx[t];
    const SampleRec r = a.rec[pib];
    const double* Ar = a.Abuf + ((size_t)pib * a.TA + slot) * M.Kp + 2 * l;
    double Ax[DEG - 1], Ay[DEG - 1];
#pragma unroll
    for (int tt = 0; tt < DEG - 1; ++tt) {
      const double2 v = *reinterpret_cast<const double2*>(Ar + (size_t)tt * M.Kp);
      Ax[tt] = v.x;
      Ay[tt] = v.y;
    }
    const double dAx = dev::anova_grad<DEG>(x, p.x, Ax);
    const double dAy = dev::anova_grad<DEG>(x, p.y, Ay);
    if (OPT == OPT_SGD) {  // sgd.nim:220-222, averaged per coordinate below
      acc.x += r.etaP * (r.dL * dAx);
      acc.y += r.etaP * (r.dL * dAy);
      seta += r.etaP;
    } else {  // adagrad.nim:122-124
      const double gx = r.dL * dAx, gy = r.dL * dAy;
      acc.x += gx;
      acc.y += gy;
      accn.x += gx * gx;
      accn.y += gy * gy;
    }
  }
  if (OPT == OPT_SGD) {
    const double c = (double)(t1 - t0);
    viol += fabs((acc.x + seta * O.beta * p.x) / c) + fabs((acc.y + seta * O.beta * p.y) / c);
    stored.x = stored.x * fP - (acc.x / c) / sPn;
    stored.y = stored.y * fP - (acc.y / c) / sPn;
    *reinterpret_cast<double2*>(M.P + e) = stored;
  } else {
    g2.x += acc.x;
    g2.y += acc.y;
    n2.x += accn.x;
    n2.y += accn.y;
    *reinterpret_cast<double2*>(O.G + e) = g2;
    *reinterpret_cast<double2*>(O.N + e) = n2;
  }
  return viol;
}

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_col_phase(ColArgs a) {
  constexpr int R = kWave / L;
  __shared__ double red[kWavesPerBlock];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t u = a.u0 + ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  double viol = 0.0;
  if (u < a.u1) {
    const int64_t j = a.ucol[u];
    const int64_t t0 = a.uptr[u], t1 = a.uptr[u + 1];
    double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0, fP = 1.0, fw = 1.0;
    if (OPT == OPT_SGD) {
      sP = a.scales_b[0];
      sw = a.scales_b[1];
      sPn = a.scales_n[0];
      swn = a.scales_n[1];
      const int64_t c = t1 - t0;
      if (c > 1) {
        if (c <= kFtab) {
          fP = a.Ftab_b[c - 1];
          fw = a.Ftab_b[kFtab + c - 1];
        } else {
          fP = pow(a.Dtab_b[0], 1.0 / (double)c) / a.Dtab_b[0];
          fw = pow(a.Dtab_b[1], 1.0 / (double)c) / a.Dtab_b[1];
        }
      }
    }
    int slot = 0;
    for (int o = 0; o < M.nb; ++o) {
      const size_t e = ((size_t)o * M.da + j) * M.Kp + 2 * l;
      const int deg = M.degree - o;
      switch (deg) {
        case 2: viol += col_block<2, OPT>(a, e, slot, l, t0, t1, sP, sPn, fP); break;
        case 3: viol += col_block<3, OPT>(a, e, slot, l, t0, t1, sP, sPn, fP); break;
        case 4: viol += col_block<4, OPT>(a, e, slot, l, t0, t1, sP, sPn, fP); break;
        case 5: viol += col_block<5, OPT>(a, e, slot, l, t0, t1, sP, sPn, fP); break;
        case 6: viol += col_block<6, OPT>(a, e, slot, l, t0, t1, sP, sPn, fP); break;
        default: break;
      }
      slot += deg - 1;
    }
    // linear term (fit_linear.nim:41-57); dummy features have no w
    if (M.fit_linear && j < M.d && l == 0) {
      if (OPT == OPT_SGD) {
        const double wt = M.w[j];
        const double wj = sw * wt;
        double accw = 0.0, setaw = 0.0;
        for (int64_t t = t0; t < t1; ++t) {
          const SampleRec r = a.rec[a.tpos[t]];
          accw += r.etaw * (r.dL * a.tx[t]);
          setaw += r.etaw;
        }
        const double c = (double)(t1 - t0);
        viol += fabs((accw + setaw * O.alpha * wj) / c);
        M.w[j] = wt * fw - (accw / c) / swn;
      } else {
        const double wt = M.w[j];
        double gw = O.Gw[j], nw = O.Nw[j];
        if (!a.use_stored) {
          const double wj = -O.eta0 * gw / ((a.it_b - 1.0) * O.eta0 * O.alpha + sqrt(nw));
          viol += fabs(wt - wj);
          M.w[j] = wj;
        }
        double ag = 0.0, an = 0.0;
        for (int64_t t = t0; t < t1; ++t) {
          const double gx = a.rec[a.tpos[t]].dL * a.tx[t];
          ag += gx;
          an += gx * gx;
        }
        O.Gw[j] = gw + ag;
        O.Nw[j] = nw + an;
      }
    }
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[w_];
    a.parts[blockIdx.x] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// batch finish
// ------------------------------------------------------------------------------------------------
struct FinArgs {
  ModelView M;
  OptView O;
  const PartA* partsA;
  const double* partsB;
  const double* Dtab_b;  // SGD: {D_P, D_w, D_0}
  double* out_acc;       // {loss_sum, viol_sum}
  double it_b, len;
  int32_t nA, nB, use_stored, opt;
};

__global__ __launch_bounds__(kBlock) void k_batch_finish(FinArgs a) {
  __shared__ double red[5][kBlock];
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < a.nA; i += kBlock) {
    const PartA p = a.partsA[i];
    s[0] += p.loss;
    s[1] += p.viol;
    s[2] += p.acc0;
    s[3] += p.acc1;
  }
  for (int i = threadIdx.x; i < a.nB; i += kBlock) s[4] += a.partsB[i];
  for (int c = 0; c < 5; ++c) red[c][threadIdx.x] = s[c];
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int c = 0; c < 5; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double viol = red[1][0] + red[4][0];
    const ModelView& M = a.M;
    const OptView& O = a.O;
    if (M.fit_intercept) {
      if (a.opt == OPT_SGD) {  // the intercept is touched by every sample of the batch: c = len
        const double b0 = M.sc[SC_INTERCEPT], D0 = a.Dtab_b[2];
        const double f0 = a.len == 1.0 ? D0 : pow(D0, 1.0 / a.len);
        viol += fabs((red[2][0] + red[3][0] * O.alpha0 * b0) / a.len);
        M.sc[SC_INTERCEPT] = f0 * b0 - red[2][0] / a.len;
      } else {
        if (!a.use_stored) {  // adagrad.nim:102-106
          const double old = M.sc[SC_INTERCEPT];
          const double nb_ = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * (a.it_b - 1.0) * O.alpha0);
          viol += fabs(old - nb_);
          M.sc[SC_INTERCEPT] = nb_;
        }
        O.gsc[0] += red[2][0];
        O.gsc[1] += red[3][0];
      }
    }
    a.out_acc[0] += red[0][0];
    a.out_acc[1] += viol;
  }
}

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
template <int L, int OPT>
static int run_batches(nfm_ctx* ctx, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                       int64_t it0, int TA) {
  constexpr int R = kWave / L;
  hipStream_t st = ctx->stream;
  const double* Stab = W.Stab.as<double>();
  const double* Dtab = W.Dtab.as<double>();
  for (int64_t b = 0; b < P.n_batches; ++b) {
    const int64_t p0 = P.bat_pos[b];
    const int len = (int)(P.bat_pos[b + 1] - p0);
    const int use_stored = (OPT == OPT_ADAGRAD && P.first_singleton && b == 0) ? 1 : 0;
    const double it_b = (double)(it0 + p0);
    const int nA = (len + kWavesPerBlock - 1) / kWavesPerBlock;
    {
      RowArgs ra{X, M, O, P.has_perm ? P.perm.as<int64_t>() : nullptr, P.begin, p0, len, use_stored, TA, 0, it_b,
                 OPT == OPT_SGD ? Stab + 2 * b : M.sc, W.Abuf.as<double>(), W.rec.as<SampleRec>(), W.partsA.as<PartA>()};
      TimedLaunch tl(ctx, "row_phase");
      hipLaunchKernelGGL((k_row_phase<L, OPT>), dim3(nA), dim3(kBlock), 0, st, ra);
    }
    const int64_t u0 = P.bat_uoff[b], u1 = P.bat_uoff[b + 1];
    const int per_block = kWavesPerBlock * R;
    int nB = (int)((u1 - u0 + per_block - 1) / per_block);
    if (nB > 0) {
      ColArgs ca{M, O, P.ucol.as<int32_t>(), P.uptr.as<int64_t>(), P.tpos.as<int32_t>(), P.tx.as<double>(), u0, u1,
                 OPT == OPT_SGD ? Stab + 2 * b : M.sc, OPT == OPT_SGD ? Stab + 2 * (b + 1) : M.sc,
                 OPT == OPT_SGD ? Dtab + 4 * b : nullptr,
                 OPT == OPT_SGD ? W.Ftab.as<double>() + (size_t)b * 2 * kFtab : nullptr, W.Abuf.as<double>(),
                 W.rec.as<SampleRec>(), W.partsB.as<double>(), it_b, TA, use_stored};
      TimedLaunch tl(ctx, "col_phase");
      hipLaunchKernelGGL((k_col_phase<L, OPT>), dim3(nB), dim3(kBlock), 0, st, ca);
    }
    {
      FinArgs fa{M, O, W.partsA.as<PartA>(), W.partsB.as<double>(), OPT == OPT_SGD ? Dtab + 4 * b : nullptr,
                 W.out_acc.as<double>(), it_b, (double)len, nA, nB, use_stored, OPT};
      TimedLaunch tl(ctx, "batch_finish");
      hipLaunchKernelGGL(k_batch_finish, dim3(1), dim3(kBlock), 0, st, fa);
    }
  }
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

template <int L>
static int run_batches_L(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P,
                         MbWork& W, int64_t it0, int TA) {
  if (opt_kind == OPT_SGD) return run_batches<L, OPT_SGD>(ctx, X, M, O, P, W, it0, TA);
  return run_batches<L, OPT_ADAGRAD>(ctx, X, M, O, P, W, it0, TA);
}

int mb_fm_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                int64_t it0, double* out2_host) {
  NFM_CHECK(M.kind == NFM_KIND_FM, NFM_ERR_UNSUPPORTED, "mb_fm_epoch: FM only");
  NFM_CHECK(M.degree <= 6, NFM_ERR_UNSUPPORTED, "mini-batch mode supports degree <= 6");
  NFM_CHECK(M.Kp <= 128, NFM_ERR_UNSUPPORTED, "mini-batch mode supports n_components <= 128");
  hipStream_t st = ctx->stream;
  int TA = 0;
  for (int o = 0; o < M.nb; ++o) TA += M.degree - o - 1;
  constexpr int kMinGroupsPerBlock = kWavesPerBlock;  // L = 64
  NFM_TRY(W.Abuf.ensure(sizeof(double) * (size_t)std::max<int64_t>(P.max_batch, 1) * std::max(TA, 1) * M.Kp));
  NFM_TRY(W.rec.ensure(sizeof(SampleRec) * (size_t)std::max<int64_t>(P.max_batch, 1)));
  NFM_TRY(W.partsA.ensure(sizeof(PartA) * (size_t)(P.max_batch / kWavesPerBlock + 1)));
  NFM_TRY(W.partsB.ensure(sizeof(double) * (size_t)(P.max_unique / kMinGroupsPerBlock + 1)));
  NFM_TRY(W.Dtab.ensure(sizeof(double) * 4 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Stab.ensure(sizeof(double) * 2 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Ftab.ensure(sizeof(double) * 2 * kFtab * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.out_acc.ensure(sizeof(double) * 2));
  NFM_HIP_CHECK(hipMemsetAsync(W.out_acc.p, 0, sizeof(double) * 2, st));
  if (P.n_batches > 0 && opt_kind == OPT_SGD) {
    TimedLaunch tl(ctx, "schedule");
    hipLaunchKernelGGL(k_schedule, dim3((unsigned)P.n_batches), dim3(kBlock), 0, st, O, M.fit_linear, M.fit_intercept,
                       P.bat_pos_dev.as<int64_t>(), (double)it0, W.Dtab.as<double>(), W.Ftab.as<double>());
    hipLaunchKernelGGL(k_scale_prefix, dim3(1), dim3(1), 0, st, M.sc, W.Dtab.as<double>(), W.Stab.as<double>(), P.n_batches);
    NFM_HIP_CHECK(hipGetLastError());
  }
  int rc = NFM_ERR_UNSUPPORTED;
  switch (M.L) {
    case 1: rc = run_batches_L<1>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 2: rc = run_batches_L<2>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 4: rc = run_batches_L<4>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 8: rc = run_batches_L<8>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 16: rc = run_batches_L<16>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 32: rc = run_batches_L<32>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
    case 64: rc = run_batches_L<64>(ctx, opt_kind, X, M, O, P, W, it0, TA); break;
  }
  NFM_TRY(rc);
  NFM_HIP_CHECK(hipMemcpyAsync(out2_host, W.out_acc.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  return NFM_OK;
}

}  // namespace nfm
