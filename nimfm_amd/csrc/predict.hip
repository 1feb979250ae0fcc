// nimfm_amd/csrc/predict.hip -- decisionFunction on the GPU.
//
// Replaces model/factorization_machine.nim:100-122 (-> kernels.nim:14-19 `linear`, :46-64
// `anova`) and model/field_aware_factorization_machine.nim:52-76.  The reference makes nOrders*k
// full passes over the CSR matrix, one per latent factor, gathering P[s][j] from k separate heap
// rows; here L*SPLIT lanes own one sample (fm_device.h), read each parameter row once (coalesced,
// 16 B per lane) from the [j][s] layout and finish the per-factor sums with shuffles.
// Bound: HBM/L2 gather of P rows; algorithmic bytes per sample 12m + 8 + O*8*m*k + 8m + 8
// (SURVEY.md 8d).
#include "fm_device.h"

namespace nfm {

template <int L, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_fm_predict(CsrView X, ModelView M, double* __restrict__ out) {
  constexpr int LPS = L * SPLIT, SPW = kWave / LPS;
  const int lane = threadIdx.x & (kWave - 1);
  const int sidx = lane / LPS, slot = (lane / L) % SPLIT, l = lane % L;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const double sw = M.sc[SC_SCALE_W], b = M.sc[SC_INTERCEPT];
  const dev::PlainParams ps{M.P, M.sc[SC_SCALE_P]};
  const double lam0 = M.lams[2 * l], lam1 = M.lams[2 * l + 1];
  for (int64_t i0 = wave0 * SPW; i0 < X.n; i0 += nwaves * SPW) {
    const int64_t i = i0 + sidx;
    const bool valid = i < X.n;
    const int64_t q0 = valid ? X.indptr[i] : 0;
    const int m = valid ? (int)(X.indptr[i + 1] - q0) : 0;
    const int m_tot = valid ? m + M.n_aug : 0;
    // linear term (kernels.nim:14-19; dummies excluded): nnz dealt round-robin to the sample's lanes
    double part = 0.0;
    for (int q = slot * L + l; q < m; q += LPS) part += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
    for (int o = 0; o < M.nb; ++o) {
      const size_t blk = M.row(o, 0) * M.Kp;
      const int rstride = (int)M.rs * M.Kp;  // doubles between consecutive features' rows of this order
      const int deg = M.deg_of(o);
      double2 ker;
      if (deg == 2) {
        double2 A1, A2;
        dev::anova_fwd_deg2<L, SPLIT>(ps, X, q0, m, m_tot, blk, rstride, slot, l, A1, A2);
        ker.x = (A1.x * A1.x - A2.x) / 2.0;
        ker.y = (A1.y * A1.y - A2.y) / 2.0;
      } else {
        double2 E[dev::kMaxDeg + 1];
        dev::anova_fwd_degn<L, SPLIT>(ps, X, q0, m, m_tot, blk, rstride, slot, l, lane, deg, E);
        ker = dev::pick(E, deg);
      }
      if (M.kc == 1) {
        if (slot == 0) part += ker.x * lam0 + ker.y * lam1;  // every slot holds the same totals
      } else if (slot == 0) {  // more than 128 factors: block o holds the factors (o % kc) * Kp ... of its order
        const double* lm = M.lams + (size_t)(o % M.kc) * M.Kp;
        part += ker.x * lm[2 * l] + ker.y * lm[2 * l + 1];
      }
    }
    // sum over the sample's L*SPLIT lanes
#pragma unroll
    for (int s = 1; s < LPS; s <<= 1) part += dev::shfl_xor_d(part, s);
    if (valid && slot == 0 && l == 0) out[i] = b + part;
  }
}

// ---- several orders (fitLower = explicit: degree 3 has a degree-3 and a degree-2 block, kernels.nim:46-64 once per order,
// factorization_machine.nim:106-120) ----
// The training layout is order-major: the two 64-byte rows of one feature (k = 8) lie a whole table apart, every 16-byte
// lane load of k_fm_predict<4, 1> uses half of the 128-byte line it opens, and the orders are walked one after the other
// (the degree >= 3 recursion one dependent row load at a time): 0.36 of the roofline where the same bytes as ONE order of
// k = 16 run at 0.87 (round 3).  Here the blocks are interleaved per feature first -- Pf[j][o][Kp], a streaming copy of the
// table, 13 MB for cfg5 against 4 GB of gathers -- and a sample's lane group covers ALL orders of a feature with one
// contiguous access (NBP * Kp * 8 bytes: 128 at k = 8, two orders); lane l belongs to order l / L and carries that order's
// recursion (degree 2: the sum-of-squares form, as the reference; degree >= 3: the DP), kFwdUnroll features in flight.
__global__ __launch_bounds__(kBlock) void k_interleave_orders(const double* __restrict__ P, double* __restrict__ Pf, int nb, int NBP, int64_t da,
                                                              int Kp) {
  const int64_t total = da * NBP * (Kp / 2);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(t % (Kp / 2));
    const int64_t r = t / (Kp / 2);
    const int o = (int)(r % NBP);
    const int64_t j = r / NBP;
    double2 v = {0.0, 0.0};
    if (o < nb) v = *reinterpret_cast<const double2*>(P + ((size_t)o * da + j) * Kp + 2 * h);
    *reinterpret_cast<double2*>(Pf + (size_t)r * Kp + 2 * h) = v;
  }
}

template <int LT, int SPLIT>  // LT = NBP * L lanes cover all orders of one feature
__global__ __launch_bounds__(kBlock) void k_fm_predict_orders(CsrView X, ModelView M, const double* __restrict__ Pf, int lgL,
                                                              double* __restrict__ out) {
  constexpr int LPS = LT * SPLIT, SPW = kWave / LPS;
  const int lane = threadIdx.x & (kWave - 1);
  const int sidx = lane / LPS, slot = (lane / LT) % SPLIT, l = lane % LT;
  const int o = l >> lgL, ll = l & ((1 << lgL) - 1);
  const int mydeg = o < M.nb ? M.deg_of(o) : 0;  // (lanes of a padding block: their rows are zero, they add nothing)
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const double sw = M.sc[SC_SCALE_W], b = M.sc[SC_INTERCEPT], sP = M.sc[SC_SCALE_P];
  const double lam0 = M.lams[2 * ll], lam1 = M.lams[2 * ll + 1];
  const int rowlen = LT * 2;  // doubles per interleaved feature
  for (int64_t i0 = wave0 * SPW; i0 < X.n; i0 += nwaves * SPW) {
    const int64_t i = i0 + sidx;
    const bool valid = i < X.n;
    const int64_t q0 = valid ? X.indptr[i] : 0;
    const int m = valid ? (int)(X.indptr[i + 1] - q0) : 0;
    const int m_tot = valid ? m + M.n_aug : 0;
    double part = 0.0;
    for (int q = slot * LT + l; q < m; q += LPS) part += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
    double2 a1 = {0.0, 0.0}, a2 = {0.0, 0.0};
    double2 E[dev::kMaxDeg + 1];
#pragma unroll
    for (int t = 0; t <= dev::kMaxDeg; ++t) E[t] = {0.0, 0.0};
    E[0] = {1.0, 1.0};
    for (int q = slot; q < m_tot; q += dev::kFwdUnroll * SPLIT) {
      int j[dev::kFwdUnroll];
      double x[dev::kFwdUnroll];
      double2 p[dev::kFwdUnroll];
#pragma unroll
      for (int u = 0; u < dev::kFwdUnroll; ++u) dev::row_entry(X, q0, m, m_tot, q + u * SPLIT, j[u], x[u]);
#pragma unroll
      for (int u = 0; u < dev::kFwdUnroll; ++u) p[u] = *reinterpret_cast<const double2*>(Pf + (size_t)j[u] * rowlen + 2 * l);
#pragma unroll
      for (int u = 0; u < dev::kFwdUnroll; ++u) {
        const double px = sP * p[u].x, py = sP * p[u].y;  // (PlainParams::load's order of operations)
        const double tx = x[u] * px, ty = x[u] * py;
        a1.x += tx;
        a1.y += ty;
        a2.x += tx * tx;
        a2.y += ty * ty;
#pragma unroll
        for (int t = dev::kMaxDeg; t >= 1; --t)
          if (t <= mydeg && mydeg > 2) {  // sgd.nim:152-159 / kernels.nim:54-58: A[t] += A[t-1] * p * x, t = deg .. 1
            E[t].x += E[t - 1].x * px * x[u];
            E[t].y += E[t - 1].y * py * x[u];
          }
      }
    }
#pragma unroll
    for (int s = LT; s < LT * SPLIT; s <<= 1) {
      a1.x += dev::shfl_xor_d(a1.x, s);
      a1.y += dev::shfl_xor_d(a1.y, s);
      a2.x += dev::shfl_xor_d(a2.x, s);
      a2.y += dev::shfl_xor_d(a2.y, s);
    }
    if constexpr (SPLIT > 1) dev::combine_slots_degn<LT, SPLIT>(E, mydeg > 2 ? mydeg : 0, lane);
    double2 ker = {0.0, 0.0};
    if (mydeg == 2) {
      ker.x = (a1.x * a1.x - a2.x) / 2.0;
      ker.y = (a1.y * a1.y - a2.y) / 2.0;
    } else if (mydeg > 2) {
      ker = dev::pick(E, mydeg);
    }
    if (slot == 0) part += ker.x * lam0 + ker.y * lam1;
#pragma unroll
    for (int s = 1; s < LPS; s <<= 1) part += dev::shfl_xor_d(part, s);
    if (valid && slot == 0 && l == 0) out[i] = b + part;
  }
}

template <int LT, int SPLIT>
static int launch_fm_predict_orders(nfm_ctx* ctx, const CsrView& X, const ModelView& M, const double* Pf, int lgL, double* out) {
  constexpr int SPW = kWave / (LT * SPLIT);
  int64_t blocks = (X.n + (int64_t)kWavesPerBlock * SPW - 1) / ((int64_t)kWavesPerBlock * SPW);
  const int64_t cap = (int64_t)ctx->n_cu * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((k_fm_predict_orders<LT, SPLIT>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, X, M, Pf, lgL, out);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

template <int LT>
static int launch_fm_predict_orders_LT(nfm_ctx* ctx, const CsrView& X, const ModelView& M, const double* Pf, int lgL, double* out) {
  constexpr int R = kWave / LT;
  int split = choose_split(LT, X.n, X.n > 0 ? (double)X.nnz / (double)X.n + M.n_aug : 0.0, ctx->n_cu);
  // two row slots per sample even when the chip is full: the degree >= 3 recursion is a dependent chain per entry, and
  // halving it measured 0.696 -> 0.718 of the roofline on cfg5 (four slots: 0.634)
  if (R >= 2 && split < 2 && !getenv("NFM_SPLIT")) split = 2;
  if constexpr (R >= 8) if (split >= 8) return launch_fm_predict_orders<LT, 8>(ctx, X, M, Pf, lgL, out);
  if constexpr (R >= 4) if (split >= 4) return launch_fm_predict_orders<LT, 4>(ctx, X, M, Pf, lgL, out);
  if constexpr (R >= 2) if (split >= 2) return launch_fm_predict_orders<LT, 2>(ctx, X, M, Pf, lgL, out);
  return launch_fm_predict_orders<LT, 1>(ctx, X, M, Pf, lgL, out);
}

// true: handled.  Used when the gathers outweigh the copy (at least two row visits per table row)
static int predict_orders(nfm_ctx* ctx, const CsrView& X, const ModelView& M, double* out, bool* handled) {
  *handled = false;
  static const bool on = !(getenv("NFM_PREDICT_ORDERS") && atoi(getenv("NFM_PREDICT_ORDERS")) == 0);
  if (!on || M.kind != NFM_KIND_FM || M.nb < 2 || M.kc != 1 || M.bs != M.da || M.rs != 1) return NFM_OK;
  int NBP = 1;
  while (NBP < M.nb) NBP <<= 1;
  const int LT = NBP * M.L;
  if (LT > kWave || X.nnz + (int64_t)M.n_aug * X.n < 2 * M.da) return NFM_OK;
  int lgL = 0;
  while ((1 << lgL) < M.L) ++lgL;
  // the interleaved table lives on the context: this call's kernels are still queued when it returns (see nfm_ctx::predict_pf)
  if (!ctx->predict_pf) ctx->predict_pf = new DevBuf();
  DevBuf& Pf = *ctx->predict_pf;
  const size_t pf_bytes = sizeof(double) * (size_t)M.da * NBP * M.Kp;
  if (!(Pf.p && pf_bytes <= Pf.bytes)) {
    NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (growing: an earlier call's kernels may still read the old block)
    NFM_TRY(Pf.alloc(pf_bytes));
  }
  TimedLaunch tl(ctx, "predict");
  {
    const int64_t total = M.da * NBP * (M.Kp / 2);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > (int64_t)ctx->n_cu * 16) blocks = (int64_t)ctx->n_cu * 16;
    hipLaunchKernelGGL(k_interleave_orders, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, M.P, Pf.as<double>(), M.nb, NBP, M.da, M.Kp);
  }
  int rc = NFM_OK;
  switch (LT) {
    case 2: rc = launch_fm_predict_orders_LT<2>(ctx, X, M, Pf.as<double>(), lgL, out); break;
    case 4: rc = launch_fm_predict_orders_LT<4>(ctx, X, M, Pf.as<double>(), lgL, out); break;
    case 8: rc = launch_fm_predict_orders_LT<8>(ctx, X, M, Pf.as<double>(), lgL, out); break;
    case 16: rc = launch_fm_predict_orders_LT<16>(ctx, X, M, Pf.as<double>(), lgL, out); break;
    case 32: rc = launch_fm_predict_orders_LT<32>(ctx, X, M, Pf.as<double>(), lgL, out); break;
    default: rc = launch_fm_predict_orders_LT<64>(ctx, X, M, Pf.as<double>(), lgL, out); break;
  }
  NFM_TRY(rc);
  *handled = true;
  return NFM_OK;
}

// FFM: yhat = b + sum w x + sum_{j1<j2} x1 x2 <P[f2][j1], P[f1][j2]>
// (model/field_aware_factorization_machine.nim:66-76).  One wavefront per sample: the m*m ordered
// pairs of the row are dealt to the R = 64/L row slots; a slot's L lanes form the dot product.
template <int L>
__global__ __launch_bounds__(kBlock) void k_ffm_predict(CsrView X, ModelView M, double* __restrict__ out) {
  constexpr int R = kWave / L;
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane / L, l = lane % L;
  const int64_t wave0 = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  const double sw = M.sc[SC_SCALE_W], b = M.sc[SC_INTERCEPT];
  const dev::PlainParams ps{M.P, M.sc[SC_SCALE_P]};
  for (int64_t i = wave0; i < X.n; i += nwaves) {
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    double lin = 0.0;
    for (int c = 0; c * kWave < m; ++c) {
      const int q = c * kWave + lane;
      if (q < m) lin += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
    }
    double acc = 0.0;
    const int64_t npairs = (int64_t)m * m;
    for (int64_t pb = 0; pb < npairs; pb += R) {
      const int64_t pr = pb + g;
      if (pr < npairs) {
        const int a = (int)(pr / m), c2 = (int)(pr % m);
        const int j1 = X.indices[q0 + a], j2 = X.indices[q0 + c2];
        if (j1 < j2) {
          const int f1 = X.fields[q0 + a], f2 = X.fields[q0 + c2];
          const double2 u = ps.load(M.row(f2, j1) * M.Kp + 2 * l);
          const double2 v = ps.load(M.row(f1, j2) * M.Kp + 2 * l);
          acc += (X.data[q0 + a] * X.data[q0 + c2]) * (u.x * v.x + u.y * v.y);
        }
      }
    }
    const double tot = dev::wave_sum(lin + acc);
    if (lane == 0) out[i] = b + tot;
  }
}

template <int L, int SPLIT>
static int launch_fm_predict(nfm_ctx* ctx, const CsrView& X, const ModelView& M, double* out) {
  constexpr int SPW = kWave / (L * SPLIT);
  int64_t blocks = (X.n + (int64_t)kWavesPerBlock * SPW - 1) / ((int64_t)kWavesPerBlock * SPW);
  const int64_t cap = (int64_t)ctx->n_cu * 32;
  if (blocks > cap) blocks = cap;
  TimedLaunch tl(ctx, "predict");
  hipLaunchKernelGGL((k_fm_predict<L, SPLIT>), dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, X, M, out);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

template <int L>
static int launch_predict_L(nfm_ctx* ctx, const CsrView& X, const ModelView& M, double* out) {
  if (X.n == 0) return NFM_OK;
  if (M.kind == NFM_KIND_FFM) {
    int64_t blocks = (X.n + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    TimedLaunch tl(ctx, "predict");
    hipLaunchKernelGGL(k_ffm_predict<L>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, X, M, out);
    NFM_HIP_CHECK(hipGetLastError());
    return NFM_OK;
  }
  constexpr int R = kWave / L;
  const int split = choose_split(L, X.n, X.n > 0 ? (double)X.nnz / (double)X.n + M.n_aug : 0.0, ctx->n_cu);
  if (R >= 16 && split >= 16) return launch_fm_predict<L, (R >= 16 ? 16 : R)>(ctx, X, M, out);
  if (R >= 8 && split >= 8) return launch_fm_predict<L, (R >= 8 ? 8 : R)>(ctx, X, M, out);
  if (R >= 4 && split >= 4) return launch_fm_predict<L, (R >= 4 ? 4 : R)>(ctx, X, M, out);
  if (R >= 2 && split >= 2) return launch_fm_predict<L, (R >= 2 ? 2 : R)>(ctx, X, M, out);
  return launch_fm_predict<L, 1>(ctx, X, M, out);
}

int launch_predict(nfm_ctx* ctx, const CsrView& X, const ModelView& M, double* out) {
  NFM_CHECK(M.Kp == 2 * M.L, NFM_ERR_UNSUPPORTED, "field-aware models: n_components > 128 is not supported by the wave-per-sample kernels");
  NFM_CHECK(M.kind == NFM_KIND_FFM || M.degree <= dev::kMaxDeg, NFM_ERR_UNSUPPORTED, "degree > %d unsupported", dev::kMaxDeg);
  if (X.n > 0) {
    bool handled = false;
    NFM_TRY(predict_orders(ctx, X, M, out, &handled));
    if (handled) return NFM_OK;
  }
  switch (M.L) {
    case 1: return launch_predict_L<1>(ctx, X, M, out);
    case 2: return launch_predict_L<2>(ctx, X, M, out);
    case 4: return launch_predict_L<4>(ctx, X, M, out);
    case 8: return launch_predict_L<8>(ctx, X, M, out);
    case 16: return launch_predict_L<16>(ctx, X, M, out);
    case 32: return launch_predict_L<32>(ctx, X, M, out);
    case 64: return launch_predict_L<64>(ctx, X, M, out);
  }
  return set_error(NFM_ERR_UNSUPPORTED, "n_components > 128 is not supported by the wave-per-sample kernels");
}

}  // namespace nfm
