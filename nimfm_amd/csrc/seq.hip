// nimfm_amd/csrc/seq.hip -- NFM_MODE_SEQUENTIAL: the reference's single-thread semantics on the GPU.
//
// One persistent workgroup walks the samples in the given order and performs the reference's
// per-sample step (SGD: optimizer/sgd.nim:246-258 = lazilyUpdate :134-143 + predictWithGrad
// :191-202 + update :205-243; AdaGrad: optimizer/adagrad.nim:169-184 = update :87-110 +
// predictWithGrad + updateG :113-134; FFM forward: optimizer/sgd_ffm.nim:11-30).  Parallelism is
// only across the latent factors: thread s owns factor s and walks the row's nnz IN STORAGE
// ORDER, so every per-factor sum has the reference's summation order.  This is the parity
// vehicle ("stochastic SGD compared at fixed seed / single thread"), not the throughput path:
// it is latency-bound by construction (one sample in flight).
//
// Lazy L2 scaling: the reference keeps scaling_P plus a per-feature snapshot scalings_P[j] and
// multiplies P[j] by their ratio on touch.  Here the stored tensor is P~ = P / scale_P with ONE
// global scale (sc[0]): the ratio trick collapses to "true value = scale * stored".  resetScaling
// (sgd.nim:116-131) becomes a dense multiply when the scale drops below 1e-9.
#include <stdlib.h>

#include "fm_device.h"
#include "opt_views.h"

namespace nfm {

struct SeqArgs {
  CsrView X;
  ModelView M;
  OptView O;
  const int64_t* perm;
  int64_t begin, end, it0;
  double* out;
  int m_cap;
};

constexpr int kSeqMaxDeg = 8;

// STAGE (FactorizationMachine): the sample's entries (index, value) and the parameter values every thread needs
// (P[blk][j_q][tid] for all q) are brought into LDS ONCE per step with independent loads; the three passes
// of the step (forward, derivative, update) then run from LDS.  Without it every pass re-reads index, value
// and parameter from global memory, one dependent round trip after the other (15 us per step at m = 16).
// The per-thread arithmetic and its order are unchanged.
template <int KIND, int OPT, bool STAGE>
__global__ void k_sequential(SeqArgs a) {
  extern __shared__ double lds[];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int T = blockDim.x, tid = threadIdx.x;
  const bool act = tid < M.Kp;
  double* red = lds;         // [T]
  double* dA = lds + T;      // [nb][m_cap][T]
  const size_t n_da = (size_t)(M.nb > 0 ? M.nb : 1) * a.m_cap * T;
  double* Pl = dA + n_da;                                        // STAGE: [nb][m_cap][T] stored parameter values
  double* vl = Pl + (STAGE ? n_da : 0);                          // STAGE: [m_cap] values
  double* wl = vl + (STAGE ? a.m_cap : 0);                       // STAGE: [m_cap] linear weights (stored values)
  int64_t* jl = reinterpret_cast<int64_t*>(wl + (STAGE ? a.m_cap : 0));  // STAGE: [m_cap] indices
  const int Kp = M.Kp, nb = M.nb, k = M.k;
  const int n_aug = (KIND == NFM_KIND_FM) ? M.n_aug : 0;
  double sP = M.sc[SC_SCALE_P], sw = M.sc[SC_SCALE_W], b = M.sc[SC_INTERCEPT];
  double gsb = 0.0, gnb = 0.0;
  if (OPT == OPT_ADAGRAD) {
    gsb = O.gsc[0];
    gnb = O.gsc[1];
  }
  double loss_acc = 0.0, viol_acc = 0.0;
  int64_t it = a.it0;

  for (int64_t pos = a.begin; pos < a.end; ++pos, ++it) {
    const int64_t i = a.perm ? a.perm[pos] : pos;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int m_tot = m + n_aug;
    const double y = dev::target_of(X.y[i], M.task);
    const double itf = (double)it;

    if (STAGE) {
      __syncthreads();  // the previous step is done with the staged row
      for (int q = tid; q < m_tot; q += T) {
        const int64_t jq = q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m);
        jl[q] = jq;
        vl[q] = q < m ? X.data[q0 + q] : 1.0;
        // (AdaGrad re-derives w in update() below and refreshes wl there)
        wl[q] = q < m ? M.w[jq] : 0.0;
      }
      __syncthreads();
    }

    if (OPT == OPT_ADAGRAD && it != 1) {
      // update(): optimizer/adagrad.nim:87-110, fit_linear.nim:50-57
      const double itp = (double)(it - 1);
      const double tmp = O.eta0 * itp * O.beta;
      if (act)
        for (int blk = 0; blk < nb; ++blk)
          for (int q = 0; q < m_tot; ++q) {
            const int64_t j = STAGE ? jl[q] : (q < m ? X.indices[q0 + q] : X.d + (q - m));
            const size_t e = M.row(blk, j) * Kp + tid;
            const double old = M.P[e];
            const double nw = dev::adagrad_param(O.G[e], O.N[e], O.eta0, tmp);
            viol_acc += fabs(old - nw);
            M.P[e] = nw;
            if (STAGE) Pl[((size_t)blk * a.m_cap + q) * T + tid] = nw;
          }
      if (M.fit_intercept) {
        const double old = b;
        const double denom = sqrt(gnb) + O.eta0 * itp * O.alpha0;
        b = -O.eta0 * gsb / denom;
        if (tid == 0) viol_acc += fabs(old - b);
      }
      if (M.fit_linear) {
        const double denom = itp * O.eta0 * O.alpha;
        for (int q = tid; q < m; q += T) {
          const int j = X.indices[q0 + q];
          const double wj = M.w[j];
          const double nw = -O.eta0 * O.Gw[j] / (denom + sqrt(O.Nw[j]));
          M.w[j] = nw;
          if (STAGE) wl[q] = nw;
          viol_acc += fabs(wj - nw);
        }
        __syncthreads();
      }
    }

    if (STAGE && act && !(OPT == OPT_ADAGRAD && it != 1)) {
      for (int blk = 0; blk < nb; ++blk) {
        int q = 0;
        for (; q + 4 <= m_tot; q += 4) {  // four independent loads in flight
          double t4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) t4[u] = M.P[M.row(blk, jl[q + u]) * Kp + tid];
#pragma unroll
          for (int u = 0; u < 4; ++u) Pl[((size_t)blk * a.m_cap + q + u) * T + tid] = t4[u];
        }
        for (; q < m_tot; ++q) Pl[((size_t)blk * a.m_cap + q) * T + tid] = M.P[M.row(blk, jl[q]) * Kp + tid];
      }
    }

    // ---- predictWithGrad (optimizer/sgd.nim:191-202 / sgd_ffm.nim:11-30) ----
    double yh = b;
    if (STAGE) {
      for (int q = 0; q < m; ++q) yh += (sw * wl[q]) * vl[q];
    } else {
      for (int q = 0; q < m; ++q) yh += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
    }
    if (KIND == NFM_KIND_FM) {
      for (int o = 0; o < nb; ++o) {
        const int deg = M.degree - o;
        double A[kSeqMaxDeg + 1];
        double kv = 0.0;
        const size_t blk = M.row(o, 0) * Kp, rstride = (size_t)M.rs * Kp;
        if (act) {
          if (deg != 2) {  // sgd.nim:152-159
            A[0] = 1.0;
#pragma unroll
            for (int t = 1; t <= kSeqMaxDeg; ++t) A[t] = 0.0;
            for (int q = 0; q < m_tot; ++q) {
              const int64_t j = STAGE ? 0 : (q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m));
              const double val = STAGE ? vl[q] : (q < m ? X.data[q0 + q] : 1.0);
              const double p = sP * (STAGE ? Pl[((size_t)o * a.m_cap + q) * T + tid] : M.P[blk + (size_t)j * rstride + tid]);
#pragma unroll
              for (int t = kSeqMaxDeg; t >= 1; --t)
                if (t <= deg) A[t] += A[t - 1] * p * val;
            }
            kv = 0.0;
#pragma unroll
            for (int t = 1; t <= kSeqMaxDeg; ++t)
              if (t == deg) kv = A[t];
          } else {  // sgd.nim:160-170
            double a1 = 0.0, a2 = 0.0;
            for (int q = 0; q < m_tot; ++q) {
              const int64_t j = STAGE ? 0 : (q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m));
              const double val = STAGE ? vl[q] : (q < m ? X.data[q0 + q] : 1.0);
              const double p = sP * (STAGE ? Pl[((size_t)o * a.m_cap + q) * T + tid] : M.P[blk + (size_t)j * rstride + tid]);
              a1 += val * p;
              a2 += (val * p) * (val * p);
            }
            A[0] = 1.0;
            A[1] = a1;
            kv = (a1 * a1 - a2) / 2;
          }
          // computeAnovaDerivative: sgd.nim:176-188
          for (int q = 0; q < m_tot; ++q) {
            const int64_t j = STAGE ? 0 : (q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m));
            const double val = STAGE ? vl[q] : (q < m ? X.data[q0 + q] : 1.0);
            const double p = sP * (STAGE ? Pl[((size_t)o * a.m_cap + q) * T + tid] : M.P[blk + (size_t)j * rstride + tid]);
            double d_;
            if (deg != 2) {
              d_ = val;
#pragma unroll
              for (int t = 1; t < kSeqMaxDeg; ++t)
                if (t < deg) d_ = val * (A[t] - p * d_);
            } else {
              d_ = val * (A[1] - p * val);
            }
            dA[((size_t)o * a.m_cap + q) * T + tid] = d_;
          }
        }
        red[tid] = (act && tid < k) ? kv : 0.0;
        __syncthreads();
        double tot = 0.0;
        for (int s = 0; s < k; ++s) tot += red[s];  // sgd.nim:172-173, ascending s
        yh += tot;
        __syncthreads();
      }
    } else {
      double part = 0.0;
      if (act) {
        for (int f = 0; f < nb; ++f)
          for (int q = 0; q < m; ++q) dA[((size_t)f * a.m_cap + q) * T + tid] = 0.0;
        for (int q1 = 0; q1 < m; ++q1)
          for (int q2 = 0; q2 < m; ++q2) {
            const int j1 = X.indices[q0 + q1], j2 = X.indices[q0 + q2];
            if (j1 < j2) {
              const int f1 = X.fields[q0 + q1], f2 = X.fields[q0 + q2];
              const double v12 = X.data[q0 + q1] * X.data[q0 + q2];
              const double pa = sP * M.P[M.row(f2, j1) * Kp + tid];
              const double pb = sP * M.P[M.row(f1, j2) * Kp + tid];
              part += (pa * pb) * v12;
              dA[((size_t)f2 * a.m_cap + q1) * T + tid] += v12 * pb;
              dA[((size_t)f1 * a.m_cap + q2) * T + tid] += v12 * pa;
            }
          }
      }
      red[tid] = (act && tid < k) ? part : 0.0;
      __syncthreads();
      double tot = 0.0;
      for (int s = 0; s < k; ++s) tot += red[s];
      yh += tot;
      __syncthreads();
    }

    if (tid == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);

    if (OPT == OPT_SGD) {
      // update(): optimizer/sgd.nim:205-243, fit_linear.nim:41-47
      const double eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      const double eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      const double sPn = sP * (1 - eta_P * O.beta);
      if (act)
        for (int blk = 0; blk < nb; ++blk)
          for (int q = 0; q < m_tot; ++q) {
            const int64_t j = STAGE ? jl[q] : (q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m));
            const size_t e = M.row(blk, j) * Kp + tid;
            const double p = sP * (STAGE ? Pl[((size_t)blk * a.m_cap + q) * T + tid] : M.P[e]);
            const double update = eta_P * (dL * dA[((size_t)blk * a.m_cap + q) * T + tid] + O.beta * p);
            viol_acc += fabs(update);
            M.P[e] = (p - update) / sPn;
          }
      sP = sPn;
      if (M.fit_intercept) {
        const double update = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf) * (dL + O.alpha0 * b);
        if (tid == 0) viol_acc += fabs(update);
        b -= update;
      }
      if (M.fit_linear) {
        const double swn = sw * (1 - eta_w * O.alpha);
        for (int q = tid; q < m; q += T) {
          const int64_t j = STAGE ? jl[q] : (int64_t)X.indices[q0 + q];
          const double wj = sw * (STAGE ? wl[q] : M.w[j]);
          const double update = eta_w * (dL * (STAGE ? vl[q] : X.data[q0 + q]) + O.alpha * wj);
          viol_acc += fabs(update);
          M.w[j] = (wj - update) / swn;
        }
        sw = swn;
      }
      // resetScaling: sgd.nim:116-131
      if (sP < 1e-9) {
        if (act)
          for (int64_t r = 0; r < (int64_t)nb * M.da; ++r) M.P[(size_t)r * Kp + tid] *= sP;
        sP = 1.0;
      }
      if (M.fit_linear && sw < 1e-9) {
        __syncthreads();
        for (int64_t j = tid; j < M.d; j += T) M.w[j] *= sw;
        sw = 1.0;
      }
      __syncthreads();
    } else {
      // updateG(): optimizer/adagrad.nim:113-134
      if (act)
        for (int blk = 0; blk < nb; ++blk)
          for (int q = 0; q < m_tot; ++q) {
            const int64_t j = STAGE ? jl[q] : (q < m ? (int64_t)X.indices[q0 + q] : X.d + (q - m));
            const size_t e = M.row(blk, j) * Kp + tid;
            const double grad = dL * dA[((size_t)blk * a.m_cap + q) * T + tid];
            O.G[e] += grad;
            O.N[e] += grad * grad;
          }
      if (M.fit_intercept) {
        gsb += dL;
        gnb += dL * dL;
      }
      if (M.fit_linear)
        for (int q = tid; q < m; q += T) {
          const int j = X.indices[q0 + q];
          const double g = dL * X.data[q0 + q];
          O.Gw[j] += g;
          O.Nw[j] += g * g;
        }
      __syncthreads();
    }
  }

  // write back scalars, reduce the running sums in a fixed order
  if (tid == 0) {
    M.sc[SC_SCALE_P] = sP;
    M.sc[SC_SCALE_W] = sw;
    M.sc[SC_INTERCEPT] = b;
    if (OPT == OPT_ADAGRAD) {
      O.gsc[0] = gsb;
      O.gsc[1] = gnb;
    }
  }
  __syncthreads();
  red[tid] = viol_acc;
  __syncthreads();
  if (tid == 0) {
    double v = 0.0;
    for (int t = 0; t < T; ++t) v += red[t];
    a.out[0] = loss_acc;
    a.out[1] = v;
  }
}

template <int KIND, int OPT, bool STAGE>
static int launch_seq_t(nfm_ctx* ctx, const SeqArgs& a, int T, size_t lds_bytes) {
  auto kern = k_sequential<KIND, OPT, STAGE>;
  if (lds_bytes > 64 * 1024)
    NFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  TimedLaunch tl(ctx, "sequential");
  hipLaunchKernelGGL(kern, dim3(1), dim3(T), lds_bytes, ctx->stream, a);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

int launch_sequential(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O,
                      const int64_t* perm_dev, int64_t begin, int64_t end, int64_t it0, int m_cap, double* out2_dev) {
  NFM_CHECK(M.Kp <= 1024, NFM_ERR_UNSUPPORTED, "sequential mode supports n_components <= 1024");
  NFM_CHECK(M.kind == NFM_KIND_FFM || M.degree <= kSeqMaxDeg, NFM_ERR_UNSUPPORTED, "degree > %d unsupported", kSeqMaxDeg);
  int T = ((M.Kp + kWave - 1) / kWave) * kWave;
  if (T < kWave) T = kWave;
  if (m_cap < 1) m_cap = 1;
  const size_t lds_bytes = sizeof(double) * ((size_t)T + (size_t)(M.nb > 0 ? M.nb : 1) * m_cap * T);
  NFM_CHECK(lds_bytes <= 160 * 1024, NFM_ERR_UNSUPPORTED,
            "sequential mode needs %zu bytes of LDS for the per-sample gradient (n_blocks=%d, max row nnz=%d, Kp=%d)",
            lds_bytes, M.nb, m_cap, M.Kp);
  SeqArgs a{X, M, O, perm_dev, begin, end, it0, out2_dev, m_cap};
  if (M.kind == NFM_KIND_FM) {
    // staged step (entries and parameter values in LDS) when the extra [nb][m_cap][T] + 2 [m_cap] doubles fit
    const size_t n_da = (size_t)(M.nb > 0 ? M.nb : 1) * m_cap * T;
    const size_t staged_bytes = lds_bytes + sizeof(double) * (n_da + 3 * (size_t)m_cap);
    static const bool stage_on = !(getenv("NFM_SEQ_STAGE") && atoi(getenv("NFM_SEQ_STAGE")) == 0);
    if (stage_on && M.nb > 0 && staged_bytes <= 160 * 1024) {
      if (opt_kind == OPT_SGD) return launch_seq_t<NFM_KIND_FM, OPT_SGD, true>(ctx, a, T, staged_bytes);
      return launch_seq_t<NFM_KIND_FM, OPT_ADAGRAD, true>(ctx, a, T, staged_bytes);
    }
    if (opt_kind == OPT_SGD) return launch_seq_t<NFM_KIND_FM, OPT_SGD, false>(ctx, a, T, lds_bytes);
    return launch_seq_t<NFM_KIND_FM, OPT_ADAGRAD, false>(ctx, a, T, lds_bytes);
  }
  if (opt_kind == OPT_SGD) return launch_seq_t<NFM_KIND_FFM, OPT_SGD, false>(ctx, a, T, lds_bytes);
  return launch_seq_t<NFM_KIND_FFM, OPT_ADAGRAD, false>(ctx, a, T, lds_bytes);
}

// finalize (optimizer/adagrad.nim:65-84): every parameter from the state, it' = it - 1
__global__ void k_adagrad_finalize(ModelView M, OptView O, double itp) {
  const int64_t nP2 = (int64_t)M.nb * M.da * M.Kp / 2;
  const double tmp = O.eta0 * itp * O.beta;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t e = t0; e < nP2; e += stride) {
    const double2 g = reinterpret_cast<const double2*>(O.G)[e];
    const double2 n = reinterpret_cast<const double2*>(O.N)[e];
    double2 v;
    v.x = -O.eta0 * g.x;
    v.x /= tmp + sqrt(n.x);
    v.y = -O.eta0 * g.y;
    v.y /= tmp + sqrt(n.y);
    reinterpret_cast<double2*>(M.P)[e] = v;
  }
  if (M.fit_linear) {
    const double den = O.eta0 * itp * O.alpha;
    for (int64_t j = t0; j < M.d; j += stride) {
      double v = -O.eta0 * O.Gw[j];
      v /= den + sqrt(O.Nw[j]);
      M.w[j] = v;
    }
  }
  if (t0 == 0) {
    if (M.fit_intercept) {
      const double den = sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0;
      M.sc[SC_INTERCEPT] = -O.eta0 * O.gsc[0] / den;
    }
    M.sc[SC_SCALE_P] = 1.0;
    M.sc[SC_SCALE_W] = 1.0;
  }
}

int launch_adagrad_finalize(nfm_ctx* ctx, const ModelView& M, const OptView& O, int64_t it) {
  const int64_t nP2 = (int64_t)M.nb * M.da * M.Kp / 2;
  int64_t work = nP2 > M.d ? nP2 : M.d;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  TimedLaunch tl(ctx, "adagrad_finalize");
  hipLaunchKernelGGL(k_adagrad_finalize, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, M, O, (double)(it - 1));
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

}  // namespace nfm
