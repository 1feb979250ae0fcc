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
  double* dA_global;  // the one-sample-in-flight kernel's per-sample gradient [nb][m_cap][T] when it does not fit the LDS, else null
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
  // [nb][m_cap][T]; thread tid only ever touches its own column (.. + tid), so the table may as well live in global memory
  // (rows of hundreds of entries: dataset.nim puts no bound on a row) -- program order of one thread is all it needs
  double* dA = a.dA_global ? a.dA_global : lds + T;
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
      double tot = 0.0;
      for (int o = 0; o < nb; ++o) {
        const int deg = M.deg_of(o);
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
        if (o % M.kc == 0) tot = 0.0;  // (more than 128 factors: the kc blocks of one order continue ONE ascending sum)
        for (int s = 0; s < k; ++s) tot += red[s];  // sgd.nim:172-173, ascending s
        if (o % M.kc == M.kc - 1) yh += tot;
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

// ------------------------------------------------------------------------------------------------
// Pipelined step (FactorizationMachine, degree 2, one order): the arithmetic of k_sequential<.., STAGE = true>, every
// sum in the same order -- what changes is WHEN things are requested and WHO does the order-independent work.
//
// 1. A step of the staged kernel is a chain of dependent round trips (order -> indptr -> entries -> linear weights /
//    parameter rows -> compute -> stores).  Here, while sample t is computed from LDS, the memory system works on the
//    samples behind it:
//        t+4: its id in the order      t+3: row bounds and target      t+2: its entries (registers, to LDS at the step's end)
//        t+1: its parameter values (AdaGrad: state and last stored parameters) and linear terms -> registers
//    Sample t+1 may share features with t, whose update lands after those values were requested.  Every thread therefore
//    leaves what it writes for entry q (new stored value; AdaGrad: new g_sum and g_norm) in the LDS slot of q as well, and
//    a map "entry of t+1 -> entry of t" (every thread looks one entry up in t's list) says which registers to refresh
//    from there before t+1 uses them: a sample sees exactly the values of the one-at-a-time order.  A dense rescale
//    (resetScaling, sgd.nim:116-131) reloads the registers instead.
// 2. With one wavefront and thread = factor, a step was bound by the ~4500 instructions it ISSUES (32 divisions, three
//    pow(), address arithmetic per entry), not by memory.  The workgroup is 256 threads = G groups x S factors: every
//    group runs the per-factor sums over ALL entries (their order is the reference's; the groups agree bit for bit),
//    but the work that is independent per entry -- requesting rows, the derivative, the update with its division, the
//    AdaGrad parameter with its square root -- is split over the groups (entry c belongs to group c mod G).
// 3. Only one barrier per step waits for memory (it stands where nothing but already-issued stores is outstanding); the
//    others order LDS only: __syncthreads() waits for every outstanding global load and would put the hidden latency
//    back on the critical path.  Loops read 8 LDS values at a time through clamped (not guarded) indices: a guarded
//    load is a branch with a wait of its own.
// ------------------------------------------------------------------------------------------------
// LDS-only barrier: __syncthreads() also waits for every outstanding global load of the wavefront
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kPipeThreads = 256;
constexpr int kPipeThreadsWide = 512;   // when a thread of 256 would hold 8 or more rows (1024 threads: 128 registers, spills: slower)
constexpr int kPipePairCap = 4096;  // FFM: entries^2 of the per-pair table (32 KB)

// KIND = FactorizationMachine (degree 2, one order): a sample's rows are its entries.  KIND = FieldAwareFactorizationMachine:
// the reference's step reads and updates ALL nFields rows of every feature of the sample (sgd_ffm.nim:11-30 and the shared
// update loops): the rows are "slots" c = field * m + q.  A slot belongs to group c mod G like an FM entry.
template <int KIND, int OPT, int RC, int T>  // RC: slots per thread (RC * G >= rows per sample), T: threads (256 or 1024)
__global__ __launch_bounds__(T) void k_sequential_pipe(SeqArgs a, int S, int lgS, int split_terms) {
  extern __shared__ double lds[];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool FFM = KIND == NFM_KIND_FFM;
  constexpr int CH = RC < 8 ? RC : 8;  // values requested together in the per-thread loops
  constexpr int FH = 8;                // ... in the loops over all entries
  const int tid = threadIdx.x;
  const int G = T >> lgS, g = tid >> lgS, s = tid & (S - 1);
  const int Kp = M.Kp, k = M.k, n_aug = FFM ? 0 : M.n_aug, nbk = FFM ? M.nb : 1;
  const bool act = s < Kp;
  const int mc = a.m_cap;
  double* red = lds;                                    // [T]
  double* dA = red + T;                                 // [nbk mc][S] derivative, then what the step wrote for the slot
  double* Pl = dA + (size_t)nbk * mc * S;               // [nbk mc][S] stored parameter values of the current sample (AdaGrad: then new g_norm)
  double* vl = Pl + (size_t)nbk * mc * S;               // [3][mc] values of samples t, t+1, t+2 (buffer = step mod 3)
  double* wl = vl + 3 * mc;                             // [mc] stored linear weights of the current sample
  double* wp = wl + mc;                                 // [mc] written linear weight (AdaGrad: g_sum of the linear term)
  double* wp2 = wp + mc;                                // [mc] AdaGrad: g_norm of the linear term
  const bool by_pair = FFM && mc * mc <= kPipePairCap;  // FFM: one thread per ordered pair of entries (below)
  double* pc = wp2 + mc;                                // FFM, by_pair: [mc][mc] a pair's term of the prediction
  double* yt = pc + (by_pair ? mc * mc : 0);            // [mc] terms of the linear part of the prediction
  double* T1 = yt + mc;                                 // FM, split_terms: [mc][S] x_q p_qs, the terms of the per-factor sums
  int64_t* jl = reinterpret_cast<int64_t*>(T1 + (!FFM && split_terms ? (size_t)mc * S : 0));  // [3][mc] indices
  int* rm = reinterpret_cast<int*>(jl + 3 * mc);        // [mc] entry of t+1 -> entry of t with the same feature, or -1
  int* fl = rm + mc;                                    // FFM: [3][mc] fields
  int* fcnt = fl + 3 * mc;                              // FFM: [nbk] entries of the current sample per field
  int* fent = fcnt + nbk;                               // FFM: [nbk][mc] ... which ones, ascending
  double sP = M.sc[SC_SCALE_P], sw = M.sc[SC_SCALE_W], b = M.sc[SC_INTERCEPT];
  double gsb = 0.0, gnb = 0.0;
  if (OPT == OPT_ADAGRAD) {
    gsb = O.gsc[0];
    gnb = O.gsc[1];
  }
  double loss_acc = 0.0, viol_acc = 0.0;
  int64_t it = a.it0;

  struct Desc {
    int64_t q0;
    int m;
    double y;
  };
  auto sample_at = [&](int64_t pos) -> int64_t { return pos < a.end ? (a.perm ? a.perm[pos] : pos) : 0; };
  auto load_desc = [&](int64_t pos, int64_t i) {
    Desc d{0, 0, 0.0};
    if (pos < a.end) {
      d.q0 = X.indptr[i];
      d.m = (int)(X.indptr[i + 1] - d.q0);
      d.y = dev::target_of(X.y[i], M.task);
    }
    return d;
  };
  // entries of a sample: thread q holds entry q (m_cap <= T)
  int64_t je = 0;
  double ve = 0.0;
  int fe = 0;
  auto load_entries = [&](const Desc& d, bool live) {
    je = 0;
    ve = 0.0;
    fe = 0;
    if (live && tid < d.m + n_aug) {
      je = tid < d.m ? (int64_t)X.indices[d.q0 + tid] : X.d + (tid - d.m);
      ve = tid < d.m ? X.data[d.q0 + tid] : 1.0;
      if (FFM) fe = X.fields[d.q0 + tid];
    }
  };
  auto store_entries = [&](int buf, const Desc& d, bool live) {
    if (live && tid < d.m + n_aug) {
      jl[buf * mc + tid] = je;
      vl[buf * mc + tid] = ve;
      if (FFM) fl[buf * mc + tid] = fe;
    }
  };
  // the row data of a sample, requested one step ahead: this thread's slots are c = u G + g, c = blk * m_tot + q;
  // (blk << 16 | q) of every slot is kept in a register (ns: of the sample being requested, cs: of the current one)
  double Pr[RC], Gr[OPT == OPT_ADAGRAD ? RC : 1], Nr[OPT == OPT_ADAGRAD ? RC : 1];
  int cs[RC], ns[RC];
  double wr = 0.0, gwr = 0.0, nwr = 0.0;
  auto load_rows = [&](int buf, const Desc& d, bool live) {
    const int mt = live ? d.m + n_aug : 0;
    const int nsl = nbk * mt;
    if (act && g < nsl) {
      // straight-line code: slots past the end repeat the first slot of their group of CH (past the end itself: the
      // thread's first slot).  Repeating work is harmless as long as a repeat reads what the original read: all reads of
      // a group of CH come before its writes, and the update loops leave at the first group that starts past the end.
#pragma unroll
      for (int ub = 0; ub < RC; ub += CH) {
        int64_t j_[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int c0 = (ub + u) * G + g, cf = ub * G + g, c = c0 < nsl ? c0 : (cf < nsl ? cf : g);
          const int blk = FFM ? c / mt : 0, q = FFM ? c - blk * mt : c;
          ns[ub + u] = (blk << 16) | q;
          j_[u] = jl[buf * mc + q];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const size_t e = M.row(ns[ub + u] >> 16, j_[u]) * Kp + s;
          Pr[ub + u] = M.P[e];
          if constexpr (OPT == OPT_ADAGRAD) {
            Gr[ub + u] = O.G[e];
            Nr[ub + u] = O.N[e];
          }
        }
      }
    }
    wr = gwr = nwr = 0.0;
    if (live && tid < d.m) {
      const int64_t j = jl[buf * mc + tid];
      wr = M.w[j];
      if (OPT == OPT_ADAGRAD && M.fit_linear) {
        gwr = O.Gw[j];
        nwr = O.Nw[j];
      }
    }
  };

  // ---- prologue: descriptors of the first three samples, entries of the first two, rows of the first ----
  Desc D0 = load_desc(a.begin, sample_at(a.begin));
  Desc D1 = load_desc(a.begin + 1, sample_at(a.begin + 1));
  Desc D2 = load_desc(a.begin + 2, sample_at(a.begin + 2));
  int64_t i3 = sample_at(a.begin + 3);
  load_entries(D0, true);
  store_entries(0, D0, true);
  load_entries(D1, a.begin + 1 < a.end);
  store_entries(1, D1, a.begin + 1 < a.end);
  for (int q = tid; q < mc; q += T) rm[q] = -1;
  __syncthreads();
  load_rows(0, D0, true);
  bool fresh = true;  // the registers hold what memory holds now: nothing to refresh
  int m_prev = 0;     // entries (incl. dummies) of the previous sample: its slot (blk, r) is blk * m_prev + r

  for (int64_t pos = a.begin; pos < a.end; ++pos, ++it) {
    const int s0 = (int)((pos - a.begin) % 3), s1 = (s0 + 1) % 3, s2 = (s0 + 2) % 3;
    const int m = D0.m, m_tot = m + n_aug, n_slots = nbk * m_tot;
    const double y = D0.y;
    const double itf = (double)it;
    const int64_t* jc = jl + s0 * mc;
    const double* vc = vl + s0 * mc;
    const int* fc = fl + s0 * mc;
    const bool live1 = pos + 1 < a.end, live2 = pos + 2 < a.end;
    const bool mine = act && g < n_slots;  // this thread has rows of the current sample
#pragma unroll
    for (int u = 0; u < RC; ++u) cs[u] = ns[u];
    // the slot of register u (past the end: this thread's first slot, whose work is then simply done again)
    auto slot_of = [&](int u) {
      const int c = u * G + g, cf = (u / CH * CH) * G + g;
      return c < n_slots ? c : (cf < n_slots ? cf : g);
    };
    auto reg_of = [&](int u) {  // the register that holds slot_of(u)'s values
      const int c = u * G + g, cf = (u / CH * CH) * G + g;
      return c < n_slots ? u : (cf < n_slots ? u / CH * CH : 0);
    };

    // ---- 1. refresh what the previous step wrote, then the rows go to LDS ----
    if (!fresh) {
      if (mine) {
#pragma unroll
        for (int ub = 0; ub < RC; ub += CH) {
          int r_[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) r_[u] = rm[cs[ub + u] & 0xFFFF];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int r = (ub + u) * G + g < n_slots ? r_[u] : -1;
            if (r >= 0) {  // rare: a feature shared by consecutive samples
              const size_t src = ((size_t)(cs[ub + u] >> 16) * m_prev + r) * S + s;
              if constexpr (OPT == OPT_ADAGRAD) {
                Gr[ub + u] = dA[src];
                Nr[ub + u] = Pl[src];
              } else {
                Pr[ub + u] = dA[src];
              }
            }
          }
        }
      }
      if (tid < m) {
        const int r = rm[tid];
        if (r >= 0) {
          if (OPT == OPT_ADAGRAD) {
            wr = wl[r];  // what update() of the previous step stored for this feature
            gwr = wp[r];
            nwr = wp2[r];
          } else {
            wr = wp[r];
          }
        }
      }
      lds_barrier();  // every thread is done with the previous step's slots
    }
    fresh = false;
    if (tid < mc) rm[tid] = -1;  // refilled in step 3, read by the next step's refresh
    bool staged = false;
    if constexpr (OPT == OPT_ADAGRAD) if (it != 1) {
      staged = true;
      // update(): optimizer/adagrad.nim:87-110, fit_linear.nim:50-57
      const double itp = (double)(it - 1);
      const double tmp = O.eta0 * itp * O.beta;
      if (mine) {
        // branch-free: the square root / division chains of several rows interleave; slots past the end repeat this
        // thread's first slot -- the same value stored again, nothing added to viol
#pragma unroll
        for (int ub = 0; ub < RC; ub += CH) {
          int64_t j_[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) j_[u] = jc[cs[ub + u] & 0xFFFF];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const bool ok = (ub + u) * G + g < n_slots;
            const size_t e = M.row(cs[ub + u] >> 16, j_[u]) * Kp + s;
            const int ru = reg_of(ub + u) == ub + u ? ub + u : (reg_of(ub + u) == ub ? ub : 0);  // static candidates only
            const double nw = dev::adagrad_param(ru == ub + u ? Gr[ub + u] : (ru == ub ? Gr[ub] : Gr[0]),
                                                 ru == ub + u ? Nr[ub + u] : (ru == ub ? Nr[ub] : Nr[0]), O.eta0, tmp);
            const double dv = fabs((ru == ub + u ? Pr[ub + u] : (ru == ub ? Pr[ub] : Pr[0])) - nw);
            viol_acc = ok ? viol_acc + dv : viol_acc;
            M.P[e] = nw;
            Pl[(size_t)slot_of(ub + u) * S + s] = nw;
          }
        }
      }
      if (M.fit_intercept) {
        const double old = b;
        const double denom = sqrt(gnb) + O.eta0 * itp * O.alpha0;
        b = -O.eta0 * gsb / denom;
        if (tid == 0) viol_acc += fabs(old - b);
      }
      if (tid < m) {
        double nw = wr;
        if (M.fit_linear) {
          const double denom = itp * O.eta0 * O.alpha;
          nw = -O.eta0 * gwr / (denom + sqrt(nwr));
          M.w[jc[tid]] = nw;
          viol_acc += fabs(wr - nw);
        }
        wl[tid] = nw;
      }
    }
    if (!staged) {
      if (mine) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
          const int ru = reg_of(u);
          Pl[(size_t)slot_of(u) * S + s] = ru == u ? Pr[u] : (ru == u / CH * CH ? Pr[u / CH * CH] : Pr[0]);
        }
      }
      if (tid < m) wl[tid] = wr;
    }
    if (FFM) {  // the current sample's entries by field, ascending (sgd_ffm.nim:24-30 walks a field's entries in this order)
      for (int f = tid; f < nbk; f += T) fcnt[f] = 0;
    }
    // The ONE full barrier of a step (it also waits for this workgroup's outstanding memory operations): it stands where
    // nothing but already-issued stores is outstanding.
    __syncthreads();
    if (FFM && tid < m) {
      const int f = fc[tid];
      int before = 0, total = 0;
      for (int qb = 0; qb < m; qb += FH) {
        int f_[FH];
#pragma unroll
        for (int u = 0; u < FH; ++u) f_[u] = fc[qb + u < m ? qb + u : qb];
#pragma unroll
        for (int u = 0; u < FH; ++u) {
          const bool same = qb + u < m && f_[u] == f;
          total += same ? 1 : 0;
          before += (same && qb + u < tid) ? 1 : 0;
        }
      }
      fent[f * mc + before] = tid;
      if (before == total - 1) fcnt[f] = total;
    }

    // ---- 2. requests for the samples behind this one (after update()'s stores: a shared row is read as rewritten) ----
    load_rows(s1, D1, live1);
    load_entries(D2, live2);
    const Desc D3 = load_desc(pos + 3, i3);
    const int64_t i4 = sample_at(pos + 4);
    // ---- 3. entry of the next sample -> entry of this one: the (entry, entry) comparisons are spread over all threads
    // (ids are distinct inside a row: at most one match per entry, no write conflicts) ----
    {
      const int mt1 = live1 ? D1.m + n_aug : 0;
      const int64_t* jn = jl + s1 * mc;
      int lg = 0;
      while ((1 << lg) < mt1) ++lg;
      const int c = tid & ((1 << lg) - 1), part = tid >> lg, nparts = T >> lg;
      const int len = (m_tot + nparts - 1) / nparts;
      if (c < mt1) {
        const int64_t j = jn[c];
        const int q_lo = part * len, q_hi = q_lo + len < m_tot ? q_lo + len : m_tot;
        for (int qb = q_lo; qb < q_hi; qb += FH) {
          int64_t jj[FH];
#pragma unroll
          for (int u = 0; u < FH; ++u) jj[u] = jc[qb + u < q_hi ? qb + u : qb];
          int r = -1;
#pragma unroll
          for (int u = 0; u < FH; ++u) r = (qb + u < q_hi && jj[u] == j) ? qb + u : r;
          if (r >= 0) rm[c] = r;
        }
      }
    }
    // the terms of the linear part, one entry per thread; FM with split_terms: x_q p_qs of this thread's entries
    if (tid < m) yt[tid] = (sw * wl[tid]) * vc[tid];
    if constexpr (!FFM) {
      if (split_terms && mine) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
          const int cc = slot_of(u);
          T1[(size_t)cc * S + s] = vc[cc] * (sP * Pl[(size_t)cc * S + s]);
        }
      }
    }
    lds_barrier();

    // ---- 4. predictWithGrad (optimizer/sgd.nim:191-202, sgd_ffm.nim:11-30) ----
    // every sum below adds its terms in the reference's order; the products were formed above, in parallel
    double yh = b;
    for (int qb = 0; qb < m; qb += FH) {
      double t_[FH];
#pragma unroll
      for (int u = 0; u < FH; ++u) t_[u] = yt[qb + u < m ? qb + u : qb];
#pragma unroll
      for (int u = 0; u < FH; ++u) yh = qb + u < m ? yh + t_[u] : yh;
    }
    double kv = 0.0;
    if constexpr (!FFM) {
      // every group runs the per-factor sums over all entries; the derivative of a thread's own entries follows
      double a1 = 0.0, a2 = 0.0;
      if (act && split_terms) {
        for (int qb = 0; qb < m_tot; qb += FH) {
          double t_[FH];
#pragma unroll
          for (int u = 0; u < FH; ++u) t_[u] = T1[(size_t)(qb + u < m_tot ? qb + u : qb) * S + s];
#pragma unroll
          for (int u = 0; u < FH; ++u) {
            const bool ok = qb + u < m_tot;
            a1 = ok ? a1 + t_[u] : a1;
            a2 = ok ? a2 + t_[u] * t_[u] : a2;
          }
        }
        kv = (a1 * a1 - a2) / 2;
      } else if (act) {
        for (int qb = 0; qb < m_tot; qb += FH) {
          double p_[FH], v_[FH];
#pragma unroll
          for (int u = 0; u < FH; ++u) {
            v_[u] = vc[qb + u < m_tot ? qb + u : qb];
            p_[u] = Pl[(size_t)(qb + u < m_tot ? qb + u : qb) * S + s];
          }
#pragma unroll
          for (int u = 0; u < FH; ++u) {
            const double p = sP * p_[u];
            const bool ok = qb + u < m_tot;
            a1 = ok ? a1 + v_[u] * p : a1;
            a2 = ok ? a2 + (v_[u] * p) * (v_[u] * p) : a2;
          }
        }
        kv = (a1 * a1 - a2) / 2;
      }
      if (mine) {  // computeAnovaDerivative (sgd.nim:176-188) of this thread's entries
#pragma unroll
        for (int u = 0; u < RC; ++u) {
          const int cc = slot_of(u);
          const double val = vc[cc];
          const double p = sP * Pl[(size_t)cc * S + s];
          dA[(size_t)cc * S + s] = val * (a1 - p * val);
        }
      }
    } else {
      lds_barrier();  // the field lists
      // sgd_ffm.nim:18-30 visits the pairs (q1, q2), j_q1 < j_q2, q1 outer / q2 inner, and adds
      //   part                 += (P[f2][j1] P[f1][j2]) x1 x2
      //   dA[f2][q1]           += x1 x2 P[f1][j2]          dA[f1][q2] += x1 x2 P[f2][j1]
      // A slot (f, q) therefore collects, over the entries q' of field f, x_q x_q' P[field(q)][j_q'] in the order
      //   q' < q with j_q' < j_q   (visited as (q', q));   then all q' with j_q' > j_q   (visited as (q, q'));
      //   then q' > q with j_q' < j_q   (visited as (q', q))
      // -- the same additions in the same order, one slot at a time, so the slots can be split over the groups.
      if (mine) {
#pragma unroll
        for (int u = 0; u < RC; ++u) {
          const int f = cs[u] >> 16, q = cs[u] & 0xFFFF;
          const int64_t j = jc[q];
          const double xq = vc[q];
          const int fq = fc[q], nf = fcnt[f];
          const int* ent = fent + f * mc;
          double acc = 0.0;
          for (int ph = 0; ph < 3; ++ph)
            for (int t = 0; t < nf; ++t) {
              const int q2 = ent[t];
              const int64_t j2 = jc[q2];
              const bool take = ph == 0 ? (q2 < q && j2 < j) : ph == 1 ? (j2 > j) : (q2 > q && j2 < j);
              if (take) {
                const double v12 = (j2 < j) ? vc[q2] * xq : xq * vc[q2];
                acc += v12 * (sP * Pl[((size_t)fq * m + q2) * S + s]);
              }
            }
          dA[(size_t)slot_of(u) * S + s] = acc;
        }
      }
      if (by_pair) {
        // The prediction exactly as sgd_ffm.nim:18-27 forms it: per pair ONE dot product over the factors, ascending s,
        // times x1, times x2, added to the running prediction in the order the pairs are visited.  The dot products of
        // different pairs are independent: thread p takes the ordered pair (q1, q2) = (p / m, p mod m).
        for (int p = tid; p < m * m; p += T) {
          const int q1 = p / m, q2 = p - q1 * m;
          double c = 0.0;
          if (jc[q1] < jc[q2]) {
            const double* pa_ = Pl + ((size_t)fc[q2] * m + q1) * S;
            const double* pb_ = Pl + ((size_t)fc[q1] * m + q2) * S;
            double tmp = 0.0;
            for (int sb = 0; sb < k; sb += FH) {
              double a_[FH], b_[FH];
#pragma unroll
              for (int u = 0; u < FH; ++u) {
                a_[u] = pa_[sb + u < k ? sb + u : sb];
                b_[u] = pb_[sb + u < k ? sb + u : sb];
              }
#pragma unroll
              for (int u = 0; u < FH; ++u) tmp = sb + u < k ? tmp + (sP * a_[u]) * (sP * b_[u]) : tmp;
            }
            c = tmp * vc[q1] * vc[q2];
          }
          pc[p] = c;
        }
        lds_barrier();
        for (int q1 = 0; q1 < m; ++q1) {
          const int64_t j1 = jc[q1];
          for (int qb = 0; qb < m; qb += FH) {
            int64_t j_[FH];
            double c_[FH];
#pragma unroll
            for (int u = 0; u < FH; ++u) {
              const int q2 = qb + u < m ? qb + u : qb;
              j_[u] = jc[q2];
              c_[u] = pc[q1 * m + q2];
            }
#pragma unroll
            for (int u = 0; u < FH; ++u) yh = (qb + u < m && j1 < j_[u]) ? yh + c_[u] : yh;
          }
        }
      } else if (act) {  // rows too long for a pair table: per-factor sums over the pairs, every group alike (the
                         // prediction then differs from the reference's grouping by rounding)
        double part = 0.0;
        for (int q1 = 0; q1 < m; ++q1) {
          const int64_t j1 = jc[q1];
          const int f1 = fc[q1];
          const double x1 = vc[q1];
          for (int qb = 0; qb < m; qb += FH) {
            int64_t j_[FH];
            int f_[FH];
            double x_[FH], pa_[FH], pb_[FH];
#pragma unroll
            for (int u = 0; u < FH; ++u) {
              const int q2 = qb + u < m ? qb + u : qb;
              j_[u] = jc[q2];
              f_[u] = fc[q2];
              x_[u] = vc[q2];
            }
#pragma unroll
            for (int u = 0; u < FH; ++u) {
              const int q2 = qb + u < m ? qb + u : qb;
              pa_[u] = Pl[((size_t)f_[u] * m + q1) * S + s];
              pb_[u] = Pl[((size_t)f1 * m + q2) * S + s];
            }
#pragma unroll
            for (int u = 0; u < FH; ++u) {
              const bool ok = qb + u < m && j1 < j_[u];
              const double v12 = x1 * x_[u];
              const double pa = sP * pa_[u], pb = sP * pb_[u];
              part = ok ? part + (pa * pb) * v12 : part;
            }
          }
        }
        kv = part;
      }
    }
    if (g == 0) red[s] = (act && s < k) ? kv : 0.0;
    lds_barrier();
    double tot = 0.0;
    for (int sb = 0; sb < k; sb += FH) {  // sgd.nim:172-173, ascending s
      double r_[FH];
#pragma unroll
      for (int u = 0; u < FH; ++u) r_[u] = red[sb + u < k ? sb + u : sb];
#pragma unroll
      for (int u = 0; u < FH; ++u) tot = sb + u < k ? tot + r_[u] : tot;
    }
    yh = by_pair ? yh : yh + tot;  // by_pair: the pair terms are in yh already
    lds_barrier();
    if (tid == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);

    // ---- 5. update; what is written for a slot is also left in its LDS place ----
    if (OPT == OPT_SGD) {
      const double eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      const double eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      const double sPn = sP * (1 - eta_P * O.beta);
      if (mine) {
#pragma unroll
        for (int ub = 0; ub < RC; ub += CH) {
          if (ub * G + g >= n_slots) break;
          double p_[CH], d_[CH];
          int64_t j_[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int cc = slot_of(ub + u);
            j_[u] = jc[cs[ub + u] & 0xFFFF];
            p_[u] = Pl[(size_t)cc * S + s];
            d_[u] = dA[(size_t)cc * S + s];
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) {  // branch-free: the division chains interleave
            const bool ok = (ub + u) * G + g < n_slots;
            const size_t e = M.row(cs[ub + u] >> 16, j_[u]) * Kp + s;
            const double p = sP * p_[u];
            const double update = eta_P * (dL * d_[u] + O.beta * p);
            viol_acc = ok ? viol_acc + fabs(update) : viol_acc;
            const double nv = (p - update) / sPn;
            M.P[e] = nv;
            dA[(size_t)slot_of(ub + u) * S + s] = nv;
          }
        }
      }
      sP = sPn;
      if (M.fit_intercept) {
        const double update = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf) * (dL + O.alpha0 * b);
        if (tid == 0) viol_acc += fabs(update);
        b -= update;
      }
      if (M.fit_linear) {
        const double swn = sw * (1 - eta_w * O.alpha);
        if (tid < m) {
          const double wj = sw * wl[tid];
          const double update = eta_w * (dL * vc[tid] + O.alpha * wj);
          viol_acc += fabs(update);
          const double nv = (wj - update) / swn;
          M.w[jc[tid]] = nv;
          wp[tid] = nv;
        }
        sw = swn;
      } else if (tid < m) {
        wp[tid] = wl[tid];
      }
      // resetScaling: sgd.nim:116-131 (rare; the requested rows are simply read again afterwards)
      bool reset = false;
      if (sP < 1e-9) {
        __syncthreads();
        if (act)
          for (int64_t r = g; r < (int64_t)M.nb * M.da; r += G) M.P[(size_t)r * Kp + s] *= sP;
        sP = 1.0;
        reset = true;
      }
      if (M.fit_linear && sw < 1e-9) {
        __syncthreads();
        for (int64_t j = tid; j < M.d; j += T) M.w[j] *= sw;
        sw = 1.0;
        reset = true;
      }
      if (reset) {
        __syncthreads();
        load_rows(s1, D1, live1);
        fresh = true;
      }
    } else {
      // updateG(): optimizer/adagrad.nim:113-134
      if (mine) {
#pragma unroll
        for (int ub = 0; ub < RC; ub += CH) {
          if (ub * G + g >= n_slots) break;
          double d_[CH], g_[CH], n_[CH];
          size_t e_[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            e_[u] = M.row(cs[ub + u] >> 16, jc[cs[ub + u] & 0xFFFF]) * Kp + s;
            d_[u] = dA[(size_t)slot_of(ub + u) * S + s];
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) {  // the state rows of CH slots requested together
            g_[u] = O.G[e_[u]];
            n_[u] = O.N[e_[u]];
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int cc = slot_of(ub + u);
            const double grad = dL * d_[u];
            const double gn = g_[u] + grad, nn = n_[u] + grad * grad;
            O.G[e_[u]] = gn;
            O.N[e_[u]] = nn;
            dA[(size_t)cc * S + s] = gn;
            Pl[(size_t)cc * S + s] = nn;
          }
        }
      }
      if (M.fit_intercept) {
        gsb += dL;
        gnb += dL * dL;
      }
      if (tid < m && M.fit_linear) {
        const int64_t j = jc[tid];
        const double gg = dL * vc[tid];
        const double gn = O.Gw[j] + gg, nn = O.Nw[j] + gg * gg;
        O.Gw[j] = gn;
        O.Nw[j] = nn;
        wp[tid] = gn;
        wp2[tid] = nn;
      }
    }
    // ---- 6. the entries of t+2 go to their buffer; the pipeline advances ----
    store_entries(s2, D2, live2);
    m_prev = m_tot;
    D0 = D1;
    D1 = D2;
    D2 = D3;
    i3 = i4;
    lds_barrier();
  }

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

template <int KIND, int OPT, int RC, int T>
static int launch_seq_pipe_t(nfm_ctx* ctx, const SeqArgs& a, int S, int lgS, int split_terms, size_t lds_bytes) {
  auto kern = k_sequential_pipe<KIND, OPT, RC, T>;
  if (lds_bytes > 64 * 1024)
    NFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  TimedLaunch tl(ctx, "sequential");
  hipLaunchKernelGGL(kern, dim3(1), dim3(T), lds_bytes, ctx->stream, a, S, lgS, split_terms);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}
template <int KIND, int OPT>
static int launch_seq_pipe(nfm_ctx* ctx, const SeqArgs& a, int S, int lgS, int rc, int threads, int split_terms, size_t lds_bytes) {
  if (threads == kPipeThreadsWide) {  // rc <= 16
    if (rc <= 2) return launch_seq_pipe_t<KIND, OPT, 2, kPipeThreadsWide>(ctx, a, S, lgS, split_terms, lds_bytes);
    if (rc <= 4) return launch_seq_pipe_t<KIND, OPT, 4, kPipeThreadsWide>(ctx, a, S, lgS, split_terms, lds_bytes);
    if (rc <= 8) return launch_seq_pipe_t<KIND, OPT, 8, kPipeThreadsWide>(ctx, a, S, lgS, split_terms, lds_bytes);
    return launch_seq_pipe_t<KIND, OPT, 16, kPipeThreadsWide>(ctx, a, S, lgS, split_terms, lds_bytes);
  }
  if (rc <= 1) return launch_seq_pipe_t<KIND, OPT, 1, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  if (rc <= 2) return launch_seq_pipe_t<KIND, OPT, 2, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  if (rc <= 4) return launch_seq_pipe_t<KIND, OPT, 4, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  if (rc <= 8) return launch_seq_pipe_t<KIND, OPT, 8, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  if (rc <= 16) return launch_seq_pipe_t<KIND, OPT, 16, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  if (rc <= 32) return launch_seq_pipe_t<KIND, OPT, 32, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
  return launch_seq_pipe_t<KIND, OPT, 64, kPipeThreads>(ctx, a, S, lgS, split_terms, lds_bytes);
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
  SeqArgs a{X, M, O, perm_dev, begin, end, it0, out2_dev, m_cap, nullptr};
  // The pipelined step (k_sequential_pipe): degree-2 FMs with one order and field-aware models; at most 256 factors and
  // 256 entries per row, at most 64 rows per thread, everything of a sample in LDS.  NFM_SEQ_PIPE=0 switches it off.
  static const bool pipe_on = !(getenv("NFM_SEQ_PIPE") && atoi(getenv("NFM_SEQ_PIPE")) == 0);
  const bool ffm = M.kind == NFM_KIND_FFM;
  if (pipe_on && (ffm || (M.nb == 1 && M.degree == 2)) && M.nb >= 1 && M.nb < 32768 && m_cap <= kPipeThreads) {
    int S = 2, lgS = 1;
    while (S < M.Kp) { S <<= 1; ++lgS; }
    const int nbk = ffm ? M.nb : 1;
    const int64_t slots = (int64_t)nbk * m_cap;
    int threads = kPipeThreads;
    int G = S <= threads ? threads / S : 0;
    int64_t rc = G > 0 ? (slots + G - 1) / G : (int64_t)1 << 30;
    if (G > 0 && rc >= 8 && (slots + 2 * G - 1) / (2 * G) <= 16) {  // 8 wavefronts: half of the rows per thread
      threads = kPipeThreadsWide;
      G *= 2;
      rc = (slots + G - 1) / G;
    }
    const size_t pair_doubles = ffm && (int64_t)m_cap * m_cap <= kPipePairCap ? (size_t)m_cap * m_cap : 0;
    // FM: the products of the per-factor sums are formed in parallel into one more [entries][S] table when it fits
    const size_t base_doubles = (size_t)threads + 2 * (size_t)slots * S + 7 * (size_t)m_cap + pair_doubles;
    const size_t tail_bytes = sizeof(int64_t) * 3 * (size_t)m_cap +
                              sizeof(int) * ((size_t)m_cap + (ffm ? 3 * (size_t)m_cap + nbk + (size_t)nbk * m_cap : 0));
    const int split_terms = !ffm && sizeof(double) * (base_doubles + (size_t)m_cap * S) + tail_bytes <= 160 * 1024 ? 1 : 0;
    const size_t pipe_bytes = sizeof(double) * (base_doubles + (split_terms ? (size_t)m_cap * S : 0)) + tail_bytes;
    if (G >= 1 && rc <= 64 && pipe_bytes <= 160 * 1024) {
      if (ffm) {
        if (opt_kind == OPT_SGD) return launch_seq_pipe<NFM_KIND_FFM, OPT_SGD>(ctx, a, S, lgS, (int)rc, threads, split_terms, pipe_bytes);
        return launch_seq_pipe<NFM_KIND_FFM, OPT_ADAGRAD>(ctx, a, S, lgS, (int)rc, threads, split_terms, pipe_bytes);
      }
      if (opt_kind == OPT_SGD) return launch_seq_pipe<NFM_KIND_FM, OPT_SGD>(ctx, a, S, lgS, (int)rc, threads, split_terms, pipe_bytes);
      return launch_seq_pipe<NFM_KIND_FM, OPT_ADAGRAD>(ctx, a, S, lgS, (int)rc, threads, split_terms, pipe_bytes);
    }
  }
  size_t lds_bytes = sizeof(double) * ((size_t)T + (size_t)(M.nb > 0 ? M.nb : 1) * m_cap * T);
  if (lds_bytes > 160 * 1024) {
    // the per-sample gradient does not fit the LDS (long rows, many blocks): global scratch, the reduction buffer stays
    if (!ctx->seq_scratch) ctx->seq_scratch = new DevBuf();
    const size_t need = sizeof(double) * (size_t)(M.nb > 0 ? M.nb : 1) * m_cap * T;
    if (!(ctx->seq_scratch->p && need <= ctx->seq_scratch->bytes)) {
      NFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (growing: an earlier launch may still use the old block)
      NFM_TRY(ctx->seq_scratch->alloc(need));
    }
    a.dA_global = ctx->seq_scratch->as<double>();
    lds_bytes = sizeof(double) * (size_t)T;
    if (M.kind == NFM_KIND_FM) {
      if (opt_kind == OPT_SGD) return launch_seq_t<NFM_KIND_FM, OPT_SGD, false>(ctx, a, T, lds_bytes);
      return launch_seq_t<NFM_KIND_FM, OPT_ADAGRAD, false>(ctx, a, T, lds_bytes);
    }
    if (opt_kind == OPT_SGD) return launch_seq_t<NFM_KIND_FFM, OPT_SGD, false>(ctx, a, T, lds_bytes);
    return launch_seq_t<NFM_KIND_FFM, OPT_ADAGRAD, false>(ctx, a, T, lds_bytes);
  }
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
