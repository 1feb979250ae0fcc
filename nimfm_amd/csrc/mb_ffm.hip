// nimfm_amd/csrc/mb_ffm.hip -- NFM_MODE_MINIBATCH for FieldAwareFactorizationMachine.
//
// Replaces the reference's FFM Hogwild drivers (optimizer/sgd_ffm_multi.nim, adagrad_ffm_multi.nim)
// with the deterministic mini-batch rule of DESIGN.md section 4; the per-sample arithmetic is
// optimizer/sgd_ffm.nim:11-30 (predictWithGrad) and sgd.nim:205-243 / adagrad.nim:87-134 with
// "order" = field: a step touches ALL nFields rows P[f][j] of every feature j of the sample
// (sgd_ffm.nim:43), and so does the rule here.
//
//   row phase     one wavefront per sample.  Output (q, f), q = nnz of the row, f = field:
//                 dA[q][f] = x_q * sum_{q': field(q') = f, j_q' != j_q} x_q' * P[field(q)][j_q']
//                 (the reference's dA[f][j_q], accumulated in the same q' order), written to the
//                 contribution buffer at slot (toff + q) * F + f; yhat's pair sum is
//                 1/2 sum_q sum_f <P[f][j_q], dA[q][f]>.
//   column phase  L lanes per unique feature j of the batch: for every field f the row P[f][j] is
//                 read once, the touches' contribution rows are combined in sample order, the row
//                 is written once.  One extra workgroup closes the batch (as in mb_fm.hip).
#include "fm_device.h"
#include "mb.h"
// Streaming hints (bit mask; measured on cfg4: F = 16, k = 8, tables of 3 x 100 MB, mini-batch 32768):
//   1  the per-touch contribution rows are WRITTEN once by the row phase   } non-temporal: row phase 411 -> 365 us,
//   2  ... and READ once by the column phase                               } column phase 333 -> 317 us
//   4  the column phase's writes of P / g_sum / g_norm rows (next read: the next batch's row phase, 300 MB later)
//   8  the column phase's reads of those rows: column phase 317 -> 290 us, the next row phase 366 -> 377 us
//  16  the row phase's reads of the state rows: row phase 377 -> 429 us (a feature is read by ~5 samples of a batch:
//      these reads WANT the cache) -- off
// 15: cfg4 4.25e7 -> 4.78e7 samples/s (roofline.frac 0.355 -> 0.40).
#ifndef NFM_FFM_NT
#define NFM_FFM_NT 15
#endif
namespace nfm {
typedef double ffm_v2d __attribute__((ext_vector_type(2)));
template <int BIT>
__device__ __forceinline__ double2 ffm_ld(const double* p) {
  if constexpr ((NFM_FFM_NT & BIT) != 0) {
    const ffm_v2d w = __builtin_nontemporal_load(reinterpret_cast<const ffm_v2d*>(p));
    return double2{w.x, w.y};
  } else {
    return *reinterpret_cast<const double2*>(p);
  }
}
template <int BIT>
__device__ __forceinline__ void ffm_st(double* p, double2 v) {
  if constexpr ((NFM_FFM_NT & BIT) != 0) {
    ffm_v2d w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<ffm_v2d*>(p));
  } else {
    *reinterpret_cast<double2*>(p) = v;
  }
}
}  // namespace nfm

namespace nfm {

struct SampleRec {
  double dL, etaP, etaw, pad;
};
struct PartA {
  double loss, viol, acc0, acc1;
};
constexpr int kFtab = 64;

// defined in mb_fm.hip
__global__ void k_schedule(OptView O, int fit_linear, int fit_intercept, const int64_t* __restrict__ bat_pos,
                           const double* __restrict__ it0p, double* __restrict__ Dtab, double* __restrict__ Ftab);
__global__ void k_scale_prefix(double* __restrict__ sc, const double* __restrict__ Dtab, double* __restrict__ Stab, int64_t nb);
__global__ void k_epoch_close(const double* __restrict__ parts, int n, double* __restrict__ out_acc);
__global__ void k_set_double(double* p, double v);

struct FRowArgs {
  CsrView X;
  ModelView M;
  OptView O;
  const int64_t* perm;
  const int64_t* toff;  // touch offset of every sample of the epoch call
  int64_t begin, p0, t_base;  // t_base = toff of the batch's first sample
  int32_t len, use_stored;
  double it_b;
  const double* it0p;
  const double* scales;
  double* contrib;  // [batch touches][F][Kp]
  SampleRec* rec;
  PartA* parts;
};

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_ffm_row_phase(FRowArgs a) {
  constexpr int R = kWave / L;
  __shared__ double red[kWavesPerBlock][4];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int pib = blockIdx.x * kWavesPerBlock + wv;
  double r_loss = 0.0, r_acc0 = 0.0, r_acc1 = 0.0;
  if (pib < a.len) {
    const int64_t pos = a.p0 + pib;
    const int64_t i = a.perm ? a.perm[pos] : a.begin + pos;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int F = M.nb;
    const double y = dev::target_of(X.y[i], M.task);
    const double itp = (a.it0p[0] + a.it_b) - 1.0;
    const bool stored = a.use_stored != 0;
    double b0 = M.sc[SC_INTERCEPT];
    const double sP = OPT == OPT_SGD ? a.scales[0] : 1.0, sw = OPT == OPT_SGD ? a.scales[1] : 1.0;
    if (OPT == OPT_ADAGRAD && !stored && M.fit_intercept)
      b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
    const double tmpP = O.eta0 * itp * O.beta, denw = itp * O.eta0 * O.alpha;
    auto load_p = [&](size_t e) -> double2 {
      if (OPT == OPT_SGD || stored) {
        double2 v = *reinterpret_cast<const double2*>(M.P + e);
        v.x *= sP;
        v.y *= sP;
        return v;
      }
      const double2 gg = *reinterpret_cast<const double2*>(O.G + e);
      const double2 nn = *reinterpret_cast<const double2*>(O.N + e);
      double2 v;
      v.x = dev::adagrad_param(gg.x, nn.x, O.eta0, tmpP);
      v.y = dev::adagrad_param(gg.y, nn.y, O.eta0, tmpP);
      return v;
    };
    double part = 0.0;
    for (int q = lane; q < m; q += kWave) {
      const int j = X.indices[q0 + q];
      double wj = sw * M.w[j];
      if (OPT == OPT_ADAGRAD && !stored && M.fit_linear) wj = -O.eta0 * O.Gw[j] / (denw + sqrt(O.Nw[j]));
      part += wj * X.data[q0 + q];
    }
    double* C = a.contrib + (size_t)(a.toff[pos] - a.t_base) * F * M.Kp;
    const int n_out = m * F;
    for (int ob = 0; ob < n_out; ob += R) {
      const int o = ob + g;
      if (o < n_out) {
        const int q = o / F, f = o % F;
        const int jq = X.indices[q0 + q], fq = X.fields[q0 + q];
        const double xq = X.data[q0 + q];
        double2 v = {0.0, 0.0};
        for (int q2 = 0; q2 < m; ++q2) {
          if (X.fields[q0 + q2] != f) continue;
          const int j2 = X.indices[q0 + q2];
          if (j2 == jq) continue;
          const double x2 = X.data[q0 + q2];
          const double2 p = load_p(M.row(fq, j2) * M.Kp + 2 * l);
          // sgd_ffm.nim:29-30: dA += val1 * val2 * P  (left to right)
          v.x += xq * x2 * p.x;
          v.y += xq * x2 * p.y;
        }
        *reinterpret_cast<double2*>(C + ((size_t)q * F + f) * M.Kp + 2 * l) = v;
        if (v.x != 0.0 || v.y != 0.0) {
          const double2 pf = load_p(M.row(f, jq) * M.Kp + 2 * l);
          part += 0.5 * (pf.x * v.x + pf.y * v.y);
        }
      }
    }
    part = dev::wave_sum(part);
    const double yh = b0 + part;
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);
    r_loss = dev::loss_value(O.loss, O.loss_param, y, yh);
    double etaP = 0.0, etaw = 0.0;
    if (OPT == OPT_SGD) {
      const double it = (a.it0p[0] + a.it_b) + (double)pib;
      etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
      etaw = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
      if (M.fit_intercept) {
        const double eta0 = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it);
        r_acc0 = eta0 * dL;
        r_acc1 = eta0;
      }
    } else if (M.fit_intercept) {
      r_acc0 = dL;
      r_acc1 = dL * dL;
    }
    if (lane == 0) a.rec[pib] = SampleRec{dL, etaP, etaw, 0.0};
  }
  if (lane == 0) {
    red[wv][0] = r_loss;
    red[wv][1] = 0.0;
    red[wv][2] = r_acc0;
    red[wv][3] = r_acc1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PartA p{0.0, 0.0, 0.0, 0.0};
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) {
      p.loss += red[w_][0];
      p.acc0 += red[w_][2];
      p.acc1 += red[w_][3];
    }
    a.parts[blockIdx.x] = p;
  }
}


// ------------------------------------------------------------------------------------------------
// row phase, LDS-resident neighbourhood (rows of at most 64 entries, at most 64 fields, m*F rows that
// fit the workgroup's LDS share): one wavefront per sample.
//   1. lane t holds entry t (index, field, value); one ballot per field gives the entries of a field
//   2. ALL m*F parameter rows P[f][j_q] of the sample are gathered into LDS, every load in flight at
//      once (16 B per lane, a row = L consecutive lanes) -- each row is needed twice, as the
//      partner row of one output and for yhat of the mirrored one
//   3. outputs (q, f) from LDS: dA[q][f] = x_q * sum_{q' in field f, j_q' != j_q} x_q' P[f_q][j_q']
//      in ascending q' (the reference's accumulation order, sgd_ffm.nim:24-30), written to the
//      contribution buffer; yhat's pair sum = 1/2 sum <P[f][j_q], dA[q][f]>
// ------------------------------------------------------------------------------------------------
// WPB wavefronts per workgroup.  COOP = false: one sample per wavefront (a sample's m*F rows fit a quarter of the
// LDS share).  COOP = true: the WPB wavefronts work on ONE sample together -- for samples whose rows need more (39 fields
// x 39 entries at k = 4: 49 KiB -> three workgroups per CU).  With one wavefront per such workgroup a CU ran three
// wavefronts, each issuing ~10k instructions per sample (1521 rows x (address, square root, division), then 1521
// outputs): the row phase was bound by instruction issue on three of four SIMDs' single wavefronts, not by memory.
template <int L, int OPT, int WPB, bool COOP>
__global__ __launch_bounds__(kWave * WPB) void k_ffm_row_phase_lds(FRowArgs a, int m_cap) {
  constexpr int R = kWave / L;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ double red[WPB][4];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int F = M.nb, Kp = M.Kp;
  // per-wavefront LDS: rows [m_cap * F][Kp] doubles, masks [F] u64, values [m_cap] doubles, indices/fields [m_cap] ints
  const size_t per_wave = ((size_t)m_cap * F * Kp * 8 + (size_t)F * 8 + (size_t)m_cap * 16 + 15) / 16 * 16;
  unsigned char* base = lds_raw + (COOP ? (size_t)0 : (size_t)wv * per_wave);
  constexpr int RW = COOP ? R * WPB : R;        // rows handled per step by the sample's lane groups
  const int gg = COOP ? wv * R + g : g;         // this lane group among them
  double* rows = reinterpret_cast<double*>(base);
  unsigned long long* fmask = reinterpret_cast<unsigned long long*>(base + (size_t)m_cap * F * Kp * 8);
  double* xs = reinterpret_cast<double*>(base + (size_t)m_cap * F * Kp * 8 + (size_t)F * 8);
  int* js = reinterpret_cast<int*>(xs + m_cap);
  int* fs = js + m_cap;
  const int pib = COOP ? (int)blockIdx.x : (int)blockIdx.x * WPB + wv;
  const bool valid = pib < a.len;
  int64_t q0 = 0;
  int m = 0;
  double y = 0.0;
  int64_t pos = 0;
  if (valid) {
    pos = a.p0 + pib;
    const int64_t i = a.perm ? a.perm[pos] : a.begin + pos;
    q0 = X.indptr[i];
    m = (int)(X.indptr[i + 1] - q0);
    y = dev::target_of(X.y[i], M.task);
  }
  const double itp = (a.it0p[0] + a.it_b) - 1.0;
  const bool stored = a.use_stored != 0;
  double b0 = M.sc[SC_INTERCEPT];
  const double sP = OPT == OPT_SGD ? a.scales[0] : 1.0, sw = OPT == OPT_SGD ? a.scales[1] : 1.0;
  if (OPT == OPT_ADAGRAD && !stored && M.fit_intercept) b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
  const double tmpP = O.eta0 * itp * O.beta, denw = itp * O.eta0 * O.alpha;
  // 1. this lane's entry; linear term; per-field entry masks
  int jt = 0, ft = -1;
  double xt = 0.0, part = 0.0;
  if (lane < m) {
    jt = X.indices[q0 + lane];
    ft = X.fields[q0 + lane];
    xt = X.data[q0 + lane];
    double wj = sw * M.w[jt];
    if (OPT == OPT_ADAGRAD && !stored && M.fit_linear) wj = -O.eta0 * O.Gw[jt] / (denw + sqrt(O.Nw[jt]));
    if (!COOP || wv == 0) part += wj * xt;
    xs[lane] = xt;  // COOP: every wavefront writes the same values
    js[lane] = jt;
    fs[lane] = ft;
  }
  for (int f = 0; f < F; ++f) {
    const unsigned long long mk = __ballot(ft == f);
    if (lane == 0) fmask[f] = mk;
  }
  // 2. gather the m * F rows
  const int n_out = m * F;
  constexpr int U = 8;  // (16 in flight for rows of 16-64 B measured no faster)
  for (int ob = 0; ob < n_out; ob += RW * U) {
    double2 r0[U], r1[U];
    int oo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int o = ob + u * RW + gg;
      oo[u] = o < n_out ? o : -1;
      const int q = o < n_out ? o / F : 0, f = o < n_out ? o % F : 0;
      const int jq = __shfl(jt, q, kWave);
      const size_t e = M.row(f, jq) * Kp + 2 * l;
      r0[u] = r1[u] = {0.0, 0.0};
      if (oo[u] >= 0) {
        if (OPT == OPT_SGD || stored) {
          r0[u] = ffm_ld<16>(M.P + e);
        } else {
          r0[u] = ffm_ld<16>(O.G + e);
          r1[u] = ffm_ld<16>(O.N + e);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (oo[u] < 0) continue;
      double2 p;
      if (OPT == OPT_SGD || stored) {
        p.x = sP * r0[u].x;
        p.y = sP * r0[u].y;
      } else {
        p.x = dev::adagrad_param(r0[u].x, r1[u].x, O.eta0, tmpP);
        p.y = dev::adagrad_param(r0[u].y, r1[u].y, O.eta0, tmpP);
      }
      *reinterpret_cast<double2*>(rows + (size_t)oo[u] * Kp + 2 * l) = p;
    }
  }
  __syncthreads();
  // 3. outputs
  double* C = a.contrib + (size_t)(valid ? a.toff[pos] - a.t_base : 0) * F * Kp;
  for (int ob = 0; ob < n_out; ob += RW) {
    const int o = ob + gg;
    if (o < n_out) {
      const int q = o / F, f = o % F;
      const int jq = js[q], fq = fs[q];
      const double xq = xs[q];
      unsigned long long mk = fmask[f] & ~(1ull << q);
      double2 v = {0.0, 0.0};
      while (mk) {
        const int q2 = __ffsll((long long)mk) - 1;
        mk &= mk - 1;
        if (js[q2] == jq) continue;
        const double2 p = *reinterpret_cast<const double2*>(rows + ((size_t)q2 * F + fq) * Kp + 2 * l);
        const double x2 = xs[q2];
        v.x += xq * x2 * p.x;  // sgd_ffm.nim:29-30: dA += val1 * val2 * P  (left to right)
        v.y += xq * x2 * p.y;
      }
#if NFM_FFM_NT & 1  // the contribution rows are written once here and read once by the column phase
      {
        typedef double v2d_ __attribute__((ext_vector_type(2)));
        v2d_ w_;
        w_.x = v.x;
        w_.y = v.y;
        __builtin_nontemporal_store(w_, reinterpret_cast<v2d_*>(C + (size_t)o * Kp + 2 * l));
      }
#else
      *reinterpret_cast<double2*>(C + (size_t)o * Kp + 2 * l) = v;
#endif
      if (v.x != 0.0 || v.y != 0.0) {
        const double2 pf = *reinterpret_cast<const double2*>(rows + (size_t)o * Kp + 2 * l);
        part += 0.5 * (pf.x * v.x + pf.y * v.y);
      }
    }
  }
  part = dev::wave_sum(part);
  if (COOP) {  // the wavefronts' shares of the prediction, added in wavefront order
    if (lane == 0) red[wv][1] = part;
    __syncthreads();
    part = 0.0;
    for (int w_ = 0; w_ < WPB; ++w_) part += red[w_][1];
    __syncthreads();
  }
  double r_loss = 0.0, r_acc0 = 0.0, r_acc1 = 0.0;
  if (valid && (!COOP || wv == 0)) {
    const double yh = b0 + part;
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);
    r_loss = dev::loss_value(O.loss, O.loss_param, y, yh);
    double etaP = 0.0, etaw = 0.0;
    if (OPT == OPT_SGD) {
      const double it = (a.it0p[0] + a.it_b) + (double)pib;
      etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
      etaw = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
      if (M.fit_intercept) {
        const double eta0 = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it);
        r_acc0 = eta0 * dL;
        r_acc1 = eta0;
      }
    } else if (M.fit_intercept) {
      r_acc0 = dL;
      r_acc1 = dL * dL;
    }
    if (lane == 0) a.rec[pib] = SampleRec{dL, etaP, etaw, 0.0};
  }
  if (lane == 0) {
    red[wv][0] = r_loss;
    red[wv][1] = 0.0;
    red[wv][2] = r_acc0;
    red[wv][3] = r_acc1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PartA p{0.0, 0.0, 0.0, 0.0};
    for (int w_ = 0; w_ < WPB; ++w_) {
      p.loss += red[w_][0];
      p.acc0 += red[w_][2];
      p.acc1 += red[w_][3];
    }
    a.parts[blockIdx.x] = p;
  }
}

struct FColArgs {
  ModelView M;
  OptView O;
  const int32_t* ucol;
  const int64_t* uptr;
  const int32_t* ucol_s;  // unique features by descending touch count (plan.h): what the column phase walks
  const int64_t* ubeg_s;
  const int32_t* ucnt_s;
  const int32_t* tpos;
  const double* tx;
  const int64_t* tq;  // sample-order touch index of every sorted touch
  int64_t u0, u1, t_base;
  const double* scales_b;
  const double* scales_n;
  const double* Dtab_b;
  const double* Ftab_b;
  const double* contrib;
  const SampleRec* rec;
  double* parts;
  const PartA* partsA;
  const double* parts_prev;
  double* out_acc;
  double it_b, len;
  const double* it0p;
  int32_t use_stored, nA, n_prev;
};

// One work unit = (unique feature j, field f): MODE 0 walks the touches [t0, t1) and applies the update;
// heavy features (more than kHeavyTouches touches in the batch -- in field-aware data the features of a
// low-cardinality field are touched by a large share of every batch) are done in two steps: MODE 1 sums ONE
// segment of the touches and stores the partial sums (no side effects), MODE 2 (one wavefront per unit) adds
// the segments' partial sums -- lane group g takes segments g, g + R, ..., then a fixed xor-shuffle tree --
// and applies.  A partial record is PW = 2 Kp + 4 doubles: [acc Kp][accn Kp][seta, a0, a1, -].
template <int L, int OPT, int MODE>
__device__ __forceinline__ double ffm_unit(const FColArgs& a, int64_t j, int f, int l, int64_t t0, int64_t t1, double c, double sP,
                                           double sPn, double sw, double swn, double fP, double fw, double itp, double* hp,
                                           int64_t nseg, int PW) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int F = M.nb;
  double viol = 0.0;
  const bool do_w = M.fit_linear && f == 0;
  const size_t e = M.row(f, j) * M.Kp + 2 * l;
  double2 st = {0.0, 0.0}, g2 = {0.0, 0.0}, n2 = {0.0, 0.0}, p = {0.0, 0.0};
  if (MODE != 1) {
    if (OPT == OPT_SGD) {
      st = ffm_ld<8>(M.P + e);
      p.x = sP * st.x; p.y = sP * st.y;
    } else {
      g2 = ffm_ld<8>(O.G + e);
      n2 = ffm_ld<8>(O.N + e);
      if (a.use_stored) {
        p = ffm_ld<8>(M.P + e);
      } else {
        const double tmp = O.eta0 * itp * O.beta;
        p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmp);
        p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmp);
        if (O.track_viol) {
          st = ffm_ld<8>(M.P + e);
          viol += fabs(st.x - p.x) + fabs(st.y - p.y);
          ffm_st<4>(M.P + e, p);  // idempotent: every lane group of a MODE 2 wavefront writes the same
        }
      }
    }
  }
  double2 acc = {0.0, 0.0}, accn = {0.0, 0.0};
  double seta = 0.0, a0 = 0.0, a1 = 0.0;
  if (MODE == 2) {
    constexpr int RG = kWave / L;
    const int g_ = (int)(threadIdx.x & (kWave - 1)) / L;
    for (int64_t sg = g_; sg < nseg; sg += RG) {
      const double* rec_ = hp + (size_t)sg * F * PW;  // consecutive segments of one feature are F records apart
      const double2 pa = *reinterpret_cast<const double2*>(rec_ + 2 * l);
      const double2 pn = *reinterpret_cast<const double2*>(rec_ + M.Kp + 2 * l);
      acc.x += pa.x; acc.y += pa.y;
      accn.x += pn.x; accn.y += pn.y;
      seta += rec_[2 * M.Kp];
      a0 += rec_[2 * M.Kp + 1];
      a1 += rec_[2 * M.Kp + 2];
    }
#pragma unroll
    for (int sh = L; sh < kWave; sh <<= 1) {
      acc.x += dev::shfl_xor_d(acc.x, sh);
      acc.y += dev::shfl_xor_d(acc.y, sh);
      accn.x += dev::shfl_xor_d(accn.x, sh);
      accn.y += dev::shfl_xor_d(accn.y, sh);
      seta += dev::shfl_xor_d(seta, sh);
      a0 += dev::shfl_xor_d(a0, sh);
      a1 += dev::shfl_xor_d(a1, sh);
    }
    if (g_ != 0) return 0.0;
  } else {
    constexpr int TU = 4;  // touches requested together
    for (int64_t tb = t0; tb < t1; tb += TU) {
      int pib[TU];
      int64_t tq[TU];
      double x[TU];
      SampleRec r[TU];
      double2 v[TU];
#pragma unroll
      for (int q = 0; q < TU; ++q) {
        const int64_t t = tb + q < t1 ? tb + q : t1 - 1;
        pib[q] = a.tpos[t];
        tq[q] = a.tq[t];
        x[q] = do_w ? a.tx[t] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < TU; ++q) {
        r[q] = a.rec[pib[q]];
#if NFM_FFM_NT & 2
        {
          typedef double v2d_ __attribute__((ext_vector_type(2)));
          const v2d_ w_ = __builtin_nontemporal_load(reinterpret_cast<const v2d_*>(a.contrib + ((size_t)(tq[q] - a.t_base) * F + f) * M.Kp + 2 * l));
          v[q] = double2{w_.x, w_.y};
        }
#else
        v[q] = *reinterpret_cast<const double2*>(a.contrib + ((size_t)(tq[q] - a.t_base) * F + f) * M.Kp + 2 * l);
#endif
      }
#pragma unroll
      for (int q = 0; q < TU; ++q) {
        if (tb + q >= t1) break;
        if (OPT == OPT_SGD) {
          acc.x += r[q].etaP * (r[q].dL * v[q].x);
          acc.y += r[q].etaP * (r[q].dL * v[q].y);
          seta += r[q].etaP;
          if (do_w) { a0 += r[q].etaw * (r[q].dL * x[q]); a1 += r[q].etaw; }
        } else {
          const double gx = r[q].dL * v[q].x, gy = r[q].dL * v[q].y;
          acc.x += gx; acc.y += gy;
          accn.x += gx * gx; accn.y += gy * gy;
          if (do_w) { const double gw = r[q].dL * x[q]; a0 += gw; a1 += gw * gw; }
        }
      }
    }
  }
  if (MODE == 1) {
    *reinterpret_cast<double2*>(hp + 2 * l) = acc;
    *reinterpret_cast<double2*>(hp + M.Kp + 2 * l) = accn;
    if (l == 0) {
      hp[2 * M.Kp] = seta;
      hp[2 * M.Kp + 1] = a0;
      hp[2 * M.Kp + 2] = a1;
      hp[2 * M.Kp + 3] = 0.0;
    }
    return 0.0;
  }
  const double cd = dev::touch_div(c, O.touch_cap);  // SGD: divisor of the summed steps (mb_fm_kernels.h)
  if (OPT == OPT_SGD) {
    viol += fabs((acc.x + seta * O.beta * p.x) / cd) + fabs((acc.y + seta * O.beta * p.y) / cd);
    st.x = st.x * fP - (acc.x / cd) / sPn;
    st.y = st.y * fP - (acc.y / cd) / sPn;
    ffm_st<4>(M.P + e, st);
  } else {
    g2.x += acc.x; g2.y += acc.y;
    n2.x += dev::ada_norm_inc(acc.x, accn.x, O.ada_cross);
    n2.y += dev::ada_norm_inc(acc.y, accn.y, O.ada_cross);
    ffm_st<4>(O.G + e, g2);
    ffm_st<4>(O.N + e, n2);
  }
  if (do_w && l == 0) {
    const double wt = M.w[j];
    if (OPT == OPT_SGD) {
      const double wj = sw * wt;
      viol += fabs((a0 + a1 * O.alpha * wj) / cd);
      M.w[j] = wt * fw - (a0 / cd) / swn;
    } else {
      const double gw = O.Gw[j], nw = O.Nw[j];
      if (!a.use_stored) {
        const double wj = -O.eta0 * gw / (itp * O.eta0 * O.alpha + sqrt(nw));
        viol += fabs(wt - wj);
        M.w[j] = wj;
      }
      O.Gw[j] = gw + a0;
      O.Nw[j] = nw + dev::ada_norm_inc(a0, a1, O.ada_cross);
    }
  }
  return viol;
}

// decay corrections of a coordinate touched c times (as mb_fm.hip's touch_factors)
__device__ __forceinline__ void ffm_touch_factors(const FColArgs& a, int64_t ci, double& fP, double& fw) {
  fP = 1.0;
  fw = 1.0;
  if (ci > 1) {
    const double c = dev::touch_div((double)ci, a.O.touch_cap);
    if (ci <= kFtab) { fP = a.Ftab_b[ci - 1]; fw = a.Ftab_b[kFtab + ci - 1]; }  // (the table has the cap in it)
    else if (c > 1.0) { fP = pow(a.Dtab_b[0], 1.0 / c) / a.Dtab_b[0]; fw = pow(a.Dtab_b[1], 1.0 / c) / a.Dtab_b[1]; }
  }
}

struct FHeavyArgs {
  const int64_t* hv_u;     // heavy feature -> index into ucol / uptr
  const int64_t* hv_seg0;  // heavy feature -> its first segment
  int64_t h0, h1, s0, s1;  // this batch's heavy features / segments
  double* hpart;           // [(s1 - s0) * F][PW]
  double* parts;           // per-block viol partials of the apply kernel
  int32_t PW, pad_;
};

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_ffm_heavy_partial(FColArgs a, FHeavyArgs hv) {
  constexpr int R = kWave / L;
  const ModelView& M = a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int F = M.nb;
  const int64_t unit = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;  // (segment, field)
  const int64_t gs = hv.s0 + unit / F;
  const int f = (int)(unit % F);
  if (gs >= hv.s1) return;
  int64_t lo = hv.h0, hi = hv.h1 - 1;  // last heavy feature whose first segment is <= gs
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (hv.hv_seg0[mid] <= gs) lo = mid; else hi = mid - 1;
  }
  const int64_t u = hv.hv_u[lo];
  const int64_t j = a.ucol[u];
  const int64_t t0 = a.uptr[u] + (gs - hv.hv_seg0[lo]) * kHeavySegment;
  const int64_t t1 = min(t0 + (int64_t)kHeavySegment, a.uptr[u + 1]);
  ffm_unit<L, OPT, 1>(a, j, f, l, t0, t1, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0,
                      hv.hpart + ((size_t)(gs - hv.s0) * F + f) * hv.PW, 0, hv.PW);
}

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_ffm_heavy_apply(FColArgs a, FHeavyArgs hv) {
  __shared__ double red[kWavesPerBlock];
  const ModelView& M = a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int l = lane % L;
  const int F = M.nb;
  const int64_t wid = (int64_t)blockIdx.x * kWavesPerBlock + wv;  // one wavefront per (heavy feature, field)
  const int64_t h = hv.h0 + wid / F;
  const int f = (int)(wid % F);
  double viol = 0.0;
  if (h < hv.h1) {
    const int64_t u = hv.hv_u[h];
    const int64_t j = a.ucol[u];
    const int64_t c = a.uptr[u + 1] - a.uptr[u];
    double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0, fP = 1.0, fw = 1.0;
    if (OPT == OPT_SGD) {
      sP = a.scales_b[0]; sw = a.scales_b[1]; sPn = a.scales_n[0]; swn = a.scales_n[1];
      ffm_touch_factors(a, c, fP, fw);
    }
    const int64_t sg0 = hv.hv_seg0[h], nseg = hv.hv_seg0[h + 1] - sg0;
    viol = ffm_unit<L, OPT, 2>(a, j, f, l, 0, 0, (double)c, sP, sPn, sw, swn, fP, fw, (a.it0p[0] + a.it_b) - 1.0,
                               hv.hpart + ((size_t)(sg0 - hv.s0) * F + f) * hv.PW, nseg, hv.PW);
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[w_];
    hv.parts[blockIdx.x] = v;
  }
}

template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_ffm_col_phase(FColArgs a) {
  constexpr int R = kWave / L;
  __shared__ double red[5][kBlock];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const bool closer = blockIdx.x == gridDim.x - 1;
  // work unit = (unique feature, field): L lanes read the row P[f][j] once, combine the touches'
  // contribution rows in sample order, write the row once; the unit of field 0 also updates w[j]
  const int F = M.nb;
  const int64_t unit = closer ? -1 : ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  const int64_t u = closer ? a.u1 : a.u0 + unit / F;
  const int f = closer ? 0 : (int)(unit % F);
  const double itp = (a.it0p[0] + a.it_b) - 1.0;
  double viol = 0.0;
  if (u < a.u1) {
    const int64_t j = a.ucol_s[u];
    const int64_t t0 = a.ubeg_s[u], t1 = t0 + a.ucnt_s[u];
    const double c = (double)(t1 - t0);
    double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0, fP = 1.0, fw = 1.0;
    if (OPT == OPT_SGD) {
      sP = a.scales_b[0]; sw = a.scales_b[1]; sPn = a.scales_n[0]; swn = a.scales_n[1];
      const int64_t ci = t1 - t0;
      ffm_touch_factors(a, ci, fP, fw);
    }
    if (t1 - t0 <= kHeavyTouches)  // heavy features: k_ffm_heavy_partial / k_ffm_heavy_apply
      viol += ffm_unit<L, OPT, 0>(a, j, f, l, t0, t1, c, sP, sPn, sw, swn, fP, fw, itp, nullptr, 0, 0);
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[0][wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[0][w_];
    a.parts[blockIdx.x] = v;
  }
  if (!closer) return;
  __syncthreads();
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < a.nA; i += kBlock) {
    const PartA p = a.partsA[i];
    s[0] += p.loss; s[1] += p.viol; s[2] += p.acc0; s[3] += p.acc1;
  }
  for (int i = threadIdx.x; i < a.n_prev; i += kBlock) s[4] += a.parts_prev[i];
  for (int cc = 0; cc < 5; ++cc) red[cc][threadIdx.x] = s[cc];
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int cc = 0; cc < 5; ++cc) red[cc][threadIdx.x] += red[cc][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = red[1][0] + red[4][0];
    if (M.fit_intercept) {
      if (OPT == OPT_SGD) {
        const double b0 = M.sc[SC_INTERCEPT], f0 = a.Dtab_b[3], lc = dev::touch_div(a.len, O.touch_cap);
        v += fabs((red[2][0] + red[3][0] * O.alpha0 * b0) / lc);
        M.sc[SC_INTERCEPT] = f0 * b0 - red[2][0] / lc;
      } else {
        if (!a.use_stored) {
          const double old = M.sc[SC_INTERCEPT];
          const double nb_ = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
          v += fabs(old - nb_);
          M.sc[SC_INTERCEPT] = nb_;
        }
        O.gsc[0] += red[2][0];
        O.gsc[1] += dev::ada_norm_inc(red[2][0], red[3][0], O.ada_cross);
      }
    }
    a.out_acc[0] += red[0][0];
    a.out_acc[1] += v;
  }
}

// AdaGrad, start of a mini-batch: update() (optimizer/adagrad.nim:87-110) of everything the batch touches, ONCE per
// (feature, field) -- P = -eta0 g_sum / (eta0 it' beta + sqrt(g_norm)), viol += |stored - new|, stored = new; likewise the
// linear weights of the batch's features and the intercept.  The row phase then reads the stored parameters (one row
// per touch instead of the two state rows, and no square root / division per touch: a feature of cfg4 is read by ~5
// samples of a batch, and the row phase was bound by instruction issue) and the column phase only adds to the state.
template <int L>
__global__ __launch_bounds__(kBlock) void k_ffm_refresh(ModelView M, OptView O, const int32_t* __restrict__ ucol, int64_t u0,
                                                        int64_t u1, const double* __restrict__ it0p, double it_b,
                                                        double* __restrict__ parts) {
  constexpr int R = kWave / L;
  __shared__ double red[kWavesPerBlock];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int F = M.nb;
  const int64_t unit = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  const int64_t u = u0 + unit / F;
  const int f = (int)(unit % F);
  const double itp = (it0p[0] + it_b) - 1.0;
  double viol = 0.0;
  if (u < u1) {
    const int64_t j = ucol[u];
    const size_t e = M.row(f, j) * M.Kp + 2 * l;
    const double2 g2 = *reinterpret_cast<const double2*>(O.G + e), n2 = *reinterpret_cast<const double2*>(O.N + e);
    const double tmp = O.eta0 * itp * O.beta;
    double2 p;
    p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmp);
    p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmp);
    if (O.track_viol) {
      const double2 st = *reinterpret_cast<const double2*>(M.P + e);
      viol += fabs(st.x - p.x) + fabs(st.y - p.y);
    }
    *reinterpret_cast<double2*>(M.P + e) = p;
    if (f == 0 && l == 0 && M.fit_linear) {  // fit_linear.nim:50-57
      const double wt = M.w[j];
      const double wj = -O.eta0 * O.Gw[j] / (itp * O.eta0 * O.alpha + sqrt(O.Nw[j]));
      viol += fabs(wt - wj);
      M.w[j] = wj;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && M.fit_intercept) {
    const double old = M.sc[SC_INTERCEPT];
    const double nb_ = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
    viol += fabs(old - nb_);
    M.sc[SC_INTERCEPT] = nb_;
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[w_];
    parts[blockIdx.x] = v;
  }
}

template <int L, int OPT>
static int run_ffm(nfm_ctx* ctx, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                   const std::vector<int64_t>& t_base) {
  constexpr int R = kWave / L;
  hipStream_t st = ctx->stream;
  const double* Stab = W.Stab.as<double>();
  const double* Dtab = W.Dtab.as<double>();
  const double* it0p = W.itbuf.as<double>();
  const size_t half = W.partsB.bytes / sizeof(double) / 2;
  int n_prev = 0;
  // LDS-resident neighbourhood (k_ffm_row_phase_lds): rows <= 64 entries, <= 64 fields, at most 64 KiB
  // per wavefront (2 workgroups per CU at 160 KiB); NFM_FFM_LDS=0 falls back to the generic kernel
  static const bool lds_on = !(getenv("NFM_FFM_LDS") && atoi(getenv("NFM_FFM_LDS")) == 0);
  static const bool refresh_on = !(getenv("NFM_FFM_REFRESH") && atoi(getenv("NFM_FFM_REFRESH")) == 0);  // AdaGrad: k_ffm_refresh
  static const bool refresh_force = getenv("NFM_FFM_REFRESH") && atoi(getenv("NFM_FFM_REFRESH")) == 2;
  const int m_cap = (int)std::max<int64_t>(X.max_row, 1);
  size_t lds_bytes = 0;
  int lds_wpb = 4;
  if (lds_on && X.max_row <= kWave && M.nb <= kWave) {
    size_t per_wave = (size_t)m_cap * M.nb * M.Kp * 8 + (size_t)M.nb * 8 + (size_t)m_cap * 16;
    per_wave = (per_wave + 15) / 16 * 16;
    if (per_wave * 4 <= 150 * 1024) {
      lds_bytes = per_wave * 4;
    } else if (per_wave <= 150 * 1024) {
      lds_bytes = per_wave;
      lds_wpb = 1;
    }
  }
  if (lds_bytes > 64 * 1024) {
    if (lds_wpb == 4)
      NFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffm_row_phase_lds<L, OPT, 4, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    else
      NFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ffm_row_phase_lds<L, OPT, 4, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  for (int64_t b = 0; b < P.n_batches; ++b) {
    const int64_t p0 = P.bat_pos[b];
    const int len = (int)(P.bat_pos[b + 1] - p0);
    int use_stored = (OPT == OPT_ADAGRAD && P.first_singleton && b == 0) ? 1 : 0;
    const int wpb = lds_bytes > 0 ? lds_wpb : kWavesPerBlock;
    const int nA = (len + wpb - 1) / wpb;
    int nR = 0;  // workgroups of the refresh pass: their viol partials sit in front of the column phase's
    // worth its pass over the batch's (feature, field) units when a unit is read by several samples of the batch (cfg4:
    // 5.2 touches per feature, 4.8e7 -> 5.1e7 samples/s; the 39-field shape: 2.7, 1.3e7 -> 1.16e7: not there)
    const bool refresh = OPT == OPT_ADAGRAD && !use_stored && refresh_on &&
                         (refresh_force || (double)(t_base[b + 1] - t_base[b]) >= 4.0 * (double)(P.bat_uoff[b + 1] - P.bat_uoff[b]));
    if (refresh) {
      const int64_t ur0 = P.bat_uoff[b], ur1 = P.bat_uoff[b + 1];
      const int per_block_r = kWavesPerBlock * R;
      nR = (int)std::max<int64_t>(1, ((ur1 - ur0) * M.nb + per_block_r - 1) / per_block_r);
      TimedLaunch tl(ctx, "refresh");
      hipLaunchKernelGGL((k_ffm_refresh<L>), dim3(nR), dim3(kBlock), 0, st, M, O, P.ucol.as<int32_t>(), ur0, ur1, it0p, (double)p0,
                         W.partsB.as<double>() + (b & 1) * half);
      use_stored = 1;  // row and column phase take the parameters as stored
    }
    {
      FRowArgs ra{X, M, O, P.has_perm ? P.perm.as<int64_t>() : nullptr, P.toff.as<int64_t>(), P.begin, p0, t_base[b], len,
                  use_stored, (double)p0, it0p, OPT == OPT_SGD ? Stab + 2 * b : M.sc, W.contrib.as<double>(),
                  W.rec.as<SampleRec>(), W.partsA.as<PartA>()};
      TimedLaunch tl(ctx, "row_phase");
      if (lds_bytes > 0 && lds_wpb == 4)
        hipLaunchKernelGGL((k_ffm_row_phase_lds<L, OPT, 4, false>), dim3(nA), dim3(kBlock), lds_bytes, st, ra, m_cap);
      else if (lds_bytes > 0)
        hipLaunchKernelGGL((k_ffm_row_phase_lds<L, OPT, 4, true>), dim3(nA), dim3(kBlock), lds_bytes, st, ra, m_cap);  // one sample per workgroup
      else
        hipLaunchKernelGGL((k_ffm_row_phase<L, OPT>), dim3(nA), dim3(kBlock), 0, st, ra);
    }
    const int64_t u0 = P.bat_uoff[b], u1 = P.bat_uoff[b + 1];
    const int per_block = kWavesPerBlock * R;
    const int nB = (int)(((u1 - u0) * M.nb + per_block - 1) / per_block) + 1;  // units = (feature, field)
    {
      FColArgs ca{M, O, P.ucol.as<int32_t>(), P.uptr.as<int64_t>(), P.ucol_s.as<int32_t>(), P.ubeg_s.as<int64_t>(),
                  P.ucnt_s.as<int32_t>(), P.tpos.as<int32_t>(), P.tx.as<double>(),
                  P.tq.as<int64_t>(), u0, u1, t_base[b], OPT == OPT_SGD ? Stab + 2 * b : M.sc,
                  OPT == OPT_SGD ? Stab + 2 * (b + 1) : M.sc, OPT == OPT_SGD ? Dtab + 4 * b : nullptr,
                  OPT == OPT_SGD ? W.Ftab.as<double>() + (size_t)b * 2 * kFtab : nullptr, W.contrib.as<double>(),
                  W.rec.as<SampleRec>(), W.partsB.as<double>() + (b & 1) * half + nR, W.partsA.as<PartA>(),
                  W.partsB.as<double>() + ((b + 1) & 1) * half, W.out_acc.as<double>(), (double)p0, (double)len, it0p,
                  use_stored, nA, n_prev};
      {
        TimedLaunch tl(ctx, "col_phase");
        hipLaunchKernelGGL((k_ffm_col_phase<L, OPT>), dim3(nB), dim3(kBlock), 0, st, ca);
      }
      int nH = 0;
      if (P.bat_hoff[b + 1] > P.bat_hoff[b]) {  // features with more than kHeavyTouches touches in this batch
        const int PW = 2 * M.Kp + 4;
        FHeavyArgs ha{P.hv_u.as<int64_t>(), P.hv_seg0.as<int64_t>(), P.bat_hoff[b], P.bat_hoff[b + 1], P.bat_soff[b],
                      P.bat_soff[b + 1], W.hpart.as<double>(), W.partsB.as<double>() + (b & 1) * half + nR + nB, PW, 0};
        const int64_t units = (ha.s1 - ha.s0) * M.nb;
        const int nsb = (int)((units + per_block - 1) / per_block);
        nH = (int)(((ha.h1 - ha.h0) * M.nb + kWavesPerBlock - 1) / kWavesPerBlock);  // one wavefront per (feature, field)
        {
          TimedLaunch tl(ctx, "heavy_partial");
          hipLaunchKernelGGL((k_ffm_heavy_partial<L, OPT>), dim3(nsb), dim3(kBlock), 0, st, ca, ha);
        }
        TimedLaunch tl(ctx, "heavy_apply");
        hipLaunchKernelGGL((k_ffm_heavy_apply<L, OPT>), dim3(nH), dim3(kBlock), 0, st, ca, ha);
      }
      n_prev = nR + nB + nH;
    }
    if (W.after_batch) NFM_TRY(W.after_batch(b));
  }
  if (P.n_batches > 0)
    hipLaunchKernelGGL(k_epoch_close, dim3(1), dim3(kBlock), 0, st, W.partsB.as<double>() + ((P.n_batches - 1) & 1) * half,
                       n_prev, W.out_acc.as<double>());
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

int mb_ffm_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                 int64_t it0, double* out2_host, bool defer_sync) {
  NFM_CHECK(M.kind == NFM_KIND_FFM, NFM_ERR_UNSUPPORTED, "mb_ffm_epoch: FFM only");
  NFM_CHECK(M.Kp == 2 * M.L && M.Kp <= 128, NFM_ERR_UNSUPPORTED, "mini-batch mode supports n_components <= 128");
  NFM_CHECK(P.toff.p && (P.TM == 0 || P.tq.p), NFM_ERR_INVALID, "FFM plan lacks the touch tables");
  hipStream_t st = ctx->stream;
  // touch base of every batch and the largest batch (in touches): from the plan's toff
  const int64_t ns = P.end - P.begin;
  std::vector<int64_t> toff_h((size_t)ns + 1);
  NFM_HIP_CHECK(hipMemcpyAsync(toff_h.data(), P.toff.p, sizeof(int64_t) * (ns + 1), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  std::vector<int64_t> t_base((size_t)P.n_batches + 1);
  int64_t max_t = 1;
  for (int64_t b = 0; b <= P.n_batches; ++b) t_base[b] = toff_h[P.bat_pos[b]];
  for (int64_t b = 0; b < P.n_batches; ++b) max_t = std::max(max_t, t_base[b + 1] - t_base[b]);
  NFM_TRY(W.contrib.ensure(sizeof(double) * (size_t)max_t * M.nb * M.Kp));
  NFM_TRY(W.rec.ensure(sizeof(SampleRec) * (size_t)std::max<int64_t>(P.max_batch, 1)));
  NFM_TRY(W.partsA.ensure(sizeof(PartA) * (size_t)(P.max_batch + 1)));  // one partial per workgroup, workgroups of >= 1 wavefront
  NFM_TRY(W.partsB.ensure(sizeof(double) * 2 * (size_t)(2 * (P.max_unique * M.nb / kWavesPerBlock) + P.max_heavy * M.nb / kWavesPerBlock + 8)));
  NFM_TRY(W.hpart.ensure(sizeof(double) * (size_t)std::max<int64_t>(P.max_segs, 1) * M.nb * (2 * M.Kp + 4)));
  NFM_TRY(W.Dtab.ensure(sizeof(double) * 4 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Stab.ensure(sizeof(double) * 2 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Ftab.ensure(sizeof(double) * 2 * kFtab * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.out_acc.ensure(sizeof(double) * 2));
  NFM_TRY(W.itbuf.ensure(sizeof(double)));
  hipLaunchKernelGGL(k_set_double, dim3(1), dim3(1), 0, st, W.itbuf.as<double>(), (double)it0);
  NFM_HIP_CHECK(hipMemsetAsync(W.out_acc.p, 0, sizeof(double) * 2, st));
  if (P.n_batches > 0 && opt_kind == OPT_SGD) {
    TimedLaunch tl(ctx, "schedule");
    hipLaunchKernelGGL(k_schedule, dim3((unsigned)P.n_batches), dim3(kBlock), 0, st, O, M.fit_linear, M.fit_intercept,
                       P.bat_pos_dev.as<int64_t>(), W.itbuf.as<double>(), W.Dtab.as<double>(), W.Ftab.as<double>());
    hipLaunchKernelGGL(k_scale_prefix, dim3(1), dim3(1), 0, st, M.sc, W.Dtab.as<double>(), W.Stab.as<double>(), P.n_batches);
  }
  int rc = NFM_ERR_UNSUPPORTED;
#define NFM_RUN(LL)                                                                         \
  case LL:                                                                                  \
    rc = opt_kind == OPT_SGD ? run_ffm<LL, OPT_SGD>(ctx, X, M, O, P, W, t_base)             \
                             : run_ffm<LL, OPT_ADAGRAD>(ctx, X, M, O, P, W, t_base);        \
    break;
  switch (M.L) {
    NFM_RUN(1) NFM_RUN(2) NFM_RUN(4) NFM_RUN(8) NFM_RUN(16) NFM_RUN(32) NFM_RUN(64)
  }
#undef NFM_RUN
  NFM_TRY(rc);
  NFM_HIP_CHECK(hipMemcpyAsync(out2_host, W.out_acc.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  if (!defer_sync) NFM_HIP_CHECK(hipStreamSynchronize(st));
  return NFM_OK;
}

}  // namespace nfm
