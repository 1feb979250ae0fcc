// nimfm_amd/csrc/mb_fm_kernels.h -- the templates of mb_fm.hip (row phase, column phase, heavy-feature path, the
// per-batch host driver run_batches): included by mb_fm.hip (epoch driver, schedule kernels) and by mb_fm_inst.hip,
// which is compiled once per lanes-per-row value L so that the seven sets of instantiations build in parallel
// (one translation unit took four minutes).
#pragma once
#include "fm_device.h"
#include "mb.h"

namespace nfm {

#ifndef NFM_WAVE_STAGE2
#define NFM_WAVE_STAGE2 1
#endif
#ifndef NFM_COL_MINW
#define NFM_COL_MINW 1
#endif
#ifndef NFM_COL_D3
#define NFM_COL_D3 1
#endif
#ifndef NFM_REG_NTSEL
#define NFM_REG_NTSEL 0
#endif
#ifndef NFM_ADA2_MINW
#define NFM_ADA2_MINW 2
#endif

struct SampleRec {
  double dL, etaP, etaw, yhat;  // yhat: the sample's prediction (read back by predictAllWithGrad)
};
struct PartA {
  double loss, viol, acc0, acc1;
};

static_assert(sizeof(SampleRec) == 32 && sizeof(PartA) == 32, "record layout");

constexpr int kFtab = 64;  // touch counts 1..kFtab have a tabulated decay correction

// ------------------------------------------------------------------------------------------------
// row phase
// ------------------------------------------------------------------------------------------------
struct RowArgs {
  CsrView X;
  ModelView M;
  OptView O;
  const int64_t* perm;  // relative to begin, or null
  int64_t begin, p0;    // first sample of the batch = begin + p0 (position), identity when perm null
  int32_t len, use_stored, TA, nt;  // nt: parameter / state rows are streamed (tables far larger than the caches)
  double it_b;            // batch start relative to the epoch call; the absolute step is it0p[0] + it_b
  const double* it0p;     // device scalar: the optimizer's `it` at the start of the epoch call
  const double* scales;  // {scale_P, scale_w} at the batch start
  const double* scales_n;  // ... at the next batch start (SGD; singles are updated here)
  const int64_t* toff;     // touch offset of every sample of the epoch call (plan), or null
  const uint8_t* single;   // per nnz in sample order: 1 = feature touched once in this batch
  double* Abuf;          // [len][TA][Kp]
  SampleRec* rec;        // [len]
  PartA* parts;          // [gridDim.x]
};

// Rows that a batch reads once and writes at most once (tables far larger than L2 / the Infinity Cache) are moved with
// the non-temporal hint: they stop evicting the samples' A rows and records the column phase gathers from L2, and
// dirty lines no longer queue behind reads.  Measured on the cfg3 shape (AdaGrad, d = 1e6, k = 64, B = 8192):
// stores 264 -> 249 us row phase and 94.7 -> 90.8 us column phase, loads as well 249 -> 238 us.
__device__ __forceinline__ void st_nt(double* p, double2 v) {
  dev::v2d_t w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<dev::v2d_t*>(p));
}
__device__ __forceinline__ double2 ld_nt(const double* p) {
  const dev::v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const dev::v2d_t*>(p));
  return double2{v.x, v.y};
}
template <bool NT>
__device__ __forceinline__ void st_row(double* p, double2 v) {
  if (NT) st_nt(p, v);
  else *reinterpret_cast<double2*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ double2 ld_row(const double* p) {
  if (NT) return ld_nt(p);
  return *reinterpret_cast<const double2*>(p);
}
// forward over all orders; returns this lane's share of sum_o sum_s kernel (non-zero in slot 0
// only) and stores the A rows the column phase needs.  GEN = false: the model is a single order of
// degree 2 (the common case), whose A1 is also returned for the in-place update of singles.
template <int L, int SPLIT, bool GEN, class PS>
__device__ __forceinline__ double row_forward(const PS& ps, const CsrView& X, const ModelView& M, int64_t q0, int m,
                                              int m_tot, int slot, int l, int lane, bool valid,
                                              double* __restrict__ Arow, double2& A1_out) {
  if (!GEN) {
    double2 A1, A2;
    dev::anova_fwd_deg2<L, SPLIT>(ps, X, q0, m, m_tot, 0, M.Kp, slot, l, A1, A2);
    if (valid && slot == 0) *reinterpret_cast<double2*>(Arow + 2 * l) = A1;
    A1_out = A1;
    return slot == 0 ? (A1.x * A1.x - A2.x) / 2 + (A1.y * A1.y - A2.y) / 2 : 0.0;
  }
  A1_out = {0.0, 0.0};
  double part = 0.0;
  int slot_a = 0;
  for (int o = 0; o < M.nb; ++o) {
    const size_t blk = M.row(o, 0) * M.Kp;
    const int rstride = (int)M.rs * M.Kp;
    const int deg = M.deg_of(o);
    double2 ker;
    if (deg == 2) {
      double2 A1, A2;
      dev::anova_fwd_deg2<L, SPLIT>(ps, X, q0, m, m_tot, blk, rstride, slot, l, A1, A2);
      ker.x = (A1.x * A1.x - A2.x) / 2;
      ker.y = (A1.y * A1.y - A2.y) / 2;
      if (valid && slot == 0) *reinterpret_cast<double2*>(Arow + (size_t)slot_a * M.Kp + 2 * l) = A1;
    } else {
      double2 E[dev::kMaxDeg + 1];
      dev::anova_fwd_degn<L, SPLIT>(ps, X, q0, m, m_tot, blk, rstride, slot, l, lane, deg, E);
      ker = dev::pick(E, deg);
      if (valid && slot == 0) {
#pragma unroll
        for (int t = 1; t < dev::kMaxDeg; ++t)
          if (t < deg) *reinterpret_cast<double2*>(Arow + (size_t)(slot_a + t - 1) * M.Kp + 2 * l) = E[t];
      }
    }
    slot_a += deg - 1;
    if (slot == 0) part += ker.x + ker.y;
  }
  return part;
}

// In-place update of a sample's singles (features no other sample of the batch touches): the same
// arithmetic as the column phase with c = 1.  Entries in groups of kUnroll: flags, then (index, value),
// then the rows are all requested before the first use.  Called from the column-phase launch (extra
// workgroups), so the bandwidth-bound singles run beside the latency-bound multi-touch features.
template <int L, int SPLIT, int OPT>
__device__ __forceinline__ double singles_update(const CsrView& X, const ModelView& M, const OptView& O,
                                                 const uint8_t* __restrict__ sg, const double* scales_b,
                                                 const double* scales_nx, int64_t q0, int m, int m_tot, int slot, int l,
                                                 double dL, double etaP, double etaw, double2 A1, double itp, bool stored) {
  double r_viol = 0.0;
  {
    const double sP = scales_b[0], sw = scales_b[1], sPn = scales_nx[0], swn = scales_nx[1];
    const double tmpP = O.eta0 * itp * O.beta, denw = itp * O.eta0 * O.alpha;
    // groups of kUnroll entries: flags, then (index, value), then the rows are all requested before
    // the first use, so the second visit of the row costs one memory round trip per group
    for (int q = slot; q < m_tot; q += dev::kUnroll * SPLIT) {
      bool f[dev::kUnroll];
      int j[dev::kUnroll];
      double x[dev::kUnroll];
      double2 r0[dev::kUnroll], r1[dev::kUnroll], r2[dev::kUnroll];
      double w0[dev::kUnroll], w1[dev::kUnroll], w2[dev::kUnroll];
#pragma unroll
      for (int u = 0; u < dev::kUnroll; ++u) {
        const int qq = q + u * SPLIT;
        f[u] = qq < m_tot && sg[qq] != 0;
      }
#pragma unroll
      for (int u = 0; u < dev::kUnroll; ++u) {
        j[u] = 0;
        x[u] = 0.0;
        if (f[u]) dev::row_entry(X, q0, m, m_tot, q + u * SPLIT, j[u], x[u]);
      }
#pragma unroll
      for (int u = 0; u < dev::kUnroll; ++u) {
        r0[u] = r1[u] = r2[u] = {0.0, 0.0};
        w0[u] = w1[u] = w2[u] = 0.0;
        if (f[u]) {
          const size_t e = (size_t)j[u] * M.Kp + 2 * l;
          const bool has_w = M.fit_linear && j[u] < M.d && l == 0;
          if (OPT == OPT_SGD) {
            r0[u] = *reinterpret_cast<const double2*>(M.P + e);
            if (has_w) w0[u] = M.w[j[u]];
          } else {
            r1[u] = *reinterpret_cast<const double2*>(O.G + e);
            r2[u] = *reinterpret_cast<const double2*>(O.N + e);
            if (stored || O.track_viol) r0[u] = *reinterpret_cast<const double2*>(M.P + e);
            if (has_w) {
              w0[u] = M.w[j[u]];
              w1[u] = O.Gw[j[u]];
              w2[u] = O.Nw[j[u]];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < dev::kUnroll; ++u) {
        if (!f[u]) continue;
        const size_t e = (size_t)j[u] * M.Kp + 2 * l;
        const bool has_w = M.fit_linear && j[u] < M.d && l == 0;
        const double xv = x[u];
        if (OPT == OPT_SGD) {
          double2 st = r0[u];
          const double px = sP * st.x, py = sP * st.y;
          const double ax = etaP * (dL * (xv * (A1.x - px * xv)));
          const double ay = etaP * (dL * (xv * (A1.y - py * xv)));
          r_viol += fabs((ax + etaP * O.beta * px) / 1.0) + fabs((ay + etaP * O.beta * py) / 1.0);
          st.x = st.x - ax / sPn;
          st.y = st.y - ay / sPn;
          *reinterpret_cast<double2*>(M.P + e) = st;
          if (has_w) {
            const double wt = w0[u], wj = sw * wt;
            const double a0 = etaw * (dL * xv);
            r_viol += fabs(a0 + etaw * O.alpha * wj);
            M.w[j[u]] = wt - a0 / swn;
          }
        } else {
          double2 g2 = r1[u], n2 = r2[u], p;
          if (stored) {
            p = r0[u];
          } else {
            p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmpP);
            p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmpP);
            if (O.track_viol) {
              r_viol += fabs(r0[u].x - p.x) + fabs(r0[u].y - p.y);
              *reinterpret_cast<double2*>(M.P + e) = p;
            }
          }
          const double gx = dL * (xv * (A1.x - p.x * xv)), gy = dL * (xv * (A1.y - p.y * xv));
          g2.x += gx;
          g2.y += gy;
          n2.x += gx * gx;
          n2.y += gy * gy;
          *reinterpret_cast<double2*>(O.G + e) = g2;
          *reinterpret_cast<double2*>(O.N + e) = n2;
          if (has_w) {
            const double wt = w0[u], gw = w1[u], nw = w2[u];
            if (!stored) {
              const double wj = -O.eta0 * gw / (denw + sqrt(nw));
              r_viol += fabs(wt - wj);
              M.w[j[u]] = wj;
            }
            const double g = dL * xv;
            O.Gw[j[u]] = gw + g;
            O.Nw[j[u]] = nw + g * g;
          }
        }
      }
    }
  }
  return r_viol;
}

// value held by lane (slot + u * SPLIT) of this wavefront, u a compile-time constant: v_readlane into
// SGPRs (one per slot) and a select -- no LDS-pipe traffic and no vector registers held, unlike
// ds_bpermute, which the scheduler hoists by the dozen next to 128 registers of resident rows
template <int SPLIT>
__device__ __forceinline__ int lane_bcast_i(int v, int u, int slot) {
  int r = __builtin_amdgcn_readlane(v, u * SPLIT);
#pragma unroll
  for (int s = 1; s < SPLIT; ++s) {
    const int t = __builtin_amdgcn_readlane(v, u * SPLIT + s);
    r = slot == s ? t : r;
  }
  return r;
}
// a value every lane of the wavefront holds identically, moved to scalar registers (one sample per
// wavefront: the sample's target, step sizes, loss derivative ... need not occupy 64 lanes each)
__device__ __forceinline__ double wave_uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int SPLIT>
__device__ __forceinline__ double lane_bcast_d(double v, int u, int slot) {
  const int lo = lane_bcast_i<SPLIT>(__double2loint(v), u, slot), hi = lane_bcast_i<SPLIT>(__double2hiint(v), u, slot);
  return __hiloint2double(hi, lo);
}

// How the row phase reads a sample's CSR row (models with one order of degree 2):
//   MODE 0  streamed: (index, value) of every entry is loaded right before its parameter row -- two
//           dependent round trips per group of rows, three in the update of the singles.  Any row length.
//   MODE 1  held entries: the sample's L*SPLIT lanes load the whole row up front, E entries per lane
//           (rows of at most held_capacity = E*L*SPLIT entries); index, value, single flag and linear
//           weight stay in registers and are handed round with ds_bpermute, so a group of parameter
//           rows costs one round trip.  Rows are streamed, singles re-read.
//   MODE 2  MODE 1 + the parameter rows stay in registers between the forward pass and the update
//           of the singles (SGD, one sample per wavefront, E = 1): HBM sees every row read once and
//           the single-touch rows written once.
//   MODE 3  MODE 1 for rows of any length: the row is taken in chunks of held_capacity entries (a
//           separate instantiation: the chunk loop costs registers -- 38 vs 26 us on cfg2's row phase).
//   MODE 4  MODE 2 with the single-touch rows written back non-temporally (tables far larger than the caches:
//           the written rows are not read again before a later batch, and as ordinary stores they evict the A rows
//           and records the column phase is about to gather -- headline shape: column phase 57.7 -> 56.0 us, row
//           phase 98 -> 97 us; non-temporal LOADS of the same rows cost the column phase its Infinity-Cache hits
//           on the multi-touch rows: 57.7 -> 64.8 us)
template <int L, int SPLIT>
constexpr int held_entries() {  // E: entries per lane; 0 = no held mode for this lane mapping
  constexpr int LPS = L * SPLIT;
  return LPS >= kWave ? 1 : (LPS >= 8 ? (kWave / LPS > 4 ? 4 : kWave / LPS) : 0);
}

// MODE 2 at k = 64 needs 170 registers as written; three wavefronts per SIMD allow 168.  Asking for
// three costs 4 spilled registers and gains 7 % (107 -> 100 us on the headline shape); with the target,
// step sizes and loss derivative still in vector registers (180) the same request spilled 68 and lost.
#ifndef NFM_REG_MINW
#define NFM_REG_MINW 3
#endif
template <int L, int SPLIT, int OPT, bool GEN, int MODE, bool SING>
__global__ __launch_bounds__(kBlock, (MODE == 2 || MODE == 4 ? NFM_REG_MINW : 1)) void k_row_phase(RowArgs a) {
  constexpr int LPS = L * SPLIT, SPW = kWave / LPS, SPB = kWavesPerBlock * SPW;  // samples per wave / block
  constexpr int E = held_entries<L, SPLIT>() > 0 ? held_entries<L, SPLIT>() : 1;
  constexpr bool HELD = MODE >= 1 && held_entries<L, SPLIT>() > 0;  // GEN models: MODE 1 / 3 only (host)
  constexpr bool REG = (MODE == 2 || MODE == 4) && HELD && OPT == OPT_SGD && LPS == kWave;
  constexpr bool CHUNKED = MODE == 3;  // rows longer than one chunk of held entries
  constexpr int NQ = REG ? L : 1;       // row pieces per lane kept in registers
  constexpr int RPS = E * L;            // rows per slot in held mode
  double2 prow[NQ];
  __shared__ double s_y[SPB], s_yh[SPB], s_dL[SPB], s_etaP[SPB], s_etaw[SPB];
  __shared__ double s_part[4], s_viol[kWavesPerBlock];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int sidx = lane / LPS, slot = (lane / L) % SPLIT, l = lane % L;
  const int sib = wv * SPW + sidx;  // sample inside the block
  // one sample per wavefront: the sample index is wave-uniform, and so is everything read through it
  // (CSR header, target: scalar loads, scalar registers)
  const int pib = SPW == 1 ? __builtin_amdgcn_readfirstlane(blockIdx.x * SPB + sib) : blockIdx.x * SPB + sib;
  const bool valid = pib < a.len;
  int64_t i = 0, q0 = 0;
  int m = 0, m_tot = 0;
  double y = 0.0;
  if (valid) {
    const int64_t pos = a.p0 + pib;
    i = a.perm ? a.perm[pos] : a.begin + pos;
    q0 = X.indptr[i];
    m = (int)(X.indptr[i + 1] - q0);
    m_tot = m + M.n_aug;
    y = dev::target_of(X.y[i], M.task);
    if (SPW == 1) y = wave_uniform(y);
  }
  double* Arow = a.Abuf + (size_t)(valid ? pib : 0) * a.TA * M.Kp;
  double b0 = M.sc[SC_INTERCEPT];
  double part = 0.0;
  double2 A1 = {0.0, 0.0};
  // held mode: this lane's entries of the sample's row (entry q = e * LPS + lane-in-sample)
  int jq[E], fq[E];
  double xq[E], wq[E], gwq[E], nwq[E];
  const int lis = lane % LPS, sbase = lane - lis;
  int m_max = 0;  // longest row among the wavefront's samples (wave-uniform loop bound)
  const double itp = (a.it0p[0] + a.it_b) - 1.0;  // AdaGrad: it' = it_b - 1 (adagrad.nim:90)
  const bool stored = a.use_stored != 0;
  // one sample per wavefront: the step sizes depend on the sample's position only, so they are
  // formed BEFORE the parameter rows are gathered -- pow / division sequences need dozens of
  // registers, which must not coincide with the rows held in registers (MODE 2)
  double etaP = 0.0, etaw = 0.0, eta_b = 0.0;
  if (SPW == 1 && NFM_WAVE_STAGE2 && OPT == OPT_SGD) {
    const double it = (a.it0p[0] + a.it_b) + (double)pib;
    etaP = wave_uniform(dev::get_eta(O.sched, O.eta0, O.power, O.beta, it));
    etaw = wave_uniform(dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it));
    if (M.fit_intercept) eta_b = wave_uniform(dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it));
  }
  // ---- 1. forward: yhat of every sample of the block ----
  // ---- held entries: a chunk of CAP = E * LPS entries of the row in one round trip (index, value, single
  // flag, linear weight), linear term lane-parallel.  Rows longer than CAP take several chunks.
  constexpr int CAP = E * LPS;
  const uint8_t* sg = (HELD && a.single != nullptr && valid) ? a.single + a.toff[a.p0 + pib] : nullptr;
  auto load_chunk = [&](int base, bool with_linear) {
    const double sw = OPT == OPT_SGD ? a.scales[1] : 1.0;
    const double denw = itp * O.eta0 * O.alpha;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int q = base + e * LPS + lis;
      dev::row_entry(X, q0, m, m_tot, q, jq[e], xq[e]);
      fq[e] = (sg != nullptr && q < m_tot) ? (int)sg[q] : 0;
      wq[e] = gwq[e] = nwq[e] = 0.0;
      if (q < m) {
        wq[e] = M.w[jq[e]];
        double wj = sw * wq[e];
        if (OPT == OPT_ADAGRAD && M.fit_linear && (!stored || a.single != nullptr)) {
          gwq[e] = O.Gw[jq[e]];
          nwq[e] = O.Nw[jq[e]];
          if (!stored) wj = -O.eta0 * gwq[e] / (denw + sqrt(nwq[e]));
        }
        if (with_linear) part += wj * xq[e];
      }
    }
  };
  if (HELD) {
    if (OPT == OPT_ADAGRAD && !stored && M.fit_intercept)
      b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
    m_max = m_tot;
#pragma unroll
    for (int s = LPS; s < kWave; s <<= 1) {
      const int o = __shfl_xor(m_max, s, kWave);
      m_max = o > m_max ? o : m_max;
    }
    m_max = __builtin_amdgcn_readfirstlane(m_max);
    // MODE 2: the first chunk's parameter rows stay in registers (prow); further chunks and MODE 1
    // stream the rows in groups of U, addressed from the held entries: one round trip per group
    // rows requested together per lane: 4 for one order of degree 2 (8 costs a wavefront of occupancy: cfg2 26.4 ->
    // 25.1 us, AdaGrad k = 64 314 -> 295 us), 8 for the multi-order walk
    constexpr int UW = GEN ? dev::kFwdUnroll : 4;
    constexpr int U = RPS < UW ? RPS : UW;
    auto held_forward = [&](auto ps) {
      double2 a1 = {0.0, 0.0}, a2 = {0.0, 0.0};
      for (int base = 0; base == 0 || (CHUNKED && base < m_max); base += CAP) {
        load_chunk(base, true);
        if (REG && base == 0) {
          const double sP = a.scales[0];
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const int jj = lane_bcast_i<SPLIT>(jq[0], u, slot);
#if NFM_REG_NTSEL
            // MODE 4: a single-touch row is read here and never again in this batch (streamed); a multi-touch row is
            // read again by the column phase and should stay in the Infinity Cache
            const int ff = MODE == 4 ? lane_bcast_i<SPLIT>(fq[0], u, slot) : 0;
            if (ff) prow[u] = ld_nt(M.P + (size_t)jj * M.Kp + 2 * l);
            else prow[u] = *reinterpret_cast<const double2*>(M.P + (size_t)jj * M.Kp + 2 * l);
#else
            prow[u] = *reinterpret_cast<const double2*>(M.P + (size_t)jj * M.Kp + 2 * l);
#endif
          }
#pragma unroll
          for (int u = 0; u < NQ; ++u) {
            const double xx = lane_bcast_d<SPLIT>(xq[0], u, slot);
            const double tx = xx * (sP * prow[u].x), ty = xx * (sP * prow[u].y);
            a1.x += tx;
            a1.y += ty;
            a2.x += tx * tx;
            a2.y += ty * ty;
          }
        } else {
#pragma unroll
          for (int u0 = 0; u0 < RPS; u0 += U) {
            if (base + u0 * SPLIT >= m_max) break;
            int jj[U];
            double xx[U];
            double2 pp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int r = u0 + u, src = sbase + slot + (r % L) * SPLIT;
              jj[u] = __shfl(jq[r / L], src, kWave);
              xx[u] = dev::shfl_d(xq[r / L], src);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) pp[u] = ps.load((size_t)jj[u] * M.Kp + 2 * l);  // past the end: (row 0, x = 0)
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const double tx = xx[u] * pp[u].x, ty = xx[u] * pp[u].y;
              a1.x += tx;
              a1.y += ty;
              a2.x += tx * tx;
              a2.y += ty * ty;
            }
          }
        }
      }
#pragma unroll
      for (int s = L; s < L * SPLIT; s <<= 1) {
        a1.x += dev::shfl_xor_d(a1.x, s);
        a1.y += dev::shfl_xor_d(a1.y, s);
        a2.x += dev::shfl_xor_d(a2.x, s);
        a2.y += dev::shfl_xor_d(a2.y, s);
      }
      A1 = a1;
      if (valid && slot == 0) *reinterpret_cast<double2*>(Arow + 2 * l) = A1;
      if (slot == 0) part += (a1.x * a1.x - a2.x) / 2 + (a1.y * a1.y - a2.y) / 2;
    };
    // models with several orders and / or degree >= 3: the held entries serve every order (one chunk) --
    // the streamed kernel reads the CSR row again for each of them
    auto held_forward_gen = [&](auto ps) {
      int slot_a = 0;
      for (int o = 0; o < M.nb; ++o) {
        const size_t blk = M.row(o, 0) * M.Kp;
        const size_t rstride = (size_t)M.rs * M.Kp;
        const int deg = M.deg_of(o);
        double2 E[dev::kMaxDeg + 1];
#pragma unroll
        for (int t = 0; t <= dev::kMaxDeg; ++t) E[t] = {0.0, 0.0};
        E[0] = {1.0, 1.0};
        double2 a2 = {0.0, 0.0};  // degree 2: sum of squares (E[1] is the plain sum)
        for (int base = 0; base == 0 || (CHUNKED && base < m_max); base += CAP) {
          if (CHUNKED || o == 0) load_chunk(base, o == 0);
#pragma unroll
          for (int u0 = 0; u0 < RPS; u0 += U) {
            if (base + u0 * SPLIT >= m_max) break;
            int jj[U];
            double xx[U];
            double2 pp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int r = u0 + u, src = sbase + slot + (r % L) * SPLIT;
              jj[u] = __shfl(jq[r / L], src, kWave);
              xx[u] = dev::shfl_d(xq[r / L], src);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) pp[u] = ps.load(blk + (size_t)jj[u] * rstride + 2 * l);
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const double tx = xx[u] * pp[u].x, ty = xx[u] * pp[u].y;
              if (deg == 2) {
                E[1].x += tx;
                E[1].y += ty;
                a2.x += tx * tx;
                a2.y += ty * ty;
              } else {  // the DP of optimizer/sgd.nim:152-159, entry by entry
#pragma unroll
                for (int t = dev::kMaxDeg; t >= 1; --t)
                  if (t <= deg) {
                    E[t].x += E[t - 1].x * pp[u].x * xx[u];  // the reference's order of the two products
                    E[t].y += E[t - 1].y * pp[u].y * xx[u];
                  }
              }
            }
          }
        }
        double2 ker;
        if (deg == 2) {
#pragma unroll
          for (int s = L; s < L * SPLIT; s <<= 1) {
            E[1].x += dev::shfl_xor_d(E[1].x, s);
            E[1].y += dev::shfl_xor_d(E[1].y, s);
            a2.x += dev::shfl_xor_d(a2.x, s);
            a2.y += dev::shfl_xor_d(a2.y, s);
          }
          ker.x = (E[1].x * E[1].x - a2.x) / 2;
          ker.y = (E[1].y * E[1].y - a2.y) / 2;
          if (valid && slot == 0) *reinterpret_cast<double2*>(Arow + (size_t)slot_a * M.Kp + 2 * l) = E[1];
        } else {
          dev::combine_slots_degn<L, SPLIT>(E, deg, lane);
          ker = dev::pick(E, deg);
          if (valid && slot == 0) {
#pragma unroll
            for (int t = 1; t < dev::kMaxDeg; ++t)
              if (t < deg) *reinterpret_cast<double2*>(Arow + (size_t)(slot_a + t - 1) * M.Kp + 2 * l) = E[t];
          }
        }
        slot_a += deg - 1;
        if (slot == 0) part += ker.x + ker.y;
      }
    };
    if (GEN) {
      if (OPT == OPT_SGD)
        held_forward_gen(dev::PlainParams{M.P, a.scales[0]});
      else if (stored)
        held_forward_gen(dev::PlainParams{M.P, 1.0});
      else
        held_forward_gen(dev::AdaParams{O.G, O.N, O.eta0, O.eta0 * itp * O.beta});
    } else if (OPT == OPT_SGD)
      held_forward(dev::PlainParams{M.P, a.scales[0]});
    else if (stored)
      held_forward(dev::PlainParams{M.P, 1.0});
    else
      held_forward(dev::AdaParams{O.G, O.N, O.eta0, O.eta0 * itp * O.beta});
  } else if (OPT == OPT_SGD) {
    const double sP = a.scales[0], sw = a.scales[1];
    for (int q = slot * L + l; q < m; q += LPS) part += (sw * M.w[X.indices[q0 + q]]) * X.data[q0 + q];
    const dev::PlainParams ps{M.P, sP};
    part += row_forward<L, SPLIT, GEN>(ps, X, M, q0, m, m_tot, slot, l, lane, valid, Arow, A1);
  } else {
    if (!stored && M.fit_intercept) b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
    const double denw = itp * O.eta0 * O.alpha;
    for (int q = slot * L + l; q < m; q += LPS) {
      const int j = X.indices[q0 + q];
      double wj = M.w[j];
      if (!stored && M.fit_linear) wj = -O.eta0 * O.Gw[j] / (denw + sqrt(O.Nw[j]));
      part += wj * X.data[q0 + q];
    }
    if (stored) {
      const dev::PlainParams ps{M.P, 1.0};
      part += row_forward<L, SPLIT, GEN>(ps, X, M, q0, m, m_tot, slot, l, lane, valid, Arow, A1);
    } else {
      const dev::AdaParams ps{O.G, O.N, O.eta0, O.eta0 * itp * O.beta};
      part += row_forward<L, SPLIT, GEN>(ps, X, M, q0, m, m_tot, slot, l, lane, valid, Arow, A1);
    }
  }
#pragma unroll
  for (int s = 1; s < LPS; s <<= 1) part += dev::shfl_xor_d(part, s);
  double dL, yh_w = 0.0;
  if (SPW == 1 && NFM_WAVE_STAGE2) {
    // ---- 2'. one sample per wavefront (k > 32): every wavefront finishes its own sample, no
    // workgroup barrier between the forward pass and the singles update, so the four wavefronts of a
    // workgroup drift apart and their load and store phases overlap ----
    const double yh = wave_uniform(b0 + part);
    yh_w = yh;
    dL = wave_uniform(dev::loss_grad(O.loss, O.loss_param, y, yh));
    double r_acc0 = 0.0, r_acc1 = 0.0;
    if (OPT == OPT_SGD) {
      if (M.fit_intercept) {
        r_acc0 = eta_b * dL;
        r_acc1 = eta_b;
      }
    } else if (M.fit_intercept) {
      r_acc0 = dL;
      r_acc1 = dL * dL;
    }
    if (lane == 0) {
      if (valid) a.rec[pib] = SampleRec{dL, etaP, etaw, yh};
      s_dL[sib] = valid ? r_acc0 : 0.0;
      s_etaP[sib] = valid ? r_acc1 : 0.0;
    }  // the loss VALUE (log / exp) waits until the rows are written back, see below
  } else {
  if (slot == 0 && l == 0) {
    s_y[sib] = y;
    s_yh[sib] = b0 + part;
  }
  __syncthreads();
  // ---- 2. loss, dL and step sizes: ONE LANE PER SAMPLE in the first wavefront, so the
  // transcendental work (exp/log of the loss, the divisions of the schedules) is issued once per
  // 64 samples instead of once per sample; the block's partial sums fall out of one wave reduction
  if (wv == 0) {
    double r_loss = 0.0, r_acc0 = 0.0, r_acc1 = 0.0;
    for (int t = lane; t < SPB; t += kWave) {
      const int pt = blockIdx.x * SPB + t;
      if (pt < a.len) {
        const double yt = s_y[t], yh = s_yh[t];
        const double dL = dev::loss_grad(O.loss, O.loss_param, yt, yh);
        r_loss += dev::loss_value(O.loss, O.loss_param, yt, yh);
        double etaP = 0.0, etaw = 0.0;
        if (OPT == OPT_SGD) {
          const double it = (a.it0p[0] + a.it_b) + (double)pt;
          etaP = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
          etaw = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
          if (M.fit_intercept) {
            const double eta0 = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it);
            r_acc0 += eta0 * dL;
            r_acc1 += eta0;
          }
        } else if (M.fit_intercept) {
          r_acc0 += dL;
          r_acc1 += dL * dL;
        }
        a.rec[pt] = SampleRec{dL, etaP, etaw, yh};
        s_dL[t] = dL;
        s_etaP[t] = etaP;
        s_etaw[t] = etaw;
      }
    }
    r_loss = dev::wave_sum(r_loss);
    r_acc0 = dev::wave_sum(r_acc0);
    r_acc1 = dev::wave_sum(r_acc1);
    if (lane == 0) {
      s_part[0] = r_loss;
      s_part[2] = r_acc0;
      s_part[3] = r_acc1;
    }
  }
  __syncthreads();
  dL = s_dL[sib];
  etaP = s_etaP[sib];
  etaw = s_etaw[sib];
  }
  double r_viol = 0.0;
  // ---- 3. singles: a feature this sample alone touches in the batch gets its update right here
  // (same arithmetic as the column phase with c = 1), so its row is read and written once ----
  if (HELD && a.single != nullptr) {
    const double sP = OPT == OPT_SGD ? a.scales[0] : 1.0, sw = OPT == OPT_SGD ? a.scales[1] : 1.0;
    const double sPn = OPT == OPT_SGD ? a.scales_n[0] : 1.0, swn = OPT == OPT_SGD ? a.scales_n[1] : 1.0;
    const double tmpP = O.eta0 * itp * O.beta, denw = itp * O.eta0 * O.alpha;
    // register-resident path: one reciprocal instead of two fp64 divisions per row (a division holds ~10
    // registers while it runs, and the scheduler overlaps several)
    const double rsPn = 1.0 / sPn;
    const bool chunks = CHUNKED && m_max > CAP;  // several chunks: the entries are loaded again per chunk
    for (int base = 0; base == 0 || (CHUNKED && base < m_max); base += CAP) {
      if (chunks) load_chunk(base, false);
      if (REG && base == 0) {
  #pragma unroll
        for (int u = 0; u < NQ; ++u) {
          const int f = lane_bcast_i<SPLIT>(fq[0], u, slot);  // read with every lane active
          const int j = lane_bcast_i<SPLIT>(jq[0], u, slot);
          const double xv = lane_bcast_d<SPLIT>(xq[0], u, slot);
          if (f) {
            double2 st = prow[u];
            const double px = sP * st.x, py = sP * st.y;
            const double ax = etaP * (dL * (xv * (A1.x - px * xv)));
            const double ay = etaP * (dL * (xv * (A1.y - py * xv)));
            r_viol += fabs((ax + etaP * O.beta * px) / 1.0) + fabs((ay + etaP * O.beta * py) / 1.0);
            st.x = st.x - ax * rsPn;
            st.y = st.y - ay * rsPn;
            st_row<MODE == 4>(M.P + (size_t)j * M.Kp + 2 * l, st);
          }
        }
      } else {
        // singles re-read, but addressed from the held entries: per group of V rows one round trip
        // (flags, indices and values come from registers), then the stores
        constexpr int V = RPS < dev::kUnroll ? RPS : dev::kUnroll;
  #pragma unroll
        for (int u0 = 0; u0 < RPS; u0 += V) {
          if (base + u0 * SPLIT >= m_max) break;
          int f[V], j[V];
          double x[V];
          double2 r0[V], r1[V], r2[V];
  #pragma unroll
          for (int u = 0; u < V; ++u) {
            const int r = u0 + u, src = sbase + slot + (r % L) * SPLIT;
            f[u] = __shfl(fq[r / L], src, kWave);
            j[u] = __shfl(jq[r / L], src, kWave);
            x[u] = dev::shfl_d(xq[r / L], src);
          }
  #pragma unroll
          for (int u = 0; u < V; ++u) {
            r0[u] = r1[u] = r2[u] = {0.0, 0.0};
            if (f[u]) {
              const size_t e = (size_t)j[u] * M.Kp + 2 * l;
              if (OPT == OPT_SGD) {
                r0[u] = *reinterpret_cast<const double2*>(M.P + e);
              } else {
                r1[u] = *reinterpret_cast<const double2*>(O.G + e);
                r2[u] = *reinterpret_cast<const double2*>(O.N + e);
                if (stored || O.track_viol) r0[u] = *reinterpret_cast<const double2*>(M.P + e);
              }
            }
          }
  #pragma unroll
          for (int u = 0; u < V; ++u) {
            if (!f[u]) continue;
            const size_t e = (size_t)j[u] * M.Kp + 2 * l;
            const double xv = x[u];
            if (OPT == OPT_SGD) {
              double2 st = r0[u];
              const double px = sP * st.x, py = sP * st.y;
              const double ax = etaP * (dL * (xv * (A1.x - px * xv)));
              const double ay = etaP * (dL * (xv * (A1.y - py * xv)));
              r_viol += fabs((ax + etaP * O.beta * px) / 1.0) + fabs((ay + etaP * O.beta * py) / 1.0);
              st.x = st.x - ax / sPn;
              st.y = st.y - ay / sPn;
              *reinterpret_cast<double2*>(M.P + e) = st;
            } else {
              double2 g2 = r1[u], n2 = r2[u], p;
              if (stored) {
                p = r0[u];
              } else {
                p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmpP);
                p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmpP);
                if (O.track_viol) {
                  r_viol += fabs(r0[u].x - p.x) + fabs(r0[u].y - p.y);
                  *reinterpret_cast<double2*>(M.P + e) = p;
                }
              }
              const double gx = dL * (xv * (A1.x - p.x * xv)), gy = dL * (xv * (A1.y - p.y * xv));
              g2.x += gx;
              g2.y += gy;
              n2.x += gx * gx;
              n2.y += gy * gy;
              *reinterpret_cast<double2*>(O.G + e) = g2;
              *reinterpret_cast<double2*>(O.N + e) = n2;
            }
          }
        }
      }
      // the linear term of the singles, one entry per lane
      if (M.fit_linear) {
  #pragma unroll
        for (int e = 0; e < E; ++e) {
          if (!fq[e] || base + e * LPS + lis >= m) continue;
          if (OPT == OPT_SGD) {
            const double wj = sw * wq[e];
            const double a0 = etaw * (dL * xq[e]);
            r_viol += fabs(a0 + etaw * O.alpha * wj);
            M.w[jq[e]] = wq[e] - a0 / swn;
          } else {
            if (!stored) {
              const double wj = -O.eta0 * gwq[e] / (denw + sqrt(nwq[e]));
              r_viol += fabs(wq[e] - wj);
              M.w[jq[e]] = wj;
            }
            const double g = dL * xq[e];
            O.Gw[jq[e]] = gwq[e] + g;
            O.Nw[jq[e]] = nwq[e] + g * g;
          }
        }
      }
    }
  } else if (SING && !GEN && a.single != nullptr && valid) {
    // stage 3 (sparse regime): singles updated right after the forward pass, while their rows are
    // as warm as they will get (measured 157 us fused vs 66 + 109 us as a separate kernel, k = 64)
    r_viol += singles_update<L, SPLIT, OPT>(X, M, O, a.single + a.toff[a.p0 + pib], a.scales, a.scales_n, q0, m, m_tot, slot, l,
                                            dL, etaP, etaw, A1, itp, stored);
  }
  r_viol = dev::wave_sum(r_viol);
  if (lane == 0) s_viol[wv] = r_viol;
  if (SPW == 1 && NFM_WAVE_STAGE2 && lane == 0) s_y[sib] = valid ? dev::loss_value(O.loss, O.loss_param, y, yh_w) : 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    PartA p{0.0, 0.0, 0.0, 0.0};
    if (SPW == 1 && NFM_WAVE_STAGE2) {
      for (int w_ = 0; w_ < kWavesPerBlock; ++w_) {
        p.loss += s_y[w_];
        p.acc0 += s_dL[w_];
        p.acc1 += s_etaP[w_];
      }
    } else {
      p = PartA{s_part[0], 0.0, s_part[2], s_part[3]};
    }
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) p.viol += s_viol[w_];
    a.parts[blockIdx.x] = p;
  }
}

// ------------------------------------------------------------------------------------------------
// row phase, AdaGrad at 32 < k <= 64 (cfg3): TWO wavefronts per sample, state resident in registers
// ------------------------------------------------------------------------------------------------
// The parameters of AdaGrad are a function of (g_sum, g_norm, it) (optimizer/adagrad.nim:96-98), so the forward pass
// reads BOTH state rows of every feature of the sample, and the in-place update of the single-touch features (about
// 60 % of a batch's touches at the cfg3 shape) needs them again: k_row_phase<.., MODE 1> reads them a second time
// (39 KB of the 182 KB a sample moves).  One wavefront cannot hold 64 rows x 2 tensors x 512 B (256 + 140 registers,
// one wavefront per SIMD: measured slower).  Here a sample is split by FACTORS over the two
// wavefronts of a pair: each holds the g_sum / g_norm HALF rows (32 factors = 256 B, 16 lanes x 16 B) of up to 64
// entries in 128 registers, the pair exchanges one double through LDS (its share of the prediction), and the update
// of the single-touch rows issues no state load at all.  Rows of at most 64 entries (incl. dummies), one order of
// degree 2, singles updated in the row phase, not the stored-parameter batch (host: run_batches).
template <int OPT, bool NT>  // NT: state / parameter rows streamed (tables far larger than the caches)
__global__ __launch_bounds__(kBlock, NFM_ADA2_MINW) void k_row_phase_ada2(RowArgs a) {
  static_assert(OPT == OPT_ADAGRAD, "AdaGrad only");
  constexpr int SPLIT = 4, NU = 16, SPB = 2;  // row slots per wavefront, rows per slot, samples per workgroup
  __shared__ double s_part[SPB][2], s_loss[SPB], s_acc0[SPB], s_acc1[SPB], s_viol[kWavesPerBlock];
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int sib = wv >> 1, half = wv & 1;
  const int slot = lane >> 4, l = lane & 15;
  const int pib = __builtin_amdgcn_readfirstlane(blockIdx.x * SPB + sib);
  const bool valid = pib < a.len;
  int64_t q0 = 0;
  int m = 0, m_tot = 0;
  double y = 0.0;
  if (valid) {
    const int64_t pos = a.p0 + pib;
    const int64_t i = a.perm ? a.perm[pos] : a.begin + pos;
    q0 = X.indptr[i];
    m = (int)(X.indptr[i + 1] - q0);
    m_tot = m + M.n_aug;
    y = wave_uniform(dev::target_of(X.y[i], M.task));
  }
  const double itp = (a.it0p[0] + a.it_b) - 1.0;  // it' = it_b - 1 (adagrad.nim:90)
  const double tmpP = O.eta0 * itp * O.beta, denw = itp * O.eta0 * O.alpha;
  const size_t eoff = (size_t)half * 32 + 2 * l;  // this lane's factor pair inside a parameter row
  // the sample's entries, one per lane (entry q = lane; past the end: (row 0, x = 0), no flag)
  int jq, fq;
  double xq;
  dev::row_entry(X, q0, m, m_tot, lane, jq, xq);
  const uint8_t* sg = valid ? a.single + a.toff[a.p0 + pib] : nullptr;
  fq = (sg != nullptr && lane < m_tot) ? (int)sg[lane] : 0;
  // linear term and intercept: the first wavefront of the pair
  double wq = 0.0, gwq = 0.0, nwq = 0.0, part = 0.0;
  if (half == 0 && lane < m) {
    wq = M.w[jq];
    double wj = wq;
    if (M.fit_linear) {
      gwq = O.Gw[jq];
      nwq = O.Nw[jq];
      wj = -O.eta0 * gwq / (denw + sqrt(nwq));
    }
    part = wj * xq;
  }
  // ---- 1. all state half rows in flight at once, then the forward pass from registers ----
  double2 G[NU], N[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int jj = lane_bcast_i<SPLIT>(jq, u, slot);
    const size_t e = (size_t)jj * M.Kp + eoff;
    G[u] = ld_row<NT>(O.G + e);
    N[u] = ld_row<NT>(O.N + e);
  }
  double2 a1 = {0.0, 0.0}, a2 = {0.0, 0.0};
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const double xx = lane_bcast_d<SPLIT>(xq, u, slot);
    const double px = dev::adagrad_param(G[u].x, N[u].x, O.eta0, tmpP), py = dev::adagrad_param(G[u].y, N[u].y, O.eta0, tmpP);
    const double tx = xx * px, ty = xx * py;
    a1.x += tx;
    a1.y += ty;
    a2.x += tx * tx;
    a2.y += ty * ty;
    // one row at a time: interleaving the sqrt / divide sequences of several rows costs dozens of registers next to
    // the 128 that hold the state
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 16; s < kWave; s <<= 1) {
    a1.x += dev::shfl_xor_d(a1.x, s);
    a1.y += dev::shfl_xor_d(a1.y, s);
    a2.x += dev::shfl_xor_d(a2.x, s);
    a2.y += dev::shfl_xor_d(a2.y, s);
  }
  if (valid && slot == 0) *reinterpret_cast<double2*>(a.Abuf + (size_t)pib * a.TA * M.Kp + eoff) = a1;
  if (slot == 0) part += (a1.x * a1.x - a2.x) / 2 + (a1.y * a1.y - a2.y) / 2;
  part = dev::wave_sum(part);
  if (lane == 0) s_part[sib][half] = part;
  __syncthreads();
  // ---- 2. prediction, loss derivative (both wavefronts of the pair form the same values) ----
  double b0 = M.sc[SC_INTERCEPT];
  if (M.fit_intercept) b0 = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * itp * O.alpha0);
  const double yh = wave_uniform(b0 + (s_part[sib][0] + s_part[sib][1]));
  const double dL = wave_uniform(dev::loss_grad(O.loss, O.loss_param, y, yh));
  if (half == 0 && lane == 0) {
    if (valid) a.rec[pib] = SampleRec{dL, 0.0, 0.0, yh};
    s_acc0[sib] = (valid && M.fit_intercept) ? dL : 0.0;
    s_acc1[sib] = (valid && M.fit_intercept) ? dL * dL : 0.0;
  }
  // ---- 3. single-touch features: updated from the resident state, written once ----
  // (the entries are made opaque here: otherwise the compiler keeps the 16 row addresses of the forward pass -- 96
  // registers for the three tensors -- alive next to the state instead of re-deriving them from jq)
  asm volatile("" : "+v"(jq), "+v"(fq), "+v"(xq));
  double r_viol = 0.0;
#ifndef NFM_ADA2_V
#define NFM_ADA2_V 8
#endif
  constexpr int V = NFM_ADA2_V;  // stored-parameter rows requested together (one round trip per group)
#pragma unroll
  for (int u0 = 0; u0 < NU; u0 += V) {
    int f[V], j[V];
    double xv[V];
    double2 po[V];
#pragma unroll
    for (int u = 0; u < V; ++u) {
      f[u] = lane_bcast_i<SPLIT>(fq, u0 + u, slot);
      j[u] = lane_bcast_i<SPLIT>(jq, u0 + u, slot);
      xv[u] = lane_bcast_d<SPLIT>(xq, u0 + u, slot);
    }
    if (O.track_viol) {  // adagrad.nim:99: sum |P_old - P_new| needs the parameters as last stored
#pragma unroll
      for (int u = 0; u < V; ++u) {
        po[u] = {0.0, 0.0};
        if (f[u]) po[u] = ld_row<NT>(M.P + (size_t)j[u] * M.Kp + eoff);
      }
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      if (!f[u]) continue;
      const size_t e = (size_t)j[u] * M.Kp + eoff;
      double2 g2 = G[u0 + u], n2 = N[u0 + u], p;
      // recomputed, not carried over from the forward pass (the compiler would park 64 more values per lane in scratch)
      asm volatile("" : "+v"(g2.x), "+v"(g2.y), "+v"(n2.x), "+v"(n2.y));
      p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmpP);
      p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmpP);
      if (O.track_viol) {
        r_viol += fabs(po[u].x - p.x) + fabs(po[u].y - p.y);
        st_row<NT>(M.P + e, p);
      }
      const double gx = dL * (xv[u] * (a1.x - p.x * xv[u])), gy = dL * (xv[u] * (a1.y - p.y * xv[u]));
      g2.x += gx;
      g2.y += gy;
      n2.x += gx * gx;
      n2.y += gy * gy;
      st_row<NT>(O.G + e, g2);
      st_row<NT>(O.N + e, n2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (half == 0 && M.fit_linear && fq && lane < m) {  // fit_linear.nim:50-57, one entry per lane
    const double wj = -O.eta0 * gwq / (denw + sqrt(nwq));
    r_viol += fabs(wq - wj);
    M.w[jq] = wj;
    const double g = dL * xq;
    O.Gw[jq] = gwq + g;
    O.Nw[jq] = nwq + g * g;
  }
  r_viol = dev::wave_sum(r_viol);
  if (lane == 0) {
    s_viol[wv] = r_viol;
    if (half == 0) s_loss[sib] = valid ? dev::loss_value(O.loss, O.loss_param, y, yh) : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PartA p{0.0, 0.0, 0.0, 0.0};
    for (int s_ = 0; s_ < SPB; ++s_) {
      p.loss += s_loss[s_];
      p.acc0 += s_acc0[s_];
      p.acc1 += s_acc1[s_];
    }
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) p.viol += s_viol[w_];
    a.parts[blockIdx.x] = p;
  }
}

// ------------------------------------------------------------------------------------------------
// column phase (+ batch close in workgroup 0)
// ------------------------------------------------------------------------------------------------
struct ColArgs {
  CsrView X;
  ModelView M;
  OptView O;
  // singles workgroups (the first nS of the launch): one wavefront per sample of the batch
  const int64_t* perm;
  const int64_t* toff;
  const uint8_t* single;
  int64_t begin, p0;
  int32_t len_i, nS;
  const int32_t* ucol;
  const int64_t* uptr;
  const int32_t* ucol_s;  // the batch's unique features by descending touch count (plan.h): what k_col_phase walks
  const int64_t* ubeg_s;
  const int32_t* ucnt_s;
  const int32_t* tpos;
  const double* tx;
  int64_t u0, u1;
  const double* scales_b;  // {scale_P, scale_w} at the batch start
  const double* scales_n;  // ... at the next batch start
  const double* Dtab_b;    // SGD: {D_P, D_w, D_0, D_0^(1/len)} of this batch
  const double* Ftab_b;    // SGD: [2][kFtab] decay corrections by touch count
  const double* Abuf;
  const SampleRec* rec;
  double* parts;            // this batch's per-block viol partials [gridDim.x]
  const PartA* partsA;      // row phase partials of this batch [nA]
  const double* parts_prev; // previous batch's per-block viol partials [n_prev]
  double* out_acc;          // {loss_sum, viol_sum}
  double it_b, len;         // it_b as in RowArgs
  const double* it0p;
  int32_t TA, use_stored, nA, n_prev;
};

struct WAcc {  // linear-term accumulators of one feature
  double a0 = 0.0, a1 = 0.0;
};

// one parameter block (order) of one unique feature: this lane's factor pair at element e.
// do_w: also accumulate the linear term's sums over the same touches.
// MODE 0: the whole feature (walk the touches, apply).  Heavy features (plan.h) are done in two steps:
// MODE 1 walks ONE segment of the touches and stores the partial sums to hp (no side effects),
// MODE 2 adds the nseg segments' partial sums in segment order and applies.  hp points at this
// lane's slot of the feature's first segment; a segment's record is PW doubles:
// [acc Kp][accn Kp][seta, wacc.a0, wacc.a1, -].
template <int OPT, bool GEN, int TU, int LG, int MODE>
__device__ __forceinline__ double col_block(const ColArgs& a, size_t e, int deg, int slot, int l, int64_t t0, int64_t t1,
                                            double sP, double sPn, double fP, bool do_w, WAcc& wacc, double c_total = 0.0,
                                            double* hp = nullptr, int64_t nseg = 0, int PW = 0, int seg_stride = 0) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  double viol = 0.0;
  double2 stored = {0.0, 0.0}, g2 = {0.0, 0.0}, n2 = {0.0, 0.0}, p;
  if (OPT == OPT_SGD) {
    stored = dev::ld_stream(M.P + e);
    p.x = sP * stored.x;
    p.y = sP * stored.y;
  } else if (OPT == OPT_PSGD) {  // minibatch_psgd.nim:72: the parameters as they stand, no lazy scale
    stored = dev::ld_stream(M.P + e);
    p = stored;
  } else {
    g2 = dev::ld_stream(O.G + e);
    n2 = dev::ld_stream(O.N + e);
    if (a.use_stored) {
      p = dev::ld_stream(M.P + e);
    } else {
      const double tmp = O.eta0 * ((a.it0p[0] + a.it_b) - 1.0) * O.beta;
      p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmp);
      p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmp);
      if (O.track_viol && MODE != 1) {  // adagrad.nim:96-99: sum |old - new| over the touched rows
        stored = dev::ld_stream(M.P + e);
        viol += fabs(stored.x - p.x) + fabs(stored.y - p.y);
        dev::st_stream(M.P + e, p);
      }
    }
  }
  double2 acc = {0.0, 0.0}, accn = {0.0, 0.0};
  double seta = 0.0;
  if (MODE == 2) {
    // a whole wavefront works on one heavy feature: lane group g adds segments g, g + R, ... in
    // order, then the groups' sums are combined by a fixed xor-shuffle tree; group 0 applies
    constexpr int RG = kWave / LG;
    const int g_ = (int)(threadIdx.x & (kWave - 1)) / LG;
#pragma unroll 4
    for (int64_t sg = g_; sg < nseg; sg += RG) {
      const double* rec_ = hp + (size_t)sg * (seg_stride ? seg_stride : PW);
      const double2 pa = *reinterpret_cast<const double2*>(rec_);
      const double2 pn = *reinterpret_cast<const double2*>(rec_ + M.Kp);
      acc.x += pa.x;
      acc.y += pa.y;
      accn.x += pn.x;
      accn.y += pn.y;
      const double* sc_ = rec_ - 2 * l + 2 * M.Kp;
      seta += sc_[0];
      if (do_w) {
        wacc.a0 += sc_[1];
        wacc.a1 += sc_[2];
      }
    }
#pragma unroll
    for (int sh = LG; sh < kWave; sh <<= 1) {
      acc.x += dev::shfl_xor_d(acc.x, sh);
      acc.y += dev::shfl_xor_d(acc.y, sh);
      accn.x += dev::shfl_xor_d(accn.x, sh);
      accn.y += dev::shfl_xor_d(accn.y, sh);
      seta += dev::shfl_xor_d(seta, sh);
      wacc.a0 += dev::shfl_xor_d(wacc.a0, sh);
      wacc.a1 += dev::shfl_xor_d(wacc.a1, sh);
    }
    if (g_ != 0) return 0.0;  // one group applies (AdaGrad's viol / stored-P side effects above are idempotent)
  } else if (!GEN || deg == 2) {
    // The touches' (sample, value) pairs are fetched L at a time, one touch per lane of the feature's
    // lane group (one coalesced load instead of L same-address loads and one dependent round trip
    // instead of one per TU touches), and handed round with ds_bpermute; the records and A rows of
    // TU touches are then requested together.  The accumulation stays in touch (= sample) order.
    const int gbase = (int)(threadIdx.x & (kWave - 1)) - l;  // first lane of this feature's group
    for (int64_t tb = t0; tb < t1; tb += LG) {
      const int64_t tl = tb + l;
      const int pib_l = tl < t1 ? a.tpos[tl] : 0;
      const double x_l = tl < t1 ? a.tx[tl] : 0.0;
      const int cnt = (int)(t1 - tb < LG ? t1 - tb : LG);
      for (int ub = 0; ub < cnt; ub += TU) {
        int pib[TU];
        double x[TU];
        SampleRec r[TU];
        double2 A1[TU];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const int src = gbase + ((ub + u) < LG ? (ub + u) : 0);
          pib[u] = __shfl(pib_l, src, kWave);
          x[u] = dev::shfl_d(x_l, src);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          r[u] = a.rec[pib[u]];
          A1[u] = *reinterpret_cast<const double2*>(a.Abuf + ((size_t)pib[u] * a.TA + slot) * M.Kp + 2 * l);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          if (ub + u < cnt) {
            const double dAx = x[u] * (A1[u].x - p.x * x[u]);
            const double dAy = x[u] * (A1[u].y - p.y * x[u]);
            if (OPT == OPT_SGD) {  // sgd.nim:220-222, averaged per coordinate below
              acc.x += r[u].etaP * (r[u].dL * dAx);
              acc.y += r[u].etaP * (r[u].dL * dAy);
              seta += r[u].etaP;
              if (do_w) {
                wacc.a0 += r[u].etaw * (r[u].dL * x[u]);
                wacc.a1 += r[u].etaw;
              }
            } else if (OPT == OPT_PSGD) {  // minibatch_psgd.nim:75-84: coef = dloss / miniBatchSize
              const double cf = r[u].dL / O.bsize;
              acc.x += cf * dAx;
              acc.y += cf * dAy;
              if (do_w) wacc.a0 += cf * x[u];
            } else {  // adagrad.nim:122-124
              const double gx = r[u].dL * dAx, gy = r[u].dL * dAy;
              acc.x += gx;
              acc.y += gy;
              accn.x += gx * gx;
              accn.y += gy * gy;
              if (do_w) {
                const double gw = r[u].dL * x[u];
                wacc.a0 += gw;
                wacc.a1 += gw * gw;
              }
            }
          }
        }
      }
    }
  } else {
    // degree >= 3: the same walk (touches fetched LG at a time by the feature's lanes, records and A rows
    // of two touches requested together); a touch needs the sample's deg - 1 A rows of this order
    constexpr int TG = 2;
    const int gbase = (int)(threadIdx.x & (kWave - 1)) - l;
    for (int64_t tb = t0; tb < t1; tb += LG) {
      const int64_t tl = tb + l;
      const int pib_l = tl < t1 ? a.tpos[tl] : 0;
      const double x_l = tl < t1 ? a.tx[tl] : 0.0;
      const int cnt = (int)(t1 - tb < LG ? t1 - tb : LG);
      for (int ub = 0; ub < cnt; ub += TG) {
        int pib[TG];
        double x[TG];
        SampleRec r[TG];
        double2 Av[TG][dev::kMaxDeg - 1];
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int src = gbase + ((ub + u) < LG ? (ub + u) : 0);
          pib[u] = __shfl(pib_l, src, kWave);
          x[u] = dev::shfl_d(x_l, src);
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          r[u] = a.rec[pib[u]];
          const double* Ar = a.Abuf + ((size_t)pib[u] * a.TA + slot) * M.Kp + 2 * l;
#pragma unroll
          for (int tt = 0; tt < dev::kMaxDeg - 1; ++tt) {
            Av[u][tt] = {0.0, 0.0};
            if (tt < deg - 1) Av[u][tt] = *reinterpret_cast<const double2*>(Ar + (size_t)tt * M.Kp);
          }
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          if (ub + u >= cnt) continue;
          double Ax[dev::kMaxDeg - 1], Ay[dev::kMaxDeg - 1];
#pragma unroll
          for (int tt = 0; tt < dev::kMaxDeg - 1; ++tt) {
            Ax[tt] = Av[u][tt].x;
            Ay[tt] = Av[u][tt].y;
          }
          const double dAx = dev::anova_grad(deg, x[u], p.x, Ax);
          const double dAy = dev::anova_grad(deg, x[u], p.y, Ay);
          if (OPT == OPT_SGD) {
            acc.x += r[u].etaP * (r[u].dL * dAx);
            acc.y += r[u].etaP * (r[u].dL * dAy);
            seta += r[u].etaP;
            if (do_w) {
              wacc.a0 += r[u].etaw * (r[u].dL * x[u]);
              wacc.a1 += r[u].etaw;
            }
          } else if (OPT == OPT_PSGD) {
            const double cf = r[u].dL / O.bsize;
            acc.x += cf * dAx;
            acc.y += cf * dAy;
            if (do_w) wacc.a0 += cf * x[u];
          } else {
            const double gx = r[u].dL * dAx, gy = r[u].dL * dAy;
            acc.x += gx;
            acc.y += gy;
            accn.x += gx * gx;
            accn.y += gy * gy;
            if (do_w) {
              const double gw = r[u].dL * x[u];
              wacc.a0 += gw;
              wacc.a1 += gw * gw;
            }
          }
        }
      }
    }
  }
  if (MODE == 1) {
    *reinterpret_cast<double2*>(hp) = acc;
    *reinterpret_cast<double2*>(hp + M.Kp) = accn;
    if (l == 0) {
      double* sc_ = hp + 2 * M.Kp;
      sc_[0] = seta;
      sc_[1] = wacc.a0;
      sc_[2] = wacc.a1;
      sc_[3] = 0.0;
    }
    return 0.0;
  }
  if (OPT == OPT_SGD) {
    const double c = dev::touch_div(MODE == 0 ? (double)(t1 - t0) : c_total, O.touch_cap);
    viol += fabs((acc.x + seta * O.beta * p.x) / c) + fabs((acc.y + seta * O.beta * p.y) / c);
    stored.x = stored.x * fP - (acc.x / c) / sPn;
    stored.y = stored.y * fP - (acc.y / c) / sPn;
    dev::st_stream(M.P + e, stored);
  } else if (OPT == OPT_PSGD) {  // Params.add with -eta_P (model/params.nim:33-41,97); sPn carries eta_P here
    if (O.gradP != nullptr) {  // predictAllWithGrad: the gradient is the product
      *reinterpret_cast<double2*>(O.gradP + e) = acc;
    } else {
      stored.x += -sPn * acc.x;
      stored.y += -sPn * acc.y;
      dev::st_stream(M.P + e, stored);
    }
  } else {
    g2.x += acc.x;
    g2.y += acc.y;
    n2.x += dev::ada_norm_inc(acc.x, accn.x, O.ada_cross);
    n2.y += dev::ada_norm_inc(acc.y, accn.y, O.ada_cross);
    dev::st_stream(O.G + e, g2);
    dev::st_stream(O.N + e, n2);
  }
  return viol;
}

// degree 3 with fitLower = explicit (cfg5): two parameter blocks -- order 0 of ANOVA degree 3 (A rows 0, 1 of the sample:
// A[1], A[2]) and order 1 of degree 2 (A row 2).  col_block walks a feature's touch list once per block (touch positions,
// values and the samples' records fetched twice); here BOTH blocks are handled in one walk: one record and the sample's
// three A rows per touch.  Same arithmetic and the same order of every sum as two col_block<.., MODE 0> calls.
template <int OPT, int LG>
__device__ __forceinline__ double col_block_d3(const ColArgs& a, int64_t j, int l, int64_t t0, int64_t t1, double sP,
                                               double sPn, double fP, bool do_w, WAcc& wacc) {
  static_assert(OPT == OPT_SGD || OPT == OPT_ADAGRAD, "SGD / AdaGrad");
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const size_t e[2] = {M.row(0, j) * M.Kp + 2 * l, M.row(1, j) * M.Kp + 2 * l};
  double viol = 0.0;
  double2 stored[2], g2[2], n2[2], p[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    stored[o] = g2[o] = n2[o] = {0.0, 0.0};
    if (OPT == OPT_SGD) {
      stored[o] = dev::ld_stream(M.P + e[o]);
      p[o].x = sP * stored[o].x;
      p[o].y = sP * stored[o].y;
    } else {
      g2[o] = dev::ld_stream(O.G + e[o]);
      n2[o] = dev::ld_stream(O.N + e[o]);
      if (a.use_stored) {
        p[o] = dev::ld_stream(M.P + e[o]);
      } else {
        const double tmp = O.eta0 * ((a.it0p[0] + a.it_b) - 1.0) * O.beta;
        p[o].x = dev::adagrad_param(g2[o].x, n2[o].x, O.eta0, tmp);
        p[o].y = dev::adagrad_param(g2[o].y, n2[o].y, O.eta0, tmp);
        if (O.track_viol) {
          stored[o] = dev::ld_stream(M.P + e[o]);
          viol += fabs(stored[o].x - p[o].x) + fabs(stored[o].y - p[o].y);
          dev::st_stream(M.P + e[o], p[o]);
        }
      }
    }
  }
  double2 acc[2] = {{0.0, 0.0}, {0.0, 0.0}}, accn[2] = {{0.0, 0.0}, {0.0, 0.0}};
  double seta = 0.0;
  constexpr int TG = 2;
  const int gbase = (int)(threadIdx.x & (kWave - 1)) - l;
  for (int64_t tb = t0; tb < t1; tb += LG) {
    const int64_t tl = tb + l;
    const int pib_l = tl < t1 ? a.tpos[tl] : 0;
    const double x_l = tl < t1 ? a.tx[tl] : 0.0;
    const int cnt = (int)(t1 - tb < LG ? t1 - tb : LG);
    for (int ub = 0; ub < cnt; ub += TG) {
      int pib[TG];
      double x[TG];
      SampleRec r[TG];
      double2 Aa[TG], Ab[TG], Ac[TG];
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        const int src = gbase + ((ub + u) < LG ? (ub + u) : 0);
        pib[u] = __shfl(pib_l, src, kWave);
        x[u] = dev::shfl_d(x_l, src);
      }
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        r[u] = a.rec[pib[u]];
        const double* Ar = a.Abuf + (size_t)pib[u] * 3 * M.Kp + 2 * l;
        Aa[u] = *reinterpret_cast<const double2*>(Ar);
        Ab[u] = *reinterpret_cast<const double2*>(Ar + M.Kp);
        Ac[u] = *reinterpret_cast<const double2*>(Ar + 2 * M.Kp);
      }
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        if (ub + u >= cnt) continue;
        const double xv = x[u];
        // order 0, degree 3 (optimizer/sgd.nim:176-188): dA = x; dA = x (A[1] - p dA); dA = x (A[2] - p dA)
        double d0x = xv * (Aa[u].x - p[0].x * xv), d0y = xv * (Aa[u].y - p[0].y * xv);
        d0x = xv * (Ab[u].x - p[0].x * d0x);
        d0y = xv * (Ab[u].y - p[0].y * d0y);
        // order 1, degree 2
        const double d1x = xv * (Ac[u].x - p[1].x * xv), d1y = xv * (Ac[u].y - p[1].y * xv);
        if (OPT == OPT_SGD) {
          acc[0].x += r[u].etaP * (r[u].dL * d0x);
          acc[0].y += r[u].etaP * (r[u].dL * d0y);
          acc[1].x += r[u].etaP * (r[u].dL * d1x);
          acc[1].y += r[u].etaP * (r[u].dL * d1y);
          seta += r[u].etaP;
          if (do_w) {
            wacc.a0 += r[u].etaw * (r[u].dL * xv);
            wacc.a1 += r[u].etaw;
          }
        } else {
          const double g0x = r[u].dL * d0x, g0y = r[u].dL * d0y, g1x = r[u].dL * d1x, g1y = r[u].dL * d1y;
          acc[0].x += g0x;
          acc[0].y += g0y;
          accn[0].x += g0x * g0x;
          accn[0].y += g0y * g0y;
          acc[1].x += g1x;
          acc[1].y += g1y;
          accn[1].x += g1x * g1x;
          accn[1].y += g1y * g1y;
          if (do_w) {
            const double gw = r[u].dL * xv;
            wacc.a0 += gw;
            wacc.a1 += gw * gw;
          }
        }
      }
    }
  }
  const double c = dev::touch_div((double)(t1 - t0), O.touch_cap);
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    if (OPT == OPT_SGD) {
      viol += fabs((acc[o].x + seta * O.beta * p[o].x) / c) + fabs((acc[o].y + seta * O.beta * p[o].y) / c);
      stored[o].x = stored[o].x * fP - (acc[o].x / c) / sPn;
      stored[o].y = stored[o].y * fP - (acc[o].y / c) / sPn;
      dev::st_stream(M.P + e[o], stored[o]);
    } else {
      g2[o].x += acc[o].x;
      g2[o].y += acc[o].y;
      n2[o].x += dev::ada_norm_inc(acc[o].x, accn[o].x, O.ada_cross);
      n2[o].y += dev::ada_norm_inc(acc[o].y, accn[o].y, O.ada_cross);
      dev::st_stream(O.G + e[o], g2[o]);
      dev::st_stream(O.N + e[o], n2[o]);
    }
  }
  return viol;
}

// degree 2 with 129 ... 256 factors (ModelView::kc == 2): two parameter blocks of degree 2 (A rows 0 and 1 of the sample).  As
// col_block_w2: BOTH blocks in one walk of the feature's touch list -- one record and the sample's two A rows per touch -- with
// the arithmetic and the order of every sum of two col_block<.., MODE 0> calls (bench.py --workload wide256: column phase 2.10 -> 1.68 ms per batch of 32768).
template <int OPT, int LG>
__device__ __forceinline__ double col_block_w2(const ColArgs& a, int64_t j, int l, int64_t t0, int64_t t1, double sP,
                                               double sPn, double fP, bool do_w, WAcc& wacc) {
  static_assert(OPT == OPT_SGD || OPT == OPT_ADAGRAD, "SGD / AdaGrad");
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const size_t e[2] = {M.row(0, j) * M.Kp + 2 * l, M.row(1, j) * M.Kp + 2 * l};
  double viol = 0.0;
  double2 stored[2], g2[2], n2[2], p[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    stored[o] = g2[o] = n2[o] = {0.0, 0.0};
    if (OPT == OPT_SGD) {
      stored[o] = dev::ld_stream(M.P + e[o]);
      p[o].x = sP * stored[o].x;
      p[o].y = sP * stored[o].y;
    } else {
      g2[o] = dev::ld_stream(O.G + e[o]);
      n2[o] = dev::ld_stream(O.N + e[o]);
      if (a.use_stored) {
        p[o] = dev::ld_stream(M.P + e[o]);
      } else {
        const double tmp = O.eta0 * ((a.it0p[0] + a.it_b) - 1.0) * O.beta;
        p[o].x = dev::adagrad_param(g2[o].x, n2[o].x, O.eta0, tmp);
        p[o].y = dev::adagrad_param(g2[o].y, n2[o].y, O.eta0, tmp);
        if (O.track_viol) {
          stored[o] = dev::ld_stream(M.P + e[o]);
          viol += fabs(stored[o].x - p[o].x) + fabs(stored[o].y - p[o].y);
          dev::st_stream(M.P + e[o], p[o]);
        }
      }
    }
  }
  double2 acc[2] = {{0.0, 0.0}, {0.0, 0.0}}, accn[2] = {{0.0, 0.0}, {0.0, 0.0}};
  double seta = 0.0;
  constexpr int TG = 2;
  const int gbase = (int)(threadIdx.x & (kWave - 1)) - l;
  for (int64_t tb = t0; tb < t1; tb += LG) {
    const int64_t tl = tb + l;
    const int pib_l = tl < t1 ? a.tpos[tl] : 0;
    const double x_l = tl < t1 ? a.tx[tl] : 0.0;
    const int cnt = (int)(t1 - tb < LG ? t1 - tb : LG);
    for (int ub = 0; ub < cnt; ub += TG) {
      int pib[TG];
      double x[TG];
      SampleRec r[TG];
      double2 Aa[TG], Ab[TG];
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        const int src = gbase + ((ub + u) < LG ? (ub + u) : 0);
        pib[u] = __shfl(pib_l, src, kWave);
        x[u] = dev::shfl_d(x_l, src);
      }
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        r[u] = a.rec[pib[u]];
        const double* Ar = a.Abuf + (size_t)pib[u] * 2 * M.Kp + 2 * l;
        Aa[u] = *reinterpret_cast<const double2*>(Ar);
        Ab[u] = *reinterpret_cast<const double2*>(Ar + M.Kp);
      }
#pragma unroll
      for (int u = 0; u < TG; ++u) {
        if (ub + u >= cnt) continue;
        const double xv = x[u];
        // degree 2 (optimizer/sgd.nim:187-188): dA = x (A[1] - p x), the factors of block 0 and of block 1
        const double d0x = xv * (Aa[u].x - p[0].x * xv), d0y = xv * (Aa[u].y - p[0].y * xv);
        const double d1x = xv * (Ab[u].x - p[1].x * xv), d1y = xv * (Ab[u].y - p[1].y * xv);
        if (OPT == OPT_SGD) {
          acc[0].x += r[u].etaP * (r[u].dL * d0x);
          acc[0].y += r[u].etaP * (r[u].dL * d0y);
          acc[1].x += r[u].etaP * (r[u].dL * d1x);
          acc[1].y += r[u].etaP * (r[u].dL * d1y);
          seta += r[u].etaP;
          if (do_w) {
            wacc.a0 += r[u].etaw * (r[u].dL * xv);
            wacc.a1 += r[u].etaw;
          }
        } else {
          const double g0x = r[u].dL * d0x, g0y = r[u].dL * d0y, g1x = r[u].dL * d1x, g1y = r[u].dL * d1y;
          acc[0].x += g0x;
          acc[0].y += g0y;
          accn[0].x += g0x * g0x;
          accn[0].y += g0y * g0y;
          acc[1].x += g1x;
          acc[1].y += g1y;
          accn[1].x += g1x * g1x;
          accn[1].y += g1y * g1y;
          if (do_w) {
            const double gw = r[u].dL * xv;
            wacc.a0 += gw;
            wacc.a1 += gw * gw;
          }
        }
      }
    }
  }
  const double c = dev::touch_div((double)(t1 - t0), O.touch_cap);
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    if (OPT == OPT_SGD) {
      viol += fabs((acc[o].x + seta * O.beta * p[o].x) / c) + fabs((acc[o].y + seta * O.beta * p[o].y) / c);
      stored[o].x = stored[o].x * fP - (acc[o].x / c) / sPn;
      stored[o].y = stored[o].y * fP - (acc[o].y / c) / sPn;
      dev::st_stream(M.P + e[o], stored[o]);
    } else {
      g2[o].x += acc[o].x;
      g2[o].y += acc[o].y;
      n2[o].x += dev::ada_norm_inc(acc[o].x, accn[o].x, O.ada_cross);
      n2[o].y += dev::ada_norm_inc(acc[o].y, accn[o].y, O.ada_cross);
      dev::st_stream(O.G + e[o], g2[o]);
      dev::st_stream(O.N + e[o], n2[o]);
    }
  }
  return viol;
}

// linear term of one feature (fit_linear.nim:41-57); every lane of the feature holds the same sums
template <int OPT>
__device__ __forceinline__ double w_epilogue(const ColArgs& a, int64_t j, int l, double c, double sw, double swn, double fw,
                                             const WAcc& wacc) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  double viol = 0.0;
  const double wt = M.w[j];
  if (OPT == OPT_SGD) {
    const double wj = sw * wt;
    const double cd = dev::touch_div(c, O.touch_cap);
    if (l == 0) {
      viol += fabs((wacc.a0 + wacc.a1 * O.alpha * wj) / cd);
      M.w[j] = wt * fw - (wacc.a0 / cd) / swn;
    }
  } else if (OPT == OPT_PSGD) {  // model/params.nim:43-45; swn carries eta_w
    if (l == 0) {
      if (O.gradw != nullptr) O.gradw[j] = wacc.a0;
      else M.w[j] = wt + -swn * wacc.a0;
    }
  } else {
    const double gw = O.Gw[j], nw = O.Nw[j];
    if (l == 0) {
      if (!a.use_stored) {
        const double wj = -O.eta0 * gw / (((a.it0p[0] + a.it_b) - 1.0) * O.eta0 * O.alpha + sqrt(nw));
        viol += fabs(wt - wj);
        M.w[j] = wj;
      }
      O.Gw[j] = gw + wacc.a0;
      O.Nw[j] = nw + dev::ada_norm_inc(wacc.a0, wacc.a1, O.ada_cross);
    }
  }
  return viol;
}

// decay corrections of a coordinate touched c times (schedule kernel's table, pow beyond it)
__device__ __forceinline__ void touch_factors(const ColArgs& a, int64_t c, double& fP, double& fw) {
  fP = 1.0;
  fw = 1.0;
  if (c > 1) {
    if (c <= kFtab) {  // (the table has the touch cap in it: k_schedule)
      fP = a.Ftab_b[c - 1];
      fw = a.Ftab_b[kFtab + c - 1];
    } else {
      const double cd = dev::touch_div((double)c, a.O.touch_cap);
      fP = cd == 1.0 ? 1.0 : pow(a.Dtab_b[0], 1.0 / cd) / a.Dtab_b[0];
      fw = cd == 1.0 ? 1.0 : pow(a.Dtab_b[1], 1.0 / cd) / a.Dtab_b[1];
    }
  }
}

// ---- heavy features (degree-2 models): segment partial sums, then per-feature apply ----
struct HeavyArgs {
  const int64_t* hv_u;     // heavy feature -> index into ucol / uptr
  const int64_t* hv_seg0;  // heavy feature -> its first segment
  int64_t h0, h1, s0, s1;  // this batch's heavy features / segments
  double* hpart;           // [s1 - s0][PW]
  double* parts;           // per-block viol partials of the apply kernel
  int32_t PW, pad_;
};

template <int L, int OPT, bool GEN>
__global__ __launch_bounds__(kBlock) void k_heavy_partial(ColArgs a, HeavyArgs hv) {
  constexpr int R = kWave / L;
  const ModelView& M = a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t gs = hv.s0 + ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
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
  const double sP = OPT == OPT_SGD ? a.scales_b[0] : 1.0;
  const bool has_w = M.fit_linear && j < M.d;
  int slot = 0;
  for (int o = 0; o < M.nb; ++o) {  // one partial record per (segment, order)
    WAcc wacc;
    const int deg = M.deg_of(o);
    col_block<OPT, GEN, 2, L, 1>(a, M.row(o, j) * M.Kp + 2 * l, deg, slot, l, t0, t1, sP, 1.0, 1.0, has_w && o == 0, wacc,
                                 0.0, hv.hpart + ((size_t)(gs - hv.s0) * M.nb + o) * hv.PW + 2 * l, 0, hv.PW);
    slot += deg - 1;
  }
}

template <int L, int OPT, bool GEN>
__global__ __launch_bounds__(kBlock) void k_heavy_apply(ColArgs a, HeavyArgs hv) {
  constexpr int R = kWave / L;
  __shared__ double red[kWavesPerBlock];
  const ModelView& M = a.M;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int64_t h = hv.h0 + (int64_t)blockIdx.x * kWavesPerBlock + wv;  // one wavefront per heavy feature
  (void)R;
  double viol = 0.0;
  if (h < hv.h1) {
    const int64_t u = hv.hv_u[h];
    const int64_t j = a.ucol[u];
    const int64_t c = a.uptr[u + 1] - a.uptr[u];
    double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0, fP = 1.0, fw = 1.0;
    if (OPT == OPT_SGD) {
      sP = a.scales_b[0];
      sw = a.scales_b[1];
      sPn = a.scales_n[0];
      swn = a.scales_n[1];
      touch_factors(a, c, fP, fw);
    } else if (OPT == OPT_PSGD) {  // this mini-batch's step sizes (minibatch_psgd.nim:112-113) ride in sPn / swn
      const OptView& O = a.O;
      const double it = a.it0p[0] + a.it_b;
      sPn = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
      swn = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
    }
    const bool has_w = M.fit_linear && j < M.d;
    WAcc wacc;
    const int64_t sg0 = hv.hv_seg0[h], nseg = hv.hv_seg0[h + 1] - sg0;
    int slot = 0;
    for (int o = 0; o < M.nb; ++o) {
      WAcc wo;
      const int deg = M.deg_of(o);
      viol += col_block<OPT, GEN, 2, L, 2>(a, M.row(o, j) * M.Kp + 2 * l, deg, slot, l, 0, 0, sP, sPn, fP, has_w && o == 0,
                                           wo, (double)c, hv.hpart + ((size_t)(sg0 - hv.s0) * M.nb + o) * hv.PW + 2 * l, nseg,
                                           hv.PW, M.nb * hv.PW);
      if (o == 0) wacc = wo;
      slot += deg - 1;
    }
    if (has_w && g == 0) viol += w_epilogue<OPT>(a, j, l, (double)c, sw, swn, fw, wacc);
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

// singles kernel: one wavefront per sample of the batch updates, in place, the features only that
// sample touches (sparse regime).  Runs between the row and the column phase; disjoint rows.
template <int L, int OPT>
__global__ __launch_bounds__(kBlock) void k_singles(ColArgs a) {
  constexpr int R = kWave / L;
  __shared__ double red[kWavesPerBlock];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int pib = blockIdx.x * kWavesPerBlock + wv;
  double viol = 0.0;
  if (pib < a.len_i) {
    const CsrView& X = a.X;
    const int64_t pos = a.p0 + pib;
    const int64_t i = a.perm ? a.perm[pos] : a.begin + pos;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int m_tot = m + M.n_aug;
    const SampleRec r = a.rec[pib];
    const double2 A1 = *reinterpret_cast<const double2*>(a.Abuf + (size_t)pib * a.TA * M.Kp + 2 * l);
    viol = singles_update<L, R, OPT>(X, M, O, a.single + a.toff[pos], a.scales_b, a.scales_n, q0, m, m_tot, g, l, r.dL,
                                     r.etaP, r.etaw, A1, (a.it0p[0] + a.it_b) - 1.0, a.use_stored != 0);
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

// the closing workgroup of a column-phase launch: fixed-order reductions of the row phase's and the previous batch's
// partials, the intercept's update; it runs beside the feature workgroups
template <int OPT>
__device__ __forceinline__ void col_closer(const ColArgs& a, double (&red)[5][kBlock]) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  __syncthreads();
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < a.nA; i += kBlock) {
    const PartA p = a.partsA[i];
    s[0] += p.loss;
    s[1] += p.viol;
    s[2] += p.acc0;
    s[3] += p.acc1;
  }
  for (int i = threadIdx.x; i < a.n_prev; i += kBlock) s[4] += a.parts_prev[i];
  for (int c = 0; c < 5; ++c) red[c][threadIdx.x] = s[c];
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int c = 0; c < 5; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = red[1][0] + red[4][0];
    if (M.fit_intercept) {
      if (OPT == OPT_SGD) {  // the intercept is touched by every sample of the batch: c = len
        const double b0 = M.sc[SC_INTERCEPT], f0 = a.Dtab_b[3], lc = dev::touch_div(a.len, O.touch_cap);
        v += fabs((red[2][0] + red[3][0] * O.alpha0 * b0) / lc);
        M.sc[SC_INTERCEPT] = f0 * b0 - red[2][0] / lc;
      } else if (OPT == OPT_PSGD) {
        // model/params.nim:47 gates the intercept's step on grad.fitLinear -- kept as the reference has it
        if (O.gradb != nullptr)
          O.gradb[0] = red[2][0] / O.bsize;
        else if (M.fit_linear)
          M.sc[SC_INTERCEPT] += -dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, a.it0p[0] + a.it_b) * (red[2][0] / O.bsize);
      } else {
        if (!a.use_stored) {  // adagrad.nim:102-106
          const double old = M.sc[SC_INTERCEPT];
          const double nb_ = -O.eta0 * O.gsc[0] / (sqrt(O.gsc[1]) + O.eta0 * ((a.it0p[0] + a.it_b) - 1.0) * O.alpha0);
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

// STRIDED: the feature workgroups loop over the features (capped grid); otherwise one lane group = one
// feature and no loop -- the loop costs registers (94 vs 80: 5 instead of 6 wavefronts per SIMD), which
// the dense regime (cfg2) pays for without needing it
template <int L, int OPT, bool GEN, int TU, bool STRIDED>
__global__ __launch_bounds__(kBlock, NFM_COL_MINW) void k_col_phase(ColArgs a) {
  constexpr int R = kWave / L;
  __shared__ double red[5][kBlock];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  double viol = 0.0;
  const int fb = blockIdx.x;                        // feature workgroup index
  const bool closer = blockIdx.x == gridDim.x - 1;  // the extra, last workgroup only closes the batch
  // the feature workgroups stride over the batch's unique features: the launch holds at most one
  // resident set of wavefronts (host: run_batches), so no wavefront waits for a slot and the
  // per-workgroup launch cost is paid once per ~8 features instead of once per feature
  const int64_t stride = (int64_t)(gridDim.x - 1) * kWavesPerBlock * R;
  for (int64_t u = closer ? a.u1 : a.u0 + ((int64_t)fb * kWavesPerBlock + wv) * R + g; u < a.u1;
       u = STRIDED ? u + stride : a.u1) {
    // heavy features are summed by k_heavy_partial / k_heavy_apply (degree-1 models have no parameter block
    // to walk them with and keep them here)
    const int64_t cnt_u = a.ucnt_s[u];
    if (M.nb > 0 && cnt_u > kHeavyTouches) continue;
    const int64_t j = a.ucol_s[u];
    const int64_t t0 = a.ubeg_s[u], t1 = t0 + cnt_u;
    double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0, fP = 1.0, fw = 1.0;
    if (OPT == OPT_SGD) {
      sP = a.scales_b[0];
      sw = a.scales_b[1];
      sPn = a.scales_n[0];
      swn = a.scales_n[1];
      touch_factors(a, t1 - t0, fP, fw);
    } else if (OPT == OPT_PSGD) {
      const double it = a.it0p[0] + a.it_b;
      sPn = dev::get_eta(O.sched, O.eta0, O.power, O.beta, it);
      swn = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it);
    }
    const bool has_w = M.fit_linear && j < M.d;  // dummy features have no w
    WAcc wacc;
    int slot = 0;
    if (GEN && OPT != OPT_PSGD && NFM_COL_D3 && M.nb == 2 && M.kc == 1 && M.degree == 3 && a.TA == 3) {
      // degree 3, explicit lower orders (cfg5): both parameter blocks in one walk of the touch list
      viol += col_block_d3<(OPT == OPT_PSGD ? OPT_SGD : OPT), L>(a, j, l, t0, t1, sP, sPn, fP, has_w, wacc);
    } else if (GEN && OPT != OPT_PSGD && NFM_COL_D3 && M.nb == 2 && M.kc == 2 && M.degree == 2 && a.TA == 2) {
      // 129 ... 256 factors: the two blocks of the one order in one walk
      viol += col_block_w2<(OPT == OPT_PSGD ? OPT_SGD : OPT), L>(a, j, l, t0, t1, sP, sPn, fP, has_w, wacc);
    } else
    for (int o = 0; o < M.nb; ++o) {
      const size_t e = M.row(o, j) * M.Kp + 2 * l;
      const int deg = M.deg_of(o);
      viol += col_block<OPT, GEN, TU, L, 0>(a, e, deg, slot, l, t0, t1, sP, sPn, fP, has_w && o == 0, wacc);
      slot += deg - 1;
    }
    if (has_w) {
      if (M.nb == 0) {  // degree-1 model: no parameter block walked the touches
        for (int64_t t = t0; t < t1; ++t) {
          const SampleRec r = a.rec[a.tpos[t]];
          const double x = a.tx[t];
          if (OPT == OPT_SGD) {
            wacc.a0 += r.etaw * (r.dL * x);
            wacc.a1 += r.etaw;
          } else if (OPT == OPT_PSGD) {
            wacc.a0 += (r.dL / O.bsize) * x;
          } else {
            wacc.a0 += r.dL * x;
            wacc.a1 += (r.dL * x) * (r.dL * x);
          }
        }
      }
      viol += w_epilogue<OPT>(a, j, l, (double)(t1 - t0), sw, swn, fw, wacc);
    }
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[0][wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[0][w_];
    a.parts[a.nS + blockIdx.x] = v;  // the singles kernel owns parts[0, nS)
  }
  if (!closer) return;
  col_closer<OPT>(a, red);
}

// ------------------------------------------------------------------------------------------------
// column phase, sparse regime (one order of degree 2, SGD / AdaGrad): software-pipelined strided walk
// ------------------------------------------------------------------------------------------------
// In the sparse regime a batch has ~1e5 multi-touch features with two or three touches each: a feature costs three
// DEPENDENT round trips (its header; its parameter row and touch list; the touching samples' records and A rows) and
// almost no arithmetic, and the register count caps a CU at 16-20 wavefronts -- k_col_phase<.., STRIDED> is latency
// bound (headline shape: 9 rounds of ~6 us).  Here a lane group keeps THREE features in flight: while the records and
// A rows of feature u are gathered, the row, the touches, the decay factors and the linear weight of feature
// u + stride are already requested, and so is the header of feature u + 2 stride.  Same arithmetic, same order of
// every sum as col_block<.., MODE 0>.
template <int OPT>
struct ColStage {  // what the second round trip of a feature delivers
  double2 st, g2, n2;  // SGD: stored row; AdaGrad: g_sum, g_norm (+ st = stored parameters when viol is tracked)
  double x, fP, fw, wt, gw, nw;
  int pib;
};
struct ColHdr {
  int cnt, j;
  int64_t t0;
};

#ifndef NFM_COL_NF
#define NFM_COL_NF 1
#endif
#ifndef NFM_COL_SPARSE_MINW
#define NFM_COL_SPARSE_MINW 1
#endif
template <int L, int OPT>
__global__ __launch_bounds__(kBlock, NFM_COL_SPARSE_MINW) void k_col_sparse(ColArgs a) {
  static_assert(OPT == OPT_SGD || OPT == OPT_ADAGRAD, "SGD / AdaGrad");
  constexpr int R = kWave / L, TU = 2;
  constexpr int NF = NFM_COL_NF;  // features a lane group gathers for at the same time
  __shared__ double red[5][kBlock];
  const ModelView& M = a.M;
  const OptView& O = a.O;
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
  const int g = lane / L, l = lane % L;
  const int gbase = lane - l;
  const bool closer = blockIdx.x == gridDim.x - 1;
  const int64_t stride = (int64_t)(gridDim.x - 1) * kWavesPerBlock * R;
  double viol = 0.0;
  double sP = 1.0, sPn = 1.0, sw = 1.0, swn = 1.0;
  if (OPT == OPT_SGD) {
    sP = a.scales_b[0];
    sw = a.scales_b[1];
    sPn = a.scales_n[0];
    swn = a.scales_n[1];
  }
  const double itp = (a.it0p[0] + a.it_b) - 1.0;
  const double tmpP = O.eta0 * itp * O.beta;
  auto load_hdr = [&](int64_t uu) {
    ColHdr h{0, 0, 0};
    if (uu < a.u1) {
      h.cnt = a.ucnt_s[uu];
      h.j = a.ucol_s[uu];
      h.t0 = a.ubeg_s[uu];
      if (h.cnt > kHeavyTouches) h.cnt = 0;  // heavy features: k_heavy_partial / k_heavy_apply
    }
    return h;
  };
  auto load_stage = [&](const ColHdr& h) {
    ColStage<OPT> s{};
    if (h.cnt > 0) {
      const size_t e = M.row(0, h.j) * M.Kp + 2 * l;
      if (OPT == OPT_SGD) {
        s.st = dev::ld_stream(M.P + e);
        touch_factors(a, h.cnt, s.fP, s.fw);
      } else {
        s.g2 = dev::ld_stream(O.G + e);
        s.n2 = dev::ld_stream(O.N + e);
        if (a.use_stored || O.track_viol) s.st = dev::ld_stream(M.P + e);
      }
      if (l < h.cnt) {
        s.pib = a.tpos[h.t0 + l];
        s.x = a.tx[h.t0 + l];
      }
      if (M.fit_linear && h.j < M.d) {
        s.wt = M.w[h.j];
        if (OPT == OPT_ADAGRAD) {
          s.gw = O.Gw[h.j];
          s.nw = O.Nw[h.j];
        }
      }
    }
    return s;
  };
  struct Gath {  // records and A rows of a feature's first TU touches
    SampleRec r[TU];
    double2 A1[TU];
    double x[TU];
  };
  auto gather = [&](const ColHdr& h, const ColStage<OPT>& s) {
    Gath G{};
    if (h.cnt > 0) {
#pragma unroll
      for (int q = 0; q < TU; ++q) {
        const int src = gbase + (q < h.cnt && q < L ? q : 0);
        const int pib = __shfl(s.pib, src, kWave);
        G.x[q] = dev::shfl_d(s.x, src);
        G.r[q] = a.rec[pib];
        G.A1[q] = *reinterpret_cast<const double2*>(a.Abuf + (size_t)pib * a.TA * M.Kp + 2 * l);
      }
    }
    return G;
  };
  // one touch into the feature's sums, in touch order: the arithmetic of col_block<.., MODE 0>
  auto add_touch = [&](const SampleRec& r, double2 A1, double x, double2 p, bool do_w, double2& acc, double2& accn, double& seta,
                       WAcc& wacc) {
    const double dAx = x * (A1.x - p.x * x);
    const double dAy = x * (A1.y - p.y * x);
    if (OPT == OPT_SGD) {  // sgd.nim:220-222, averaged per coordinate below
      acc.x += r.etaP * (r.dL * dAx);
      acc.y += r.etaP * (r.dL * dAy);
      seta += r.etaP;
      if (do_w) {
        wacc.a0 += r.etaw * (r.dL * x);
        wacc.a1 += r.etaw;
      }
    } else {  // adagrad.nim:122-124
      const double gx = r.dL * dAx, gy = r.dL * dAy;
      acc.x += gx;
      acc.y += gy;
      accn.x += gx * gx;
      accn.y += gy * gy;
      if (do_w) {
        const double gw = r.dL * x;
        wacc.a0 += gw;
        wacc.a1 += gw * gw;
      }
    }
  };
  auto finish = [&](const ColHdr& h, const ColStage<OPT>& s, const Gath& G) {
    if (h.cnt <= 0) return;
    const int64_t j = h.j;
    const size_t e = M.row(0, j) * M.Kp + 2 * l;
    const int64_t t0 = h.t0, t1 = t0 + h.cnt;
    const bool do_w = M.fit_linear && j < M.d;
    double2 stored = s.st, g2 = s.g2, n2 = s.n2, p;
    if (OPT == OPT_SGD) {
      p.x = sP * stored.x;
      p.y = sP * stored.y;
    } else if (a.use_stored) {
      p = stored;
    } else {
      p.x = dev::adagrad_param(g2.x, n2.x, O.eta0, tmpP);
      p.y = dev::adagrad_param(g2.y, n2.y, O.eta0, tmpP);
      if (O.track_viol) {  // adagrad.nim:96-99
        viol += fabs(stored.x - p.x) + fabs(stored.y - p.y);
        dev::st_stream(M.P + e, p);
      }
    }
    double2 acc = {0.0, 0.0}, accn = {0.0, 0.0};
    double seta = 0.0;
    WAcc wacc;
#pragma unroll
    for (int q = 0; q < TU; ++q)
      if (q < h.cnt) add_touch(G.r[q], G.A1[q], G.x[q], p, do_w, acc, accn, seta, wacc);
    if (h.cnt > TU) {  // longer lists (a minority in this regime): the rest is fetched here, TU touches at a time
      for (int64_t tb = t0; tb < t1; tb += L) {
        int pib_l = s.pib;
        double x_l = s.x;
        if (tb != t0) {
          const int64_t tl = tb + l;
          pib_l = tl < t1 ? a.tpos[tl] : 0;
          x_l = tl < t1 ? a.tx[tl] : 0.0;
        }
        const int cnt = (int)(t1 - tb < L ? t1 - tb : L);
        for (int ub = tb == t0 ? TU : 0; ub < cnt; ub += TU) {
          int pib[TU];
          double x[TU];
          SampleRec r[TU];
          double2 A1[TU];
#pragma unroll
          for (int q = 0; q < TU; ++q) {
            const int src = gbase + ((ub + q) < cnt ? (ub + q) : ub);
            pib[q] = __shfl(pib_l, src, kWave);
            x[q] = dev::shfl_d(x_l, src);
          }
#pragma unroll
          for (int q = 0; q < TU; ++q) {
            r[q] = a.rec[pib[q]];
            A1[q] = *reinterpret_cast<const double2*>(a.Abuf + (size_t)pib[q] * a.TA * M.Kp + 2 * l);
          }
#pragma unroll
          for (int q = 0; q < TU; ++q)
            if (ub + q < cnt) add_touch(r[q], A1[q], x[q], p, do_w, acc, accn, seta, wacc);
        }
      }
    }
    const double c = OPT == OPT_SGD ? dev::touch_div((double)(t1 - t0), O.touch_cap) : (double)(t1 - t0);
    if (OPT == OPT_SGD) {
      viol += fabs((acc.x + seta * O.beta * p.x) / c) + fabs((acc.y + seta * O.beta * p.y) / c);
      stored.x = stored.x * s.fP - (acc.x / c) / sPn;
      stored.y = stored.y * s.fP - (acc.y / c) / sPn;
      dev::st_stream(M.P + e, stored);
      if (do_w) {  // fit_linear.nim:41-47
        const double wj = sw * s.wt;
        if (l == 0) {
          viol += fabs((wacc.a0 + wacc.a1 * O.alpha * wj) / c);
          M.w[j] = s.wt * s.fw - (wacc.a0 / c) / swn;
        }
      }
    } else {
      g2.x += acc.x;
      g2.y += acc.y;
      n2.x += dev::ada_norm_inc(acc.x, accn.x, O.ada_cross);
      n2.y += dev::ada_norm_inc(acc.y, accn.y, O.ada_cross);
      dev::st_stream(O.G + e, g2);
      dev::st_stream(O.N + e, n2);
      if (do_w && l == 0) {  // fit_linear.nim:50-57
        if (!a.use_stored) {
          const double wj = -O.eta0 * s.gw / (itp * O.eta0 * O.alpha + sqrt(s.nw));
          viol += fabs(s.wt - wj);
          M.w[j] = wj;
        }
        O.Gw[j] = s.gw + wacc.a0;
        O.Nw[j] = s.nw + dev::ada_norm_inc(wacc.a0, wacc.a1, O.ada_cross);
      }
    }
  };
  // a lane group takes features u, u + stride, ...; NF of them per iteration: their records and A rows are gathered
  // together, while the rows / touch lists of the next NF and the headers of the NF after those are requested
  int64_t u = closer ? a.u1 : a.u0 + ((int64_t)blockIdx.x * kWavesPerBlock + wv) * R + g;
  ColHdr h0[NF], h1[NF];
  ColStage<OPT> s0[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    h0[f] = load_hdr(u + f * stride);
    h1[f] = load_hdr(u + (NF + f) * stride);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) s0[f] = load_stage(h0[f]);
  for (; u < a.u1; u += NF * stride) {
    Gath G[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) G[f] = gather(h0[f], s0[f]);
    ColHdr h2[NF];
    ColStage<OPT> s1[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      h2[f] = load_hdr(u + (2 * NF + f) * stride);
      s1[f] = load_stage(h1[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) finish(h0[f], s0[f], G[f]);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      h0[f] = h1[f];
      h1[f] = h2[f];
      s0[f] = s1[f];
    }
  }
  viol = dev::wave_sum(viol);
  if (lane == 0) red[0][wv] = viol;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < kWavesPerBlock; ++w_) v += red[0][w_];
    a.parts[a.nS + blockIdx.x] = v;
  }
  if (!closer) return;
  col_closer<OPT>(a, red);
}

// adds the last batch's per-block viol partials (every other batch's are folded in by the next batch's closing
// workgroup); defined once, in mb_fm.hip (mb_ffm.hip launches it as well)
__global__ void k_epoch_close(const double* __restrict__ parts, int n, double* __restrict__ out_acc);

// ------------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------------
template <int L, int SPLIT, int OPT, bool GEN>
static void launch_row(hipStream_t st, const RowArgs& ra, int mode, int n_cu, int* n_blocks) {
  const bool sing = ra.single != nullptr;
  constexpr int SPW = kWave / (L * SPLIT);
  const int nA = (ra.len + kWavesPerBlock * SPW - 1) / (kWavesPerBlock * SPW);
  *n_blocks = nA;
  constexpr bool CAN_HOLD = held_entries<L, SPLIT>() > 0;
  constexpr bool CAN_REG = CAN_HOLD && !GEN && OPT == OPT_SGD && L * SPLIT == kWave;
  if (CAN_REG && mode == 2 && ra.nt)
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, (CAN_REG ? 4 : 0), false>), dim3(nA), dim3(kBlock), 0, st, ra);
  else if (CAN_REG && mode == 2)
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, (CAN_REG ? 2 : 0), false>), dim3(nA), dim3(kBlock), 0, st, ra);
  else if (CAN_HOLD && mode == 3)
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, (CAN_HOLD ? 3 : 0), false>), dim3(nA), dim3(kBlock), 0, st, ra);
  else if (CAN_HOLD && mode >= 1)
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, (CAN_HOLD ? 1 : 0), false>), dim3(nA), dim3(kBlock), 0, st, ra);
  else if (sing && !GEN)
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, 0, !GEN>), dim3(nA), dim3(kBlock), 0, st, ra);
  else
    hipLaunchKernelGGL((k_row_phase<L, SPLIT, OPT, GEN, 0, false>), dim3(nA), dim3(kBlock), 0, st, ra);
}

// rows the held mode of lane mapping (L, s) can take: E * L * s entries (k_row_phase)
static int held_capacity(int L, int s) {
  const int lps = L * s;
  const int e = lps >= kWave ? 1 : (lps >= 8 ? (kWave / lps > 4 ? 4 : kWave / lps) : 0);
  return e * lps;
}

template <int L, int OPT, bool GEN>
static int run_batches(nfm_ctx* ctx, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                       int TA) {
  constexpr int R = kWave / L;
  hipStream_t st = ctx->stream;
  const double* Stab = W.Stab.as<double>();
  const double* Dtab = W.Dtab.as<double>();
  const double* it0p = W.itbuf.as<double>();
  const double avg_row = X.n > 0 ? (double)X.nnz / (double)X.n + M.n_aug : 0.0;
  const size_t partsB_half = W.partsB.bytes / sizeof(double) / 2;
  // records / A rows requested together per feature; 2 measured best on cfg2 and the headline shape
  // (4: 104 VGPRs -> 4 waves per SIMD, 2: 80 -> 6)
  const char* tu_env = getenv("NFM_TU");
  const int tu = tu_env ? atoi(tu_env) : 2;
  // Where the singles are updated: as stage 3 of the row phase (default), or by their own kernel
  // between the row and the column phase (NFM_SINGLES_KERNEL=1: 66 + 109 us vs 157 us fused at k = 64;
  // as extra workgroups of the column launch the sum of the times was conserved as well: the memory
  // system is the limit, not latency).
  const bool have_singles = !GEN && P.use_singles;
  // tables far beyond the 256 MB Infinity Cache, batches that visit most rows once: stream them (st_nt / ld_nt above);
  // NFM_STREAM=0|1 overrides
  static const int stream_env = getenv("NFM_STREAM") ? atoi(getenv("NFM_STREAM")) : -1;
  const size_t table_bytes = sizeof(double) * (size_t)M.nb * M.da * M.Kp * (OPT == OPT_ADAGRAD ? 3 : 1);
  const bool stream_rows = stream_env >= 0 ? stream_env != 0 : (have_singles && table_bytes > ((size_t)384 << 20));
  static const bool env_kernel = getenv("NFM_SINGLES_KERNEL") && atoi(getenv("NFM_SINGLES_KERNEL")) != 0;
  const bool singles_in_row = have_singles && !env_kernel;
  const bool singles_in_col = have_singles && !singles_in_row;
  int n_prev = 0;
  for (int64_t b = 0; b < P.n_batches; ++b) {
    const int64_t p0 = P.bat_pos[b];
    const int len = (int)(P.bat_pos[b + 1] - p0);
    // MBPSGD: the forward pass is AdaGrad's row phase reading the stored parameters (its record carries the
    // raw dloss); the optimizer's `it` advances once per mini-batch (minibatch_psgd.nim:121), not per sample
    const int use_stored = (OPT == OPT_PSGD || (OPT == OPT_ADAGRAD && P.first_singleton && b == 0)) ? 1 : 0;
    const double it_b = OPT == OPT_PSGD ? (double)b : (double)p0;
    constexpr int ROPT = OPT == OPT_PSGD ? OPT_ADAGRAD : OPT;
    const int split = choose_split(L, len, avg_row, ctx->n_cu);
    // row-phase mode (k_row_phase): 2 = held entries + register-resident rows (SGD, one sample per
    // wavefront, a batch with singles: 106 vs 141 us per batch on the headline shape), 1 = held entries,
    // 3 = held entries in chunks (rows longer than the held capacity), 0 = streamed (models with several
    // orders, samples spread over fewer than 8 lanes).
    // NFM_HELD=0 / NFM_NQ=0 switch the modes off (tuning).
    auto mode_for = [&](int s_used) {
      static const bool held_on = !(getenv("NFM_HELD") && atoi(getenv("NFM_HELD")) == 0);
      static const bool reg_on = !(getenv("NFM_NQ") && atoi(getenv("NFM_NQ")) == 0);
      if (!held_on || held_capacity(L, s_used) == 0 || M.nb == 0) return 0;  // degree-1 models: linear term only, streamed
      if (X.max_row + M.n_aug > held_capacity(L, s_used)) return 3;  // long rows: chunks of held entries
      if (!GEN && reg_on && OPT == OPT_SGD && singles_in_row && L * s_used == kWave) return 2;
      return 1;
    };
    int nA;
    {
      RowArgs ra{X, M, O, P.has_perm ? P.perm.as<int64_t>() : nullptr, P.begin, p0, len, use_stored, TA, stream_rows ? 1 : 0, it_b, it0p,
                 OPT == OPT_SGD ? Stab + 2 * b : M.sc, OPT == OPT_SGD ? Stab + 2 * (b + 1) : M.sc,
                 singles_in_row ? P.toff.as<int64_t>() : nullptr,
                 singles_in_row ? P.single.as<uint8_t>() : nullptr, W.Abuf.as<double>(), W.rec.as<SampleRec>(),
                 W.partsA.as<PartA>()};
      TimedLaunch tl(ctx, "row_phase");
      int s_used;
      // AdaGrad, 32 < k <= 64, rows of at most 64 entries, singles in the row phase: two wavefronts per sample with the
      // state rows resident in registers (k_row_phase_ada2; NFM_ADA2=0 switches it off)
      static const bool ada2_on = !(getenv("NFM_ADA2") && atoi(getenv("NFM_ADA2")) == 0);
      if (OPT == OPT_ADAGRAD && !GEN && L == 32 && ada2_on && singles_in_row && !use_stored && X.max_row + M.n_aug <= 64) {
        nA = (len + 1) / 2;
        if (stream_rows)
          hipLaunchKernelGGL((k_row_phase_ada2<OPT_ADAGRAD, true>), dim3(nA), dim3(kBlock), 0, st, ra);
        else
          hipLaunchKernelGGL((k_row_phase_ada2<OPT_ADAGRAD, false>), dim3(nA), dim3(kBlock), 0, st, ra);
        s_used = 2;
      } else
      if (R >= 16 && split >= 16) { launch_row<L, (R >= 16 ? 16 : R), ROPT, GEN>(st, ra, mode_for(R >= 16 ? 16 : R), ctx->n_cu, &nA); s_used = R >= 16 ? 16 : R; }
      else if (R >= 8 && split >= 8) { launch_row<L, (R >= 8 ? 8 : R), ROPT, GEN>(st, ra, mode_for(R >= 8 ? 8 : R), ctx->n_cu, &nA); s_used = R >= 8 ? 8 : R; }
      else if (R >= 4 && split >= 4) { launch_row<L, (R >= 4 ? 4 : R), ROPT, GEN>(st, ra, mode_for(R >= 4 ? 4 : R), ctx->n_cu, &nA); s_used = R >= 4 ? 4 : R; }
      else if (R >= 2 && split >= 2) { launch_row<L, (R >= 2 ? 2 : R), ROPT, GEN>(st, ra, mode_for(R >= 2 ? 2 : R), ctx->n_cu, &nA); s_used = R >= 2 ? 2 : R; }
      else { launch_row<L, 1, ROPT, GEN>(st, ra, mode_for(1), ctx->n_cu, &nA); s_used = 1; }
      (void)s_used;  // nA = workgroups launched (their per-workgroup partials are what the closer adds up)
    }
    const int64_t u0 = P.bat_uoff[b], u1 = P.bat_uoff[b + 1];
    const int per_block = kWavesPerBlock * R;
    int nB = (int)((u1 - u0 + per_block - 1) / per_block);
    // Many short-lived wavefronts (sparse batches at large k: ~49k wavefronts of two features with two
    // touches each) spend their time being launched: beyond four resident sets the grid is capped at
    // 8 workgroups per CU and the workgroups stride over the features (headline shape: 70 -> 58 us).
    // Fewer, longer wavefronts (cfg2: 12.5k wavefronts of 8 features x 10 touches) balance better
    // when the hardware hands out workgroups one by one (capped: 46 us, uncapped: 38 us).
    // The capped grid is exactly ONE resident set: what the strided kernel's register count lets a CU hold (headline
    // shape, SGD: 95 registers = 5 workgroups per CU: 8 per CU left three waiting for a slot, a second round with a long
    // tail -- column phase 56.1 -> 50.9 us; 6 per CU: 64 us; 10: 53.8 us).  NFM_COL_WG overrides.
    // one order of degree 2, SGD / AdaGrad: the software-pipelined walk (k_col_sparse); NFM_COL_PIPE=0 switches it off
    static const bool col_pipe_env = !(getenv("NFM_COL_PIPE") && atoi(getenv("NFM_COL_PIPE")) == 0);
    const bool col_pipe = col_pipe_env && !GEN && M.nb == 1 && OPT != OPT_PSGD;
    static const int col_occ = [] {
      int nb_ = 0, np_ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, k_col_phase<L, OPT, GEN, 2, true>, kBlock, 0) != hipSuccess || nb_ < 1) nb_ = 8;
      if (!GEN && OPT != OPT_PSGD && col_pipe_env &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&np_, k_col_sparse<L, (OPT == OPT_PSGD ? OPT_SGD : OPT)>, kBlock, 0) == hipSuccess && np_ >= 1)
        nb_ = np_;
      return nb_;
    }();
    static const int col_wg_per_cu = getenv("NFM_COL_WG") ? atoi(getenv("NFM_COL_WG")) : col_occ;
    // the tuning variants TU = 1 / 4 (NFM_TU) are not strided
    static const int cap_sets = getenv("NFM_COL_CAP_SETS") ? atoi(getenv("NFM_COL_CAP_SETS")) : 4;  // tuning
    const bool nB_capped = tu == 2 && col_wg_per_cu > 0 && nB > cap_sets * ctx->n_cu * col_wg_per_cu;
    if (nB_capped) nB = ctx->n_cu * col_wg_per_cu;
    nB += 1;  // + the closing workgroup
    const int nS = singles_in_col ? (len + kWavesPerBlock - 1) / kWavesPerBlock : 0;
    // the singles kernel writes parts[0, nS), the column phase parts[nS, nS + nB)
    double* parts_cur = W.partsB.as<double>() + (b & 1) * partsB_half;
    const double* parts_prev = W.partsB.as<double>() + ((b + 1) & 1) * partsB_half;
    {
      ColArgs ca{X, M, O, P.has_perm ? P.perm.as<int64_t>() : nullptr, singles_in_col ? P.toff.as<int64_t>() : nullptr,
                 singles_in_col ? P.single.as<uint8_t>() : nullptr, P.begin, p0, len, nS,
                 P.ucol.as<int32_t>(), P.uptr.as<int64_t>(), P.ucol_s.as<int32_t>(), P.ubeg_s.as<int64_t>(), P.ucnt_s.as<int32_t>(),
                 P.tpos.as<int32_t>(), P.tx.as<double>(), u0, u1,
                 OPT == OPT_SGD ? Stab + 2 * b : M.sc, OPT == OPT_SGD ? Stab + 2 * (b + 1) : M.sc,
                 OPT == OPT_SGD ? Dtab + 4 * b : nullptr,
                 OPT == OPT_SGD ? W.Ftab.as<double>() + (size_t)b * 2 * kFtab : nullptr, W.Abuf.as<double>(),
                 W.rec.as<SampleRec>(), parts_cur, W.partsA.as<PartA>(), parts_prev, W.out_acc.as<double>(), it_b,
                 (double)len, it0p, TA, use_stored, nA, n_prev};
      if (nS > 0) {
        TimedLaunch tls(ctx, "singles");
        hipLaunchKernelGGL((k_singles<L, OPT>), dim3(nS), dim3(kBlock), 0, st, ca);
      }
      TimedLaunch tl(ctx, "col_phase");
      const bool strided = nB_capped;
      if (tu == 1 && OPT != OPT_PSGD)  // the tuning variants are not instantiated for MBPSGD
        hipLaunchKernelGGL((k_col_phase<L, (OPT == OPT_PSGD ? OPT_SGD : OPT), GEN, 1, false>), dim3(nB), dim3(kBlock), 0, st, ca);
      else if (tu == 4 && OPT != OPT_PSGD)
        hipLaunchKernelGGL((k_col_phase<L, (OPT == OPT_PSGD ? OPT_SGD : OPT), GEN, 4, false>), dim3(nB), dim3(kBlock), 0, st, ca);
      else if (strided && col_pipe)
        hipLaunchKernelGGL((k_col_sparse<L, (OPT == OPT_PSGD ? OPT_SGD : OPT)>), dim3(nB), dim3(kBlock), 0, st, ca);
      else if (strided)
        hipLaunchKernelGGL((k_col_phase<L, OPT, GEN, 2, true>), dim3(nB), dim3(kBlock), 0, st, ca);
      else
        hipLaunchKernelGGL((k_col_phase<L, OPT, GEN, 2, false>), dim3(nB), dim3(kBlock), 0, st, ca);
    }
    int nH = 0;
    if (M.nb > 0 && P.bat_hoff[b + 1] > P.bat_hoff[b]) {
      const int PW = 2 * M.Kp + 4;
      HeavyArgs ha{P.hv_u.as<int64_t>(), P.hv_seg0.as<int64_t>(), P.bat_hoff[b], P.bat_hoff[b + 1], P.bat_soff[b],
                   P.bat_soff[b + 1], W.hpart.as<double>(), parts_cur + nS + nB, PW, 0};
      ColArgs ca{X, M, O, P.has_perm ? P.perm.as<int64_t>() : nullptr, nullptr, nullptr, P.begin, p0, len, nS,
                 P.ucol.as<int32_t>(), P.uptr.as<int64_t>(), P.ucol_s.as<int32_t>(), P.ubeg_s.as<int64_t>(), P.ucnt_s.as<int32_t>(),
                 P.tpos.as<int32_t>(), P.tx.as<double>(), u0, u1,
                 OPT == OPT_SGD ? Stab + 2 * b : M.sc, OPT == OPT_SGD ? Stab + 2 * (b + 1) : M.sc,
                 OPT == OPT_SGD ? Dtab + 4 * b : nullptr,
                 OPT == OPT_SGD ? W.Ftab.as<double>() + (size_t)b * 2 * kFtab : nullptr, W.Abuf.as<double>(),
                 W.rec.as<SampleRec>(), parts_cur, W.partsA.as<PartA>(), parts_prev, W.out_acc.as<double>(), it_b,
                 (double)len, it0p, TA, use_stored, nA, n_prev};
      const int nsb = (int)((ha.s1 - ha.s0 + per_block - 1) / per_block);
      nH = (int)((ha.h1 - ha.h0 + kWavesPerBlock - 1) / kWavesPerBlock);  // one wavefront per heavy feature
      {
        TimedLaunch tl(ctx, "heavy_partial");
        hipLaunchKernelGGL((k_heavy_partial<L, OPT, GEN>), dim3(nsb), dim3(kBlock), 0, st, ca, ha);
      }
      TimedLaunch tl(ctx, "heavy_apply");
      hipLaunchKernelGGL((k_heavy_apply<L, OPT, GEN>), dim3(nH), dim3(kBlock), 0, st, ca, ha);
    }
    n_prev = nB + nS + nH;
    if (OPT == OPT_PSGD && O.gradP == nullptr) launch_psgd_step(ctx, M, O, W, it0p, it_b);
    if (W.after_batch) {
      // (recording a data-parallel epoch as graphs: the stretch up to here ends, the exchange runs outside the capture)
      const bool cut = W.seg_recording && W.is_sync && W.is_sync(b);  // (also after the last mini-batch: the hook must never be captured)
      if (cut) NFM_TRY(W.seg_cut_here(ctx, b));
      NFM_TRY(W.after_batch(b));
      if (cut) NFM_TRY(W.seg_resume(ctx));
    }
  }
  if (P.n_batches > 0) {
    const double* parts_last = W.partsB.as<double>() + ((P.n_batches - 1) & 1) * partsB_half;
    hipLaunchKernelGGL(k_epoch_close, dim3(1), dim3(kBlock), 0, st, parts_last, n_prev, W.out_acc.as<double>());
  }
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

}  // namespace nfm
