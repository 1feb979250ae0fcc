// nimfm_amd/csrc/seqwin.hip -- NFM_MODE_SEQUENTIAL at throughput: the reference's one-sample-at-a-time order
// (optimizer/sgd.nim:246-258,294-308; optimizer/adagrad.nim:169-184) run as a DEPENDENCY WINDOW over the whole chip.
//
// What the order really serialises.  Sample t of the order reads and rewrites (a) the parameter rows of its own
// features, (b) the intercept (SGD) / the intercept's state (AdaGrad), (c) the step counter and, for SGD, the two lazy
// L2 scales.  (c) does not depend on the data: a product of factors that are functions of `it` alone.  (b) is a scalar
// chain  b_t -> yhat_t = ((b_t + w x ...) + ...) + anova -> dL_t -> b_{t+1}  that nothing can shorten without changing
// the rounding.  (a) ties t only to the EARLIER samples that share a feature with it -- at d = 1e6, 64 entries per row,
// two samples do with probability 0.4 %.  So:
//
//   * W workers (one workgroup = one CU each; one wavefront, four for field-aware models and degrees >= 3) take the samples round-robin: worker s owns
//     positions s, s + W, s + 2W ...  A worker waits until every earlier sample that shares a feature with its sample has
//     written its rows (a per-entry "previous position with this feature" table, built once per order by one sort, and a
//     completion counter per worker), gathers the rows, forms everything of predictWithGrad (sgd.nim:191-202) that does
//     not need the intercept -- the per-factor sums in the reference's entry order, their sum over the factors in
//     ascending order, the products w_j x_j -- and posts them to its mailbox.
//   * ONE conductor workgroup walks the samples in order: a fetch wavefront collects the mailboxes (several loads in
//     flight) into an LDS ring, a chain wavefront adds  b + w_1 x_1 + ... + w_m x_m + anova  in the reference's order,
//     takes dloss, steps the intercept (or its AdaGrad state) and posts {dL, yhat} back.  This chain -- about m + 100
//     dependent fp64 operations per sample -- is the one thing that runs at the speed of a single thread; everything else
//     of the step is spread over the workers.
//   * The worker then performs update() (sgd.nim:205-243 / adagrad.nim:113-134) for its rows, drains its stores and
//     bumps its completion counter.
//   * SGD's scale chain (scaling_P *= 1 - eta_P beta, sgd.nim:233-234) is produced ahead of the launch by one
//     wavefront (k_win_scales: step sizes of 64 samples in parallel, the two products in order); it also finds the
//     sample after which a scale drops below 1e-9 (resetScaling, sgd.nim:116-131): the launch ends there, the dense
//     rescale runs, the next launch continues.
//
// Four workers: win_worker_k64 (degree 2, rows of 64 factors, up to 64 entries: the rows stay in registers), win_worker
// (degree 2, any row shape: rows in LDS), win_worker_fmx (several orders / degree <= 6), win_worker_ffm (field-aware: one
// chain term per pair of entries).  A dependency on a sample fewer than W positions back does not wait for that sample's
// update: the writer posts a RECIPE (row as used + per-factor sums or derivative) before the conductor answers, the
// successor forms the new row itself from it and the conductor's dL for the writer's sample.
//
// Same arithmetic, same order of every sum as the one-workgroup kernels (seq.hip): parameters, linear weights, intercept and
// AdaGrad state come out BIT FOR BIT equal to the one-workgroup kernels (tests/test_gpu_seqwin.py); only the epoch's
// loss / viol totals are associated differently (per worker, then in worker order).
//
// Cross-CU visibility (MI355X: per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by another
// CU's stores): every byte that one workgroup writes and another reads during the launch -- parameter rows, linear
// weights, AdaGrad state, mailboxes, counters -- is stored AND loaded with agent-scope relaxed atomics (global_load /
// global_store ... sc1, 8 bytes), every signalling store follows an s_waitcnt vmcnt(0) of the wavefront that wrote the
// data, and a consumer loads the data only after ITS OWN poll has seen the signal.  Mailbox words are self-validating
// (a reserved signalling-NaN pattern means "not written yet"), so they need no separate flag.  One workgroup per CU
// (LDS request), W + 1 <= number of CUs: all workgroups are resident, every wait is on a workgroup that is running.
// Every spin has a wall-clock limit and watches an abort word: a launch always drains.
#include <hipcub/hipcub.hpp>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "fm_device.h"
#include "opt_views.h"

namespace nfm {

typedef unsigned long long ull;
constexpr ull kWinSentinel = 0x7FF4DEADBEEF0000ull;  // a signalling NaN no arithmetic produces (results are quiet NaNs)
constexpr ull kWinQuietNaN = 0x7FF8000000000000ull;
constexpr int kWinRing = 8;    // LDS ring slots between the conductor's fetch and chain wavefronts (power of two)
constexpr int kWinDepth = 8;   // mailboxes the fetch wavefront has requested ahead (must stay < W, see below)
constexpr int kWinHdr = 4;     // mailbox tail, after the MC term slots: anova sum, target, eta(alpha0) | eta0 (it-1) alpha0, number of chain terms
constexpr int kWinMaxNL = 5;   // 64-lane loads per mailbox: rows of up to 316 entries
constexpr long long kWinTimeoutTicks = 100000000ll;  // 1 s of the 100 MHz wall clock without progress: abort (every legitimate wait is microseconds)

struct WinArgs {
  CsrView X;
  ModelView M;
  OptView O;
  const int64_t* perm;    // absolute positions, or null
  int64_t begin;          // first position of the call
  int64_t seg0, n_seg;    // this launch: positions begin + seg0 ... begin + seg0 + n_seg - 1
  int64_t it0;            // step counter of the launch's first sample
  const int32_t* prev;    // [nnz] position (relative to `begin`) of the previous sample of the call with this feature, -1: none
  const uint8_t* prevq;   // [nnz] the feature's entry index in that sample's row (rows of up to 64 entries: the forwarding path)
  const int32_t* next;    // [nnz] position of the next sample of the call with this feature, -1: none
  ull* fw;                // [W][2][kFwSlot] forwarding areas: the recipes of a worker's rows that a near successor asks for
  long long* trace;       // debugging (NFM_SEQ_WIN_TRACE=1): [n_seg][8] wall-clock stamps per sample, or null
  const double* scales;   // SGD: [ns][2] {scale_P, scale_w} BEFORE each sample of the call
  ull* fwd;               // [W][2][FW] worker -> conductor
  ull* res;               // [W][2][4]  conductor -> worker(s): {dL, yhat} as tagged granules
  unsigned* completed;    // [W] samples of this launch a worker has finished
  unsigned* ctrl;         // [0]: abort
  double* partial;        // [W + 1][2] {loss, viol} per worker, last: the conductor's
  int W, lgW, m_cap, FW, lgKp;  // FW = MC + kWinHdr words per mailbox, MC = m_cap rounded up to the chain's chunk
  int near_r;             // a dependency on a sample fewer than near_r positions back takes the recipe path (W with a conductor)
  int thr;                // without a conductor: before sample u every sample below u - thr W is complete (thr W + near_r <= np W:
                          // a worker's buffer set is reused np W positions later, and only readers within near_r look at it)
  int np;                 // mailboxes / answer words / forwarding areas per worker (a power of two): sample u of a worker uses
                          // number (u / W) mod np, i.e. they are reused np samples of that worker later
  int no_cond;            // 1: fitIntercept = false -- no scalar chain ties the samples, there is NO conductor: a worker adds up its
                          // sample's prediction itself (intercept constant, the entries' terms in storage order, the interaction
                          // sum: predictWithGrad, sgd.nim:193-201), takes dloss and posts {dL, yhat} for its near successors
  int dead_slot;          // -DNFM_TEST_HOOKS builds only (NFM_SEQ_WIN_TEST_DEAD_SLOT): this worker leaves at once, as a workgroup that
                          // never became resident would; -1 otherwise.  The others time out, the launch aborts, the host restores + falls back
  int par_min;            // one-term conductor: the shortest chunk of ready samples whose chain is solved in parallel (0: never)
  int first_worker;       // workgroups ahead of the workers: 1 (workgroup 0 is the conductor) or 0 (no conductor: none is launched)
  int one_term;           // 1 (the default with a conductor; NFM_SEQ_WIN_EXACT=1 turns it off): the worker adds up everything of its
                          // sample's prediction but the intercept -- S = sum_j w_j x_j (storage order) + the interaction sum -- and posts
                          // ONE term; the conductor's chain is  yhat = b + S -> dloss -> b'  (win_conductor_sum).  Same sample order and
                          // dependencies; yhat is rounded as b + (sum) instead of ((b + t1) + t2) + ...: results agree with the
                          // term-by-term chain to ~1e-15 relative per step, not to the last bit
};

__device__ __forceinline__ ull ld_u64(const ull* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u64(ull* p, ull v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a store through an address that was SELECTED between two arrays: the pointer is given its address space explicitly (a
// select of pointers loses it, and an sc1 store through a flat_ instruction is not the hand-off this kernel relies on)
typedef __attribute__((address_space(1))) double global_double;
__device__ __forceinline__ void st_f64_at(unsigned long long addr, double v) {
  __hip_atomic_store(reinterpret_cast<global_double*>(addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ ull mail_bits(double v) {  // a value as a mailbox word: never the "empty" pattern
  return v != v ? kWinQuietNaN : (ull)__double_as_longlong(v);
}
__device__ __forceinline__ void compiler_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_fence_order() { asm volatile("" ::: "memory"); }  // LDS operations of one wavefront execute in order

// A spin that always ends: the abort word of the launch and a wall-clock limit, looked at every 64 rounds.
struct Spin {
  long long t0;
  int n = 0;
  __device__ Spin() : t0(wall_clock64()) {}
  // true: give up (the launch is being aborted)
  __device__ __forceinline__ bool wait(unsigned* ctrl) {
    __builtin_amdgcn_s_sleep(2);
    if ((++n & 63) != 0) return false;
    if (ld_u32(ctrl) != 0u) return true;
    if (wall_clock64() - t0 > kWinTimeoutTicks) {
      st_u32(ctrl, 1u);
      return true;
    }
    return false;
  }
};

// A forwarded value: two 8-byte granules {32-bit tag, half of the double}; the tag is the writer's sample index + 1, so
// a granule needs no reset and no flag -- a reader takes the value when both tags are the sample it waits for.
__device__ __forceinline__ void fw_store(ull* p, unsigned tag, double v) {
  st_u64(p, ((ull)tag << 32) | (ull)(unsigned)__double2loint(v));
  st_u64(p + 1, ((ull)tag << 32) | (ull)(unsigned)__double2hiint(v));
}
__device__ __forceinline__ bool fw_load(const ull* p, unsigned tag, double& v) {
  const ull g0 = ld_u64(p), g1 = ld_u64(p + 1);
  v = __hiloint2double((int)(unsigned)(g1 & 0xffffffffull), (int)(unsigned)(g0 & 0xffffffffull));
  return (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
}
// The ONE-TERM mailbox (WinArgs::one_term): {S, y, the intercept's step size, C1, C2, C12, two writers} as thirteen tagged
// granules, laid out granule-major -- fwd[(granule * np + parity) * W + worker] -- so that the conductor's fetch wavefront
// reads one granule of 64 workers with one 512-byte access.  Tags instead of the "empty" pattern: nothing to reset after use.
// C1 / C12 / writers (SGD): the sample shares ONE feature with a sample u inside the window whose dL need not exist yet.
// The writer's step of the shared row is AFFINE in its dL (sgd.nim:217-223: p' = A + dL B, A and B known at the writer's forward
// pass) and the prediction is multilinear in the rows of different features, so S = S(A) + dL_u C1 exactly: the worker posts
// S(A) and C1 without waiting for dL_u, the conductor -- which made that dL -- finishes the sum (win_conductor_sum).  When u's
// own recipe is affine in the dL of ITS writer w (see the forwarding area below), S = S(A) + dL_u (C1 + dL_w C12).  The format
// carries S = S0 + dL_1 C1 + dL_2 (C2 + dL_1 C12) with two writers as distances (sample - writer; 0: none; the second one may
// lie up to 254 positions back).  (Tried with the C2 slot: TWO writers' rows affine at once, S bilinear in their dL -- measured
// +-0 on the headline and cfg2 shapes, twice: what the conductor waits for are chains of recipes, not second writers.  Not kept.)
constexpr int kSumGran = 13;
__device__ __forceinline__ void post_sum(ull* fwd, int np, int W, int slot, int par, unsigned tag, int lane, double S, double y, double h2,
                                         double C1 = 0.0, double C2 = 0.0, double C12 = 0.0, unsigned dists = 0u) {
  if (lane < kSumGran) {
    const double v = lane < 2 ? S : lane < 4 ? y : lane < 6 ? h2 : lane < 8 ? C1 : lane < 10 ? C2 : C12;
    unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
    if (lane == 12) half = dists;
    st_u64(fwd + ((size_t)(lane * np + par) * W + slot), ((ull)tag << 32) | (ull)half);
  }
}
// A worker's forwarding area (one per parity of its sample count): what a successor within W positions needs to form
// the rows it shares with this sample BY ITSELF as soon as the conductor's dL for this sample lands -- the RECIPE, known
// at this sample's forward pass, not the result, known only after its update:
//   [2][64 lanes]        the per-factor sums a1, and their slope T in the dL of THIS sample's own affine writer (register-
//                        resident worker, SGD: a sample whose shared row is still affine in an earlier sample's dL posts its
//                        recipe at once, with a1 = a1(A) + dL_w T, instead of holding it back until that dL exists)
//   [3][64 q][64 lanes]  per hot row: its value as this sample used it; AdaGrad: g_sum and g_norm as loaded
//   [5][64 q]            per hot entry: linear weight as used, AdaGrad g_sum / g_norm of it, the entry's value; [4][0]: the
//                        distance to that affine writer (0: the recipe is exact, T is unused)
constexpr int kFwVals = 3;
constexpr size_t kFwRows = (size_t)kWave * 4;                                    // granules of a1 and T
constexpr size_t kFwLin = kFwRows + (size_t)kFwVals * kWave * kWave * 2;         // start of the per-entry part
constexpr size_t kFwSlot = kFwLin + (size_t)5 * kWave * 2;                       // granules per (worker, parity)
constexpr int kResWords = 4;  // conductor -> worker: {dL, yhat} as tagged granules (several workers may read them)

// ------------------------------------------------------------------------------------------------------------------
// worker: one wavefront, lanes (r, s) = (row slot, factor), Kp lanes per row, R = 64 / Kp rows per instruction
// ------------------------------------------------------------------------------------------------------------------
template <int OPT>
__device__ __forceinline__ void win_worker(const WinArgs& a, const int slot, double* lds) {
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  constexpr int U = 4;  // rows a lane requests together
  const int lane = threadIdx.x;
  const int Kp = M.Kp, lgK = a.lgKp, k = M.k;
  const int R = kWave >> lgK, r = lane >> lgK, s = lane & (Kp - 1);
  // rows are handled R * U at a time WITHOUT a branch per row (a row past the sample's end: feature 0's row with value 0,
  // its stores go to a scratch row): the per-sample arrays are padded to whole groups
  const int mc = (a.m_cap + R * U - 1) / (R * U) * (R * U);
  const int W = a.W, lgW = a.lgW;
  double* Pl = lds;                                   // [mc][Kp] stored parameter values of the sample's rows
  double* Tl = Pl + (size_t)mc * Kp;                  // [mc][Kp] x_q p_qs, then the derivative
  double* Gl = Tl + (size_t)mc * Kp;                  // AdaGrad: [mc][Kp] g_sum
  double* Nl = Gl + (ADA ? (size_t)mc * Kp : 0);      // AdaGrad: [mc][Kp] g_norm
  double* red = Nl + (ADA ? (size_t)mc * Kp : 0);     // [64]
  double* vl = red + kWave;                           // [mc] values
  double* wl = vl + mc;                               // [mc] stored linear weights (AdaGrad: after update())
  double* gwl = wl + mc;                              // AdaGrad: [mc] g_sum of the linear term
  double* nwl = gwl + (ADA ? mc : 0);                 // AdaGrad: [mc] g_norm
  int* jl = reinterpret_cast<int*>(nwl + (ADA ? mc : 0));  // [mc] feature ids
  int* pl = jl + mc;                                       // [mc] previous position with the same feature
  int* ll = pl + mc;                                       // [mc] 1: the row comes from its writer's recipe (below), not from memory
  unsigned* cnt = reinterpret_cast<unsigned*>(ll + mc);    // [W] completion counters as last seen
  for (int l = lane; l < W; l += kWave) cnt[l] = 0u;
  double loss_acc = 0.0, viol_acc = 0.0;
  const double b_const = M.sc[SC_INTERCEPT];  // (no_cond: the intercept is not fitted and stays what it is)
  auto fw_area = [&](int64_t u_) { return a.fw + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kFwSlot; };
  // (the forwarding areas are laid out for 64 entries of up to 64 factors: entry q's row at stride Kp)
  auto fw_row = [&](ull* base, int v, int q) { return base + kFwRows + ((size_t)(v * kWave + q) * Kp + s) * 2; };
  auto fw_lin = [&](ull* base, int v, int q) { return base + kFwLin + (size_t)(v * kWave + q) * 2; };
  auto res_of = [&](int64_t u_) { return a.res + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kResWords; };

  for (int64_t u = slot; u < a.n_seg; u += W) {
    const int64_t pos = a.seg0 + u, pa = a.begin + pos;
    const int64_t i = a.perm ? a.perm[pa] : pa;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const double y = dev::target_of(X.y[i], M.task);
    const int64_t it = a.it0 + u;
    const double itf = (double)it;
    const int par = (int)((u >> lgW) & (a.np - 1));
    ull* mb = a.fwd + (size_t)(slot * a.np + par) * a.FW;
    const ull* rp = a.res + (size_t)(slot * a.np + par) * kResWords;
    if (a.no_cond && u >= (int64_t)a.thr * W) {
      // No conductor walks the samples in order, so nothing else keeps a fast worker from running ahead: its forwarding
      // area and answer words of sample u - np W are about to be reused, and a near successor of that sample (a position
      // below u - (np - 1) W) may not have read them yet.  Every sample below that must be complete: the workers before
      // this one have finished c - (np - 2) samples, the others one fewer (c = this worker's count) -- the window spans
      // at most np W positions.
      const unsigned c_ = (unsigned)(u >> lgW) - (unsigned)(a.thr - 1);
      Spin sp;
      bool first = true;
      while (true) {
        bool ok = true;
        for (int l = lane; l < W; l += kWave) ok = ok && cnt[l] >= (l < slot ? c_ : c_ - 1u);
        if (__all(ok)) break;
        if (!first && sp.wait(a.ctrl)) return;
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }
    // the sample's first 64 entries also sit one per lane: they are the ones that can take the forwarding path
    const bool e_in = lane < m;
    const int pq = e_in ? a.prev[q0 + lane] : -1;
    const int pqu = e_in ? (int)a.prevq[q0 + lane] : 0;
    const int nq = e_in ? a.next[q0 + lane] : -1;
    bool near;
    {
      const int64_t v = (int64_t)pq - a.seg0;
      const bool pend = v >= 0 && cnt[v & (W - 1)] <= (unsigned)(v >> lgW);
      // a NEAR previous sample (fewer than W positions back, the shared feature among ITS first 64 entries) left the recipe
      // of the shared row in its forwarding area; every other one is waited for by its counter
      near = pend && (pos - (int64_t)pq) < a.near_r && pqu < kWave;
    }
    // One-term window, SGD: the shared row of the MOST RECENT earlier sample inside the window is treated as AFFINE in that
    // sample's dL (post_sum) -- chosen by POSITIONS alone (never by what happens to be finished when this worker looks), so that
    // the arithmetic, and with it every bit of the result, is the same from run to run; provided that writer shares exactly
    // one feature with this sample (two rows moving with the same dL would make S quadratic in it).  The earlier writers' dL
    // come first (the conductor makes them in order) and are waited for.
    int aff_q = -1;
    ull aff_bit = 0ull;
    if constexpr (!ADA) {
      if (a.one_term) {
        const int64_t vq_ = (int64_t)pq - a.seg0;
        const bool cand = vq_ >= 0 && (pos - (int64_t)pq) < a.near_r && pqu < kWave;
        int latest = cand ? (int)vq_ : -1;
#pragma unroll
        for (int sh = 1; sh < kWave; sh <<= 1) {
          const int o_ = __shfl_xor(latest, sh, kWave);
          latest = o_ > latest ? o_ : latest;
        }
        const ull who = __ballot(cand && (int)vq_ == latest);
        if (who != 0ull && (who & (who - 1)) == 0ull) {
          aff_q = __builtin_ctzll(who);
          aff_bit = who;
        }
      }
    }
    near = near || lane == aff_q;
    const ull fwdmask = __ballot(near);
    const ull hotmask = __ballot(nq >= 0 && ((int64_t)nq - pos) < a.near_r);  // rows a near successor will ask the recipe of
    if (a.trace && lane == 0) a.trace[u * 8 + 0] = wall_clock64();  // sample taken up
    for (int q = lane; q < mc; q += kWave) {
      const bool in = q < m;
      jl[q] = in ? X.indices[q0 + q] : 0;
      vl[q] = in ? X.data[q0 + q] : 0.0;
      pl[q] = in ? a.prev[q0 + q] : -1;
      ll[q] = (q < kWave && near) ? 1 : 0;
    }
    compiler_fence();
    double* const junk = reinterpret_cast<double*>(a.fw + (size_t)a.np * W * kFwSlot) + (size_t)slot * 2 * kWave + (lane & (kWave - 1));

    // ---- A. every earlier sample of this launch that shares a feature has written its rows (far ones only) ----
    {
      Spin sp;
      bool first = true;
      while (true) {
        bool need = false;
        for (int q = lane; q < m; q += kWave) {
          const int64_t v = (int64_t)pl[q] - a.seg0;
          if (v >= 0 && !ll[q] && cnt[v & (W - 1)] <= (unsigned)(v >> lgW)) need = true;
        }
        if (!__any(need)) break;
        if (!first && sp.wait(a.ctrl)) return;
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }

    // ---- B. rows -> LDS (AdaGrad: update(), adagrad.nim:87-110, for the sample's rows first) ----
    double sP = 1.0, sw = 1.0;
    if constexpr (!ADA) {
      sP = a.scales[2 * pos];
      sw = a.scales[2 * pos + 1];
    }
    const double itp = (double)(it - 1);
    const double tmpP = O.eta0 * itp * O.beta;
    for (int qb = 0; qb < m; qb += R * U) {
      double v_[U], g_[ADA ? U : 1], n_[ADA ? U : 1];
      int j_[U];
#pragma unroll
      for (int t = 0; t < U; ++t) j_[t] = jl[qb + t * R + r];  // (0 past the end)
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const size_t e = (size_t)j_[t] * Kp + s;
        v_[t] = ld_f64(M.P + e);
        if constexpr (ADA) {
          g_[t] = ld_f64(O.G + e);
          n_[t] = ld_f64(O.N + e);
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int q = qb + t * R + r;
        const bool in = q < m && !ll[q];  // (a forwarded row is done again in C)
        double p = v_[t];
        if constexpr (ADA) {
          if (it != 1) {
            p = dev::adagrad_param(g_[t], n_[t], O.eta0, tmpP);
            viol_acc += in ? fabs(v_[t] - p) : 0.0;
            st_f64_at(in ? (ull)(M.P + (size_t)j_[t] * Kp + s) : (ull)junk, p);
          }
          Gl[(size_t)q * Kp + s] = g_[t];
          Nl[(size_t)q * Kp + s] = n_[t];
        }
        Pl[(size_t)q * Kp + s] = p;
        Tl[(size_t)q * Kp + s] = vl[q] * (sP * p);
      }
    }
    const double denw = itp * O.eta0 * O.alpha;
    for (int q = lane; q < m; q += kWave) {
      const int j = jl[q];
      double wv = ld_f64(M.w + j);
      if constexpr (ADA) {
        if (M.fit_linear) {
          const double gw = ld_f64(O.Gw + j), nw_ = ld_f64(O.Nw + j);
          gwl[q] = gw;
          nwl[q] = nw_;
          if (it != 1 && !ll[q]) {
            const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
            viol_acc += fabs(wv - nv);
            st_f64(M.w + j, nv);
            wv = nv;
          }
        }
      }
      wl[q] = wv;
    }
    compiler_fence();

    // ---- B2. near dependencies: the writer's recipe + the conductor's dL for the WRITER's sample -> the row as the writer
    // will (or did) write it, formed here with the writer's own arithmetic (no wait for its update, store and counter) ----
    // One-term window, SGD, ONE such row whose writer's dL is not there yet: the row is affine in that dL (A + dL B) and the
    // sample's sum S with it (post_sum): S(A) and the slope are posted at once, the exact row is formed when dL arrives
    bool affine = false;  // (uniform)
    double aff_A = 0.0, aff_B = 0.0, aff_Bw = 0.0;
    const bool mine = (fwdmask >> lane) & 1ull;  // lane = entry
    const int64_t upl = (int64_t)pq - a.seg0;
    const unsigned tagl = (unsigned)(upl + 1);
    ull* srcl = fw_area(mine ? upl : 0);
    const ull* rsrcl = res_of(mine ? upl : 0);
    double wu = 0.0, gwu = 0.0, nwu = 0.0, vsl = 0.0, dLl = 0.0;
    double sPul = 1.0, etaPul = 0.0, sPnul = 1.0, swul = 1.0, etawul = 0.0;
    bool okl = true;
    auto load_lin = [&]() {
      okl = true;
      if (mine) {
        okl = fw_load(fw_lin(srcl, 0, pqu), tagl, wu);
        if (ADA && M.fit_linear) {
          okl = fw_load(fw_lin(srcl, 1, pqu), tagl, gwu) && okl;
          okl = fw_load(fw_lin(srcl, 2, pqu), tagl, nwu) && okl;
        }
        okl = fw_load(fw_lin(srcl, 3, pqu), tagl, vsl) && okl;
      }
    };
    // the recipe of shared row q (all row slots load it): the writer's per-factor sums, the row as it used it (+ AdaGrad's state)
    auto load_recipe = [&](int q, double& a1u, double& pv, double& gv, double& nv) -> bool {
      const int64_t up = (int64_t)__builtin_amdgcn_readlane(pq, q) - a.seg0;
      const unsigned tag = (unsigned)(up + 1);
      ull* src = fw_area(up);
      const int qu = __builtin_amdgcn_readlane(pqu, q);
      bool ok = fw_load(src + (size_t)s * 2, tag, a1u);
      ok = fw_load(fw_row(src, 0, qu), tag, pv) && ok;
      if constexpr (ADA) {
        ok = fw_load(fw_row(src, 1, qu), tag, gv) && ok;
        ok = fw_load(fw_row(src, 2, qu), tag, nv) && ok;
      }
      return ok;
    };
    // the exact rows of the near dependencies in `which` (and the shared features' linear weights) once the writers' dL are in hand
    int have_q = -1;  // the shared row whose recipe is kept in (have_a1, have_pv): read before the mailbox was posted
    double have_a1 = 0.0, have_pv = 0.0;
    auto form_near_rows = [&](ull which) -> bool {
      for (ull mk = which; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        double a1u = have_a1, pv = have_pv, gv = 0.0, nv = 0.0;
        if (q != have_q) {
          Spin sp;
          while (!__all(load_recipe(q, a1u, pv, gv, nv)))
            if (sp.wait(a.ctrl)) return false;
        }
        const double vsu = dev::shfl_d(vsl, q), dLu = dev::shfl_d(dLl, q);
        const size_t e = (size_t)jl[q] * Kp + s;
        double p;
        if constexpr (ADA) {  // the writer's updateG of this row (adagrad.nim:113-134), then this sample's update() of it
          const double grad = dLu * (vsu * (a1u - pv * vsu));
          const double g = gv + grad, n = nv + grad * grad;
          p = pv;
          if (it != 1) {
            p = dev::adagrad_param(g, n, O.eta0, tmpP);
            if (r == 0) {
              viol_acc += fabs(pv - p);
              st_f64(M.P + e, p);
            }
          }
          Gl[(size_t)q * Kp + s] = g;
          Nl[(size_t)q * Kp + s] = n;
        } else {  // the writer's update() of this row (sgd.nim:217-223), with ITS scale and step size
          const double sPu = dev::shfl_d(sPul, q), etaPu = dev::shfl_d(etaPul, q), sPnu = dev::shfl_d(sPnul, q);
          const double pw = sPu * pv;
          const double update = etaPu * (dLu * (vsu * (a1u - pw * vsu)) + O.beta * pw);
          p = (pw - update) / sPnu;
        }
        Pl[(size_t)q * Kp + s] = p;  // (all row slots write the same value)
        Tl[(size_t)q * Kp + s] = vl[q] * (sP * p);
      }
      if (mine && ((which >> lane) & 1ull)) {  // the linear weight of the shared feature, the same way
        double wv = wu;
        if (M.fit_linear) {
          if constexpr (ADA) {
            const double gg = dLl * vsl;
            const double gw = gwu + gg, nw_ = nwu + gg * gg;
            gwl[lane] = gw;
            nwl[lane] = nw_;
            if (it != 1) {
              const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
              viol_acc += fabs(wv - nv);
              st_f64(M.w + jl[lane], nv);
              wv = nv;
            }
          } else {  // fit_linear.nim:41-47 with the writer's scale and step size
            const double wj = swul * wu;
            wv = (wj - etawul * (dLl * vsl + O.alpha * wj)) / (swul * (1 - etawul * O.alpha));
          }
        }
        wl[lane] = wv;
      }
      compiler_fence();
      return true;
    };
    if (fwdmask) {
      load_lin();
      if constexpr (!ADA) {  // the writers' scales and step sizes: functions of their step counters alone
        if (mine) {
          sPul = a.scales[2 * (a.seg0 + upl)];
          swul = a.scales[2 * (a.seg0 + upl) + 1];
          etaPul = dev::get_eta(O.sched, O.eta0, O.power, O.beta, (double)(a.it0 + upl));
          etawul = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, (double)(a.it0 + upl));
          sPnul = sPul * (1 - etaPul * O.beta);
        }
      }
      double a1f = 0.0, pvf = 0.0, gvf = 0.0, nvf = 0.0;
      {
        Spin sp;
        while (true) {  // the writers' dL (their recipes were posted before it could exist)
          bool ok = true;
          if (mine) ok = fw_load(rsrcl, tagl, dLl);
          if (!__all(okl)) load_lin();
          // every dL but the affine writer's is here, and so is that writer's recipe: S as a function of its dL -- whether or
          // not that dL exists already (the same arithmetic every run)
          if (aff_q >= 0) {
            if (__all(ok || lane == aff_q) && __all(okl) && __all(load_recipe(aff_q, a1f, pvf, gvf, nvf))) {
              affine = true;
              break;
            }
          } else if (__all(ok)) {
            break;
          }
          if (sp.wait(a.ctrl)) return;
        }
      }
      if (!affine) {
        {
          Spin sp;
          while (!__all(okl)) {
            if (sp.wait(a.ctrl)) return;
            load_lin();
          }
        }
        if (!form_near_rows(fwdmask)) return;
      } else {
        if constexpr (!ADA) {
          if (!form_near_rows(fwdmask & ~aff_bit)) return;  // (the earlier writers' rows: exact)
          // row'(dL) = (p - eta (dL g + beta p)) / s' = A + dL B  with the writer's scale, step size and per-factor sums
          const int q = aff_q;
          const double vsu = dev::shfl_d(vsl, q);
          const double sPu = dev::shfl_d(sPul, q), etaPu = dev::shfl_d(etaPul, q), sPnu = dev::shfl_d(sPnul, q);
          const double pw = sPu * pvf;
          aff_A = (pw - etaPu * (O.beta * pw)) / sPnu;
          aff_B = -(etaPu * (vsu * (a1f - pw * vsu))) / sPnu;
          have_q = q;  // (kept: the writer's forwarding area may be reused by the time this sample's second half runs)
          have_a1 = a1f;
          have_pv = pvf;
          Pl[(size_t)q * Kp + s] = aff_A;
          Tl[(size_t)q * Kp + s] = vl[q] * (sP * aff_A);
          if (lane == q) {  // the shared feature's linear weight the same way (fit_linear.nim:41-47): Aw + dL Bw
            double wv = wu;
            if (M.fit_linear) {
              const double wj = swul * wu, den = swul * (1 - etawul * O.alpha);
              wv = (wj - etawul * (O.alpha * wj)) / den;
              aff_Bw = -(etawul * vsl) / den;
            }
            wl[lane] = wv;
          }
          compiler_fence();
        }
      }
    }

    if (a.trace && lane == 0)  // dependencies resolved (x 16) + which way: 2 near rows, 4 waited for a writer's dL, 8 affine
      a.trace[u * 8 + 1] = wall_clock64() * 16 + ((affine ? 8 : 0) + ((fwdmask & ~aff_bit) ? 4 : 0) + (fwdmask ? 2 : 0));
    // ---- C. the per-factor sums over all entries in storage order (sgd.nim:160-170), their sum over the factors in
    // ascending order (:172-173); every row slot runs them, so every lane ends with its factor's sums ----
    double a1 = 0.0, a2 = 0.0;
    for (int qb = 0; qb < m; qb += 8) {
      double t_[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) t_[t] = Tl[(size_t)(qb + t < m ? qb + t : qb) * Kp + s];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bool ok = qb + t < m;
        a1 = ok ? a1 + t_[t] : a1;
        a2 = ok ? a2 + t_[t] * t_[t] : a2;
      }
    }
    const double kv = (a1 * a1 - a2) / 2;
    if (r == 0) red[s] = s < k ? kv : 0.0;
    compiler_fence();
    double tot = 0.0;
    for (int sb = 0; sb < k; sb += 8) {
      double r_[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) r_[t] = red[sb + t < k ? sb + t : sb];
#pragma unroll
      for (int t = 0; t < 8; ++t) tot = sb + t < k ? tot + r_[t] : tot;
    }
    // the mailbox: the terms of the linear part (past the row's end -0.0, which changes no sum), then the header
    const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
    double dL_own = 0.0, yh_own = 0.0;
    if (a.no_cond || a.one_term) {
      // no conductor: the chain of predictWithGrad (sgd.nim:193-201) right here, in every lane -- the constant intercept,
      // then the linear terms in storage order, then the interaction sum -- and {dL, yhat} posted as the conductor would.
      // One-term window: the same sum WITHOUT the intercept (from 0.0) is the sample's one mailbox term.
      double yh_ = a.no_cond ? b_const : 0.0;
      for (int eb = 0; eb < m; eb += 8) {
        double w_[8], v_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int e = eb + t < m ? eb + t : eb;
          w_[t] = wl[e];
          v_[t] = vl[e];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) yh_ = eb + t < m ? yh_ + (sw * w_[t]) * v_[t] : yh_;
      }
      yh_ += tot;
      if (a.no_cond) {
        yh_own = yh_;
        dL_own = dev::loss_grad(O.loss, O.loss_param, y, yh_);
        if (lane < kResWords) {
          const double v = lane < 2 ? dL_own : yh_own;
          const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
          st_u64(a.res + (size_t)(slot * a.np + par) * kResWords + lane, ((ull)(unsigned)(u + 1) << 32) | (ull)half);
        }
      } else if (!affine) {
        post_sum(a.fwd, a.np, W, slot, par, (unsigned)(u + 1), lane, yh_, y, h2);
      } else {
        // dS / d dL_writer: the kernel is multilinear in the rows, so the slope is  sum_s x sP B_s (a1_s - x sP A_s)  -- the
        // per-factor sums WITHOUT the shared row's own term -- plus the linear term's  sw Bw x
        const double xs = vl[aff_q];
        const double rest = a1 - xs * (sP * aff_A);
        double c1 = dev::wave_sum(r == 0 && s < k ? (xs * (sP * aff_B)) * rest : 0.0);
        c1 += dev::shfl_d((sw * aff_Bw) * xs, aff_q);
        post_sum(a.fwd, a.np, W, slot, par, (unsigned)(u + 1), lane, yh_, y, h2, c1, 0.0, 0.0,
                 (unsigned)(u - ((int64_t)__builtin_amdgcn_readlane(pq, aff_q) - a.seg0)));  // (the writer's distance: 1 ... near_r - 1 <= 127)
        // ... and now the exact row: the writer's dL, its update of the shared row, this sample's sums again with it
        {
          Spin sp;
          while (true) {
            bool ok = true;
            if (lane == aff_q) ok = fw_load(rsrcl, tagl, dLl);
            if (__all(ok)) break;
            if (sp.wait(a.ctrl)) return;
          }
        }
        if (!form_near_rows(aff_bit)) return;
        a1 = 0.0;
        a2 = 0.0;
        for (int qb = 0; qb < m; qb += 8) {
          double t_[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) t_[t] = Tl[(size_t)(qb + t < m ? qb + t : qb) * Kp + s];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const bool ok = qb + t < m;
            a1 = ok ? a1 + t_[t] : a1;
            a2 = ok ? a2 + t_[t] * t_[t] : a2;
          }
        }
      }
    } else {
      const int MC = a.FW - kWinHdr;
      for (int e = lane; e < a.FW; e += kWave) {
        double val;
        if (e < MC) val = e < m ? (sw * wl[e]) * vl[e] : -0.0;
        else if (e == MC) val = tot;
        else if (e == MC + 1) val = y;
        else if (e == MC + 2) val = h2;
        else val = (double)m;  // the number of chain terms
        st_u64(mb + e, mail_bits(val));
      }
    }
    // (after the mailbox: a successor needs them together with this sample's dL, which the conductor forms from the mailbox)
    if (hotmask) {
      const unsigned mytag = (unsigned)(u + 1);
      ull* fwm = fw_area(u);
      if (r == 0) fw_store(fwm + (size_t)s * 2, mytag, a1);
      for (ull mk = hotmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        if (r == 0) {
          fw_store(fw_row(fwm, 0, q), mytag, Pl[(size_t)q * Kp + s]);
          if constexpr (ADA) {
            fw_store(fw_row(fwm, 1, q), mytag, Gl[(size_t)q * Kp + s]);
            fw_store(fw_row(fwm, 2, q), mytag, Nl[(size_t)q * Kp + s]);
          }
        }
      }
      if ((hotmask >> lane) & 1ull) {
        fw_store(fw_lin(fwm, 0, lane), mytag, wl[lane]);
        if constexpr (ADA) {
          if (M.fit_linear) {
            fw_store(fw_lin(fwm, 1, lane), mytag, gwl[lane]);
            fw_store(fw_lin(fwm, 2, lane), mytag, nwl[lane]);
          }
        }
        fw_store(fw_lin(fwm, 3, lane), mytag, vl[lane]);
      }
    }

    if (a.trace && lane == 0) a.trace[u * 8 + 2] = wall_clock64();  // mailbox (and recipes) posted
    // ---- D. while the conductor works: the derivative (sgd.nim:176-188) and the step sizes ----
    for (int qb = 0; qb < m; qb += R * U) {
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int q = qb + t * R + r;
        const double val = vl[q];
        const double p = sP * Pl[(size_t)q * Kp + s];
        Tl[(size_t)q * Kp + s] = val * (a1 - p * val);
      }
    }
    double eta_w = 0.0, eta_P = 0.0, sPn = 1.0, swn = 1.0;
    if constexpr (!ADA) {
      eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      sPn = sP * (1 - eta_P * O.beta);
      swn = sw * (1 - eta_w * O.alpha);
    }

    // ---- E. {dL, yhat} from the conductor (tagged granules) ----
    double dL, yh;
    if (a.no_cond) {
      dL = dL_own;
      yh = yh_own;
    } else {
      Spin sp;
      double rd;
      while (true) {
        const bool ok = fw_load(rp + (size_t)(lane & 1) * 2, (unsigned)(u + 1), rd);
        if (__all(ok)) break;
        if (sp.wait(a.ctrl)) return;
      }
      dL = dev::shfl_d(rd, 0);
      yh = dev::shfl_d(rd, 1);
      // both mailboxes back to "empty" for their next use, two samples of this worker from now: these stores have
      // completed (vmcnt(0) below) before this worker posts its next sample, which the conductor consumes before it can
      // look at these words again (kWinDepth < W).  (One-term mailboxes carry tags: nothing to reset.)
      if (!a.one_term)
        for (int e = lane; e < a.FW; e += kWave) st_u64(mb + e, kWinSentinel);
    }
    if (lane == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    if (a.trace && lane == 0) a.trace[u * 8 + 3] = wall_clock64();  // dL received

    // ---- E2. a row formed from its writer's recipe must not be STORED before the writer's own store of it has landed (two
    // stores to one address from two CUs are not ordered by anything else: with a conductor the writer's store is several
    // microseconds ahead by construction, without one the two updates run nearly side by side -- found by the field-aware
    // no-conductor test: one stale row in a thousand samples).  The writer bumps its counter after its stores have drained. ----
    if (fwdmask) {
      Spin sp;
      bool first = true;
      while (true) {
        bool need = false;
        for (int q = lane; q < m; q += kWave) {
          const int64_t v2 = (int64_t)pl[q] - a.seg0;
          if (v2 >= 0 && ll[q] && cnt[v2 & (W - 1)] <= (unsigned)(v2 >> lgW)) need = true;
        }
        if (!__any(need)) break;
        if (!first && sp.wait(a.ctrl)) return;
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }
    // ---- F. update(): sgd.nim:205-243 / updateG(): adagrad.nim:113-134 ----
    for (int qb = 0; qb < m; qb += R * U) {
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int q = qb + t * R + r;
        const bool in = q < m;
        const size_t e = (size_t)jl[q] * Kp + s;
        const double d_ = Tl[(size_t)q * Kp + s];
        if constexpr (ADA) {
          const double grad = dL * d_;
          st_f64_at(in ? (ull)(O.G + e) : (ull)junk, Gl[(size_t)q * Kp + s] + grad);
          st_f64_at(in ? (ull)(O.N + e) : (ull)(junk + kWave), Nl[(size_t)q * Kp + s] + grad * grad);
        } else {
          const double p = sP * Pl[(size_t)q * Kp + s];
          const double update = eta_P * (dL * d_ + O.beta * p);
          viol_acc += in ? fabs(update) : 0.0;
          st_f64_at(in ? (ull)(M.P + e) : (ull)junk, (p - update) / sPn);
        }
      }
    }
    if (M.fit_linear) {
      for (int q = lane; q < m; q += kWave) {
        const int j = jl[q];
        if constexpr (ADA) {
          const double gg = dL * vl[q];
          st_f64(O.Gw + j, gwl[q] + gg);
          st_f64(O.Nw + j, nwl[q] + gg * gg);
        } else {
          const double wj = sw * wl[q];
          const double update = eta_w * (dL * vl[q] + O.alpha * wj);
          viol_acc += fabs(update);
          st_f64(M.w + j, (wj - update) / swn);
        }
      }
    }
    // ---- G. rows written: tell the waiters ----
    // (no conductor: the other workers' completion counters requested here ride along with this wait -- the run-ahead
    // check of the next sample then finds them in LDS instead of paying a round trip of its own)
    unsigned cr0_ = 0u, cr1_ = 0u, cr2_ = 0u, cr3_ = 0u;
    if (a.no_cond) {
      if (lane < W) cr0_ = ld_u32(a.completed + lane);
      if (kWave + lane < W) cr1_ = ld_u32(a.completed + kWave + lane);
      if (2 * kWave + lane < W) cr2_ = ld_u32(a.completed + 2 * kWave + lane);
      if (3 * kWave + lane < W) cr3_ = ld_u32(a.completed + 3 * kWave + lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.no_cond) {
      if (lane < W) cnt[lane] = cr0_;
      if (kWave + lane < W) cnt[kWave + lane] = cr1_;
      if (2 * kWave + lane < W) cnt[2 * kWave + lane] = cr2_;
      if (3 * kWave + lane < W) cnt[3 * kWave + lane] = cr3_;
    }
    if (lane == 0) st_u32(a.completed + slot, (unsigned)(u >> lgW) + 1u);
    if (a.trace && lane == 0) a.trace[u * 8 + 4] = wall_clock64();  // rows written
  }
  viol_acc = dev::wave_sum(viol_acc);
  if (lane == 0) {
    a.partial[2 * slot] = loss_acc;
    a.partial[2 * slot + 1] = viol_acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// worker for field-aware models (optimizer/sgd_ffm.nim:11-30, 33-106; adagrad_ffm.nim:11-66): the reference's step reads
// and updates ALL nFields rows of every feature of the sample, so a sample's rows are its "slots" c = q * F + f (entry q,
// field f) -- in the feature-major layout (common.h) the F rows of a feature are one contiguous run, and a slot is
// handled like a row of the general worker above.  The prediction is the intercept, then the linear terms in storage
// order, then ONE term per pair (q1, q2) with j_q1 < j_q2 in the order of the reference's double loop: all of them go to
// the conductor's chain as mailbox terms.  A slot's derivative is collected over its field's entries in the order that
// loop visits them (three passes, as in seq.hip).  Dependencies by counters only.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kFfmWaves = 4;  // wavefronts of a field-aware worker (its workgroup: 256 threads)
template <int OPT>
__device__ __forceinline__ void win_worker_ffm(const WinArgs& a, const int slot, double* lds) {
  // FOUR wavefronts per worker: a field-aware step handles nFields rows per entry (cfg4: 256 rows, the pairs' dot products,
  // the three-pass derivative) -- one wavefront took 45 us for it, and a dependent sample waits for most of a step.  The
  // slots / pairs are dealt to the wavefronts, the phases are separated by workgroup barriers; wavefront 0 alone talks to
  // the counters and writes the mailbox in order.
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  constexpr int U = 4, NW = kFfmWaves;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  const int Kp = M.Kp, lgK = a.lgKp, k = M.k, F = M.nb, mcap = a.m_cap;
  const int R = kWave >> lgK, r = lane >> lgK, s = lane & (Kp - 1);
  const int mcs = (mcap * F + R * U - 1) / (R * U) * (R * U);  // slots, padded to whole groups
  const int W = a.W, lgW = a.lgW;
  double* Pl = lds;                                   // [mcs][Kp] stored parameter values of the sample's slots
  double* Tl = Pl + (size_t)mcs * Kp;                 // [mcs][Kp] the slots' derivative
  double* Gl = Tl + (size_t)mcs * Kp;                 // AdaGrad: g_sum
  double* Nl = Gl + (ADA ? (size_t)mcs * Kp : 0);     // AdaGrad: g_norm
  double* pc = Nl + (ADA ? (size_t)mcs * Kp : 0);     // [mcap][mcap] the pairs' terms of the prediction
  double* bc = pc + (size_t)mcap * mcap;              // [4][64] per near entry: the writer's dL, scale, step size, next scale
  double* vsum = bc + 4 * kWave;                      // [NW] the wavefronts' viol
  double* vl = vsum + NW;                             // [mcap] values
  double* wl = vl + mcap;                             // [mcap] stored linear weights (AdaGrad: after update())
  double* gwl = wl + mcap;                            // AdaGrad: [mcap]
  double* nwl = gwl + (ADA ? mcap : 0);               // AdaGrad: [mcap]
  ull* mk_l = reinterpret_cast<ull*>(nwl + (ADA ? mcap : 0));  // [2] {near entries, hot entries} of the sample
  int* jl = reinterpret_cast<int*>(mk_l + 2);                  // [mcap] feature ids
  int* fl = jl + mcap;                                       // [mcap] fields
  int* pl = fl + mcap;                                       // [mcap] previous position with the same feature
  int* ql = pl + mcap;                                       // [mcap] the feature's entry index in that sample
  int* ll = ql + mcap;                                       // [mcap] 1: the entry's rows come from their writer's recipe
  int* fcnt = ll + mcap;                                     // [F] entries of the sample per field
  int* fent = fcnt + F;                                      // [F][mcap] ... which ones, ascending
  unsigned* cnt = reinterpret_cast<unsigned*>(fent + (size_t)F * mcap);  // [W] (wavefront 0's)
  if (wv == 0)
    for (int l = lane; l < W; l += kWave) cnt[l] = 0u;
  const double b_const = M.sc[SC_INTERCEPT];  // (no_cond: the intercept is not fitted and stays what it is)
  double loss_acc = 0.0, viol_acc = 0.0;
  const ull lt_mask = lane == 0 ? 0ull : (~0ull >> (kWave - lane));
  const int nb = F;
  // Recipe forwarding per slot, as in win_worker_fmx below: of a hot entry the writer posts, per field, the row as it used
  // it and the row's derivative, AdaGrad's state rows as loaded, and the linear weight.
  const int hot_cap = (int)((size_t)kFwVals * kWave * kWave / ((size_t)nb * 4 * Kp));
  const int kFxHot = hot_cap < kWave ? hot_cap : kWave;
  const bool fwd_on = kFxHot >= 1;
  auto fw_area = [&](int64_t u_) { return a.fw + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kFwSlot; };
  auto fw_slot = [&](ull* base, int q, int o, int v) { return base + kFwRows + ((size_t)((q * nb + o) * 4 + v) * Kp + s) * 2; };
  auto fw_lin = [&](ull* base, int v, int q) { return base + kFwLin + (size_t)(v * kWave + q) * 2; };
  auto res_of = [&](int64_t u_) { return a.res + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kResWords; };
  __syncthreads();

  for (int64_t u = slot; u < a.n_seg; u += W) {
    const int64_t pos = a.seg0 + u, pa = a.begin + pos;
    const int64_t i = a.perm ? a.perm[pa] : pa;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int nsl = m * F;
    const double y = dev::target_of(X.y[i], M.task);
    const int64_t it = a.it0 + u;
    const double itf = (double)it;
    const int par = (int)((u >> lgW) & (a.np - 1));
    ull* mb = a.fwd + (size_t)(slot * a.np + par) * a.FW;
    const ull* rp = a.res + (size_t)(slot * a.np + par) * kResWords;
    // ---- 0. the entries; which of them take the forwarding path (wavefront 0 decides for all) ----
    if (wv == 0) {
      if (a.no_cond && u >= (int64_t)a.thr * W) {  // the run-ahead bound of a window without a conductor (see win_worker)
        const unsigned c_ = (unsigned)(u >> lgW) - (unsigned)(a.thr - 1);
        Spin sp;
        bool first = true;
        while (true) {
          bool ok = true;
          for (int l = lane; l < W; l += kWave) ok = ok && cnt[l] >= (l < slot ? c_ : c_ - 1u);
          if (__all(ok)) break;
          if (!first && sp.wait(a.ctrl)) break;  // (aborting: every wait below ends the same way)
          first = false;
          for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
          compiler_fence();
        }
      }
      const bool e_in = lane < m;
      const int pq = e_in ? a.prev[q0 + lane] : -1;
      const int pqu = e_in ? (int)a.prevq[q0 + lane] : 0;
      const int nq = e_in ? a.next[q0 + lane] : -1;
      const int64_t v = (int64_t)pq - a.seg0;
      const bool pend = v >= 0 && cnt[v & (W - 1)] <= (unsigned)(v >> lgW);
      const bool near = fwd_on && pend && (pos - (int64_t)pq) < a.near_r && pqu < kFxHot;
      const ull fm_ = __ballot(near), hm_ = __ballot(fwd_on && lane < kFxHot && nq >= 0 && ((int64_t)nq - pos) < a.near_r);
      if (lane == 0) {
        mk_l[0] = fm_;
        mk_l[1] = hm_;
      }
      for (int q = lane; q < mcap; q += kWave) {
        const bool in = q < m;
        jl[q] = in ? X.indices[q0 + q] : 0;
        fl[q] = in ? X.fields[q0 + q] : 0;
        vl[q] = in ? X.data[q0 + q] : 0.0;
        pl[q] = in ? a.prev[q0 + q] : -1;
        ql[q] = in ? (int)a.prevq[q0 + q] : 0;
        ll[q] = (q < kWave && near) ? 1 : 0;
      }
      compiler_fence();
      // ---- A. every earlier sample of this launch that shares a feature has written its rows (far ones only) ----
      Spin sp;
      bool first = true;
      while (true) {
        bool need = false;
        for (int q = lane; q < m; q += kWave) {
          const int64_t v2 = (int64_t)pl[q] - a.seg0;
          if (v2 >= 0 && !ll[q] && cnt[v2 & (W - 1)] <= (unsigned)(v2 >> lgW)) need = true;
        }
        if (!__any(need)) break;
        if (!first && sp.wait(a.ctrl)) break;  // (aborting: every wait below ends the same way)
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }
    if (a.trace && tid == 0) a.trace[u * 8 + 0] = wall_clock64();  // sample taken up, far dependencies waited for
    __syncthreads();
    const ull fwdmask = mk_l[0], hotmask = mk_l[1];
    double* const junk = reinterpret_cast<double*>(a.fw + (size_t)a.np * W * kFwSlot) + (size_t)slot * 2 * kWave + (lane & (kWave - 1));

    // ---- B. all F rows of every feature -> LDS (AdaGrad: update() first, adagrad.nim:87-110): the slots dealt to the wavefronts ----
    double sP = 1.0, sw = 1.0;
    if constexpr (!ADA) {
      sP = a.scales[2 * pos];
      sw = a.scales[2 * pos + 1];
    }
    const double itp = (double)(it - 1);
    const double tmpP = O.eta0 * itp * O.beta;
    const double denw = itp * O.eta0 * O.alpha;
    auto slot_row = [&](int c) {  // where slot c's row starts (a slot past the end: slot 0)
      const int cc = c < nsl ? c : 0;
      const int q = cc / F, f = cc - q * F;
      return M.row(f, jl[q]) * (size_t)Kp + s;
    };
    for (int cb = wv * R * U; cb < nsl; cb += NW * R * U) {
      double v_[U], g_[ADA ? U : 1], n_[ADA ? U : 1];
      size_t e_[U];
#pragma unroll
      for (int t = 0; t < U; ++t) e_[t] = slot_row(cb + t * R + r);
#pragma unroll
      for (int t = 0; t < U; ++t) {
        v_[t] = ld_f64(M.P + e_[t]);
        if constexpr (ADA) {
          g_[t] = ld_f64(O.G + e_[t]);
          n_[t] = ld_f64(O.N + e_[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int c = cb + t * R + r;
        const bool in = c < nsl && !ll[c / F];  // (a forwarded entry's rows are done again in B2)
        double p = v_[t];
        if constexpr (ADA) {
          if (it != 1) {
            p = dev::adagrad_param(g_[t], n_[t], O.eta0, tmpP);
            viol_acc += in ? fabs(v_[t] - p) : 0.0;
            st_f64_at(in ? (ull)(M.P + e_[t]) : (ull)junk, p);
          }
          Gl[(size_t)c * Kp + s] = g_[t];
          Nl[(size_t)c * Kp + s] = n_[t];
        }
        Pl[(size_t)c * Kp + s] = p;
      }
    }
    if (wv == NW - 1) {  // the linear weights (the last wavefront has the fewest slots)
      for (int q = lane; q < m; q += kWave) {
        const int j = jl[q];
        double wvv = ld_f64(M.w + j);
        if constexpr (ADA) {
          if (M.fit_linear) {
            const double gw = ld_f64(O.Gw + j), nw_ = ld_f64(O.Nw + j);
            gwl[q] = gw;
            nwl[q] = nw_;
            if (it != 1 && !ll[q]) {
              const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
              viol_acc += fabs(wvv - nv);
              st_f64(M.w + j, nv);
              wvv = nv;
            }
          }
        }
        wl[q] = wvv;
      }
      // the sample's entries by field, ascending (sgd_ffm.nim:24-30 walks a field's entries in this order)
      for (int f = lane; f < F; f += kWave) {
        int c_ = 0;
        for (int q = 0; q < m; ++q)
          if (fl[q] == f) fent[(size_t)f * mcap + c_++] = q;
        fcnt[f] = c_;
      }
    }
    __syncthreads();

    // ---- B2. near dependencies: the writer's recipe + the conductor's dL for the WRITER's sample -> the rows as the writer
    // will (or did) write them.  Wavefront 0 polls what is per entry, all wavefronts form the rows (fields dealt out) ----
    if (fwdmask) {
      if (wv == 0) {
        const bool mine = (fwdmask >> lane) & 1ull;  // lane = entry
        const int64_t upl = (int64_t)pl[mine ? lane : 0] - a.seg0;
        const int pqu = ql[mine ? lane : 0];
        const unsigned tagl = (unsigned)(upl + 1);
        ull* srcl = fw_area(mine ? upl : 0);
        const ull* rsrcl = res_of(mine ? upl : 0);
        double wu = 0.0, gwu = 0.0, nwu = 0.0, vsl = 0.0, dLl = 0.0;
        bool okl = true, dead = false;
        auto load_lin = [&]() {
          okl = true;
          if (mine) {
            okl = fw_load(fw_lin(srcl, 0, pqu), tagl, wu);
            if (ADA && M.fit_linear) {
              okl = fw_load(fw_lin(srcl, 1, pqu), tagl, gwu) && okl;
              okl = fw_load(fw_lin(srcl, 2, pqu), tagl, nwu) && okl;
            }
            okl = fw_load(fw_lin(srcl, 3, pqu), tagl, vsl) && okl;
          }
        };
        load_lin();
        double sPul = 1.0, etaPul = 0.0, sPnul = 1.0, swul = 1.0, etawul = 0.0;
        if constexpr (!ADA) {
          if (mine) {
            sPul = a.scales[2 * (a.seg0 + upl)];
            swul = a.scales[2 * (a.seg0 + upl) + 1];
            etaPul = dev::get_eta(O.sched, O.eta0, O.power, O.beta, (double)(a.it0 + upl));
            etawul = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, (double)(a.it0 + upl));
            sPnul = sPul * (1 - etaPul * O.beta);
          }
        }
        {
          Spin sp;
          while (true) {
            bool ok = true;
            if (mine) ok = fw_load(rsrcl, tagl, dLl);
            if (!__all(okl)) load_lin();
            if (__all(ok) && __all(okl)) break;
            if (sp.wait(a.ctrl)) {
              dead = true;
              break;
            }
          }
        }
        if (mine && !dead) {
          bc[lane] = dLl;
          bc[kWave + lane] = sPul;
          bc[2 * kWave + lane] = etaPul;
          bc[3 * kWave + lane] = sPnul;
          double wvv = wu;  // the linear weight of the shared feature, the same way
          if (M.fit_linear) {
            if constexpr (ADA) {
              const double gg = dLl * vsl;
              const double gw = gwu + gg, nw_ = nwu + gg * gg;
              gwl[lane] = gw;
              nwl[lane] = nw_;
              if (it != 1) {
                const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
                viol_acc += fabs(wvv - nv);
                st_f64(M.w + jl[lane], nv);
                wvv = nv;
              }
            } else {
              const double wj = swul * wu;
              wvv = (wj - etawul * (dLl * vsl + O.alpha * wj)) / (swul * (1 - etawul * O.alpha));
            }
          }
          wl[lane] = wvv;
        }
      }
      __syncthreads();
      for (ull mk = fwdmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        const int64_t up = (int64_t)pl[q] - a.seg0;
        const unsigned tag = (unsigned)(up + 1);
        ull* src = fw_area(up);
        const int qu = ql[q];
        const double dLu = bc[q], sPu = bc[kWave + q], etaPu = bc[2 * kWave + q], sPnu = bc[3 * kWave + q];
        for (int o = wv; o < nb; o += NW) {
          double pv, dv, gv = 0.0, nv = 0.0;
          {
            Spin sp;
            while (true) {
              bool ok = fw_load(fw_slot(src, qu, o, 0), tag, pv);
              ok = fw_load(fw_slot(src, qu, o, 1), tag, dv) && ok;
              if constexpr (ADA) {
                ok = fw_load(fw_slot(src, qu, o, 2), tag, gv) && ok;
                ok = fw_load(fw_slot(src, qu, o, 3), tag, nv) && ok;
              }
              if (__all(ok)) break;
              if (sp.wait(a.ctrl)) break;
            }
          }
          const size_t c = (size_t)q * nb + o;
          const size_t e = M.row(o, jl[q]) * (size_t)Kp + s;
          double p;
          if constexpr (ADA) {
            const double grad = dLu * dv;
            const double g = gv + grad, n = nv + grad * grad;
            p = pv;
            if (it != 1) {
              p = dev::adagrad_param(g, n, O.eta0, tmpP);
              if (r == 0) {
                viol_acc += fabs(pv - p);
                st_f64(M.P + e, p);
              }
            }
            Gl[c * Kp + s] = g;
            Nl[c * Kp + s] = n;
          } else {
            const double pw = sPu * pv;
            const double update = etaPu * (dLu * dv + O.beta * pw);
            p = (pw - update) / sPnu;
          }
          Pl[c * Kp + s] = p;
        }
      }
      __syncthreads();
    }

    // ---- C. predictWithGrad, sgd_ffm.nim:11-30.  The pairs (q1, q2), j_q1 < j_q2, q1 outer / q2 inner, add
    //   result += (P[f2][j1] . P[f1][j2]) x1 x2,   dA[f2][q1] += x1 x2 P[f1][j2],   dA[f1][q2] += x1 x2 P[f2][j1]
    // so slot (q, f) collects, over the entries q' of field f, x_q x_q' P[field(q)][j_q'] in the order q' < q with
    // j_q' < j_q (visited as (q', q)), then all q' with j_q' > j_q (visited as (q, q')), then q' > q with j_q' < j_q ----
    for (int cb = wv * R; cb < nsl; cb += NW * R) {
      const int c = cb + r;
      if (c < nsl) {
        const int q = c / F, f = c - q * F;
        const int j = jl[q], fq = fl[q], nf = fcnt[f];
        const double xq = vl[q];
        const int* ent = fent + (size_t)f * mcap;
        double acc = 0.0;
        for (int ph = 0; ph < 3; ++ph)
          for (int t = 0; t < nf; ++t) {
            const int q2 = ent[t];
            const int j2 = jl[q2];
            const bool take = ph == 0 ? (q2 < q && j2 < j) : ph == 1 ? (j2 > j) : (q2 > q && j2 < j);
            if (take) {
              const double v12 = (j2 < j) ? vl[q2] * xq : xq * vl[q2];
              acc += v12 * (sP * Pl[((size_t)q2 * F + fq) * Kp + s]);
            }
          }
        Tl[(size_t)c * Kp + s] = acc;
      }
    }
    // the pairs' terms: thread = ordered pair (q1, q2); per pair ONE dot product over the factors, ascending s, times x1, times x2
    for (int p = tid; p < m * m; p += NW * kWave) {
      const int q1 = p / m, q2 = p - q1 * m;
      double term = 0.0;
      if (jl[q1] < jl[q2]) {
        const double* pa_ = Pl + ((size_t)q1 * F + fl[q2]) * Kp;  // P[f2][j1]
        const double* pb_ = Pl + ((size_t)q2 * F + fl[q1]) * Kp;  // P[f1][j2]
        double tmp = 0.0;
        for (int tb = 0; tb < k; tb += 8) {
          double a_[8], b_[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int tt = tb + e < k ? tb + e : tb;
            a_[e] = pa_[tt];
            b_[e] = pb_[tt];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp = tb + e < k ? tmp + (sP * a_[e]) * (sP * b_[e]) : tmp;
        }
        term = tmp * vl[q1] * vl[q2];
      }
      pc[p] = term;
    }
    __syncthreads();
    const int MC = a.FW - kWinHdr;
    if (wv == 0 && (a.no_cond || a.one_term)) {
      // no conductor: predictWithGrad's chain (sgd_ffm.nim:13-27) in this wavefront -- the constant intercept, the linear terms
      // in storage order, then ONE term per pair j_q1 < j_q2 in the order of the reference's double loop (q1 outer, q2 inner) --
      // dloss, and {dL, yhat} posted as the conductor would.  One-term window: the same sum from 0.0 is the mailbox term.
      double yh_ = a.no_cond ? b_const : 0.0;
      for (int eb = 0; eb < m; eb += 8) {
        double w_[8], v_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int e = eb + t < m ? eb + t : eb;
          w_[t] = wl[e];
          v_[t] = vl[e];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) yh_ = eb + t < m ? yh_ + (sw * w_[t]) * v_[t] : yh_;
      }
      for (int q1 = 0; q1 < m; ++q1) {
        const int j1 = jl[q1];
        for (int qb = 0; qb < m; qb += 8) {
          double t_[8];
          int j_[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int q2 = qb + t < m ? qb + t : qb;
            t_[t] = pc[q1 * m + q2];
            j_[t] = jl[q2];
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) yh_ = (qb + t < m && j1 < j_[t]) ? yh_ + t_[t] : yh_;
        }
      }
      if (a.no_cond) {
        const double dL_ = dev::loss_grad(O.loss, O.loss_param, y, yh_);
        if (lane < kResWords) {
          const double v = lane < 2 ? dL_ : yh_;
          const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
          st_u64(a.res + (size_t)(slot * a.np + par) * kResWords + lane, ((ull)(unsigned)(u + 1) << 32) | (ull)half);
        }
      } else {
        const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
        post_sum(a.fwd, a.np, W, slot, par, (unsigned)(u + 1), lane, yh_, y, h2);
      }
    } else if (wv == 0) {  // the mailbox: linear terms, then the valid pairs' terms in the order the double loop visits them
      int n_pairs = 0;
      for (int pb = 0; pb < m * m; pb += kWave) {
        const int p = pb + lane;
        bool valid = false;
        if (p < m * m) {
          const int q1 = p / m, q2 = p - q1 * m;
          valid = jl[q1] < jl[q2];
        }
        const ull mask = __ballot(valid);
        if (valid) st_u64(mb + m + n_pairs + __popcll(mask & lt_mask), mail_bits(pc[p]));
        n_pairs += __popcll(mask);
      }
      const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
      for (int e = lane; e < a.FW; e += kWave) {
        double val;
        bool put = true;
        if (e < m) val = (sw * wl[e]) * vl[e];
        else if (e < m + n_pairs) put = false;  // a pair's term, stored above
        else if (e < MC) val = -0.0;            // (changes no sum)
        else if (e == MC) val = -0.0;           // no separate interaction sum: the pairs are chain terms
        else if (e == MC + 1) val = y;
        else if (e == MC + 2) val = h2;
        else val = (double)(m + n_pairs);       // the number of chain terms
        if (put) st_u64(mb + e, mail_bits(val));
      }
    }
    // (after the mailbox: a successor needs them together with this sample's dL, which the conductor forms from the mailbox)
    if (hotmask) {
      const unsigned mytag = (unsigned)(u + 1);
      ull* fwm = fw_area(u);
      for (ull mk = hotmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        if (r == 0) {
          for (int o = wv; o < nb; o += NW) {
            const size_t c = (size_t)q * nb + o;
            fw_store(fw_slot(fwm, q, o, 0), mytag, Pl[c * Kp + s]);
            fw_store(fw_slot(fwm, q, o, 1), mytag, Tl[c * Kp + s]);
            if constexpr (ADA) {
              fw_store(fw_slot(fwm, q, o, 2), mytag, Gl[c * Kp + s]);
              fw_store(fw_slot(fwm, q, o, 3), mytag, Nl[c * Kp + s]);
            }
          }
        }
      }
      if (wv == NW - 1 && ((hotmask >> lane) & 1ull)) {
        fw_store(fw_lin(fwm, 0, lane), mytag, wl[lane]);
        if constexpr (ADA) {
          if (M.fit_linear) {
            fw_store(fw_lin(fwm, 1, lane), mytag, gwl[lane]);
            fw_store(fw_lin(fwm, 2, lane), mytag, nwl[lane]);
          }
        }
        fw_store(fw_lin(fwm, 3, lane), mytag, vl[lane]);
      }
    }

    if (a.trace && tid == 0)  // mailbox (and recipes) posted; [1]: the same stamp x 16 + which way (2 near rows, 8 affine)
      a.trace[u * 8 + 2] = wall_clock64(), a.trace[u * 8 + 1] = wall_clock64() * 16 + (fwdmask ? 2 : 0);
    // ---- D. the step sizes while the conductor works ----
    double eta_w = 0.0, eta_P = 0.0, sPn = 1.0, swn = 1.0;
    if constexpr (!ADA) {
      eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      sPn = sP * (1 - eta_P * O.beta);
      swn = sw * (1 - eta_w * O.alpha);
    }

    // ---- E. {dL, yhat} from the conductor (tagged granules): every wavefront takes them itself ----
    double dL, yh;
    bool dead = false;
    {
      Spin sp;
      double rd;
      while (true) {
        const bool ok = fw_load(rp + (size_t)(lane & 1) * 2, (unsigned)(u + 1), rd);
        if (__all(ok)) break;
        if (sp.wait(a.ctrl)) {
          dead = true;
          break;
        }
      }
      dL = dev::shfl_d(rd, 0);
      yh = dev::shfl_d(rd, 1);
    }
    if (dead) break;  // (the launch is being aborted: every wavefront finds the abort word set)
    if (a.trace && tid == 0) a.trace[u * 8 + 3] = wall_clock64();  // dL received
    if (wv == 0) {
      if (!a.no_cond && !a.one_term)
        for (int e = lane; e < a.FW; e += kWave) st_u64(mb + e, kWinSentinel);  // (as in the general worker)
      if (lane == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    }

    // ---- E2. rows formed from a near writer's recipe are stored only after that writer's own stores have landed (see win_worker) ----
    if (fwdmask) {
      if (wv == 0) {
        Spin sp;
        bool first = true;
        while (true) {
          bool need = false;
          for (int q = lane; q < m; q += kWave) {
            const int64_t v2 = (int64_t)pl[q] - a.seg0;
            if (v2 >= 0 && ll[q] && cnt[v2 & (W - 1)] <= (unsigned)(v2 >> lgW)) need = true;
          }
          if (!__any(need)) break;
          if (!first && sp.wait(a.ctrl)) break;  // (aborting)
          first = false;
          for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
          compiler_fence();
        }
      }
      __syncthreads();
    }
    // ---- F. update() / updateG() over all slots (the shared loops of sgd.nim:205-243, adagrad.nim:113-134) ----
    for (int cb = wv * R * U; cb < nsl; cb += NW * R * U) {
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int c = cb + t * R + r;
        const bool in = c < nsl;
        const size_t e = slot_row(c);
        const double d_ = Tl[(size_t)c * Kp + s];
        if constexpr (ADA) {
          const double grad = dL * d_;
          st_f64_at(in ? (ull)(O.G + e) : (ull)junk, Gl[(size_t)c * Kp + s] + grad);
          st_f64_at(in ? (ull)(O.N + e) : (ull)(junk + kWave), Nl[(size_t)c * Kp + s] + grad * grad);
        } else {
          const double p = sP * Pl[(size_t)c * Kp + s];
          const double update = eta_P * (dL * d_ + O.beta * p);
          viol_acc += in ? fabs(update) : 0.0;
          st_f64_at(in ? (ull)(M.P + e) : (ull)junk, (p - update) / sPn);
        }
      }
    }
    if (M.fit_linear && wv == NW - 1) {
      for (int q = lane; q < m; q += kWave) {
        const int j = jl[q];
        if constexpr (ADA) {
          const double gg = dL * vl[q];
          st_f64(O.Gw + j, gwl[q] + gg);
          st_f64(O.Nw + j, nwl[q] + gg * gg);
        } else {
          const double wj = sw * wl[q];
          const double update = eta_w * (dL * vl[q] + O.alpha * wj);
          viol_acc += fabs(update);
          st_f64(M.w + j, (wj - update) / swn);
        }
      }
    }
    // ---- G. rows written (every wavefront's stores): tell the waiters ----
    unsigned cr_[4] = {0u, 0u, 0u, 0u};
    if (a.no_cond && wv == 0) {  // (the other workers' counters for the next sample's run-ahead check ride along with the drain)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * kWave + lane < W) cr_[t] = ld_u32(a.completed + t * kWave + lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.no_cond && wv == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * kWave + lane < W) cnt[t * kWave + lane] = cr_[t];
    }
    __syncthreads();
    if (tid == 0) st_u32(a.completed + slot, (unsigned)(u >> lgW) + 1u);
    if (a.trace && tid == 0) a.trace[u * 8 + 4] = wall_clock64();  // rows written
  }
  viol_acc = dev::wave_sum(viol_acc);
  if (lane == 0) vsum[wv] = viol_acc;
  __syncthreads();
  if (tid == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < NW; ++w_) v += vsum[w_];
    a.partial[2 * slot] = loss_acc;
    a.partial[2 * slot + 1] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// worker for FMs of any degree / several orders (fitLower = explicit: order o has degree `degree - o`; kernels of
// optimizer/sgd.nim:146-188): a sample's rows are its slots c = q * nb + o (entry q, order o), handled like the rows of the
// general worker.  Per order: the ANOVA recursion over the entries in storage order (every lane runs it for its factor),
// the derivative of the order's slots, the sum over the factors in ascending order -- ONE chain term per order, behind the
// linear terms (predictWithGrad adds the orders' kernels to the prediction one after the other, sgd.nim:198-201).
// Dependencies by counters only.  No dummy features (fitLower = augment: every sample would depend on its predecessor).
// ------------------------------------------------------------------------------------------------------------------
template <int OPT>
__device__ __forceinline__ void win_worker_fmx(const WinArgs& a, const int slot, double* lds) {
  // Four wavefronts per worker, as in win_worker_ffm: the slots are dealt to the wavefronts, the phases are separated by
  // workgroup barriers; wavefront 0 alone talks to the counters and writes the mailbox.
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  constexpr int U = 4, NW = kFfmWaves, DG = dev::kMaxDeg;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid >> 6;
  const int Kp = M.Kp, lgK = a.lgKp, k = M.k, F = M.nb, mcap = a.m_cap;
  const int R = kWave >> lgK, r = lane >> lgK, s = lane & (Kp - 1);
  const int mcs = (mcap * F + R * U - 1) / (R * U) * (R * U);  // slots, padded to whole groups
  const int W = a.W, lgW = a.lgW;
  double* Pl = lds;                                   // [mcs][Kp] stored parameter values of the sample's slots
  double* Tl = Pl + (size_t)mcs * Kp;                 // [mcs][Kp] the slots' derivative
  double* Gl = Tl + (size_t)mcs * Kp;                 // AdaGrad: g_sum
  double* Nl = Gl + (ADA ? (size_t)mcs * Kp : 0);     // AdaGrad: g_norm
  double* red = Nl + (ADA ? (size_t)mcs * Kp : 0);    // [64] an order's kernel per factor
  double* bc = red + kWave;              // [4][64] per near entry: the writer's dL, scale, step size, next scale
  double* affB = bc + 4 * kWave;                      // [DG][64] the affine entry's rows: slope in its writer's dL, per order
  double* affP = affB + DG * kWave;                   // [DG][64] ... the rows as the writer used them, and
  double* affD = affP + DG * kWave;                   // [DG][64] ... their derivatives (the recipe, kept for the exact rows)
  double* vsum = affD + DG * kWave;                   // [NW] the wavefronts' viol
  double* vl = vsum + NW;                             // [mcap] values
  double* wl = vl + mcap;                             // [mcap] stored linear weights (AdaGrad: after update())
  double* gwl = wl + mcap;                            // AdaGrad: [mcap]
  double* nwl = gwl + (ADA ? mcap : 0);               // AdaGrad: [mcap]
  ull* mk_l = reinterpret_cast<ull*>(nwl + (ADA ? mcap : 0));  // [3] {near entries, hot entries, affine entry + 1} of the sample
  int* jl = reinterpret_cast<int*>(mk_l + 3);                  // [mcap] feature ids
  int* pl = jl + mcap;                                       // [mcap] previous position with the same feature
  int* ql = pl + mcap;                                       // [mcap] the feature's entry index in that sample
  int* ll = ql + mcap;                                       // [mcap] 1: the entry's rows come from their writer's recipe
  unsigned* cnt = reinterpret_cast<unsigned*>(ll + mcap);    // [W] (wavefront 0's)
  if (wv == 0)
    for (int l = lane; l < W; l += kWave) cnt[l] = 0u;
  double loss_acc = 0.0, viol_acc = 0.0;
  const double b_const = M.sc[SC_INTERCEPT];  // (no_cond: the intercept is not fitted and stays what it is)
  const int nb = F;
  // Recipe forwarding as in the general worker, per slot: of a hot entry (its feature is asked for again within W
  // positions; among the sample's first kFxHot entries -- as many as the area holds) the writer posts, per order, the row
  // as it used it and the row's DERIVATIVE (known before the conductor answers), AdaGrad's state rows as loaded, and the
  // linear weight; the successor forms the rows the writer will write from them and the conductor's dL for the writer's sample.
  const int hot_cap = (int)((size_t)kFwVals * kWave * kWave / ((size_t)nb * 4 * Kp));
  const int kFxHot = hot_cap < kWave ? hot_cap : kWave;
  const bool fwd_on = kFxHot >= 1;
  auto fw_area = [&](int64_t u_) { return a.fw + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kFwSlot; };
  auto fw_slot = [&](ull* base, int q, int o, int v) { return base + kFwRows + ((size_t)((q * nb + o) * 4 + v) * Kp + s) * 2; };
  auto fw_lin = [&](ull* base, int v, int q) { return base + kFwLin + (size_t)(v * kWave + q) * 2; };
  auto res_of = [&](int64_t u_) { return a.res + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kResWords; };
  __syncthreads();

  for (int64_t u = slot; u < a.n_seg; u += W) {
    const int64_t pos = a.seg0 + u, pa = a.begin + pos;
    const int64_t i = a.perm ? a.perm[pa] : pa;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const int nsl = m * F;
    const double y = dev::target_of(X.y[i], M.task);
    const int64_t it = a.it0 + u;
    const double itf = (double)it;
    const int par = (int)((u >> lgW) & (a.np - 1));
    ull* mb = a.fwd + (size_t)(slot * a.np + par) * a.FW;
    const ull* rp = a.res + (size_t)(slot * a.np + par) * kResWords;
    // ---- 0. the entries; which of them take the forwarding path (wavefront 0 decides for all) ----
    if (wv == 0) {
      if (a.no_cond && u >= (int64_t)a.thr * W) {  // the run-ahead bound of a window without a conductor (see win_worker)
        const unsigned c_ = (unsigned)(u >> lgW) - (unsigned)(a.thr - 1);
        Spin sp;
        bool first = true;
        while (true) {
          bool ok = true;
          for (int l = lane; l < W; l += kWave) ok = ok && cnt[l] >= (l < slot ? c_ : c_ - 1u);
          if (__all(ok)) break;
          if (!first && sp.wait(a.ctrl)) break;  // (aborting: every wait below ends the same way)
          first = false;
          for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
          compiler_fence();
        }
      }
      const bool e_in = lane < m;
      const int pq = e_in ? a.prev[q0 + lane] : -1;
      const int pqu = e_in ? (int)a.prevq[q0 + lane] : 0;
      const int nq = e_in ? a.next[q0 + lane] : -1;
      const int64_t v = (int64_t)pq - a.seg0;
      const bool pend = v >= 0 && cnt[v & (W - 1)] <= (unsigned)(v >> lgW);
      // One-term window, SGD: the rows of the entry shared with the MOST RECENT earlier sample inside the window are treated as
      // AFFINE in that sample's dL (post_sum; see win_worker) -- chosen by positions alone, so that the arithmetic is the same
      // from run to run; that writer must share exactly one feature with this sample
      int aff_q = -1;
      if constexpr (!ADA) {
        if (a.one_term && fwd_on) {
          const bool cand = v >= 0 && (pos - (int64_t)pq) < a.near_r && pqu < kFxHot;
          int latest = cand ? (int)v : -1;
#pragma unroll
          for (int sh = 1; sh < kWave; sh <<= 1) {
            const int o_ = __shfl_xor(latest, sh, kWave);
            latest = o_ > latest ? o_ : latest;
          }
          const ull who = __ballot(cand && (int)v == latest);
          if (who != 0ull && (who & (who - 1)) == 0ull) aff_q = __builtin_ctzll(who);
        }
      }
      const bool near = (fwd_on && pend && (pos - (int64_t)pq) < a.near_r && pqu < kFxHot) || lane == aff_q;
      const ull fm_ = __ballot(near), hm_ = __ballot(fwd_on && lane < kFxHot && nq >= 0 && ((int64_t)nq - pos) < a.near_r);
      if (lane == 0) {
        mk_l[0] = fm_;
        mk_l[1] = hm_;
        mk_l[2] = (ull)(aff_q + 1);
      }
      for (int q = lane; q < mcap; q += kWave) {
        const bool in = q < m;
        jl[q] = in ? X.indices[q0 + q] : 0;
        vl[q] = in ? X.data[q0 + q] : 0.0;
        pl[q] = in ? a.prev[q0 + q] : -1;
        ql[q] = in ? (int)a.prevq[q0 + q] : 0;
        ll[q] = (q < kWave && near) ? 1 : 0;
      }
      compiler_fence();
      // ---- A. every earlier sample of this launch that shares a feature has written its rows (far ones only) ----
      Spin sp;
      bool first = true;
      while (true) {
        bool need = false;
        for (int q = lane; q < m; q += kWave) {
          const int64_t v2 = (int64_t)pl[q] - a.seg0;
          if (v2 >= 0 && !ll[q] && cnt[v2 & (W - 1)] <= (unsigned)(v2 >> lgW)) need = true;
        }
        if (!__any(need)) break;
        if (!first && sp.wait(a.ctrl)) break;  // (aborting: every wait below ends the same way)
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }
    if (a.trace && tid == 0) a.trace[u * 8 + 0] = wall_clock64();  // sample taken up, far dependencies waited for
    __syncthreads();
    const ull fwdmask = mk_l[0], hotmask = mk_l[1];
    const int aff_q = (int)mk_l[2] - 1;  // (uniform over the workgroup) the entry whose rows are posted as affine in its writer's dL
    const bool affine = aff_q >= 0;
    double* const junk = reinterpret_cast<double*>(a.fw + (size_t)a.np * W * kFwSlot) + (size_t)slot * 2 * kWave + (lane & (kWave - 1));

    // ---- B. the rows of every (entry, order) -> LDS (AdaGrad: update() first, adagrad.nim:87-110): the slots dealt to the wavefronts ----
    double sP = 1.0, sw = 1.0;
    if constexpr (!ADA) {
      sP = a.scales[2 * pos];
      sw = a.scales[2 * pos + 1];
    }
    const double itp = (double)(it - 1);
    const double tmpP = O.eta0 * itp * O.beta;
    const double denw = itp * O.eta0 * O.alpha;
    auto slot_row = [&](int c) {  // where slot c's row starts (a slot past the end: slot 0)
      const int cc = c < nsl ? c : 0;
      const int q = cc / F, f = cc - q * F;
      return M.row(f, jl[q]) * (size_t)Kp + s;
    };
    for (int cb = wv * R * U; cb < nsl; cb += NW * R * U) {
      double v_[U], g_[ADA ? U : 1], n_[ADA ? U : 1];
      size_t e_[U];
#pragma unroll
      for (int t = 0; t < U; ++t) e_[t] = slot_row(cb + t * R + r);
#pragma unroll
      for (int t = 0; t < U; ++t) {
        v_[t] = ld_f64(M.P + e_[t]);
        if constexpr (ADA) {
          g_[t] = ld_f64(O.G + e_[t]);
          n_[t] = ld_f64(O.N + e_[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int c = cb + t * R + r;
        const bool in = c < nsl && !ll[c / F];  // (a forwarded entry's rows are done again in B2)
        double p = v_[t];
        if constexpr (ADA) {
          if (it != 1) {
            p = dev::adagrad_param(g_[t], n_[t], O.eta0, tmpP);
            viol_acc += in ? fabs(v_[t] - p) : 0.0;
            st_f64_at(in ? (ull)(M.P + e_[t]) : (ull)junk, p);
          }
          Gl[(size_t)c * Kp + s] = g_[t];
          Nl[(size_t)c * Kp + s] = n_[t];
        }
        Pl[(size_t)c * Kp + s] = p;
      }
    }
    if (wv == NW - 1) {  // the linear weights (the last wavefront has the fewest slots)
      for (int q = lane; q < m; q += kWave) {
        const int j = jl[q];
        double wvv = ld_f64(M.w + j);
        if constexpr (ADA) {
          if (M.fit_linear) {
            const double gw = ld_f64(O.Gw + j), nw_ = ld_f64(O.Nw + j);
            gwl[q] = gw;
            nwl[q] = nw_;
            if (it != 1 && !ll[q]) {
              const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
              viol_acc += fabs(wvv - nv);
              st_f64(M.w + j, nv);
              wvv = nv;
            }
          }
        }
        wl[q] = wvv;
      }
    }
    __syncthreads();

    // ---- B2. near dependencies: the writer's recipe + the conductor's dL for the WRITER's sample -> the rows as the writer
    // will (or did) write them.  Wavefront 0 polls what is per entry, all wavefronts form the rows (fields dealt out) ----
    // (wavefront 0, lane = entry: the shared feature's linear weight as its writer used it, the entry's value there, the writer's
    // scale and step size -- kept for the affine entry's second half)
    double wu = 0.0, gwu = 0.0, nwu = 0.0, vsl = 0.0, swul = 1.0, etawul = 0.0, aff_Bw = 0.0;
    if (fwdmask) {
      if (wv == 0) {
        const bool mine = (fwdmask >> lane) & 1ull;  // lane = entry
        const int64_t upl = (int64_t)pl[mine ? lane : 0] - a.seg0;
        const int pqu = ql[mine ? lane : 0];
        const unsigned tagl = (unsigned)(upl + 1);
        ull* srcl = fw_area(mine ? upl : 0);
        const ull* rsrcl = res_of(mine ? upl : 0);
        double dLl = 0.0;
        bool okl = true, dead = false;
        auto load_lin = [&]() {
          okl = true;
          if (mine) {
            okl = fw_load(fw_lin(srcl, 0, pqu), tagl, wu);
            if (ADA && M.fit_linear) {
              okl = fw_load(fw_lin(srcl, 1, pqu), tagl, gwu) && okl;
              okl = fw_load(fw_lin(srcl, 2, pqu), tagl, nwu) && okl;
            }
            okl = fw_load(fw_lin(srcl, 3, pqu), tagl, vsl) && okl;
          }
        };
        load_lin();
        double sPul = 1.0, etaPul = 0.0, sPnul = 1.0;
        if constexpr (!ADA) {
          if (mine) {
            sPul = a.scales[2 * (a.seg0 + upl)];
            swul = a.scales[2 * (a.seg0 + upl) + 1];
            etaPul = dev::get_eta(O.sched, O.eta0, O.power, O.beta, (double)(a.it0 + upl));
            etawul = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, (double)(a.it0 + upl));
            sPnul = sPul * (1 - etaPul * O.beta);
          }
        }
        {
          Spin sp;
          while (true) {
            bool ok = true;
            if (mine && lane != aff_q) ok = fw_load(rsrcl, tagl, dLl);  // (the affine writer's dL is not waited for here)
            if (!__all(okl)) load_lin();
            if (__all(ok) && __all(okl)) break;
            if (sp.wait(a.ctrl)) {
              dead = true;
              break;
            }
          }
        }
        if (mine && !dead) {
          bc[lane] = dLl;
          bc[kWave + lane] = sPul;
          bc[2 * kWave + lane] = etaPul;
          bc[3 * kWave + lane] = sPnul;
          double wvv = wu;  // the linear weight of the shared feature, the same way
          if (M.fit_linear) {
            if constexpr (ADA) {
              const double gg = dLl * vsl;
              const double gw = gwu + gg, nw_ = nwu + gg * gg;
              gwl[lane] = gw;
              nwl[lane] = nw_;
              if (it != 1) {
                const double nv = -O.eta0 * gw / (denw + sqrt(nw_));
                viol_acc += fabs(wvv - nv);
                st_f64(M.w + jl[lane], nv);
                wvv = nv;
              }
            } else if (lane != aff_q) {
              const double wj = swul * wu;
              wvv = (wj - etawul * (dLl * vsl + O.alpha * wj)) / (swul * (1 - etawul * O.alpha));
            } else {  // the affine entry's linear weight (fit_linear.nim:41-47): Aw + dL Bw
              const double wj = swul * wu, den = swul * (1 - etawul * O.alpha);
              wvv = (wj - etawul * (O.alpha * wj)) / den;
              aff_Bw = -(etawul * vsl) / den;
            }
          }
          wl[lane] = wvv;
        }
      }
      __syncthreads();
      for (ull mk = fwdmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        const int64_t up = (int64_t)pl[q] - a.seg0;
        const unsigned tag = (unsigned)(up + 1);
        ull* src = fw_area(up);
        const int qu = ql[q];
        const double dLu = bc[q], sPu = bc[kWave + q], etaPu = bc[2 * kWave + q], sPnu = bc[3 * kWave + q];
        for (int o = wv; o < nb; o += NW) {
          double pv, dv, gv = 0.0, nv = 0.0;
          {
            Spin sp;
            while (true) {
              bool ok = fw_load(fw_slot(src, qu, o, 0), tag, pv);
              ok = fw_load(fw_slot(src, qu, o, 1), tag, dv) && ok;
              if constexpr (ADA) {
                ok = fw_load(fw_slot(src, qu, o, 2), tag, gv) && ok;
                ok = fw_load(fw_slot(src, qu, o, 3), tag, nv) && ok;
              }
              if (__all(ok)) break;
              if (sp.wait(a.ctrl)) break;
            }
          }
          const size_t c = (size_t)q * nb + o;
          const size_t e = M.row(o, jl[q]) * (size_t)Kp + s;
          double p;
          if constexpr (ADA) {
            const double grad = dLu * dv;
            const double g = gv + grad, n = nv + grad * grad;
            p = pv;
            if (it != 1) {
              p = dev::adagrad_param(g, n, O.eta0, tmpP);
              if (r == 0) {
                viol_acc += fabs(pv - p);
                st_f64(M.P + e, p);
              }
            }
            Gl[c * Kp + s] = g;
            Nl[c * Kp + s] = n;
          } else if (q != aff_q) {
            const double pw = sPu * pv;
            const double update = etaPu * (dLu * dv + O.beta * pw);
            p = (pw - update) / sPnu;
          } else {
            // row'(dL) = (pw - eta (dL dv + beta pw)) / s' = A + dL B: A goes into the forward pass, B into the slope, the recipe
            // is kept (the writer's forwarding area may be reused by the time the exact row is formed)
            const double pw = sPu * pv;
            p = (pw - etaPu * (O.beta * pw)) / sPnu;
            if (r == 0) {
              affB[o * kWave + s] = -(etaPu * dv) / sPnu;
              affP[o * kWave + s] = pv;
              affD[o * kWave + s] = dv;
            }
          }
          Pl[c * Kp + s] = p;
        }
      }
      __syncthreads();
    }

    // ---- C. per order: computeAnova (sgd.nim:146-173) in every wavefront, computeAnovaDerivative (:176-188) of the
    // order's slots dealt to the wavefronts; one chain term per order behind the linear terms ----
    const int MC = a.FW - kWinHdr;
    double ktot[DG];  // (no conductor: the orders' kernels, chain terms behind the linear ones -- wavefront 0's)
    auto forward_orders = [&]() {
#pragma unroll
    for (int t = 0; t < DG; ++t) ktot[t] = 0.0;
    for (int o = 0; o < nb; ++o) {
      const int deg = M.deg_of(o);
      double A[DG + 1];
      double kv;
      if (deg != 2) {  // sgd.nim:152-159
        A[0] = 1.0;
#pragma unroll
        for (int t = 1; t <= DG; ++t) A[t] = 0.0;
        for (int qb = 0; qb < m; qb += 8) {  // eight entries' values requested together (a dependent LDS read per entry otherwise)
          double v_[8], p_[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int qq = qb + e < m ? qb + e : qb;
            v_[e] = vl[qq];
            p_[e] = Pl[((size_t)qq * nb + o) * Kp + s];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool ok = qb + e < m;
            const double val = v_[e];
            const double p = sP * p_[e];
#pragma unroll
            for (int t = DG; t >= 1; --t)
              if (t <= deg) {
                const double nx = A[t] + A[t - 1] * p * val;
                A[t] = ok ? nx : A[t];
              }
          }
        }
        kv = 0.0;
#pragma unroll
        for (int t = 1; t <= DG; ++t)
          if (t == deg) kv = A[t];
      } else {  // sgd.nim:160-170
        double a1 = 0.0, a2 = 0.0;
        for (int qb = 0; qb < m; qb += 8) {
          double v_[8], p_[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int qq = qb + e < m ? qb + e : qb;
            v_[e] = vl[qq];
            p_[e] = Pl[((size_t)qq * nb + o) * Kp + s];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool ok = qb + e < m;
            const double vp = v_[e] * (sP * p_[e]);
            a1 = ok ? a1 + vp : a1;
            a2 = ok ? a2 + vp * vp : a2;
          }
        }
        A[0] = 1.0;
        A[1] = a1;
#pragma unroll
        for (int t = 2; t <= DG; ++t) A[t] = 0.0;
        kv = (a1 * a1 - a2) / 2;
      }
      for (int qb = wv * R; qb < m; qb += NW * R) {  // the order's slots: row slot r of wavefront wv takes entries wv R + r, ...
        const int q = qb + r;
        if (q < m) {
          const double val = vl[q];
          const double p = sP * Pl[((size_t)q * nb + o) * Kp + s];
          double d_;
          if (deg != 2) {
            d_ = val;
#pragma unroll
            for (int t = 1; t < DG; ++t)
              if (t < deg) d_ = val * (A[t] - p * d_);
          } else {
            d_ = val * (A[1] - p * val);
          }
          Tl[((size_t)q * nb + o) * Kp + s] = d_;
        }
      }
      if (wv == 0) {
        if (r == 0) red[s] = s < k ? kv : 0.0;
        compiler_fence();
        double tot = 0.0;
        for (int sb = 0; sb < k; sb += 8) {  // sgd.nim:172-173, ascending s
          double r_[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) r_[t] = red[sb + t < k ? sb + t : sb];
#pragma unroll
          for (int t = 0; t < 8; ++t) tot = sb + t < k ? tot + r_[t] : tot;
        }
        compiler_fence();  // (red is written again by the next order)
        if (a.no_cond || a.one_term) {
#pragma unroll
          for (int t = 0; t < DG; ++t) ktot[t] = t == o ? tot : ktot[t];
        } else if (lane == 0) {
          st_u64(mb + m + o, mail_bits(tot));  // the order's kernel: a chain term behind the linear terms
        }
      }
    }
    };
    forward_orders();
    if (affine) __syncthreads();  // (the slope needs the derivatives of the affine entry's slots: all wavefronts')
    if (wv == 0 && (a.no_cond || a.one_term)) {
      // predictWithGrad's chain (sgd.nim:193-201) in this wavefront: the constant intercept, the linear terms in storage order,
      // the orders' kernels one after the other; dloss; {dL, yhat} posted as the conductor would (all wavefronts and the near
      // successors read them from there).  One-term window: the same sum from 0.0 is the mailbox term.
      double yh_ = a.no_cond ? b_const : 0.0;
      for (int eb = 0; eb < m; eb += 8) {
        double w_[8], v_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int e = eb + t < m ? eb + t : eb;
          w_[t] = wl[e];
          v_[t] = vl[e];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) yh_ = eb + t < m ? yh_ + (sw * w_[t]) * v_[t] : yh_;
      }
#pragma unroll
      for (int t = 0; t < DG; ++t) yh_ = t < nb ? yh_ + ktot[t] : yh_;
      if (a.no_cond) {
        const double dL_ = dev::loss_grad(O.loss, O.loss_param, y, yh_);
        if (lane < kResWords) {
          const double v = lane < 2 ? dL_ : yh_;
          const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
          st_u64(a.res + (size_t)(slot * a.np + par) * kResWords + lane, ((ull)(unsigned)(u + 1) << 32) | (ull)half);
        }
      } else if (!affine) {
        const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
        post_sum(a.fwd, a.np, W, slot, par, (unsigned)(u + 1), lane, yh_, y, h2);
      } else {
        // dS / d dL_writer: the orders' kernels are multilinear in the rows, so the slope is  sum_o sum_s B_os dA_os  -- the
        // derivative of the affine entry's slots (it does not depend on their own rows) -- plus the linear term's  sw Bw x
        const double h2 = dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
        double part = 0.0;
        if (r == 0 && s < k)
          for (int o = 0; o < nb; ++o) part += (sP * affB[o * kWave + s]) * Tl[((size_t)aff_q * nb + o) * Kp + s];  // (d true value = sP d stored)
        double c1 = dev::wave_sum(part);
        c1 += dev::shfl_d((sw * aff_Bw) * vl[aff_q < m ? aff_q : 0], aff_q);
        const unsigned dist = (unsigned)(u - ((int64_t)pl[aff_q] - a.seg0));  // (1 ... near_r - 1 <= 127)
        post_sum(a.fwd, a.np, W, slot, par, (unsigned)(u + 1), lane, yh_, y, h2, c1, 0.0, 0.0, dist);
      }
    } else if (wv == 0) {
      const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
      for (int e = lane; e < a.FW; e += kWave) {
        double val;
        bool put = true;
        if (e < m) val = (sw * wl[e]) * vl[e];
        else if (e < m + nb) put = false;  // an order's kernel, stored above
        else if (e < MC) val = -0.0;       // (changes no sum)
        else if (e == MC) val = -0.0;      // no separate interaction sum
        else if (e == MC + 1) val = y;
        else if (e == MC + 2) val = h2;
        else val = (double)(m + nb);       // the number of chain terms
        if (put) st_u64(mb + e, mail_bits(val));
      }
    }
    __syncthreads();  // (the derivatives of all slots, for the recipes and the update)
    if (affine) {
      // ... and now the exact rows of the affine entry: the writer's dL, its update of the entry's rows, this sample's kernels
      // and derivatives again with them (they enter the update and this sample's own recipes)
      if constexpr (!ADA) {
        const int64_t upa = (int64_t)pl[aff_q] - a.seg0;
        if (wv == 0) {
          const ull* rsa = res_of(upa);
          double dLa = 0.0;
          Spin sp;
          while (true) {
            const bool ok = fw_load(rsa, (unsigned)(upa + 1), dLa);
            if (__all(ok)) break;
            if (sp.wait(a.ctrl)) break;  // (aborting: every wait below ends the same way)
          }
          if (lane == aff_q) {
            bc[lane] = dLa;
            double wvv = wu;
            if (M.fit_linear) {
              const double wj = swul * wu;
              wvv = (wj - etawul * (dLa * vsl + O.alpha * wj)) / (swul * (1 - etawul * O.alpha));
            }
            wl[lane] = wvv;
          }
        }
        __syncthreads();
        const double dLu = bc[aff_q], sPu = bc[kWave + aff_q], etaPu = bc[2 * kWave + aff_q], sPnu = bc[3 * kWave + aff_q];
        for (int o = wv; o < nb; o += NW) {
          const double pw = sPu * affP[o * kWave + s];
          const double update = etaPu * (dLu * affD[o * kWave + s] + O.beta * pw);
          Pl[((size_t)aff_q * nb + o) * Kp + s] = (pw - update) / sPnu;
        }
        __syncthreads();
        forward_orders();
        __syncthreads();
      }
    }
    // (after the mailbox: a successor needs them together with this sample's dL, which the conductor forms from the mailbox)
    if (hotmask) {
      const unsigned mytag = (unsigned)(u + 1);
      ull* fwm = fw_area(u);
      for (ull mk = hotmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        if (r == 0) {
          for (int o = wv; o < nb; o += NW) {
            const size_t c = (size_t)q * nb + o;
            fw_store(fw_slot(fwm, q, o, 0), mytag, Pl[c * Kp + s]);
            fw_store(fw_slot(fwm, q, o, 1), mytag, Tl[c * Kp + s]);
            if constexpr (ADA) {
              fw_store(fw_slot(fwm, q, o, 2), mytag, Gl[c * Kp + s]);
              fw_store(fw_slot(fwm, q, o, 3), mytag, Nl[c * Kp + s]);
            }
          }
        }
      }
      if (wv == NW - 1 && ((hotmask >> lane) & 1ull)) {
        fw_store(fw_lin(fwm, 0, lane), mytag, wl[lane]);
        if constexpr (ADA) {
          if (M.fit_linear) {
            fw_store(fw_lin(fwm, 1, lane), mytag, gwl[lane]);
            fw_store(fw_lin(fwm, 2, lane), mytag, nwl[lane]);
          }
        }
        fw_store(fw_lin(fwm, 3, lane), mytag, vl[lane]);
      }
    }

    if (a.trace && tid == 0)  // mailbox (and recipes) posted; [1]: the same stamp x 16 + which way (2 near rows, 8 affine)
      a.trace[u * 8 + 2] = wall_clock64(), a.trace[u * 8 + 1] = wall_clock64() * 16 + (fwdmask ? 2 : 0) + (affine ? 8 : 0);
    // ---- D. the step sizes while the conductor works ----
    double eta_w = 0.0, eta_P = 0.0, sPn = 1.0, swn = 1.0;
    if constexpr (!ADA) {
      eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      sPn = sP * (1 - eta_P * O.beta);
      swn = sw * (1 - eta_w * O.alpha);
    }

    // ---- E. {dL, yhat} from the conductor (tagged granules): every wavefront takes them itself ----
    double dL, yh;
    bool dead = false;
    {
      Spin sp;
      double rd;
      while (true) {
        const bool ok = fw_load(rp + (size_t)(lane & 1) * 2, (unsigned)(u + 1), rd);
        if (__all(ok)) break;
        if (sp.wait(a.ctrl)) {
          dead = true;
          break;
        }
      }
      dL = dev::shfl_d(rd, 0);
      yh = dev::shfl_d(rd, 1);
    }
    if (dead) break;  // (the launch is being aborted: every wavefront finds the abort word set)
    if (a.trace && tid == 0) a.trace[u * 8 + 3] = wall_clock64();  // dL received
    if (wv == 0) {
      if (!a.no_cond && !a.one_term)
        for (int e = lane; e < a.FW; e += kWave) st_u64(mb + e, kWinSentinel);  // (as in the general worker)
      if (lane == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    }

    // ---- E2. rows formed from a near writer's recipe are stored only after that writer's own stores have landed (see win_worker) ----
    if (fwdmask) {
      if (wv == 0) {
        Spin sp;
        bool first = true;
        while (true) {
          bool need = false;
          for (int q = lane; q < m; q += kWave) {
            const int64_t v2 = (int64_t)pl[q] - a.seg0;
            if (v2 >= 0 && ll[q] && cnt[v2 & (W - 1)] <= (unsigned)(v2 >> lgW)) need = true;
          }
          if (!__any(need)) break;
          if (!first && sp.wait(a.ctrl)) break;  // (aborting)
          first = false;
          for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
          compiler_fence();
        }
      }
      __syncthreads();
    }
    // ---- F. update() / updateG() over all slots (the shared loops of sgd.nim:205-243, adagrad.nim:113-134) ----
    for (int cb = wv * R * U; cb < nsl; cb += NW * R * U) {
#pragma unroll
      for (int t = 0; t < U; ++t) {
        const int c = cb + t * R + r;
        const bool in = c < nsl;
        const size_t e = slot_row(c);
        const double d_ = Tl[(size_t)c * Kp + s];
        if constexpr (ADA) {
          const double grad = dL * d_;
          st_f64_at(in ? (ull)(O.G + e) : (ull)junk, Gl[(size_t)c * Kp + s] + grad);
          st_f64_at(in ? (ull)(O.N + e) : (ull)(junk + kWave), Nl[(size_t)c * Kp + s] + grad * grad);
        } else {
          const double p = sP * Pl[(size_t)c * Kp + s];
          const double update = eta_P * (dL * d_ + O.beta * p);
          viol_acc += in ? fabs(update) : 0.0;
          st_f64_at(in ? (ull)(M.P + e) : (ull)junk, (p - update) / sPn);
        }
      }
    }
    if (M.fit_linear && wv == NW - 1) {
      for (int q = lane; q < m; q += kWave) {
        const int j = jl[q];
        if constexpr (ADA) {
          const double gg = dL * vl[q];
          st_f64(O.Gw + j, gwl[q] + gg);
          st_f64(O.Nw + j, nwl[q] + gg * gg);
        } else {
          const double wj = sw * wl[q];
          const double update = eta_w * (dL * vl[q] + O.alpha * wj);
          viol_acc += fabs(update);
          st_f64(M.w + j, (wj - update) / swn);
        }
      }
    }
    // ---- G. rows written (every wavefront's stores): tell the waiters ----
    unsigned cr_[4] = {0u, 0u, 0u, 0u};
    if (a.no_cond && wv == 0) {  // (the other workers' counters for the next sample's run-ahead check ride along with the drain)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * kWave + lane < W) cr_[t] = ld_u32(a.completed + t * kWave + lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.no_cond && wv == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * kWave + lane < W) cnt[t * kWave + lane] = cr_[t];
    }
    __syncthreads();
    if (tid == 0) st_u32(a.completed + slot, (unsigned)(u >> lgW) + 1u);
    if (a.trace && tid == 0) a.trace[u * 8 + 4] = wall_clock64();  // rows written
  }
  viol_acc = dev::wave_sum(viol_acc);
  if (lane == 0) vsum[wv] = viol_acc;
  __syncthreads();
  if (tid == 0) {
    double v = 0.0;
    for (int w_ = 0; w_ < NW; ++w_) v += vsum[w_];
    a.partial[2 * slot] = loss_acc;
    a.partial[2 * slot + 1] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// conductor: wavefront 1 fetches the mailboxes in sample order into an LDS ring, wavefront 0 runs the scalar chain
// ------------------------------------------------------------------------------------------------------------------
// the conductor's three LDS words (ready, consumed, abort): volatile accesses in the LDS address space (ds_read / ds_write)
typedef __attribute__((address_space(3))) unsigned lds_uint;
__device__ __forceinline__ unsigned ldsv_load(unsigned* p) { return *(volatile lds_uint*)(lds_uint*)p; }
__device__ __forceinline__ void ldsv_store(unsigned* p, unsigned v) { *(volatile lds_uint*)(lds_uint*)p = v; }

template <int OPT, int CH>  // CH: terms of the linear part the chain requests from LDS together (MC is a multiple)
__device__ __forceinline__ void win_conductor(const WinArgs& a, double* lds) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int FW = a.FW, W = a.W, lgW = a.lgW;
  unsigned* c_ready = reinterpret_cast<unsigned*>(lds);  // samples the fetch wavefront has put into the ring
  unsigned* c_consumed = c_ready + 1;                      // samples the chain wavefront is done with
  unsigned* c_abort = c_ready + 2;
  ull* ring = reinterpret_cast<ull*>(lds + 2);  // [kWinRing][FW]
  if (threadIdx.x == 0) {
    ldsv_store(c_ready, 0u);
    ldsv_store(c_consumed, 0u);
    ldsv_store(c_abort, 0u);
  }
  __syncthreads();
  const int64_t n = a.n_seg;
  if (wave == 1) {
    // ---- fetch: kWinDepth mailboxes requested ahead; a word still "empty" means the worker has not posted yet ----
    const int NL = (FW + kWave - 1) / kWave;
    ull v[kWinDepth][kWinMaxNL];
    // Always kWinMaxNL loads per mailbox (a lane past the mailbox's end reads its last word again) and never a load
    // behind a branch: the loads of kWinDepth mailboxes are in flight, and the wait before a mailbox is looked at must be
    // for ITS loads only -- the compiler counts the loads issued since, which it can only do when every path issues the
    // same number (with conditional loads it fell back to vmcnt(0): every sample waited for the mailbox requested last,
    // a full round trip per sample; the prefetch bought nothing).
    auto issue = [&](int dd, int64_t u) {
      const int64_t uu = u < n ? u : n - 1;  // (past the end: the last mailbox again, never looked at)
      const int slot = (int)(uu & (W - 1)), par = (int)((uu >> lgW) & (a.np - 1));
      const ull* mb = a.fwd + (size_t)(slot * a.np + par) * FW;
#pragma unroll
      for (int l = 0; l < kWinMaxNL; ++l) {
        const int e = lane + kWave * l;
        v[dd][l] = ld_u64(mb + (e < FW ? e : FW - 1));
      }
    };
#pragma unroll
    for (int dd = 0; dd < kWinDepth; ++dd) issue(dd, dd);
    int64_t consumed_seen = 0;
    for (int64_t ub = 0; ub < n; ub += kWinDepth) {
#pragma unroll
      for (int dd = 0; dd < kWinDepth; ++dd) {
        const int64_t u = ub + dd;
        if (u >= n) break;
        bool bad = false;
#pragma unroll
        for (int l = 0; l < kWinMaxNL; ++l) bad = bad || v[dd][l] == kWinSentinel;
        if (__any(bad)) {  // not posted yet: ask again until it is (this path drains every load before it rejoins)
          Spin sp;
          do {
            if (sp.wait(a.ctrl)) {
              ldsv_store(c_abort, 1u);
              return;
            }
            issue(dd, u);
            bad = false;
#pragma unroll
            for (int l = 0; l < kWinMaxNL; ++l) bad = bad || v[dd][l] == kWinSentinel;
          } while (__any(bad));
        }
        if (consumed_seen + kWinRing <= u) {
          Spin sp2;
          while ((consumed_seen = (int64_t)ldsv_load(c_consumed)) + kWinRing <= u) {
            if (sp2.wait(a.ctrl)) {
              ldsv_store(c_abort, 1u);
              return;
            }
          }
        }
        ull* dst = ring + (size_t)(u & (kWinRing - 1)) * FW;
#pragma unroll
        for (int l = 0; l < kWinMaxNL; ++l) {
          const int e = lane + kWave * l;
          if (l < NL && e < FW) dst[e] = v[dd][l];
        }
        lds_fence();  // LDS operations of a wavefront execute in order: data, then the counter
        if (lane == 0) ldsv_store(c_ready, (unsigned)(u + 1));
        if (a.trace && lane == 0) a.trace[u * 8 + 5] = wall_clock64();  // mailbox fetched
        issue(dd, u + kWinDepth);
      }
    }
    return;
  }
  // ---- chain ----
  // Two samples are held in registers at any time: while the additions of sample u run -- each waits for the one
  // before it -- the LDS reads of sample u + 1 are already in flight (one wavefront issues in order: the reads have
  // to be REQUESTED ahead of the chain they would otherwise stall).
  double b = M.sc[SC_INTERCEPT];
  double gsb = 0.0, gnb = 0.0, viol_b = 0.0;
  if (ADA) {
    gsb = O.gsc[0];
    gnb = O.gsc[1];
  }
  const int MC = FW - kWinHdr;  // a multiple of CH
  int64_t ready_seen = 0;
  constexpr int PF = CH < 32 ? CH : 32;  // terms of the NEXT sample requested ahead (two samples' worth must fit the registers)
  struct Terms {
    double2 t[PF / 2];
    double2 h01, h23;  // {the interaction sum, y}, {the intercept's step size, the number of chain terms}
  };
  auto slot_of = [&](int64_t u) { return reinterpret_cast<const double*>(ring + (size_t)(u & (kWinRing - 1)) * FW); };
  auto request = [&](int64_t u, Terms& T) {
    const double* sl = slot_of(u);
    T.h01 = *reinterpret_cast<const double2*>(sl + MC);
    T.h23 = *reinterpret_cast<const double2*>(sl + MC + 2);
#pragma unroll
    for (int t = 0; t < PF / 2; ++t) T.t[t] = *reinterpret_cast<const double2*>(sl + 2 * t);
  };
  // false: the launch is being aborted
  // The chain wavefront issues NO vector-memory load inside its loop: the compiler's wait-count insertion is conservative
  // at joins -- one load on a slow path (the abort word) made it wait for vmcnt(0) before every sample's additions, i.e.
  // for the previous sample's answer STORE to be acknowledged (gfx9 counts stores in vmcnt): 1.5 us per sample.  The
  // abort word is watched by the fetch wavefront, which raises the LDS flag; this wavefront only has its clock.
  auto wait_ready = [&](int64_t u) {
    if (ready_seen > u) return true;
    int spins = 0;
    const long long t0 = wall_clock64();
    while ((ready_seen = (int64_t)ldsv_load(c_ready)) <= u) {
      if (ldsv_load(c_abort)) return false;
      if ((++spins & 1023) == 0 && wall_clock64() - t0 > kWinTimeoutTicks) {
        st_u32(a.ctrl, 1u);
        return false;
      }
    }
    compiler_fence();
    return true;
  };
  auto try_request = [&](int64_t u, Terms& T) {  // the next sample's terms, if the fetch wavefront has them already
    if (u >= n) return false;
    if (ready_seen <= u) ready_seen = (int64_t)ldsv_load(c_ready);
    if (ready_seen <= u) return false;
    compiler_fence();
    request(u, T);
    return true;
  };
  auto step = [&](int64_t u, const Terms& T) {
    if (a.trace && lane == 0) a.trace[u * 8 + 6] = wall_clock64();  // chain starts
    const double tot = T.h01.x, y = T.h01.y, h2 = T.h23.x;
    const int nt = (int)T.h23.y;  // terms beyond it are -0.0 padding: not added (they would change nothing but cost an addition each)
    const int64_t it = a.it0 + u;
    if (ADA && it != 1 && M.fit_intercept) {  // adagrad.nim:101-106
      const double old = b;
      const double denom = sqrt(gnb) + h2;
      b = -O.eta0 * gsb / denom;
      viol_b += fabs(old - b);
    }
    const double* sl = slot_of(u);
    double2 t2_[CH > PF ? (CH - PF) / 2 : 1];  // the rest of the first chunk: requested now, used after PF additions
    if constexpr (CH > PF) {
#pragma unroll
      for (int t = 0; t < (CH - PF) / 2; ++t) t2_[t] = *reinterpret_cast<const double2*>(sl + PF + 2 * t);
    }
    double yh = b;  // predictWithGrad, sgd.nim:193-196: intercept first, then the entries in storage order
#pragma unroll
    for (int t = 0; t < PF / 2; ++t) {
      yh += T.t[t].x;
      yh += T.t[t].y;
    }
    if constexpr (CH > PF) {
#pragma unroll
      for (int t = 0; t < (CH - PF) / 2; ++t) {
        yh += t2_[t].x;
        yh += t2_[t].y;
      }
    }
    if (MC > CH) {  // rows longer than one chunk: sixteen terms at a time, as far as the sample's terms go
                    // (requesting the next sixteen before the current ones are added measured SLOWER: 1.16 -> 1.40 us on
                    // cfg4's rows -- the copies between the two sets cost more than the overlap gains)
      for (int qb = CH; qb < nt; qb += 16) {
        double2 t_[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) t_[t] = *reinterpret_cast<const double2*>(sl + qb + 2 * t);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          yh += t_[t].x;
          yh += t_[t].y;
        }
      }
    }
    yh += tot;
    const double dL = dev::loss_grad(O.loss, O.loss_param, y, yh);
    if (M.fit_intercept) {
      if (ADA) {
        gsb += dL;
        gnb += dL * dL;
      } else {
        const double update = h2 * (dL + O.alpha0 * b);
        viol_b += fabs(update);
        b -= update;
      }
    }
    const int slot = (int)(u & (W - 1)), par = (int)((u >> lgW) & (a.np - 1));
    ull* rp = a.res + (size_t)(slot * a.np + par) * kResWords;
    if (lane < kResWords) {  // {dL, yhat} as granules tagged with the sample: its worker and near successors read them
      const double v = lane < 2 ? dL : yh;
      const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
      st_u64(rp + lane, ((ull)(unsigned)(u + 1) << 32) | (ull)half);
    }
    lds_fence();  // the ring slot has been read
    if (lane == 0) ldsv_store(c_consumed, (unsigned)(u + 1));
    if (a.trace && lane == 0) a.trace[u * 8 + 7] = wall_clock64();  // answer posted
  };
  Terms A, B;
  bool haveA = false, haveB = false;
  for (int64_t u = 0; u < n; u += 2) {
    if (!haveA) {
      if (!wait_ready(u)) return;
      request(u, A);
    }
    haveB = try_request(u + 1, B);
    step(u, A);
    if (u + 1 >= n) break;
    if (!haveB) {
      if (!wait_ready(u + 1)) return;
      request(u + 1, B);
    }
    haveA = try_request(u + 2, A);
    step(u + 1, B);
  }
  if (lane == 0) {
    M.sc[SC_INTERCEPT] = b;
    if (ADA) {
      O.gsc[0] = gsb;
      O.gsc[1] = gnb;
    }
    a.partial[2 * W] = 0.0;
    a.partial[2 * W + 1] = viol_b;
  }
}

__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// ------------------------------------------------------------------------------------------------------------------
// conductor of the ONE-TERM window (WinArgs::one_term; the default when the intercept is fitted).  What the reference's order
// really chains through the intercept is  yhat_t = b_t + S_t -> dL_t = dloss(y_t, yhat_t) -> b_{t+1}  (sgd.nim:193-201,224-229;
// adagrad.nim:101-110,121-123): S_t -- the linear terms and the interaction sum -- does not depend on b.  The term-by-term
// conductor above adds S_t's m terms onto b one by one (the reference's rounding, 0.17 us of dependent additions + 0.2 us of
// LDS hand-off per sample); here the WORKER adds them up (same storage order, starting from 0.0) and the chain is one addition,
// dloss and the intercept's step.  north_star asks for 1e-6 relative on predictions; this variant agrees with the term-by-term
// chain to ~1e-15 relative per step (tests: rtol 1e-8 against the oracle).
//   * fetch wavefront: lane = worker (W / 64 workers per lane).  Every poll round loads the six tagged granules of each
//     worker's CURRENT mailbox (kSumDepth rounds in flight: a fresh snapshot of all W mailboxes every few hundred ns), moves
//     the ones that arrived into an LDS ring indexed by sample, and publishes the length of the in-order prefix.
//   * chain wavefront: takes up to 64 consecutive samples out of the ring with ONE LDS access (lane = sample) and walks them
//     with v_readlane: the LDS hand-off is paid per chunk, not per sample.  It issues no vector-memory load (see above).
// A worker posts sample u + W only after it has seen the answer for u, so at most W samples are outstanding: the ring
// (kSumRing >= 2 W slots) needs no "consumed" counter, and a mailbox can be reused two samples of its worker later.
// ------------------------------------------------------------------------------------------------------------------
// dloss(y, p) and its derivative in p (loss.nim:15-102, the Huber quirk kept): what the chunk-parallel chain linearises
template <int LOSS>
__device__ __forceinline__ double loss_grad_slope(double param, double y, double p, double& slope) {
  if constexpr (LOSS == NFM_LOSS_SQUARED) {
    slope = 1.0;
    return p - y;
  } else if constexpr (LOSS == NFM_LOSS_SQUARED_HINGE) {
    const double z = 1 - p * y;
    slope = z > 0 ? 2 * y * y : 0.0;
    return z > 0 ? -2 * y * z : 0.0;
  } else if constexpr (LOSS == NFM_LOSS_LOGISTIC) {
    const double z = p * y;
    const double e = exp(-fabs(z));
    const double r = 1.0 / (1 + e);
    slope = (y * y) * (e * r) * r;  // y^2 sigma (1 - sigma)
    const double num = z > 0 ? -y * e : -y;
    return num / (1 + e);
  } else {
    const double z = fabs(y - p);
    slope = z < param ? -1.0 : 0.0;
    return z < param ? y - p : param;
  }
}

constexpr int kSumRing = 256;  // LDS ring, in samples (a power of two >= 2 W)
constexpr int kSumDepth = 2;   // poll rounds in flight
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ double ldsv_load_d(const double* p) { return *(volatile lds_double*)(lds_double*)p; }

template <int OPT, int NLW>  // NLW = W / 64 workers per lane of the fetch wavefront
__device__ __forceinline__ void win_conductor_sum(const WinArgs& a, double* lds) {
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int W = a.W, lgW = a.lgW, np = a.np;
  unsigned* c_ready = reinterpret_cast<unsigned*>(lds);  // samples [0, c_ready) are in the ring
  unsigned* c_abort = c_ready + 1;
  unsigned* c_done = c_ready + 2;                          // samples [0, c_done) have their {dL, yhat} in rD / rP
  double* rS = lds + 2;            // [kSumRing] the samples' sums
  double* rY = rS + kSumRing;      // [kSumRing] targets
  double* rH = rY + kSumRing;      // [kSumRing] the intercept's step size (AdaGrad: eta0 (it-1) alpha0)
  double* rC = rH + kSumRing;      // [kSumRing] the slope of S in the first writer's dL (post_sum)
  double* rC2 = rC + kSumRing;     // [kSumRing] ... in the second writer's
  double* rC12 = rC2 + kSumRing;   // [kSumRing] ... in their product
  double* rD = rC12 + kSumRing;    // [kSumRing] the chain's dL per sample: what the post wavefront sends out, and the history a
  double* rP = rD + kSumRing;      // [kSumRing]   dependent sample's slope is multiplied by; the predictions yhat
  unsigned* rW = reinterpret_cast<unsigned*>(rP + kSumRing);  // [kSumRing] the writers' distances (first | second << 8), 0: none
  if (threadIdx.x == 0) {
    ldsv_store(c_ready, 0u);
    ldsv_store(c_abort, 0u);
    ldsv_store(c_done, 0u);
  }
  __syncthreads();
  const int64_t n = a.n_seg;
  if (wave == 1) {
    // ---- fetch ----
    int64_t uw[NLW];  // the next sample of worker lane + 64 w that is not in the ring yet
#pragma unroll
    for (int w = 0; w < NLW; ++w) uw[w] = lane + kWave * w < W ? lane + kWave * w : n;  // (fewer than 64 workers: idle lanes)
    ull g[kSumDepth][NLW][kSumGran];
    // a fixed number of loads per round, none behind a branch (the wait before a round is looked at is then for ITS loads only)
    auto issue = [&](int dd) {
#pragma unroll
      for (int w = 0; w < NLW; ++w) {
        const int slot = (lane + kWave * w) & (W - 1);
        const int64_t u = uw[w] < n ? uw[w] : slot;  // (a worker that is done: any of its mailboxes, never taken)
        const int par = (int)((u >> lgW) & (np - 1));
#pragma unroll
        for (int gi = 0; gi < kSumGran; ++gi) g[dd][w][gi] = ld_u64(a.fwd + ((size_t)(gi * np + par) * W + slot));
      }
    };
#pragma unroll
    for (int dd = 0; dd < kSumDepth; ++dd) issue(dd);
    int64_t cf = 0;  // samples [0, cf) are in the ring
    long long t_last = wall_clock64();
    int rounds = 0;
    while (cf < n) {
#pragma unroll
      for (int dd = 0; dd < kSumDepth; ++dd) {
        ull done_m[NLW];
#pragma unroll
        for (int w = 0; w < NLW; ++w) {
          const unsigned tag = (unsigned)(uw[w] + 1);
          bool ok = uw[w] < n;
#pragma unroll
          for (int gi = 0; gi < kSumGran; ++gi) ok = ok && (unsigned)(g[dd][w][gi] >> 32) == tag;
          if (ok) {
            const int idx = (int)(uw[w] & (kSumRing - 1));
            rS[idx] = __hiloint2double((int)(unsigned)g[dd][w][1], (int)(unsigned)g[dd][w][0]);
            rY[idx] = __hiloint2double((int)(unsigned)g[dd][w][3], (int)(unsigned)g[dd][w][2]);
            rH[idx] = __hiloint2double((int)(unsigned)g[dd][w][5], (int)(unsigned)g[dd][w][4]);
            rC[idx] = __hiloint2double((int)(unsigned)g[dd][w][7], (int)(unsigned)g[dd][w][6]);
            rC2[idx] = __hiloint2double((int)(unsigned)g[dd][w][9], (int)(unsigned)g[dd][w][8]);
            rC12[idx] = __hiloint2double((int)(unsigned)g[dd][w][11], (int)(unsigned)g[dd][w][10]);
            rW[idx] = (unsigned)g[dd][w][12];
            if (a.trace) a.trace[uw[w] * 8 + 5] = wall_clock64();  // mailbox fetched
            uw[w] += W;
          }
          // the worker's sample inside [cf, cf + W) is in the ring (its next one lies beyond), or the worker has none left
          done_m[w] = __ballot(uw[w] >= cf + W || uw[w] >= n);
        }
        // the in-order prefix: the workers' bits as a ring of W, read from worker cf mod W on
        int adv;
        if constexpr (NLW == 1) {  // W <= 64: lanes past W have no worker
          const ull full = W >= kWave ? ~0ull : (1ull << W) - 1ull;
          const ull m0 = done_m[0] & full;
          const int r = (int)(cf & (W - 1));
          const ull rot = r ? ((m0 >> r) | (m0 << (W - r))) & full : m0;
          adv = rot == full ? W : __builtin_ctzll(~rot);
        } else {
          int r = (int)(cf & 127);
          ull m0 = done_m[0], m1 = done_m[NLW - 1];
          if (r >= 64) {
            const ull t_ = m0;
            m0 = m1;
            m1 = t_;
            r -= 64;
          }
          const ull lo = r ? (m0 >> r) | (m1 << (64 - r)) : m0;
          const ull hi = r ? (m1 >> r) | (m0 << (64 - r)) : m1;
          adv = lo != ~0ull ? __builtin_ctzll(~lo) : hi != ~0ull ? 64 + __builtin_ctzll(~hi) : 128;
        }
        if (adv > 0) {
          cf = cf + adv < n ? cf + adv : n;
          lds_fence();  // LDS operations of a wavefront execute in order: data, then the counter
          if (lane == 0) ldsv_store(c_ready, (unsigned)cf);
          t_last = wall_clock64();
        }
        issue(dd);
      }
      if ((++rounds & 15) == 0) {  // the launch's abort word and the wall-clock limit (no progress for 1 s)
        if (ld_u32(a.ctrl) != 0u || wall_clock64() - t_last > kWinTimeoutTicks) {
          st_u32(a.ctrl, 1u);
          ldsv_store(c_abort, 1u);
          return;
        }
      }
    }
    return;
  }
  if (wave == 2) {
    // ---- post: {dL, yhat} of the samples the chain has finished go out as tagged granules (their workers and near successors
    // poll them), lane = sample.  A wavefront issues one instruction every few cycles whatever it is: the address arithmetic,
    // the tagging and the store were a quarter of the chain's instructions per sample -- here they cost the chain two LDS writes.
    int64_t pdone = 0;
    long long t_last = wall_clock64();
    int spins = 0;
    while (pdone < n) {
      const int64_t done = (int64_t)__builtin_amdgcn_readfirstlane((int)ldsv_load(c_done));
      if (done <= pdone) {
        if (ldsv_load(c_abort)) return;
        if ((++spins & 1023) == 0 && wall_clock64() - t_last > kWinTimeoutTicks) {
          st_u32(a.ctrl, 1u);
          ldsv_store(c_abort, 1u);
          return;
        }
        continue;
      }
      compiler_fence();
      const int cntb = (int)(done - pdone < kWave ? done - pdone : kWave);
      if (lane < cntb) {
        const int64_t uu = pdone + lane;
        const int idx = (int)(uu & (kSumRing - 1));
        const double dL = ldsv_load_d(rD + idx), yh = ldsv_load_d(rP + idx);
        ull* rp = a.res + ((size_t)(uu & (W - 1)) * np + (size_t)((uu >> lgW) & (np - 1))) * kResWords;
        const ull tg = (ull)(unsigned)(uu + 1) << 32;
        st_u64(rp + 0, tg | (ull)(unsigned)__double2loint(dL));
        st_u64(rp + 1, tg | (ull)(unsigned)__double2hiint(dL));
        st_u64(rp + 2, tg | (ull)(unsigned)__double2loint(yh));
        st_u64(rp + 3, tg | (ull)(unsigned)__double2hiint(yh));
        if (a.trace) a.trace[uu * 8 + 7] = wall_clock64();  // answer posted
      }
      pdone += cntb;
      t_last = wall_clock64();
    }
    return;
  }
  // ---- chain ----
  double b = M.sc[SC_INTERCEPT];
  double gsb = 0.0, gnb = 0.0, viol_b = 0.0;
  if (ADA) {
    gsb = O.gsc[0];
    gnb = O.gsc[1];
  }
  const bool fit_b_rt = M.fit_intercept != 0;
  // The loop is compiled once per loss (and with / without the debugging stamps): with the loss a run-time switch every
  // sample paid a dozen scalar branches that wait on vector compares -- as much as the arithmetic itself.
  auto run = [&](auto loss_c, auto trace_c, auto fitb_c) -> bool {
    constexpr int LOSS = decltype(loss_c)::value;
    constexpr bool TRACE = decltype(trace_c)::value;
    constexpr bool fit_b = decltype(fitb_c)::value;  // (fitIntercept: a select per intercept word and sample otherwise)
    int64_t u = 0;
    int64_t ready = 0;
    [[maybe_unused]] const long long cyc0 = clock64(), rt0 = wall_clock64();
    while (u < n) {
      if (ready <= u) {
        int spins = 0;
        const long long t0 = wall_clock64();
        while ((ready = (int64_t)__builtin_amdgcn_readfirstlane((int)ldsv_load(c_ready))) <= u) {
          if (ldsv_load(c_abort)) return false;
          if ((++spins & 1023) == 0 && wall_clock64() - t0 > kWinTimeoutTicks) {
            st_u32(a.ctrl, 1u);
            return false;
          }
        }
      }
      compiler_fence();
      // up to 64 consecutive samples, lane = sample (lanes past `ready` read slots that are not theirs yet: never used)
      const int idx = (int)((u + lane) & (kSumRing - 1));
      const double Sv = ldsv_load_d(rS + idx), yv = ldsv_load_d(rY + idx), hv = ldsv_load_d(rH + idx), Cv = ldsv_load_d(rC + idx);
      const double C2v = ldsv_load_d(rC2 + idx), C12v = ldsv_load_d(rC12 + idx);
      const int Wv = (int)ldsv_load(rW + idx);
      const int cnt = (int)(ready - u < kWave ? ready - u : kWave);
      const ull affmask = __ballot(lane < cnt && Wv != 0);  // samples whose S waits for a writer's dL
      // ---- the chunk in parallel (SGD).  A wavefront issues an instruction every ~8 cycles, and the sample-by-sample loop
      // below costs ~90 of them per sample: with a backlog it IS the window's bound.  But the chain is a recurrence in ONE
      // number:  b' = b - h (dloss(y, b + S) + alpha0 b).  Linearised around estimates bt_t of the intercept before each sample,
      // dloss = d_t + g_t (b - bt_t), it is an affine map b' = A_t b + B_t per sample, and affine maps compose associatively:
      // a prefix scan over the lanes (lane = sample) gives every b_t at once; new estimates, again -- Newton's method on the whole
      // chunk, quadratic: from bt_t = b_0 the estimates are exact to rounding after three or four passes (squared loss: the
      // recurrence IS affine, one pass).  A pass costs ~100 instructions for 64 samples.  The result satisfies the reference's
      // recurrence to ~1e-16 relative (the scan associates the products differently): the one-term flavour's tolerance, not bits.
      // A sample whose S waits for the dL of a writer inside the same chunk takes that dL from the previous pass.  No
      // convergence in eight passes (or a NaN): the sample-by-sample loop.
      bool chunk_done = false;
      if constexpr (!ADA) {
        if (a.par_min > 0 && cnt >= a.par_min) {
          const bool in = lane < cnt;
          const int64_t me = u + lane;
          const int d1 = Wv & 255, d2 = (Wv >> 8) & 255;             // the writers' distances (0: none)
          const bool w1_here = d1 != 0 && me - d1 >= u, w2_here = d2 != 0 && me - d2 >= u;  // ... inside this chunk: their dL are being found, too
          const bool w_here = w1_here || w2_here;
          // the dL of writers of EARLIER chunks: final, in the ring
          const double dL1_old = (in && d1 != 0 && !w1_here) ? ldsv_load_d(rD + ((me - d1) & (kSumRing - 1))) : 0.0;
          const double dL2_old = (in && d2 != 0 && !w2_here) ? ldsv_load_d(rD + ((me - d2) & (kSumRing - 1))) : 0.0;
          auto S_of = [&](double dl1, double dl2) { return Sv + dl1 * Cv + dl2 * (C2v + dl1 * C12v); };  // (no writer: dl = 0, C = 0)
          const double S0 = S_of(dL1_old, dL2_old);
          double bt = b, St = S0, dLt = 0.0, yht = 0.0, updt = 0.0, bnext = b;
          bool conv = false;
          const bool coupled = __any(in && w_here);
          // Newton converges quadratically: estimates that moved by less than 1e-9 of their scale are exact to rounding after
          // this pass.  The coupling through a writer inside the chunk converges linearly (its dL enters one pass late): 1e-13.
          const double tol_rel = coupled ? 1e-13 : 1e-9;
          for (int pass = 0; pass < 8 && !conv; ++pass) {
            const double dLprev = dLt;
            if (pass > 0 && coupled) {  // S of the samples that wait for a dL of this chunk
              const double dLa = __shfl(dLt, (lane - d1) & (kWave - 1), kWave), dLb = __shfl(dLt, (lane - d2) & (kWave - 1), kWave);
              St = (in && w_here) ? S_of(w1_here ? dLa : dL1_old, w2_here ? dLb : dL2_old) : S0;
            }
            double g;
            yht = bt + St;
            dLt = loss_grad_slope<LOSS>(O.loss_param, yv, yht, g);
            // b' = b - h (d + g (b - bt) + alpha0 b) = A b + B
            double A = 1.0, B = 0.0;
            if (fit_b && in) {
              A = 1.0 - hv * (g + O.alpha0);
              B = -hv * (dLt - g * bt);
            }
            double PA = A, PB = B;
#pragma unroll
            for (int dd = 1; dd < kWave; dd <<= 1) {
              const double Ap = __shfl_up(PA, dd, kWave), Bp = __shfl_up(PB, dd, kWave);
              if (lane >= dd) {
                PB = PA * Bp + PB;
                PA = PA * Ap;
              }
            }
            const double after = PA * b + PB;                       // the intercept after this lane's sample
            const double before = __shfl_up(after, 1, kWave);
            const double bt_new = lane == 0 ? b : before;
            // converged: no estimate moved (scale: the intercept and a chunk's worth of its steps), nor a writer's dL
            const double scale = fabs(bt_new) + 64.0 * fabs(hv * dLt) + 1e-300;
            bool moved = in && !(fabs(bt_new - bt) <= tol_rel * scale);
            if (coupled) moved = moved || pass == 0 || (in && !(fabs(dLt - dLprev) <= tol_rel * (fabs(dLt) + 1e-300)));
            conv = !__any(moved);
            bt = bt_new;
            bnext = after;
            updt = in && fit_b ? hv * (dLt + O.alpha0 * bt) : 0.0;
            if (conv && pass == 0) {  // (the estimates did not move: dLt, yht are already those of bt)
            } else if (conv) {
              double g2;
              yht = bt + St;
              dLt = loss_grad_slope<LOSS>(O.loss_param, yv, yht, g2);
              updt = in && fit_b ? hv * (dLt + O.alpha0 * bt) : 0.0;
            }
          }
          if (conv) {
            const int di = (int)((u + lane) & (kSumRing - 1));
            if (in) {
              rD[di] = dLt;
              rP[di] = yht;
            }
            viol_b += dev::wave_sum(fabs(updt));
            b = readlane_d(bnext, cnt - 1);
            lds_fence_order();
            if (lane == 0) ldsv_store(c_done, (unsigned)(u + cnt));
            if constexpr (TRACE) {
              if (in) a.trace[(u + lane) * 8 + 6] = wall_clock64();
            }
            chunk_done = true;
          }
        }
      }
      // (AdaGrad: the same Newton iteration on the PAIR (g_sum, g_norm) of the intercept's state -- 2 x 2 affine maps composed by
      // the scan -- was built and measured: rows that share nothing 0.34 -> 0.24 us per sample at 128 workers, the headline /
      // cfg2 / cfg4 shapes +-0: with AdaGrad a dependent sample waits for its writer's dL whatever the chain costs (the
      // writer's step is not affine in dL), and those turnarounds are the bound.  Not kept.)
      for (int t = 0; t < cnt && !chunk_done; ++t) {
        double S = readlane_d(Sv, t);
        const double y = readlane_d(yv, t), h2 = readlane_d(hv, t);
        const int64_t uu = u + t;
        const bool aff_t = (affmask >> t) & 1ull;
        if (aff_t) {  // S waits for the dL of one or two earlier samples still inside the window (post_sum): made here
          const int wp = readlane_i(Wv, t);
          const int d1 = wp & 255, d2 = (wp >> 8) & 255;  // (fewer than W <= 128 positions back: still in the ring)
          const double dL1 = ldsv_load_d(rD + ((uu - d1) & (kSumRing - 1)));
          S += dL1 * readlane_d(Cv, t);
          if (d2 != 0) {
            const double dL2 = ldsv_load_d(rD + ((uu - d2) & (kSumRing - 1)));
            S += dL2 * (readlane_d(C2v, t) + dL1 * readlane_d(C12v, t));
          }
        }
        if (ADA && fit_b && a.it0 + uu != 1) {  // adagrad.nim:101-106
          const double old = b;
          b = -O.eta0 * gsb / (sqrt(gnb) + h2);
          viol_b += fabs(old - b);
        }
        const double yh = b + S;  // predictWithGrad, sgd.nim:193-201, with everything but the intercept added up by the worker
        double dL;
        if constexpr (LOSS == NFM_LOSS_LOGISTIC) {
          // loss.nim:66-71 without its branch: z > 0: -y e^{-z} / (1 + e^{-z}), else -y / (e^{z} + 1) -- one exponential of
          // -|z| serves both (the same operations on the same values: the same bits).  (Tried: exp(-|z|) from a per-chunk
          // lane-parallel exp(-|z0|) times a degree-11 polynomial of |z0| - |z| -- 20 dependent operations fewer per sample,
          // measured +-0: a wavefront issues an instruction every ~8 cycles whatever it is, and the sample costs ~90 of them.)
          const double z = yh * y;
          const double e = exp(-fabs(z));
          const double num = z > 0 ? -y * e : -y;
          dL = num / (1 + e);
        } else {
          dL = dev::loss_grad(LOSS, O.loss_param, y, yh);
        }
        if (fit_b) {
          if (ADA) {  // adagrad.nim:121-123
            gsb += dL;
            gnb += dL * dL;
          } else {  // sgd.nim:224-229
            const double update = h2 * (dL + O.alpha0 * b);
            viol_b += fabs(update);
            b -= update;
          }
        }
        // {dL, yhat} to the post wavefront (LDS operations of a wavefront execute in order: data, then the counter)
        if (lane == 0) {
          const int di = (int)(uu & (kSumRing - 1));
          rD[di] = dL;
          rP[di] = yh;
          lds_fence_order();
          ldsv_store(c_done, (unsigned)(uu + 1));
        }
        if constexpr (TRACE) {
          if (lane == 0) a.trace[uu * 8 + 6] = wall_clock64();  // chain done with the sample
        }
      }
      u += cnt;
    }
    if constexpr (TRACE) {  // (debugging: the shader clock against the 100 MHz wall clock over the launch, in the last sample's row)
      if (lane == 0 && n >= 1) {
        a.trace[(n - 1) * 8 + 0] = clock64() - cyc0;
        a.trace[(n - 1) * 8 + 1] = wall_clock64() - rt0;
      }
    }
    return true;
  };
  auto run_l = [&](auto loss_c) {
    if (a.trace) return fit_b_rt ? run(loss_c, std::true_type(), std::true_type()) : run(loss_c, std::true_type(), std::false_type());
    return fit_b_rt ? run(loss_c, std::false_type(), std::true_type()) : run(loss_c, std::false_type(), std::false_type());
  };
  bool done;
  switch (O.loss) {
    case NFM_LOSS_SQUARED: done = run_l(std::integral_constant<int, NFM_LOSS_SQUARED>()); break;
    case NFM_LOSS_SQUARED_HINGE: done = run_l(std::integral_constant<int, NFM_LOSS_SQUARED_HINGE>()); break;
    case NFM_LOSS_LOGISTIC: done = run_l(std::integral_constant<int, NFM_LOSS_LOGISTIC>()); break;
    default: done = run_l(std::integral_constant<int, NFM_LOSS_HUBER>()); break;
  }
  if (!done) return;
  if (lane == 0) {
    M.sc[SC_INTERCEPT] = b;
    if (ADA) {
      O.gsc[0] = gsb;
      O.gsc[1] = gnb;
    }
    a.partial[2 * W] = 0.0;
    a.partial[2 * W + 1] = viol_b;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// worker for rows of 64 factors (Kp = 64) and at most 64 entries: lane = factor, lane q also holds entry q of the sample;
// the sample's 64 stored rows stay in REGISTERS between predictWithGrad and update() (no LDS round trips, no branches
// that depend on a lane), an entry travels by v_readlane.  SGD gathers its rows BEFORE it has waited for the samples it
// depends on and reads again only the rows that wait concerned: the gather's latency is off the path of a sample that
// has to wait.  One divisor -- the new scale -- divides all of a sample's values: a/b is formed as Markstein's twice-
// corrected a * (1/b), which equals the IEEE quotient (tools/divtest.hip: 2.7e11 pairs, no mismatch; a divisor whose
// significand is all ones, where the theorem does not hold, takes the division instruction).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double div_by(double a, double b, double y /* = 1 / b */) {
  const double q0 = a * y;
  const double e0 = fma(-b, q0, a);
  const double q1 = fma(e0, y, q0);
  const double e1 = fma(-b, q1, a);
  return fma(e1, y, q1);
}

template <int OPT>
__device__ __forceinline__ void win_worker_k64(const WinArgs& a, const int slot, double* lds) {
  const CsrView& X = a.X;
  const ModelView& M = a.M;
  const OptView& O = a.O;
  constexpr bool ADA = OPT == OPT_ADAGRAD;
  constexpr int K = 64;
  const int lane = threadIdx.x;
  const int k = M.k, W = a.W, lgW = a.lgW, MC = a.FW - kWinHdr;
  double* Fl = lds;                                    // [64][64] rows that arrived late (forwarded or read again)
  double* Gl = Fl + K * K;                             // AdaGrad: [64][64] g_sum of the sample's rows
  double* Nl = Gl + (ADA ? K * K : 0);                 // AdaGrad: [64][64] g_norm
  double* red = Nl + (ADA ? K * K : 0);                // [64]
  unsigned* cnt = reinterpret_cast<unsigned*>(red + K);  // [W] completion counters as last seen
  for (int l = lane; l < W; l += kWave) cnt[l] = 0u;
  double loss_acc = 0.0, viol_acc = 0.0;
  const double b_const = M.sc[SC_INTERCEPT];  // (no_cond: the intercept is not fitted and stays what it is)
  double Pr[K];
  auto fw_area = [&](int64_t u_) { return a.fw + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kFwSlot; };
  auto fw_row = [&](ull* base, int v, int q) { return base + kFwRows + ((size_t)(v * K + q) * K + lane) * 2; };
  auto fw_lin = [&](ull* base, int v, int q) { return base + kFwLin + (size_t)(v * K + q) * 2; };
  auto res_of = [&](int64_t u_) { return a.res + ((size_t)(u_ & (W - 1)) * a.np + (size_t)((u_ >> lgW) & (a.np - 1))) * kResWords; };

  for (int64_t u = slot; u < a.n_seg; u += W) {
    const int64_t pos = a.seg0 + u, pa = a.begin + pos;
    const int64_t i = a.perm ? a.perm[pa] : pa;
    const int64_t q0 = X.indptr[i];
    const int m = (int)(X.indptr[i + 1] - q0);
    const double y = dev::target_of(X.y[i], M.task);
    const int64_t it = a.it0 + u;
    const double itf = (double)it;
    const int par = (int)((u >> lgW) & (a.np - 1));
    ull* mb = a.fwd + (size_t)(slot * a.np + par) * a.FW;
    if (a.no_cond && u >= (int64_t)a.thr * W) {
      // No conductor walks the samples in order, so nothing else keeps a fast worker from running ahead: its forwarding
      // area and answer words of sample u - np W are about to be reused, and a near successor of that sample (a position
      // below u - (np - 1) W) may not have read them yet.  Every sample below that must be complete: the workers before
      // this one have finished c - (np - 2) samples, the others one fewer (c = this worker's count) -- the window spans
      // at most np W positions.
      const unsigned c_ = (unsigned)(u >> lgW) - (unsigned)(a.thr - 1);
      Spin sp;
      bool first = true;
      while (true) {
        bool ok = true;
        for (int l = lane; l < W; l += kWave) ok = ok && cnt[l] >= (l < slot ? c_ : c_ - 1u);
        if (__all(ok)) break;
        if (!first && sp.wait(a.ctrl)) return;
        first = false;
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
      }
    }
    int jq = 0, pq = -1, pqu = 0, nq = -1;
    double vq = 0.0;
    if (lane < m) {
      jq = X.indices[q0 + lane];
      vq = X.data[q0 + lane];
      pq = a.prev[q0 + lane];
      pqu = a.prevq[q0 + lane];
      nq = a.next[q0 + lane];
    }
    double sP = 1.0, sw = 1.0;
    if constexpr (!ADA) {
      sP = a.scales[2 * pos];
      sw = a.scales[2 * pos + 1];
    }
    const double itp = (double)(it - 1);
    const double tmpP = O.eta0 * itp * O.beta;
    auto pending = [&]() {
      const int64_t v = (int64_t)pq - a.seg0;
      return v >= 0 && cnt[v & (W - 1)] <= (unsigned)(v >> lgW);
    };
    compiler_fence();
    // entries whose previous sample is not known to be finished: a NEAR one (fewer than W positions back) left the recipe
    // of the shared row in its forwarding area; a far one is waited for by its counter
    const bool pend = pending();
    // One-term window, SGD: the shared row of the MOST RECENT earlier sample inside the window is treated as AFFINE in that
    // sample's dL (post_sum) -- chosen by POSITIONS alone (never by what happens to be finished when this worker looks), so that
    // the arithmetic, and with it every bit of the result, is the same from run to run; provided that writer shares exactly
    // one feature with this sample (two rows moving with the same dL would make S quadratic in it).  The earlier writers' dL
    // come first (the conductor makes them in order) and are waited for.
    int aff_q = -1;
    ull aff_bit = 0ull;
    if constexpr (!ADA) {
      if (a.one_term) {
        const int64_t vq_ = (int64_t)pq - a.seg0;
        const bool cand = vq_ >= 0 && (pos - (int64_t)pq) < a.near_r;
        int latest = cand ? (int)vq_ : -1;
#pragma unroll
        for (int sh = 1; sh < kWave; sh <<= 1) {
          const int o_ = __shfl_xor(latest, sh, kWave);
          latest = o_ > latest ? o_ : latest;
        }
        const ull who = __ballot(cand && (int)vq_ == latest);
        if (who != 0ull && (who & (who - 1)) == 0ull) {
          aff_q = __builtin_ctzll(who);
          aff_bit = who;
        }
      }
    }
    // (one-term SGD: EVERY earlier sample inside the window takes the recipe path, finished or not -- a recipe may carry its
    // per-factor sums as a function of an earlier dL, and the row formed from it then differs in the last bit from the one in
    // memory: which of the two is used must not depend on timing)
    const bool near_pos = (int64_t)pq - a.seg0 >= 0 && (pos - (int64_t)pq) < a.near_r;
    const bool near = (!ADA && a.one_term) ? near_pos : (pend && near_pos);
    const ull fwdmask = __ballot(near), farmask = __ballot(pend && !near), latemask = fwdmask | farmask;
    if (a.trace && lane == 0) a.trace[u * 8 + 0] = wall_clock64();  // sample taken up
    const bool late = (latemask >> lane) & 1ull;
    // the next sample with one of this sample's features within W positions will ask for that row's recipe
    const ull hotmask = __ballot(nq >= 0 && ((int64_t)nq - pos) < a.near_r);
    // step sizes first: nothing below waits for them
    const double h2 = ADA ? O.eta0 * itp * O.alpha0 : dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, itf);
    double eta_w = 0.0, eta_P = 0.0, sPn = 1.0, swn = 1.0;
    if constexpr (!ADA) {
      eta_w = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf);
      eta_P = dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf);
      sPn = sP * (1 - eta_P * O.beta);
      swn = sw * (1 - eta_w * O.alpha);
    }
    const double ry = 1.0 / sPn;
    const bool odd_divisor = (__double_as_longlong(sPn) & 0xFFFFFFFFFFFFFll) == 0xFFFFFFFFFFFFFll;

    // ---- A. all rows requested right away (the late ones are replaced below) ----
    double wv = 0.0, gwr = 0.0, nwr = 0.0;
    // AdaGrad update() of one row from (stored P, g_sum, g_norm): adagrad.nim:87-100
    auto ada_row = [&](double pold, double g, double n, size_t e, bool ok) {
      double p = pold;
      if (it != 1) {
        p = dev::adagrad_param(g, n, O.eta0, tmpP);
        viol_acc = ok ? viol_acc + fabs(pold - p) : viol_acc;
        if (ok) st_f64(M.P + e, p);
      }
      return p;
    };
    auto ada_lin = [&](double wold, double gw, double nw_) {  // fit_linear.nim:50-57
      double nv = wold;
      if (it != 1) {
        const double denom = itp * O.eta0 * O.alpha;
        nv = -O.eta0 * gw / (denom + sqrt(nw_));
        viol_acc += fabs(wold - nv);
        st_f64(M.w + jq, nv);
      }
      return nv;
    };
    if constexpr (!ADA) {
      // no branch per row anywhere on the hot path: a row past the sample's end is feature 0's row with value 0 (its terms
      // are +0.0: they change no sum), its store goes to a scratch row -- per-row uniform branches made the compiler keep 64
      // predicates as spilled masks and wait for ALL memory at every row
#pragma unroll
      for (int q = 0; q < K; ++q) Pr[q] = ld_f64(M.P + (size_t)readlane_i(jq, q) * K + lane);
      if (lane < m) wv = ld_f64(M.w + jq);
    } else {
      // Software-pipelined over groups of eight rows: the next group's 24 loads (stored P, g_sum, g_norm) are in flight while
      // this group's square roots and divisions run -- group by group, each waited a round trip of its own (8 x 2 us of the
      // worker's 34 us per sample).  No branch per group: rows past the sample's end read entry 0's row and store nothing.
      double p_[2][8], g_[2][8], n_[2][8];
      auto row_of = [&](int q) { return (size_t)readlane_i(jq, q < m ? q : 0) * K + lane; };
      auto issue = [&](int buf, int qb) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const size_t e = row_of(qb + t);
          p_[buf][t] = ld_f64(M.P + e);
          g_[buf][t] = ld_f64(O.G + e);
          n_[buf][t] = ld_f64(O.N + e);
        }
      };
      issue(0, 0);
#pragma unroll
      for (int qb = 0; qb < K; qb += 8) {
        const int cur = (qb >> 3) & 1;
        if (qb + 8 < K) issue(cur ^ 1, qb + 8);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const bool ok = qb + t < m && !((latemask >> (qb + t)) & 1ull);  // a late row is done again below
          Pr[qb + t] = ada_row(p_[cur][t], g_[cur][t], n_[cur][t], row_of(qb + t), ok);  // (past the end: a number, times x = 0)
          Gl[(qb + t) * K + lane] = g_[cur][t];
          Nl[(qb + t) * K + lane] = n_[cur][t];
        }
      }
      if (lane < m) {
        wv = ld_f64(M.w + jq);
        if (M.fit_linear) {
          gwr = ld_f64(O.Gw + jq);
          nwr = ld_f64(O.Nw + jq);
          if (!late) wv = ada_lin(wv, gwr, nwr);
        }
      }
    }
    // ---- B. far dependencies: wait for the counters, read those rows again ----
    if (farmask) {
      Spin sp;
      while (true) {
        for (int l = lane; l < W; l += kWave) cnt[l] = ld_u32(a.completed + l);
        compiler_fence();
        if (!__any(pend && !near && pending())) break;
        if (sp.wait(a.ctrl)) return;
      }
      for (ull mk = farmask; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        const size_t e = (size_t)readlane_i(jq, q) * K + lane;
        Fl[q * K + lane] = ld_f64(M.P + e);
        if constexpr (ADA) {
          Gl[q * K + lane] = ld_f64(O.G + e);
          Nl[q * K + lane] = ld_f64(O.N + e);
        }
      }
      if ((farmask >> lane) & 1ull) {
        wv = ld_f64(M.w + jq);
        if (ADA && M.fit_linear) {
          gwr = ld_f64(O.Gw + jq);
          nwr = ld_f64(O.Nw + jq);
        }
      }
    }
    // ---- C. near dependencies: the writer's recipe + the conductor's dL for the writer's sample -> the row as the writer
    // will (or did) write it, formed here with the writer's own arithmetic ----
    // One-term window, SGD, ONE such row whose writer's dL is not there yet: the row is affine in that dL (A + dL B), the
    // sample's sum S with it (post_sum): S(A) and the slope are posted at once, the exact row is formed when dL arrives
    bool affine = false;  // (uniform)
    double aff_A = 0.0, aff_B = 0.0, aff_B1 = 0.0, aff_Bw = 0.0, aff_a1u = 0.0, aff_pv = 0.0, aff_tu = 0.0;
    int aff_dw = 0;          // the affine writer's OWN affine writer, as a distance from it (its recipe: a1 = a1u + dL_w' tu); 0: none
    bool posted_early = false;  // this sample's recipes went out before its second half (a1 as a function of its writer's dL)
    const unsigned mytag = (unsigned)(u + 1);
    const int q_first = fwdmask ? __builtin_ctzll(fwdmask) : 0;
    // the writers' side of the recipes (kept for the affine path's second half)
    const bool mine = (fwdmask >> lane) & 1ull;
    const int64_t upl = (int64_t)pq - a.seg0;
    const unsigned tagl = (unsigned)(upl + 1);
    double wu = 0.0, gwu = 0.0, nwu = 0.0, vsl = 0.0, dLl = 0.0;
    double a1u = 0.0, pv = 0.0, gv = 0.0, nv = 0.0, tu = 0.0, dwu = 0.0;
    double sPul = 1.0, etaPul = 0.0, sPnul = 1.0, swul = 1.0, etawul = 0.0, rynul = 1.0;
    ull* srcl = fw_area(mine ? upl : 0);
    const ull* rsrcl = res_of(mine ? upl : 0);
    bool okl = true, okr = true;
    int q_loaded = -1;  // the shared row whose recipe sits in (a1u, pv, gv, nv)
    auto load_lin = [&]() {
      okl = true;
      if (mine) {
        okl = fw_load(fw_lin(srcl, 0, pqu), tagl, wu);
        if (ADA && M.fit_linear) {
          okl = fw_load(fw_lin(srcl, 1, pqu), tagl, gwu) && okl;
          okl = fw_load(fw_lin(srcl, 2, pqu), tagl, nwu) && okl;
        }
        okl = fw_load(fw_lin(srcl, 3, pqu), tagl, vsl) && okl;
      }
    };
    auto load_row = [&](int q) {
      const int64_t up = (int64_t)readlane_i(pq, q) - a.seg0;
      const unsigned tag = (unsigned)(up + 1);
      ull* src = fw_area(up);
      const int qu = readlane_i(pqu, q);
      okr = fw_load(src + (size_t)lane * 2, tag, a1u);
      okr = fw_load(fw_row(src, 0, qu), tag, pv) && okr;
      if constexpr (!ADA) {  // (the recipe's affine part: slope of a1 in its writer's own affine writer's dL, that writer's distance)
        okr = fw_load(src + (size_t)(K + lane) * 2, tag, tu) && okr;
        okr = fw_load(fw_lin(src, 4, 0), tag, dwu) && okr;
      }
      if constexpr (ADA) {
        okr = fw_load(fw_row(src, 1, qu), tag, gv) && okr;
        okr = fw_load(fw_row(src, 2, qu), tag, nv) && okr;
      }
      q_loaded = q;
    };
    // the exact rows of the near dependencies in `which`, once their writers' dL are in hand (lane `mine`: dLl)
    auto form_near_rows = [&](ull which) -> bool {
      for (ull mk = which; mk; mk &= mk - 1) {
        const int q = __builtin_ctzll(mk);
        const double vsu = readlane_d(vsl, q), dLu = readlane_d(dLl, q);
        if (q != q_loaded) load_row(q);
        Spin sp;
        while (!__all(okr)) {
          if (sp.wait(a.ctrl)) return false;
          load_row(q);
        }
        double a1x = a1u;
        if constexpr (!ADA) {
          if (dwu != 0.0) {  // an affine recipe: the writer's per-factor sums with ITS writer's dL (which exists: it is earlier)
            const int64_t uw = (int64_t)readlane_i(pq, q) - a.seg0 - (int64_t)dwu;
            const ull* rw = res_of(uw);
            double dLw;
            Spin spw;
            while (!__all(fw_load(rw, (unsigned)(uw + 1), dLw)))
              if (spw.wait(a.ctrl)) return false;
            a1x = a1u + dLw * tu;
          }
        }
        if constexpr (ADA) {  // the writer's updateG of this row (adagrad.nim:113-134)
          const double grad = dLu * (vsu * (a1u - pv * vsu));
          Fl[q * K + lane] = pv;
          Gl[q * K + lane] = gv + grad;
          Nl[q * K + lane] = nv + grad * grad;
        } else {  // the writer's update() of this row (sgd.nim:217-223), with ITS scale and step size
          const double sPu = readlane_d(sPul, q), etaPu = readlane_d(etaPul, q), sPnu = readlane_d(sPnul, q);
          const double p = sPu * pv;
          const double update = etaPu * (dLu * (vsu * (a1x - p * vsu)) + O.beta * p);
          const bool oddu = (__double_as_longlong(sPnu) & 0xFFFFFFFFFFFFFll) == 0xFFFFFFFFFFFFFll;
          Fl[q * K + lane] = oddu ? (p - update) / sPnu : div_by(p - update, sPnu, readlane_d(rynul, q));
        }
      }
      if (mine && ((which >> lane) & 1ull)) {
        wv = wu;
        if (M.fit_linear) {
          if constexpr (ADA) {
            const double gg = dLl * vsl;
            gwr = gwu + gg;
            nwr = nwu + gg * gg;
          } else {  // fit_linear.nim:41-47 with the writer's scale and step size
            const double wj = swul * wu;
            wv = (wj - etawul * (dLl * vsl + O.alpha * wj)) / (swul * (1 - etawul * O.alpha));
          }
        }
      }
      return true;
    };
    if (fwdmask) {
      // Everything the recipes hold was posted at the writers' forward passes, long ago: it is requested FIRST (the
      // shared entries' lanes: their entry's part; all lanes: one shared row's part), the poll for the writers' dL
      // runs while those loads are in flight, and only then are the recipes' tags looked at.
      const int q_pre = aff_q >= 0 ? aff_q : q_first;
      load_lin();
      load_row(q_pre);
      // the writers' scales and step sizes (functions of their step counters alone) while those loads are in flight
      if constexpr (!ADA) {
        if (mine) {
          sPul = a.scales[2 * (a.seg0 + upl)];
          swul = a.scales[2 * (a.seg0 + upl) + 1];
          etaPul = dev::get_eta(O.sched, O.eta0, O.power, O.beta, (double)(a.it0 + upl));
          etawul = dev::get_eta(O.sched, O.eta0, O.power, O.alpha, (double)(a.it0 + upl));
          sPnul = sPul * (1 - etaPul * O.beta);
        }
      }
      rynul = 1.0 / sPnul;
      {
        // one loop: the dL of the writers, and -- as long as they do not carry the writers' tags yet (this worker may have
        // arrived before the writers' forward passes) -- the recipes again, so that they are in hand when dL lands
        Spin sp;
        while (true) {
          bool ok = true;
          if (mine) ok = fw_load(rsrcl, tagl, dLl);
          if (!__all(okl)) load_lin();
          if (!__all(okr)) load_row(q_pre);
          // every dL but the affine writer's is here, and so is that writer's recipe: S as a function of its dL -- whether or
          // not that dL exists already (the same arithmetic every run)
          if (aff_q >= 0) {
            if (__all(ok || lane == aff_q) && __all(okl) && __all(okr)) {
              affine = true;
              break;
            }
          } else if (__all(ok)) {
            break;
          }
          if (sp.wait(a.ctrl)) return;
        }
      }
      if (a.trace && lane == 0) a.trace[u * 8 + 0] = wall_clock64();  // (tracing: the writers' dL seen)
      if (!affine) {
        {
          Spin sp;
          while (!__all(okl)) {
            if (sp.wait(a.ctrl)) return;
            load_lin();
          }
        }
        if (!form_near_rows(fwdmask)) return;
      } else {
        if constexpr (!ADA) {
          // row'(dL) = (p - eta (dL g + beta p)) / s' = A + dL B  with the writer's scale, step size and per-factor sums
          const int q = aff_q;
          const double vsu = readlane_d(vsl, q);
          const double sPu = readlane_d(sPul, q), etaPu = readlane_d(etaPul, q), sPnu = readlane_d(sPnul, q);
          const double p = sPu * pv;
          aff_A = (p - etaPu * (O.beta * p)) / sPnu;
          aff_B = -(etaPu * (vsu * (a1u - p * vsu))) / sPnu;
          aff_a1u = a1u;  // (kept: the writer's forwarding area may be reused by the time this sample's second half runs)
          aff_pv = pv;
          aff_tu = tu;
          aff_dw = (int)dwu;
          // the writer's recipe is itself affine in ITS writer's dL (a1 = a1u + dL_w' tu): B = B0 + dL_w' B1
          aff_B1 = aff_dw != 0 ? -(etaPu * (vsu * tu)) / sPnu : 0.0;
          if (!form_near_rows(fwdmask & ~aff_bit)) return;  // (the earlier writers' rows: exact)
          Fl[q * K + lane] = aff_A;
          if (lane == q) {  // the shared feature's linear weight the same way (fit_linear.nim:41-47): Aw + dL Bw
            wv = wu;
            if (M.fit_linear) {
              const double wj = swul * wu, den = swul * (1 - etawul * O.alpha);
              wv = (wj - etawul * (O.alpha * wj)) / den;
              aff_Bw = -(etawul * vsl) / den;
            }
          }
        }
      }
    }
    auto apply_late = [&](ull which) {
      compiler_fence();
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if (q < m && ((which >> q) & 1ull)) {
          if constexpr (ADA) Pr[q] = ada_row(Fl[q * K + lane], Gl[q * K + lane], Nl[q * K + lane], (size_t)readlane_i(jq, q) * K + lane, true);
          else Pr[q] = Fl[q * K + lane];
        }
      }
    };
    if (latemask) {
      apply_late(latemask);
      if (ADA && M.fit_linear && late) wv = ada_lin(wv, gwr, nwr);
    }
    if (a.trace && lane == 0)  // dependencies resolved (x 16) + which way: 1 far wait, 2 near rows, 4 waited for a writer's dL, 8 affine
      a.trace[u * 8 + 1] = wall_clock64() * 16 + ((affine ? 8 : 0) + ((fwdmask & ~aff_bit) ? 4 : 0) + (fwdmask ? 2 : 0) + (farmask ? 1 : 0));
    // ---- D. per-factor sums over the entries in storage order, their sum over the factors in ascending order ----
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int q = 0; q < K; ++q) {  // (entries past the end have value 0: + 0.0)
      const double t = readlane_d(vq, q) * (sP * Pr[q]);
      a1 += t;
      a2 += t * t;
    }
    const double kv = (a1 * a1 - a2) / 2;
    red[lane] = lane < k ? kv : 0.0;  // (never -0.0: adding the padding changes nothing)
    compiler_fence();
    // what the update needs of a row (scale x stored value, the entry's value, its address) is formed again there: kept
    // alive from here, 64 rows' worth of it would not fit the registers
#pragma unroll
    for (int q = 0; q < K; ++q) asm volatile("" : "+v"(Pr[q]));
    asm volatile("" : "+v"(vq), "+v"(jq));
    double tot = 0.0;
#pragma unroll
    for (int tb = 0; tb < K / 2; tb += 8) {  // 16 values requested together (the rows keep 128 registers)
      double2 r_[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) r_[t] = *reinterpret_cast<const double2*>(red + 2 * (tb + t));
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        tot += r_[t].x;
        tot += r_[t].y;
      }
    }
    double dL_own = 0.0, yh_own = 0.0;
    if (a.no_cond || a.one_term) {
      // no conductor (fitIntercept = false): predictWithGrad's chain (sgd.nim:193-201) in this wavefront -- the constant
      // intercept, the entries' terms w_j x_j in storage order (lane q holds entry q's), the interaction sum -- then dloss,
      // and {dL, yhat} posted as tagged granules exactly as the conductor would (near successors read them).
      // One-term window: the same sum WITHOUT the intercept (from 0.0) is the sample's one mailbox term.
      const double term = lane < m ? (sw * wv) * vq : -0.0;
      double yh_ = a.no_cond ? b_const : 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) yh_ += readlane_d(term, q);  // (past the row's end: -0.0, which changes no sum)
      yh_ += tot;
      if (a.no_cond) {
        yh_own = yh_;
        dL_own = dev::loss_grad(O.loss, O.loss_param, y, yh_);
        if (lane < kResWords) {
          const double v = lane < 2 ? dL_own : yh_own;
          const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
          st_u64(a.res + ((size_t)(u & (W - 1)) * a.np + (size_t)((u >> lgW) & (a.np - 1))) * kResWords + lane, ((ull)mytag << 32) | (ull)half);
        }
      } else if (!affine) {
        post_sum(a.fwd, a.np, W, slot, par, mytag, lane, yh_, y, h2);
      } else {
        // dS / d dL_writer: the kernel is multilinear in the rows, so the slope is  sum_s x sP B_s (a1_s - x sP A_s)  -- the
        // per-factor sums WITHOUT the shared row's own term -- plus the linear term's  sw Bw x
        const double xs = readlane_d(vq, aff_q);
        const double rest = a1 - xs * (sP * aff_A);
        double c1 = dev::wave_sum(lane < k ? (xs * (sP * aff_B)) * rest : 0.0);
        c1 += readlane_d((sw * aff_Bw) * vq, aff_q);
        const unsigned d1 = (unsigned)(u - ((int64_t)readlane_i(pq, aff_q) - a.seg0));  // (the writer's distance: 1 ... near_r - 1 <= 127)
        double c12 = 0.0;
        unsigned dists = d1;
        if (aff_dw != 0) {  // S = S(A) + dL_u (C1 + dL_w' C12): the writer's writer, d1 + its distance back (<= 254: still in the conductor's ring)
          c12 = dev::wave_sum(lane < k ? (xs * (sP * aff_B1)) * rest : 0.0);
          dists |= (d1 + (unsigned)aff_dw) << 8;
        }
        post_sum(a.fwd, a.np, W, slot, par, mytag, lane, yh_, y, h2, c1, 0.0, c12, dists);
        // This sample's own recipes NOW, with a1 as a function of the writer's dL -- a1 = a1(A) + dL_u T, T_s = x sP B_s -- so that
        // its successors need not wait for that dL either (tools/seqwin_profile.py: what the conductor idled for were samples
        // waiting for nothing but such a recipe).  Only when S is affine in ONE dL (a recipe that is bilinear already has no
        // place in this format) and the affine row is not itself one a successor asks for.
        if (aff_dw == 0 && hotmask && (hotmask & aff_bit) == 0ull) {
          posted_early = true;
          size_t fwo = ((size_t)(u & (W - 1)) * a.np + (size_t)((u >> lgW) & (a.np - 1))) * kFwSlot;
          asm volatile("" : "+s"(fwo));
          ull* fwm = a.fw + fwo;
          fw_store(fwm + (size_t)lane * 2, mytag, a1);
          fw_store(fwm + (size_t)(K + lane) * 2, mytag, lane < k ? xs * (sP * aff_B) : 0.0);
          if (lane == 0) fw_store(fw_lin(fwm, 4, 0), mytag, (double)d1);
#pragma unroll
          for (int q = 0; q < K; ++q) {
            if (q < m && ((hotmask >> q) & 1ull)) {
              size_t fqo = ((size_t)(u & (W - 1)) * a.np + (size_t)((u >> lgW) & (a.np - 1))) * kFwSlot;
              asm volatile("" : "+s"(fqo));
              fw_store(fw_row(a.fw + fqo, 0, q), mytag, Pr[q]);
            }
          }
          if ((hotmask >> lane) & 1ull) {
            fw_store(fw_lin(fwm, 0, lane), mytag, wv);
            fw_store(fw_lin(fwm, 3, lane), mytag, vq);
          }
        }
        // ... and now the exact row: the writer's dL, its update of the shared row, this sample's sums again with it
        {
          Spin sp;
          while (true) {
            bool ok = true;
            if (lane == aff_q) ok = fw_load(rsrcl, tagl, dLl);
            if (__all(ok)) break;
            if (sp.wait(a.ctrl)) return;
          }
        }
        a1u = aff_a1u;  // (the recipe as read before the mailbox was posted)
        pv = aff_pv;
        tu = aff_tu;
        dwu = (double)aff_dw;
        q_loaded = aff_q;
        okr = true;
        if (!form_near_rows(aff_bit)) return;
        apply_late(aff_bit);
        a1 = 0.0;
        a2 = 0.0;
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const double t = readlane_d(vq, q) * (sP * Pr[q]);
          a1 += t;
          a2 += t * t;
        }
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("" : "+v"(Pr[q]));
        asm volatile("" : "+v"(vq), "+v"(jq));
      }
    } else {
      if (lane < MC) st_u64(mb + lane, mail_bits(lane < m ? (sw * wv) * vq : -0.0));
      if (lane < kWinHdr) st_u64(mb + MC + lane, mail_bits(lane == 0 ? tot : lane == 1 ? y : lane == 2 ? h2 : (double)m));
    }
    if (a.trace && lane == 0) a.trace[u * 8 + 2] = wall_clock64();  // mailbox posted
    // (after the mailbox: a successor needs them together with this sample's dL, which the conductor forms from the mailbox)
    if (hotmask && !posted_early) {  // the recipes of the rows a near successor shares (the base is made opaque per use: left to itself the
                    // compiler hoists the 192 row addresses out of the sample loop and spills them)
      // (an opaque OFFSET, not an opaque pointer: the address must stay provably global -- sc1 through flat_ instructions
      // is not the hand-off this kernel relies on)
      size_t fwo = ((size_t)(u & (W - 1)) * a.np + (size_t)((u >> lgW) & (a.np - 1))) * kFwSlot;
      asm volatile("" : "+s"(fwo));
      ull* fwm = a.fw + fwo;
      fw_store(fwm + (size_t)lane * 2, mytag, a1);
      if constexpr (!ADA) {  // (an exact recipe: no slope, no writer)
        fw_store(fwm + (size_t)(K + lane) * 2, mytag, 0.0);
        if (lane == 0) fw_store(fw_lin(fwm, 4, 0), mytag, 0.0);
      }
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if (q < m && ((hotmask >> q) & 1ull)) {
          size_t fqo = ((size_t)(u & (W - 1)) * a.np + (size_t)((u >> lgW) & (a.np - 1))) * kFwSlot;
          asm volatile("" : "+s"(fqo));
          ull* fq_ = a.fw + fqo;
          fw_store(fw_row(fq_, 0, q), mytag, Pr[q]);
          if constexpr (ADA) {
            fw_store(fw_row(fq_, 1, q), mytag, Gl[q * K + lane]);
            fw_store(fw_row(fq_, 2, q), mytag, Nl[q * K + lane]);
          }
        }
      }
      if ((hotmask >> lane) & 1ull) {
        fw_store(fw_lin(fwm, 0, lane), mytag, wv);
        if constexpr (ADA) {
          fw_store(fw_lin(fwm, 1, lane), mytag, gwr);
          fw_store(fw_lin(fwm, 2, lane), mytag, nwr);
        }
        fw_store(fw_lin(fwm, 3, lane), mytag, vq);
      }
    }

    // ---- E. {dL, yhat} from the conductor: tagged granules (a near successor reads them too) ----
    // E2 rides along: rows formed from a near writer's recipe are stored only after that writer's own stores have landed (see
    // win_worker) -- the lanes with such an entry ask for their writer's completion counter in the same rounds that poll for
    // this sample's dL, so the answer is normally there when dL is (a separate poll cost the headline shape 13 %)
    const int64_t vnear = (int64_t)pq - a.seg0;
    bool wdone = !near;
    auto ask_writer = [&]() {
      if (!wdone) wdone = ld_u32(a.completed + (vnear & (W - 1))) > (unsigned)(vnear >> lgW);
    };
    double dL, yh;
    if (a.no_cond) {
      dL = dL_own;
      yh = yh_own;
    } else {
      const ull* rp = res_of(u);
      Spin sp;
      double rd;
      while (true) {
        const bool ok = fw_load(rp + (size_t)(lane & 1) * 2, mytag, rd);
        ask_writer();
        if (__all(ok)) break;
        if (sp.wait(a.ctrl)) return;
      }
      dL = readlane_d(rd, 0);
      yh = readlane_d(rd, 1);
    }
    if (a.trace && lane == 0) a.trace[u * 8 + 3] = wall_clock64();  // dL received
    if (fwdmask) {
      Spin sp;
      while (true) {
        ask_writer();
        if (__all(wdone)) break;
        if (sp.wait(a.ctrl)) return;
      }
    }
    // ---- F. update(): sgd.nim:205-243 / updateG(): adagrad.nim:113-134 ----
    if (!ADA && odd_divisor) {
      // a divisor with an all-ones significand (one sample in 2^52): the division instruction, rows through LDS
#pragma unroll
      for (int q = 0; q < K; ++q) Fl[q * K + lane] = Pr[q];
      compiler_fence();
      for (int q = 0; q < m; ++q) {
        const size_t e = (size_t)readlane_i(jq, q) * K + lane;
        const double vs = readlane_d(vq, q);
        const double p = sP * Fl[q * K + lane];
        const double update = eta_P * (dL * (vs * (a1 - p * vs)) + O.beta * p);
        viol_acc += fabs(update);
        st_f64(M.P + e, (p - update) / sPn);
      }
    } else {
      double* const junk = reinterpret_cast<double*>(a.fw + (size_t)a.np * W * kFwSlot) + (size_t)slot * 2 * K + lane;  // scratch rows
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const bool in = q < m;
        const size_t e = (size_t)readlane_i(jq, q) * K + lane;
        const double vs = readlane_d(vq, q);
        const double p = sP * Pr[q];
        const double d_ = vs * (a1 - p * vs);
        if constexpr (ADA) {
          const double grad = dL * d_;
          st_f64_at(in ? (ull)(O.G + e) : (ull)junk, Gl[q * K + lane] + grad);
          st_f64_at(in ? (ull)(O.N + e) : (ull)(junk + K), Nl[q * K + lane] + grad * grad);
        } else {
          const double update = eta_P * (dL * d_ + O.beta * p);
          viol_acc += in ? fabs(update) : 0.0;
          st_f64_at(in ? (ull)(M.P + e) : (ull)junk, div_by(p - update, sPn, ry));
        }
      }
    }
    if (M.fit_linear && lane < m) {
      if constexpr (ADA) {
        const double gg = dL * vq;
        st_f64(O.Gw + jq, gwr + gg);
        st_f64(O.Nw + jq, nwr + gg * gg);
      } else {
        const double wj = sw * wv;
        const double update = eta_w * (dL * vq + O.alpha * wj);
        viol_acc += fabs(update);
        st_f64(M.w + jq, (wj - update) / swn);
      }
    }
    // the mailbox back to "empty" for its next use, two samples of this worker from now: these stores have completed
    // (vmcnt(0) below) before this worker posts its next sample, which the conductor consumes before it can look at
    // these words again (kWinDepth < W)
    if (!a.no_cond && !a.one_term) {
      if (lane < a.FW) st_u64(mb + lane, kWinSentinel);
      if (lane < a.FW - kWave) st_u64(mb + kWave + lane, kWinSentinel);
    }
    if (lane == 0) loss_acc += dev::loss_value(O.loss, O.loss_param, y, yh);
    // ---- G. rows written: tell the far waiters ----
    // (no conductor: the other workers' completion counters requested here ride along with this wait -- the run-ahead
    // check of the next sample then finds them in LDS instead of paying a round trip of its own)
    unsigned cr0_ = 0u, cr1_ = 0u, cr2_ = 0u, cr3_ = 0u;
    if (a.no_cond) {
      if (lane < W) cr0_ = ld_u32(a.completed + lane);
      if (kWave + lane < W) cr1_ = ld_u32(a.completed + kWave + lane);
      if (2 * kWave + lane < W) cr2_ = ld_u32(a.completed + 2 * kWave + lane);
      if (3 * kWave + lane < W) cr3_ = ld_u32(a.completed + 3 * kWave + lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.no_cond) {
      if (lane < W) cnt[lane] = cr0_;
      if (kWave + lane < W) cnt[kWave + lane] = cr1_;
      if (2 * kWave + lane < W) cnt[2 * kWave + lane] = cr2_;
      if (3 * kWave + lane < W) cnt[3 * kWave + lane] = cr3_;
    }
    if (lane == 0) st_u32(a.completed + slot, (unsigned)(u >> lgW) + 1u);
    if (a.trace && lane == 0) a.trace[u * 8 + 4] = wall_clock64();  // rows written
  }
  viol_acc = dev::wave_sum(viol_acc);
  if (lane == 0) {
    a.partial[2 * slot] = loss_acc;
    a.partial[2 * slot + 1] = viol_acc;
  }
}

enum { WK_GENERAL = 0, WK_K64 = 1, WK_FFM = 2, WK_FMX = 3 };  // which worker
template <int OPT, int CH, int WK>
__global__ __launch_bounds__(kFfmWaves * kWave) void k_seq_window(WinArgs a) {
  extern __shared__ double lds[];
  const int nb0 = a.first_worker;
#ifdef NFM_TEST_HOOKS
  if (a.dead_slot >= 0 && (int)blockIdx.x == a.dead_slot + nb0) return;
#endif
  if ((int)blockIdx.x < nb0) {
    if (threadIdx.x < (a.one_term ? 3 : 2) * kWave) {  // (one-term: fetch, chain and post wavefronts)
      if (a.one_term) {
        if (a.W == 2 * kWave) win_conductor_sum<OPT, 2>(a, lds);
        else win_conductor_sum<OPT, 1>(a, lds);
      } else {
        win_conductor<OPT, CH>(a, lds);
      }
    } else {
      __syncthreads();  // (the conductor's one barrier: the workgroup's idle wavefronts pass it and leave)
    }
  } else if constexpr (WK == WK_FFM) {
    win_worker_ffm<OPT>(a, (int)blockIdx.x - nb0, lds);  // all wavefronts of the workgroup
  } else if constexpr (WK == WK_FMX) {
    win_worker_fmx<OPT>(a, (int)blockIdx.x - nb0, lds);
  } else if (threadIdx.x < kWave) {
    if constexpr (WK == WK_K64) win_worker_k64<OPT>(a, (int)blockIdx.x - nb0, lds);
    else win_worker<OPT>(a, (int)blockIdx.x - nb0, lds);
  }
}

// loss / viol of the launch: the workers' partial sums in worker order, then the conductor's
__global__ void k_win_finish(const double* partial, int W, double* out) {
  if (threadIdx.x == 0) {
    double l = 0.0, v = 0.0;
    for (int s = 0; s <= W; ++s) {
      l += partial[2 * s];
      v += partial[2 * s + 1];
    }
    out[0] += l;
    out[1] += v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// SGD: the lazy-scaling chain (sgd.nim:233-234, 116-131) ahead of the launch: scale *= 1 - eta(it) reg, one sample after the
// other (lane 0 of the chain wavefront: scale_P, lane 1: scale_w) -- 2e6 dependent multiplications per 2e6-sample call.  Three
// wavefronts in a pipeline over blocks of 64 samples (round 5; one wavefront did everything at 32 ns per sample, 8 % of an
// exact-order epoch): wavefront 1 forms the factors of block i + 1 (64 step sizes in parallel), wavefront 0 multiplies block i
// through -- nothing else on its path --, wavefront 2 writes block i - 1 out.  The products are formed in the same order as
// before: the same bits.
//   scales[pos] = {scale_P, scale_w} before the sample at `pos`;  info[0] = last position of the stretch (the sample after
//   which a scale falls below 1e-9, or the call's last), info[1] = 1: rescale P, 2: rescale w.  The model's two scale words
//   get the values after that sample.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(3 * 64) void k_win_scales(ModelView M, OptView O, int64_t it0, int64_t pos0, int64_t ns, double* scales,
                                                       int64_t* info) {
  __shared__ double f[2][2][kWave];  // [buffer][chain][t] the factors 1 - eta reg of a block
  __shared__ double o[2][2][kWave];  // [buffer][chain][t] the scale AFTER sample t of a block (before sample t: the one after t - 1)
  __shared__ double cin[2][2];       // [buffer][chain] the scale before the block's first sample
  __shared__ int stp[2][2];          // [buffer][chain]: the first sample of the block after which the chain's scale is below 1e-9, or -1
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  const int64_t nblk = (ns - pos0 + kWave - 1) / kWave;
  const bool fit_linear = M.fit_linear != 0;
  auto factors = [&](int64_t blk) {  // (wavefront 1)
    const double itf = (double)(it0 + blk * kWave + lane);
    f[blk & 1][0][lane] = 1 - dev::get_eta(O.sched, O.eta0, O.power, O.beta, itf) * O.beta;
    // the linear scale moves only when the linear term is fitted (sgd.nim:229-234 as kept by k_sequential_pipe)
    f[blk & 1][1][lane] = fit_linear ? 1 - dev::get_eta(O.sched, O.eta0, O.power, O.alpha, itf) * O.alpha : 1.0;
  };
  auto write_out = [&](int64_t blk, int upto) {  // (wavefront 2)
    if (lane < upto) {
      const int64_t pos = pos0 + blk * kWave + lane;
      scales[2 * pos] = lane == 0 ? cin[blk & 1][0] : o[blk & 1][0][lane - 1];
      scales[2 * pos + 1] = lane == 0 ? cin[blk & 1][1] : o[blk & 1][1][lane - 1];
    }
  };
  double sc = lane == 0 ? M.sc[SC_SCALE_P] : M.sc[SC_SCALE_W];  // (chain wavefront: lane 0 carries scale_P, lane 1 scale_w)
  if (wave == 1 && nblk > 0) factors(0);
  __syncthreads();
  int64_t last = ns - 1;
  int flags = 0;
  for (int64_t blk = 0; blk < nblk; ++blk) {
    const int cnt = (int)(ns - (pos0 + blk * kWave) < kWave ? ns - (pos0 + blk * kWave) : kWave);
    if (wave == 0) {
      if (lane < 2) {
        // the 64 factors into registers at once, then the chain IN PLACE: one multiplication per sample and nothing else on
        // its path (a lone wavefront issues an instruction every ~8 cycles: every copy beside the product costs as much as it)
        double fr[kWave];
#pragma unroll
        for (int t = 0; t < kWave; t += 2) {
          const double2 v = *reinterpret_cast<const double2*>(&f[blk & 1][lane][t]);
          fr[t] = v.x;
          fr[t + 1] = v.y;
        }
        cin[blk & 1][lane] = sc;
        fr[0] = sc * fr[0];
#pragma unroll
        for (int t = 1; t < kWave; ++t) fr[t] = fr[t - 1] * fr[t];  // the value after sample t
#pragma unroll
        for (int t = 0; t < kWave; t += 2) *reinterpret_cast<double2*>(&o[blk & 1][lane][t]) = double2{fr[t], fr[t + 1]};
        const double c = fr[kWave - 1];
        double c_end = c;
        if (cnt < kWave) {  // a short last block: the value after its last sample
#pragma unroll
          for (int t = 0; t < kWave; ++t)
            if (t == cnt - 1) c_end = fr[t];
        }
        // resetScaling after the first sample that leaves the scale below 1e-9 (sgd.nim:116-131).  The factors are <= 1, so
        // the scale never grows: only a block that ENDS below the limit has to be searched.
        int stop = -1;
        if ((lane == 0 || fit_linear) && c_end < 1e-9) {
#pragma unroll
          for (int t = kWave - 1; t >= 0; --t)
            if (t < cnt && fr[t] < 1e-9) stop = t;
        }
        stp[blk & 1][lane] = stop;
        sc = c_end;
      }
    } else if (wave == 1) {
      if (blk + 1 < nblk) factors(blk + 1);
    } else if (blk > 0) {
      write_out(blk - 1, kWave);
    }
    __syncthreads();
    const int s0 = stp[blk & 1][0], s1 = stp[blk & 1][1];
    if (s0 >= 0 || s1 >= 0) {  // (uniform: every wavefront reads the same two words after the barrier)
      const int stop = s0 >= 0 && (s1 < 0 || s0 <= s1) ? s0 : s1;
      last = pos0 + blk * kWave + stop;
      flags = (s0 == stop ? 1 : 0) | (s1 == stop ? 2 : 0);
      if (wave == 2) write_out(blk, stop + 1);
      if (wave == 0 && lane < 2) sc = o[blk & 1][lane][stop];  // both chains: the value after sample `stop`
      break;
    }
    if (blk + 1 == nblk && wave == 2) write_out(blk, cnt);
  }
  if (wave == 0) {
    if (lane == 0) {
      M.sc[SC_SCALE_P] = sc;
      info[0] = last;
      info[1] = flags;
    }
    if (lane == 1) M.sc[SC_SCALE_W] = sc;
  }
}

__global__ void k_win_fill(ull* p, int64_t n, ull v) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = v;
}

__global__ void k_win_rescale(double* p, int64_t n, double* scale_word, const int64_t* info, int bit) {
  if (!(info[1] & bit)) return;
  const double sc = *scale_word;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) p[e] *= sc;
}
__global__ void k_win_rescale_done(double* sc, const int64_t* info) {
  if (info[1] & 1) sc[SC_SCALE_P] = 1.0;
  if (info[1] & 2) sc[SC_SCALE_W] = 1.0;
}

// ------------------------------------------------------------------------------------------------------------------
// the dependency table: for every stored entry of the samples of the call, the position of the previous sample of the
// call with the same feature.  One sort of (feature, position) keys.
// ------------------------------------------------------------------------------------------------------------------
__global__ void k_win_rowlen(CsrView X, const int64_t* perm, int64_t begin, int64_t ns, int64_t* len) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > ns) return;
  if (p == ns) {
    len[p] = 0;
    return;
  }
  const int64_t i = perm ? perm[begin + p] : begin + p;
  len[p] = X.indptr[i + 1] - X.indptr[i];
}
__global__ void k_win_expand(CsrView X, const int64_t* perm, int64_t begin, int64_t ns, const int64_t* off, ull* keys, uint32_t* vals) {
  const int64_t p = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
  if (p >= ns) return;
  const int lane = threadIdx.x & 63;
  const int64_t i = perm ? perm[begin + p] : begin + p;
  const int64_t q0 = X.indptr[i], m = X.indptr[i + 1] - q0, o = off[p];
  for (int64_t q = lane; q < m; q += kWave) {
    keys[o + q] = ((ull)(uint32_t)X.indices[q0 + q] << 32) | (ull)(uint32_t)p;
    vals[o + q] = (uint32_t)(q0 + q);
  }
}
__global__ void k_win_prev(CsrView X, const int64_t* perm, int64_t begin, const ull* keys, const uint32_t* vals, int64_t T,
                           int32_t* prev, uint8_t* prevq, int32_t* next) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= T) return;
  const ull kcur = keys[e];
  int32_t p = -1, nx = -1;
  uint8_t pqu = 0;
  if (e > 0) {
    const ull kp = keys[e - 1];
    if ((kp >> 32) == (kcur >> 32)) {
      p = (int32_t)(uint32_t)(kp & 0xffffffffull);
      const int64_t ip = perm ? perm[begin + p] : begin + p;
      const int64_t qu = (int64_t)vals[e - 1] - X.indptr[ip];  // the feature's place in the previous sample's row
      pqu = (uint8_t)(qu < 256 ? qu : 255);
    }
  }
  if (e + 1 < T) {
    const ull kn = keys[e + 1];
    if ((kn >> 32) == (kcur >> 32)) nx = (int32_t)(uint32_t)(kn & 0xffffffffull);
  }
  const uint32_t at = vals[e];
  prev[at] = p;
  prevq[at] = pqu;
  next[at] = nx;
}

static int build_prev(nfm_ctx* ctx, const CsrView& X, const int64_t* perm_dev, int64_t begin, int64_t ns, SeqWin* sw) {
  hipStream_t st = ctx->stream;
  NFM_TRY(sw->prev.ensure(sizeof(int32_t) * (size_t)(X.nnz > 0 ? X.nnz : 1)));
  NFM_TRY(sw->next.ensure(sizeof(int32_t) * (size_t)(X.nnz > 0 ? X.nnz : 1)));
  NFM_TRY(sw->prevq.ensure((size_t)(X.nnz > 0 ? X.nnz : 1)));
  DevBuf len, off, k0, k1, v0, v1, tmp;
  NFM_TRY(len.alloc(sizeof(int64_t) * (ns + 1)));
  NFM_TRY(off.alloc(sizeof(int64_t) * (ns + 1)));
  hipLaunchKernelGGL(k_win_rowlen, dim3((unsigned)((ns + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, X, perm_dev, begin, ns, len.as<int64_t>());
  size_t bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, len.as<int64_t>(), off.as<int64_t>(), (int)(ns + 1), st));
  NFM_TRY(tmp.alloc(bytes));
  NFM_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, len.as<int64_t>(), off.as<int64_t>(), (int)(ns + 1), st));
  int64_t T = 0;
  NFM_HIP_CHECK(hipMemcpyAsync(&T, off.as<int64_t>() + ns, sizeof(T), hipMemcpyDeviceToHost, st));
  NFM_HIP_CHECK(hipStreamSynchronize(st));
  if (T == 0) return NFM_OK;
  NFM_CHECK(T < ((int64_t)1 << 31), NFM_ERR_UNSUPPORTED, "too many entries for the dependency table");
  NFM_TRY(k0.alloc(sizeof(ull) * T));
  NFM_TRY(k1.alloc(sizeof(ull) * T));
  NFM_TRY(v0.alloc(sizeof(uint32_t) * T));
  NFM_TRY(v1.alloc(sizeof(uint32_t) * T));
  const int per = kBlock / kWave;
  hipLaunchKernelGGL(k_win_expand, dim3((unsigned)((ns + per - 1) / per)), dim3(kBlock), 0, st, X, perm_dev, begin, ns, off.as<int64_t>(),
                     k0.as<ull>(), v0.as<uint32_t>());
  int fbits = 1;
  while (((int64_t)1 << fbits) < X.d + 1 && fbits < 31) ++fbits;
  hipcub::DoubleBuffer<ull> dk(k0.as<ull>(), k1.as<ull>());
  hipcub::DoubleBuffer<uint32_t> dv(v0.as<uint32_t>(), v1.as<uint32_t>());
  bytes = 0;
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (int)T, 0, 32 + fbits, st));
  NFM_TRY(tmp.alloc(bytes));
  NFM_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, dk, dv, (int)T, 0, 32 + fbits, st));
  hipLaunchKernelGGL(k_win_prev, dim3((unsigned)((T + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, X, perm_dev, begin, dk.Current(), dv.Current(), T,
                     sw->prev.as<int32_t>(), sw->prevq.as<uint8_t>(), sw->next.as<int32_t>());
  NFM_HIP_CHECK(hipGetLastError());
  NFM_HIP_CHECK(hipStreamSynchronize(st));  // the temporaries go back to the pool
  return NFM_OK;
}

static int win_ffm_terms(int m_cap) { return m_cap + m_cap * (m_cap - 1) / 2; }
static size_t win_ffm_lds(const ModelView& M, int m_cap, bool ada, int W) {  // the carve-up of win_worker_ffm
  int lgKp = 1;
  while ((1 << lgKp) < M.Kp) ++lgKp;
  const int grp = (kWave >> lgKp) * 4;
  const size_t mcs = ((size_t)m_cap * M.nb + grp - 1) / grp * grp;
  return sizeof(double) * ((ada ? 4 : 2) * mcs * M.Kp + (size_t)m_cap * m_cap + 4 * kWave + kFfmWaves + (ada ? 4 : 2) * (size_t)m_cap) +
         sizeof(ull) * 2 + sizeof(int) * (5 * (size_t)m_cap + M.nb + (size_t)M.nb * m_cap) + sizeof(unsigned) * W + 64;
}

static size_t win_fmx_lds(const ModelView& M, int m_cap, bool ada, int W) {  // the carve-up of win_worker_fmx
  int lgKp = 1;
  while ((1 << lgKp) < M.Kp) ++lgKp;
  const int grp = (kWave >> lgKp) * 4;
  const size_t mcs = ((size_t)m_cap * M.nb + grp - 1) / grp * grp;
  return sizeof(double) * ((ada ? 4 : 2) * mcs * M.Kp + kWave + 4 * kWave + 3 * dev::kMaxDeg * kWave + kFfmWaves + (ada ? 4 : 2) * (size_t)m_cap) +
         sizeof(ull) * 3 + sizeof(int) * 4 * (size_t)m_cap + sizeof(unsigned) * W + 64;
}

// which flavour of the window a model gets (read per call: tests switch the environment)
static bool win_no_cond(const ModelView& M) {  // fitIntercept = false: no scalar chain, no conductor
  const char* env = getenv("NFM_SEQ_WIN_NOCOND");
  return !(env && atoi(env) == 0) && !M.fit_intercept;
}
static bool win_one_term(const ModelView& M) {  // the intercept is fitted: the one-term chain unless NFM_SEQ_WIN_EXACT=1
  const char* env = getenv("NFM_SEQ_WIN_EXACT");
  return !win_no_cond(M) && !(env && atoi(env) != 0);
}

// A degree-2 FM of 65 ... 128 factors is ONE block of 128-double rows: too wide for the window's workers (rows of at most 64
// factors).  The same table read as feature-major blocks of 64 -- row(b, j) = 2 j + b in units of 64 doubles, exactly
// ModelView::row with bs = 1, rs = 2 -- is a two-block model of the same degree (common.h: kc), which the several-orders worker
// takes as it is: no copy, no second layout.  The factors' sum is then formed block by block -- results within rounding of the
// one-sample-in-flight kernel, not bit-equal: NFM_SEQ_WIN_EXACT=1 keeps such models on that kernel.
ModelView seq_window_view(const ModelView& M) {
  const char* exact = getenv("NFM_SEQ_WIN_EXACT");
  if (exact && atoi(exact) != 0) return M;
  if (!(M.kind == NFM_KIND_FM && M.degree == 2 && M.n_aug == 0 && M.Kp == 2 * kWave)) return M;
  // one block of 128-double rows (65 ... 128 factors), or a wide model of one order whose kc blocks of 128 lie feature-major
  // (api.hip: wide_rows): either way a feature's row is nb * 128 contiguous doubles = 2 nb blocks of 64
  const bool one_block = M.nb == 1 && M.kc == 1 && M.bs == M.da && M.rs == 1;
  const bool wide_rows = M.kc > 1 && M.nb == M.kc && M.bs == 1 && M.rs == M.nb;
  if (!one_block && !wide_rows) return M;
  ModelView V = M;
  V.nb = 2 * M.nb;
  V.kc = V.nb;
  V.Kp = kWave;
  V.L = kWave / 2;
  V.k = kWave;
  V.bs = 1;
  V.rs = V.nb;
  return V;
}

bool seq_window_supported(const ModelView& M, int m_cap, int64_t ns, int64_t nnz, int n_cu, bool ada) {
  const char* env = getenv("NFM_SEQ_WIN");  // 0: off, 1 (default): when it pays, 2: whenever possible (read per call: tests switch it)
  const int mode = env ? atoi(env) : 1;
  if (mode == 0) return false;
  if (M.Kp > kWave || M.Kp < 2) return false;
  // a mailbox of one word per chain term bounds the row length only where the term-by-term conductor runs: not without a
  // conductor (fitIntercept = false) and not in the one-term window (the default; launch_sequential_window)
  const bool term_mail = !(win_no_cond(M) || win_one_term(M));
  if (M.kind == NFM_KIND_FFM) {
    // field-aware: one chain term per entry and per pair of entries; all nFields rows of every feature in LDS
    if (M.n_aug != 0 || M.nb < 1 || m_cap < 1) return false;
    if (term_mail && kWinHdr + (win_ffm_terms(m_cap) + 63) / 64 * 64 > kWave * kWinMaxNL) return false;
    if (win_ffm_lds(M, m_cap, ada, 128) > 160 * 1024) return false;  // (AdaGrad keeps g_sum / g_norm of every slot beside the rows)
  } else if (M.kind == NFM_KIND_FM && (M.nb != 1 || M.degree != 2)) {
    // several orders / degree >= 3: one chain term per entry and per order; the rows of every (entry, order) in LDS
    if (M.n_aug != 0 || M.nb < 1 || M.degree < 2 || M.degree > dev::kMaxDeg || m_cap < 1) return false;
    if (term_mail && kWinHdr + (m_cap + M.nb + 63) / 64 * 64 > kWave * kWinMaxNL) return false;
    if (win_fmx_lds(M, m_cap, ada, 128) > 160 * 1024) return false;
  } else {
    if (M.kind != NFM_KIND_FM || M.nb != 1 || M.degree != 2 || M.n_aug != 0) return false;
    if (term_mail && kWinHdr + (m_cap + 63) / 64 * 64 > kWave * kWinMaxNL) return false;
    // (the general worker keeps the sample's rows in LDS)
    if (!(M.Kp == kWave && m_cap <= kWave)) {
      int lgKp_ = 1;
      while ((1 << lgKp_) < M.Kp) ++lgKp_;
      const int grp_ = (kWave >> lgKp_) * 4;
      const size_t mcp_ = (size_t)(m_cap + grp_ - 1) / grp_ * grp_;
      if (sizeof(double) * ((ada ? 4 : 2) * mcp_ * M.Kp + kWave + (ada ? 4 : 2) * mcp_) + sizeof(int) * 3 * mcp_ + sizeof(unsigned) * 256 > 160 * 1024) return false;
    }
  }
  if (nnz >= ((int64_t)1 << 32) || M.d >= ((int64_t)1 << 31) || ns >= ((int64_t)1 << 31)) return false;
  if (n_cu < 17) return false;  // the smallest window: 16 workers + the conductor, one CU each (launch_sequential_window)
  return mode == 2 || ns >= 2048;
}

template <int OPT, int CH, int WK>
static int launch_window_t(nfm_ctx* ctx, const WinArgs& a, size_t lds_bytes) {
  auto kern = k_seq_window<OPT, CH, WK>;
  NFM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  // the W + 1 workgroups wait for each other: every one of them must be resident.  One workgroup per CU (the LDS request
  // sees to that) and W + 1 <= CUs is the launcher's rule; here the kernel itself is asked whether a CU can hold it at all
  constexpr int threads = kFfmWaves * kWave;  // (the workers of degree-2 FMs use the first wavefront; the conductor needs three)
  int per_cu = 0;
  NFM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds_bytes));
  if (per_cu < 1) return NFM_WIN_FALLBACK;
  ctx->timing.acc["seq_window_launch"].launches += 1;  // (counted with timing on or off: the tests ask which kernel ran)
  TimedLaunch tl(ctx, "sequential");
  const unsigned grid = (unsigned)(a.W + a.first_worker);
  if ((int)grid >= ctx->n_cu) {
    // a workgroup on EVERY CU, all waiting for each other: a cooperative launch, which the runtime starts only when the
    // whole grid can be resident at once (the occupancy query above says one workgroup fits a CU; it cannot say the CUs are free)
    WinArgs a_ = a;
    void* params[] = {&a_};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(grid), dim3(threads), params, (unsigned)lds_bytes, ctx->stream);
    if (e == hipSuccess) {
      ctx->timing.acc["seq_window_cooperative"].launches += 1;
      return NFM_OK;
    }
    (void)hipGetLastError();  // (not supported / too large for this device: the plain launch, guarded by its time limits)
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, ctx->stream, a);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}
template <int OPT>
static int launch_window(nfm_ctx* ctx, const WinArgs& a, size_t lds_bytes, int ch, int wk) {
  if (wk == WK_K64) return ch == 32 ? launch_window_t<OPT, 32, WK_K64>(ctx, a, lds_bytes) : launch_window_t<OPT, 64, WK_K64>(ctx, a, lds_bytes);
  if (wk == WK_FFM) return ch == 32 ? launch_window_t<OPT, 32, WK_FFM>(ctx, a, lds_bytes) : launch_window_t<OPT, 64, WK_FFM>(ctx, a, lds_bytes);
  if (wk == WK_FMX) return ch == 32 ? launch_window_t<OPT, 32, WK_FMX>(ctx, a, lds_bytes) : launch_window_t<OPT, 64, WK_FMX>(ctx, a, lds_bytes);
  return ch == 32 ? launch_window_t<OPT, 32, WK_GENERAL>(ctx, a, lds_bytes) : launch_window_t<OPT, 64, WK_GENERAL>(ctx, a, lds_bytes);
}

int launch_sequential_window(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const int64_t* perm_dev,
                             int64_t begin, int64_t end, int64_t it0, int m_cap, double* out2_dev, SeqWin* sw, uint64_t ds_uid,
                             bool perm_is_callers) {
  hipStream_t st = ctx->stream;
  const int64_t ns = end - begin;
  const bool ada = opt_kind == OPT_ADAGRAD;
  if (m_cap < 1) m_cap = 1;
  // worker count: a power of two, one workgroup per CU with one CU left for the conductor
  // fitIntercept = false (degree-2 FMs): no conductor, and nothing but the features ties the samples -- twice the workers
  const bool no_cond = win_no_cond(M);  // (every worker: degree-2 FMs, several orders / degree >= 3, field-aware models)
  // the intercept is fitted: the ONE-TERM chain (the worker adds up its sample's prediction but the intercept) unless
  // NFM_SEQ_WIN_EXACT=1 asks for the reference's term-by-term rounding
  bool one_term = win_one_term(M);
  // with a conductor: 128 workers for the register-resident worker (64-factor rows: headline shape 2.36e6 -> 2.58e6 samples/s,
  // 2.8e6 with the chunk-parallel chain), 64 for the others (cfg2's shape: 2.36e6 against 2.2e6 at 128 -- its samples conflict
  // ten times as often, and every sample in flight is one more writer to wait for)
  const bool k64_shape = M.kind == NFM_KIND_FM && M.nb == 1 && M.degree == 2 && M.Kp == kWave && m_cap <= kWave;
  int W = no_cond || k64_shape ? 128 : 64;
  // rows of 64 factors without a conductor: a worker on EVERY CU (headline shape 5.8e6 -> 7.1e6 samples/s, AdaGrad 3.8e6 -> 5.9e6;
  // shorter rows conflict too often to gain) -- until a launch of this optimizer has aborted once: such a launch needs the
  // whole chip resident, and a tenant that holds a single CU would cost every call its 1 s limit.  (After a second abort
  // nfm_opt_epoch stops offering the window to this optimizer at all.)
  if (no_cond && M.kind == NFM_KIND_FM && M.nb == 1 && M.degree == 2 && M.Kp == kWave && m_cap <= kWave && sw->fallbacks == 0) W = 256;
  if (const char* env = getenv("NFM_SEQ_WIN_W")) W = atoi(env);
  int lgW = 4;
  while ((2 << lgW) <= W && lgW < (no_cond ? 8 : 7)) ++lgW;
  W = 1 << lgW;
  const int extra = no_cond ? 0 : 1;  // the conductor's CU (without a conductor workgroup 0 leaves at once)
  while (W + extra > ctx->n_cu && lgW > 4) W = 1 << --lgW;
  if (!(W > kWinDepth && W + extra <= ctx->n_cu)) return NFM_WIN_FALLBACK;  // (seq_window_supported keeps such devices out)
  int lgKp = 1;
  while ((1 << lgKp) < M.Kp) ++lgKp;
  const bool ffm = M.kind == NFM_KIND_FFM;
  const bool fmx = !ffm && (M.nb != 1 || M.degree != 2);  // several orders / degree >= 3
  const int terms = ffm ? win_ffm_terms(m_cap) : fmx ? m_cap + M.nb : m_cap;  // what a sample hands to the conductor's chain
  const int ch = one_term || terms <= 32 ? 32 : 64;  // the chain's chunk of terms; a mailbox holds MC = a multiple of it
  const int FW = one_term ? kWinHdr + 32 : kWinHdr + (terms + ch - 1) / ch * ch;  // (one-term mailboxes: kSumGran granules, see post_sum)
  const bool k64 = !ffm && !fmx && M.Kp == kWave && m_cap <= kWave;  // the register-resident worker
  const int wk = ffm ? WK_FFM : fmx ? WK_FMX : k64 ? WK_K64 : WK_GENERAL;
  // the previous-position table of this order (kept while the same samples are walked in storage order)
  const bool reuse = sw->valid && !perm_is_callers && !sw->had_perm && sw->ds_uid == ds_uid && sw->begin == begin && sw->end == end && sw->nnz == X.nnz;
  if (!reuse) {
    sw->valid = false;
    TimedLaunch tl(ctx, "seq_window_deps");
    NFM_TRY(build_prev(ctx, X, perm_dev, begin, ns, sw));
    sw->valid = true;
    sw->had_perm = perm_dev != nullptr;
    sw->ds_uid = ds_uid;
    sw->begin = begin;
    sw->end = end;
    sw->nnz = X.nnz;
  }
  // without a conductor the workers may run further apart (see the workers' run-ahead check): four buffer sets per worker
  // (measured: 2 / 4 / 8 sets 6.5e6 / 7.6e6 / 7.6e6 samples/s on the headline shape)
  const int np = no_cond ? 4 : 2;
  const size_t n_fwd = one_term ? (size_t)kSumGran * np * W : (size_t)W * np * FW, n_res = (size_t)W * np * kResWords;
  NFM_TRY(sw->mail.ensure(sizeof(ull) * (n_fwd + n_res)));
  NFM_TRY(sw->ctl.ensure(sizeof(unsigned) * (W + 64) + sizeof(double) * 2 * (W + 1) + sizeof(int64_t) * 2));
  if (!ada) NFM_TRY(sw->scales.ensure(sizeof(double) * 2 * (size_t)ns));
  WinArgs a{};
  a.X = X;
  a.M = M;
  a.O = O;
  a.perm = perm_dev;
  a.begin = begin;
  a.prev = sw->prev.as<int32_t>();
  a.prevq = sw->prevq.as<uint8_t>();
  a.next = sw->next.as<int32_t>();
  NFM_TRY(sw->fw.ensure(sizeof(ull) * (kFwSlot * np + 2 * kWave) * (size_t)W));  // + two scratch rows per worker
  a.fw = sw->fw.as<ull>();
  a.trace = nullptr;
  if (getenv("NFM_SEQ_WIN_TRACE") && atoi(getenv("NFM_SEQ_WIN_TRACE")) != 0) {  // debugging: stamps of the first launch's samples
    NFM_TRY(sw->trace.ensure(sizeof(long long) * 8 * (size_t)ns));
    NFM_HIP_CHECK(hipMemsetAsync(sw->trace.p, 0, sizeof(long long) * 8 * (size_t)ns, st));
    a.trace = sw->trace.as<long long>();
  }
  a.scales = ada ? nullptr : sw->scales.as<double>();
  a.fwd = sw->mail.as<ull>();
  a.res = a.fwd + n_fwd;
  a.completed = sw->ctl.as<unsigned>();
  a.ctrl = a.completed + W;
  a.partial = reinterpret_cast<double*>(a.completed + W + 64);
  int64_t* info = reinterpret_cast<int64_t*>(a.partial + 2 * (W + 1));
  a.W = W;
  a.lgW = lgW;
  a.no_cond = no_cond ? 1 : 0;
  a.first_worker = no_cond ? 0 : 1;
  a.one_term = one_term ? 1 : 0;
  a.par_min = 6;  // (a parallel solve costs about as much as five or six samples of the serial loop)
  if (const char* env = getenv("NFM_SEQ_WIN_PAR")) a.par_min = atoi(env);  // 0: the sample-by-sample chain only
  a.np = np;
  // without a conductor: how far back a dependency may lie and still take the recipe path, and how far apart the workers may run
  // without a conductor: how far back a dependency may lie and still take the recipe path (W), and how far apart the workers
  // may run (measured, headline shape at 256 workers / cfg2's at 128: recipes for dependencies up to W back 7.3e6 / 5.5e6
  // samples/s; up to 2W or 4W back 7.1e6 / 5.2e6: the longer reach only adds recipe stores)
  const int thr = np - 1, near_r = W;
  a.thr = thr;
  a.near_r = near_r;
  a.dead_slot = -1;
#ifdef NFM_TEST_HOOKS  // (libnimfm_hip_testhooks.so only: the product library has no test hooks)
  if (const char* env = getenv("NFM_SEQ_WIN_TEST_DEAD_SLOT")) a.dead_slot = atoi(env);  // see WinArgs
#endif
  a.m_cap = m_cap;
  a.FW = FW;
  a.lgKp = lgKp;
  const int grp = (kWave >> lgKp) * 4;  // rows the general worker handles together (R * U): its arrays are padded to whole groups
  const size_t mcp = (size_t)(m_cap + grp - 1) / grp * grp;
  const size_t rows = mcp * M.Kp;
  size_t lds_worker = sizeof(double) * ((ada ? 4 : 2) * rows + kWave + (ada ? 4 : 2) * mcp) + sizeof(int) * 3 * mcp + sizeof(unsigned) * W;
  if (k64) lds_worker = sizeof(double) * ((ada ? 3 : 1) * (size_t)kWave * kWave + kWave) + sizeof(unsigned) * W;
  if (ffm) lds_worker = win_ffm_lds(M, m_cap, ada, W);
  if (fmx) lds_worker = win_fmx_lds(M, m_cap, ada, W);
  const size_t lds_cond = one_term ? sizeof(double) * (2 + 9 * (size_t)kSumRing) + sizeof(unsigned) * kSumRing : sizeof(double) * 2 + sizeof(ull) * (size_t)kWinRing * FW;
  size_t lds_bytes = lds_worker > lds_cond ? lds_worker : lds_cond;
  if (lds_bytes > 160 * 1024) return NFM_WIN_FALLBACK;  // (NFM_SEQ_WIN_W beyond what seq_window_supported assumed: the one-workgroup kernel)
  if (lds_bytes < 81 * 1024) lds_bytes = 81 * 1024;  // one workgroup per CU
  NFM_HIP_CHECK(hipMemsetAsync(out2_dev, 0, sizeof(double) * 2, st));
  int64_t pos = 0;
  while (pos < ns) {
    int64_t last = ns - 1;
    int64_t host_info[2] = {ns - 1, 0};
    if (!ada) {
      TimedLaunch tls(ctx, "seq_window_scales");
      hipLaunchKernelGGL(k_win_scales, dim3(1), dim3(3 * kWave), 0, st, M, O, it0 + pos, pos, ns, sw->scales.as<double>(), info);
      NFM_HIP_CHECK(hipMemcpyAsync(host_info, info, sizeof(host_info), hipMemcpyDeviceToHost, st));
      NFM_HIP_CHECK(hipStreamSynchronize(st));
      last = host_info[0];
    }
    // mailboxes empty, counters and abort word zero
    {
      NFM_HIP_CHECK(hipMemsetAsync(sw->ctl.p, 0, sizeof(unsigned) * (W + 64) + sizeof(double) * 2 * (W + 1), st));
      NFM_HIP_CHECK(hipMemsetAsync(sw->fw.p, 0, sizeof(ull) * kFwSlot * np * (size_t)W, st));  // tag 0: nobody's
      const int64_t nm = (int64_t)n_fwd;  // the workers' mailboxes "empty"; the conductor's answers carry tags (0: nobody's)
      if (one_term) NFM_HIP_CHECK(hipMemsetAsync(sw->mail.p, 0, sizeof(ull) * n_fwd, st));  // (one-term mailboxes are tagged as well)
      else hipLaunchKernelGGL(k_win_fill, dim3((unsigned)((nm + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, sw->mail.as<ull>(), nm, kWinSentinel);
      NFM_HIP_CHECK(hipMemsetAsync(sw->mail.as<ull>() + n_fwd, 0, sizeof(ull) * n_res, st));
    }
    a.seg0 = pos;
    a.n_seg = last - pos + 1;
    a.it0 = it0 + pos;
    if (ada) NFM_TRY(launch_window<OPT_ADAGRAD>(ctx, a, lds_bytes, ch, wk));
    else NFM_TRY(launch_window<OPT_SGD>(ctx, a, lds_bytes, ch, wk));
    hipLaunchKernelGGL(k_win_finish, dim3(1), dim3(kWave), 0, st, a.partial, W, out2_dev);
    if (!ada && host_info[1]) {
      const int64_t nP = (int64_t)M.nb * M.da * M.Kp;
      hipLaunchKernelGGL(k_win_rescale, dim3(1024), dim3(kBlock), 0, st, M.P, nP, M.sc + SC_SCALE_P, info, 1);
      hipLaunchKernelGGL(k_win_rescale, dim3(256), dim3(kBlock), 0, st, M.w, M.d, M.sc + SC_SCALE_W, info, 2);
      hipLaunchKernelGGL(k_win_rescale_done, dim3(1), dim3(1), 0, st, M.sc, info);
    }
    unsigned aborted = 0;
    NFM_HIP_CHECK(hipMemcpyAsync(&aborted, a.ctrl, sizeof(aborted), hipMemcpyDeviceToHost, st));
    NFM_HIP_CHECK(hipStreamSynchronize(st));
    if (aborted != 0) {
      // a wait ran into its wall-clock limit (a workgroup was not resident -- other tenants on the CUs -- or a hand-off was
      // lost): the samples of this call are partly applied.  The caller restores its snapshot and runs the one-workgroup kernel
      fprintf(stderr, "[nimfm_hip] the dependency-window kernel gave up waiting at positions [%lld, %lld]; falling back to the one-workgroup kernel\n",
              (long long)(begin + pos), (long long)(begin + last));
      return NFM_WIN_FALLBACK;
    }
    if (a.trace) {  // debugging: the stamps go to a file (100 MHz ticks: taken up, deps, posted, dL, written | fetched, chain, answered)
      std::vector<long long> h((size_t)8 * a.n_seg);
      NFM_HIP_CHECK(hipMemcpy(h.data(), a.trace, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      const char* path = getenv("NFM_SEQ_WIN_TRACE_FILE") ? getenv("NFM_SEQ_WIN_TRACE_FILE") : "/tmp/seqwin_trace.bin";
      if (FILE* f = fopen(path, "wb")) {
        fwrite(h.data(), sizeof(long long), h.size(), f);
        fclose(f);
      }
    }
    pos = last + 1;
  }
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

}  // namespace nfm
