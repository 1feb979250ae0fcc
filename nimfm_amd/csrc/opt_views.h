// nimfm_amd/csrc/opt_views.h -- optimizer parameter/state views passed by value to kernels.
#pragma once
#include "common.h"

namespace nfm {

// OPT_PSGD: the reference's mini-batch proximal SGD (optimizer/minibatch_psgd.nim)
enum { OPT_SGD = 0, OPT_ADAGRAD = 1, OPT_PSGD = 2 };

struct OptView {
  // hyper-parameters (newSGD optimizer/sgd.nim:23-52, newAdaGrad optimizer/adagrad.nim:20-44)
  double eta0, alpha0, alpha, beta, power, eps, loss_param;
  // SGD mini-batch rule: at most this many of a batch's per-sample steps on one coordinate are SUMMED (the reference's
  // Hogwild threads apply theirs at full strength too, optimizer/sgd_multi.nim:83-101); a coordinate touched c > cap
  // times receives cap / c of the sum.  1 (default): the per-coordinate mean
  double touch_cap;
  // AdaGrad mini-batch rule (round 5): g_norm of a coordinate grows by the batch's  sum g^2 + ada_cross ((sum g)^2 - sum g^2)  -- the
  // cross products of the samples' gradients, all taken from the batch-start parameters: the norm sees how far the batch AGREES on a
  // coordinate (what NFM_DP_STATE_CROSS does for the ranks of a data-parallel group, dp.h).  0 (default): the samples' squares alone
  double ada_cross;
  int32_t loss, sched, track_viol, pad_;
  // AdaGrad state, device layout: G/N [nb][da][Kp] (padding: G = 0, N = eps), Gw/Nw [d],
  // gsc[0] = g_sum.intercept, gsc[1] = g_norm.intercept
  double* G;
  double* N;
  double* Gw;
  double* Nw;
  double* gsc;
  // MBPSGD (newMBPSGD, optimizer/minibatch_psgd.nim:24-65): gamma, miniBatchSize as a double, NFM_REG_*
  double gamma, bsize;
  int32_t reg, reg_transpose;
  // pgd.predictAllWithGrad (optimizer/pgd.nim:70-103): when set, the column phase of OPT_PSGD stores the batch
  // gradient here (device layout of P, [d], one scalar) instead of stepping the parameters
  double* gradP;
  double* gradw;
  double* gradb;
};

// ---- seq.hip ----
int launch_sequential(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O,
                      const int64_t* perm_dev, int64_t begin, int64_t end, int64_t it0, int m_cap,
                      double* out2_dev /*{loss_sum, viol_sum}*/);
// ---- seqwin.hip: the same order as launch_sequential, run as a dependency window over the chip ----
struct SeqWin {           // kept on the optimizer between calls
  DevBuf prev;            // [nnz] per stored entry: previous position of the call with the same feature
  DevBuf prevq, next;     // [nnz] that feature's entry index in the previous sample's row; next position with the feature
  DevBuf scales, mail, ctl, fw, trace;
  DevBuf snap;            // the model's arena (+ AdaGrad: the state arena) as it was when the call began (abort: put back)
  int64_t fallbacks = 0;  // calls that ended in the one-workgroup kernel after an abort (forgotten one at a time: 16 clean calls each)
  int64_t clean_calls = 0;
  bool valid = false, had_perm = false;
  uint64_t ds_uid = 0;
  int64_t begin = 0, end = 0, nnz = 0;
};
// the model as the window kernels see it (seqwin.hip: a 65 ... 128-factor FM read as two blocks of 64)
ModelView seq_window_view(const ModelView& M);
// the model as the one-sample-in-flight kernel sees it: a wide FM of one order whose kc blocks lie feature-major is ONE row of
// kc * Kp factors per feature (padding zeros inside the row add nothing to the ascending factor sum, sgd.nim:172-173)
inline ModelView seq_row_view(const ModelView& M) {
  if (!(M.kind == NFM_KIND_FM && M.kc > 1 && M.nb == M.kc && M.bs == 1 && M.rs == M.nb)) return M;
  ModelView V = M;
  V.Kp = M.nb * M.Kp;
  V.k = V.Kp;
  V.L = 64;
  V.nb = 1;
  V.kc = 1;
  V.bs = M.da;
  V.rs = 1;
  return V;
}
bool seq_window_supported(const ModelView& M, int m_cap, int64_t ns, int64_t nnz, int n_cu, bool ada);
// launch_sequential_window's third outcome besides NFM_OK and an error: the window could not run to its end (the kernel does
// not fit a CU, or a wait inside it timed out and the launch aborted).  Parameters and state of the call may be partly
// updated: the caller puts its snapshot back and runs launch_sequential over the same range.
constexpr int NFM_WIN_FALLBACK = 1;
int launch_sequential_window(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O,
                             const int64_t* perm_dev, int64_t begin, int64_t end, int64_t it0, int m_cap,
                             double* out2_dev /*{loss_sum, viol_sum}*/, SeqWin* sw, uint64_t ds_uid, bool perm_is_callers);
// AdaGrad finalize (optimizer/adagrad.nim:65-84): all parameters from the state with it' = it-1
int launch_adagrad_finalize(nfm_ctx* ctx, const ModelView& M, const OptView& O, int64_t it);

}  // namespace nfm
