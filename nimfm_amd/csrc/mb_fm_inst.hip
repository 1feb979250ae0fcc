// nimfm_amd/csrc/mb_fm_inst.hip -- the mini-batch kernels of mb_fm_kernels.h for ONE lanes-per-row value
// (Makefile: -DNFM_INST_L=1 ... 64 -> mb_fm_L1.o ... mb_fm_L64.o).
#include "mb_fm_kernels.h"

#ifndef NFM_INST_L
#error "compile with -DNFM_INST_L=<lanes per row>"
#endif
#define NFM_CAT2(a, b) a##b
#define NFM_CAT(a, b) NFM_CAT2(a, b)

namespace nfm {

int NFM_CAT(mb_fm_run_L, NFM_INST_L)(nfm_ctx* ctx, int opt_kind, bool gen, const CsrView& X, const ModelView& M, const OptView& O,
                                     const Plan& P, MbWork& W, int TA) {
  constexpr int LL = NFM_INST_L;
  if (opt_kind == OPT_PSGD)
    return gen ? run_batches<LL, OPT_PSGD, true>(ctx, X, M, O, P, W, TA) : run_batches<LL, OPT_PSGD, false>(ctx, X, M, O, P, W, TA);
  if (gen)
    return opt_kind == OPT_SGD ? run_batches<LL, OPT_SGD, true>(ctx, X, M, O, P, W, TA)
                               : run_batches<LL, OPT_ADAGRAD, true>(ctx, X, M, O, P, W, TA);
  return opt_kind == OPT_SGD ? run_batches<LL, OPT_SGD, false>(ctx, X, M, O, P, W, TA)
                             : run_batches<LL, OPT_ADAGRAD, false>(ctx, X, M, O, P, W, TA);
}

}  // namespace nfm
