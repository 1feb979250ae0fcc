// nimfm_amd/csrc/mb_ffm.hip -- NFM_MODE_MINIBATCH for FieldAwareFactorizationMachine (placeholder
// until the FFM row/column kernels land; the sequential mode covers FFM meanwhile).
#include "mb.h"

namespace nfm {

int mb_ffm_epoch(nfm_ctx*, int, const CsrView&, const ModelView&, const OptView&, const Plan&, MbWork&, int64_t, double*) {
  return set_error(NFM_ERR_UNSUPPORTED, "mini-batch mode for FFM is not implemented yet; use NFM_MODE_SEQUENTIAL");
}

}  // namespace nfm
