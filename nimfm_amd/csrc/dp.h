// nimfm_amd/csrc/dp.h -- data-parallel groups inside the library (include/nimfm_hip.h, nfm_dp_*).
//
// The reference's only parallel strategy is shared-memory Hogwild over contiguous sample slices
// (optimizer/sgd_multi.nim:83-101: every thread owns a slice, all threads share ONE model, nothing is
// synchronised).  Across GPUs the slices become per-rank shards resident in each GPU's HBM and the shared model
// becomes replicas that are reconciled every `sync_period` mini-batches:
//   Both optimizers exchange INCREMENTS since the last agreed state:
//   SGD      the ranks' increments are averaged (default: local SGD, as stable as one rank) or summed (NFM_DP_SUM: every
//            rank's steps land in the model, as every Hogwild thread's do in the reference's shared one) -- dp.hip;
//   AdaGrad  the state is additive over samples (optimizer/adagrad.nim:113-134): the sum is the state ONE process would
//            hold after all shards' samples.
// A mid-epoch exchange is DELAYED by one period so that the collective (on its own stream) runs beside the next
// period's mini-batches: at sync point k a rank snapshots what it contributes and starts the all-reduce; the result
// is folded in at sync point k+1.  The fold-in point is fixed, so the result is deterministic.  The exchange that ends
// an nfm_opt_epoch call is exact and blocking: all replicas leave the call bitwise identical.
#pragma once
#include "common.h"

namespace nfm {

enum { DP_SUM = 0, DP_MAX = 1 };

// How the ranks of a group talk.  allreduce: out-of-place reduction of n doubles, enqueued on stream st of the calling
// rank's device; every rank receives the same bits.
struct DpTransport {
  int rank = 0, world = 1;
  // the ranks are host threads of ONE process (nfm_dp_create_local).  Stream capture is not used then: another rank's
  // thread may synchronise the device (the block cache's cross-thread reuse) while this one captures, which voids the capture
  bool in_process = false;
  virtual ~DpTransport() {}
  virtual int allreduce(const double* send, double* recv, int64_t n, int op, hipStream_t st) = 0;
};

}  // namespace nfm

struct nfm_dp {
  nfm_ctx* ctx = nullptr;
  nfm::DpTransport* t = nullptr;
  hipStream_t comm = nullptr;               // the exchange runs here, beside the mini-batches on ctx->stream
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  // per attached optimizer (one optimizer per group at a time)
  nfm::DevBuf snap, recv, base, scal;       // what this rank contributed / the reduced result / AdaGrad: agreed state
  bool pending = false;                     // a collective is in flight (its result is folded in at the next sync)
  int64_t pending_n = 0;
  // statistics (bench.py): collectives issued and bytes moved per rank
  int64_t n_collectives = 0, bytes = 0;
  uint64_t uid = 0;                         // process-unique: an optimizer refers to its group by (pointer, uid)
  ~nfm_dp();                                // releases the stream, the events and the transport (dp.hip)
};

namespace nfm {

// One nfm_opt_epoch call of an optimizer with a group attached.
struct DpEpoch {
  nfm_dp* dp = nullptr;
  int opt_kind = 0;            // OPT_SGD / OPT_ADAGRAD
  double* arena = nullptr;     // SGD: [P | w | scalars]; AdaGrad: [G | N | Gw | Nw | gscalars]
  int64_t n = 0;               // doubles in the arena
  int64_t skip_lo = 0, skip_hi = 0;  // SGD: the two scale slots (identical on all ranks at a sync point; never exchanged)
  int64_t seg_w = 0, seg_sc = 0;     // SGD: the arena is [P: 0 .. seg_w | w: seg_w .. seg_sc | scalars]
  int64_t sync_period = 0;     // mini-batches between exchanges, 0 = only the closing exchange
  int64_t n_sync = 0;          // mid-epoch sync points every rank reaches (agreed at dp_epoch_begin)
  bool overlap = true;
  double combine_w = 1.0;      // SGD: 1 / world (mean of the ranks' increments, the default) or 1 (their sum); AdaGrad: 1
  // AdaGrad, NFM_DP_STATE_CROSS: the ranks' g_sum increments are SUMMED, and the squared norm takes what the sum of the ranks'
  // g_norm increments cannot see -- how far the ranks AGREE: g_norm += sum_r dN_r + gamma ((sum_r dG_r)^2 - sum_r dG_r^2), never
  // less than before.  Increments formed from the same stale point push the same way; their cross products inflate the norm
  // exactly where adding them up would over-shoot (what the square of a mini-batch's gradient SUM does for large-batch AdaGrad),
  // and vanish where the ranks saw different things.  One all-reduce as before: a rank sends dN_r - gamma dG_r^2 in place of dN_r.
  // pair[q] = {offset of a g_sum span, offset of its g_norm span, length} in the arena (P, w, intercept); n_pairs = 0: off
  double cross_gamma = 0.0;
  int n_pairs = 0;
  int64_t pair[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
};

// a group handle that nfm_dp_destroy has not seen yet (an optimizer may outlive the group it was attached to)
bool dp_is_live(const nfm_dp* dp, uint64_t uid);

// agree on the number of mid-epoch sync points: min over ranks of (full mini-batches / sync_period), strictly before
// the rank's last batch; AdaGrad: base <- state
int dp_epoch_begin(DpEpoch& e, int64_t n_full_batches, int64_t n_batches);
// called after mini-batch b (0-based) has been enqueued on ctx->stream
int dp_after_batch(DpEpoch& e, int64_t b);
// folds a pending (delayed) exchange in (SGD: stored units)
int dp_fold_pending(DpEpoch& e);
// closing exchange (exact, blocking).  SGD: in true values under one agreed scale; leaves the arena in true values with
// both scales 1 (no launch_rescale afterwards).
// sums[3] = {loss_sum, viol_sum, samples} of this rank in, of all ranks out (device buffer of 3 doubles).
int dp_epoch_end(DpEpoch& e, double* sums_dev);

}  // namespace nfm
