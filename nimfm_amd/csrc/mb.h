// nimfm_amd/csrc/mb.h -- mini-batch mode: work buffers and entry points.
#pragma once
#include <functional>
#include <vector>

#include "opt_views.h"
#include "plan.h"

namespace nfm {

// Scratch owned by an optimizer and reused by every batch/epoch.
struct MbWork {
  DevBuf Abuf;     // per-sample per-factor sums of the current batch  [B][TA][Kp]
  DevBuf rec;      // per-sample record {dL, eta_P, eta_w, -}          [B]
  DevBuf partsA;   // per-block partial sums of the row phase
  DevBuf partsB;   // per-block partial viol of the column phase
  DevBuf Dtab;     // per-batch decay products {D_P, D_w, D_0, -}
  DevBuf Stab;     // scales at every batch boundary {scale_P, scale_w}
  DevBuf Ftab;     // per-batch decay corrections by touch count [2][64]
  DevBuf out_acc;  // {loss_sum, viol_sum} of the epoch call
  DevBuf contrib;  // FFM: per-touch gradient rows
  DevBuf itbuf;    // device scalar: `it` at the start of the epoch call
  DevBuf hpart;    // heavy features: per-segment partial sums
  DevBuf prox;     // MBPSGD: row norms [nb][da] + thresholds [nb][Kp] of the coupled prox operators
  // hipGraph of one epoch call over a reusable plan
  bool use_graph = true;
  void* graph_exec = nullptr;
  uint64_t graph_plan_serial = 0;
  uint64_t graph_data_serial = 0;  // nfm_dataset::uid ^ serial mix the graph's captured dataset pointers belong to
  int graph_opt = -1;
  // data-parallel hook (dp.h): called after every mini-batch has been enqueued; set only for the duration of one
  // nfm_opt_epoch call of an optimizer with a group attached (such an epoch is never replayed as a graph)
  std::function<int(int64_t)> after_batch;
  // Data-parallel epochs as hipGraphs (FM): the mini-batches BETWEEN two exchange points are captured as one graph each;
  // the exchange itself (increment passes, the collective on the group's stream, events) stays outside.  is_sync(b): the
  // hook has work after mini-batch b.  seg_execs[s] covers the launches up to and including mini-batch seg_cut[s] (the last
  // one: to the end of the call); replayed while the key (plan, data, optimizer, period, exchange points) stays the same.
  std::function<bool(int64_t)> is_sync;
  std::vector<void*> seg_execs;
  std::vector<int64_t> seg_cut;
  uint64_t seg_plan_serial = 0, seg_data_serial = 0;
  int seg_opt = -1;
  int64_t seg_key = -1;  // sync_period * 2^32 + exchange points of the call
  int64_t seg_key_now = -1;
  bool seg_recording = false;
  int seg_cut_here(nfm_ctx* ctx, int64_t b);  // ends the capture of the current segment, instantiates and launches it
  int seg_resume(nfm_ctx* ctx);               // begins the capture of the next one
  void drop_graph();
  ~MbWork();
};

// data_serial: changes whenever one of X's device pointers does (a captured graph holds them)
// defer_sync: return right after the epoch has been enqueued (out2_host must then be pinned memory; the caller
// synchronises ctx->stream before reading it)
int mb_fm_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                int64_t it0, double* out2_host, uint64_t data_serial = 0, bool defer_sync = false);
// psgd.hip: Params.step's shrink + the regulariser's prox after one mini-batch of MBPSGD (it = it0p[0] + it_b)
void launch_psgd_step(nfm_ctx* ctx, const ModelView& M, const OptView& O, MbWork& W, const double* it0p, double it_b);
// yhat / dloss of the samples of the last epoch call's batch (a single batch: pgd.predictAllWithGrad), device arrays
int mb_fm_records(nfm_ctx* ctx, MbWork& W, int64_t n, double* yhat_dev, double* dL_dev);
int mb_ffm_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                 int64_t it0, double* out2_host, bool defer_sync = false);

}  // namespace nfm
