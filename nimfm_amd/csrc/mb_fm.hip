// nimfm_amd/csrc/mb_fm.hip -- NFM_MODE_MINIBATCH for FactorizationMachine: the throughput path.
//
// Replaces the reference's Hogwild drivers (optimizer/sgd_multi.nim:21-37,83-101,
// adagrad_multi.nim:15-36,78-96) -- T threads racing on shared P/w/intercept/it -- with a
// deterministic rule: all samples of a batch see the batch-start
// parameters; their per-sample updates (the reference's expressions: sgd.nim:205-243,
// fit_linear.nim:41-47; adagrad.nim:87-134) are combined per coordinate in sample order
// (SGD: averaged over the samples touching the coordinate; AdaGrad: summed into the state).
//
// Two kernels per batch, no atomics, every sum in a fixed order:
//   row phase     L*SPLIT lanes per SAMPLE: gathers the row's parameter rows (16 B per lane,
//                 coalesced segments of Kp*8 bytes), forms A = sum x p (and sum (x p)^2) per factor,
//                 yhat, loss, dL; writes the per-factor sums A[s] (Kp doubles) and a 32-byte record
//                 {dL, eta_P, eta_w} per sample.
//   column phase  L lanes per UNIQUE FEATURE of the batch (plan.hip): reads the parameter row once,
//                 walks the feature's touches (sample, x) in sample order, recomputes
//                 dA = x (A[s] - p x) from the sample's A row (L2-resident), accumulates, writes the
//                 row once.  Rows touched c times in a batch are read and written once, not c times.
//                 One extra workgroup closes the batch: fixed-order reduction of the row phase's
//                 per-block partial sums (loss, intercept gradient), intercept update, and the
//                 previous batch's per-block viol partials.
// L2 decay is carried by the global scales (common.h): the schedule kernel forms the per-batch
// products of (1 - eta_t * reg) and the prefix kernel the scale at every batch boundary.
#include "mb_fm_kernels.h"

namespace nfm {

// run_batches<L, OPT, GEN> for one L: defined in mb_fm_inst.hip, compiled once per L (-DNFM_INST_L=...)
#define NFM_DECL(LL) \
  int mb_fm_run_L##LL(nfm_ctx* ctx, int opt_kind, bool gen, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W, int TA);
NFM_DECL(1) NFM_DECL(2) NFM_DECL(4) NFM_DECL(8) NFM_DECL(16) NFM_DECL(32) NFM_DECL(64)
#undef NFM_DECL

// ------------------------------------------------------------------------------------------------
// schedule: per-batch decay products and scales (SGD only)
// ------------------------------------------------------------------------------------------------

__global__ void k_schedule(OptView O, int fit_linear, int fit_intercept, const int64_t* __restrict__ bat_pos,
                           const double* __restrict__ it0p, double* __restrict__ Dtab /*[nb][4]*/,
                           double* __restrict__ Ftab /*[nb][2][kFtab]*/) {
  __shared__ double red[3][kBlock];
  const double it0 = *it0p;
  const int b = blockIdx.x;
  const int64_t p0 = bat_pos[b], p1 = bat_pos[b + 1];
  double dP = 1.0, dw = 1.0, d0 = 1.0;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += kBlock) {
    const double it = it0 + (double)p;
    dP *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.beta, it) * O.beta;
    if (fit_linear) dw *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.alpha, it) * O.alpha;
    if (fit_intercept) d0 *= 1 - dev::get_eta(O.sched, O.eta0, O.power, O.alpha0, it) * O.alpha0;
  }
  red[0][threadIdx.x] = dP;
  red[1][threadIdx.x] = dw;
  red[2][threadIdx.x] = d0;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] *= red[0][threadIdx.x + s];
      red[1][threadIdx.x] *= red[1][threadIdx.x + s];
      red[2][threadIdx.x] *= red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Dtab[4 * b + 0] = red[0][0];
    Dtab[4 * b + 1] = red[1][0];
    Dtab[4 * b + 2] = red[2][0];
    // the intercept is touched by every sample: c = len (as a divisor: len / touch_cap beyond the cap)
    const double lc = dev::touch_div((double)(p1 - p0), O.touch_cap);
    Dtab[4 * b + 3] = lc == 1.0 ? red[2][0] : pow(red[2][0], 1.0 / lc);
  }
  // a coordinate touched c times receives D^(1/c) instead of D; relative to the global scale
  // (which advances by D) that is the factor D^(1/c) / D, tabulated for c = 1..kFtab
  if (threadIdx.x < 2 * kFtab) {
    const int which = threadIdx.x / kFtab, c = threadIdx.x % kFtab + 1;
    const double D = red[which][0];
    const double cd = dev::touch_div((double)c, O.touch_cap);
    Ftab[((size_t)b * 2 + which) * kFtab + (c - 1)] = cd == 1.0 ? 1.0 : pow(D, 1.0 / cd) / D;
  }
}

__global__ void k_scale_prefix(double* __restrict__ sc, const double* __restrict__ Dtab, double* __restrict__ Stab, int64_t nb) {
  double sP = sc[SC_SCALE_P], sw = sc[SC_SCALE_W];
  for (int64_t b = 0; b < nb; ++b) {
    Stab[2 * b] = sP;
    Stab[2 * b + 1] = sw;
    sP *= Dtab[4 * b];
    sw *= Dtab[4 * b + 1];
  }
  Stab[2 * nb] = sP;
  Stab[2 * nb + 1] = sw;
  sc[SC_SCALE_P] = sP;
  sc[SC_SCALE_W] = sw;
}

// adds the last batch's per-block viol partials (every other batch's are folded in by the next
// batch's closing workgroup)
__global__ __launch_bounds__(kBlock) void k_epoch_close(const double* __restrict__ parts, int n, double* __restrict__ out_acc) {
  __shared__ double red[kBlock];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) s += parts[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out_acc[1] += red[0];
}

// everything one epoch call enqueues; all kernel arguments are independent of the optimizer's
// `it` (read from the device scalar W.itbuf), so the sequence can be captured once per plan and
// replayed as a hipGraph.
static int enqueue_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P,
                         MbWork& W, int TA) {
  hipStream_t st = ctx->stream;
  NFM_HIP_CHECK(hipMemsetAsync(W.out_acc.p, 0, sizeof(double) * 2, st));
  if (P.n_batches > 0 && opt_kind == OPT_SGD) {
    TimedLaunch tl(ctx, "schedule");
    hipLaunchKernelGGL(k_schedule, dim3((unsigned)P.n_batches), dim3(kBlock), 0, st, O, M.fit_linear, M.fit_intercept,
                       P.bat_pos_dev.as<int64_t>(), W.itbuf.as<double>(), W.Dtab.as<double>(), W.Ftab.as<double>());
    hipLaunchKernelGGL(k_scale_prefix, dim3(1), dim3(1), 0, st, M.sc, W.Dtab.as<double>(), W.Stab.as<double>(), P.n_batches);
    NFM_HIP_CHECK(hipGetLastError());
  }
  const bool gen = !(M.nb == 1 && M.degree == 2);  // anything but a single order of degree 2
  NFM_CHECK(gen || !P.use_singles || (P.toff.p && P.single.p), NFM_ERR_INVALID, "plan lacks the singles tables");
  NFM_CHECK(!gen || !P.use_singles, NFM_ERR_INVALID, "a plan with singles needs the degree-2 kernels");
#define NFM_RUN(LL) \
  case LL:          \
    return mb_fm_run_L##LL(ctx, opt_kind, gen, X, M, O, P, W, TA);
  switch (M.L) {
    NFM_RUN(1)
    NFM_RUN(2)
    NFM_RUN(4)
    NFM_RUN(8)
    NFM_RUN(16)
    NFM_RUN(32)
    NFM_RUN(64)
  }
#undef NFM_RUN
  return set_error(NFM_ERR_UNSUPPORTED, "unsupported lanes-per-row %d", M.L);
}

__global__ void k_set_double(double* p, double v) { *p = v; }

void MbWork::drop_graph() {
  if (graph_exec) (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec));
  graph_exec = nullptr;
  for (void* e : seg_execs)
    if (e) (void)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(e));
  seg_execs.clear();
  seg_cut.clear();
  seg_key = -1;
}
int MbWork::seg_cut_here(nfm_ctx* ctx, int64_t b) {
  hipGraph_t graph = nullptr;
  NFM_HIP_CHECK(hipStreamEndCapture(ctx->stream, &graph));
  hipGraphExec_t exec = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  NFM_HIP_CHECK(e2);
  seg_execs.push_back(exec);
  seg_cut.push_back(b);
  NFM_HIP_CHECK(hipGraphLaunch(exec, ctx->stream));
  return NFM_OK;
}
int MbWork::seg_resume(nfm_ctx* ctx) {
  NFM_HIP_CHECK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  return NFM_OK;
}
MbWork::~MbWork() { drop_graph(); }

int mb_fm_epoch(nfm_ctx* ctx, int opt_kind, const CsrView& X, const ModelView& M, const OptView& O, const Plan& P, MbWork& W,
                int64_t it0, double* out2_host, uint64_t data_serial, bool defer_sync) {
  NFM_CHECK(M.kind == NFM_KIND_FM, NFM_ERR_UNSUPPORTED, "mb_fm_epoch: FM only");
  NFM_CHECK(M.degree <= dev::kMaxDeg, NFM_ERR_UNSUPPORTED, "mini-batch mode supports degree <= %d", dev::kMaxDeg);
  NFM_CHECK(M.Kp == 2 * M.L && M.Kp <= 128, NFM_ERR_UNSUPPORTED, "mini-batch mode: factor blocks of at most 128 (api.hip splits wider FMs)");
  hipStream_t st = ctx->stream;
  int TA = 0;
  for (int o = 0; o < M.nb; ++o) TA += M.deg_of(o) - 1;
  constexpr int kMinGroupsPerBlock = kWavesPerBlock;  // L = 64
  const void* before[] = {W.Abuf.p, W.rec.p, W.partsA.p, W.partsB.p, W.Dtab.p, W.Stab.p, W.Ftab.p, W.out_acc.p, W.itbuf.p, W.hpart.p, W.prox.p};
  NFM_TRY(W.Abuf.ensure(sizeof(double) * (size_t)std::max<int64_t>(P.max_batch, 1) * std::max(TA, 1) * M.Kp));
  NFM_TRY(W.rec.ensure(sizeof(SampleRec) * (size_t)std::max<int64_t>(P.max_batch, 1)));
  // one partial per row-phase workgroup: >= 4 samples per workgroup, 2 in k_row_phase_ada2
  NFM_TRY(W.partsA.ensure(sizeof(PartA) * (size_t)(P.max_batch / 2 + 2)));
  NFM_TRY(W.partsB.ensure(sizeof(double) * 2 * (size_t)(P.max_unique / kMinGroupsPerBlock + P.max_batch / kWavesPerBlock + P.max_heavy / kWavesPerBlock + 6)));
  NFM_TRY(W.hpart.ensure(sizeof(double) * (size_t)std::max<int64_t>(P.max_segs, 1) * std::max(M.nb, 1) * (2 * M.Kp + 4)));
  NFM_TRY(W.Dtab.ensure(sizeof(double) * 4 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Stab.ensure(sizeof(double) * 2 * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.Ftab.ensure(sizeof(double) * 2 * kFtab * (size_t)(P.n_batches + 1)));
  NFM_TRY(W.out_acc.ensure(sizeof(double) * 2));
  NFM_TRY(W.itbuf.ensure(sizeof(double)));
  // MBPSGD scratch: row norms [nb][da] | per-component pass state 3 x [nb][Kp] + counter | per-workgroup partials [nb][1024][2 Kp]
  if (opt_kind == OPT_PSGD)
    NFM_TRY(W.prox.ensure(sizeof(double) * ((size_t)M.nb * M.da + (size_t)std::max(M.nb, 1) * M.Kp * (3 + 2 * 1024) + 8)));
  const void* after[] = {W.Abuf.p, W.rec.p, W.partsA.p, W.partsB.p, W.Dtab.p, W.Stab.p, W.Ftab.p, W.out_acc.p, W.itbuf.p, W.hpart.p, W.prox.p};
  for (size_t q = 0; q < sizeof(before) / sizeof(before[0]); ++q)
    if (before[q] != after[q]) W.drop_graph();
  hipLaunchKernelGGL(k_set_double, dim3(1), dim3(1), 0, st, W.itbuf.as<double>(), (double)it0);
  // A plan that is reused (shuffle off) is replayed as a hipGraph: two dependent launches per
  // batch make the epoch launch-bound on the host otherwise.  Timing mode and one-off plans
  // (explicit permutations) launch directly.
  const bool want_graph = W.use_graph && !W.after_batch && !ctx->timing.enabled && !P.has_perm && P.n_batches >= 8;
  if (want_graph) {
    if (!W.graph_exec || W.graph_plan_serial != P.serial || W.graph_opt != opt_kind || W.graph_data_serial != data_serial) {
      W.drop_graph();
      hipGraph_t graph = nullptr;
      NFM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue_epoch(ctx, opt_kind, X, M, O, P, W, TA);
      const hipError_t e = hipStreamEndCapture(st, &graph);
      if (rc != NFM_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      NFM_HIP_CHECK(e);
      hipGraphExec_t exec = nullptr;
      const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      NFM_HIP_CHECK(e2);
      W.graph_exec = exec;
      W.graph_plan_serial = P.serial;
      W.graph_data_serial = data_serial;
      W.graph_opt = opt_kind;
    }
    NFM_HIP_CHECK(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(W.graph_exec), st));
  } else if (W.use_graph && W.after_batch && W.is_sync && !ctx->timing.enabled && !P.has_perm && P.n_batches >= 8) {
    // a data-parallel epoch over a reusable plan: one graph per stretch of mini-batches between exchange points
    const bool seg_on = !(getenv("NFM_DP_GRAPH") && atoi(getenv("NFM_DP_GRAPH")) == 0);  // (read per call: tests compare both ways)
    if (!seg_on) {
      NFM_TRY(enqueue_epoch(ctx, opt_kind, X, M, O, P, W, TA));
    } else if (!W.seg_execs.empty() && W.seg_plan_serial == P.serial && W.seg_data_serial == data_serial && W.seg_opt == opt_kind &&
               W.seg_key == W.seg_key_now) {
      for (size_t sgi = 0; sgi < W.seg_execs.size(); ++sgi) {
        NFM_HIP_CHECK(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(W.seg_execs[sgi]), st));
        if (sgi + 1 < W.seg_execs.size()) NFM_TRY(W.after_batch(W.seg_cut[sgi]));
      }
    } else {
      W.drop_graph();
      W.seg_recording = true;
      int rc = W.seg_resume(ctx);
      if (rc == NFM_OK) rc = enqueue_epoch(ctx, opt_kind, X, M, O, P, W, TA);  // run_batches cuts at the exchange points
      W.seg_recording = false;
      if (rc != NFM_OK) {
        hipGraph_t junk = nullptr;
        (void)hipStreamEndCapture(st, &junk);  // (whatever state the capture is in)
        if (junk) (void)hipGraphDestroy(junk);
        (void)hipGetLastError();
        W.drop_graph();
        return rc;
      }
      NFM_TRY(W.seg_cut_here(ctx, P.n_batches - 1));
      W.seg_plan_serial = P.serial;
      W.seg_data_serial = data_serial;
      W.seg_opt = opt_kind;
      W.seg_key = W.seg_key_now;
    }
  } else {
    NFM_TRY(enqueue_epoch(ctx, opt_kind, X, M, O, P, W, TA));
  }
  NFM_HIP_CHECK(hipMemcpyAsync(out2_host, W.out_acc.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  if (!defer_sync) NFM_HIP_CHECK(hipStreamSynchronize(st));
  return NFM_OK;
}

// predictAllWithGrad: the records of the (single) batch the last epoch call ran, unpacked
__global__ void k_unpack_rec(const SampleRec* __restrict__ rec, int64_t n, double* __restrict__ yhat, double* __restrict__ dL) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const SampleRec r = rec[i];
  if (yhat) yhat[i] = r.yhat;
  if (dL) dL[i] = r.dL;
}

int mb_fm_records(nfm_ctx* ctx, MbWork& W, int64_t n, double* yhat_dev, double* dL_dev) {
  NFM_CHECK(W.rec.p && W.rec.bytes >= sizeof(SampleRec) * (size_t)n, NFM_ERR_INVALID, "no records of %lld samples", (long long)n);
  if (n > 0)
    hipLaunchKernelGGL(k_unpack_rec, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, W.rec.as<SampleRec>(), n,
                       yhat_dev, dL_dev);
  NFM_HIP_CHECK(hipGetLastError());
  return NFM_OK;
}

}  // namespace nfm
