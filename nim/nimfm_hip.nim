## nimfm_hip.nim -- Nim side of the MI355X hot path: nimfm's FM surface over libnimfm_hip.so
## (include/nimfm_hip.h).  This module holds everything that needs no private field of nimfm's
## objects: the FFI declarations, device-resident datasets, model push/pull, `decisionFunction`
## (and with it nimfm's generic `predict` / `predictProba` / `score`, model/fm_base.nim:18-48, which
## call `self.decisionFunction(X)` for whatever dataset type they are instantiated with), and the epoch
## driver the `fit` overloads share.
##
## The `fit` overloads themselves read private hyper-parameters (SGD.eta0 / scheduling / power,
## optimizer/sgd.nim:15-17; AdaGrad.eps, optimizer/adagrad.nim:13), so they live in four small files that
## nimfm's optimizer modules `include` (INTEGRATION.md):
##     optimizer/sgd.nim          when defined(nimfmHip): include hip_sgd        # fit(SGD, HipCSRDataset, ..)
##     optimizer/adagrad.nim      when defined(nimfmHip): include hip_adagrad
##     optimizer/sgd_ffm.nim      when defined(nimfmHip): include hip_sgd_ffm    # .. HipCSRFieldDataset, FFM
##     optimizer/adagrad_ffm.nim  when defined(nimfmHip): include hip_adagrad_ffm
## After that
##     var X: HipCSRDataset; var y: seq[float64]
##     loadSVMLightFile("train.svm", X, y)          # text parsed on the GPU, CSR only ever in HBM
##     newSGD(loss = newLogistic()).fit(X, y, fm)   # the reference's own call, resolved by overloading
##     echo fm.score(X, y)
## and `toHip(X)` turns an existing CSRDataset / CSRFieldDataset into its device-resident twin.
##
## The epoch loop, shuffle (Nim's global RNG, optimizer/sgd.nim:297), stopping criterion, verbose lines and
## callbacks stay in Nim where the reference has them; only per-sample work crosses the FFI.
##
## NOTE: written against the C header and the reference's sources; NOT compiled in the build image (no Nim
## toolchain there, SURVEY.md section 0).  tests/test_nim_shim.py checks every `proc nfm_*` below against
## include/nimfm_hip.h (name, arity, parameter widths, struct layouts).  The executable stand-in with the same
## control flow is nimfm_amd/host.py.
import os, strutils
import nimfm/[dataset, loss, metrics, utils]
import nimfm/tensor/[tensor, sparse]
import nimfm/model/[fm_base, factorization_machine, field_aware_factorization_machine, params]
import nimfm/optimizer/utils as optutils
import std/[math, random, sequtils, strformat, sugar]

const libnfm = "libnimfm_hip.so"

type
  NfmCtx* = pointer
  NfmDataset* = pointer
  NfmModel* = pointer
  NfmOpt* = pointer
  NfmDp* = pointer
  NfmStream* = pointer
  NfmModelCfg* {.bycopy.} = object
    kind*, task*, degree*, nComponents*, fitLower*, fitIntercept*, fitLinear*, reserved*: int32
    nFeatures*, nFields*: int64
  NfmSgdCfg* {.bycopy.} = object
    eta0*, alpha0*, alpha*, beta*, power*, lossParam*: float64
    loss*, scheduling*, mode*, reserved*: int32
    batch*: int64
  NfmAdaGradCfg* {.bycopy.} = object
    eta0*, alpha0*, alpha*, beta*, eps*, lossParam*: float64
    loss*, mode*, trackViol*, reserved*: int32
    batch*: int64
  NfmMbpsgdCfg* {.bycopy.} = object
    eta0*, alpha0*, alpha*, beta*, gamma*, power*, lossParam*: float64
    loss*, scheduling*, reg*, regTranspose*: int32
    batch*: int64

const
  nfmModeSequential* = 0'i32   ## the reference's single-thread order (sgd.nim:294-308)
  nfmModeMinibatch* = 1'i32    ## the library's deterministic data-parallel rule (replaces *_multi.nim)
  nfmKindFM = 0'i32
  nfmKindFFM = 1'i32

{.push importc, cdecl, dynlib: libnfm.}
proc nfm_last_error*(): cstring
proc nfm_version*(): int32
proc nfm_device_count*(n: ptr int32): int32
proc nfm_ctx_create*(deviceId: int32, hipStream: pointer, outp: ptr NfmCtx): int32
proc nfm_ctx_destroy*(ctx: NfmCtx): int32
proc nfm_ctx_synchronize*(ctx: NfmCtx): int32
proc nfm_dataset_create_csr*(ctx: NfmCtx, nSamples, nFeatures: int64, indptr, indices: ptr int64, data: ptr float64,
                             fields: ptr int64, nFields: int64, y: ptr float64, outp: ptr NfmDataset): int32
proc nfm_dataset_load_svmlight*(ctx: NfmCtx, path: cstring, nFeatures: int64, outp: ptr NfmDataset): int32
proc nfm_dataset_load_ffm*(ctx: NfmCtx, path: cstring, nFeatures, nFields: int64, outp: ptr NfmDataset): int32
proc nfm_dataset_load_stream*(ctx: NfmCtx, xPath, yPath: cstring, outp: ptr NfmDataset): int32
proc nfm_stream_open*(ctx: NfmCtx, xPath, yPath: cstring, outp: ptr NfmStream): int32
proc nfm_stream_shape*(s: NfmStream, nSamples, nFeatures, nnz, nFields: ptr int64): int32
proc nfm_stream_load_rows*(s: NfmStream, rowBegin, rowEnd: int64, outp: ptr NfmDataset): int32
proc nfm_stream_prefetch_rows*(s: NfmStream, rowBegin, rowEnd: int64): int32
proc nfm_stream_close*(s: NfmStream): int32
proc nfm_convert_svmlight*(ctx: NfmCtx, fIn, fOutX, fOutY: cstring): int32
proc nfm_dataset_shape*(ds: NfmDataset, nSamples, nFeatures, nnz, nFields: ptr int64): int32
proc nfm_dataset_get_targets*(ds: NfmDataset, y: ptr float64): int32
proc nfm_dataset_get_csr*(ds: NfmDataset, indptr, indices: ptr int64, data: ptr float64, fields: ptr int64): int32
proc nfm_dataset_set_targets*(ds: NfmDataset, y: ptr float64): int32
proc nfm_dataset_destroy*(ds: NfmDataset): int32
proc nfm_model_create*(ctx: NfmCtx, cfg: ptr NfmModelCfg, outp: ptr NfmModel): int32
proc nfm_model_shape*(m: NfmModel, nBlocks, nAug: ptr int32): int32
proc nfm_model_set_params*(m: NfmModel, P, w: ptr float64, intercept: float64, lams: ptr float64): int32
proc nfm_model_get_params*(m: NfmModel, P, w: ptr float64, intercept: ptr float64): int32
proc nfm_decision_function*(m: NfmModel, ds: NfmDataset, outp: ptr float64): int32
proc nfm_score*(m: NfmModel, ds: NfmDataset, outp: ptr float64): int32
proc nfm_metrics*(m: NfmModel, ds: NfmDataset, rmse, accuracy, rocauc: ptr float64): int32
proc nfm_model_sqnorms*(m: NfmModel, pSq, wSq: ptr float64): int32
proc nfm_model_destroy*(m: NfmModel): int32
proc nfm_sgd_create*(m: NfmModel, cfg: ptr NfmSgdCfg, outp: ptr NfmOpt): int32
proc nfm_adagrad_create*(m: NfmModel, cfg: ptr NfmAdaGradCfg, outp: ptr NfmOpt): int32
proc nfm_mbpsgd_create*(m: NfmModel, cfg: ptr NfmMbpsgdCfg, outp: ptr NfmOpt): int32
proc nfm_opt_predict_all_with_grad*(o: NfmOpt, ds: NfmDataset, yPred, dL, gradP, gradW, gradB, lossSum: ptr float64): int32
proc nfm_opt_set_it*(o: NfmOpt, it: int64): int32
proc nfm_opt_get_it*(o: NfmOpt, it: ptr int64): int32
proc nfm_opt_get_state*(o: NfmOpt, gsumP, gnormP, gsumW, gnormW, gsumB, gnormB: ptr float64): int32
proc nfm_opt_set_state*(o: NfmOpt, gsumP, gnormP, gsumW, gnormW: ptr float64, gsumB, gnormB: float64): int32
proc nfm_opt_epoch*(o: NfmOpt, ds: NfmDataset, perm: ptr int64, first, last: int64,
                    lossSum, violSum: ptr float64): int32
proc nfm_opt_set_shuffle*(o: NfmOpt, seed: int64): int32
proc nfm_opt_get_perm*(o: NfmOpt, perm: ptr int64, n: int64): int32
proc nfm_opt_announce_perm*(o: NfmOpt, permNext: ptr int64, first, last: int64): int32
proc nfm_opt_finalize*(o: NfmOpt): int32
proc nfm_opt_destroy*(o: NfmOpt): int32
# data-parallel groups: one process per GPU (RCCL over xGMI) or the ranks of one process (threads + peer access)
proc nfm_dp_unique_id*(id: pointer): int32
proc nfm_dp_create*(ctx: NfmCtx, id: pointer, rank, world: int32, outp: ptr NfmDp): int32
proc nfm_dp_create_local*(ctxs: ptr NfmCtx, world: int32, outp: ptr NfmDp): int32
proc nfm_dp_info*(dp: NfmDp, rank, world: ptr int32, nCollectives, bytes: ptr int64): int32
proc nfm_dp_destroy*(dp: NfmDp): int32
proc nfm_opt_set_dp*(o: NfmOpt, dp: NfmDp, syncPeriod: int64, overlap: int32): int32
proc nfm_opt_set_dp_combine*(o: NfmOpt, combine: int32): int32  # -1 (default): auto (SGD the mean; AdaGrad summed at syncPeriod 1, else the cross rule), 0: mean, 1: sum, 2: averaged state, 3: 1/sqrt(world), 4: cross (g_sum summed, g_norm + the ranks' agreement)
proc nfm_opt_set_touch_cap*(o: NfmOpt, cap: float64): int32     # SGD mini-batch rule: steps per coordinate summed before averaging sets in
proc nfm_opt_set_ada_cross*(o: NfmOpt, gamma: float64): int32   # AdaGrad mini-batch rule: weight of the batch's gradient cross products in g_norm
{.pop.}

proc check*(rc: int32) =
  ## NFM_ERR_INVALID (-1) is the reference's ValueError, NFM_ERR_NOT_FITTED (-3) its NotFittedError
  ## (model/fm_base.nim:10-15); the rest are runtime failures.
  if rc == 0: return
  let msg = $nfm_last_error()
  if rc == -1: raise newException(ValueError, msg)
  if rc == -3: raise newException(ValueError, msg)   # NotFittedError is private to fm_base (fm_base.nim:10)
  raise newException(IOError, fmt"libnimfm_hip error {rc}: {msg}")

var gCtx: NfmCtx

proc hipContext*(): NfmCtx =
  ## one context (device 0, library-owned stream) per process; a data-parallel host creates one process per GPU
  if gCtx.isNil: check nfm_ctx_create(0, nil, addr gCtx)
  gCtx

# ---------------------------------------------------------------------------------------------------------
# device-resident datasets
# ---------------------------------------------------------------------------------------------------------
type
  HipDatasetObj = object
    handle*: NfmDataset
    nSamples*, nnz*: int
    nFeaturesStored: int      ## columns of the matrix, without augments (what the library is told)
    nFields*: int
    nAugments*: int           ## kept for source compatibility with BaseDataset (dataset.nim:11); dummy features
                              ## are generated inside the kernels, never stored
  HipCSRDataset* = ref HipDatasetObj       ## BaseDataset[CSRMatrix] in HBM (dataset.nim:10-16)
  HipCSRFieldDataset* = ref HipDatasetObj  ## BaseDataset[CSRFieldMatrix] in HBM

proc release(X: ref HipDatasetObj) =
  if not X.handle.isNil:
    discard nfm_dataset_destroy(X.handle)
    X.handle = nil

proc nFeatures*(X: ref HipDatasetObj): int = X.nFeaturesStored + X.nAugments  # dataset.nim:47-48
proc shape*(X: ref HipDatasetObj): array[2, int] = [X.nSamples, X.nFeatures]
proc nCached*(X: ref HipDatasetObj): int = X.nSamples                         # everything is resident

proc adopt(h: NfmDataset): ref HipDatasetObj =
  new(result, release)
  result.handle = h
  var n, d, nnz, nf: int64
  check nfm_dataset_shape(h, addr n, addr d, addr nnz, addr nf)
  (result.nSamples, result.nFeaturesStored, result.nnz, result.nFields) = (n.int, d.int, nnz.int, nf.int)

proc toHip*(X: CSRDataset): HipCSRDataset =
  ## tensor/sparse.nim:9-12: data / indices / indptr are exported seqs of float64 / int / int (Nim int = int64)
  var h: NfmDataset
  let nnz = X.data.data.len
  check nfm_dataset_create_csr(hipContext(), X.nSamples.int64, X.data.shape[1].int64,
                               cast[ptr int64](unsafeAddr X.data.indptr[0]),
                               (if nnz > 0: cast[ptr int64](unsafeAddr X.data.indices[0]) else: nil),
                               (if nnz > 0: unsafeAddr X.data.data[0] else: nil), nil, 0, nil, addr h)
  result = adopt(h)

proc toHip*(X: CSRFieldDataset): HipCSRFieldDataset =
  var h: NfmDataset
  let nnz = X.data.data.len
  check nfm_dataset_create_csr(hipContext(), X.nSamples.int64, X.data.shape[1].int64,
                               cast[ptr int64](unsafeAddr X.data.indptr[0]),
                               (if nnz > 0: cast[ptr int64](unsafeAddr X.data.indices[0]) else: nil),
                               (if nnz > 0: unsafeAddr X.data.data[0] else: nil),
                               (if nnz > 0: cast[ptr int64](unsafeAddr X.data.fields[0]) else: nil),
                               X.nFields.int64, nil, addr h)
  result = adopt(h)

proc targets(X: ref HipDatasetObj): seq[float64] =
  result = newSeq[float64](X.nSamples)
  if X.nSamples > 0: check nfm_dataset_get_targets(X.handle, addr result[0])

proc loadSVMLightFile*(f: string, dataset: var HipCSRDataset, y: var seq[float64], nFeatures: int = -1) =
  ## dataset.nim:616-632, same signature with the device dataset type: the text is parsed on the GPU
  var h: NfmDataset
  check nfm_dataset_load_svmlight(hipContext(), f.cstring, nFeatures.int64, addr h)
  dataset = adopt(h)
  y = targets(dataset)

proc loadFFMFile*(f: string, dataset: var HipCSRFieldDataset, y: var seq[float64], nFeatures: int = -1,
                  nFields: int = -1) =
  ## dataset.nim:768-790
  var h: NfmDataset
  check nfm_dataset_load_ffm(hipContext(), f.cstring, nFeatures.int64, nFields.int64, addr h)
  dataset = adopt(h)
  y = targets(dataset)

proc newHipStreamCSRDataset*(f: string, fY: string = ""): HipCSRDataset =
  ## dataset.nim:170-174 newStreamCSRDataset: the STREAMCSR(FIELD) file becomes resident in HBM
  var h: NfmDataset
  check nfm_dataset_load_stream(hipContext(), f.cstring, (if fY.len > 0: fY.cstring else: nil), addr h)
  result = adopt(h)

# ---------------------------------------------------------------------------------------------------------
# models: flat copies of the reference's jagged containers (the reference itself pays a transpose copy per
# fit: sgd.nim:292,328)
# ---------------------------------------------------------------------------------------------------------
proc flatten*(P: Tensor): seq[float64] =
  result = newSeqOfCap[float64](P.shape[0] * P.shape[1] * P.shape[2])
  for a in 0..<P.shape[0]:
    for b in 0..<P.shape[1]:
      for c in 0..<P.shape[2]: result.add(P[a, b, c])

proc unflatten*(P: var Tensor, flat: seq[float64]) =
  var t = 0
  for a in 0..<P.shape[0]:
    for b in 0..<P.shape[1]:
      for c in 0..<P.shape[2]:
        P[a, b, c] = flat[t]
        inc t

proc ptrOrNil(s: var seq[float64]): ptr float64 = (if s.len > 0: addr s[0] else: nil)

proc push*(fm: FactorizationMachine, nFeatures: int): NfmModel =
  ## nFeatures: columns of the data (fm.P.shape[2] - fm.nAugments)
  var cfg = NfmModelCfg(kind: nfmKindFM, task: (if fm.task == classification: 1 else: 0), degree: fm.degree.int32,
                        nComponents: fm.nComponents.int32, fitLower: ord(fm.fitLower).int32,
                        fitIntercept: fm.fitIntercept.int32, fitLinear: fm.fitLinear.int32,
                        nFeatures: nFeatures.int64, nFields: 0)
  check nfm_model_create(hipContext(), addr cfg, addr result)
  var P = flatten(fm.P)            # [nOrders][nComponents][nFeatures+nAugments], the ABI's layout
  var w = fm.w
  var lams = fm.lams
  check nfm_model_set_params(result, ptrOrNil(P), ptrOrNil(w), fm.intercept, ptrOrNil(lams))

proc push*(ffm: FieldAwareFactorizationMachine): NfmModel =
  var cfg = NfmModelCfg(kind: nfmKindFFM, task: (if ffm.task == classification: 1 else: 0), degree: 2,
                        nComponents: ffm.nComponents.int32, fitLower: 0,
                        fitIntercept: ffm.fitIntercept.int32, fitLinear: ffm.fitLinear.int32,
                        nFeatures: ffm.P.shape[1].int64, nFields: ffm.P.shape[0].int64)
  check nfm_model_create(hipContext(), addr cfg, addr result)
  var P = flatten(ffm.P)           # [nFields][nFeatures][nComponents]
  var w = ffm.w
  check nfm_model_set_params(result, ptrOrNil(P), ptrOrNil(w), ffm.intercept, nil)

proc pull*[FM](fm: FM, m: NfmModel) =
  ## the finalised parameters back into fm.P / fm.w / fm.intercept (call nfm_opt_finalize first)
  var P = newSeq[float64](fm.P.shape[0] * fm.P.shape[1] * fm.P.shape[2])
  check nfm_model_get_params(m, ptrOrNil(P), (if fm.w.len > 0: addr fm.w[0] else: nil), addr fm.intercept)
  unflatten(fm.P, P)

proc decisionFunction*(self: FactorizationMachine, X: HipCSRDataset): seq[float64] =
  ## model/factorization_machine.nim:100-122 on a device-resident dataset.  predict / predictProba / score
  ## (model/fm_base.nim:18-48) are generic over the dataset type and pick this overload up.
  self.checkInitialized()
  if X.nFeaturesStored + self.nAugments != self.P.shape[2]:
    raise newException(ValueError, "Invalid nFeatures.")   # factorization_machine.nim:114-115
  let m = push(self, X.nFeaturesStored)
  result = newSeq[float64](X.nSamples)
  if X.nSamples > 0: check nfm_decision_function(m, X.handle, addr result[0])
  discard nfm_model_destroy(m)

proc decisionFunction*(self: FieldAwareFactorizationMachine, X: HipCSRFieldDataset): seq[float64] =
  ## model/field_aware_factorization_machine.nim:52-76
  self.checkInitialized()
  if X.nFeaturesStored != self.P.shape[1]: raise newException(ValueError, "Invalid nFeatures.")
  if X.nFields != self.P.shape[0]: raise newException(ValueError, "Invalid nFields.")   # :63-64
  let m = push(self)
  result = newSeq[float64](X.nSamples)
  if X.nSamples > 0: check nfm_decision_function(m, X.handle, addr result[0])
  discard nfm_model_destroy(m)

proc hipScore*[FM](self: FM, X: ref HipDatasetObj, y: seq[float64]): float64 =
  ## score (model/fm_base.nim:39-48) reduced on the device: only the scalar comes back
  self.checkInitialized()
  var yy = y
  check nfm_dataset_set_targets(X.handle, addr yy[0])
  let m = (when FM is FactorizationMachine: push(self, X.nFeaturesStored) else: push(self))
  check nfm_score(m, X.handle, addr result)
  discard nfm_model_destroy(m)

proc init*(self: FieldAwareFactorizationMachine, X: HipCSRFieldDataset, force = false) =
  ## model/field_aware_factorization_machine.nim:79-92 (the reference's init takes RowFieldDataset, a closed
  ## type class; FactorizationMachine.init is generic over the dataset and needs no twin)
  if force or not (self.warmStart and self.isInitialized):
    randomize(self.randomState)
    self.w = zeros([X.nFeatures])
    self.P = randomNormal([X.nFields, X.nFeatures, self.nComponents], scale = self.scale)
    self.intercept = 0.0
  self.isInitialized = true

# ---------------------------------------------------------------------------------------------------------
# losses -> ids.  Huber's threshold is private (loss.nim:12); its dloss returns it for any residual beyond
# it (loss.nim:88-91), which reads it back without touching the field.
# ---------------------------------------------------------------------------------------------------------
proc lossId*[L](loss: L): int32 =
  when L is Squared: 0 elif L is SquaredHinge: 1 elif L is Logistic: 2 else: 3

proc lossParam*[L](loss: L): float64 =
  when L is Huber: loss.dloss(0.0, Inf) else: 1.0

# ---------------------------------------------------------------------------------------------------------
# the epoch loop of optimizer/sgd.nim:294-328 / adagrad.nim:164-203 (and their FFM twins), shared by the four
# `fit` overloads.  `it` is the optimizer's own counter (self.it); `pull` finalises on the device and copies the
# parameters into the model; `callback` is the user's callback bound to (self, fm) or nil.
# ---------------------------------------------------------------------------------------------------------
type HipFitCfg* = object
  maxIter*, verbose*, nCalls*: int
  tol*, alpha0*, alpha*, beta*: float64
  shuffle*: bool
  callbackEveryEpochOnly*: bool  ## SGD: per-epoch callback only when nCalls <= 0 (sgd.nim:312); AdaGrad: always (:188)
  minibatch*: bool               ## the maxThreads overloads: callbacks per epoch only (sgd_multi.nim:104-108)

proc hipEpochLoop*(o: NfmOpt, m: NfmModel, ds: NfmDataset, nSamples: int, c: HipFitCfg, it: var int,
                   pull: proc () {.closure.}, callback: proc () {.closure.}) =
  # two index arrays: while epoch e runs over one, the other already holds epoch e+1's order (the same sequence of
  # shuffles of the same array as the reference's, sgd.nim:297 -- drawn one epoch early) and is announced to the library,
  # which builds its batch plan beside the running epoch
  var idx = [toSeq(0..<nSamples), newSeq[int](0)]
  var isConverged = false
  check nfm_opt_set_it(o, it.int64)
  let wholeEpochs = callback.isNil or c.nCalls <= 0 or c.minibatch
  if c.shuffle: shuffle(idx[0])               # sgd.nim:297, Nim's global RNG exactly as in the reference
  for epoch in 0..<c.maxIter:
    var viol, runningLoss: float64
    var perm: ptr int64 = nil
    template indices: untyped = idx[epoch mod 2]
    if c.shuffle:
      perm = cast[ptr int64](addr indices[0])
      if epoch + 1 < c.maxIter:
        idx[(epoch + 1) mod 2] = indices      # a copy
        shuffle(idx[(epoch + 1) mod 2])
        if wholeEpochs and c.minibatch:
          check nfm_opt_announce_perm(o, cast[ptr int64](addr idx[(epoch + 1) mod 2][0]), 0, nSamples.int64)
    if not callback.isNil and c.nCalls > 0 and not c.minibatch:
      # sgd.nim:303-308 / adagrad.nim:180-184: callback whenever it mod nCalls == 0 -- the epoch runs in pieces
      var pos = 0
      while pos < nSamples:
        let toNext = (c.nCalls - it mod c.nCalls) mod c.nCalls + 1
        let last = min(nSamples, pos + toNext)
        var ls, vs: float64
        check nfm_opt_epoch(o, ds, perm, pos.int64, last.int64, addr ls, addr vs)
        runningLoss += ls
        viol += vs
        it += last - pos
        pos = last
        if (it - 1) mod c.nCalls == 0:
          pull()
          dec it                              # the reference calls back before inc(self.it)
          callback()
          inc it
    else:
      check nfm_opt_epoch(o, ds, perm, 0, nSamples.int64, addr runningLoss, addr viol)
      it += nSamples
    runningLoss /= float(nSamples)
    if not callback.isNil and (c.nCalls <= 0 or c.minibatch or not c.callbackEveryEpochOnly):
      pull()
      callback()
    # stoppingCriterion, sgd.nim:72-89 (the regulariser is evaluated on the device copy)
    var isContinue = true
    if runningLoss.classify == fcNan:
      echo("Loss is NaN. Use smaller learning rate.")
      isContinue = false
    if c.verbose > 0:
      var pSq, wSq, b: float64
      check nfm_model_sqnorms(m, addr pSq, addr wSq)
      check nfm_model_get_params(m, nil, nil, addr b)
      echoInfo(epoch+1, c.maxIter, viol, runningLoss, 0.5*c.alpha0*b*b + 0.5*c.alpha*wSq + 0.5*c.beta*pSq)
    if viol < c.tol:
      if c.verbose > 0: echo(fmt"Converged at epoch {epoch}.")
      isConverged = true
      isContinue = false
    if not isContinue: break
  if not isConverged and c.verbose > 0:
    echo("Objective did not converge. Increase maxIter.")
  pull()                                      # finalize + transpose back, sgd.nim:327-328 / adagrad.nim:202-203

proc defaultBatch*(): int =
  ## mini-batch size of the maxThreads overloads when the caller does not pass `miniBatchSize`: the environment's
  ## NIMFM_HIP_BATCH, else 8192 (BASELINE.json configs[2]).  `maxThreads` itself only SELECTS the mini-batch mode -- its
  ## value (a thread count in the reference, optimizer/sgd_multi.nim:13-18) says nothing about a batch size.
  let e = getEnv("NIMFM_HIP_BATCH")
  if e.len > 0: max(1, parseInt(e)) else: 8192

type HipGroup* = ref object
  ## one rank's handle of a data-parallel group (nfm_dp_*): one process per GPU, every rank calls `fit` on its own
  ## contiguous slice of the samples (optimizer/sgd_multi.nim:85-88) and passes its handle as `group`
  handle*: NfmDp

proc newHipGroup*(ctx: NfmCtx, id: array[128, byte], rank, world: int): HipGroup =
  ## `id`: made once by rank 0 (hipGroupId) and carried to the other ranks by whatever the host has (a file, MPI ...)
  new(result)
  var idv = id
  check nfm_dp_create(ctx, addr idv[0], rank.int32, world.int32, addr result.handle)

proc hipGroupId*(): array[128, byte] =
  check nfm_dp_unique_id(addr result[0])

proc close*(g: HipGroup) =
  if not g.isNil and not g.handle.isNil:
    discard nfm_dp_destroy(g.handle)
    g.handle = nil

proc attach*(o: NfmOpt, group: HipGroup, syncPeriod: int) =
  ## the maxThreads overloads across GPUs: replicas reconciled every `syncPeriod` mini-batches (0: at the end of every
  ## epoch only) and exactly at the end of every epoch call
  if not group.isNil: check nfm_opt_set_dp(o, group.handle, syncPeriod.int64, 1)
