## nimfm_hip.nim -- Nim shim: nimfm's FM surface over libnimfm_hip.so (include/nimfm_hip.h).
##
## Drop this file next to nimfm (`import nimfm, nimfm_hip`) and replace
##     sgd.fit(X, y, fm)                 optimizer/sgd.nim:261-328
##     adagrad.fit(X, y, fm)             optimizer/adagrad.nim:137-203
##     sgd.fit(X, y, fm, maxThreads)     optimizer/sgd_multi.nim:40-120 (Hogwild -> mini-batch mode)
##     fm.decisionFunction(X)            model/factorization_machine.nim:100-122
## by `hipFit(sgd, X, y, fm)` / `hipDecisionFunction(fm, X)`; everything else (datasets, loaders,
## dump/load, metrics, CLI) stays nimfm's.  The epoch loop, shuffle (Nim's global RNG, sgd.nim:297),
## stopping criterion and verbose output stay in Nim exactly as in the reference; only the
## per-sample work crosses the FFI.
##
## NOTE: written against the C header; NOT compiled in the build image (no Nim toolchain there,
## SURVEY.md section 0).  The executable stand-in with the same control flow is nimfm_amd/host.py.
import nimfm
import std/[math, random, sequtils, strformat]

const libnfm = "libnimfm_hip.so"

type
  NfmCtx = pointer
  NfmDataset = pointer
  NfmModel = pointer
  NfmOpt = pointer
  NfmModelCfg {.bycopy.} = object
    kind, task, degree, nComponents, fitLower, fitIntercept, fitLinear, reserved: int32
    nFeatures, nFields: int64
  NfmSgdCfg {.bycopy.} = object
    eta0, alpha0, alpha, beta, power, lossParam: float64
    loss, scheduling, mode, reserved: int32
    batch: int64
  NfmAdaGradCfg {.bycopy.} = object
    eta0, alpha0, alpha, beta, eps, lossParam: float64
    loss, mode, trackViol, reserved: int32
    batch: int64
  NfmMbpsgdCfg {.bycopy.} = object
    eta0, alpha0, alpha, beta, gamma, power, lossParam: float64
    loss, scheduling, reg, regTranspose: int32
    batch: int64

{.push importc, cdecl, dynlib: libnfm.}
proc nfm_last_error(): cstring
proc nfm_ctx_create(deviceId: int32, stream: pointer, outp: ptr NfmCtx): int32
proc nfm_ctx_destroy(ctx: NfmCtx): int32
proc nfm_dataset_create_csr(ctx: NfmCtx, n, d: int64, indptr, indices: ptr int64, data: ptr float64,
                            fields: ptr int64, nFields: int64, y: ptr float64, outp: ptr NfmDataset): int32
proc nfm_dataset_set_targets(ds: NfmDataset, y: ptr float64): int32
proc nfm_dataset_load_svmlight(ctx: NfmCtx, path: cstring, nFeatures: int64, outp: ptr NfmDataset): int32
proc nfm_dataset_load_ffm(ctx: NfmCtx, path: cstring, nFeatures, nFields: int64, outp: ptr NfmDataset): int32
proc nfm_dataset_shape(ds: NfmDataset, nSamples, nFeatures, nnz, nFields: ptr int64): int32
proc nfm_dataset_get_targets(ds: NfmDataset, y: ptr float64): int32
proc nfm_dataset_destroy(ds: NfmDataset): int32
proc nfm_model_create(ctx: NfmCtx, cfg: ptr NfmModelCfg, outp: ptr NfmModel): int32
proc nfm_model_set_params(m: NfmModel, P, w: ptr float64, intercept: float64, lams: ptr float64): int32
proc nfm_model_get_params(m: NfmModel, P, w: ptr float64, intercept: ptr float64): int32
proc nfm_decision_function(m: NfmModel, ds: NfmDataset, outp: ptr float64): int32
proc nfm_model_sqnorms(m: NfmModel, pSq, wSq: ptr float64): int32
proc nfm_model_destroy(m: NfmModel): int32
proc nfm_sgd_create(m: NfmModel, cfg: ptr NfmSgdCfg, outp: ptr NfmOpt): int32
proc nfm_adagrad_create(m: NfmModel, cfg: ptr NfmAdaGradCfg, outp: ptr NfmOpt): int32
proc nfm_mbpsgd_create(m: NfmModel, cfg: ptr NfmMbpsgdCfg, outp: ptr NfmOpt): int32
proc nfm_opt_predict_all_with_grad(o: NfmOpt, ds: NfmDataset, yPred, dL, gradP, gradW, gradB, lossSum: ptr float64): int32
proc nfm_opt_set_it(o: NfmOpt, it: int64): int32
proc nfm_opt_epoch(o: NfmOpt, ds: NfmDataset, perm: ptr int64, first, last: int64,
                   lossSum, violSum: ptr float64): int32
proc nfm_opt_finalize(o: NfmOpt): int32
proc nfm_opt_destroy(o: NfmOpt): int32
{.pop.}

proc check(rc: int32) =
  ## NFM_ERR_INVALID (-1) is the reference's ValueError; the rest are runtime failures.
  if rc == 0: return
  let msg = $nfm_last_error()
  if rc == -1: raise newException(ValueError, msg)
  raise newException(IOError, fmt"libnimfm_hip error {rc}: {msg}")

var gCtx: NfmCtx

proc ctx(): NfmCtx =
  if gCtx.isNil: check nfm_ctx_create(0, nil, addr gCtx)
  gCtx

# ---- flat copies of the reference's jagged containers (the reference itself pays a transpose
# ---- copy per fit: sgd.nim:292,328) ----
proc flatten(P: Tensor): seq[float64] =
  result = newSeqOfCap[float64](P.shape[0] * P.shape[1] * P.shape[2])
  for a in 0..<P.shape[0]:
    for b in 0..<P.shape[1]:
      for c in 0..<P.shape[2]: result.add(P[a, b, c])

proc unflatten(P: var Tensor, flat: seq[float64]) =
  var t = 0
  for a in 0..<P.shape[0]:
    for b in 0..<P.shape[1]:
      for c in 0..<P.shape[2]:
        P[a, b, c] = flat[t]
        inc t

proc toDevice(X: CSRDataset): NfmDataset =
  ## tensor/sparse.nim:9-12: data / indices / indptr are exported seqs of float64 / int / int
  var indptr = X.data.indptr
  var indices = X.data.indices
  var data = X.data.data
  check nfm_dataset_create_csr(ctx(), X.nSamples.int64, (X.nFeatures - X.nAugments).int64,
                               cast[ptr int64](addr indptr[0]), cast[ptr int64](addr indices[0]),
                               addr data[0], nil, 0, nil, addr result)

type
  HipCSRDataset* = ref object
    ## a dataset that lives in HBM only: made by the GPU loaders below, consumed by hipFit / hipDecisionFunction
    ## overloads that take the handle instead of uploading a CSRDataset
    handle*: NfmDataset
    nSamples*, nFeatures*, nnz*, nFields*: int

proc hipLoadSVMLightFile*(f: string, dataset: var HipCSRDataset, y: var seq[float64], nFeatures: int = -1) =
  ## dataset.nim:616-632 loadSVMLightFile: the text is parsed on the GPU, the CSR never exists on the host
  new(dataset)
  check nfm_dataset_load_svmlight(ctx(), f.cstring, nFeatures.int64, addr dataset.handle)
  var n, d, nnz, nf: int64
  check nfm_dataset_shape(dataset.handle, addr n, addr d, addr nnz, addr nf)
  (dataset.nSamples, dataset.nFeatures, dataset.nnz, dataset.nFields) = (n.int, d.int, nnz.int, nf.int)
  y = newSeq[float64](n.int)
  if n > 0: check nfm_dataset_get_targets(dataset.handle, addr y[0])

proc hipLoadFFMFile*(f: string, dataset: var HipCSRDataset, y: var seq[float64], nFeatures: int = -1, nFields: int = -1) =
  ## dataset.nim:768-790 loadFFMFile
  new(dataset)
  check nfm_dataset_load_ffm(ctx(), f.cstring, nFeatures.int64, nFields.int64, addr dataset.handle)
  var n, d, nnz, nf: int64
  check nfm_dataset_shape(dataset.handle, addr n, addr d, addr nnz, addr nf)
  (dataset.nSamples, dataset.nFeatures, dataset.nnz, dataset.nFields) = (n.int, d.int, nnz.int, nf.int)
  y = newSeq[float64](n.int)
  if n > 0: check nfm_dataset_get_targets(dataset.handle, addr y[0])

proc lossId[L](loss: L): int32 =
  when L is Squared: 0 elif L is SquaredHinge: 1 elif L is Logistic: 2 else: 3

proc modelHandle(fm: FactorizationMachine, d: int): NfmModel =
  var cfg = NfmModelCfg(kind: 0, task: (if fm.task == classification: 1 else: 0), degree: fm.degree.int32,
                        nComponents: fm.nComponents.int32, fitLower: ord(fm.fitLower).int32,
                        fitIntercept: fm.fitIntercept.int32, fitLinear: fm.fitLinear.int32,
                        nFeatures: d.int64, nFields: 0)
  check nfm_model_create(ctx(), addr cfg, addr result)
  var P = flatten(fm.P)
  var w = fm.w
  check nfm_model_set_params(result, (if P.len > 0: addr P[0] else: nil), addr w[0], fm.intercept, addr fm.lams[0])

proc pull(fm: FactorizationMachine, m: NfmModel) =
  var P = newSeq[float64](fm.P.shape[0] * fm.P.shape[1] * fm.P.shape[2])
  check nfm_model_get_params(m, (if P.len > 0: addr P[0] else: nil), addr fm.w[0], addr fm.intercept)
  unflatten(fm.P, P)

proc hipDecisionFunction*(fm: FactorizationMachine, X: CSRDataset): seq[float64] =
  ## model/factorization_machine.nim:100-122
  fm.checkInitialized()
  let ds = toDevice(X)
  let m = modelHandle(fm, X.nFeatures)
  result = newSeq[float64](X.nSamples)
  check nfm_decision_function(m, ds, addr result[0])
  discard nfm_model_destroy(m)
  discard nfm_dataset_destroy(ds)

proc runFit(o: NfmOpt, m: NfmModel, ds: NfmDataset, fm: FactorizationMachine, nSamples, maxIter, verbose: int,
            tol, alpha0, alpha, beta: float64, shuffleOn: bool, it: var int) =
  ## the epoch loop of optimizer/sgd.nim:294-328 / adagrad.nim:164-203
  var indices = toSeq(0..<nSamples)
  var isConverged = false
  check nfm_opt_set_it(o, it.int64)
  for epoch in 0..<maxIter:
    var viol, runningLoss: float64
    var perm: ptr int64 = nil
    if shuffleOn:
      shuffle(indices)                       # sgd.nim:297, Nim's global RNG as in the reference
      perm = cast[ptr int64](addr indices[0])
    check nfm_opt_epoch(o, ds, perm, 0, nSamples.int64, addr runningLoss, addr viol)
    it += nSamples
    runningLoss /= float(nSamples)
    if runningLoss.classify == fcNan:        # stoppingCriterion, sgd.nim:72-89
      echo("Loss is NaN. Use smaller learning rate.")
      break
    if verbose > 0:
      var pSq, wSq, b: float64
      check nfm_model_sqnorms(m, addr pSq, addr wSq)
      check nfm_model_get_params(m, nil, nil, addr b)
      echo fmt"{epoch+1:<5}   {viol:<10.4e}   {runningLoss:<10.4e}   {0.5*alpha0*b*b + 0.5*alpha*wSq + 0.5*beta*pSq:<10.4e}"
    if viol < tol:
      if verbose > 0: echo(fmt"Converged at epoch {epoch}.")
      isConverged = true
      break
  if not isConverged and verbose > 0:
    echo("Objective did not converge. Increase maxIter.")
  check nfm_opt_finalize(o)                  # sgd.nim:327-328 / adagrad.nim:202-203
  pull(fm, m)

proc hipFit*[L](self: SGD[L], X: CSRDataset, y: seq[float64], fm: FactorizationMachine,
                maxThreads = 0, batch = 8192) =
  ## optimizer/sgd.nim:261-328; with maxThreads != 0 the Hogwild overload (sgd_multi.nim:40-42)
  ## is served by the deterministic mini-batch mode.
  fm.init(X)
  if not fm.warmStart: self.it = 1
  let ds = toDevice(X)
  var yy = y
  check nfm_dataset_set_targets(ds, addr yy[0])   # checkTarget is applied on the device
  let m = modelHandle(fm, X.nFeatures)
  var cfg = NfmSgdCfg(eta0: self.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta,
                      power: self.power, lossParam: 1.0, loss: lossId(self.loss),
                      scheduling: ord(self.scheduling).int32, mode: (if maxThreads != 0: 1 else: 0),
                      batch: batch.int64)
  var o: NfmOpt
  check nfm_sgd_create(m, addr cfg, addr o)
  runFit(o, m, ds, fm, X.nSamples, self.maxIter, self.verbose, self.tol, self.alpha0, self.alpha, self.beta,
         self.shuffle, self.it)
  discard nfm_opt_destroy(o); discard nfm_model_destroy(m); discard nfm_dataset_destroy(ds)

proc hipFit*[L](self: AdaGrad[L], X: CSRDataset, y: seq[float64], fm: FactorizationMachine,
                maxThreads = 0, batch = 8192) =
  ## optimizer/adagrad.nim:137-203 (state kept across calls needs a persistent NfmOpt: omitted here
  ## for brevity, see nimfm_amd/host.py::_OptimizerBase._handle).
  fm.init(X)
  if not fm.warmStart: self.it = 1
  let ds = toDevice(X)
  var yy = y
  check nfm_dataset_set_targets(ds, addr yy[0])
  let m = modelHandle(fm, X.nFeatures)
  var cfg = NfmAdaGradCfg(eta0: self.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta,
                          eps: 1e-10, lossParam: 1.0, loss: lossId(self.loss),
                          mode: (if maxThreads != 0: 1 else: 0), trackViol: 1, batch: batch.int64)
  var o: NfmOpt
  check nfm_adagrad_create(m, addr cfg, addr o)
  runFit(o, m, ds, fm, X.nSamples, self.maxIter, self.verbose, self.tol, self.alpha0, self.alpha, self.beta,
         self.shuffle, self.it)
  discard nfm_opt_destroy(o); discard nfm_model_destroy(m); discard nfm_dataset_destroy(ds)

# ---- mini-batch proximal SGD (optimizer/minibatch_psgd.nim; SURVEY 8f rank 3) ----
# MBPSGD keeps eta0 / scheduling / power / miniBatchSize / maxIterInner / shuffle / it private
# (minibatch_psgd.nim:13-22), so a binding outside that module takes them as arguments; inside the module
# this is `proc hipFit*[L, R](self: MBPSGD[L, R], X, y, sfm)` reading the fields.
proc regId(reg: L1): int32 = 0
proc regId(reg: L21): int32 = 1
proc regId(reg: SquaredL12): int32 = 2
proc regId(reg: SquaredL21): int32 = 3

proc hipFitMBPSGD*[L, R](X: CSRDataset, y: seq[float64], sfm: FactorizationMachine, loss: L, reg: R,
                         regTranspose: bool, maxIter = 100, eta0 = 0.1, alpha0 = 1e-6, alpha = 1e-3, beta = 1e-4,
                         gamma = 1e-4, miniBatchSize = -1, maxIterInner = -1, scheduling = optimal, power = 1.0,
                         verbose = 1, tol = 1e-6, shuffleOn = true, it: var int) =
  sfm.init(X)
  if not sfm.warmStart: it = 1                                  # minibatch_psgd.nim:153-154
  let nSamples = X.nSamples
  var B = miniBatchSize
  if B <= 0: B = max((X.nFeatures * nSamples) div X.nnz, 1)     # :160-163
  var inner = maxIterInner
  if inner <= 0: inner = max((nSamples-1) div B + 1, 1)         # :164-167
  let ds = toDevice(X)
  var yy = y
  check nfm_dataset_set_targets(ds, addr yy[0])
  let m = modelHandle(sfm, X.nFeatures)
  var cfg = NfmMbpsgdCfg(eta0: eta0, alpha0: alpha0, alpha: alpha, beta: beta, gamma: gamma, power: power,
                         lossParam: 1.0, loss: lossId(loss), scheduling: ord(scheduling).int32,
                         reg: regId(reg), regTranspose: regTranspose.int32, batch: B.int64)
  var o: NfmOpt
  check nfm_mbpsgd_create(m, addr cfg, addr o)                  # ValueError for SquaredL12/21 with degree != 2
  check nfm_opt_set_it(o, it.int64)
  var indices = toSeq(0..<nSamples)
  var stream = newSeq[int](B * inner)
  var ii = 0
  if shuffleOn: shuffle(indices)                                # :169-170
  var oldLossVal = Inf
  var isConverged = false
  for epoch in 0..<maxIter:
    for q in 0..<stream.len:                                    # :98-108: indices[ii], wrap and reshuffle
      stream[q] = indices[ii]
      inc(ii)
      if ii >= nSamples:
        ii = 0
        if shuffleOn: shuffle(indices)
    var lossSum, viol: float64
    check nfm_opt_epoch(o, ds, cast[ptr int64](addr stream[0]), 0, stream.len.int64, addr lossSum, addr viol)
    it += inner
    let runningLoss = lossSum / float(B * inner)                # :122
    if runningLoss.classify == fcNan:
      echo("Loss is NaN. Use smaller learning rate.")
      break
    if verbose > 0: echo fmt"{epoch+1:<5}   {runningLoss:<10.4e}"
    if abs(oldLossVal - runningLoss) < tol:                     # :201-204
      if verbose > 0: echo("Converged at epoch ", epoch+1, ".")
      isConverged = true
      break
    oldLossVal = runningLoss
  if not isConverged and verbose > 0:
    echo("Objective did not converge. Increase maxIter.")
  check nfm_opt_finalize(o)
  pull(sfm, m)                                                  # pgd.finalize, optimizer/pgd.nim:45-51
  discard nfm_opt_destroy(o); discard nfm_model_destroy(m); discard nfm_dataset_destroy(ds)
