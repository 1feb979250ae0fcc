## hip_sgd_ffm.nim -- INCLUDED by nimfm's optimizer/sgd_ffm.nim, after its `fit` (`when defined(nimfmHip): include
## hip_sgd_ffm`).  sgd_ffm.nim imports sgd.nim, whose private SGD fields are NOT visible here; the step-size
## parameters are therefore read through hipSgdHyper, which hip_sgd.nim's host module exports for this purpose:
##     proc hipSgdHyper*[L](self: SGD[L]): tuple[eta0, power: float64, scheduling: int32] =
##       (self.eta0, self.power, ord(self.scheduling).int32)          # add next to `include hip_sgd` in sgd.nim
## Overloads of fit(self: SGD[L], X: RowFieldDataset, y, ffm, callback = nil) (optimizer/sgd_ffm.nim:49-51) and of
## its maxThreads twin (optimizer/sgd_ffm_multi.nim) for nimfm_hip.HipCSRFieldDataset.
## Not compiled in the build image (no Nim toolchain); see nimfm_hip.nim.
import nimfm_hip

proc hipFitSGDFFM[L](self: SGD[L], X: HipCSRFieldDataset, y: seq[float64], ffm: FieldAwareFactorizationMachine,
                     mode: int32, batch: int, callback: (SGD[L], FieldAwareFactorizationMachine)->void,
                     group: HipGroup = nil, syncPeriod = 0, touchCap = 1.0) =
  ffm.init(X)                                   # nimfm_hip.init: field_aware_factorization_machine.nim:79-92
  var yy = ffm.checkTarget(y)
  if yy.len != X.nSamples: raise newException(ValueError, "len(y) != nSamples")
  check nfm_dataset_set_targets(X.handle, addr yy[0])
  if not ffm.warmStart: self.init()             # sgd_ffm.nim:70-71
  let m = push(ffm)
  let h = hipSgdHyper(self)
  var cfg = NfmSgdCfg(eta0: h.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, power: h.power,
                      lossParam: lossParam(self.loss), loss: lossId(self.loss), scheduling: h.scheduling,
                      mode: mode, batch: batch.int64)
  var o: NfmOpt
  check nfm_sgd_create(m, addr cfg, addr o)
  attach(o, group, syncPeriod)
  if mode == nfmModeMinibatch and touchCap != 1.0: check nfm_opt_set_touch_cap(o, touchCap)
  let fc = HipFitCfg(maxIter: self.maxIter, verbose: self.verbose, nCalls: self.nCalls, tol: self.tol,
                     alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, shuffle: self.shuffle,
                     callbackEveryEpochOnly: true, minibatch: mode == nfmModeMinibatch)
  let pullBack = proc () =
    check nfm_opt_finalize(o)
    pull(ffm, m)
  var cb: proc () {.closure.} = nil
  if not callback.isNil: cb = proc () = callback(self, ffm)
  try:
    hipEpochLoop(o, m, X.handle, X.nSamples, fc, self.it, pullBack, cb)
  finally:
    discard nfm_opt_destroy(o)
    discard nfm_model_destroy(m)

proc fit*[L](self: SGD[L], X: HipCSRFieldDataset, y: seq[float64], ffm: FieldAwareFactorizationMachine,
             callback: (SGD[L], FieldAwareFactorizationMachine)->void = nil) =
  ## optimizer/sgd_ffm.nim:49-106
  hipFitSGDFFM(self, X, y, ffm, nfmModeSequential, 1, callback)

proc fit*[L](self: SGD[L], X: HipCSRFieldDataset, y: seq[float64], ffm: FieldAwareFactorizationMachine,
             maxThreads: int, callback: (SGD[L], FieldAwareFactorizationMachine)->void = nil,
             miniBatchSize: int = defaultBatch(), syncPeriod: int = 0, group: HipGroup = nil, touchCap: float64 = 1.0) =
  ## optimizer/sgd_ffm_multi.nim -> the deterministic mini-batch mode; maxThreads only selects it (hip_sgd.nim)
  discard maxThreads
  hipFitSGDFFM(self, X, y, ffm, nfmModeMinibatch, miniBatchSize, callback, group, syncPeriod, touchCap)
