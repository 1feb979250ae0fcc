## hip_sgd.nim -- INCLUDED by nimfm's optimizer/sgd.nim (`when defined(nimfmHip): include hip_sgd`), so it sees the
## private fields SGD.eta0 / scheduling / power (optimizer/sgd.nim:15-17).  Adds the overloads of the reference's
##     fit(self: SGD[L], X, y, fm, callback = nil)                 optimizer/sgd.nim:261-263
##     fit(self: SGD[L], X, y, fm, maxThreads, callback = nil)     optimizer/sgd_multi.nim:40-42
## for a device-resident dataset (nimfm_hip.HipCSRDataset); everything per-sample runs in libnimfm_hip.so.
## Not compiled in the build image (no Nim toolchain); see nimfm_hip.nim.
import nimfm_hip

proc hipFitSGD[L](self: SGD[L], X: HipCSRDataset, y: seq[float64], fm: FactorizationMachine, mode: int32,
                  batch: int, callback: (SGD[L], FactorizationMachine)->void, group: HipGroup = nil, syncPeriod = 0,
                  touchCap = 1.0) =
  fm.init(X)                                    # generic over the dataset: factorization_machine.nim:125-139
  var yy = fm.checkTarget(y)                    # fm_base.nim:29-36 (the device applies the same rule by task)
  if yy.len != X.nSamples: raise newException(ValueError, "len(y) != nSamples")
  check nfm_dataset_set_targets(X.handle, addr yy[0])
  if not fm.warmStart: self.init()              # sgd.nim:54-56,288-289: it = 1, echoHeader
  let m = push(fm, fm.P.shape[2] - fm.nAugments)
  var cfg = NfmSgdCfg(eta0: self.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, power: self.power,
                      lossParam: lossParam(self.loss), loss: lossId(self.loss),
                      scheduling: ord(self.scheduling).int32, mode: mode, batch: batch.int64)
  var o: NfmOpt
  check nfm_sgd_create(m, addr cfg, addr o)
  attach(o, group, syncPeriod)
  if mode == nfmModeMinibatch and touchCap != 1.0: check nfm_opt_set_touch_cap(o, touchCap)
  let fc = HipFitCfg(maxIter: self.maxIter, verbose: self.verbose, nCalls: self.nCalls, tol: self.tol,
                     alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, shuffle: self.shuffle,
                     callbackEveryEpochOnly: true, minibatch: mode == nfmModeMinibatch)
  let pullBack = proc () =
    check nfm_opt_finalize(o)                   # finalize, sgd.nim:99-113
    pull(fm, m)
  var cb: proc () {.closure.} = nil
  if not callback.isNil: cb = proc () = callback(self, fm)
  try:
    hipEpochLoop(o, m, X.handle, X.nSamples, fc, self.it, pullBack, cb)
  finally:
    discard nfm_opt_destroy(o)
    discard nfm_model_destroy(m)

proc fit*[L](self: SGD[L], X: HipCSRDataset, y: seq[float64], fm: FactorizationMachine,
             callback: (SGD[L], FactorizationMachine)->void = nil) =
  ## optimizer/sgd.nim:261-328 -- the reference's sample-by-sample order (NFM_MODE_SEQUENTIAL)
  hipFitSGD(self, X, y, fm, nfmModeSequential, 1, callback)

proc fit*[L](self: SGD[L], X: HipCSRDataset, y: seq[float64], fm: FactorizationMachine, maxThreads: int,
             callback: (SGD[L], FactorizationMachine)->void = nil, miniBatchSize: int = defaultBatch(),
             syncPeriod: int = 0, group: HipGroup = nil, touchCap: float64 = 1.0) =
  ## optimizer/sgd_multi.nim:40-120: the Hogwild overload is served by the deterministic mini-batch mode.  `maxThreads`
  ## keeps its place in the signature and only selects this mode: a thread count is not a batch size.  The knobs of the
  ## mode are explicit and defaulted: `miniBatchSize` (NIMFM_HIP_BATCH, else 8192), and across GPUs -- one process per
  ## GPU, X being this rank's slice -- `group` with `syncPeriod` mini-batches between exchanges.  `touchCap`: how many of a
  ## batch's steps on one coordinate are summed before averaging sets in (nfm_opt_set_touch_cap; 1 = the mean; about twice
  ## miniBatchSize * entries per row / nFeatures keeps the sequential order's epochs to a given loss, INTEGRATION.md).
  discard maxThreads
  hipFitSGD(self, X, y, fm, nfmModeMinibatch, miniBatchSize, callback, group, syncPeriod, touchCap)
