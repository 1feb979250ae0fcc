## hip_adagrad.nim -- INCLUDED by nimfm's optimizer/adagrad.nim (`when defined(nimfmHip): include hip_adagrad`): it
## reads the private AdaGrad.eps (optimizer/adagrad.nim:13).  Overloads of
##     fit(self: AdaGrad[L], X, y, fm, callback = nil)              optimizer/adagrad.nim:137-139
##     fit(self: AdaGrad[L], X, y, fm, maxThreads, callback = nil)  optimizer/adagrad_multi.nim:39-41
## for nimfm_hip.HipCSRDataset.  The optimizer's state stays where the reference keeps it -- self.g_sum / self.g_norm
## (:15-16), created or shape-checked by the reference's own init (:47-62) -- and travels to the device at the
## start of a warm-started fit and back at its end, so `warmStart` continues a run exactly as in the reference.
## Not compiled in the build image (no Nim toolchain); see nimfm_hip.nim.
import nimfm_hip

proc statePush(o: NfmOpt, s, n: Params) =
  var gs = flatten(s.P)     # [nOrders][nFeatures+nAugments][nComponents]: the ABI's state layout
  var gn = flatten(n.P)
  var sw = s.w
  var nw = n.w
  check nfm_opt_set_state(o, (if gs.len > 0: addr gs[0] else: nil), (if gn.len > 0: addr gn[0] else: nil),
                          addr sw[0], addr nw[0], s.intercept, n.intercept)

proc statePull(o: NfmOpt, s, n: Params) =
  var gs = newSeq[float64](s.P.shape[0] * s.P.shape[1] * s.P.shape[2])
  var gn = newSeq[float64](gs.len)
  check nfm_opt_get_state(o, (if gs.len > 0: addr gs[0] else: nil), (if gn.len > 0: addr gn[0] else: nil),
                          addr s.w[0], addr n.w[0], addr s.intercept, addr n.intercept)
  unflatten(s.P, gs)
  unflatten(n.P, gn)

proc hipFitAdaGrad[L, FM, DS](self: AdaGrad[L], X: DS, y: seq[float64], fm: FM, m: NfmModel, stateShape: array[3, int],
                              mode: int32, batch: int, callback: (AdaGrad[L], FM)->void, group: HipGroup = nil,
                              syncPeriod = 0, adaCross = 0.0) =
  var yy = fm.checkTarget(y)
  if yy.len != X.nSamples: raise newException(ValueError, "len(y) != nSamples")
  check nfm_dataset_set_targets(X.handle, addr yy[0])
  # adagrad.nim:47-62: it = 1 unless warmStart; state created at it == 1, ValueError on a shape mismatch otherwise
  var shapeOnly: Tensor = zeros(stateShape)
  init(self, shapeOnly, fm.w, fm.warmStart, fm.fitLinear, fm.fitIntercept)
  var cfg = NfmAdaGradCfg(eta0: self.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, eps: self.eps,
                          lossParam: lossParam(self.loss), loss: lossId(self.loss), mode: mode, trackViol: 1,
                          batch: batch.int64)
  var o: NfmOpt
  check nfm_adagrad_create(m, addr cfg, addr o)
  attach(o, group, syncPeriod)
  if mode == nfmModeMinibatch and adaCross != 0.0: check nfm_opt_set_ada_cross(o, adaCross)
  if self.it != 1: statePush(o, self.g_sum, self.g_norm)   # a warm start continues from the object's state
  let fc = HipFitCfg(maxIter: self.maxIter, verbose: self.verbose, nCalls: self.nCalls, tol: self.tol,
                     alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, shuffle: self.shuffle,
                     callbackEveryEpochOnly: false, minibatch: mode == nfmModeMinibatch)
  let pullBack = proc () =
    check nfm_opt_finalize(o)                   # finalize, adagrad.nim:65-84
    pull(fm, m)
  var cb: proc () {.closure.} = nil
  if not callback.isNil: cb = proc () = callback(self, fm)
  try:
    hipEpochLoop(o, m, X.handle, X.nSamples, fc, self.it, pullBack, cb)
    statePull(o, self.g_sum, self.g_norm)       # g_sum / g_norm are exported fields users may inspect (:15-16)
  finally:
    discard nfm_opt_destroy(o)
    discard nfm_model_destroy(m)

proc fit*[L](self: AdaGrad[L], X: HipCSRDataset, y: seq[float64], fm: FactorizationMachine,
             callback: (AdaGrad[L], FactorizationMachine)->void = nil) =
  ## optimizer/adagrad.nim:137-203
  fm.init(X)
  let m = push(fm, fm.P.shape[2] - fm.nAugments)
  hipFitAdaGrad(self, X, y, fm, m, [fm.P.shape[0], fm.P.shape[2], fm.P.shape[1]], nfmModeSequential, 1, callback)

proc fit*[L](self: AdaGrad[L], X: HipCSRDataset, y: seq[float64], fm: FactorizationMachine, maxThreads: int,
             callback: (AdaGrad[L], FactorizationMachine)->void = nil, miniBatchSize: int = defaultBatch(),
             syncPeriod: int = 0, group: HipGroup = nil, adaCross: float64 = 0.0) =
  ## optimizer/adagrad_multi.nim:39-115 -> the deterministic mini-batch mode; maxThreads only selects it, the mode's own
  ## knobs are the defaulted miniBatchSize / syncPeriod / group (hip_sgd.nim) and adaCross: the weight of the batch's gradient
  ## cross products in g_norm (nfm_opt_set_ada_cross; 0.1 lets batches of ten touches per coordinate learn like small ones)
  discard maxThreads
  fm.init(X)
  let m = push(fm, fm.P.shape[2] - fm.nAugments)
  hipFitAdaGrad(self, X, y, fm, m, [fm.P.shape[0], fm.P.shape[2], fm.P.shape[1]], nfmModeMinibatch, miniBatchSize, callback,
                group, syncPeriod, adaCross)
