## hip_mbpsgd.nim -- INCLUDED by nimfm's optimizer/minibatch_psgd.nim (`when defined(nimfmHip): include hip_mbpsgd`):
## MBPSGD keeps eta0 / scheduling / power / miniBatchSize / maxIterInner / shuffle / it private
## (optimizer/minibatch_psgd.nim:13-22).  Overload of fit(self: MBPSGD[L, R], X, y, sfm, callback = nil) (:125-210)
## for nimfm_hip.HipCSRDataset (SURVEY 8f rank 3): the gradient of a mini-batch, the step on all parameters and the
## regulariser's proximal operator run on the device; the outer loop, the index stream (indices[ii] with wrap-around and
## reshuffle, :98-108), the stopping rule (:201-204) and the verbose lines stay here.
## Not compiled in the build image (no Nim toolchain); see nimfm_hip.nim.
import nimfm_hip

proc regId(reg: L1): int32 = 0
proc regId(reg: L21): int32 = 1
proc regId(reg: SquaredL12): int32 = 2
proc regId(reg: SquaredL21): int32 = 3

proc fit*[L, R](self: MBPSGD[L, R], X: HipCSRDataset, y: seq[float64], sfm: FactorizationMachine,
                callback: (MBPSGD[L, R], FactorizationMachine)->void = nil) =
  sfm.init(X)
  var yy = sfm.checkTarget(y)
  check nfm_dataset_set_targets(X.handle, addr yy[0])
  if not sfm.warmStart: self.it = 1                               # :153-154
  let nSamples = X.nSamples
  var B = self.miniBatchSize
  if B <= 0: B = max((X.nFeatures * nSamples) div X.nnz, 1)       # :160-163
  var inner = self.maxIterInner
  if inner <= 0: inner = max((nSamples-1) div B + 1, 1)           # :164-167
  let m = push(sfm, sfm.P.shape[2] - sfm.nAugments)
  var cfg = NfmMbpsgdCfg(eta0: self.eta0, alpha0: self.alpha0, alpha: self.alpha, beta: self.beta, gamma: self.gamma,
                         power: self.power, lossParam: lossParam(self.loss), loss: lossId(self.loss),
                         scheduling: ord(self.scheduling).int32, reg: regId(self.reg),
                         regTranspose: (when compiles(self.reg.transpose): self.reg.transpose.int32 else: 0), batch: B.int64)
  var o: NfmOpt
  check nfm_mbpsgd_create(m, addr cfg, addr o)                    # ValueError for SquaredL12/21 with degree != 2
  check nfm_opt_set_it(o, self.it.int64)
  var indices = toSeq(0..<nSamples)
  var stream = newSeq[int](B * inner)
  var ii = 0
  if self.shuffle: shuffle(indices)                               # :169-170
  var oldLossVal = Inf
  var isConverged = false
  try:
    for epoch in 0..<self.maxIter:
      for q in 0..<stream.len:                                    # :98-108: indices[ii], wrap and reshuffle
        stream[q] = indices[ii]
        inc(ii)
        if ii >= nSamples:
          ii = 0
          if self.shuffle: shuffle(indices)
      var lossSum, viol: float64
      check nfm_opt_epoch(o, X.handle, cast[ptr int64](addr stream[0]), 0, stream.len.int64, addr lossSum, addr viol)
      self.it += inner
      let runningLoss = lossSum / float(B * inner)                # :122
      if not callback.isNil:                                      # :185-187
        check nfm_opt_finalize(o)
        pull(sfm, m)
        callback(self, sfm)
      if runningLoss.classify == fcNan:
        echo("Loss is NaN. Use smaller learning rate.")
        break
      if self.verbose > 0: echo fmt"{epoch+1:<5}   {runningLoss:<10.4e}"
      if abs(oldLossVal - runningLoss) < self.tol:                # :201-204
        if self.verbose > 0: echo("Converged at epoch ", epoch+1, ".")
        isConverged = true
        break
      oldLossVal = runningLoss
    if not isConverged and self.verbose > 0:
      echo("Objective did not converge. Increase maxIter.")
    check nfm_opt_finalize(o)
    pull(sfm, m)                                                  # pgd.finalize, optimizer/pgd.nim:45-51
  finally:
    discard nfm_opt_destroy(o)
    discard nfm_model_destroy(m)
