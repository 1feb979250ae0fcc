## hip_adagrad_ffm.nim -- INCLUDED by nimfm's optimizer/adagrad_ffm.nim (`when defined(nimfmHip): include
## hip_adagrad_ffm`).  Reuses hipFitAdaGrad of hip_adagrad.nim, which adagrad.nim must export for it:
##     export hipFitAdaGrad        # next to `include hip_adagrad` in optimizer/adagrad.nim
## Overloads of fit(self: AdaGrad[L], X: RowFieldDataset, y, ffm, callback = nil) (optimizer/adagrad_ffm.nim:11-13)
## and of its maxThreads twin (optimizer/adagrad_ffm_multi.nim) for nimfm_hip.HipCSRFieldDataset; the state shape is
## ffm.P's own [nFields][nFeatures][nComponents] (adagrad_ffm.nim:30).
## Not compiled in the build image (no Nim toolchain); see nimfm_hip.nim.
import nimfm_hip

proc fit*[L](self: AdaGrad[L], X: HipCSRFieldDataset, y: seq[float64], ffm: FieldAwareFactorizationMachine,
             callback: (AdaGrad[L], FieldAwareFactorizationMachine)->void = nil) =
  ffm.init(X)
  hipFitAdaGrad(self, X, y, ffm, push(ffm), ffm.P.shape, nfmModeSequential, 1, callback)

proc fit*[L](self: AdaGrad[L], X: HipCSRFieldDataset, y: seq[float64], ffm: FieldAwareFactorizationMachine,
             maxThreads: int, callback: (AdaGrad[L], FieldAwareFactorizationMachine)->void = nil,
             miniBatchSize: int = defaultBatch(), syncPeriod: int = 0, group: HipGroup = nil, adaCross: float64 = 0.0) =
  discard maxThreads  # selects the mini-batch mode; its knobs are the defaulted arguments (hip_sgd.nim, hip_adagrad.nim)
  ffm.init(X)
  hipFitAdaGrad(self, X, y, ffm, push(ffm), ffm.P.shape, nfmModeMinibatch, miniBatchSize, callback, group, syncPeriod, adaCross)
