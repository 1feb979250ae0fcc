"""nimfm_amd -- MI355X (gfx950) hot path for nimfm factorization machines.

Layout:
  csrc/      hand-written HIP kernels + the C ABI (include/nimfm_hip.h) -> lib/libnimfm_hip.so
  _capi.py   ctypes binding of that ABI (fails loudly when the library is missing)
  host.py    host-side mirror of the reference's Nim surface (newSGD(...).fit(X, y, fm), ...)
  dp.py      handles of the library's data-parallel groups (nfm_dp_*: RCCL between processes, peer sums inside one)
"""
from ._capi import NfmError, NotFittedError, build, lib  # noqa: F401
from .host import (L1, L21, MBPSGD, SquaredL12, SquaredL21, newL1, newL21, newMBPSGD, predictAllWithGrad, newSquaredL12, newSquaredL21,  # noqa: F401
                   AdaGrad, Context, CSRDataset, StreamCSRDataset, NimRand, randomNormal, randomize, FactorizationMachine, FieldAwareFactorizationMachine, SGD,  # noqa: F401
                   accuracy, convertSVMLightFile, default_context, expit, load, loadFFMFile, loadSVMLightFile, newAdaGrad, newCSRDataset, newCSRFieldDataset,
                   newFactorizationMachine, newFieldAwareFactorizationMachine, newSGD, newStreamCSRDataset, parseText, rmse, suggestTouchCap,
                   set_default_context)
