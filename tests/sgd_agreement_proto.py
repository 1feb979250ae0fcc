"""(Not a test: a research script on the oracle.)  Can the SGD mini-batch rule keep one sequential epoch's progress per epoch at
batches where every coordinate is touched 40-80 times?  The touch cap (up to 16 steps summed, then averaged) against an agreement-
weighted divisor (oracle/nimfm_mb.c: orc_mb_sgd_agree): held-out loss after 1 / 3 epochs on a planted FM.
usage: python tests/sgd_agreement_proto.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402

n, nt, d, m, k = 200_000, 30_000, 20_000, 16, 8
rng = np.random.default_rng(3)
idx = np.sort(rng.integers(0, d, size=(n + nt, m)), axis=1)
for _ in range(50):
    dup = np.zeros_like(idx, dtype=bool)
    dup[:, 1:] = idx[:, 1:] == idx[:, :-1]
    if not dup.any():
        break
    idx[dup] = rng.integers(0, d, size=int(dup.sum()))
    idx.sort(axis=1)
val = rng.uniform(-1, 1, size=(n + nt, m))
Xall = O.Dataset(np.arange(n + nt + 1, dtype=np.int64) * m, idx.ravel(), val.ravel(), n + nt, d)
Pp, wp = rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.3
yall = np.sign(O.fm_decision_function(Xall, 2, Pp, wp, 0.0) + 0.3 * rng.standard_normal(n + nt))
Xtr = O.Dataset(np.arange(n + 1, dtype=np.int64) * m, idx[:n].ravel(), val[:n].ravel(), n, d)
Xte = O.Dataset(np.arange(nt + 1, dtype=np.int64) * m, idx[n:].ravel(), val[n:].ravel(), nt, d)
ytr, yte = yall[:n], yall[n:]
cfg = O.sgd_cfg(loss="logistic", eta0=0.02, alpha0=1e-6, alpha=1e-5, beta=1e-5)
P0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01


def held_out(P, w, b):
    z = O.fm_decision_function(Xte, 2, P, w, b) * yte
    return float(np.mean(np.log1p(np.exp(-z))))


def run(B, cap=1.0, agree=0.0, epochs=3):
    var = C.c_double.in_dll(O.lib(), "orc_mb_sgd_agree")
    var.value = agree
    P, w, b, it = P0.copy(), np.zeros(d), 0.0, 1
    out = []
    try:
        for e in range(epochs):
            if B == 1:
                P, w, b, it, *_ = O.fm_sgd_fit(Xtr, ytr, 2, P, w, b, cfg, 1, 0, it=it)
            else:
                b, it, ls, vs = O.fm_sgd_epoch_mb(Xtr, ytr, 2, P, w, b, cfg, B, it=it, touch_cap=cap)
            out.append(round(held_out(P, w, b), 4))
    finally:
        var.value = 0.0
    return out


print("sequential order:", run(1), flush=True)
for B in (12_500, 50_000, 100_000):
    lam = B * m / d
    print("batch %d (every coordinate touched ~%.0f times):" % (B, lam), "cap 16", run(B, cap=16.0), "| cap 64", run(B, cap=64.0),
          "| agreement p=1", run(B, agree=1.0), "| p=0.5", run(B, agree=0.5), flush=True)
