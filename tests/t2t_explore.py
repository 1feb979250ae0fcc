"""CPU exploration for bench.py's time_to_target leg: which planted model / step size lets ONE to TEN epochs in the
reference's order (oracle: optimizer/sgd.nim:294-321) close a real share of the gap between the start and the planted
model's own held-out loss, at the bench shapes' samples-per-feature ratio (d scaled down, n / d and nnz per row kept).
usage: python tests/t2t_explore.py cfg2|headline [eta0 ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
etas = [float(v) for v in sys.argv[2:]] or [0.01, 0.05, 0.2]
SH = {"cfg2": dict(d=10_000, m=32, k=16, n_t=100_000, n_h=20_000),
      "headline": dict(d=20_000, m=64, k=64, n_t=20_000, n_h=10_000),
      "headline4": dict(d=20_000, m=64, k=64, n_t=80_000, n_h=10_000)}[name]
d, m, k, n_t, n_h = SH["d"], SH["m"], SH["k"], SH["n_t"], SH["n_h"]
rng = np.random.default_rng(42)
n = n_t + n_h
idx = np.sort(rng.integers(0, d, size=(n, m)), axis=1)
ok = (np.diff(idx, axis=1) > 0).all(axis=1)
idx = idx[ok][: n]
n = len(idx)
n_t = n - n_h
val = rng.uniform(-1, 1, (n, m))


def ds(lo, hi):
    return O.Dataset(np.arange(hi - lo + 1, dtype=np.int64) * m, idx[lo:hi].ravel().astype(np.int64), val[lo:hi].ravel().copy(), hi - lo, d)


Xt, Xh = ds(0, n_t), ds(n_t, n)
Xall = ds(0, n)


def logloss(p, y):
    z = p * y
    return float(np.mean(np.log1p(np.exp(-np.abs(z))) - np.minimum(z, 0)))


for (sw, sp, kp) in [(0.1, 0.1, k), (0.3, 0.1, k), (0.5, 0.3, 4), (0.3, 0.3, 4)]:
    prng = np.random.default_rng(1234)
    Pp = prng.standard_normal((1, kp, d)) * sp
    wp = prng.standard_normal(d) * sw
    f = O.fm_decision_function(Xall, 2, Pp, wp, 0.0)
    lin = O.fm_decision_function(Xall, 2, Pp * 0, wp, 0.0)
    y = np.sign(f)
    l_pl = logloss(f[n_t:], y[n_t:])
    print("planted w %.2f P %.2f (k=%d): std f %.3f, linear %.3f; planted model's own loss %.4f" % (sw, sp, kp, f.std(), lin.std(), l_pl), flush=True)
    for eta0 in etas:
        for reg in (1e-3, 1e-5):
            P = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01
            w, b, it = np.zeros(d), 0.0, 1
            cfg = O.sgd_cfg(eta0=eta0, alpha=reg, beta=reg, loss="logistic")
            l0 = logloss(O.fm_decision_function(Xh, 2, P, w, b), y[n_t:])
            out = []
            t0 = time.time()
            for e in range(1, 11):
                P, w, b, it, *_ = O.fm_sgd_fit(Xt, y[:n_t], 2, P, w, b, cfg, 1, 0, it=it)
                if e in (1, 3, 10):
                    l = logloss(O.fm_decision_function(Xh, 2, P, w, b), y[n_t:])
                    out.append("%d: %.4f (%.0f%%)" % (e, l, 100 * (l0 - l) / (l0 - l_pl)))
            print("   eta0 %.3g reg %.0e: start %.4f -> %s   [%.1f s]" % (eta0, reg, l0, ", ".join(out), time.time() - t0), flush=True)
