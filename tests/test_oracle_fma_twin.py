"""The conditioning yardstick of tests/fuzz_mb.py (FUZZ_CPU_TWIN=1): the oracle's sources built with fused multiply-adds
(oracle/Makefile target `fma`; never a parity build) against the parity build.  On a well-conditioned draw the two agree to
rounding; they need not be bit-equal -- that they may differ is the point of the yardstick."""
import numpy as np

import oracle as O
from common import random_csr


def _fit(Xo, y, P0, w0, solver):
    P, w, b, it = P0.copy(), w0.copy(), 0.1, 1
    if solver == "sgd":
        cfg = O.sgd_cfg(eta0=0.01, loss="squared")
        for _ in range(2):
            b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, cfg, 64, it=it, touch_cap=16.0)
    else:
        cfg = O.adagrad_cfg(loss="squared")
        st = O.AdaState(1, P0.shape[2], P0.shape[1], P0.shape[2])
        for _ in range(2):
            b, it, _, _ = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, 64, st, it=it, ada_cross=0.1)
        b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    return P, w, b


def test_fma_build_agrees_with_the_parity_build_on_a_well_conditioned_draw():
    n, d, m, k = 1500, 300, 8, 8
    Xo = random_csr(n, d, m, seed=5)
    rng = np.random.default_rng(6)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    for solver in ("sgd", "adagrad"):
        P, w, b = _fit(Xo, y, P0, w0, solver)
        with O.variant("fma"):
            P2, w2, b2 = _fit(Xo, y, P0, w0, solver)
        P3, w3, b3 = _fit(Xo, y, P0, w0, solver)  # the parity build is back after the block, and deterministic
        assert np.array_equal(P, P3) and np.array_equal(w, w3) and b == b3
        scale = np.abs(P).max()
        assert np.abs(P2 - P).max() <= 1e-9 * scale and np.abs(w2 - w).max() <= 1e-9 * max(np.abs(w).max(), 1e-3) and abs(b2 - b) <= 1e-9
