"""Oracle pin 4: field-aware FM, fast restatement vs the brute-force FFMSlow.

tests/test_sgd_ffm.nim:87-115 and tests/test_adagrad_ffm.nim:88-116: n=80, d=20,
nFields=5, k=4, shuffle=false, maxIter=5, rtol 1e-6 / atol 1e-9; warm start :58-86.
"""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_ffm, make_ffm_dataset

N, D, F, K = 80, 20, 5, 4


def test_decision_function_vs_bruteforce():
    X, Xd, field_of, _ = make_ffm_dataset(N, D, F, K, 42, threshold=0.3)
    rng = np.random.default_rng(2)
    P, w = rng.standard_normal((F, D, K)) * 0.3, rng.standard_normal(D)
    assert_close(O.ffm_decision_function(X, P, w, -0.5), O.slow_ffm_decision_function(Xd, field_of, F, P, w, -0.5),
                 rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("fit_linear,fit_intercept", itertools.product([False, True], [False, True]))
def test_sgd_fast_vs_naive(fit_linear, fit_intercept):
    X, Xd, field_of, y = make_ffm_dataset(N, D, F, K, 42, threshold=0.3)
    P0, w0, b0 = init_ffm(D, F, K)
    cfg = O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    Ps, ws, bs, _ = O.slow_ffm_sgd_fit(Xd, field_of, F, y, P0, w0, b0, cfg, 5)
    Pf, wf, bf, it, *_ = O.ffm_sgd_fit(X, y, P0, w0, b0, cfg, 5)
    assert it == 5 * N + 1
    assert abs(bf - bs) < 1e-7
    assert_close(wf, ws, what="w")
    assert_close(Pf, Ps, what="P")


@pytest.mark.parametrize("fit_linear,fit_intercept", itertools.product([False, True], [False, True]))
def test_adagrad_fast_vs_naive(fit_linear, fit_intercept):
    X, Xd, field_of, y = make_ffm_dataset(N, D, F, K, 42, threshold=0.3)
    P0, w0, b0 = init_ffm(D, F, K)
    cfg = O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    Ps, ws, bs, _ = O.slow_ffm_adagrad_fit(Xd, field_of, F, y, P0, w0, b0, cfg, 5)
    Pf, wf, bf, *_ = O.ffm_adagrad_fit(X, y, P0, w0, b0, cfg, 5)
    assert abs(bf - bs) < 1e-7
    assert_close(wf, ws, what="w")
    assert_close(Pf, Ps, what="P")


def test_warm_start():
    X, Xd, field_of, y = make_ffm_dataset(N, D, F, K, 42)
    P0, w0, b0 = init_ffm(D, F, K)
    cfg = O.sgd_cfg()
    P, w, b, it = P0, w0, b0, 1
    for _ in range(10):
        P, w, b, it, *_ = O.ffm_sgd_fit(X, y, P, w, b, cfg, 1, it=it)
    P1, w1, b1, *_ = O.ffm_sgd_fit(X, y, P0, w0, b0, cfg, 10)
    assert abs(b - b1) < 1e-8
    assert_close(w, w1, atol=1e-8)
    assert_close(P, P1, atol=1e-8)
    acfg = O.adagrad_cfg()
    P, w, b, it, st = P0, w0, b0, 1, None
    for _ in range(10):
        P, w, b, it, _, _, _, st = O.ffm_adagrad_fit(X, y, P, w, b, acfg, 1, it=it, state=st)
    P1, w1, b1, *_ = O.ffm_adagrad_fit(X, y, P0, w0, b0, acfg, 10)
    assert abs(b - b1) < 1e-8
    assert_close(w, w1, atol=1e-8)
    assert_close(P, P1, atol=1e-8)
