"""Oracle pin 3: AdaGrad.fit, fast restatement vs the brute-force AdaGradSlow.

The reference's suite (tests/test_adagrad.nim): warmStart equivalence (:58-89, atol
1e-8), fast == slow for degree 2..3 x fitLower with fitLinear = fitIntercept = false
(:92-126, rtol 1e-6); here the linear/intercept combinations are run as well.
"""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_fm, make_fm_dataset, make_perms

N, D, K = 80, 8, 4


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept",
                         itertools.product([2, 3], ["explicit", "none", "augment"], [False, True], [False, True]))
def test_fast_vs_naive(degree, fit_lower, fit_linear, fit_intercept):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    cfg = O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    perms = make_perms(N, 5)
    Ps, ws, bs, _ = O.slow_fm_adagrad_fit(Xd, y, degree, P0, w0, b0, cfg, 5, n_aug, perms)
    Pf, wf, bf, it, el, ev, nrun, _ = O.fm_adagrad_fit(X, y, degree, P0, w0, b0, cfg, 5, n_aug, perms=perms)
    assert it == 5 * N + 1
    assert abs(bf - bs) < 1e-6
    assert_close(wf, ws, rtol=1e-6, what="w")
    assert_close(Pf, Ps, rtol=1e-6, what="P")


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept",
                         itertools.product([2, 3, 4], ["explicit", "augment", "none"], [False, True], [False, True]))
def test_warm_start(degree, fit_lower, fit_linear, fit_intercept):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    cfg = O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    P, w, b, it, st = P0, w0, b0, 1, None
    for _ in range(10):
        P, w, b, it, _, _, _, st = O.fm_adagrad_fit(X, y, degree, P, w, b, cfg, 1, n_aug, it=it, state=st)
    P1, w1, b1, it1, *_ = O.fm_adagrad_fit(X, y, degree, P0, w0, b0, cfg, 10, n_aug)
    assert it == it1
    assert abs(b - b1) < 1e-8
    assert_close(w, w1, atol=1e-8)
    assert_close(P, P1, atol=1e-8)


def test_fit_flags_off():
    X, Xd, y = make_fm_dataset(N, D, 2, K, 42)
    P0, w0, b0, n_aug = init_fm(D, 2, K, "explicit", False)
    _, w, b, *_ = O.fm_adagrad_fit(X, y, 2, P0, w0, b0, O.adagrad_cfg(fit_linear=False, fit_intercept=False), 5)
    assert (w == 0.0).all() and b == 0.0


def test_score_improves():
    for degree, fit_lower in itertools.product([2, 3], ["explicit", "none", "augment"]):
        X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower)
        P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
        cfg = O.adagrad_cfg(alpha0=1e-9, alpha=1e-9, beta=1e-9)
        before = np.sqrt(np.mean((O.fm_decision_function(X, degree, P0, w0, b0, n_aug) - y) ** 2))
        P, w, b, *_ = O.fm_adagrad_fit(X, y, degree, P0, w0, b0, cfg, 20, n_aug, perms=make_perms(N, 20))
        after = np.sqrt(np.mean((O.fm_decision_function(X, degree, P, w, b, n_aug) - y) ** 2))
        assert after < before
