"""Oracle pin 2: SGD.fit, fast restatement vs the brute-force SGDSlow.

The reference's own suite (tests/test_sgd.nim) on the two restatements:
  :16-55  fitLinear=false => w == 0, fitIntercept=false => intercept == 0
  :58-89  warmStart: 10 x fit(maxIter=1) == fit(maxIter=10), shuffle off, atol 1e-8
  :92-126 fast == slow, maxIter=5, rtol 1e-6 / atol 1e-9, degree 2..4 x fitLower x
          fitLinear x fitIntercept (threshold 0.3 data)
  :129-151 score (rmse) improves
Sizes n=80, d=8, k=4 (:10-13).
"""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_fm, make_fm_dataset, make_perms

N, D, K = 80, 8, 4
GRID = list(itertools.product([2, 3, 4], ["explicit", "none", "augment"], [False, True], [False, True]))


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept", GRID)
def test_fast_vs_naive(degree, fit_lower, fit_linear, fit_intercept):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    cfg = O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    perms = make_perms(N, 5)
    Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, degree, P0, w0, b0, cfg, 5, n_aug, perms)
    Pf, wf, bf, it, el, ev, nrun = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 5, n_aug, perms=perms)
    assert it == 5 * N + 1 and nrun == 5
    assert abs(bf - bs) < 1e-7
    assert_close(wf, ws, what="w")
    assert_close(Pf, Ps, what="P")


@pytest.mark.parametrize("degree,fit_lower,flag", itertools.product([2, 3, 4], ["explicit", "none", "augment"], [True, False]))
def test_fit_linear_and_intercept_off(degree, fit_lower, flag):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, False, flag)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, False)
    perms = make_perms(N, 10)
    _, w, _, _, _, _, _ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, O.sgd_cfg(fit_linear=False, fit_intercept=flag),
                                       10, n_aug, perms=perms)
    assert (w == 0.0).all()
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, flag, False)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, flag)
    _, _, b, _, _, _, _ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, O.sgd_cfg(fit_linear=flag, fit_intercept=False),
                                       10, n_aug, perms=perms)
    assert b == 0.0


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept", GRID)
def test_warm_start(degree, fit_lower, fit_linear, fit_intercept):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    cfg = O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    P, w, b, it = P0, w0, b0, 1
    for _ in range(10):
        P, w, b, it, _, _, _ = O.fm_sgd_fit(X, y, degree, P, w, b, cfg, 1, n_aug, it=it)
    P1, w1, b1, it1, _, _, _ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 10, n_aug)
    assert it == it1
    assert abs(b - b1) < 1e-8
    assert_close(w, w1, atol=1e-8)
    assert_close(P, P1, atol=1e-8)


@pytest.mark.parametrize("loss", ["squared", "squared_hinge", "logistic", "huber"])
@pytest.mark.parametrize("scheduling", ["constant", "optimal", "invscaling", "pegasos"])
def test_losses_and_schedules_vs_naive(loss, scheduling):
    """Beyond the reference's grid: every loss x schedule through both restatements."""
    degree, fit_lower = 3, "explicit"
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3, scale=0.3)
    if loss in ("squared_hinge", "logistic"):
        y = np.sign(y)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    # pegasos: eta = 1/(reg*it) needs reg ~ 1 to stay finite
    kw = dict(alpha0=0.5, alpha=0.5, beta=0.5) if scheduling == "pegasos" else {}
    cfg = O.sgd_cfg(loss=loss, scheduling=scheduling, power=0.75, **kw)
    perms = make_perms(N, 3)
    # pegasos at it == 1 has 1 - eta*reg == 0: the reference's lazy scaling then computes 0/0
    # (sgd.nim:234-243,125-131), so that schedule is only meaningful from a later `it`.
    it0 = 20 if scheduling == "pegasos" else 1
    Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, degree, P0, w0, b0, cfg, 3, n_aug, perms, it=it0)
    Pf, wf, bf, _, el, _, _ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 3, n_aug, perms=perms, it=it0)
    assert np.isfinite(el).all()
    assert abs(bf - bs) < 1e-7
    assert_close(wf, ws, what="w")
    assert_close(Pf, Ps, what="P")


@pytest.mark.parametrize("degree,fit_lower", itertools.product([2, 3, 4], ["explicit", "none", "augment"]))
def test_score_improves(degree, fit_lower):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    cfg = O.sgd_cfg(alpha0=1e-9, alpha=1e-9, beta=1e-9)
    before = np.sqrt(np.mean((O.fm_decision_function(X, degree, P0, w0, b0, n_aug) - y) ** 2))
    P, w, b, *_ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 20, n_aug, perms=make_perms(N, 20))
    after = np.sqrt(np.mean((O.fm_decision_function(X, degree, P, w, b, n_aug) - y) ** 2))
    assert after < before


def test_decision_function_vs_bruteforce():
    """model/factorization_machine.nim:100-122 vs tests/model/fm_slow.nim:42-73."""
    for degree, fit_lower, fit_linear in itertools.product([2, 3, 4], ["explicit", "none", "augment"], [True, False]):
        X, Xd, _ = make_fm_dataset(N, D, degree, K, 5, fit_lower, fit_linear, threshold=0.3)
        rng = np.random.default_rng(3)
        P0, _, _, n_aug = init_fm(D, degree, K, fit_lower, fit_linear, scale=0.5)
        w = rng.standard_normal(D)
        assert_close(O.fm_decision_function(X, degree, P0, w, 0.25, n_aug),
                     O.slow_fm_decision_function(Xd, degree, P0, w, 0.25, n_aug), rtol=1e-9, atol=1e-12)


def test_stopping_and_scaling_reset():
    """viol < tol stops the loop (sgd.nim:85-89); huge beta drives scaling_P below 1e-9 so
    resetScaling (sgd.nim:116-131) runs and must not change the result vs the dense model."""
    degree, fit_lower = 2, "explicit"
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    *_, nrun = O.fm_sgd_fit(X, y, degree, P0, w0, b0, O.sgd_cfg(), 50, n_aug, tol=1e9)
    assert nrun == 1
    cfg = O.sgd_cfg(eta0=0.5, alpha=1.5, beta=1.5, scheduling="constant")
    Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, degree, P0, w0, b0, cfg, 4, n_aug)
    Pf, wf, bf, *_ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 4, n_aug)
    assert_close(wf, ws, what="w")
    assert_close(Pf, Ps, what="P")


def test_jagged_layout_restatement_is_bit_identical():
    """oracle/nimfm_jagged.c (the reference's seq-of-seq storage, timed by bench.py's cpu_baseline) == the flat restatement"""
    from common import make_perms, random_csr
    n, d, m, k = 300, 40, 6, 5
    X = random_csr(n, d, m, seed=2)
    rng = np.random.default_rng(1)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.01
    perms = make_perms(n, 3)
    for loss, sched, fl, fi in (("squared", "optimal", True, True), ("logistic", "invscaling", False, True), ("huber", "constant", True, False)):
        cfg = O.sgd_cfg(loss=loss, scheduling=sched, fit_linear=fl, fit_intercept=fi, eta0=0.05)
        P, w, b, it, el, ev, _ = O.fm_sgd_fit(X, y, 2, P0, w0, 0.1, cfg, 3, perms=perms)
        Pj, wj, bj, itj, elj, evj = O.fm_sgd_fit_jagged(X, y, P0, w0, 0.1, cfg, 3, perms=perms)
        assert np.array_equal(P, Pj) and np.array_equal(w, wj) and b == bj and it == itj
        assert np.array_equal(el, elj) and np.array_equal(ev, evj)
