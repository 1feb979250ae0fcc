"""the oracle's restatement of mini-batch proximal SGD
(oracle/nimfm_psgd.c <- optimizer/minibatch_psgd.nim:67-122, model/params.nim:33-98, regularizer/*.nim).

The reference's only test for this row is tests/test_squaredl12.nim:10-27 (proxSquaredL12 against the sort-based
proxSquaredL12Slow); its grid is re-run here on the two restatements.  The solver is checked against a dense numpy
statement of the same update built from the brute-force model (tests/model/fm_slow.nim)."""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_fm, make_fm_dataset

LAMS = [0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 3, 4]  # test_squaredl12.nim:19


def test_prox_squaredl12_matches_slow():
    """test_squaredl12.nim:10-27: d = 100, q_j = rand(400)/100 - 2, 13 lambdas, |fast - slow| < 1e-10"""
    rng = np.random.default_rng(0)
    for it in range(200):
        q = rng.integers(0, 401, size=100) / 100.0 - 2.0
        for lam in LAMS:
            p1 = O.prox_squaredl12(q, lam, seed=it + 1)
            p2 = O.prox_squaredl12_slow(q, lam)
            assert np.max(np.abs(p1 - p2)) < 1e-10


def test_prox_squaredl12_pivot_independent():
    rng = np.random.default_rng(1)
    q = rng.normal(size=257)
    a = O.prox_squaredl12(q, 0.05, seed=3)
    b = O.prox_squaredl12(q, 0.05, seed=12345)
    assert np.max(np.abs(a - b)) < 1e-14
    # fixed point of the operator's defining equation: tau = 2 lam sum max(|q| - tau, 0)
    nz = np.abs(a) > 0
    tau = np.abs(q[nz])[0] - np.abs(a[nz])[0]
    assert abs(tau - 2 * 0.05 * np.sum(np.maximum(np.abs(q) - tau, 0.0))) < 1e-12


def test_prox_edge_cases():
    assert np.all(O.prox_squaredl12(np.zeros(7), 0.1) == 0.0)
    assert len(O.prox_squaredl12(np.zeros(0), 0.1)) == 0
    one = O.prox_squaredl12(np.array([2.0]), 0.5)  # S = 2 / (1 + 2 lam) = 1, threshold 2 lam S = 1
    assert abs(one[0] - 1.0) < 1e-15
    assert np.all(O.prox("l1", np.array([[0.5, -0.2], [-3.0, 0.0]]), 0.3) == np.array([[0.2, 0.0], [-2.7, 0.0]]))


def test_matrix_prox_closed_forms():
    rng = np.random.default_rng(2)
    Pt = rng.normal(size=(37, 5))
    lam = 0.4
    assert_close(O.prox("l1", Pt, lam), np.sign(Pt) * np.maximum(np.abs(Pt) - lam, 0), rtol=0, atol=1e-15)
    nr = np.sqrt((Pt ** 2).sum(1, keepdims=True))
    assert_close(O.prox("l21", Pt, lam), np.where(nr > lam, Pt * (1 - lam / np.maximum(nr, 1e-300)), 0.0), rtol=1e-14, atol=1e-15)
    # squaredl12: column-wise (transpose = true, the default) and row-wise
    got = O.prox("squaredl12", Pt, lam)
    for s in range(Pt.shape[1]):
        assert_close(got[:, s], O.prox_squaredl12_slow(Pt[:, s], lam), rtol=0, atol=1e-13)
    got = O.prox("squaredl12", Pt, lam, transpose=False)
    for j in range(Pt.shape[0]):
        assert_close(got[j], O.prox_squaredl12_slow(Pt[j], lam), rtol=0, atol=1e-13)
    # squaredl21: the vector operator on the row norms, rows rescaled
    got = O.prox("squaredl21", Pt, lam)
    nn = O.prox_squaredl12_slow(nr[:, 0], lam)
    assert_close(got, Pt / nr * nn[:, None], rtol=1e-13, atol=1e-15)
    for reg in O.REG:  # eval: the verbose line's regulariser value
        v = O.reg_eval(reg, Pt)
        want = {"l1": np.abs(Pt).sum(), "l21": nr.sum(), "squaredl12": (np.abs(Pt).sum(0) ** 2).sum(),
                "squaredl21": nr.sum() ** 2}[reg]
        assert abs(v - want) < 1e-10 * want


def dense_step(Xd, y, degree, P, w, b, cfg, idx, batch, it, n_aug, fit_lower_orders):
    """the same mini-batch in dense numpy: gradient of the brute-force model by central differences"""
    O_, k, da = P.shape
    d = Xd.shape[1]

    def f(Pm, wm, bm, i):
        return O.slow_fm_decision_function(Xd[i:i + 1], degree, Pm, wm, bm, n_aug)[0]

    gP, gw, gb = np.zeros_like(P), np.zeros_like(w), 0.0
    loss_sum = 0.0
    h = 1e-6
    for i in idx:
        yp = f(P, w, b, i)
        loss_sum += O.lib().orc_loss(cfg.loss, cfg.loss_param, y[i], yp)
        coef = O.lib().orc_dloss(cfg.loss, cfg.loss_param, y[i], yp) / batch
        row = np.nonzero(Xd[i])[0]
        touched = list(row) + list(range(d, da))
        for o in range(O_):
            for s in range(k):
                for j in touched:
                    Pp, Pm = P.copy(), P.copy()
                    Pp[o, s, j] += h
                    Pm[o, s, j] -= h
                    gP[o, s, j] += coef * (f(Pp, w, b, i) - f(Pm, w, b, i)) / (2 * h)
        if cfg.fit_linear:
            gw[row] += coef * Xd[i, row]
        if cfg.fit_intercept:
            gb += coef
    eta = lambda reg: O.lib().orc_get_eta(cfg.scheduling, cfg.eta0, cfg.power, reg, it)
    eP, ew, e0 = eta(cfg.beta), eta(cfg.alpha), eta(cfg.alpha0)
    P = (P - eP * gP) / (1 + eP * cfg.beta)
    if cfg.fit_linear:
        w = (w - ew * gw) / (1 + ew * cfg.alpha)
    if cfg.fit_intercept:
        b = (b - (e0 * gb if cfg.fit_linear else 0.0)) / (1 + e0 * cfg.alpha0)
    lam = cfg.gamma * eP / (1 + eP * cfg.beta)
    names = {v: n for n, v in O.REG.items()}
    for o in range(O_):
        P[o] = O.prox(names[cfg.reg], P[o].T.copy(), lam, transpose=bool(cfg.reg_transpose)).T
    return P, w, b, loss_sum


@pytest.mark.parametrize("degree,fit_lower,reg,loss", [
    (2, "explicit", "squaredl12", "squared"), (2, "explicit", "l1", "logistic"), (2, "explicit", "l21", "squared"),
    (2, "explicit", "squaredl21", "squared"), (3, "explicit", "l1", "squared"), (3, "augment", "l21", "squared"),
    (2, "none", "l1", "squared_hinge")])
def test_epoch_matches_dense_statement(degree, fit_lower, reg, loss):
    n, d, k, B = 24, 6, 3, 5
    X, Xd, y = make_fm_dataset(n, d, degree, k, 11, fit_lower, threshold=0.3)
    if loss != "squared":
        y = np.sign(y)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.3)
    b0 = 0.1
    cfg = O.psgd_cfg(eta0=0.2, gamma=0.05, beta=1e-2, alpha=1e-2, alpha0=1e-2, loss=loss, reg=reg)
    inner = (n - 1) // B + 1
    stream = np.concatenate([np.random.default_rng(5).permutation(n), np.random.default_rng(6).permutation(n)])[:B * inner]
    P, w = P0.copy(), w0.copy()
    b, it, ls = O.fm_mbpsgd_epoch(X, y, degree, P, w, b0, cfg, stream, B, n_aug, it=1)
    assert it == 1 + inner
    Pd, wd, bd, lsd = P0.copy(), w0.copy(), b0, 0.0
    for t in range(inner):
        Pd, wd, bd, l1 = dense_step(Xd, y, degree, Pd, wd, bd, cfg, stream[t * B:(t + 1) * B], B, 1 + t, n_aug, None)
        lsd += l1
    assert abs(ls - lsd) < 1e-6 * max(1.0, abs(lsd))
    assert abs(b - bd) < 1e-7
    assert_close(w, wd, rtol=1e-6, atol=1e-8)
    assert_close(P, Pd, rtol=1e-5, atol=1e-7)


def test_intercept_step_gated_on_fit_linear():
    """model/params.nim:47 as written: without fitLinear the intercept only shrinks"""
    n, d, k, B = 12, 5, 2, 4
    X, Xd, y = make_fm_dataset(n, d, 2, k, 3, "explicit", fit_linear=False, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", False, scale=0.3)
    cfg = O.psgd_cfg(eta0=0.1, reg="l1", scheduling="constant", alpha0=0.5, fit_linear=False)
    P, w = P0.copy(), w0.copy()
    b, it, ls = O.fm_mbpsgd_epoch(X, y, 2, P, w, 2.0, cfg, np.arange(12), B, n_aug, it=1)
    assert abs(b - 2.0 / (1 + 0.1 * 0.5) ** 3) < 1e-14
    assert np.all(w == w0)


@pytest.mark.parametrize("degree,fit_lower,loss", [(2, "explicit", "squared"), (3, "explicit", "logistic"),
                                                   (3, "augment", "squared"), (2, "none", "squared_hinge")])
def test_predict_all_with_grad_is_the_gradient_of_the_mean_loss(degree, fit_lower, loss):
    """pgd.nim:70-103 against central differences of the brute-force model's mean loss"""
    n, d, k = 15, 5, 2
    X, Xd, y = make_fm_dataset(n, d, degree, k, 4, fit_lower, threshold=0.3)
    if loss != "squared":  # (Huber is left out: the reference's dloss for it, loss.nim:90-93, is not the derivative of its loss)
        y = np.sign(y)
    P, w, b, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.4)
    w = np.random.default_rng(2).normal(size=d) * 0.1
    b = 0.3
    yp, dL, gP, gw, gb = O.fm_predict_all_with_grad(X, y, degree, P, w, b, loss, n_aug)
    assert_close(yp, O.slow_fm_decision_function(Xd, degree, P, w, b, n_aug), rtol=1e-10, atol=1e-12)
    lid = O.LOSS[loss]

    def mean_loss(Pm, wm, bm):
        f = O.slow_fm_decision_function(Xd, degree, Pm, wm, bm, n_aug)
        return np.mean([O.lib().orc_loss(lid, 1.0, y[i], f[i]) for i in range(n)])

    h = 1e-6
    tol = 1e-6 if loss in ("squared", "logistic") else 1e-4  # the squared hinge's derivative has a kink: central differences lose an order
    for o, s, j in itertools.product(range(P.shape[0]), range(k), range(P.shape[2])):
        Pp, Pm = P.copy(), P.copy()
        Pp[o, s, j] += h
        Pm[o, s, j] -= h
        assert abs(gP[o, j, s] - (mean_loss(Pp, w, b) - mean_loss(Pm, w, b)) / (2 * h)) < tol
    for j in range(d):
        wp, wm = w.copy(), w.copy()
        wp[j] += h
        wm[j] -= h
        assert abs(gw[j] - (mean_loss(P, wp, b) - mean_loss(P, wm, b)) / (2 * h)) < tol
    assert abs(gb - (mean_loss(P, w, b + h) - mean_loss(P, w, b - h)) / (2 * h)) < tol
    assert abs(gb - dL.mean()) < 1e-14
