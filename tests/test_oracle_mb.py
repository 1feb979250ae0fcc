"""The mini-batch rule (oracle/nimfm_mb.c) reduces to the
reference's sequential step at batch == 1, and keeps its invariants for batch > 1."""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_ffm, init_fm, make_ffm_dataset, make_fm_dataset, make_perms

N, D, K = 80, 8, 4


@pytest.mark.parametrize("degree,fit_lower,loss", itertools.product([2, 3, 4], ["explicit", "none", "augment"],
                                                                    ["squared", "logistic"]))
def test_sgd_batch1_is_sequential(degree, fit_lower, loss):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
    if loss == "logistic":
        y = np.sign(y)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    cfg = O.sgd_cfg(loss=loss)
    perms = make_perms(N, 3)
    Pf, wf, bf, it_f, el, ev, _ = O.fm_sgd_fit(X, y, degree, P0, w0, b0, cfg, 3, n_aug, perms=perms)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    for e in range(3):
        b, it, ls, vs = O.fm_sgd_epoch_mb(X, y, degree, P, w, b, cfg, 1, n_aug, perm=perms[e], it=it)
        assert abs(ls / N - el[e]) < 1e-12 * max(1.0, abs(el[e]))
        assert abs(vs - ev[e]) < 1e-9 * max(1.0, abs(ev[e]))
    assert it == it_f
    assert abs(b - bf) < 1e-12
    assert_close(w, wf, rtol=1e-10, atol=1e-13)
    assert_close(P, Pf, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("degree,fit_lower", itertools.product([2, 3], ["explicit", "none", "augment"]))
def test_adagrad_batch1_is_sequential(degree, fit_lower):
    X, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    cfg = O.adagrad_cfg()
    perms = make_perms(N, 3)
    Pf, wf, bf, it_f, el, ev, _, _ = O.fm_adagrad_fit(X, y, degree, P0, w0, b0, cfg, 3, n_aug, perms=perms)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(P.shape[0], P.shape[2], K, D)
    for e in range(3):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(X, y, degree, P, w, b, cfg, 1, st, n_aug, perm=perms[e], it=it)
        assert abs(ls / N - el[e]) < 1e-12 * max(1.0, abs(el[e]))
        assert abs(vs - ev[e]) < 1e-9 * max(1.0, abs(ev[e]))
    b = O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st, n_aug)
    assert it == it_f and abs(b - bf) < 1e-12
    assert_close(w, wf, rtol=1e-10, atol=1e-13)
    assert_close(P, Pf, rtol=1e-10, atol=1e-13)


def test_ffm_batch1_is_sequential():
    n, d, F, k = 80, 20, 5, 4
    X, Xd, field_of, y = make_ffm_dataset(n, d, F, k, 42, threshold=0.3)
    P0, w0, b0 = init_ffm(d, F, k)
    cfg = O.sgd_cfg()
    Pf, wf, bf, it_f, el, ev, _ = O.ffm_sgd_fit(X, y, P0, w0, b0, cfg, 2)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    for e in range(2):
        b, it, ls, vs = O.ffm_sgd_epoch_mb(X, y, P, w, b, cfg, 1, it=it)
        assert abs(vs - ev[e]) < 1e-9 * max(1.0, abs(ev[e]))
    assert abs(b - bf) < 1e-12
    assert_close(w, wf, rtol=1e-10, atol=1e-13)
    assert_close(P, Pf, rtol=1e-10, atol=1e-13)
    acfg = O.adagrad_cfg()
    Pf, wf, bf, it_f, el, ev, _, _ = O.ffm_adagrad_fit(X, y, P0, w0, b0, acfg, 2)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(F, d, k, d)
    for e in range(2):
        b, it, ls, vs = O.ffm_adagrad_epoch_mb(X, y, P, w, b, acfg, 1, st, it=it)
        assert abs(vs - ev[e]) < 1e-9 * max(1.0, abs(ev[e]))
    b = O.ffm_adagrad_finalize(P, w, b, acfg, it, st)
    assert abs(b - bf) < 1e-12
    assert_close(w, wf, rtol=1e-10, atol=1e-13)
    assert_close(P, Pf, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("batch", [4, 16, 80, 1000])
def test_minibatch_learns_and_keeps_flags(batch):
    X, Xd, y = make_fm_dataset(N, D, 2, K, 42)
    P0, w0, b0, n_aug = init_fm(D, 2, K, "explicit", True)
    before = np.sqrt(np.mean((O.fm_decision_function(X, 2, P0, w0, b0) - y) ** 2))
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    cfg = O.sgd_cfg(alpha0=1e-9, alpha=1e-9, beta=1e-9)
    for e in range(30):
        b, it, ls, vs = O.fm_sgd_epoch_mb(X, y, 2, P, w, b, cfg, batch, it=it)
    assert it == 30 * N + 1
    assert np.sqrt(np.mean((O.fm_decision_function(X, 2, P, w, b) - y) ** 2)) < before
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    cfg = O.sgd_cfg(fit_linear=False, fit_intercept=False)
    b, it, _, _ = O.fm_sgd_epoch_mb(X, y, 2, P, w, b, cfg, batch, it=it)
    assert (w == 0).all() and b == 0.0


def test_minibatch_subrange_composition():
    """[0,n) in one call == [0,48) then [48,n) when 48 is a batch boundary (nCalls callbacks)."""
    X, Xd, y = make_fm_dataset(N, D, 3, K, 42, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, 3, K, "explicit", True)
    cfg = O.sgd_cfg()
    Pa, wa = P0.copy(), w0.copy()
    ba, ita, la, va = O.fm_sgd_epoch_mb(X, y, 3, Pa, wa, b0, cfg, 16, it=1)
    Pb, wb = P0.copy(), w0.copy()
    bb, itb, l1, v1 = O.fm_sgd_epoch_mb(X, y, 3, Pb, wb, b0, cfg, 16, begin=0, end=48, it=1)
    bb, itb, l2, v2 = O.fm_sgd_epoch_mb(X, y, 3, Pb, wb, bb, cfg, 16, begin=48, end=N, it=itb)
    assert ita == itb and ba == bb and (Pa == Pb).all() and (wa == wb).all()
    assert abs(la - (l1 + l2)) < 1e-12 and abs(va - (v1 + v2)) < 1e-12
