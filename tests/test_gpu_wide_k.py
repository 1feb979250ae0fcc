"""-m gpu: FactorizationMachines with more than 128 factors (nComponents has no cap in the reference,
model/factorization_machine.nim:68-70).  On the device the factors of one order are cut into blocks of at most 128
(ModelView::kc, csrc/common.h): an ANOVA kernel is a sum over the factors of terms that do not mix them
(kernels.nim:46-64), so every kernel that walks "orders" takes the blocks as they come.  Held against the oracle on
every route a model takes: decisionFunction, the parameter / AdaGrad state layouts at the C ABI, NFM_MODE_SEQUENTIAL
(the one-sample-in-flight kernel; sums over the factors continue across the blocks of an order, sgd.nim:172-173)
and NFM_MODE_MINIBATCH (SGD with touch cap 1 and 16, AdaGrad)."""
import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_fm, make_perms, random_csr
from gpu_common import gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [129, 200, 256, 300])
@pytest.mark.parametrize("degree,fit_lower", [(2, "explicit"), (3, "explicit"), (3, "augment"), (4, "none")])
def test_decision_function(k, degree, fit_lower):
    n, d = 300, 60
    Xo = ragged_csr(n, d, seed=k + degree, max_m=40)
    rng = np.random.default_rng(k)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.3)
    w0 = rng.standard_normal(d) * 0.1
    fm = gpu_fm("regression", degree, k, fit_lower, True, True, P0, w0, 0.25)
    want = O.fm_decision_function(Xo, degree, P0, w0, 0.25, n_aug)
    assert_close(fm.decisionFunction(to_gpu(Xo)), want, 1e-10, 1e-12, "decisionFunction")
    # the reference layout survives the block layout bit for bit (nfm_model_set_params -> nfm_model_get_params)
    fm._pull()
    assert np.array_equal(np.asarray(fm.P), P0) and np.array_equal(np.asarray(fm.w), w0) and fm.intercept == 0.25


def test_lams_follow_their_factors():
    """decisionFunction weights factor s by lams[s] (model/factorization_machine.nim:120)"""
    n, d, k = 200, 40, 200
    Xo = random_csr(n, d, 8, seed=3)
    rng = np.random.default_rng(4)
    P0 = rng.standard_normal((1, k, d)) * 0.2
    lams = rng.uniform(0.5, 1.5, size=k)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, np.zeros(d), 0.0)
    fm.lams = lams
    fm.set_params(P0, np.zeros(d), 0.0)
    X = to_gpu(Xo)
    got = fm.decisionFunction(X)
    want = np.zeros(n)
    for s in range(k):
        Ps = np.zeros((1, 1, d))
        Ps[0, 0] = P0[0, s]
        want += lams[s] * O.fm_decision_function(Xo, 2, Ps, np.zeros(d), 0.0, 0)
    assert_close(got, want, 1e-10, 1e-12, "decisionFunction with lams")


@pytest.mark.parametrize("k", [131, 200])
@pytest.mark.parametrize("degree,fit_lower", [(2, "explicit"), (3, "explicit"), (3, "augment")])
def test_sequential_sgd_and_adagrad(k, degree, fit_lower):
    n, d = 120, 30
    Xo = ragged_csr(n, d, seed=11, max_m=12)
    rng = np.random.default_rng(12)
    y = rng.standard_normal(n)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.05)
    perms = make_perms(n, 3)
    X = to_gpu(Xo)
    Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(), 3, n_aug, perms=perms)
    fm = gpu_fm("regression", degree, k, fit_lower, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=3, verbose=0, tol=0)
    sgd.fit(X, y, fm, perms=perms)
    assert sgd.it == it and abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, 1e-8, 1e-11, "w")
    assert_close(fm.P, Pf, 1e-8, 1e-11, "P")
    assert_close([h[1] for h in sgd.history], el, 1e-9, 1e-12, "loss")
    assert_close([h[0] for h in sgd.history], ev, 1e-8, 1e-11, "viol")
    cfg = O.adagrad_cfg()
    Pf, wf, bf, it, el, ev, _, st = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, cfg, 3, n_aug, perms=perms)
    fm = gpu_fm("regression", degree, k, fit_lower, True, True, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=3, verbose=0, tol=0)
    ada.fit(X, y, fm, perms=perms)
    assert ada.it == it and abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, 1e-8, 1e-11, "w")
    assert_close(fm.P, Pf, 1e-8, 1e-11, "P")
    assert_close([h[0] for h in ada.history], ev, 1e-8, 1e-11, "viol")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(fm)  # the state tensors come back in the reference's layout
    assert_close(gs, st.gsum_P, 1e-8, 1e-11, "g_sum.P")
    assert_close(gn, st.gnorm_P, 1e-8, 1e-11, "g_norm.P")


@pytest.mark.parametrize("k,cap", [(131, 1.0), (200, 16.0), (300, 16.0)])
def test_minibatch_sgd(k, cap):
    n, d, m, B = 3000, 400, 12, 256
    Xo = random_csr(n, d, m, seed=21)
    rng = np.random.default_rng(22)
    y = np.sign(rng.standard_normal(n))
    P0, w0 = rng.standard_normal((1, k, d)) * 0.03, np.zeros(d)
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    hist = []
    perm = np.random.default_rng(23).permutation(n).astype(np.int64)
    for e in range(2):
        b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(loss="logistic"), B, it=it, touch_cap=cap,
                                          perm=perm if e else None)
        hist.append((ls, vs))
    X = to_gpu(Xo)
    X.set_targets(y)
    fm = gpu_fm("classification", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=B, touchCap=cap)
    sgd._handle(fm, X.ctx, "minibatch")
    got = []
    for e in range(2):
        got.append(sgd._epoch(X, perm if e else None, 0, n))
        sgd.it += n
    sgd._finalize_into(fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close(got, hist, 1e-9, 0, "loss / viol sums")


@pytest.mark.parametrize("k,degree", [(150, 2), (136, 3)])
def test_minibatch_adagrad(k, degree):
    n, d, m, B = 2500, 300, 10, 200
    Xo = random_csr(n, d, m, seed=31)
    rng = np.random.default_rng(32)
    y = rng.standard_normal(n)
    no = degree - 1
    P0, w0 = rng.standard_normal((no, k, d)) * 0.05, np.zeros(d)
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(no, d, k, d)
    hv = []
    for e in range(2):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, B, st, it=it)
        hv.append(vs)
    b = O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st)
    fm = gpu_fm("regression", degree, k, "explicit", True, True, P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    ada.fit(to_gpu(Xo), y, fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close([h[0] for h in ada.history], hv, 1e-9, 1e-12, "viol")
    # warm start through the C ABI's state layout: get -> a fresh optimizer -> set -> the next epoch agrees
    state = ada.get_state(fm)
    b2, it2, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, B, st, it=it)
    b2 = O.fm_adagrad_finalize(degree, P, w, b2, cfg, it2, st)
    from nimfm_amd import _capi as capi
    from nimfm_amd.host import _vp
    X = to_gpu(Xo)
    X.set_targets(y)
    ada2 = nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    ada2._handle(fm, X.ctx, "minibatch")
    ada2.it = ada.it
    capi.check(capi.lib().nfm_opt_set_it(ada2._h, ada.it))
    gs, gn, gsw, gnw, gsb, gnb = state
    capi.check(capi.lib().nfm_opt_set_state(ada2._h, _vp(gs), _vp(gn), _vp(gsw), _vp(gnw), gsb, gnb))
    ada2._epoch(X, None, 0, n)
    ada2.it += n
    ada2._finalize_into(fm)
    assert abs(fm.intercept - b2) < 1e-11
    assert_close(fm.P, P, 1e-9, 1e-13, "P after a warm-started epoch")


def test_mbpsgd_refuses_wide_models():
    """the matrix prox of MBPSGD's regularisers needs all factors of a feature in one row"""
    d, k = 50, 200
    rng = np.random.default_rng(1)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, rng.standard_normal((1, k, d)) * 0.01, np.zeros(d), 0.0)
    Xo = random_csr(100, d, 5, seed=2)
    opt = nf.newMBPSGD(maxIter=1, verbose=0)
    with pytest.raises(Exception):
        opt.fit(to_gpu(Xo), rng.standard_normal(100), fm)


@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_minibatch_sparse_regime(solver):
    """few touches per feature and batch (lambda = B m / d = 0.1): the regime in which narrow degree-2 models update
    once-touched features in the row phase; wide models have several blocks and take the column phase for everything"""
    n, d, m, B, k = 2000, 20000, 8, 256, 160
    Xo = random_csr(n, d, m, seed=41)
    rng = np.random.default_rng(42)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    if solver == "sgd":
        for e in range(2):
            b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(), B, it=it)
        opt = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    else:
        cfg = O.adagrad_cfg()
        st = O.AdaState(1, d, k, d)
        for e in range(2):
            b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, B, st, it=it)
        b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
        opt = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    opt.fit(to_gpu(Xo), y, fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")


def test_minibatch_of_one_is_the_sequential_rule():
    """batch = 1: the mini-batch rule IS the reference's per-sample step"""
    n, d, m, k = 150, 60, 6, 140
    Xo = random_csr(n, d, m, seed=51)
    rng = np.random.default_rng(52)
    y = rng.standard_normal(n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.05)
    Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(), 1, 0, perms=None)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=1)
    sgd.fit(to_gpu(Xo), y, fm)
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, 1e-8, 1e-11, "w")
    assert_close(fm.P, Pf, 1e-8, 1e-11, "P")
