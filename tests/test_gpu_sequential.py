"""-m gpu: NFM_MODE_SEQUENTIAL vs the reference-faithful CPU oracle, on the reference's own grids.

tests/test_sgd.nim:92-126, test_adagrad.nim:92-126, test_sgd_ffm.nim:87-115, test_adagrad_ffm.nim:88-116
compare fast and slow at rtol 1e-6 / atol 1e-9; the GPU is held to rtol 1e-8 / atol 1e-11 against the
oracle's fast path (differences: one global L2 scale instead of per-feature snapshots, libm vs ocml)."""
import itertools

import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_ffm, init_fm, make_ffm_dataset, make_fm_dataset, make_perms
from gpu_common import gpu_ffm, gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu
N, D, K = 80, 8, 4
RTOL, ATOL = 1e-8, 1e-11
GRID = list(itertools.product([2, 3, 4], ["explicit", "none", "augment"], [False, True], [False, True]))


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept", GRID)
def test_sgd_reference_grid(degree, fit_lower, fit_linear, fit_intercept):
    Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    perms = make_perms(N, 5)
    Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0,
                                             O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept), 5, n_aug,
                                             perms=perms)
    fm = gpu_fm("regression", degree, K, fit_lower, fit_linear, fit_intercept, P0, w0, b0)
    sgd = nf.newSGD(maxIter=5, verbose=0, tol=0)
    sgd.fit(to_gpu(Xo), y, fm, perms=perms)
    assert sgd.it == it
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[1] for h in sgd.history], el, 1e-9, 1e-12, "loss")
    assert_close([h[0] for h in sgd.history], ev, 1e-8, 1e-11, "viol")
    # the brute-force SGDSlow at the reference's own tolerance
    Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, degree, P0, w0, b0,
                                      O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept), 5, n_aug, perms)
    assert abs(fm.intercept - bs) < 1e-7
    assert_close(fm.w, ws, 1e-6, 1e-9, "w vs slow")
    assert_close(fm.P, Ps, 1e-6, 1e-9, "P vs slow")
    if not fit_linear:
        assert (fm.w == 0.0).all()  # test_sgd.nim:16-34
    if not fit_intercept:
        assert fm.intercept == 0.0  # test_sgd.nim:37-55


@pytest.mark.parametrize("loss,scheduling", itertools.product(["squared", "squared_hinge", "logistic", "huber"],
                                                              ["constant", "optimal", "invscaling", "pegasos"]))
def test_sgd_losses_and_schedules(loss, scheduling):
    degree, fit_lower = 3, "explicit"
    Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3, scale=0.3)
    task = "classification" if loss in ("squared_hinge", "logistic") else "regression"
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    kw = dict(alpha0=0.5, alpha=0.5, beta=0.5) if scheduling == "pegasos" else {}
    it0 = 20 if scheduling == "pegasos" else 1
    perms = make_perms(N, 3)
    yo = np.sign(y) if task == "classification" else y  # fm_base.nim:32-34 is applied by the library
    Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, yo, degree, P0, w0, b0, O.sgd_cfg(loss=loss, scheduling=scheduling, power=0.75, **kw),
                                  3, n_aug, perms=perms, it=it0)
    fm = gpu_fm(task, degree, K, fit_lower, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=3, verbose=0, tol=0, loss=loss, scheduling=scheduling, power=0.75, **kw)
    sgd.it = it0
    sgd.fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")


@pytest.mark.parametrize("degree,fit_lower,fit_linear,fit_intercept",
                         itertools.product([2, 3], ["explicit", "none", "augment"], [False, True], [False, True]))
def test_adagrad_reference_grid(degree, fit_lower, fit_linear, fit_intercept):
    Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, fit_linear, fit_intercept, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, fit_linear)
    perms = make_perms(N, 5)
    cfg = O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
    Pf, wf, bf, it, el, ev, _, st = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, cfg, 5, n_aug, perms=perms)
    fm = gpu_fm("regression", degree, K, fit_lower, fit_linear, fit_intercept, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=5, verbose=0, tol=0)
    ada.fit(to_gpu(Xo), y, fm, perms=perms)
    assert ada.it == it
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[1] for h in ada.history], el, 1e-9, 1e-12, "loss")
    assert_close([h[0] for h in ada.history], ev, 1e-8, 1e-11, "viol")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(fm)
    assert_close(gs, st.gsum_P, RTOL, ATOL, "g_sum.P")
    assert_close(gn, st.gnorm_P, RTOL, ATOL, "g_norm.P")
    if fit_linear:
        assert_close(gsw, st.gsum_w, RTOL, ATOL)
        assert_close(gnw, st.gnorm_w, RTOL, ATOL)
    if fit_intercept:
        assert abs(gsb - st.gsum_b.value) < 1e-9 and abs(gnb - st.gnorm_b.value) < 1e-9
    Ps, ws, bs, _ = O.slow_fm_adagrad_fit(Xd, y, degree, P0, w0, b0, cfg, 5, n_aug, perms)
    assert abs(fm.intercept - bs) < 1e-6
    assert_close(fm.w, ws, 1e-6, 1e-9, "w vs slow")
    assert_close(fm.P, Ps, 1e-6, 1e-9, "P vs slow")


@pytest.mark.parametrize("opt", ["sgd", "adagrad"])
def test_warm_start(opt):
    """test_sgd.nim:58-89 / test_adagrad.nim:58-89: 10 x fit(maxIter=1) == fit(maxIter=10), atol 1e-8."""
    for degree, fit_lower in [(2, "explicit"), (3, "augment"), (4, "none")]:
        Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower)
        P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
        X = to_gpu(Xo)
        mk = (lambda mi: nf.newSGD(maxIter=mi, verbose=0, tol=0, shuffle=False)) if opt == "sgd" else (
            lambda mi: nf.newAdaGrad(maxIter=mi, verbose=0, tol=0, shuffle=False))
        fmw = gpu_fm("regression", degree, K, fit_lower, True, True, P0, w0, b0)
        ow = mk(1)
        for _ in range(10):
            ow.fit(X, y, fmw)
        fm = gpu_fm("regression", degree, K, fit_lower, True, True, P0, w0, b0)
        o1 = mk(10)
        o1.fit(X, y, fm)
        assert ow.it == o1.it == 10 * N + 1
        assert abs(fm.intercept - fmw.intercept) < 1e-8
        assert_close(fm.w, fmw.w, atol=1e-8)
        assert_close(fm.P, fmw.P, atol=1e-8)


def test_callbacks_ncalls():
    """sgd.nim:303-307: callback sees a finalised model every nCalls steps; training is unaffected."""
    Xo, Xd, y = make_fm_dataset(N, D, 2, K, 42)
    P0, w0, b0, _ = init_fm(D, 2, K, "explicit", True)
    X = to_gpu(Xo)
    seen = []
    fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, nCalls=32)
    sgd.fit(X, y, fm, callback=lambda o, m: seen.append((o.it, m.P.copy())))
    assert [s[0] for s in seen] == [32, 64, 96, 128, 160]
    fm2 = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
    nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False).fit(X, y, fm2)
    assert_close(fm.P, fm2.P, 1e-10, 1e-13)
    Pq, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(), 1)
    # the callback after it=32 holds the model after 32 sequential steps of epoch 0
    Xo32 = O.Dataset(Xo.indptr[:33], Xo.indices[:Xo.indptr[32]], Xo.data[:Xo.indptr[32]], 32, D)
    P32, *_ = O.fm_sgd_fit(Xo32, y[:32], 2, P0, w0, b0, O.sgd_cfg(), 1)
    assert_close(seen[0][1], P32, RTOL, ATOL)


@pytest.mark.parametrize("fit_linear,fit_intercept", itertools.product([False, True], [False, True]))
def test_ffm_reference_grid(fit_linear, fit_intercept):
    n, d, F, k = 80, 20, 5, 4
    Xo, Xd, field_of, y = make_ffm_dataset(n, d, F, k, 42, threshold=0.3)
    P0, w0, b0 = init_ffm(d, F, k)
    X = to_gpu(Xo)
    Pf, wf, bf, it, el, ev, _ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept), 5)
    ffm = gpu_ffm("regression", k, fit_linear, fit_intercept, P0, w0, b0)
    sgd = nf.newSGD(maxIter=5, verbose=0, tol=0, shuffle=False)
    sgd.fit(X, y, ffm)
    assert abs(ffm.intercept - bf) < 1e-9
    assert_close(ffm.w, wf, RTOL, ATOL, "w")
    assert_close(ffm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], ev, 1e-8, 1e-11, "viol")
    Pf, wf, bf, it, el, ev, _, _ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0,
                                                     O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept), 5)
    ffm = gpu_ffm("regression", k, fit_linear, fit_intercept, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=5, verbose=0, tol=0, shuffle=False)
    ada.fit(X, y, ffm)
    assert abs(ffm.intercept - bf) < 1e-9
    assert_close(ffm.w, wf, RTOL, ATOL, "w")
    assert_close(ffm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[0] for h in ada.history], ev, 1e-8, 1e-11, "viol")


def test_stopping_criterion_and_scale_reset():
    Xo, Xd, y = make_fm_dataset(N, D, 2, K, 42, threshold=0.3)
    P0, w0, b0, _ = init_fm(D, 2, K, "explicit", True)
    X = to_gpu(Xo)
    fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=50, verbose=0, tol=1e9, shuffle=False)
    sgd.fit(X, y, fm)
    assert len(sgd.history) == 1  # viol < tol after the first epoch (sgd.nim:85-89)
    # scale_P drops below 1e-9 inside the epoch: the dense reset (sgd.nim:116-131) must be invisible
    cfg = dict(eta0=0.5, alpha=1.5, beta=1.5, scheduling="constant")
    Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(**cfg), 4)
    fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
    nf.newSGD(maxIter=4, verbose=0, tol=0, shuffle=False, **cfg).fit(X, y, fm)
    assert_close(fm.w, wf, 1e-7, 1e-11, "w")
    assert_close(fm.P, Pf, 1e-7, 1e-11, "P")


@pytest.mark.parametrize("k,max_m,d", [(1, 5, 9), (3, 40, 60), (16, 100, 150), (50, 64, 90), (64, 30, 40), (100, 12, 30)])
@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_pipelined_step_shapes(solver, k, max_m, d):
    """k_sequential_pipe (csrc/seq.hip): group counts G = 256 / S from 128 down to 2, 1 to 50 rows per thread, ragged and
    empty rows, unsorted storage order, consecutive samples that share most of their features (d is small: nearly every
    requested value is refreshed through LDS), a permuted order -- bit-level agreement with the one-sample-at-a-time
    oracle is the tolerance of the reference grids"""
    n = 120
    Xo = ragged_csr(n, d, seed=k + max_m, max_m=max_m, empty_every=7)
    rng = np.random.default_rng(k)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    perms = make_perms(n, 3)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.1)
    if solver == "sgd":
        Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, 2, P0, w0, 0.1, O.sgd_cfg(eta0=0.02), 3, 0, perms=perms)
        opt = nf.newSGD(maxIter=3, eta0=0.02, verbose=0, tol=0)
    else:
        Pf, wf, bf, it, el, ev, _, _ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, 0.1, O.adagrad_cfg(), 3, 0, perms=perms)
        opt = nf.newAdaGrad(maxIter=3, verbose=0, tol=0)
    opt.fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[1] for h in opt.history], el, 1e-9, 1e-12, "loss")
    assert_close([h[0] for h in opt.history], ev, 1e-8, 1e-11, "viol")


@pytest.mark.parametrize("F,k,d", [(3, 2, 12), (16, 8, 64), (39, 4, 120), (7, 20, 30)])
@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_pipelined_step_ffm_shapes(solver, F, k, d):
    """the field-aware step through the same kernel: fields with several entries, with one and with none in a sample,
    storage order not sorted by index (the pair order of sgd_ffm.nim:18-30 depends on it)"""
    n = 90
    rng = np.random.default_rng(F * 100 + k)
    field_of = rng.integers(0, F, size=d)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = 0 if i % 11 == 5 else int(rng.integers(1, min(d, max(F, 6)) + 1))  # F = 39: 39 x 39 rows of a sample in LDS
        idx = rng.choice(d, size=m, replace=False)
        if i % 3:
            idx = np.sort(idx)
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m))
        indptr.append(indptr[-1] + m)
    idx = np.concatenate(rows).astype(np.int64)
    Xo = O.Dataset(np.array(indptr), idx, np.concatenate(vals), n, d, field_of[idx], F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, k)
    X = to_gpu(Xo)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    if solver == "sgd":
        Pf, wf, bf, it, el, ev, _ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.05), 3)
        opt = nf.newSGD(maxIter=3, eta0=0.05, verbose=0, tol=0, shuffle=False)
    else:
        Pf, wf, bf, it, el, ev, _, _ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0, O.adagrad_cfg(), 3)
        opt = nf.newAdaGrad(maxIter=3, verbose=0, tol=0, shuffle=False)
    opt.fit(X, y, ffm)
    assert abs(ffm.intercept - bf) < 1e-9
    assert_close(ffm.w, wf, RTOL, ATOL, "w")
    assert_close(ffm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[0] for h in opt.history], ev, 1e-8, 1e-11, "viol")


@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_sequential_mode_is_bit_exact_where_the_arithmetic_is_the_same(solver):
    """The device code is built without fused multiply-adds (csrc/Makefile: the reference's generated C has none on
    x86-64).  Where the step then performs the reference's operations in the reference's order -- AdaGrad, and SGD without
    L2 decay (with decay the device keeps one global scale where the reference keeps per-feature snapshots) -- and the loss
    needs no exp / log, the result is the oracle's BIT FOR BIT: division and square root are correctly rounded on both
    sides.  viol is a sum over threads and stays within rounding."""
    n, d, k = 200, 70, 8
    Xo = ragged_csr(n, d, seed=11, max_m=30, empty_every=9)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(n) * 0.5
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    perms = make_perms(n, 3)
    for loss in ("squared", "squared_hinge", "huber"):
        yy = np.sign(y) if loss == "squared_hinge" else y
        task = "classification" if loss == "squared_hinge" else "regression"
        fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.1)
        if solver == "sgd":
            Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, yy, 2, P0, w0, 0.1, O.sgd_cfg(loss=loss, alpha=0.0, beta=0.0), 3, 0, perms=perms)
            nf.newSGD(maxIter=3, loss=loss, alpha=0.0, beta=0.0, verbose=0, tol=0).fit(to_gpu(Xo), yy, fm, perms=perms)
        else:
            Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, yy, 2, P0, w0, 0.1, O.adagrad_cfg(loss=loss), 3, 0, perms=perms)
            nf.newAdaGrad(maxIter=3, loss=loss, verbose=0, tol=0).fit(to_gpu(Xo), yy, fm, perms=perms)
        assert fm.intercept == bf, loss
        assert np.array_equal(fm.w, wf), loss
        assert np.array_equal(fm.P, Pf), loss


def test_sequential_ffm_adagrad_is_bit_exact():
    """field-aware AdaGrad: the pipelined step forms the prediction as sgd_ffm.nim:18-27 does -- one dot product per
    pair of entries, the pairs' terms added in visiting order -- so it equals the oracle bit for bit as well"""
    n, d, F, k = 150, 50, 6, 8
    rng = np.random.default_rng(9)
    Xr = ragged_csr(n, d, seed=21, max_m=12, empty_every=8)
    field_of = rng.integers(0, F, size=d)
    Xo = O.Dataset(Xr.indptr, Xr.indices, Xr.data, n, d, field_of[Xr.indices], F)
    y = rng.standard_normal(n) * 0.5
    P0, w0, b0 = init_ffm(d, F, k)
    Pf, wf, bf, *_ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0, O.adagrad_cfg(), 3)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    nf.newAdaGrad(maxIter=3, verbose=0, tol=0, shuffle=False).fit(to_gpu(Xo), y, ffm)
    assert ffm.intercept == bf
    assert np.array_equal(ffm.w, wf)
    assert np.array_equal(ffm.P, Pf)


def _long_row_csr(n, d, seed, long_every=5, long_m=700, short_m=9):
    """rows of several hundred entries between short ones (dataset.nim puts no bound on a row; text data has such rows)"""
    rng = np.random.default_rng(seed)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = long_m if i % long_every == 2 else short_m
        rows.append(np.sort(rng.choice(d, size=m, replace=False)))
        vals.append(rng.uniform(-1, 1, size=m) / np.sqrt(m))
        indptr.append(indptr[-1] + m)
    return O.Dataset(np.array(indptr), np.concatenate(rows), np.concatenate(vals), n, d)


@pytest.mark.parametrize("k,degree", [(64, 2), (24, 3), (200, 2)])
def test_rows_longer_than_the_lds_gradient_table(k, degree):
    """NFM_MODE_SEQUENTIAL on rows of 700 entries: the one-sample-in-flight kernel's per-sample gradient ([blocks][row][factors]:
    358 KB at k = 64) does not fit the 160 KB of LDS and lives in global memory (csrc/seq.hip, SeqArgs::dA_global); before
    round 5 such a dataset was NFM_ERR_UNSUPPORTED in this mode.  SGD and AdaGrad against the oracle."""
    n, d = 40, 1500
    Xo = _long_row_csr(n, d, seed=61)
    rng = np.random.default_rng(62)
    y = rng.standard_normal(n)
    P0, w0, b0, n_aug = init_fm(d, degree, k, "explicit", True, scale=0.05)
    perms = make_perms(n, 2)
    X = to_gpu(Xo)
    Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(), 2, n_aug, perms=perms)
    fm = gpu_fm("regression", degree, k, "explicit", True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0)
    sgd.fit(X, y, fm, perms=perms)
    assert sgd.it == it and abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], ev, 1e-8, 1e-11, "viol")
    cfg = O.adagrad_cfg()
    Pf, wf, bf, it, el, ev, _, st = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, cfg, 2, n_aug, perms=perms)
    fm = gpu_fm("regression", degree, k, "explicit", True, True, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=2, verbose=0, tol=0)
    ada.fit(X, y, fm, perms=perms)
    assert abs(fm.intercept - bf) < 1e-9
    assert_close(fm.w, wf, RTOL, ATOL, "w")
    assert_close(fm.P, Pf, RTOL, ATOL, "P")


def test_field_aware_rows_longer_than_the_lds_gradient_table():
    """the same for a field-aware model: 12 fields x 300 entries x k = 8"""
    n, d, F, k = 30, 900, 12, 8
    rng = np.random.default_rng(71)
    rows, vals, flds, indptr = [], [], [], [0]
    for i in range(n):
        m = 300 if i % 4 == 1 else 7
        idx = np.sort(rng.choice(d, size=m, replace=False))
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m) / np.sqrt(m))
        flds.append(idx % F)
        indptr.append(indptr[-1] + m)
    Xo = O.Dataset(np.array(indptr), np.concatenate(rows), np.concatenate(vals), n, d, fields=np.concatenate(flds), n_fields=F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
    perms = make_perms(n, 2)
    Pf, wf, bf, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(), 2, perms=perms)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0)
    sgd.fit(to_gpu(Xo), y, ffm, perms=perms)
    assert abs(ffm.intercept - bf) < 1e-9
    assert_close(ffm.w, wf, RTOL, ATOL, "w")
    assert_close(ffm.P, Pf, RTOL, ATOL, "P")
