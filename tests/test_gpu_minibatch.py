"""-m gpu: NFM_MODE_MINIBATCH (row phase / column phase kernels) vs the CPU restatement of the same
rule (oracle/nimfm_mb.c), and vs the reference-faithful sequential oracle at batch == 1."""
import itertools

import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_fm, make_fm_dataset, make_perms, random_csr
from gpu_common import gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu
N, D, K = 80, 8, 4
RTOL, ATOL = 1e-9, 1e-12


def run_oracle_sgd_mb(Xo, y, degree, P0, w0, b0, cfg, batch, n_aug, perms, epochs):
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    hist = []
    for e in range(epochs):
        b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, degree, P, w, b, cfg, batch, n_aug,
                                          perm=None if perms is None else perms[e], it=it)
        hist.append((vs, ls / Xo.n))
    return P, w, b, it, hist


@pytest.mark.parametrize("degree,fit_lower,batch", itertools.product([2, 3, 4], ["explicit", "none", "augment"],
                                                                     [1, 5, 16, 80, 1000]))
def test_sgd_vs_mb_oracle(degree, fit_lower, batch):
    Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    perms = make_perms(N, 3)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, degree, P0, w0, b0, O.sgd_cfg(), batch, n_aug, perms, 3)
    fm = gpu_fm("regression", degree, K, fit_lower, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=3, verbose=0, tol=0, mode="minibatch", batch=batch)
    sgd.fit(to_gpu(Xo), y, fm, perms=perms)
    assert sgd.it == it
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[1] for h in sgd.history], [h[1] for h in hist], 1e-10, 1e-13, "loss")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    if batch == 1:  # the rule collapses to the reference's sequential step
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(), 3, n_aug, perms=perms)
        assert_close(fm.P, Pf, 1e-8, 1e-11, "P vs sequential")
        assert_close(fm.w, wf, 1e-8, 1e-11, "w vs sequential")


@pytest.mark.parametrize("fit_linear,fit_intercept,loss", itertools.product([False, True], [False, True],
                                                                            ["squared", "logistic"]))
def test_sgd_flags_and_losses(fit_linear, fit_intercept, loss):
    Xo, Xd, y = make_fm_dataset(N, D, 2, K, 42, "explicit", fit_linear, fit_intercept, threshold=0.3)
    task = "classification" if loss == "logistic" else "regression"
    yo = np.sign(y) if task == "classification" else y
    P0, w0, b0, n_aug = init_fm(D, 2, K, "explicit", fit_linear)
    cfg = O.sgd_cfg(loss=loss, fit_linear=fit_linear, fit_intercept=fit_intercept, scheduling="invscaling", power=0.5)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, yo, 2, P0, w0, b0, cfg, 16, n_aug, None, 3)
    fm = gpu_fm(task, 2, K, "explicit", fit_linear, fit_intercept, P0, w0, b0)
    sgd = nf.newSGD(maxIter=3, verbose=0, tol=0, shuffle=False, loss=loss, scheduling="invscaling", power=0.5,
                    mode="minibatch", batch=16)
    sgd.fit(to_gpu(Xo), y, fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    if not fit_linear:
        assert (fm.w == 0).all()
    if not fit_intercept:
        assert fm.intercept == 0.0


@pytest.mark.parametrize("degree,fit_lower,batch,track", itertools.product([2, 3], ["explicit", "augment"],
                                                                           [1, 7, 80], [True]))
def test_adagrad_vs_mb_oracle(degree, fit_lower, batch, track):
    Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True)
    perms = make_perms(N, 3)
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(P.shape[0], P.shape[2], K, D)
    hist = []
    for e in range(3):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, batch, st, n_aug, perm=perms[e], it=it)
        hist.append((vs, ls / N))
    b = O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st, n_aug)
    fm = gpu_fm("regression", degree, K, fit_lower, True, True, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=3, verbose=0, tol=0, mode="minibatch", batch=batch, trackViol=track)
    ada.fit(to_gpu(Xo), y, fm, perms=perms)
    assert ada.it == it
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[1] for h in ada.history], [h[1] for h in hist], 1e-10, 1e-13, "loss")
    assert_close([h[0] for h in ada.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(fm)
    assert_close(gs, st.gsum_P, RTOL, ATOL, "g_sum")
    assert_close(gn, st.gnorm_P, RTOL, 1e-20, "g_norm")
    if batch == 1:
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, cfg, 3, n_aug, perms=perms)
        assert_close(fm.P, Pf, 1e-8, 1e-11, "P vs sequential")


@pytest.mark.parametrize("k", [1, 3, 8, 16, 30, 64, 128])
def test_component_counts_ragged(k):
    """every lanes-per-row instantiation, rows from empty to > 64 nnz, unsorted storage order"""
    n, d = 300, 200
    Xo = ragged_csr(n, d, seed=k)
    rng = np.random.default_rng(k)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * (0.1 / np.sqrt(k)), np.zeros(d)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, O.sgd_cfg(), 64, 0, None, 2)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=64)
    sgd.fit(to_gpu(Xo), y, fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")


def test_medium_problem_sgd_and_adagrad():
    """cfg2's shape scaled down to what the CPU oracle finishes in seconds: n=2e4, d=2e3, m=32, k=16."""
    n, d, m, k, B = 20000, 2000, 32, 16, 1024
    Xo = random_csr(n, d, m, seed=42)
    rng = np.random.default_rng(1)
    Pt = rng.standard_normal((1, k, d)) * 0.1
    y = np.sign(O.fm_decision_function(Xo, 2, Pt, np.zeros(d), 0.0))
    P0, w0 = rng.standard_normal((1, k, d)) * 0.01, np.zeros(d)
    X = to_gpu(Xo)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, O.sgd_cfg(loss="logistic"), B, 0, None, 2)
    fm = gpu_fm("classification", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=B)
    sgd.fit(X, y, fm)
    assert abs(fm.intercept - b) < 1e-10
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[1] for h in sgd.history], [h[1] for h in hist], 1e-10, 1e-13, "loss")
    cfg = O.adagrad_cfg(loss="logistic")
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(1, d, k, d)
    for e in range(2):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, B, st, it=it)
    b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    fm = gpu_fm("classification", 2, k, "explicit", True, True, P0, w0, 0.0)
    nf.newAdaGrad(maxIter=2, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=B).fit(X, y, fm)
    assert abs(fm.intercept - b) < 1e-10
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")


def test_maxthreads_overload_learns():
    """fit(X, y, fm, maxThreads) (sgd_multi.nim:40-42) maps to the mini-batch mode; score improves
    (tests/test_sgd.nim:129-151 on the parallel path)."""
    Xo, Xd, y = make_fm_dataset(400, 16, 2, 4, 42)
    P0, w0, b0, _ = init_fm(16, 2, 4, "explicit", True)
    X = to_gpu(Xo)
    fm = gpu_fm("regression", 2, 4, "explicit", True, True, P0, w0, b0)
    before = fm.score(X, y)
    nf.newSGD(maxIter=30, verbose=0, tol=0, alpha0=1e-9, alpha=1e-9, beta=1e-9, batch=32).fit(X, y, fm, maxThreads=4)
    assert fm.score(X, y) < before


@pytest.mark.parametrize("k,batch", [(4, 64), (16, 256), (64, 100)])
def test_sparse_regime_singles(k, batch):
    """d >> batch * nnz/row: most features are touched once per batch and are updated by the row phase
    itself (the "singles" path); the rest go through the column phase.  Same rule, same oracle."""
    n, d, m = 1500, 6000, 8
    Xo = random_csr(n, d, m, seed=3)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * (0.1 / np.sqrt(k)), np.zeros(d)
    perms = make_perms(n, 2)
    X = to_gpu(Xo)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, O.sgd_cfg(), batch, 0, perms, 2)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch)
    sgd.fit(X, y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(1, d, k, d)
    hist = []
    for e in range(2):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, batch, st, perm=perms[e], it=it)
        hist.append(vs)
    b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch)
    ada.fit(X, y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in ada.history], hist, 1e-9, 1e-12, "viol")
    # no permutation: the plan is reused and replayed as a hipGraph from the second epoch on
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, O.sgd_cfg(), batch, 0, None, 12)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    nf.newSGD(maxIter=12, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=batch).fit(X, y, fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")


@pytest.mark.parametrize("k,fit_linear,max_m", [(64, True, 64), (64, False, 63), (32, True, 64), (16, True, 40), (40, False, 17),
                                                (64, True, 150), (48, False, 64), (16, True, 100), (8, True, 70)])
def test_register_resident_rows(k, fit_linear, max_m):
    """Sparse regime, held-entries row phase: with one sample per wavefront (k > 8) the first 64 entries' parameter
    rows stay in registers and the singles are updated without a second visit (k_row_phase MODE 2); longer rows
    (max_m > 64, or 64 + the dummy feature) continue in further chunks.  Ragged rows, empty rows, unsorted storage order;
    fit_linear=False with fitLower=augment adds a dummy feature every sample touches (a heavy column)."""
    n, d = 700, 30000
    Xo = ragged_csr(n, d, seed=k + max_m, max_m=max_m)
    rng = np.random.default_rng(k)
    y = rng.standard_normal(n)
    fit_lower = "explicit" if fit_linear else "augment"
    n_aug = 0 if fit_linear else 1
    P0, w0 = rng.standard_normal((1, k, d + n_aug)) * (0.1 / np.sqrt(k)), np.zeros(d)
    perms = make_perms(n, 2)
    cfg = O.sgd_cfg(fit_linear=fit_linear)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, cfg, 128, n_aug, perms, 2)
    fm = gpu_fm("regression", 2, k, fit_lower, fit_linear, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=128)
    sgd.fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")


@pytest.mark.parametrize("k,F,batch", [(4, 5, 1), (4, 5, 7), (8, 16, 64), (30, 3, 80), (4, 4, 16), (8, 2, 32), (16, 16, 40)])
def test_ffm_minibatch_vs_mb_oracle(k, F, batch):
    """FieldAwareFactorizationMachine through the mini-batch kernels (mb_ffm.hip), SGD and AdaGrad."""
    from common import init_ffm, make_ffm_dataset
    from gpu_common import gpu_ffm
    # d divisible by F (tests/utils.nim:66-68); (16, 16): a sample's m*F rows need > 1/4 of the LDS (one wavefront per
    # workgroup); F = 4: rows of ~80 entries (> 64: the generic row kernel),
    # F = 2: rows that fit the LDS-resident kernel with many entries per field
    n, d = 120, {5: 60, 16: 48, 3: 48, 4: 200, 2: 64}[F]
    Xo, Xd, field_of, y = make_ffm_dataset(n, d, F, k, 42, threshold=0.6)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
    perms = make_perms(n, 3)
    X = to_gpu(Xo)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    hist = []
    for e in range(3):
        b, it, ls, vs = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, O.sgd_cfg(eta0=0.002), batch, perm=perms[e], it=it)
        hist.append((vs, ls / n))
    assert np.isfinite(b)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=3, verbose=0, tol=0, eta0=0.002, mode="minibatch", batch=batch)
    sgd.fit(X, y, ffm, perms=perms)
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, RTOL, ATOL, "w")
    assert_close(ffm.P, P, RTOL, ATOL, "P")
    assert_close([h[1] for h in sgd.history], [h[1] for h in hist], 1e-10, 1e-13, "loss")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    if batch == 1:
        Pf, wf, bf, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.002), 3, perms=perms)
        assert_close(ffm.P, Pf, 1e-8, 1e-11, "P vs sequential")
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(F, d, k, d)
    hist = []
    for e in range(3):
        b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, batch, st, perm=perms[e], it=it)
        hist.append(vs)
    b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=3, verbose=0, tol=0, mode="minibatch", batch=batch)
    ada.fit(X, y, ffm, perms=perms)
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, RTOL, ATOL, "w")
    assert_close(ffm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in ada.history], hist, 1e-9, 1e-12, "viol")


@pytest.mark.parametrize("k,batch", [(4, 700), (8, 257)])
def test_ffm_low_cardinality_fields(k, batch):
    """Field-aware data where some fields have only a few distinct features (each touched by a large share of every
    batch): those features go through k_ffm_heavy_partial / k_ffm_heavy_apply (segments of 64 touches, then one
    wavefront per (feature, field)); the rest through the ordinary column phase.  SGD and AdaGrad vs the oracle."""
    from common import init_ffm
    from gpu_common import gpu_ffm
    rng = np.random.default_rng(k)
    n, F, per = 1500, 4, 10
    cards = [2, 3, 10, 10]
    d = F * per
    idx = np.stack([f * per + rng.integers(0, cards[f], size=n) for f in range(F)], axis=1)
    val = rng.uniform(-1, 1, size=(n, F))
    Xo = O.Dataset(np.arange(n + 1) * F, idx.ravel(), val.ravel(), n, d, fields=np.tile(np.arange(F), n), n_fields=F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
    perms = make_perms(n, 2)
    X = to_gpu(Xo)
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    hist = []
    for e in range(2):
        b, it, ls, vs = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, O.sgd_cfg(eta0=0.01), batch, perm=perms[e], it=it)
        hist.append((vs, ls / n))
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, eta0=0.01, mode="minibatch", batch=batch)
    sgd.fit(X, y, ffm, perms=perms)
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, RTOL, ATOL, "w")
    assert_close(ffm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(F, d, k, d)
    hist = []
    for e in range(2):
        b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, batch, st, perm=perms[e], it=it)
        hist.append(vs)
    b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    ada = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch)
    ada.fit(X, y, ffm, perms=perms)
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, RTOL, ATOL, "w")
    assert_close(ffm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in ada.history], hist, 1e-9, 1e-12, "viol")


@pytest.mark.parametrize("degree,fit_lower,fit_linear", [(3, "augment", True), (4, "augment", False), (3, "explicit", True),
                                                         (1, "explicit", True)])
def test_heavy_features_higher_degree(degree, fit_lower, fit_linear):
    """Dummy features of fitLower=augment are touched by EVERY sample of a batch; with degree >= 3 (or several orders)
    they and other heavy columns go through the segmented kernels too (one partial record per (segment, order))."""
    n, d, k, batch = 900, 12, 4, 400
    Xo = random_csr(n, d, 6, seed=11)
    rng = np.random.default_rng(degree)
    y = rng.standard_normal(n)
    n_ord, n_aug = O.n_orders(degree, fit_lower), O.n_augments(degree, fit_lower, fit_linear)
    P0, w0 = rng.standard_normal((n_ord, k, d + n_aug)) * 0.1, np.zeros(d)
    perms = make_perms(n, 2)
    cfg = O.sgd_cfg(fit_linear=fit_linear)
    P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, degree, P0, w0, 0.0, cfg, batch, n_aug, perms, 2)
    fm = gpu_fm("regression", degree, k, fit_lower, fit_linear, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch)
    sgd.fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 1e-12, "viol")
    acfg = O.adagrad_cfg(fit_linear=fit_linear)
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(n_ord, d + n_aug, k, d)
    for e in range(2):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, acfg, batch, st, n_aug, perm=perms[e], it=it)
    b = O.fm_adagrad_finalize(degree, P, w, b, acfg, it, st, n_aug)
    fm = gpu_fm("regression", degree, k, fit_lower, fit_linear, True, P0, w0, 0.0)
    nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch).fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")


@pytest.mark.parametrize("solver,k", [("sgd", 4), ("sgd", 16), ("adagrad", 8)])
def test_heavy_features(solver, k):
    """Features touched by (almost) every sample of a batch -- Zipf heads, dummy features -- have their touch
    lists cut into segments summed by separate lane groups (k_heavy_partial / k_heavy_apply); the result must
    still be the rule's (sums are re-associated per segment, hence rtol 1e-9)."""
    n, d, batch = 3000, 60, 1000
    rng = np.random.default_rng(11)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        rest = rng.choice(np.arange(2, d), size=4, replace=False)
        idx = np.concatenate([[0], [1] if i % 3 else [], rest]).astype(np.int64)  # feature 0 in every row, 1 in 2/3
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=len(idx)))
        indptr.append(indptr[-1] + len(idx))
    Xo = O.Dataset(np.array(indptr), np.concatenate(rows), np.concatenate(vals), n, d)
    y = rng.standard_normal(n)
    # fit_lower=augment with fit_linear=False: one dummy feature, touched by every sample
    for fit_lower, fit_linear in (("explicit", True), ("augment", False)):
        n_aug = O.n_augments(2, fit_lower, fit_linear)
        P0, w0 = rng.standard_normal((1, k, d + n_aug)) * 0.05, np.zeros(d)
        X = to_gpu(Xo)
        fm = gpu_fm("regression", 2, k, fit_lower, fit_linear, True, P0, w0, 0.0)
        if solver == "sgd":
            cfg = O.sgd_cfg(fit_linear=fit_linear)
            P, w, b, it, hist = run_oracle_sgd_mb(Xo, y, 2, P0, w0, 0.0, cfg, batch, n_aug, None, 3)
            opt = nf.newSGD(maxIter=3, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=batch)
            hv = [h[0] for h in hist]
        else:
            cfg = O.adagrad_cfg(fit_linear=fit_linear)
            P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
            st = O.AdaState(1, d + n_aug, k, d)
            hv = []
            for e in range(3):
                b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, batch, st, n_aug, it=it)
                hv.append(vs)
            b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st, n_aug)
            opt = nf.newAdaGrad(maxIter=3, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=batch)
        opt.fit(X, y, fm)
        assert abs(fm.intercept - b) < 1e-11
        assert_close(fm.w, w, RTOL, ATOL, "w")
        assert_close(fm.P, P, RTOL, ATOL, "P")
        assert_close([h[0] for h in opt.history], hv, 1e-9, 1e-12, "viol")


@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_device_shuffle_replays_on_the_oracle(solver):
    """shuffle = true with the order drawn on the device (nfm_opt_set_shuffle): every epoch gets a fresh permutation, the
    NEXT epoch's plan is built beside the current epoch on a second stream, and nfm_opt_get_perm hands back the order
    that was used -- replayed on the CPU restatement of the mini-batch rule it must give the same parameters."""
    n, d, m, k, B, epochs = 5000, 300, 8, 8, 512, 4
    Xo = random_csr(n, d, m, seed=31)
    rng = np.random.default_rng(6)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    X = to_gpu(Xo)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=1, verbose=0, tol=0, shuffle=True, mode="minibatch", batch=B,
                                                             deviceShuffle=True)
    X.set_targets(y)
    opt._handle(fm, X.ctx, "minibatch")
    import ctypes as C
    from nimfm_amd import _capi as capi
    capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, 7))
    perms, hist = [], []
    for e in range(epochs):
        ls, vs = opt._epoch(X, None, 0, n)
        opt.it += n
        perms.append(opt.last_permutation(n))
        hist.append((ls, vs))
    opt._finalize_into(fm)
    for p in perms:
        assert np.array_equal(np.sort(p), np.arange(n))
    assert not any(np.array_equal(perms[a], perms[b]) for a in range(epochs) for b in range(a))
    assert abs(np.mean(perms[0][: n // 2]) - n / 2) < n / 10  # no obvious structure
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    if solver == "sgd":
        for e in range(epochs):
            b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(), B, perm=perms[e], it=it)
            assert_close([ls, vs], hist[e], 1e-9, 0, "loss / viol of epoch %d" % e)
    else:
        cfg = O.adagrad_cfg()
        st = O.AdaState(1, d, k, d)
        for e in range(epochs):
            b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, B, st, perm=perms[e], it=it)
            assert_close([ls, vs], hist[e], 1e-9, 1e-12, "loss / viol of epoch %d" % e)
        b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")
    # the same seed draws the same orders again; through fit() the seed is the model's randomState
    capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, 7))
    opt._epoch(X, None, 0, n)
    assert np.array_equal(opt.last_permutation(n), perms[0])


@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 64, 1000, 4097, 65537])
def test_device_shuffle_is_a_permutation_at_awkward_sizes(n):
    """the device order is a cycle-walked Feistel bijection over the next even power of two (csrc/plan.hip): it must be a
    permutation of the range for every n (also 1, 2^k and 2^k + 1), spread positions evenly, and change with the epoch"""
    d, m, k = 50, 4, 4
    Xo = random_csr(n, d, m, seed=5)
    X = to_gpu(Xo)
    X.set_targets(np.zeros(n))
    rng = np.random.default_rng(2)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, rng.standard_normal((1, k, d)) * 0.05, np.zeros(d), 0.0)
    opt = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=True, mode="minibatch", batch=32, deviceShuffle=True)
    opt._handle(fm, X.ctx, "minibatch")
    from nimfm_amd import _capi as capi
    capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, 11))
    epochs = 24 if n <= 4097 else 3
    perms = []
    for e in range(epochs):
        opt._epoch(X, None, 0, n)
        opt.it += n
        perms.append(opt.last_permutation(n))
        assert np.array_equal(np.sort(perms[-1]), np.arange(n))
    if n >= 1000:
        assert not any(np.array_equal(perms[a], perms[b]) for a in range(epochs) for b in range(a))
        pos = np.stack([np.argsort(p) for p in perms])  # position of every sample in every epoch
        # the mean position of a sample over the epochs is n/2 +- n/sqrt(12 epochs); 6 sigma over all samples
        assert np.max(np.abs(pos.mean(0) - (n - 1) / 2)) < 6 * n / np.sqrt(12 * epochs)
        # neighbours in the data are not neighbours in the order more often than chance (2/n per pair)
        adj = np.mean([np.mean(np.abs(np.diff(pp)) == 1) for pp in pos])
        assert adj < 2.0 / n + 0.01


def test_announced_permutation_is_only_a_hint():
    """nfm_opt_announce_perm: the plan built beside the current epoch is used only when the next call passes the SAME
    array; another array (or a changed one) is planned afresh -- results never depend on the announcement."""
    from nimfm_amd import _capi as capi
    n, d, m, k, B = 3000, 200, 8, 8, 256
    Xo = random_csr(n, d, m, seed=17)
    rng = np.random.default_rng(4)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    perms = make_perms(n, 4, seed=3)
    decoy = np.ascontiguousarray(perms[3][::-1])
    X = to_gpu(Xo)
    X.set_targets(y)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, mode="minibatch", batch=B)
    sgd._handle(fm, X.ctx, "minibatch")
    announce = lambda a: capi.check(capi.lib().nfm_opt_announce_perm(sgd._h, a.ctypes.data, 0, n))  # noqa: E731
    announce(perms[1])
    sgd._epoch(X, perms[0], 0, n)          # epoch 0; the plan of perms[1] is built beside it
    sgd.it += n
    sgd._epoch(X, perms[1], 0, n)          # ... and used here
    sgd.it += n
    announce(decoy)
    sgd._epoch(X, perms[2], 0, n)          # the announcement is not followed up: planned afresh
    sgd.it += n
    changed = perms[3].copy()
    announce(changed)
    sgd._epoch(X, perms[2], 0, n)
    sgd.it += n
    changed[[0, -1]] = changed[[-1, 0]]    # the promised array was modified: the probes notice, planned afresh
    sgd._epoch(X, changed, 0, n)
    sgd._finalize_into(fm)
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    for p in (perms[0], perms[1], perms[2], perms[2], changed):
        b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(), B, perm=p, it=it)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert_close(fm.P, P, RTOL, ATOL, "P")


@pytest.mark.parametrize("dense", [True, False])
@pytest.mark.parametrize("with_perm", [False, True])
def test_epoch_over_a_sub_range(dense, with_perm):
    """nfm_opt_epoch over [begin, end) of the order (what the nCalls callbacks and the Hogwild-style slices use): a dense
    shape takes its plan from the column-major twin -- the samples outside the range must drop out of every column's touch
    list --, a sparse one from the sort; three consecutive ranges must equal the CPU restatement run over the same ranges."""
    n, k, B = 3000, 8, 256
    d, m = (100, 12) if dense else (20000, 12)
    Xo = random_csr(n, d, m, seed=77)
    rng = np.random.default_rng(8)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    perm = rng.permutation(n).astype(np.int64) if with_perm else None
    X = to_gpu(Xo)
    X.set_targets(y)
    for solver in ("sgd", "adagrad"):
        fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
        opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
        opt._handle(fm, X.ctx, "minibatch")
        P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
        st = O.AdaState(1, d, k, d)
        for lo, hi in ((0, 700), (700, 2950), (2950, n)):
            ls, vs = opt._epoch(X, perm, lo, hi)
            opt.it += hi - lo
            if solver == "sgd":
                b, it, lw, vw = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(), B, perm=perm, begin=lo, end=hi, it=it)
            else:
                b, it, lw, vw = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, O.adagrad_cfg(), B, st, perm=perm, begin=lo, end=hi, it=it)
            assert_close([ls, vs], [lw, vw], 1e-9, 1e-12, "%s loss / viol of [%d, %d)" % (solver, lo, hi))
        opt._finalize_into(fm)
        if solver == "adagrad":
            b = O.fm_adagrad_finalize(2, P, w, b, O.adagrad_cfg(), it, st)
        assert abs(fm.intercept - b) < 1e-11
        assert_close(fm.w, w, RTOL, ATOL, solver + " w")
        assert_close(fm.P, P, RTOL, ATOL, solver + " P")


@pytest.mark.parametrize("cap,model,batch", itertools.product([2.0, 4.0, 16.0], ["fm2", "fm3", "ffm"], [7, 64, 1000]))
def test_sgd_touch_cap_vs_mb_oracle(cap, model, batch):
    """nfm_opt_set_touch_cap: up to `cap` of a batch's per-sample steps on one coordinate are summed, beyond that the sum
    is scaled by cap / c (the reference's Hogwild threads apply their steps at full strength, sgd_multi.nim:83-101;
    cap = 1, the default of every other test here, is the per-coordinate mean).  Sparse and dense coordinates, the
    intercept (touched by every sample), features past the 64-entry decay table and the segment path (> 128 touches)."""
    from common import init_ffm
    from gpu_common import gpu_ffm
    rng = np.random.default_rng(int(cap) * 100 + batch)
    n, d, k = 1200, 90, 8
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = int(rng.integers(1, 12))
        p = np.full(d, 1.0)
        p[:2] = d  # two features most samples have: far more than 64 / 128 touches per batch of 1000
        idx = np.sort(rng.choice(d, size=m, replace=False, p=p / p.sum()))
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m))
        indptr.append(indptr[-1] + m)
    idx, val, indptr = np.concatenate(rows).astype(np.int64), np.concatenate(vals), np.array(indptr)
    y = rng.standard_normal(n)
    perms = make_perms(n, 2)
    cfg = O.sgd_cfg(eta0=0.02, scheduling="invscaling", power=0.5)
    kw = dict(maxIter=2, eta0=0.02, scheduling="invscaling", power=0.5, verbose=0, tol=0, mode="minibatch", batch=batch, touchCap=cap)
    if model == "ffm":
        F = 5
        field_of = rng.integers(0, F, size=d)
        Xo = O.Dataset(indptr, idx, val, n, d, field_of[idx], F)
        P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
        P, w, b, it = P0.copy(), w0.copy(), 0.1, 1
        for e in range(2):
            b, it, _, _ = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, cfg, batch, perm=perms[e], it=it, touch_cap=cap)
        mdl = gpu_ffm("regression", k, True, True, P0, w0, 0.1)
    else:
        degree = 2 if model == "fm2" else 3
        Xo = O.Dataset(indptr, idx, val, n, d)
        P0, w0 = rng.standard_normal((degree - 1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
        P, w, b, it = P0.copy(), w0.copy(), 0.1, 1
        for e in range(2):
            b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, degree, P, w, b, cfg, batch, perm=perms[e], it=it, touch_cap=cap)
        mdl = gpu_fm("regression", degree, k, "explicit", True, True, P0, w0, 0.1)
    opt = nf.newSGD(**kw)
    opt.fit(to_gpu(Xo), y, mdl, perms=perms)
    assert np.isfinite(P).all()
    assert opt.it == it and abs(mdl.intercept - b) < 1e-10
    assert_close(mdl.w, w, RTOL, ATOL, "w")
    assert_close(mdl.P, P, RTOL, ATOL, "P")


@pytest.mark.parametrize("batch,cap", [(64, 1.0), (300, 16.0), (2048, 1.0)])
def test_ffm_sparse_regime_small_batches(batch, cap):
    """Field-aware models in the sparse regime (batch x entries / d << 1: the small batches bench.py runs BASELINE configs[3]
    at): most features of a batch are touched by ONE sample, a few popular ones by many; empty rows, several entries per
    field, unsorted storage order; SGD (touch cap 1 / 16) and AdaGrad, two permuted epochs, against the CPU restatement of
    the rule (optimizer/sgd_ffm.nim:43, adagrad_ffm.nim:11-66: all nFields rows of every touched feature) at rtol 1e-9.
    (A variant of the row phase that updated the single-touch features itself was built in round 4, passed this test bit
    for bit against the two-phase path, measured +-0 and was not kept: DESIGN.md section 11.)"""
    import os
    from common import init_ffm
    from gpu_common import gpu_ffm
    rng = np.random.default_rng(batch)
    n, d, F, k = 2500, 24_000, 8, 8
    field_of = rng.integers(0, F, size=d)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = 0 if i % 13 == 5 else int(rng.integers(1, 9))
        p = np.full(d, 1.0)
        p[:40] = 60.0  # a few popular features: shared by several samples of most batches (the column phase's share)
        idx = rng.choice(d, size=m, replace=False, p=p / p.sum())
        if i % 3:
            idx = np.sort(idx)
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m))
        indptr.append(indptr[-1] + m)
    idx = np.concatenate(rows).astype(np.int64)
    Xo = O.Dataset(np.array(indptr), idx, np.concatenate(vals), n, d, field_of[idx], F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
    w0 = rng.standard_normal(d) * 0.01
    perms = make_perms(n, 2)
    X = to_gpu(Xo)

    def gpu(kind, singles):
        old = os.environ.get("NFM_SINGLES")
        os.environ["NFM_SINGLES"] = "1" if singles else "0"
        try:
            ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
            if kind == "sgd":
                opt = nf.newSGD(maxIter=2, verbose=0, tol=0, eta0=0.01, mode="minibatch", batch=batch, touchCap=cap)
            else:
                opt = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch)
            ctx = nf.default_context()
            ctx.timing_enable(True)
            ctx.timing_reset()
            opt.fit(X, y, ffm, perms=perms)
            n_col = ctx.timing_get("col_phase")[0]
            ctx.timing_enable(False)
            st = opt.get_state(ffm) if kind == "adagrad" else None
            return ffm.P.copy(), ffm.w.copy(), ffm.intercept, list(opt.history), st, n_col
        finally:
            if old is None:
                os.environ.pop("NFM_SINGLES", None)
            else:
                os.environ["NFM_SINGLES"] = old

    for kind in ("sgd", "adagrad"):
        a = gpu(kind, True)
        P, w, b, it = P0.copy(), w0.copy(), b0, 1
        hist = []
        if kind == "sgd":
            for e in range(2):
                b, it, ls, vs = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, O.sgd_cfg(eta0=0.01), batch, perm=perms[e], it=it, touch_cap=cap)
                hist.append((vs, ls / n))
        else:
            cfg = O.adagrad_cfg()
            st = O.AdaState(F, d, k, d)
            for e in range(2):
                b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, batch, st, perm=perms[e], it=it)
                hist.append((vs, ls / n))
            b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
            assert_close(a[4][0], st.gsum_P, 1e-9, 1e-13, "g_sum")
            assert_close(a[4][1], st.gnorm_P, 1e-9, 1e-16, "g_norm")
        assert abs(a[2] - b) < 1e-11
        assert_close(a[1], w, RTOL, ATOL, kind + " w")
        assert_close(a[0], P, RTOL, ATOL, kind + " P")
        assert_close([h[1] for h in a[3]], [h[1] for h in hist], 1e-10, 1e-13, kind + " loss")
        assert_close([h[0] for h in a[3]], [h[0] for h in hist], 1e-9, 1e-12, kind + " viol")
