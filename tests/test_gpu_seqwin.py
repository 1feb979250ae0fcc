"""-m gpu: the dependency-window form of NFM_MODE_SEQUENTIAL (seqwin.hip) -- the reference's one-sample-at-a-time order
(optimizer/sgd.nim:246-258,294-308, optimizer/adagrad.nim:169-184) spread over the chip.

Two bars: (1) BIT FOR BIT the parameters / linear weights / intercept / AdaGrad state of the one-workgroup kernels
(NFM_SEQ_WIN=0), which test_gpu_sequential.py holds to the oracle -- on conflict-heavy data (8 features: every sample
depends on its predecessor), ragged rows, every lanes-per-row value, small and large windows; (2) the oracle's
reference-faithful fit (O.fm_sgd_fit / O.fm_adagrad_fit) at rtol 1e-8 on the reference's grids and on sparse shapes
where most samples run concurrently.  Hand-offs are exercised under uneven load by construction: rows of 0 ... 100
entries make the workers' step times differ by two orders of magnitude."""
import itertools
import os

import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_fm, make_fm_dataset, make_perms, random_csr
from gpu_common import gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu

# Two flavours of the window when the intercept is fitted (seqwin.hip): "one_term" (the default: the worker adds up its
# sample's prediction but the intercept, the conductor's chain is b + S -> dloss -> b') and "exact" (NFM_SEQ_WIN_EXACT=1: the
# conductor adds the sample's terms onto the intercept one by one, the reference's rounding).  The exact flavour and every fit
# without an intercept (no conductor at all) are held to the one-workgroup kernel BIT FOR BIT; the one-term flavour rounds
# yhat differently (b + (sum) instead of ((b + t1) + t2) + ...), so it is held to the same results at rtol 1e-8 -- the
# tolerance of the oracle comparisons (north_star: 1e-6 relative on predictions).
_ONE_TERM = {"variant": False, "last_fit": False}


@pytest.fixture(autouse=True, params=["one_term", "exact"])
def window_flavour(request):
    with env(NFM_SEQ_WIN_EXACT=1 if request.param == "exact" else 0):
        _ONE_TERM["variant"] = request.param == "one_term"
        _ONE_TERM["last_fit"] = False
        yield request.param


def _note_fit(win, fit_intercept=True):
    """a fit through the window with a fitted intercept in the one-term flavour: its results are compared with a tolerance"""
    if int(win) != 0 and fit_intercept and _ONE_TERM["variant"]:
        _ONE_TERM["last_fit"] = True


def _tol(rtol, atol):
    """the per-epoch totals: tight when the fits are equal bit for bit, the oracle comparisons' tolerance otherwise"""
    return (1e-8, 1e-11) if _ONE_TERM["last_fit"] else (rtol, atol)


def same_b(a, b):
    """intercepts (or any scalars) equal: as bits, or at the one-term flavour's tolerance"""
    return abs(a - b) <= 1e-11 + 1e-8 * abs(b) if _ONE_TERM["last_fit"] else a == b


class env:
    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fallbacks():
    """calls of this context that began in the window kernel and ended in the one-workgroup kernel (an aborted launch)"""
    return nf.default_context().timing_get("seq_window_fallback")[0]


def fit(kind, win, W, Xo, y, task, k, P0, w0, b0, epochs, perms=None, it0=1, nCalls=-1, fit_linear=True, fit_intercept=True,
        expect_fallbacks=0, **kw):
    """one fit in sequential mode; win = 0: the one-workgroup kernels, 2: the window kernel with W workers.  nCalls > 0:
    the reference's per-nCalls callbacks (sgd.nim:303-308): the epoch becomes a series of calls over sub-ranges of the
    order with a finalize in between"""
    _note_fit(win, fit_intercept)
    with env(NFM_SEQ_WIN=win, NFM_SEQ_WIN_W=W):
        fm = gpu_fm(task, 2, k, "explicit", fit_linear, fit_intercept, P0, w0, b0)
        mk = nf.newSGD if kind == "sgd" else nf.newAdaGrad
        opt = mk(maxIter=epochs, verbose=0, tol=0, shuffle=False, mode="sequential", nCalls=nCalls, **kw)
        opt.it = it0
        seen = []
        fb0 = _fallbacks()
        opt.fit(to_gpu(Xo), y, fm, perms=perms, callback=(lambda o_, m_: seen.append(o_.it)) if nCalls > 0 else None)
        assert _fallbacks() - fb0 == expect_fallbacks, "window launches that aborted and were re-run by the one-workgroup kernel"
        state = opt.get_state(fm) if kind == "adagrad" else None
        return fm.P.copy(), fm.w.copy(), fm.intercept, opt.it, list(opt.history), state


def same_bits(a, b, what):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, what
    if _ONE_TERM["last_fit"]:  # (the one-term chain ran: same results to rounding, see the top of the file)
        return assert_close(a, b, 1e-8, 1e-11, what)
    bad = a.view(np.uint64) != b.view(np.uint64)
    assert not bad.any(), "%s: %d of %d words differ, max |diff| %.3e" % (what, bad.sum(), bad.size, np.abs(a - b)[bad].max())


def check_pair(kind, Xo, y, task, k, W, epochs=2, seed=0, perms=None, oracle_rtol=1e-8, **kw):
    rng = np.random.default_rng(seed)
    d = Xo.d
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.1 / np.sqrt(k), rng.standard_normal(d) * 0.01, 0.05
    ref = fit(kind, 0, W, Xo, y, task, k, P0, w0, b0, epochs, perms, **kw)
    win = fit(kind, 2, W, Xo, y, task, k, P0, w0, b0, epochs, perms, **kw)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3]
    if ref[4]:
        assert_close([h[1] for h in win[4]], [h[1] for h in ref[4]], *_tol(1e-12, 1e-15), "loss per epoch")
        assert_close([h[0] for h in win[4]], [h[0] for h in ref[4]], *_tol(1e-11, 1e-14), "viol per epoch")
    if kind == "adagrad":
        for g, h, name in zip(win[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
    return win, (P0, w0, b0)


@pytest.mark.parametrize("kind,k,W", itertools.product(["sgd", "adagrad"], [1, 3, 4, 8, 16, 30, 64], [8, 64]))
def test_conflict_heavy_ragged_rows_bitwise(kind, k, W):
    """8 ... 40 features, rows from empty to longer than a wavefront, unsorted storage order: nearly every sample waits for
    its predecessors, the waits resolve in every possible order"""
    d = 8 if k % 2 == 0 else 40
    Xo = ragged_csr(400, d, seed=k, max_m=d)
    y = np.random.default_rng(k).standard_normal(Xo.n)
    check_pair(kind, Xo, y, "regression", k, W, epochs=2, seed=k)


@pytest.mark.parametrize("kind,W", itertools.product(["sgd", "adagrad"], [8, 32, 128]))
def test_long_rows_and_more_features_bitwise(kind, W):
    Xo = ragged_csr(1500, 300, seed=3, max_m=100)  # rows of up to 100 entries: two 64-lane chunks
    rng = np.random.default_rng(4)
    y = np.sign(rng.standard_normal(Xo.n))
    perms = make_perms(Xo.n, 2)
    check_pair(kind, Xo, y, "classification", 16, W, epochs=2, perms=perms, loss="logistic")


@pytest.mark.parametrize("loss,scheduling", itertools.product(["squared", "squared_hinge", "logistic", "huber"],
                                                              ["constant", "optimal", "invscaling", "pegasos"]))
def test_sgd_losses_and_schedules_bitwise_and_oracle(loss, scheduling):
    n, d, k = 600, 50, 8
    Xo = random_csr(n, d, 6, seed=11)
    rng = np.random.default_rng(12)
    task = "classification" if loss in ("squared_hinge", "logistic") else "regression"
    y = rng.standard_normal(n)
    kw = dict(alpha0=0.5, alpha=0.5, beta=0.5) if scheduling == "pegasos" else {}
    it0 = 20 if scheduling == "pegasos" else 1
    perms = make_perms(n, 2)
    win, (P0, w0, b0) = check_pair("sgd", Xo, y, task, k, 16, epochs=2, perms=perms, it0=it0, loss=loss, scheduling=scheduling,
                                   power=0.75, **kw)
    yo = np.sign(y) if task == "classification" else y
    Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, yo, 2, P0, w0, b0, O.sgd_cfg(loss=loss, scheduling=scheduling, power=0.75, **kw), 2,
                                  0, perms=perms, it=it0)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("fit_linear,fit_intercept,kind", itertools.product([False, True], [False, True], ["sgd", "adagrad"]))
def test_reference_grid_flags(fit_linear, fit_intercept, kind):
    """the reference's degree-2 grid (tests/test_sgd.nim:92-126, test_adagrad.nim:92-126: n = 80, d = 8, k = 4) through the
    window kernel, against the oracle's fit and the brute-force model at the reference's tolerance"""
    N, D, K = 80, 8, 4
    Xo, Xd, y = make_fm_dataset(N, D, 2, K, 42, "explicit", fit_linear, fit_intercept, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(D, 2, K, "explicit", fit_linear)
    perms = make_perms(N, 5)
    win = fit(kind, 2, 8, Xo, y, "regression", K, P0, w0, b0, 5, perms, fit_linear=fit_linear, fit_intercept=fit_intercept)
    if kind == "sgd":
        cfg = O.sgd_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
        Pf, wf, bf, it, el, ev, _ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, cfg, 5, n_aug, perms=perms)
        Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, 2, P0, w0, b0, cfg, 5, n_aug, perms)
    else:
        cfg = O.adagrad_cfg(fit_linear=fit_linear, fit_intercept=fit_intercept)
        Pf, wf, bf, it, el, ev, _, st = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, cfg, 5, n_aug, perms=perms)
        Ps, ws, bs, _ = O.slow_fm_adagrad_fit(Xd, y, 2, P0, w0, b0, cfg, 5, n_aug, perms)
    assert win[3] == it
    assert_close(win[0], Pf, 1e-8, 1e-11, "P")
    assert_close(win[1], wf, 1e-8, 1e-11, "w")
    assert abs(win[2] - bf) < 1e-9
    assert_close([h[1] for h in win[4]], el, 1e-9, 1e-12, "loss")
    assert_close([h[0] for h in win[4]], ev, 1e-8, 1e-11, "viol")
    assert_close(win[0], Ps, 1e-6, 1e-9, "P vs slow")
    assert_close(win[1], ws, 1e-6, 1e-9, "w vs slow")
    if not fit_linear:
        assert (win[1] == 0.0).all()
    if not fit_intercept:
        assert win[2] == 0.0


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_bit_exact_against_the_oracle_where_the_arithmetic_is_the_same(kind):
    """no L2 decay (scale stays 1) and a loss without exp / log: the device's arithmetic is the oracle's, operation for
    operation -- the window kernel must then equal the reference restatement to the last bit (AdaGrad always does)"""
    n, d, k = 3000, 2000, 16
    Xo = random_csr(n, d, 12, seed=21)
    rng = np.random.default_rng(22)
    y = rng.standard_normal(n)
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d), 0.0
    perms = make_perms(n, 2)
    if kind == "sgd":
        kw = dict(alpha0=0.0, alpha=0.0, beta=0.0, scheduling="constant", loss="squared")
        cfg = O.sgd_cfg(**kw)
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, cfg, 2, 0, perms=perms)
    else:
        kw = dict(loss="squared")
        cfg = O.adagrad_cfg(**kw)
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, cfg, 2, 0, perms=perms)
    win = fit(kind, 2, 64, Xo, y, "regression", k, P0, w0, b0, 2, perms, **kw)
    same_bits(win[0], Pf, "P vs oracle")
    same_bits(win[1], wf, "w vs oracle")
    assert same_b(win[2], bf)


def test_reset_scaling_mid_epoch():
    """a step size x beta large enough that scaling_P falls below 1e-9 several times inside one epoch (resetScaling,
    sgd.nim:116-131): the launch is cut there, the dense rescale runs, the next launch continues"""
    n, d, k = 2500, 400, 8
    Xo = random_csr(n, d, 8, seed=31)
    y = np.random.default_rng(32).standard_normal(n)
    perms = make_perms(n, 2)
    win, (P0, w0, b0) = check_pair("sgd", Xo, y, "regression", k, 32, epochs=2, perms=perms, eta0=0.05, beta=0.5, alpha=0.3,
                                   scheduling="constant")
    Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(eta0=0.05, beta=0.5, alpha=0.3, scheduling="constant"), 2, 0,
                                  perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_sparse_shape_most_samples_concurrent(kind):
    """d = 1e5, 32 entries per row: two samples share a feature with probability 1 % -- the window is full of samples
    running side by side; fixed order, no permutation (the dependency table is reused by the second epoch)"""
    n, d, k = 20000, 100000, 16
    rng = np.random.default_rng(41)
    idx = np.sort(rng.integers(0, d, size=(n, 32)), axis=1)
    idx += np.arange(32)  # distinct inside a row
    idx %= d
    idx.sort(axis=1)
    ok = (np.diff(idx, axis=1) > 0).all(axis=1)
    idx = idx[ok]
    n = len(idx)
    Xo = O.Dataset(np.arange(n + 1, dtype=np.int64) * 32, idx.ravel().astype(np.int64), rng.uniform(-1, 1, n * 32), n, d)
    y = np.sign(rng.standard_normal(n))
    win, (P0, w0, b0) = check_pair(kind, Xo, y, "classification", k, 64, epochs=2, loss="logistic")
    if kind == "sgd":
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(loss="logistic"), 2, 0)
    else:
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, O.adagrad_cfg(loss="logistic"), 2, 0)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_sub_ranges_of_the_order(kind):
    """nCalls = 37: every epoch is a series of calls over 37-sample stretches of the permuted order (positions relative to
    the call's first sample in the dependency table), a finalize + read-back between them"""
    Xo = ragged_csr(500, 60, seed=5, max_m=30)
    y = np.random.default_rng(6).standard_normal(Xo.n)
    perms = make_perms(Xo.n, 2)
    check_pair(kind, Xo, y, "regression", 8, 8, epochs=2, perms=perms, nCalls=37)


def _ffm_data(n, d, F, max_m, seed, one_per_field=False):
    rng = np.random.default_rng(seed)
    field_of = rng.integers(0, F, size=d) if not one_per_field else np.arange(d) // (d // F)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        if one_per_field:  # one entry per field (the benchmark shape, cfg4)
            idx = rng.integers(0, d // F, size=F) + np.arange(F) * (d // F)
        else:
            m = 0 if i % 11 == 5 else int(rng.integers(1, max_m + 1))
            idx = rng.choice(d, size=m, replace=False)
            if i % 3:
                idx = np.sort(idx)  # the pair order of sgd_ffm.nim:18-30 depends on the storage order
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=len(idx)))
        indptr.append(indptr[-1] + len(idx))
    idx = np.concatenate(rows).astype(np.int64)
    return O.Dataset(np.array(indptr), idx, np.concatenate(vals), n, d, field_of[idx], F), rng.standard_normal(n)


def _ffm_fit(kind, win, W, Xo, y, k, P0, w0, b0, epochs, perms=None, **kw):
    from gpu_common import gpu_ffm
    _note_fit(win)
    with env(NFM_SEQ_WIN=win, NFM_SEQ_WIN_W=W):
        ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
        mk = nf.newSGD if kind == "sgd" else nf.newAdaGrad
        opt = mk(maxIter=epochs, verbose=0, tol=0, shuffle=False, mode="sequential", **kw)
        ctx = nf.default_context()
        ctx.timing_enable(True)
        ctx.timing_reset()
        opt.fit(to_gpu(Xo), y, ffm, perms=perms)
        windowed = ctx.timing_get("seq_window_deps")[0] > 0
        assert _fallbacks() == 0, "a window launch aborted"
        ctx.timing_enable(False)
        assert windowed == (int(win) != 0), "the %s kernel ran" % ("one-workgroup" if int(win) else "window")
        state = opt.get_state(ffm) if kind == "adagrad" else None
        return ffm.P.copy(), ffm.w.copy(), ffm.intercept, opt.it, list(opt.history), state


@pytest.mark.parametrize("kind,F,k,d,max_m,W", [
    ("sgd", 3, 2, 12, 6, 8), ("adagrad", 3, 2, 12, 6, 8),          # few features: every sample waits
    ("sgd", 16, 8, 64, 16, 16), ("adagrad", 16, 8, 64, 16, 64),    # cfg4's fields and factors
    ("sgd", 7, 20, 30, 7, 8), ("adagrad", 7, 20, 30, 7, 32),       # factors not a power of two
    ("sgd", 5, 4, 400, 20, 64), ("adagrad", 5, 4, 400, 20, 64),    # 20 entries: 210 chain terms, several entries per field
    ("sgd", 4, 64, 40, 8, 16), ("adagrad", 2, 33, 25, 10, 8),      # 64 factors per row (one row per instruction)
])
def test_field_aware_window_bitwise_and_oracle(kind, F, k, d, max_m, W):
    """field-aware models in the dependency window (win_worker_ffm): parameters, linear weights, intercept, AdaGrad state
    bit for bit those of the one-workgroup kernel, and the oracle's fit (O.ffm_*_fit, sgd_ffm.nim / adagrad_ffm.nim)"""
    from common import init_ffm
    Xo, y = _ffm_data(300, d, F, max_m, seed=F * 100 + k)
    P0, w0, b0 = init_ffm(d, F, k)
    perms = make_perms(Xo.n, 2)
    kw = dict(eta0=0.05) if kind == "sgd" else {}
    ref = _ffm_fit(kind, 0, W, Xo, y, k, P0, w0, b0, 2, perms, **kw)
    win = _ffm_fit(kind, 2, W, Xo, y, k, P0, w0, b0, 2, perms, **kw)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3]
    assert_close([h[1] for h in win[4]], [h[1] for h in ref[4]], *_tol(1e-12, 1e-15), "loss per epoch")
    assert_close([h[0] for h in win[4]], [h[0] for h in ref[4]], *_tol(1e-11, 1e-14), "viol per epoch")
    if kind == "adagrad":
        for g, h, name in zip(win[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
        Pf, wf, bf, *_ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0, O.adagrad_cfg(), 2, perms=perms)
    else:
        Pf, wf, bf, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.05), 2, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("F,k,n_feat", [(39, 4, 39 * 40), (30, 8, 30 * 25)])
def test_field_aware_rows_longer_than_a_mailbox_in_the_one_term_window(F, k, n_feat):
    """Rows of more than 22 entries -- 39 fields x one entry is the shape of click-through data -- are 780 chain terms: no
    mailbox of the term-by-term conductor holds them (that flavour sends them to the one-workgroup kernel).  The one-term window
    posts ONE term whatever the row length: SGD fits (its slots' rows + derivatives: 98 KB of LDS at F = 39, k = 4; AdaGrad's
    state rows beside them do not).  Same results as the one-workgroup kernel at the one-term tolerance, and the oracle's
    (sgd_ffm.nim:11-106)."""
    if not _ONE_TERM["variant"]:
        pytest.skip("the term-by-term flavour has no mailbox for rows this long")
    from common import init_ffm
    Xo, y = _ffm_data(2500, n_feat, F, F, seed=F, one_per_field=True)
    P0, w0, b0 = init_ffm(n_feat, F, k)
    perms = make_perms(Xo.n, 2)
    ref = _ffm_fit("sgd", 0, 64, Xo, y, k, P0, w0, b0, 2, perms, eta0=0.02)
    win = _ffm_fit("sgd", 2, 64, Xo, y, k, P0, w0, b0, 2, perms, eta0=0.02)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3]
    Pf, wf, bf, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.02), 2, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_field_aware_window_benchmark_shape(kind):
    """16 fields x one entry, k = 8 (BASELINE configs[3]'s row shape), 3000 samples over 1600 features, 64 workers"""
    from common import init_ffm
    Xo, y = _ffm_data(3000, 1600, 16, 16, seed=9, one_per_field=True)
    P0, w0, b0 = init_ffm(1600, 16, 8)
    ref = _ffm_fit(kind, 0, 64, Xo, y, 8, P0, w0, b0, 1)
    win = _ffm_fit(kind, 2, 64, Xo, y, 8, P0, w0, b0, 1)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2])


def _fmx_fit(kind, win, W, Xo, y, degree, fit_lower, k, P0, w0, b0, epochs, perms=None, **kw):
    _note_fit(win)
    with env(NFM_SEQ_WIN=win, NFM_SEQ_WIN_W=W):
        fm = gpu_fm("regression", degree, k, fit_lower, True, True, P0, w0, b0)
        mk = nf.newSGD if kind == "sgd" else nf.newAdaGrad
        opt = mk(maxIter=epochs, verbose=0, tol=0, shuffle=False, mode="sequential", **kw)
        ctx = nf.default_context()
        ctx.timing_enable(True)
        ctx.timing_reset()
        opt.fit(to_gpu(Xo), y, fm, perms=perms)
        windowed = ctx.timing_get("seq_window_deps")[0] > 0
        assert _fallbacks() == 0, "a window launch aborted"
        ctx.timing_enable(False)
        assert windowed == (int(win) != 0), "the %s kernel ran" % ("one-workgroup" if int(win) else "window")
        state = opt.get_state(fm) if kind == "adagrad" else None
        return fm.P.copy(), fm.w.copy(), fm.intercept, opt.it, list(opt.history), state


@pytest.mark.parametrize("kind,degree,fit_lower,k,d,W", [
    ("sgd", 3, "explicit", 8, 10, 8), ("adagrad", 3, "explicit", 8, 10, 16),     # two orders (degrees 3 and 2): cfg5's model
    ("sgd", 4, "explicit", 4, 40, 64), ("adagrad", 4, "explicit", 4, 40, 64),    # three orders
    ("sgd", 3, "none", 5, 30, 32), ("adagrad", 5, "none", 3, 30, 8),             # one order of degree >= 3
    ("sgd", 2, "explicit", 16, 300, 64),                                         # (degree 2: the general worker, for reference)
    ("sgd", 3, "explicit", 64, 30, 16), ("adagrad", 3, "none", 40, 30, 8),       # 64 factors per row
    ("sgd", 6, "explicit", 2, 60, 32),                                           # the highest degree: five orders
])
def test_higher_degree_window_bitwise_and_oracle(kind, degree, fit_lower, k, d, W):
    """several orders / degree >= 3 in the dependency window (win_worker_fmx): bit for bit the one-workgroup kernel's
    parameters and state, and the oracle's fit (computeAnova's recursion, sgd.nim:146-188)"""
    Xo = ragged_csr(400, d, seed=degree * 10 + k, max_m=min(d, 24))
    rng = np.random.default_rng(degree)
    y = rng.standard_normal(Xo.n)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.1)
    assert n_aug == 0
    perms = make_perms(Xo.n, 2)
    kw = dict(eta0=0.02) if kind == "sgd" else {}
    ref = _fmx_fit(kind, 0, W, Xo, y, degree, fit_lower, k, P0, w0, b0, 2, perms, **kw)
    win = _fmx_fit(kind, 2, W, Xo, y, degree, fit_lower, k, P0, w0, b0, 2, perms, **kw)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3]
    assert_close([h[1] for h in win[4]], [h[1] for h in ref[4]], *_tol(1e-12, 1e-15), "loss per epoch")
    assert_close([h[0] for h in win[4]], [h[0] for h in ref[4]], *_tol(1e-11, 1e-14), "viol per epoch")
    if kind == "adagrad":
        for g, h, name in zip(win[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, O.adagrad_cfg(), 2, 0, perms=perms)
    else:
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(eta0=0.02), 2, 0, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_higher_degree_window_long_rows(kind):
    """rows of up to 150 entries (the entries beyond the first 64 take the counter path, the forwarding area holds fewer
    hot entries than a row has), 20 features only: every sample waits"""
    Xo = ragged_csr(300, 150, seed=8, max_m=150)
    y = np.random.default_rng(8).standard_normal(Xo.n)
    P0, w0, b0, _ = init_fm(150, 3, 8, "explicit", True, scale=0.05)
    kw = dict(eta0=0.005) if kind == "sgd" else {}
    ref = _fmx_fit(kind, 0, 16, Xo, y, 3, "explicit", 8, P0, w0, b0, 2, None, **kw)
    win = _fmx_fit(kind, 2, 16, Xo, y, 3, "explicit", 8, P0, w0, b0, 2, None, **kw)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2])


def _distinct_rows(n, d, m, seed):
    """n rows of exactly m distinct sorted features, uniform over d (rows with a repeat are re-drawn)"""
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.integers(0, d, size=(n, m)), axis=1)
    while True:
        bad = np.nonzero((idx[:, 1:] == idx[:, :-1]).any(axis=1))[0]
        if len(bad) == 0:
            break
        idx[bad] = np.sort(rng.integers(0, d, size=(len(bad), m)), axis=1)
    return O.Dataset(np.arange(n + 1, dtype=np.int64) * m, idx.ravel().astype(np.int64), rng.uniform(-1, 1, n * m), n, d)


def _fit_checked(kind, win, Xo, y, task, k, P0, w0, b0, epochs, perms, W=64, **kw):
    """like fit(), and asserts WHICH kernel ran (the window builds a dependency table, the one-workgroup kernel does not)"""
    ctx = nf.default_context()
    ctx.timing_enable(True)
    ctx.timing_reset()
    try:
        out = fit(kind, win, W, Xo, y, task, k, P0, w0, b0, epochs, perms, **kw)
        windowed = ctx.timing_get("seq_window_deps")[0] > 0
    finally:
        ctx.timing_enable(False)
    assert windowed == (int(win) != 0), "the %s kernel ran" % ("one-workgroup" if int(win) else "window")
    return out


@pytest.mark.parametrize("kind,d,fit_intercept", itertools.product(["sgd", "adagrad"], [2_000, 200_000], [True, False]))
def test_k64_worker_at_the_headline_row_shape(kind, d, fit_intercept):
    """win_worker_k64 (selected for 64 factors and rows of up to 64 entries: the register-resident worker behind bench.py's
    headline `exact_order`) at ITS shape: 64 entries per row, k = 64, 20 000 samples, two permuted epochs.  d = 2 000: a sample
    shares a feature with one of its 63 predecessors almost surely (speculative gather re-read, hot-row and recipe
    forwarding on every sample); d = 200 000: 2 % per pair -- most samples run side by side, the dependent ones are near
    successors.  Bit for bit the one-workgroup kernel, and the oracle's fit (optimizer/sgd.nim:246-258,294-308,
    adagrad.nim:169-184) at rtol 1e-8.  fit_intercept = False: the window without a conductor (128 workers, each adds up
    its own sample's prediction: no scalar chain ties the samples, only their features do)."""
    n, m, k = 20_000, 64, 64
    fl = dict(fit_intercept=fit_intercept, W=64 if fit_intercept else (256 if d > 2_000 else 128))  # (256: a worker on every CU)
    Xo = _distinct_rows(n, d, m, seed=d + 1)
    rng = np.random.default_rng(d)
    y = np.sign(rng.standard_normal(n))
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.01, rng.standard_normal(d) * 0.01, 0.02
    perms = make_perms(n, 2)
    ref = _fit_checked(kind, 0, Xo, y, "classification", k, P0, w0, b0, 2, perms, loss="logistic", **fl)
    win = _fit_checked(kind, 2, Xo, y, "classification", k, P0, w0, b0, 2, perms, loss="logistic", **fl)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3] == 2 * n + 1
    assert fit_intercept or win[2] == b0
    assert np.isfinite(win[0]).all() and not np.array_equal(win[0], P0)
    assert_close([h[1] for h in win[4]], [h[1] for h in ref[4]], *_tol(1e-12, 1e-15), "loss per epoch")
    if kind == "adagrad":
        for g, h, name in zip(win[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, O.adagrad_cfg(loss="logistic", fit_intercept=fit_intercept), 2, 0, perms=perms)
    else:
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(loss="logistic", fit_intercept=fit_intercept), 2, 0, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("kind,d,fit_intercept", itertools.product(["sgd", "adagrad"], [2_000, 100_000], [True, False]))
def test_general_worker_at_cfg2_row_shape(kind, d, fit_intercept):
    """the LDS-resident worker at BASELINE configs[1]'s row shape (32 entries, k = 16), conflict-heavy and sparse, two
    permuted epochs: bit for bit the one-workgroup kernel, rtol 1e-8 the oracle (fit_intercept = False: without a conductor)"""
    n, m, k = 20_000, 32, 16
    fl = dict(fit_intercept=fit_intercept, W=64 if fit_intercept else 128)
    Xo = _distinct_rows(n, d, m, seed=d + 2)
    rng = np.random.default_rng(d + 3)
    y = np.sign(rng.standard_normal(n))
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.01, rng.standard_normal(d) * 0.01, -0.01
    perms = make_perms(n, 2)
    ref = _fit_checked(kind, 0, Xo, y, "classification", k, P0, w0, b0, 2, perms, loss="logistic", **fl)
    win = _fit_checked(kind, 2, Xo, y, "classification", k, P0, w0, b0, 2, perms, loss="logistic", **fl)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert same_b(win[2], ref[2]) and win[3] == ref[3]
    assert_close([h[1] for h in win[4]], [h[1] for h in ref[4]], *_tol(1e-12, 1e-15), "loss per epoch")
    if kind == "adagrad":
        for g, h, name in zip(win[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, O.adagrad_cfg(loss="logistic", fit_intercept=fit_intercept), 2, 0, perms=perms)
    else:
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(loss="logistic", fit_intercept=fit_intercept), 2, 0, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert abs(win[2] - bf) < 1e-9


@pytest.mark.parametrize("kind,degree,fit_lower,k,d,W", [
    ("sgd", 3, "explicit", 8, 12, 16), ("adagrad", 3, "explicit", 8, 12, 128),   # cfg5's model, every sample waits
    ("sgd", 4, "explicit", 4, 2000, 128), ("adagrad", 3, "none", 16, 3000, 128),  # mostly concurrent samples
    ("sgd", 6, "explicit", 2, 60, 32),
])
def test_higher_degree_window_without_a_conductor(kind, degree, fit_lower, k, d, W):
    """fitIntercept = false with several orders / degree >= 3 (win_worker_fmx): no conductor -- wavefront 0 of a worker adds up
    the prediction (the linear terms in storage order, then the orders' kernels: predictWithGrad, sgd.nim:193-201) and posts
    {dL, yhat} for the other wavefronts and the near successors.  Bit for bit the one-workgroup kernel, rtol 1e-8 the oracle."""
    n = 3000 if d >= 2000 else 400
    Xo = ragged_csr(n, d, seed=degree * 7 + k, max_m=min(d, 24))
    rng = np.random.default_rng(degree + 11)
    y = rng.standard_normal(Xo.n)
    P0, w0, _, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.1)
    b0 = 0.3  # (stays where it is)
    perms = make_perms(Xo.n, 2)
    kw = dict(eta0=0.02) if kind == "sgd" else {}

    def run(win):
        with env(NFM_SEQ_WIN=win, NFM_SEQ_WIN_W=W):
            fm = nf.newFactorizationMachine("regression", degree=degree, nComponents=k, fitLower=fit_lower, fitLinear=True,
                                            fitIntercept=False, warmStart=True)
            fm.set_params(P0, w0, b0)
            mk = nf.newSGD if kind == "sgd" else nf.newAdaGrad
            opt = mk(maxIter=2, verbose=0, tol=0, shuffle=False, mode="sequential", **kw)
            ctx = nf.default_context()
            ctx.timing_enable(True)
            ctx.timing_reset()
            opt.fit(to_gpu(Xo), y, fm, perms=perms)
            windowed = ctx.timing_get("seq_window_deps")[0] > 0
            assert _fallbacks() == 0, "a window launch aborted"
            ctx.timing_enable(False)
            assert windowed == (int(win) != 0)
            return fm.P.copy(), fm.w.copy(), fm.intercept, list(opt.history)

    ref, win = run(0), run(2)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert win[2] == ref[2] == b0
    assert_close([h[1] for h in win[3]], [h[1] for h in ref[3]], *_tol(1e-12, 1e-15), "loss per epoch")
    if kind == "adagrad":
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, O.adagrad_cfg(fit_intercept=False), 2, 0, perms=perms)
    else:
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(eta0=0.02, fit_intercept=False), 2, 0, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")
    assert bf == b0


@pytest.mark.parametrize("kind,F,k,d,max_m,W", [
    ("sgd", 3, 2, 12, 6, 16), ("adagrad", 3, 2, 12, 6, 128),          # few features: every sample waits
    ("sgd", 16, 8, 1600, 16, 128), ("adagrad", 16, 8, 1600, 16, 128),  # cfg4's fields and factors, mostly concurrent samples
    ("sgd", 5, 4, 400, 20, 64),                                       # 20 entries: 210 pair terms, several entries per field
])
def test_field_aware_window_without_a_conductor(kind, F, k, d, max_m, W):
    """fitIntercept = false, field-aware (win_worker_ffm): wavefront 0 of a worker adds up the prediction itself -- the linear
    terms, then one term per pair in the order of the reference's double loop (sgd_ffm.nim:13-27).  Bit for bit the
    one-workgroup kernel, rtol 1e-8 the oracle."""
    from common import init_ffm
    n = 2500 if d >= 1000 else 300
    Xo, y = _ffm_data(n, d, F, max_m, seed=F * 31 + k)
    P0, w0, _ = init_ffm(d, F, k)
    b0 = -0.2
    perms = make_perms(Xo.n, 2)
    kw = dict(eta0=0.05) if kind == "sgd" else {}

    def run(win):
        with env(NFM_SEQ_WIN=win, NFM_SEQ_WIN_W=W):
            ffm = nf.newFieldAwareFactorizationMachine("regression", nComponents=k, fitLinear=True, fitIntercept=False, warmStart=True)
            ffm.set_params(P0, w0, b0)
            mk = nf.newSGD if kind == "sgd" else nf.newAdaGrad
            opt = mk(maxIter=2, verbose=0, tol=0, shuffle=False, mode="sequential", **kw)
            ctx = nf.default_context()
            ctx.timing_enable(True)
            ctx.timing_reset()
            opt.fit(to_gpu(Xo), y, ffm, perms=perms)
            windowed = ctx.timing_get("seq_window_deps")[0] > 0
            assert _fallbacks() == 0, "a window launch aborted"
            ctx.timing_enable(False)
            assert windowed == (int(win) != 0)
            return ffm.P.copy(), ffm.w.copy(), ffm.intercept, list(opt.history)

    ref, win = run(0), run(2)
    same_bits(win[0], ref[0], "P")
    same_bits(win[1], ref[1], "w")
    assert win[2] == ref[2] == b0
    assert_close([h[1] for h in win[3]], [h[1] for h in ref[3]], *_tol(1e-12, 1e-15), "loss per epoch")
    if kind == "adagrad":
        Pf, wf, bf, *_ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0, O.adagrad_cfg(fit_intercept=False), 2, perms=perms)
    else:
        Pf, wf, bf, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.05, fit_intercept=False), 2, perms=perms)
    assert_close(win[0], Pf, 1e-8, 1e-11, "P vs oracle")
    assert_close(win[1], wf, 1e-8, 1e-11, "w vs oracle")


@pytest.mark.parametrize("kind,shape", [("sgd", "k64"), ("sgd", "general"), ("adagrad", "k64")])
def test_one_term_window_is_bitwise_reproducible_with_the_sample_by_sample_chain(kind, shape):
    """The one-term flavour's arithmetic is fixed by POSITIONS (which row is treated as affine in its writer's dL never
    depends on what happens to be finished when a worker looks), so with the conductor's sample-by-sample chain
    (NFM_SEQ_WIN_PAR=0) two runs give the same bits whatever the timing.  The chunk-parallel chain (the default for SGD) solves
    the chunk of samples that happen to be ready: its results agree with these to rounding (rtol 1e-8 here), run to run as well."""
    if not _ONE_TERM["variant"]:
        pytest.skip("the term-by-term flavour is bit-equal to the one-workgroup kernel (every other test of this file)")
    n, d, m, k = (6000, 3000, 64, 64) if shape == "k64" else (6000, 1500, 20, 16)
    Xo = _distinct_rows(n, d, m, seed=77)
    rng = np.random.default_rng(78)
    y = np.sign(rng.standard_normal(n))
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.01, rng.standard_normal(d) * 0.01, 0.02
    perms = make_perms(n, 2)
    runs = []
    for par in ("0", "0", "6", "6"):
        with env(NFM_SEQ_WIN_PAR=par):
            runs.append(fit(kind, 2, 128 if shape == "k64" else 64, Xo, y, "classification", k, P0, w0, b0, 2, perms, loss="logistic"))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2], \
        "two runs with the sample-by-sample chain differ"
    for r_ in runs[2:]:
        assert_close(r_[0], runs[0][0], 1e-8, 1e-11, "P, chunk-parallel chain")
        assert_close(r_[1], runs[0][1], 1e-8, 1e-11, "w, chunk-parallel chain")
        assert abs(r_[2] - runs[0][2]) <= 1e-11 + 1e-8 * abs(runs[0][2])


def test_abort_paths_in_the_test_hooks_build():
    """The launch-abort paths (a worker that never shows up: snapshot put back, the one-workgroup kernel re-runs the call; two
    aborts: the optimizer stops asking; no memory for the snapshot: the one-workgroup kernel) need hooks inside the library.
    The product library has none: they are compiled into libnimfm_hip_testhooks.so only (-DNFM_TEST_HOOKS, same objects
    otherwise).  ONE child process runs tests/gpu_hooks_cases.py against that build."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "nimfm_amd", "lib", "libnimfm_hip_testhooks.so")
    assert os.path.exists(lib), "build it: make -C nimfm_amd/csrc (the `all` target builds both libraries)"
    if _ONE_TERM["variant"] is False:  # (once, not per flavour: the child runs both)
        return
    envc = dict(os.environ, NIMFM_HIP_LIB=lib, NFM_TEST_HOOKS_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(root, "tests", "gpu_hooks_cases.py")],
                       cwd=root, env=envc, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-4000:], r.stderr[-2000:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("kind,k,m", [("sgd", 100, 40), ("sgd", 128, 64), ("sgd", 65, 9), ("adagrad", 100, 20), ("adagrad", 128, 30),
                                      ("sgd", 200, 24), ("sgd", 300, 10), ("adagrad", 160, 12)])
def test_65_to_128_factors_read_as_two_blocks_of_64(kind, k, m, window_flavour):
    """A degree-2 FM of 65 ... 128 factors is one block of 128-double rows: the window's workers take rows of at most 64
    factors, so before round 5 such a model ran its exact order one sample at a time.  The same table read as feature-major
    blocks of 64 (ModelView::row with bs = 1, rs = 2, seqwin.hip::seq_window_view) is a two-block model of the same degree,
    which the several-orders worker takes as it is -- in the one-term window (the exact flavour keeps the one-workgroup kernel:
    its factor sum is one ascending pass).  Held to the one-workgroup kernel and to the oracle; which kernel ran is asked.
    More than 128 factors (one order): the kc blocks of 128 lie feature-major (api.hip: wide_rows), a feature's row is kc * 128
    contiguous doubles -- 2 kc blocks of 64 to the window as long as the worker's LDS holds the sample's rows, ONE row to the
    one-sample-in-flight kernel."""
    n, d = 2500, 700
    Xo = random_csr(n, d, m, seed=k)
    rng = np.random.default_rng(k + 1)
    y = rng.standard_normal(n)
    perms = make_perms(n, 2)
    ctx = nf.default_context()
    before = ctx.timing_get("seq_window_launch")[0]
    if kind == "sgd":
        win, (P0, w0, b0) = check_pair(kind, Xo, y, "regression", k, 64, epochs=2, perms=perms)
    else:
        # AdaGrad's 1 / sqrt(g_norm) (g_norm starts at eps = 1e-10) amplifies the few-ulp differences of a prediction whose
        # hundred factors are summed block by block instead of in one pass (tools/seqwin_soak.py saw 1e-10 ... 7e-7 per parameter
        # for the one-term window itself): held at the reference's own fast-against-slow tolerance (tests/utils.nim:82-105)
        P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.1 / np.sqrt(k), rng.standard_normal(d) * 0.01, 0.05
        ref = fit(kind, 0, 64, Xo, y, "regression", k, P0, w0, b0, 2, perms)
        win = fit(kind, 2, 64, Xo, y, "regression", k, P0, w0, b0, 2, perms)
        for a_, b_, name in [(win[0], ref[0], "P"), (win[1], ref[1], "w")] + list(zip(win[5][:4], ref[5][:4], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w"])):
            if window_flavour == "one_term":
                assert_close(a_, b_, 1e-6, 1e-9, name)
            else:
                same_bits(a_, b_, name)
        assert abs(win[2] - ref[2]) < 1e-9
    launched = ctx.timing_get("seq_window_launch")[0] - before
    if window_flavour == "one_term":
        assert launched >= 2, "the window did not run for %d factors" % k
    else:
        assert launched == 0
    if kind == "sgd":
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, y, 2, P0, w0, b0, O.sgd_cfg(), 2, 0, perms=perms)
    else:
        Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, y, 2, P0, w0, b0, O.adagrad_cfg(), 2, 0, perms=perms)
    tol = (1e-8, 1e-11) if kind == "sgd" or window_flavour != "one_term" else (1e-6, 1e-9)
    if kind == "sgd" and k == 100:  # without an intercept: no conductor, the worker sums the prediction itself -- the same view
        before = ctx.timing_get("seq_window_launch")[0]
        refn = fit(kind, 0, 64, Xo, y, "regression", k, P0, w0, 0.0, 2, perms, fit_intercept=False)
        winn = fit(kind, 2, 64, Xo, y, "regression", k, P0, w0, 0.0, 2, perms, fit_intercept=False)
        assert (ctx.timing_get("seq_window_launch")[0] - before >= 2) == (window_flavour == "one_term")
        assert_close(winn[0], refn[0], 1e-8, 1e-11, "P, fitIntercept = false")
        assert_close(winn[1], refn[1], 1e-8, 1e-11, "w, fitIntercept = false")
        assert winn[2] == 0.0
    assert_close(win[0], Pf, *tol, "P vs oracle")
    assert_close(win[1], wf, *tol, "w vs oracle")
    assert abs(win[2] - bf) < (1e-9 if kind == "sgd" else 1e-7)
