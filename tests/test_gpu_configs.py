"""-m gpu: the BASELINE.json configurations at their OWN shapes (everything that decides which kernel instantiation
runs and how a batch's touches collide: d, nnz/row, k, degree, number of fields, batch size), with n cut to what the
CPU restatement of the mini-batch rule finishes in seconds, plus the statistical clause of SURVEY 8(e): the mini-batch
rule against the exact sequential order on a planted problem.

  cfg5  degree 3, fitLower = explicit (two orders), k = 8, 32 nnz/row, d = 1e5, mini-batch 32768, SGD Squared
        (reference: tests/test_sgd.nim:92-151 grids at degree 2..4; kernels.nim:54-58 the degree >= 3 recursion)
  cfg4  FieldAwareFactorizationMachine, 16 fields, one nnz per field, k = 8, d = 1e5, mini-batch 32768, AdaGrad
        (reference: optimizer/sgd_ffm.nim:11-30, adagrad_ffm.nim:11-66)
  cfg3  d = 1e6, 64 nnz/row, k = 64, mini-batch 8192, AdaGrad, Squared and Logistic, 20 full batches
        (reference: optimizer/adagrad.nim:87-134)
"""
import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, random_csr
from gpu_common import gpu_ffm, gpu_fm, to_gpu
from test_gpu_fullsize import big_csr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cap", [1.0, 16.0])
def test_cfg5_shape_degree3_sgd_vs_mb_oracle(cap):
    """(cap = 16: the touch cap bench.py trains this config with, nfm_opt_set_touch_cap)"""
    n, d, m, k, B = 131_072 + 500, 100_000, 32, 8, 32768  # four full batches and a ragged tail
    Xo = big_csr(n, d, m, 45)
    rng = np.random.default_rng(6)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((2, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    cfg = O.sgd_cfg(loss="squared")
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    hist = []
    for _ in range(2):
        b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 3, P, w, b, cfg, B, it=it, touch_cap=cap)
        hist.append((vs, ls / n))
    X = to_gpu(Xo)
    fm = gpu_fm("regression", 3, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, loss="squared", mode="minibatch", batch=B, touchCap=cap)
    sgd.fit(X, y, fm)
    assert sgd.it == it == 2 * n + 1
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close([h[1] for h in sgd.history], [h[1] for h in hist], 1e-11, 0, "mean loss")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 0, "viol")
    assert_close(fm.decisionFunction(X), O.fm_decision_function(Xo, 3, P, w, b), 1e-10, 1e-13, "decision")
    # a fresh permutation (the reference's default shuffle = true): the plan is rebuilt for it
    perm = np.random.default_rng(3).permutation(n).astype(np.int64)
    P, w = P0.copy(), w0.copy()
    b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 3, P, w, 0.0, cfg, B, perm=perm, it=1, touch_cap=cap)
    fm = gpu_fm("regression", 3, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, loss="squared", mode="minibatch", batch=B, touchCap=cap)
    sgd.fit(X, y, fm, perms=perm[None, :])
    assert_close(fm.P, P, 1e-9, 1e-13, "P, permuted")
    assert_close(sgd.history[0][0], vs, 1e-9, 0, "viol, permuted")


@pytest.mark.parametrize("gamma,B", [(0.0, 32768), (0.1, 32768), (0.1, 65536)])
def test_cfg4_shape_ffm_adagrad_vs_mb_oracle(gamma, B):
    """BASELINE configs[3] at the batches bench.py quotes it at since round 5 (32768, then 65536: the dense regime with the refresh
    pass); gamma = 0.1: WITH the batch's gradient cross products in g_norm (nfm_opt_set_ada_cross), the rule bench.py trains it with"""
    n, d, F, k = 6 * 32768 + 4097, 100_000, 16, 8  # it == 1 singleton + six (three) full batches + a tail
    rng = np.random.default_rng(8)
    per = d // F
    idx = rng.integers(0, per, size=(n, F)) + np.arange(F) * per  # field f owns [f d/F, (f+1) d/F) (tests/utils.nim:66-68)
    val = rng.uniform(-1, 1, size=(n, F))
    Xo = O.Dataset(np.arange(n + 1, dtype=np.int64) * F, idx.ravel(), val.ravel(), n, d, fields=np.tile(np.arange(F), n),
                   n_fields=F)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((F, d, k)) * 0.05, np.zeros(d)
    cfg = O.adagrad_cfg(loss="squared")
    P, w = P0.copy(), w0.copy()
    st = O.AdaState(F, d, k, d)
    b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, 0.0, cfg, B, st, it=1, ada_cross=gamma)
    b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
    X = to_gpu(Xo)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, loss="squared", mode="minibatch", batch=B, adaCross=gamma)
    ada.fit(X, y, ffm)
    assert ada.it == it == n + 1
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, 1e-9, 1e-13, "w")
    assert_close(ffm.P, P, 1e-9, 1e-13, "P")
    assert_close(ada.history[0][0], vs, 1e-9, 0, "viol")
    assert_close(ada.history[0][1], ls / n, 1e-11, 0, "mean loss")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(ffm)
    assert_close(gs, st.gsum_P, 1e-9, 1e-13, "g_sum")
    assert_close(gn, st.gnorm_P, 1e-9, 1e-16, "g_norm")
    assert_close(ffm.decisionFunction(X), O.ffm_decision_function(Xo, P, w, b), 1e-10, 1e-13, "decision")


def _cfg4_data(n, d, F, seed):
    rng = np.random.default_rng(seed)
    per = d // F
    idx = rng.integers(0, per, size=(n, F)) + np.arange(F) * per  # field f owns [f d/F, (f+1) d/F) (tests/utils.nim:66-68)
    val = rng.uniform(-1, 1, size=(n, F))
    Xo = O.Dataset(np.arange(n + 1, dtype=np.int64) * F, idx.ravel(), val.ravel(), n, d, fields=np.tile(np.arange(F), n),
                   n_fields=F)
    return Xo, rng


@pytest.mark.parametrize("kind", ["adagrad", "sgd_cap16"])
def test_cfg4_shape_at_the_bench_batch_2048(kind):
    """BASELINE configs[3] at the batch bench.py QUOTES it at (2048: the sparse regime -- 1.17 touches per unique feature,
    k_ffm_row_phase_lds + k_ffm_col_phase WITHOUT the per-batch refresh pass; the B = 32768 test above runs the dense regime
    with the refresh): F = 16, one entry per field, d = 1e5, k = 8, 19 full batches + the it == 1 singleton + a ragged tail,
    two epochs -- the second over a fresh permutation -- against the CPU statement of the mini-batch rule
    (optimizer/adagrad_ffm.nim:11-66, sgd_ffm.nim:43 through oracle/nimfm_mb.c) at rtol 1e-9, decisionFunction at 1e-10."""
    n, d, F, k, B = 1 + 19 * 2048 + 777, 100_000, 16, 8, 2048
    Xo, rng = _cfg4_data(n, d, F, 18)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((F, d, k)) * 0.05, rng.standard_normal(d) * 0.01
    perm = np.random.default_rng(5).permutation(n).astype(np.int64)
    perms = np.stack([np.arange(n, dtype=np.int64), perm])
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    hist = []
    if kind == "adagrad":
        cfg = O.adagrad_cfg(loss="squared")
        st = O.AdaState(F, d, k, d)
        for e in range(2):
            b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, B, st, perm=perms[e], it=it)
            hist.append((vs, ls / n))
        b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
        opt = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, loss="squared", mode="minibatch", batch=B)
    else:
        cfg = O.sgd_cfg(loss="squared", eta0=0.02)
        for e in range(2):
            b, it, ls, vs = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, cfg, B, perm=perms[e], it=it, touch_cap=16.0)
            hist.append((vs, ls / n))
        opt = nf.newSGD(maxIter=2, verbose=0, tol=0, loss="squared", eta0=0.02, mode="minibatch", batch=B, touchCap=16.0)
    X = to_gpu(Xo)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, 0.0)
    ctx = nf.default_context()
    ctx.timing_enable(True)
    ctx.timing_reset()
    try:
        opt.fit(X, y, ffm, perms=perms)
        refresh, rows, cols = (ctx.timing_get(name)[0] for name in ("refresh", "row_phase", "col_phase"))
    finally:
        ctx.timing_enable(False)
    # which path ran: the sparse regime's (no refresh pass), one row + one column launch per mini-batch and epoch
    n_batches = 1 + 19 + 1 if kind == "adagrad" else 19 + 1  # (AdaGrad: the very first sample is a batch of its own, adagrad.nim:171)
    assert refresh == 0, "the per-batch refresh pass ran: this is not the path bench.py quotes cfg4 on"
    assert rows >= 2 * n_batches - 1 and cols >= 2 * n_batches - 1, (rows, cols)
    assert opt.it == it == 2 * n + 1
    assert abs(ffm.intercept - b) < 1e-11
    assert_close(ffm.w, w, 1e-9, 1e-13, "w")
    assert_close(ffm.P, P, 1e-9, 1e-13, "P")
    assert_close([h[0] for h in opt.history], [h[0] for h in hist], 1e-9, 0, "viol")
    assert_close([h[1] for h in opt.history], [h[1] for h in hist], 1e-11, 0, "mean loss")
    if kind == "adagrad":
        gs, gn, gsw, gnw, gsb, gnb = opt.get_state(ffm)
        assert_close(gs, st.gsum_P, 1e-9, 1e-13, "g_sum")
        assert_close(gn, st.gnorm_P, 1e-9, 1e-16, "g_norm")
        assert_close(gsw, st.gsum_w, 1e-9, 1e-13, "g_sum.w")
    assert_close(ffm.decisionFunction(X), O.ffm_decision_function(Xo, P, w, b), 1e-10, 1e-13, "decision")


@pytest.mark.parametrize("loss", ["squared", "logistic"])
def test_cfg3_shape_adagrad_20_batches(loss):
    n, d, m, k, B = 20 * 8192 + 1, 1_000_000, 64, 64, 8192  # the it == 1 singleton + twenty full batches
    Xo = big_csr(n, d, m, 46)
    rng = np.random.default_rng(4)
    y = np.sign(rng.standard_normal(n)) if loss == "logistic" else rng.standard_normal(n)
    task = "classification" if loss == "logistic" else "regression"
    P0, w0 = rng.standard_normal((1, k, d)) * 0.01, np.zeros(d)
    cfg = O.adagrad_cfg(loss=loss)
    P, w = P0.copy(), w0.copy()
    st = O.AdaState(1, d, k, d)
    b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, 0.0, cfg, B, st, it=1)
    # a second epoch over the first five batches: rows whose state is no longer at its initial value
    n2 = 5 * B
    b, it, ls2, vs2 = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, B, st, begin=0, end=n2, it=it)
    b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    X = to_gpu(Xo)
    fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, loss=loss, mode="minibatch", batch=B)
    X.set_targets(y)
    ada._handle(fm, X.ctx, "minibatch")
    ls_g, vs_g = ada._epoch(X, None, 0, n)
    ada.it += n
    ls2_g, vs2_g = ada._epoch(X, None, 0, n2)
    ada.it += n2
    ada._finalize_into(fm)
    assert ada.it == it
    assert_close([ls_g, ls2_g], [ls, ls2], 1e-11, 0, "loss sums")
    assert_close([vs_g, vs2_g], [vs, vs2], 1e-9, 0, "viol")
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(fm)
    assert_close(gs, st.gsum_P, 1e-9, 1e-14, "g_sum")
    assert_close(gn, st.gnorm_P, 1e-9, 1e-18, "g_norm")


# ---- "final-loss/RMSE within tolerance of the exact-sequential run" ----
# The mini-batch rule makes ONE averaged step per coordinate and batch: a coordinate touched
# c times in a batch advances once where the sequential order advances c times.  The clause is therefore stated at
# EQUAL step size with the number of epochs scaled by the mean touch count per touched coordinate,
#   c = lambda / (1 - exp(-lambda)),  lambda = batch * nnz_per_row / d,
# and asserts that the mini-batch run is then at least as good as the sequential run on held-out data
# (regression: RMSE <= 1.05 x sequential; classification: accuracy >= sequential - 0.02).
def _planted(task):
    n, nt, d, m, k = 20_000, 10_000, 1_000, 16, 8
    full = random_csr(n + nt, d, m, seed=5)
    rng = np.random.default_rng(9)
    Pt, wt = rng.standard_normal((1, k, d)) * 0.3, rng.standard_normal(d) * 0.3
    yfull = O.fm_decision_function(full, 2, Pt, wt, 0.1) + 0.1 * rng.standard_normal(n + nt)
    if task == "classification":
        yfull = np.sign(yfull)

    def sub(lo, hi):
        a, b = full.indptr[lo], full.indptr[hi]
        return O.Dataset(full.indptr[lo:hi + 1] - a, full.indices[a:b], full.data[a:b], hi - lo, d)

    return sub(0, n), sub(n, n + nt), yfull[:n], yfull[n:], n, d, m, k


@pytest.mark.parametrize("task,batch", [("regression", 64), ("regression", 256), ("classification", 256)])
def test_minibatch_rule_statistical_parity_with_sequential(task, batch):
    Xtr, Xte, ytr, yte, n, d, m, k = _planted(task)
    loss = "squared" if task == "regression" else "logistic"
    E, eta0 = 10, 0.05
    P0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01
    cfg = O.sgd_cfg(eta0=eta0, loss=loss, alpha=1e-5, beta=1e-5)
    perms = np.stack([np.random.default_rng(100 + e).permutation(n) for e in range(E)]).astype(np.int64)
    Ps, ws, bs, *_ = O.fm_sgd_fit(Xtr, ytr, 2, P0, np.zeros(d), 0.0, cfg, E, perms=perms)  # the reference's order
    lam = batch * m / d
    Emb = int(np.ceil(E * lam / (1 - np.exp(-lam))))
    perms_mb = np.stack([np.random.default_rng(100 + e).permutation(n) for e in range(Emb)]).astype(np.int64)
    fm = gpu_fm(task, 2, k, "explicit", True, True, P0, np.zeros(d), 0.0)
    sgd = nf.newSGD(maxIter=Emb, eta0=eta0, alpha=1e-5, beta=1e-5, loss=loss, verbose=0, tol=0, mode="minibatch", batch=batch)
    sgd.fit(to_gpu(Xtr), ytr, fm, perms=perms_mb)
    dec_s = O.fm_decision_function(Xte, 2, Ps, ws, bs)
    dec_m = fm.decisionFunction(to_gpu(Xte))
    if task == "regression":
        rs, rm = np.sqrt(np.mean((dec_s - yte) ** 2)), np.sqrt(np.mean((dec_m - yte) ** 2))
        assert rs < 0.5 * yte.std(), rs  # the sequential run has learnt the planted model
        assert rm <= 1.05 * rs, (rm, rs, Emb)
    else:
        as_, am = np.mean(np.sign(dec_s) == yte), np.mean(np.sign(dec_m) == yte)
        assert as_ > 0.65, as_
        assert am >= as_ - 0.02, (am, as_, Emb)
    assert sgd.history[-1][1] < sgd.history[0][1]


def test_plan_cache_is_keyed_by_dataset_identity_not_address():
    """a dataset destroyed and replaced by another of the same shape (very likely at the same heap address)
    must not be served the old dataset's batch plan."""
    n, d, m, k, B = 3000, 200, 8, 8, 256
    rng = np.random.default_rng(12)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    for seed in (1, 2, 3):  # same n, nnz, begin, end, batch every time; different structure
        Xo = random_csr(n, d, m, seed=seed)
        X = to_gpu(Xo)
        fm.set_params(P0, w0, 0.0)
        sgd.it = 1  # a warm-started model keeps the optimizer's step count (sgd.nim:288-289)
        sgd.fit(X, y, fm)
        P, w = P0.copy(), w0.copy()
        b, *_ = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, 0.0, O.sgd_cfg(), B, it=1)
        assert_close(fm.P, P, 1e-9, 1e-13, "P, dataset %d" % seed)
        g_want = O.fm_predict_all_with_grad(Xo, y, 2, P, w, b, "squared")
        g_got = nf.predictAllWithGrad(X, y, fm)
        assert_close(g_got[2]["P"], g_want[2], 1e-8, 1e-15, "grad P, dataset %d" % seed)
        del X


def test_optimizer_refuses_a_destroyed_model():
    """the optimizer's model is looked up by id, not trusted by address."""
    import ctypes as C

    from nimfm_amd import _capi as capi
    n, d, m, k = 500, 50, 4, 4
    Xo = random_csr(n, d, m, seed=1)
    X = to_gpu(Xo)
    X.set_targets(np.zeros(n))
    fm = gpu_fm("regression", 2, k, "explicit", True, True, np.zeros((1, k, d)), np.zeros(d), 0.0)
    ada = nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=64)
    ada._handle(fm, X.ctx, "minibatch")
    ada._epoch(X, None, 0, n)
    old = ada._h
    # a model of ANOTHER size takes the old one's place (set_params with a different nFeatures releases the handle)
    fm.set_params(np.zeros((1, k, 4 * d)), np.zeros(4 * d), 0.0)
    fm._push(X.ctx)
    ls, vs = C.c_double(), C.c_double()
    assert capi.lib().nfm_opt_epoch(old, X.h, None, 0, n, C.byref(ls), C.byref(vs)) == capi.ERR_INVALID
    assert b"destroyed" in capi.lib().nfm_last_error()
    assert capi.lib().nfm_opt_finalize(old) == capi.ERR_INVALID
    # the host mirror notices the new generation and builds a fresh optimizer
    X4 = to_gpu(random_csr(n, 4 * d, m, seed=2))
    X4.set_targets(np.zeros(n))
    h2 = ada._handle(fm, X4.ctx, "minibatch")
    assert ada._h is h2 and ada._epoch(X4, None, 0, n) == (0.0, 0.0)


def test_repeated_ids_in_a_row_predict_but_do_not_train():
    """A column id repeated inside a row (dataset.nim:597-612 reads it without complaint): decisionFunction takes it as the
    reference does -- every entry is one more term of the ANOVA recursion (kernels.nim:46-64; field-aware: pairs of equal ids
    are skipped, field_aware_factorization_machine.nim:66-76) -- while the training kernels need distinct ids per row (what
    the reference's step computes for a repeat is an accident of its lazy scaling, include/nimfm_hip.h): fit refuses, naming
    the first such row.  Any storage order is fine."""
    indptr = np.array([0, 3, 6, 8], dtype=np.int64)
    ok = np.array([5, 1, 3, 0, 2, 9, 7, 4], dtype=np.int64)  # unsorted rows, all distinct
    nf.newCSRDataset(np.ones(8), ok, indptr, 3, 10)
    bad = ok.copy()
    bad[5] = 0  # row 1 = [0, 2, 0]
    vals = np.array([0.5, -1.0, 2.0, 1.5, 0.25, -0.75, 1.0, 3.0])
    rng = np.random.default_rng(3)
    Xo = O.Dataset(indptr, bad, vals, 3, 10)
    X = nf.newCSRDataset(vals, bad, indptr, 3, 10)
    for degree, fit_lower in [(2, "explicit"), (3, "explicit"), (4, "none")]:
        no = O.n_orders(degree, fit_lower)
        P0, w0 = rng.standard_normal((no, 5, 10)) * 0.3, rng.standard_normal(10) * 0.2
        fm = gpu_fm("regression", degree, 5, fit_lower, True, True, P0, w0, 0.1)
        assert_close(fm.decisionFunction(X), O.fm_decision_function(Xo, degree, P0, w0, 0.1, 0), 1e-12, 1e-14, "degree %d" % degree)
    y = np.array([1.0, -1.0, 0.5])
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0)
    with pytest.raises(nf.NfmError, match="row 1"):
        sgd.fit(X, y, fm)
    with pytest.raises(nf.NfmError, match="distinct"):
        nf.newSGD(maxIter=1, verbose=0, tol=0, mode="minibatch", batch=2).fit(X, y, fm)
    # field-aware: ids 3 and 3 in row 0 (fields 0 and 1)
    fld = np.array([0, 1, 2, 0, 1, 2, 0, 1], dtype=np.int64)
    idx = np.array([3, 3, 5, 0, 2, 9, 7, 4], dtype=np.int64)
    Xfo = O.Dataset(indptr, idx, vals, 3, 10, fields=fld, n_fields=3)
    Xf = nf.newCSRFieldDataset(vals, idx, indptr, fld, 3, 10, 3)
    Pf, wf = rng.standard_normal((3, 10, 4)) * 0.3, rng.standard_normal(10) * 0.2
    ffm = gpu_ffm("regression", 4, True, True, Pf, wf, -0.2)
    assert_close(ffm.decisionFunction(Xf), O.ffm_decision_function(Xfo, Pf, wf, -0.2), 1e-12, 1e-14, "field-aware")
    # the text loaders accept such a file too; sorted rows with an adjacent repeat
    Xt, yt = nf.parseText(b"1 3:1.0 5:2.0 3:0.5\n-1 1:1\n")
    with pytest.raises(nf.NfmError, match="distinct"):
        nf.newSGD(maxIter=1, verbose=0, tol=0).fit(Xt, yt, gpu_fm("regression", 2, 2, "explicit", True, True, np.zeros((1, 2, Xt.nFeatures)),
                                                                 np.zeros(Xt.nFeatures), 0.0))
    nf.newCSRDataset(np.ones(4), np.array([1, 1, 2, 3]), np.array([0, 2, 4]), 2, 5)
