"""-m gpu: mini-batch proximal SGD through the C ABI (nfm_mbpsgd_create / nfm_opt_epoch)
against the CPU restatement of optimizer/minibatch_psgd.nim (oracle/nimfm_psgd.c).

Tolerances: the device sums a mini-batch's gradient per coordinate in sample order like the reference, but forms
dloss / miniBatchSize, the row norms and the coupled thresholds with a different association (tree sums, fixed-point
threshold instead of randomised pivoting) -- 1e-9 relative on the parameters after a few outer iterations; the
north-star bound is 1e-6 relative."""
import itertools

import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_fm, make_fm_dataset, random_csr
from gpu_common import gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-12
REGS = {"l1": nf.newL1, "l21": nf.newL21, "squaredl12": nf.newSquaredL12, "squaredl21": nf.newSquaredL21}


def make_stream(n, need_total, seed):
    """indices[ii] as the reference's inner loops consume it: permutations back to back (wrap + reshuffle)"""
    rng = np.random.default_rng(seed)
    out = []
    while sum(len(o) for o in out) < need_total:
        out.append(rng.permutation(n))
    return np.concatenate(out)[:need_total].astype(np.int64)


def run_oracle(Xo, y, degree, P0, w0, b0, cfg, stream, B, inner, outer, n_aug):
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    losses = []
    need = B * inner
    for t in range(outer):
        b, it, ls = O.fm_mbpsgd_epoch(Xo, y, degree, P, w, b, cfg, stream[t * need:(t + 1) * need], B, n_aug, it=it, seed=t + 1)
        losses.append(ls / need)
    return P, w, b, it, losses


def check(Xo, y, task, degree, fit_lower, k, P0, w0, b0, n_aug, B, outer, reg, transpose=None, loss="squared",
          fit_linear=True, fit_intercept=True, scheduling="optimal", gamma=0.02, eta0=0.2, rtol=RTOL, atol=ATOL):
    n = Xo.n
    inner = (n - 1) // B + 1
    stream = make_stream(n, B * inner * outer, 9)
    kw = {} if transpose is None else {"transpose": transpose}
    cfg = O.psgd_cfg(eta0=eta0, gamma=gamma, beta=1e-2, alpha=1e-2, alpha0=1e-2, loss=loss, reg=reg, scheduling=scheduling,
                     fit_linear=fit_linear, fit_intercept=fit_intercept, **kw)
    P, w, b, it, losses = run_oracle(Xo, y, degree, P0, w0, b0, cfg, stream, B, inner, outer, n_aug)
    fm = gpu_fm(task, degree, k, fit_lower, fit_linear, fit_intercept, P0, w0, b0)
    opt = nf.newMBPSGD(maxIter=outer, eta0=eta0, alpha0=1e-2, alpha=1e-2, beta=1e-2, gamma=gamma, loss=loss,
                       reg=REGS[reg](**kw), miniBatchSize=B, scheduling=scheduling, verbose=0, tol=-1.0)
    opt.it = 1  # gpu_fm warm-starts (injected parameters); a fresh fit sets it = 1 (minibatch_psgd.nim:153-154)
    opt.fit(to_gpu(Xo), y, fm, stream=stream)
    assert opt.it == it
    assert_close([h[1] for h in opt.history], losses, 1e-10, 1e-13, "running loss")
    assert abs(fm.intercept - b) < 1e-10
    assert_close(fm.w, w, rtol, atol, "w")
    assert_close(fm.P, P, rtol, atol, "P")
    return fm, P


@pytest.mark.parametrize("reg,degree,fit_lower", [
    ("l1", 2, "explicit"), ("l21", 2, "explicit"), ("squaredl12", 2, "explicit"), ("squaredl21", 2, "explicit"),
    ("l1", 3, "explicit"), ("l21", 3, "augment"), ("l1", 4, "none"), ("l21", 2, "none"), ("l1", 2, "augment")])
def test_mbpsgd_vs_oracle(reg, degree, fit_lower):
    n, d, k, B = 83, 9, 4, 16  # 6 mini-batches of 16 = 96 indices per outer iteration: the stream wraps every time
    Xo, Xd, y = make_fm_dataset(n, d, degree, k, 42, fit_lower, threshold=0.3)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.3)
    fm, P = check(Xo, y, "regression", degree, fit_lower, k, P0, w0, 0.1, n_aug, B, 3, reg)
    if reg != "l21" or degree == 2:  # the penalty bites: exact zeros where the oracle has them
        assert np.array_equal(fm.P == 0.0, P == 0.0)


def test_degree_one_model():
    """degree = 1: no interaction block at all (nOrders = 0) -- only the linear term and the intercept are stepped"""
    n, d, B = 90, 12, 16
    Xo, Xd, y = make_fm_dataset(n, d, 2, 2, 7, "explicit", threshold=0.4)
    P0 = np.zeros((0, 3, d))
    w0 = np.random.default_rng(1).normal(size=d) * 0.1
    inner = (n - 1) // B + 1
    stream = make_stream(n, B * inner * 2, 2)
    cfg = O.psgd_cfg(eta0=0.1, reg="l1", alpha=1e-2, alpha0=1e-2)
    P, w, b, it, losses = run_oracle(Xo, y, 1, P0, w0, 0.2, cfg, stream, B, inner, 2, 0)
    fm = nf.newFactorizationMachine("regression", degree=1, nComponents=3, warmStart=True)
    fm.set_params(P0, w0, 0.2)
    opt = nf.newMBPSGD(maxIter=2, eta0=0.1, alpha=1e-2, alpha0=1e-2, reg=nf.newL1(), miniBatchSize=B, verbose=0, tol=-1.0)
    opt.it = 1
    opt.fit(to_gpu(Xo), y, fm, stream=stream)
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert abs(fm.intercept - b) < 1e-11
    assert_close([h[1] for h in opt.history], losses, 1e-10, 1e-13, "loss")


@pytest.mark.parametrize("loss,fit_linear,fit_intercept,scheduling", [
    ("logistic", True, True, "constant"), ("squared_hinge", False, True, "invscaling"), ("huber", True, False, "optimal"),
    ("squared", False, False, "constant")])
def test_flags_losses_schedules(loss, fit_linear, fit_intercept, scheduling):
    n, d, k, B = 64, 8, 3, 8
    Xo, Xd, y = make_fm_dataset(n, d, 2, k, 5, "explicit", fit_linear, fit_intercept, threshold=0.3)
    task = "regression" if loss in ("squared", "huber") else "classification"
    yo = y if task == "regression" else np.sign(y)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", fit_linear, scale=0.3)
    check(Xo, yo, task, 2, "explicit", k, P0, w0, 0.2, n_aug, B, 2, "squaredl12", loss=loss, fit_linear=fit_linear,
          fit_intercept=fit_intercept, scheduling=scheduling, eta0=0.05)


@pytest.mark.parametrize("reg,transpose,k", [("squaredl12", True, 5), ("squaredl12", False, 5), ("squaredl21", False, 3),
                                             ("l21", None, 7), ("squaredl12", True, 17), ("squaredl12", False, 33)])
def test_padded_components_and_wide_rows(reg, transpose, k):
    """k that does not fill the lane mapping (Kp > k): the padding stays zero and out of every norm / threshold"""
    n, d, B = 60, 40, 12
    Xo = random_csr(n, d, 6, 3)
    rng = np.random.default_rng(4)
    y = rng.normal(size=n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.3)
    check(Xo, y, "regression", 2, "explicit", k, P0, w0, 0.0, n_aug, B, 2, reg, transpose=transpose, gamma=0.05)


def test_heavy_features_and_many_rows():
    """few features, large mini-batches: every feature is touched > 128 times per batch (the segment path of the
    column phase), 3000 rows for the coupled threshold's strided passes"""
    n, d, k, B = 1200, 3000, 8, 600
    rng = np.random.default_rng(8)
    Xo = random_csr(n, 12, 5, 1)  # 12 hot features ...
    idx = Xo.indices.reshape(n, 5).copy()
    idx[:, 4] = rng.integers(12, d, size=n)  # ... and one cold one per row
    Xo = O.Dataset(Xo.indptr, idx.reshape(-1), Xo.data, n, d)
    y = rng.normal(size=n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.2)
    for reg in ("squaredl12", "squaredl21", "l1"):
        check(Xo, y, "regression", 2, "explicit", k, P0, w0, 0.0, n_aug, B, 2, reg, gamma=0.01, eta0=0.1)


@pytest.mark.parametrize("reg,k,gamma", [("squaredl12", 2, 1e-4), ("squaredl21", 2, 1e-4), ("squaredl12", 5, 1e-2),
                                         ("squaredl12", 3, 30.0), ("squaredl12", 16, 1.0)])
def test_more_features_than_the_register_resident_step_holds(reg, k, gamma):
    """d > 16384: the coupled threshold runs row-parallel passes over memory (k_prox_pass_*; a large gamma needs more
    passes than are enqueued blindly and ends in k_prox_finish) / k_psgd_prox_norms"""
    n, d, B = 64, 20000, 16
    Xo = random_csr(n, d, 40, 6)
    y = np.random.default_rng(2).normal(size=n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.3)
    fm, P = check(Xo, y, "regression", 2, "explicit", k, P0, w0, 0.0, n_aug, B, 2, reg, gamma=gamma)
    assert np.array_equal(fm.P == 0.0, P == 0.0)


@pytest.mark.parametrize("degree,fit_lower", [(3, "augment"), (2, "augment"), (4, "explicit")])
def test_dummy_features_through_the_segment_path(degree, fit_lower):
    """fitLower = augment: the dummy features are touched by every sample of a mini-batch of 300 (> 128 touches: the
    column phase's segment path, for models with several orders too)"""
    n, d, k, B = 600, 30, 4, 300
    Xo, Xd, y = make_fm_dataset(n, d, degree, k, 9, fit_lower, threshold=0.7)
    P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, True, scale=0.2)
    check(Xo, y, "regression", degree, fit_lower, k, P0, w0, 0.05, n_aug, B, 2, "l1", gamma=1e-3, eta0=0.05)
    check(Xo, y, "regression", degree, fit_lower, k, P0, w0, 0.05, n_aug, B, 2, "l21", gamma=1e-3, eta0=0.05)


def test_ragged_rows_and_default_batch():
    """empty rows, rows longer than a wavefront; miniBatchSize / maxIterInner defaults (minibatch_psgd.nim:160-167)"""
    n, d, k = 150, 300, 4
    Xo = ragged_csr(n, d, 2, max_m=90)
    rng = np.random.default_rng(3)
    y = rng.normal(size=n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.1)
    nnz = int(Xo.indptr[-1])
    B = max((d * n) // nnz, 1)
    inner = (n - 1) // B + 1
    stream = make_stream(n, B * inner * 2, 1)
    cfg = O.psgd_cfg(reg="squaredl12")
    P, w, b, it, losses = run_oracle(Xo, y, 2, P0, w0, b0, cfg, stream, B, inner, 2, n_aug)
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, b0)
    opt = nf.newMBPSGD(maxIter=2, verbose=0, tol=-1.0)
    opt.it = 1
    opt.fit(to_gpu(Xo), y, fm, stream=stream)
    assert opt.batch == B and opt.it == it == 1 + 2 * inner
    assert_close(fm.P, P, RTOL, ATOL, "P")
    assert_close(fm.w, w, RTOL, ATOL, "w")
    assert abs(fm.intercept - b) < 1e-11


def test_internal_shuffle_stopping_and_verbose(capsys):
    """no stream: the host mirror shuffles, wraps and stops on |loss change| < tol (minibatch_psgd.nim:201-204)"""
    n, d, k = 200, 30, 4
    Xo, Xd, y = make_fm_dataset(n, d, 2, k, 1, "explicit", threshold=0.5)
    fm = nf.newFactorizationMachine("regression", nComponents=k)
    opt = nf.newMBPSGD(maxIter=200, eta0=0.02, gamma=1e-3, miniBatchSize=32, verbose=1, tol=2e-2, reg=nf.newL1())
    opt.fit(to_gpu(Xo), y, fm)
    out = capsys.readouterr().out
    assert "Minibatch size: 32" in out and "Number of inner iteration: 7" in out
    assert "Converged at epoch" in out and len(opt.history) < 200
    assert opt.history[-1][1] < opt.history[0][1]
    assert np.isfinite(fm.P).all()


@pytest.mark.parametrize("degree,fit_lower,loss,fit_linear,fit_intercept", [
    (2, "explicit", "squared", True, True), (3, "explicit", "logistic", True, True), (3, "augment", "squared", True, False),
    (2, "none", "squared_hinge", False, True), (4, "explicit", "huber", True, True)])
def test_predict_all_with_grad(degree, fit_lower, loss, fit_linear, fit_intercept):
    """pgd.predictAllWithGrad (optimizer/pgd.nim:70-103) through nfm_opt_predict_all_with_grad vs its restatement"""
    n, d, k = 300, 40, 5  # every feature is touched by ~100 samples of the single batch: heavy and light paths
    Xo, Xd, y = make_fm_dataset(n, d, degree, k, 21, fit_lower, fit_linear, fit_intercept, threshold=0.6)
    Xo2 = random_csr(n, d, 3, 5)
    task = "regression" if loss in ("squared", "huber") else "classification"
    for X_ in (Xo, Xo2):
        yo = y if task == "regression" else np.sign(y)
        P0, w0, b0, n_aug = init_fm(d, degree, k, fit_lower, fit_linear, scale=0.3)
        w0 = np.random.default_rng(3).normal(size=d) * (0.1 if fit_linear else 0.0)
        b0 = 0.25 if fit_intercept else 0.0
        yp, dL, gP, gw, gb = O.fm_predict_all_with_grad(X_, yo, degree, P0, w0, b0, loss, n_aug, fit_linear, fit_intercept)
        fm = gpu_fm(task, degree, k, fit_lower, fit_linear, fit_intercept, P0, w0, b0)
        yp_g, dL_g, g = nf.predictAllWithGrad(to_gpu(X_), yo, fm, loss=loss)
        assert_close(yp_g, yp, 1e-11, 1e-13, "yPred")
        assert_close(dL_g, dL, 1e-10, 1e-13, "dL")
        assert_close(g["P"], gP, 1e-9, 1e-13, "grad P")
        assert_close(g["w"], gw, 1e-9, 1e-13, "grad w")
        assert abs(g["intercept"] - gb) < 1e-12
        assert np.array_equal(fm.P, P0) and fm.intercept == b0  # the parameters are not stepped


@pytest.mark.parametrize("d,reg", [(3000, "squaredl12"), (20000, "squaredl12"), (3000, "squaredl21"), (3000, "l21")])
def test_bitwise_reproducible(d, reg):
    """no atomics, fixed summation orders, no random pivots: two fits from the same start give the same bits"""
    n, k, B = 900, 8, 300
    rng = np.random.default_rng(8)
    Xo = random_csr(n, 12, 5, 1)
    idx = Xo.indices.reshape(n, 5).copy()
    idx[:, 4] = rng.integers(12, d, size=n)
    Xo = O.Dataset(Xo.indptr, idx.reshape(-1), Xo.data, n, d)
    y = rng.normal(size=n)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.2)
    stream = make_stream(n, 3 * n, 4)
    out = []
    for _ in range(2):
        fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, b0)
        opt = nf.newMBPSGD(maxIter=3, eta0=0.1, gamma=0.01, reg=REGS[reg](), miniBatchSize=B, verbose=0, tol=-1.0)
        opt.it = 1
        opt.fit(to_gpu(Xo), y, fm, stream=stream)
        out.append((fm.P.copy(), fm.w.copy(), fm.intercept))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    assert (out[0][0] != P0).any()


def test_changed_hyperparameters_rebuild_the_device_optimizer():
    """the same MBPSGD object fitted with another gamma / regulariser must not reuse the cached device handle"""
    n, d, k, B = 64, 10, 3, 16
    Xo, Xd, y = make_fm_dataset(n, d, 2, k, 3, "explicit", threshold=0.3)
    P0, w0, b0, n_aug = init_fm(d, 2, k, "explicit", True, scale=0.3)
    stream = make_stream(n, 2 * n, 1)
    opt = nf.newMBPSGD(maxIter=2, eta0=0.2, gamma=1e-3, reg=nf.newL1(), miniBatchSize=B, verbose=0, tol=-1.0)
    got = []
    for gamma, reg in ((1e-3, nf.newL1()), (0.5, nf.newL1()), (0.5, nf.newL21())):
        opt.gamma, opt.reg, opt.it = gamma, reg, 1
        fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, b0)
        opt.fit(to_gpu(Xo), y, fm, stream=stream)
        cfg = O.psgd_cfg(eta0=0.2, gamma=gamma, reg=reg.name)
        P, w, b, it, _ = run_oracle(Xo, y, 2, P0, w0, b0, cfg, stream, B, 4, 2, n_aug)
        assert_close(fm.P, P, RTOL, ATOL, "P gamma=%g %s" % (gamma, reg.name))
        got.append(fm.P.copy())
    assert not np.array_equal(got[0], got[1]) and not np.array_equal(got[1], got[2])


def test_errors():
    n, d, k = 20, 6, 2
    Xo, Xd, y = make_fm_dataset(n, d, 3, k, 1, "explicit", threshold=0.3)
    fm = nf.newFactorizationMachine("regression", degree=3, nComponents=k)
    with pytest.raises(ValueError, match="supports only degree=2"):  # squaredl12.nim:103-105
        nf.newMBPSGD(verbose=0).fit(to_gpu(Xo), y, fm)
    with pytest.raises(ValueError):
        nf.newMBPSGD(reg="l1")
    ffm = nf.newFieldAwareFactorizationMachine("regression", nComponents=k)
    with pytest.raises(ValueError):
        nf.newMBPSGD(verbose=0).fit(to_gpu(Xo), y, ffm)
    fm2 = nf.newFactorizationMachine("regression", nComponents=k)
    with pytest.raises(ValueError, match="stream holds fewer"):
        nf.newMBPSGD(verbose=0, maxIter=3, miniBatchSize=8).fit(to_gpu(Xo), y, fm2, stream=np.arange(10))


def test_abi_error_paths():
    """the C entry points themselves (nimfm_hip.h): bad regulariser id, batch < 1, a stream that is not a whole number of
    mini-batches, predictAllWithGrad on an SGD optimizer"""
    import ctypes as C

    from nimfm_amd import _capi as capi

    n, d, k = 20, 6, 2
    Xo, Xd, y = make_fm_dataset(n, d, 2, k, 1, "explicit", threshold=0.3)
    X = to_gpu(Xo)
    X.set_targets(y)
    fm = nf.newFactorizationMachine("regression", nComponents=k)
    fm.init(X)
    mh = fm._push(X.ctx)
    L = capi.lib()

    def create(reg=0, batch=4, transpose=0):
        h = C.c_void_p()
        cfg = capi.MBPSGDCfg(0.1, 1e-6, 1e-3, 1e-4, 1e-4, 1.0, 1.0, 0, 1, reg, transpose, batch)
        return L.nfm_mbpsgd_create(mh, C.byref(cfg), C.byref(h)), h

    assert create(reg=7)[0] == capi.ERR_INVALID
    assert create(batch=0)[0] == capi.ERR_INVALID
    assert create(reg=capi.REG["squaredl21"], transpose=1)[0] != 0
    rc, h = create()
    assert rc == 0
    ls, vs = C.c_double(), C.c_double()
    perm = np.arange(10, dtype=np.int64)
    assert L.nfm_opt_epoch(h, X.h, perm.ctypes.data_as(C.c_void_p), 0, 10, C.byref(ls), C.byref(vs)) == capi.ERR_INVALID  # 10 % 4
    assert b"whole number of mini-batches" in L.nfm_last_error()
    assert L.nfm_opt_epoch(h, X.h, None, 0, 40, C.byref(ls), C.byref(vs)) == capi.ERR_INVALID  # past the data without a stream
    assert L.nfm_opt_epoch(h, X.h, perm.ctypes.data_as(C.c_void_p), 0, 8, C.byref(ls), C.byref(vs)) == 0
    it = C.c_int64()
    L.nfm_opt_get_it(h, C.byref(it))
    assert it.value == 3  # two mini-batches
    L.nfm_opt_destroy(h)
    sgd = nf.newSGD(verbose=0)
    sgd._handle(fm, X.ctx, "sequential")
    gb = C.c_double()
    assert L.nfm_opt_predict_all_with_grad(sgd._h, X.h, None, None, None, None, C.byref(gb), None) == capi.ERR_INVALID
