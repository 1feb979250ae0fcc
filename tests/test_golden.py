"""Golden vectors (tests/golden/fm_hotpath_golden.npz, provenance in tests/golden/make_golden.py):
produced by the brute-force definition; checked here against the oracle's fast path on CPU and,
with -m gpu, against the HIP sequential mode through the C ABI.  Tolerance: the reference's own
fast-vs-slow tolerance, rtol 1e-6 / atol 1e-9 (tests/utils.nim:82-105)."""
import itertools
import os

import numpy as np
import pytest

import oracle as O
from common import assert_close

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fm_hotpath_golden.npz"))
CASES = list(itertools.product([2, 3], ["explicit", "augment", "none"]))
K, EPOCHS = 4, 2


@pytest.mark.parametrize("degree,fit_lower", CASES)
def test_oracle_fast_path_vs_golden(degree, fit_lower):
    t = "fm_d%d_%s" % (degree, fit_lower)
    Xd, y, P0, perms = G[t + "_X"], G[t + "_y"], G[t + "_P0"], G[t + "_perms"]
    d = Xd.shape[1]
    n_aug = O.n_augments(degree, fit_lower, True)
    Xo = O.Dataset.from_dense(Xd)
    assert_close(O.fm_decision_function(Xo, degree, P0, G[t + "_wq"], 0.25, n_aug), G[t + "_decision"])
    P, w, b, *_ = O.fm_sgd_fit(Xo, y, degree, P0, np.zeros(d), 0.0, O.sgd_cfg(), EPOCHS, n_aug, perms=perms)
    assert_close(P, G[t + "_sgd_P"]); assert_close(w, G[t + "_sgd_w"]); assert abs(b - G[t + "_sgd_b"]) < 1e-7
    P, w, b, *_ = O.fm_adagrad_fit(Xo, y, degree, P0, np.zeros(d), 0.0, O.adagrad_cfg(), EPOCHS, n_aug, perms=perms)
    assert_close(P, G[t + "_ada_P"]); assert_close(w, G[t + "_ada_w"]); assert abs(b - G[t + "_ada_b"]) < 1e-7


def test_oracle_ffm_vs_golden():
    Xd, y, P0, field_of = G["ffm_X"], G["ffm_y"], G["ffm_P0"], G["ffm_field_of"]
    d, F = Xd.shape[1], P0.shape[0]
    Xo = O.Dataset.from_dense(Xd, field_of, F)
    assert_close(O.ffm_decision_function(Xo, P0, np.linspace(-1, 1, d), -0.5), G["ffm_decision"])
    P, w, b, *_ = O.ffm_sgd_fit(Xo, y, P0, np.zeros(d), 0.0, O.sgd_cfg(), EPOCHS)
    assert_close(P, G["ffm_sgd_P"]); assert_close(w, G["ffm_sgd_w"]); assert abs(b - G["ffm_sgd_b"]) < 1e-7
    P, w, b, *_ = O.ffm_adagrad_fit(Xo, y, P0, np.zeros(d), 0.0, O.adagrad_cfg(), EPOCHS)
    assert_close(P, G["ffm_ada_P"]); assert_close(w, G["ffm_ada_w"]); assert abs(b - G["ffm_ada_b"]) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("degree,fit_lower", CASES)
def test_gpu_vs_golden(degree, fit_lower):
    import nimfm_amd as nf
    from gpu_common import gpu_fm, to_gpu
    t = "fm_d%d_%s" % (degree, fit_lower)
    Xd, y, P0, perms = G[t + "_X"], G[t + "_y"], G[t + "_P0"], G[t + "_perms"]
    d = Xd.shape[1]
    X = to_gpu(O.Dataset.from_dense(Xd))
    fm = gpu_fm("regression", degree, K, fit_lower, True, True, P0, G[t + "_wq"], 0.25)
    assert_close(fm.decisionFunction(X), G[t + "_decision"])
    for opt, key in ((nf.newSGD(maxIter=EPOCHS, verbose=0, tol=0), "sgd"), (nf.newAdaGrad(maxIter=EPOCHS, verbose=0, tol=0), "ada")):
        fm = gpu_fm("regression", degree, K, fit_lower, True, True, P0, np.zeros(d), 0.0)
        opt.fit(X, y, fm, perms=perms)
        assert_close(fm.P, G[t + "_%s_P" % key]); assert_close(fm.w, G[t + "_%s_w" % key])
        assert abs(fm.intercept - G[t + "_%s_b" % key]) < 1e-7


@pytest.mark.gpu
def test_gpu_ffm_vs_golden():
    import nimfm_amd as nf
    from gpu_common import gpu_ffm, to_gpu
    Xd, y, P0, field_of = G["ffm_X"], G["ffm_y"], G["ffm_P0"], G["ffm_field_of"]
    d, F = Xd.shape[1], P0.shape[0]
    X = to_gpu(O.Dataset.from_dense(Xd, field_of, F))
    ffm = gpu_ffm("regression", K, True, True, P0, np.linspace(-1, 1, d), -0.5)
    assert_close(ffm.decisionFunction(X), G["ffm_decision"])
    for opt, key in ((nf.newSGD(maxIter=EPOCHS, verbose=0, tol=0, shuffle=False), "sgd"),
                     (nf.newAdaGrad(maxIter=EPOCHS, verbose=0, tol=0, shuffle=False), "ada")):
        ffm = gpu_ffm("regression", K, True, True, P0, np.zeros(d), 0.0)
        opt.fit(X, y, ffm)
        assert_close(ffm.P, G["ffm_%s_P" % key]); assert_close(ffm.w, G["ffm_%s_w" % key])
        assert abs(ffm.intercept - G["ffm_%s_b" % key]) < 1e-7


# ---- mini-batch proximal SGD (tests/golden/psgd_golden.npz, provenance in tests/golden/make_psgd_golden.py: a dense
# numpy statement of minibatch_psgd.nim with sort-based proximal operators) ----
GP = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psgd_golden.npz"))
PSGD_REGS = ["l1", "l21", "squaredl12", "squaredl21"]


@pytest.mark.parametrize("reg", PSGD_REGS)
def test_oracle_psgd_vs_golden(reg):
    Xd, y, P0, w0, stream = GP["X"], GP["y"], GP["P0"], GP["w0"], GP["stream"]
    eta0, alpha0, alpha, beta, gamma = GP["hyper"]
    B, outer = int(GP["batch"]), int(GP["outer"])
    Xo = O.Dataset.from_dense(Xd)
    cfg = O.psgd_cfg(eta0=eta0, alpha0=alpha0, alpha=alpha, beta=beta, gamma=gamma, reg=reg)
    P, w, b, it = P0.copy(), w0.copy(), float(GP["b0"]), 1
    need = len(stream) // outer
    for t in range(outer):
        b, it, ls = O.fm_mbpsgd_epoch(Xo, y, 2, P, w, b, cfg, stream[t * need:(t + 1) * need], B, 0, it=it)
        assert abs(ls / need - GP[reg + "_loss"][t]) < 1e-9
    assert_close(P, GP[reg + "_P"]); assert_close(w, GP[reg + "_w"]); assert abs(b - GP[reg + "_b"]) < 1e-9
    assert np.array_equal(P == 0.0, GP[reg + "_P"] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("reg", PSGD_REGS)
def test_gpu_psgd_vs_golden(reg):
    import nimfm_amd as nf
    from gpu_common import gpu_fm, to_gpu
    Xd, y, P0, w0, stream = GP["X"], GP["y"], GP["P0"], GP["w0"], GP["stream"]
    eta0, alpha0, alpha, beta, gamma = GP["hyper"]
    regs = {"l1": nf.newL1, "l21": nf.newL21, "squaredl12": nf.newSquaredL12, "squaredl21": nf.newSquaredL21}
    fm = gpu_fm("regression", 2, P0.shape[1], "explicit", True, True, P0, w0, float(GP["b0"]))
    opt = nf.newMBPSGD(maxIter=int(GP["outer"]), eta0=eta0, alpha0=alpha0, alpha=alpha, beta=beta, gamma=gamma,
                       reg=regs[reg](), miniBatchSize=int(GP["batch"]), verbose=0, tol=-1.0)
    opt.it = 1
    opt.fit(to_gpu(O.Dataset.from_dense(Xd)), y, fm, stream=stream)
    assert_close(fm.P, GP[reg + "_P"]); assert_close(fm.w, GP[reg + "_w"]); assert abs(fm.intercept - GP[reg + "_b"]) < 1e-9
    assert_close([h[1] for h in opt.history], GP[reg + "_loss"], 1e-9, 1e-12)
    assert np.array_equal(fm.P == 0.0, GP[reg + "_P"] == 0.0)
