"""Which paths agree with the CPU oracle BIT FOR BIT (max |difference| == 0) now that the device code is built without
fused multiply-adds?  Prints the largest absolute difference per path; run on a GPU box."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import nimfm_amd as nf, oracle as O
from common import init_ffm, make_perms, random_csr
from gpu_common import gpu_ffm, gpu_fm, ragged_csr, to_gpu

def md(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)))) if np.size(a) else 0.0

n, d, k = 300, 80, 8
Xo = ragged_csr(n, d, seed=3, max_m=30, empty_every=9)
rng = np.random.default_rng(1)
y = rng.standard_normal(n) * 0.5
P0, w0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
X = to_gpu(Xo)
perms = make_perms(n, 3)
print("decisionFunction FM degree 2:", md(gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.1).decisionFunction(X),
                                          O.fm_decision_function(Xo, 2, P0, w0, 0.1)))
P3 = rng.standard_normal((2, k, d)) * 0.05
print("decisionFunction FM degree 3:", md(gpu_fm("regression", 3, k, "explicit", True, True, P3, w0, 0.1).decisionFunction(X),
                                          O.fm_decision_function(Xo, 3, P3, w0, 0.1)))
for loss in ("squared", "logistic"):
    yy = y if loss == "squared" else np.sign(y)
    task = "regression" if loss == "squared" else "classification"
    for beta in (1e-3, 0.0):
        Pf, wf, bf, *_ = O.fm_sgd_fit(Xo, yy, 2, P0, w0, 0.1, O.sgd_cfg(loss=loss, beta=beta, alpha=beta), 3, 0, perms=perms)
        fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.1)
        nf.newSGD(maxIter=3, loss=loss, beta=beta, alpha=beta, verbose=0, tol=0).fit(X, yy, fm, perms=perms)
        print("sequential SGD %-8s beta=alpha=%g: P %.3g  w %.3g  b %.3g" % (loss, beta, md(fm.P, Pf), md(fm.w, wf), abs(fm.intercept - bf)))
    Pf, wf, bf, *_ = O.fm_adagrad_fit(Xo, yy, 2, P0, w0, 0.1, O.adagrad_cfg(loss=loss), 3, 0, perms=perms)
    fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.1)
    nf.newAdaGrad(maxIter=3, loss=loss, verbose=0, tol=0).fit(X, yy, fm, perms=perms)
    print("sequential AdaGrad %-8s: P %.3g  w %.3g  b %.3g" % (loss, md(fm.P, Pf), md(fm.w, wf), abs(fm.intercept - bf)))
    for B in (32, 100):
        P, w = P0.copy(), w0.copy()
        b, it = 0.1, 1
        for e in range(3):
            b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, yy, 2, P, w, b, O.sgd_cfg(loss=loss), B, perm=perms[e], it=it)
        fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.1)
        nf.newSGD(maxIter=3, loss=loss, verbose=0, tol=0, mode="minibatch", batch=B).fit(X, yy, fm, perms=perms)
        print("mini-batch SGD %-8s B=%d: P %.3g  w %.3g  b %.3g" % (loss, B, md(fm.P, P), md(fm.w, w), abs(fm.intercept - b)))
        P, w = P0.copy(), w0.copy()
        b, it = 0.1, 1
        st = O.AdaState(1, d, k, d)
        cfg = O.adagrad_cfg(loss=loss)
        for e in range(3):
            b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, yy, 2, P, w, b, cfg, B, st, perm=perms[e], it=it)
        b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
        fm = gpu_fm(task, 2, k, "explicit", True, True, P0, w0, 0.1)
        nf.newAdaGrad(maxIter=3, loss=loss, verbose=0, tol=0, mode="minibatch", batch=B).fit(X, yy, fm, perms=perms)
        print("mini-batch AdaGrad %-8s B=%d: P %.3g  w %.3g  b %.3g" % (loss, B, md(fm.P, P), md(fm.w, w), abs(fm.intercept - b)))
# field-aware
F = 6
field_of = rng.integers(0, F, size=d)
Xf = O.Dataset(Xo.indptr, Xo.indices, Xo.data, n, d, field_of[Xo.indices], F)
XF = to_gpu(Xf)
Pf0, wf0, bf0 = init_ffm(d, F, k)
print("decisionFunction FFM:", md(gpu_ffm("regression", k, True, True, Pf0, w0, 0.1).decisionFunction(XF), O.ffm_decision_function(Xf, Pf0, w0, 0.1)))
Pf, wf, bf, *_ = O.ffm_sgd_fit(Xf, y, Pf0, wf0, bf0, O.sgd_cfg(), 3)
m_ = gpu_ffm("regression", k, True, True, Pf0, wf0, bf0)
nf.newSGD(maxIter=3, verbose=0, tol=0, shuffle=False).fit(XF, y, m_)
print("sequential FFM SGD: P %.3g  w %.3g  b %.3g" % (md(m_.P, Pf), md(m_.w, wf), abs(m_.intercept - bf)))
Pf, wf, bf, *_ = O.ffm_adagrad_fit(Xf, y, Pf0, wf0, bf0, O.adagrad_cfg(), 3)
m_ = gpu_ffm("regression", k, True, True, Pf0, wf0, bf0)
nf.newAdaGrad(maxIter=3, verbose=0, tol=0, shuffle=False).fit(XF, y, m_)
print("sequential FFM AdaGrad: P %.3g  w %.3g  b %.3g" % (md(m_.P, Pf), md(m_.w, wf), abs(m_.intercept - bf)))
