"""Random configurations of NFM_MODE_SEQUENTIAL (the reference's sample-by-sample order) against the reference-faithful CPU
restatement: shapes, orders, solvers, losses, ragged / empty rows, popular features, with and without a permutation.
Not part of the test suite -- a robustness sweep to run on a GPU box after changes to csrc/seq.hip.
usage: python tests/fuzz_seq.py [n_cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import nimfm_amd as nf, oracle as O
from gpu_common import gpu_ffm, gpu_fm, to_gpu
from common import init_ffm

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    n = int(rng.integers(1, 400))
    d = int(rng.integers(2, 600))
    k = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 20, 32, 50, 64, 65, 100, 128, 129, 200, 260]))  # (round 5: no cap on k)
    max_m = int(min(d, rng.choice([1, 3, 8, 20, 64, 100, 400])))
    B = 0
    solver = str(rng.choice(["sgd", "adagrad"]))
    loss = str(rng.choice(["squared", "logistic", "squared_hinge", "huber"]))
    degree = int(rng.choice([2, 2, 2, 3]))
    ffm = degree == 2 and rng.random() < 0.25 and k <= 16
    hot = rng.random() < 0.3  # a few features that most samples have
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = 0 if rng.random() < 0.05 else int(rng.integers(1, max_m + 1))
        if hot and d > 4:
            p = np.full(d, 1.0); p[:3] = d
            idx = rng.choice(d, size=m, replace=False, p=p / p.sum())
        else:
            idx = rng.choice(d, size=m, replace=False)
        if rng.random() < 0.5:
            idx = np.sort(idx)
        rows.append(idx); vals.append(rng.uniform(-1, 1, size=m)); indptr.append(indptr[-1] + m)
    idx = np.concatenate(rows).astype(np.int64) if indptr[-1] else np.zeros(0, np.int64)
    val = np.concatenate(vals) if indptr[-1] else np.zeros(0)
    y = rng.standard_normal(n)
    task = "classification" if loss in ("logistic", "squared_hinge") else "regression"
    if task == "classification":
        y = np.sign(y) + (y == 0)
    epochs = 2
    perms = np.stack([rng.permutation(n) for _ in range(epochs)]).astype(np.int64) if rng.random() < 0.6 else None
    tag = "case %d: n=%d d=%d k=%d m<=%d B=%d %s %s deg=%d ffm=%s hot=%s perm=%s" % (case, n, d, k, max_m, B, solver, loss, degree, ffm, hot, perms is not None)
    try:
        if ffm:
            F = int(rng.integers(2, 9))
            field_of = rng.integers(0, F, size=d)
            Xo = O.Dataset(np.array(indptr), idx, val, n, d, field_of[idx] if len(idx) else np.zeros(0, np.int64), F)
            P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
            P, w, b, it = P0.copy(), w0.copy(), b0, 1
            mdl = gpu_ffm(task, k, True, True, P0, w0, b0)
            if solver == "sgd":
                P, w, b, *_ = O.ffm_sgd_fit(Xo, y, P0, w0, b0, O.sgd_cfg(eta0=0.01, loss=loss), epochs, perms=perms)
                opt = nf.newSGD(maxIter=epochs, eta0=0.01, loss=loss, verbose=0, tol=0, shuffle=False)
            else:
                P, w, b, *_ = O.ffm_adagrad_fit(Xo, y, P0, w0, b0, O.adagrad_cfg(loss=loss), epochs, perms=perms)
                opt = nf.newAdaGrad(maxIter=epochs, loss=loss, verbose=0, tol=0, shuffle=False)
        else:
            Xo = O.Dataset(np.array(indptr), idx, val, n, d)
            nb = degree - 1
            P0, w0, b0 = rng.standard_normal((nb, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
            P, w, b, it = P0.copy(), w0.copy(), b0, 1
            mdl = gpu_fm(task, degree, k, "explicit", True, True, P0, w0, b0)
            if solver == "sgd":
                P, w, b, *_ = O.fm_sgd_fit(Xo, y, degree, P0, w0, b0, O.sgd_cfg(eta0=0.01, loss=loss), epochs, 0, perms=perms)
                opt = nf.newSGD(maxIter=epochs, eta0=0.01, loss=loss, verbose=0, tol=0, shuffle=False)
            else:
                P, w, b, *_ = O.fm_adagrad_fit(Xo, y, degree, P0, w0, b0, O.adagrad_cfg(loss=loss), epochs, 0, perms=perms)
                opt = nf.newAdaGrad(maxIter=epochs, loss=loss, verbose=0, tol=0, shuffle=False)
        opt.fit(to_gpu(Xo), y, mdl, perms=perms)
        if not np.isfinite(P).all() or float(np.abs(P).max()) > 1e3:
            print("diverged on the CPU as well (step size too large for this draw), skipped:", tag, flush=True)
            continue
        scale = max(1e-3, float(np.abs(P).max()))
        err = max(float(np.abs(mdl.P - P).max()) / scale, float(np.abs(mdl.w - w).max()) / max(1e-3, float(np.abs(w).max())), abs(mdl.intercept - b))
        worst = max(worst, err)
        # (NFM_SEQ_WIN=2 sends even these short fits through the window; AdaGrad there, with 65 ... 128 factors summed block by
        # block, is held at the reference's own fast-against-slow tolerance: tests/test_gpu_seqwin.py)
        tol = 1e-6 if (os.environ.get("NFM_SEQ_WIN") == "2" and solver == "adagrad") else 1e-8
        if not np.isfinite(err) or err > tol:
            print("MISMATCH", tag, "err", err, flush=True)
    except Exception as e:  # noqa: BLE001
        print("ERROR", tag, repr(e)[:300], flush=True)
print("fuzz: %d cases, worst relative error %.3g" % (n_cases, worst))
