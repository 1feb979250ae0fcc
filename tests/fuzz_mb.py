"""Random mini-batch configurations against the CPU restatement of the rule (oracle/nimfm_mb.c): shapes, batch sizes,
orders, solvers, losses, row lengths (ragged, empty rows, a few very popular features), with and without a permutation.
Not part of the test suite -- a robustness sweep to run on a GPU box after kernel or plan changes.
usage: python tests/fuzz_mb.py [n_cases] [seed]      (FUZZ_CPU_TWIN=1: no GPU -- the oracle's parity build against its -O3 -march=native
build on the same draws: what two correct implementations of the rule differ by, case by case)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np
import oracle as O
CPU_TWIN = os.environ.get("FUZZ_CPU_TWIN") == "1"  # no GPU: two CPU builds of the rule against each other (how ill-conditioned is a draw)
if not CPU_TWIN:
    import nimfm_amd as nf
    from gpu_common import gpu_ffm, gpu_fm, to_gpu
from common import init_ffm

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(n_cases):
    n = int(rng.integers(1, 2500))
    d = int(rng.integers(2, 600))
    k = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 20, 32, 50, 64, 65, 100, 128, 129, 200, 260]))  # (round 5: no cap on k)
    max_m = int(min(d, rng.choice([1, 3, 8, 20, 64, 100, 400])))
    B = int(rng.choice([1, 7, 64, 256, 1000, 4096]))
    solver = str(rng.choice(["sgd", "adagrad"]))
    loss = str(rng.choice(["squared", "logistic", "squared_hinge", "huber"]))
    degree = int(rng.choice([2, 2, 2, 3]))
    ffm = degree == 2 and rng.random() < 0.25 and k <= 16
    hot = rng.random() < 0.3  # a few features that most samples have
    cap = float(rng.choice([1.0, 1.0, 2.0, 16.0, 32.0]))    # SGD: nfm_opt_set_touch_cap (round 5: the bench runs 16 / 32)
    gamma = float(rng.choice([0.0, 0.0, 0.1, 0.5]))         # AdaGrad: nfm_opt_set_ada_cross (round 5)
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
    tag = "case %d: n=%d d=%d k=%d m<=%d B=%d %s %s deg=%d ffm=%s hot=%s perm=%s cap=%g gamma=%g" % (case, n, d, k, max_m, B, solver, loss, degree, ffm, hot, perms is not None, cap, gamma)
    try:
        def cpu_fit(Xo, P0, w0, b0, F):
            """the rule on the CPU (oracle/nimfm_mb.c) from this start: (P, w, b)"""
            P, w, b, it = P0.copy(), w0.copy(), b0, 1
            pe = lambda e: None if perms is None else perms[e]
            if solver == "sgd":
                cfg = O.sgd_cfg(eta0=0.01, loss=loss)
                for e in range(epochs):
                    if F:
                        b, it, _, _ = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, cfg, B, perm=pe(e), it=it, touch_cap=cap)
                    else:
                        b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, degree, P, w, b, cfg, B, perm=pe(e), it=it, touch_cap=cap)
            else:
                cfg = O.adagrad_cfg(loss=loss)
                st = O.AdaState(F if F else degree - 1, d, k, d)
                for e in range(epochs):
                    if F:
                        b, it, _, _ = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, B, st, perm=pe(e), it=it, ada_cross=gamma)
                    else:
                        b, it, _, _ = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, B, st, perm=pe(e), it=it, ada_cross=gamma)
                b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st) if F else O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st)
            return P, w, b

        F = 0
        if ffm:
            F = int(rng.integers(2, 9))
            field_of = rng.integers(0, F, size=d)
            Xo = O.Dataset(np.array(indptr), idx, val, n, d, field_of[idx] if len(idx) else np.zeros(0, np.int64), F)
            if np.diff(Xo.indptr).max(initial=0) > 64:
                continue
            P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
        else:
            Xo = O.Dataset(np.array(indptr), idx, val, n, d)
            P0, w0, b0 = rng.standard_normal((degree - 1, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
        P, w, b = cpu_fit(Xo, P0, w0, b0, F)
        if CPU_TWIN:  # the yardstick: the SAME rule and sources built with fused multiply-adds (oracle/Makefile target fma) against the parity build
            with O.variant("fma"):
                P2, w2, b2 = cpu_fit(Xo, P0, w0, b0, F)
            class _M: pass
            mdl = _M(); mdl.P, mdl.w, mdl.intercept = P2, w2, b2
        else:
            mdl = gpu_ffm(task, k, True, True, P0, w0, b0) if F else gpu_fm(task, degree, k, "explicit", True, True, P0, w0, b0)
            if solver == "sgd":
                opt = nf.newSGD(maxIter=epochs, eta0=0.01, loss=loss, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B, touchCap=cap)
            else:
                opt = nf.newAdaGrad(maxIter=epochs, loss=loss, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B, adaCross=gamma)
            opt.fit(to_gpu(Xo), y, mdl, perms=perms)
        if not np.isfinite(P).all() or float(np.abs(P).max()) > 1e3:
            print("diverged on the CPU as well (step size too large for this draw), skipped:", tag, flush=True)
            continue
        scale = max(1e-3, float(np.abs(P).max()))
        err = max(float(np.abs(mdl.P - P).max()) / scale, float(np.abs(mdl.w - w).max()) / max(1e-3, float(np.abs(w).max())), abs(mdl.intercept - b))
        worst = max(worst, err)
        if not np.isfinite(err) or err > 1e-8:
            print("MISMATCH", tag, "err", err, flush=True)
    except Exception as e:  # noqa: BLE001
        print("ERROR", tag, repr(e)[:300], flush=True)
print("fuzz: %d cases, worst relative error %.3g" % (n_cases, worst))
