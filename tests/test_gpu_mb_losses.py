"""-m gpu: NFM_MODE_MINIBATCH with the two losses the other mini-batch tests leave out -- squared hinge
(loss.nim:33-48) and Huber (loss.nim:84-93, sign quirk kept) -- x SGD / AdaGrad x FM (degree 2 and 3) / field-aware,
on seeded random shapes (ragged rows, empty rows, unsorted storage order, a few very popular features, with and without
a permutation per epoch), against the CPU restatement of the rule (oracle/nimfm_mb.c: O.*_epoch_mb).  The cases are the
generator of tests/fuzz_mb.py with fixed seeds."""
import itertools

import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import init_ffm
from gpu_common import gpu_ffm, gpu_fm, to_gpu

pytestmark = pytest.mark.gpu


def draw_case(seed, model):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(50, 1500))
    d = int(rng.integers(8, 400))
    k = int(rng.choice([1, 3, 4, 8, 16] if model == "ffm" else [1, 3, 4, 8, 16, 20, 32, 64]))
    max_m = int(min(d, rng.choice([3, 8, 20, 64])))
    B = int(rng.choice([1, 7, 64, 256, 1000]))
    hot = seed % 3 == 0  # a few features that most samples have
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        m = 0 if rng.random() < 0.05 else int(rng.integers(1, max_m + 1))
        if hot and d > 4:
            p = np.full(d, 1.0)
            p[:3] = d
            idx = rng.choice(d, size=m, replace=False, p=p / p.sum())
        else:
            idx = rng.choice(d, size=m, replace=False)
        if rng.random() < 0.5:
            idx = np.sort(idx)
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m))
        indptr.append(indptr[-1] + m)
    idx = np.concatenate(rows).astype(np.int64)
    val = np.concatenate(vals)
    y = rng.standard_normal(n)
    perms = np.stack([rng.permutation(n) for _ in range(2)]).astype(np.int64) if seed % 2 == 0 else None
    return rng, n, d, k, B, np.array(indptr), idx, val, y, perms


CASES = list(itertools.product(["squared_hinge", "huber"], ["sgd", "adagrad"], ["fm2", "fm3", "ffm"], [0, 1, 2, 3]))


@pytest.mark.parametrize("loss,solver,model,seed", CASES)
def test_minibatch_squared_hinge_and_huber(loss, solver, model, seed):
    rng, n, d, k, B, indptr, idx, val, y, perms = draw_case(seed * 7 + len(loss) + len(solver), model)
    task = "classification" if loss == "squared_hinge" else "regression"
    if task == "classification":
        y = np.sign(y) + (y == 0)
    epochs = 2
    pe = lambda e: None if perms is None else perms[e]  # noqa: E731
    kw = dict(loss=loss, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B, maxIter=epochs)
    if model == "ffm":
        F = int(rng.integers(2, 9))
        field_of = rng.integers(0, F, size=d)
        Xo = O.Dataset(indptr, idx, val, n, d, field_of[idx], F)
        P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
        P, w, b, it = P0.copy(), w0.copy(), b0, 1
        mdl = gpu_ffm(task, k, True, True, P0, w0, b0)
        if solver == "sgd":
            cfg = O.sgd_cfg(eta0=0.01, loss=loss)
            for e in range(epochs):
                b, it, _, _ = O.ffm_sgd_epoch_mb(Xo, y, P, w, b, cfg, B, perm=pe(e), it=it)
            opt = nf.newSGD(eta0=0.01, **kw)
        else:
            cfg = O.adagrad_cfg(loss=loss)
            st = O.AdaState(F, d, k, d)
            for e in range(epochs):
                b, it, _, _ = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, B, st, perm=pe(e), it=it)
            b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
            opt = nf.newAdaGrad(**kw)
    else:
        degree = 2 if model == "fm2" else 3
        nb = degree - 1
        Xo = O.Dataset(indptr, idx, val, n, d)
        P0, w0, b0 = rng.standard_normal((nb, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
        P, w, b, it = P0.copy(), w0.copy(), b0, 1
        mdl = gpu_fm(task, degree, k, "explicit", True, True, P0, w0, b0)
        if solver == "sgd":
            cfg = O.sgd_cfg(eta0=0.01, loss=loss)
            for e in range(epochs):
                b, it, _, _ = O.fm_sgd_epoch_mb(Xo, y, degree, P, w, b, cfg, B, perm=pe(e), it=it)
            opt = nf.newSGD(eta0=0.01, **kw)
        else:
            cfg = O.adagrad_cfg(loss=loss)
            st = O.AdaState(nb, d, k, d)
            for e in range(epochs):
                b, it, _, _ = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, B, st, perm=pe(e), it=it)
            b = O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st)
            opt = nf.newAdaGrad(**kw)
    opt.fit(to_gpu(Xo), y, mdl, perms=perms)
    assert np.isfinite(P).all() and float(np.abs(P).max()) < 1e3, "the draw diverges on the CPU as well: pick another seed"
    assert opt.it == it
    scale = max(1e-3, float(np.abs(P).max()))
    assert float(np.abs(mdl.P - P).max()) / scale < 1e-8
    assert float(np.abs(mdl.w - w).max()) / max(1e-3, float(np.abs(w).max())) < 1e-8
    assert abs(mdl.intercept - b) < 1e-8
