"""What the data-parallel exchange does to convergence, measured on ONE GPU with local groups of 2 / 4 / 8 ranks
(nfm_dp_create_local: a context, a stream, a shard, a replica and a host thread per rank).

A planted degree-2 FM (labels + noise), n training samples split into contiguous shards (optimizer/sgd_multi.nim:85-88),
held-out loss after E epochs for
   one rank over ALL samples (what N ranks should match per epoch),
   N ranks: SGD with the ranks' increments averaged / summed / averaged at a step size x N, exchanges every S mini-batches,
            AdaGrad with the state increments summed.
progress = (L_start - L_run) / (L_start - L_one_rank): 1.0 = N ranks make one rank's progress per epoch (throughput then IS
speed-up), 1/N = the ranks only share the work of one.   usage: python tools/dp_convergence.py [E] [touch_cap]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import nimfm_amd as nf  # noqa: E402
import oracle as O  # noqa: E402
from common import random_csr  # noqa: E402
from nimfm_amd import dp  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CAP = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
n, nt, d, m, k, B = 160_000, 20_000, 20_000, 16, 8, 512
full = random_csr(n + nt, d, m, seed=5)
rng = np.random.default_rng(9)
Pt, wt = rng.standard_normal((1, k, d)) * 0.3, rng.standard_normal(d) * 0.3
yfull = O.fm_decision_function(full, 2, Pt, wt, 0.1) + 0.1 * rng.standard_normal(n + nt)


def sub(lo, hi):
    a, b = full.indptr[lo], full.indptr[hi]
    return O.Dataset(full.indptr[lo:hi + 1] - a, full.indices[a:b], full.data[a:b], hi - lo, d)


Xtr, Xte, ytr, yte = sub(0, n), sub(n, n + nt), yfull[:n], yfull[n:]
P0, w0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01, np.zeros(d)


def rmse(P, w, b):
    return float(np.sqrt(np.mean((O.fm_decision_function(Xte, 2, P, w, b) - yte) ** 2)))


def make_opt(solver, eta_scale=1.0):
    if solver == "sgd":
        return nf.newSGD(maxIter=E, eta0=0.05 * eta_scale, alpha=1e-5, beta=1e-5, verbose=0, tol=0, shuffle=False, mode="minibatch",
                         batch=B, touchCap=CAP)
    return nf.newAdaGrad(maxIter=E, eta0=0.1, alpha=1e-5, beta=1e-5, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)


def single(solver):
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    fm.set_params(P0, w0, 0.0)
    make_opt(solver).fit(nf.newCSRDataset(Xtr.data, Xtr.indices, Xtr.indptr, n, d), ytr, fm)
    return rmse(fm.P, fm.w, fm.intercept)


def ranks(solver, world, S, combine="mean", eta_scale=1.0):
    ctxs = [nf.Context(0) for _ in range(world)]
    groups = dp.Group.local(ctxs)
    res, err = [None] * world, []

    def body(r):
        try:
            lo, hi = dp.shard_bounds(n, r, world)
            a, b = Xtr.indptr[lo], Xtr.indptr[hi]
            X = nf.newCSRDataset(Xtr.data[a:b], Xtr.indices[a:b], Xtr.indptr[lo:hi + 1] - a, hi - lo, d, ctx=ctxs[r])
            fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
            fm.set_params(P0, w0, 0.0)
            opt = make_opt(solver, eta_scale)
            opt.setDataParallel(groups[r], S, True, combine)
            opt.fit(X, ytr[lo:hi], fm)
            res[r] = (fm.P.copy(), fm.w.copy(), fm.intercept)
        except BaseException as e:  # noqa: BLE001
            err.append((r, e))

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in groups:
        g.close()
    if err:
        return float("nan")
    v = rmse(*res[0])
    return v if np.isfinite(v) else float("nan")


L0 = rmse(P0, w0, 0.0)
print("planted FM: %d train / %d held-out samples, d=%d, m=%d, k=%d, mini-batch %d, %d epochs, SGD touch cap %g; held-out RMSE at start %.4f"
      % (n, nt, d, m, k, B, E, CAP, L0), flush=True)
for solver in ("sgd", "adagrad"):
    one = single(solver)
    print("%s: one rank over all samples: %.4f" % (solver, one), flush=True)
    for world in (2, 4, 8):
        nb = (n // world) // B
        for S in (1, 4, 16, 0):
            row = []
            variants = [("mean", 1.0), ("sum", 1.0), ("mean", float(world))] if solver == "sgd" else [("mean", 1.0)]
            for combine, es in variants:
                v = ranks(solver, world, S, combine, es)
                prog = (L0 - v) / (L0 - one) if np.isfinite(v) else float("nan")
                row.append("%s%s %.4f (progress %.2f)" % (combine if solver == "sgd" else "state-sum", " x%d step" % world if es != 1.0 else "", v, prog))
            print("  %d ranks, exchange every %s (%d mini-batches per rank and epoch): %s" % (world, "%d mini-batches" % S if S else "epoch", nb, "; ".join(row)), flush=True)
