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
import numpy as np  # noqa: E402

import nimfm_amd as nf  # noqa: E402
from nimfm_amd import dp  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CAP = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
n, nt, d, m, k, B = 160_000, 20_000, 20_000, 16, 8, 512
rng = np.random.default_rng(5)
idx = np.sort(rng.integers(0, d, size=(n + nt, m)), axis=1)
for _ in range(50):  # distinct ids inside a row
    dup = np.zeros_like(idx, dtype=bool)
    dup[:, 1:] = idx[:, 1:] == idx[:, :-1]
    if not dup.any():
        break
    idx[dup] = rng.integers(0, d, size=int(dup.sum()))
    idx.sort(axis=1)
val = rng.uniform(-1.0, 1.0, size=(n + nt, m))


class Part:
    def __init__(self, lo, hi):
        self.n = hi - lo
        self.indptr = np.arange(self.n + 1, dtype=np.int64) * m
        self.indices = np.ascontiguousarray(idx[lo:hi].ravel())
        self.data = np.ascontiguousarray(val[lo:hi].ravel())


Xtr, Xte = Part(0, n), Part(n, n + nt)
rng = np.random.default_rng(9)
planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
planted.set_params(rng.standard_normal((1, k, d)) * 0.3, rng.standard_normal(d) * 0.3, 0.1)
Xall_gpu = nf.newCSRDataset(np.concatenate([Xtr.data, Xte.data]), np.concatenate([Xtr.indices, Xte.indices]),
                            np.arange(n + nt + 1, dtype=np.int64) * m, n + nt, d)
yfull = planted.decisionFunction(Xall_gpu) + 0.1 * rng.standard_normal(n + nt)
del planted, Xall_gpu
ytr, yte = yfull[:n], yfull[n:]
Xte_gpu = nf.newCSRDataset(Xte.data, Xte.indices, Xte.indptr, nt, d)
P0, w0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01, np.zeros(d)


def rmse(P, w, b):
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    fm.set_params(P, w, b)
    return float(np.sqrt(np.mean((fm.decisionFunction(Xte_gpu) - yte) ** 2)))


def make_opt(solver, eta_scale=1.0):
    if solver == "sgd":
        return nf.newSGD(maxIter=E, eta0=0.05 * eta_scale, alpha=1e-5, beta=1e-5, verbose=0, tol=0, shuffle=False, mode="minibatch",
                         batch=B, touchCap=CAP)
    return nf.newAdaGrad(maxIter=E, eta0=0.1, alpha=1e-5, beta=1e-5, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)


def single(solver):
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    fm.set_params(P0, w0, 0.0)
    opt = make_opt(solver)
    opt.fit(nf.newCSRDataset(Xtr.data, Xtr.indices, Xtr.indptr, n, d), ytr, fm)
    single.train = opt.history[-1][1]  # the last epoch's running training loss (what the reference prints)
    return rmse(fm.P, fm.w, fm.intercept)


def ranks(solver, world, S, combine="mean", eta_scale=1.0):
    ctxs = [nf.Context(0) for _ in range(world)]
    groups = dp.Group.local(ctxs)
    res, err = [None] * world, []

    def body(r):
        try:
            lo, hi = dp.shard_bounds(n, r, world)
            a, b = lo * m, hi * m
            X = nf.newCSRDataset(Xtr.data[a:b], Xtr.indices[a:b], Xtr.indptr[lo:hi + 1] - a, hi - lo, d, ctx=ctxs[r])
            fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
            fm.set_params(P0, w0, 0.0)
            opt = make_opt(solver, eta_scale)
            opt.setDataParallel(groups[r], S, True, combine)
            opt.fit(X, ytr[lo:hi], fm)
            res[r] = (fm.P.copy(), fm.w.copy(), fm.intercept)
            if r == 0:
                ranks.train = opt.history[-1][1]  # (over the samples of all ranks)
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
if len(sys.argv) > 3:  # one row in a process of its own: "solver world S one_rank_value"
    solver, world, S, one = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
    nb = (n // world) // B
    row = []
    variants = [("mean", 1.0), ("sum", 1.0)] if solver == "sgd" else [("sum", 1.0), ("state_mean", 1.0), ("state_rsqrt", 1.0), ("state_cross", 1.0)]
    for combine, es in variants:
        v = ranks(solver, world, S, combine, es)
        prog = (L0 - v) / (L0 - one) if np.isfinite(v) else float("nan")
        row.append("%s %.4f (progress %.2f; last epoch's training loss %.4f)" % (combine if solver == "sgd" else ("state-" + combine.replace("state_", "")), v, prog,
                                                                                      getattr(ranks, "train", float("nan"))))
    print("  %d ranks, exchange every %s (%d mini-batches per rank and epoch): %s" % (world, "%d mini-batches" % S if S else "epoch", nb, "; ".join(row)), flush=True)
    sys.exit(0)
import subprocess  # noqa: E402

print("planted FM: %d train / %d held-out samples, d=%d, m=%d, k=%d, mini-batch %d, %d epochs, SGD touch cap %g; held-out RMSE at start %.4f"
      % (n, nt, d, m, k, B, E, CAP, L0), flush=True)
for solver in (os.environ.get("DPC_SOLVERS") or "sgd,adagrad").split(","):
    one = single(solver)
    print("%s: one rank over all samples: %.4f (last epoch's training loss %.4f)" % (solver, one, getattr(single, "train", float("nan"))), flush=True)
    for world in [int(v) for v in (os.environ.get("DPC_WORLDS") or "2,4,8").split(",")]:
        for S in [int(v) for v in (os.environ.get("DPC_PERIODS") or "1,4,16,0").split(",")]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(E), str(CAP), solver, str(world), str(S), repr(one)],
                                 capture_output=True, text=True, timeout=600)
            lines = [ln for ln in out.stdout.splitlines() if ln.startswith("  ")]
            print(lines[-1] if lines else "  %d ranks, S=%d: FAILED rc=%d %s" % (world, S, out.returncode, (out.stderr or "")[-300:]), flush=True)
            if out.returncode != 0 and "Memory access fault" in (out.stderr or ""):
                raise SystemExit("GPU fault in %s world=%d S=%d: stopping" % (solver, world, S))
