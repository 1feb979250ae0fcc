"""time per step of NFM_MODE_SEQUENTIAL (the reference's exact per-sample order) on cfg2's and the headline's row shape;
run once with NFM_SEQ_PIPE=0 and once without to compare the staged and the pipelined kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
for name, n, d, m, k, solver in (("cfg2 shape", 40_000, 100_000, 32, 16, "sgd"), ("headline shape", 20_000, 1_000_000, 64, 64, "sgd"),
                                 ("cfg2 shape", 40_000, 100_000, 32, 16, "adagrad"), ("cfg3 shape", 20_000, 1_000_000, 64, 64, "adagrad")):
    indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
    y = np.sign(np.random.default_rng(0).standard_normal(n))
    fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1, warmStart=True)
    fm.init(X)
    mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
    opt = mk(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential")
    opt.fit(X, y, fm)  # warm-up
    ctx.timing_reset(); ctx.timing_enable(True)
    t0 = time.perf_counter()
    opt.fit(X, y, fm)
    dt = time.perf_counter() - t0
    kt = ctx.timing_get("sequential")
    ctx.timing_enable(False)
    print("%-15s %-8s n=%d m=%d k=%d: kernel %.2f us per step (%.3g samples/s); fit() wall %.2f us per step, NFM_SEQ_PIPE=%s" %
          (name, solver, n, m, k, kt[1] / n * 1e3, n / (kt[1] * 1e-3), dt / n * 1e6, os.environ.get("NFM_SEQ_PIPE", "1")))

# field-aware models (sgd_ffm.nim / adagrad_ffm.nim), cfg4's row shape: 16 fields, one entry per field, k = 8
from bench import make_dataset
for solver in ("sgd", "adagrad"):
    n, d, F, k = 10_000, 100_000, 16, 8
    X, *_keep = make_dataset(torch, nf, ctx, dev, dict(d=d, m=F, fields=F), n, 0)
    y = np.random.default_rng(0).standard_normal(n)
    fm = nf.newFieldAwareFactorizationMachine("regression", nComponents=k, randomState=1, warmStart=True)
    fm.init(X)
    mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
    opt = mk(maxIter=1, loss="squared", verbose=0, tol=0, shuffle=False, mode="sequential")
    opt.fit(X, y, fm)
    ctx.timing_reset(); ctx.timing_enable(True)
    opt.fit(X, y, fm)
    kt = ctx.timing_get("sequential")
    ctx.timing_enable(False)
    print("FFM cfg4 shape  %-8s n=%d F=%d k=%d: kernel %.2f us per step (%.3g samples/s)" % (solver, n, F, k, kt[1] / n * 1e3, n / (kt[1] * 1e-3)))
