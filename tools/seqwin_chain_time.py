"""the conductor's chain wavefront, per sample (NFM_SEQ_WIN_TRACE stamps 6 = chain starts, 7 = answer posted): how long the
chain itself takes and how long it waits between two samples, on rows that share nothing (the conductor alone is the bound).
usage: python tools/seqwin_chain_time.py [n] [loss]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
os.environ["NFM_SEQ_WIN"] = "2"
for m, k in ((64, 64), (32, 64), (16, 64)):
    d = n * m
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
    indices = torch.arange(n * m, device=dev, dtype=torch.int32)
    data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
    for loss in ("logistic", "squared"):
        y = np.sign(np.random.default_rng(0).standard_normal(n))
        X.set_targets(y)
        for solver in ("sgd", "adagrad"):
            task = "classification" if loss == "logistic" else "regression"
            fm = nf.newFactorizationMachine(task, nComponents=k, randomState=1, warmStart=True)
            fm.init(X)
            opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=1, loss=loss, verbose=0, tol=0, shuffle=False, mode="sequential")
            opt._handle(fm, ctx, "sequential")
            os.environ.pop("NFM_SEQ_WIN_TRACE", None)
            opt._epoch(X, None, 0, n); opt.it += n
            path = "/tmp/seqwin_trace.bin"
            os.environ["NFM_SEQ_WIN_TRACE"] = "1"; os.environ["NFM_SEQ_WIN_TRACE_FILE"] = path
            opt._epoch(X, None, 0, n); opt.it += n
            ctx.synchronize()
            t = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64) / 100.0
            t = t[1000:-1000]
            print("m=%2d %-8s %-7s: per sample %.3f us = chain %.3f + between samples %.3f; fetched -> chain starts %.2f" % (
                m, loss, solver, np.mean(np.diff(t[:, 7])), np.mean(t[:, 7] - t[:, 6]), np.mean(t[1:, 6] - t[:-1, 7]), np.mean(t[:, 6] - t[:, 5])), flush=True)
            del opt, fm
