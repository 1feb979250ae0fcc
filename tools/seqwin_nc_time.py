"""the window WITHOUT a conductor (fitIntercept = false, seqwin.hip `no_cond`): samples/s by worker count on data without /
with dependencies, and where a worker's time per sample goes (NFM_SEQ_WIN_TRACE stamps).
usage: python tools/seqwin_nc_time.py [n] [W,W,...] [shapes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
Ws = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128]
shapes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["nodep64", "headline", "cfg2"]
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
SH = {"cfg2": (100_000, 32, 16), "headline": (1_000_000, 64, 64), "nodep64": (0, 64, 64), "nodep32": (0, 32, 16)}
os.environ["NFM_SEQ_WIN"] = "2"
for name in shapes:
    d, m, k = SH[name]
    if d == 0:
        d = n * m
        indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
        indices = torch.arange(n * m, device=dev, dtype=torch.int32)
        data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
    else:
        indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
    y = np.sign(np.random.default_rng(0).standard_normal(n))
    X.set_targets(y)
    for solver in ("sgd", "adagrad"):
        for W in Ws:
            os.environ["NFM_SEQ_WIN_W"] = str(W)
            os.environ.pop("NFM_SEQ_WIN_TRACE", None)
            fm = nf.newFactorizationMachine("classification", nComponents=k, fitIntercept=False, randomState=1, warmStart=True)
            fm.init(X)
            opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential")
            opt._handle(fm, ctx, "sequential")
            opt._epoch(X, None, 0, n); opt.it += n
            ctx.synchronize()
            t0 = time.perf_counter()
            opt._epoch(X, None, 0, n); opt.it += n
            ctx.synchronize()
            dt = time.perf_counter() - t0
            line = "%-9s %-7s W=%3d: %.3g samples/s (%.2f us per sample per worker)" % (name, solver, W, n / dt, dt / n * W * 1e6)
            if k == 64 and m <= 64:  # the register-resident worker carries the stamps
                path = "/tmp/seqwin_trace.bin"
                os.environ["NFM_SEQ_WIN_TRACE"] = "1"; os.environ["NFM_SEQ_WIN_TRACE_FILE"] = path
                opt._epoch(X, None, 0, n); opt.it += n
                ctx.synchronize()
                t = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64) / 100.0
                t = t[4 * W:-4 * W]
                gap = t[W:, 0] - t[:-W, 4]
                line += " | taken->deps %.2f, deps->posted %.2f, posted->dL %.2f, dL->written %.2f, written->next taken %.2f us" % (
                    np.mean(t[:, 1] - t[:, 0]), np.mean(t[:, 2] - t[:, 1]), np.mean(t[:, 3] - t[:, 2]), np.mean(t[:, 4] - t[:, 3]), np.mean(gap))
            print(line, flush=True)
            del opt, fm
