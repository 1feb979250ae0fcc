"""samples/s of NFM_MODE_SEQUENTIAL as a dependency window (seqwin.hip) on cfg2's and the headline's row shape, by
worker count; NFM_SEQ_WIN=0 is the one-workgroup kernel.  usage: python tools/seqwin_time.py [n] [W,W,...] [shapes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
Ws = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 32, 64, 128]
shapes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["cfg2", "headline"]
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
SH = {"cfg2": (100_000, 32, 16), "headline": (1_000_000, 64, 64), "nodep64": (0, 64, 64), "nodep32": (0, 32, 16),
      "dep1_64": (-1, 64, 64), "dep4_64": (-4, 64, 64), "dep16_64": (-16, 64, 64), "dep1_32": (-1, 32, 16)}
for name in shapes:
    d, m, k = SH[name]
    if d < 0:  # sample t shares exactly ONE feature with sample t - delta: time per sample x delta = the turnaround of a dependency
        delta = -d
        d = n * m + delta
        indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
        idx = torch.arange(n * m, device=dev, dtype=torch.int64).reshape(n, m)
        idx[delta:, 0] = idx[:-delta, 1]
        indices = idx.reshape(-1).to(torch.int32)
        data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
    elif d == 0:  # no two samples share a feature: what the conductor alone sustains
        d = n * m
        indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
        indices = torch.arange(n * m, device=dev, dtype=torch.int32)
        data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
    else:
        indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
    y = np.sign(np.random.default_rng(0).standard_normal(n))
    for solver in ("sgd", "adagrad"):
        for W in Ws:
            os.environ["NFM_SEQ_WIN"] = "0" if W == 0 else "2"
            os.environ["NFM_SEQ_WIN_W"] = str(max(W, 8))
            nn = n if W else min(n, 20_000)
            fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1, warmStart=True)
            fm.init(X)
            mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
            opt = mk(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential", nCalls=-1)
            if W == 0:  # the one-workgroup kernel on a prefix
                Xs = X
            opt.fit(X, y, fm) if W else None  # warm-up: dependency table built, kept for the next epoch (fixed order)
            ctx.timing_reset(); ctx.timing_enable(True)
            t0 = time.perf_counter()
            if W:
                opt.fit(X, y, fm)
            else:
                opt.maxIter = 1
                opt.fit(X, y, fm) if n <= 20_000 else None
            dt = time.perf_counter() - t0
            kt = ctx.timing_get("sequential")
            kd = ctx.timing_get("seq_window_deps")
            ks = ctx.timing_get("seq_window_scales")
            ctx.timing_enable(False)
            if W == 0 and n > 20_000:
                print("%-9s %-8s one-workgroup kernel: skipped at n=%d (run with n <= 20000)" % (name, solver, n), flush=True)
                continue
            print("%-9s %-8s W=%-3d n=%d: kernel %.3f us per sample = %.3g samples/s; wall %.3g samples/s (%.1f ms); deps table %.1f ms; scale chain %.1f ms" %
                  (name, solver, W, n, kt[1] / n * 1e3, n / (kt[1] * 1e-3), n / dt, dt * 1e3, kd[1], ks[1]), flush=True)
