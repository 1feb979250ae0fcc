"""Exact order (NFM_MODE_SEQUENTIAL) of degree-2 FMs with 65 ... 128 factors: the table read as two blocks of 64 goes through the
window (seqwin.hip::seq_window_view); NFM_SEQ_WIN=0 is the one-sample-in-flight kernel such models ran on before.
usage: python tools/seqwin_k128_time.py [k] [m] [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import nimfm_amd as nf  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
d = 1_000_000
dev = torch.device("cuda:0")
ctx = nf.default_context()
wl = dict(n=n, d=d, m=m, k=k, degree=2, solver="sgd", loss="logistic", batch=8192)
X, *_keep = bench.make_dataset(torch, nf, ctx, dev, wl, n, 0)
y = np.sign(np.random.default_rng(1).standard_normal(n))
X.set_targets(y)
for solver in ("sgd", "adagrad"):
    for win in ("1", "0"):
        os.environ["NFM_SEQ_WIN"] = win
        fm = nf.newFactorizationMachine("classification", nComponents=k, warmStart=True, randomState=1)
        fm.init(X)
        opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential")
        opt._handle(fm, ctx, "sequential")
        ns = n if win == "1" else min(n, 20_000)
        l0 = ctx.timing_get("seq_window_launch")[0]
        opt._epoch(X, None, 0, ns)
        opt.it += ns
        ctx.synchronize()
        t0 = time.perf_counter()
        opt._epoch(X, None, 0, ns)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("%s k=%d m=%d NFM_SEQ_WIN=%s: %.3g samples/s (window launches %d)" % (solver, k, m, win, ns / dt, ctx.timing_get("seq_window_launch")[0] - l0), flush=True)
