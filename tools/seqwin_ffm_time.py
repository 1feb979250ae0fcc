"""samples/s of NFM_MODE_SEQUENTIAL on a field-aware model (16 fields x one entry, k = 8: BASELINE configs[3]'s row shape):
the one-workgroup kernel (NFM_SEQ_WIN=0) against the dependency window (win_worker_ffm).  usage: python tools/seqwin_ffm_time.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import nimfm_amd as nf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
wl = bench.WORKLOADS["cfg4"]
dev = torch.device("cuda", 0)
ctx = nf.default_context()
X, *_keep = bench.make_dataset(torch, nf, ctx, dev, wl, n, 0)
y = np.random.default_rng(0).standard_normal(n)
X.set_targets(y)
for solver in ("sgd", "adagrad"):
    for win, nn in (("0", min(n, 20_000)), ("2", n)):
        os.environ["NFM_SEQ_WIN"] = win
        fm = nf.newFieldAwareFactorizationMachine("regression", nComponents=wl["k"], randomState=1, warmStart=True)
        fm.init(X)
        mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
        opt = mk(maxIter=1, verbose=0, tol=0, shuffle=False, mode="sequential")
        opt._handle(fm, ctx, "sequential")
        opt._epoch(X, None, 0, nn)
        opt.it += nn
        ctx.synchronize()
        t0 = time.perf_counter()
        opt._epoch(X, None, 0, nn)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("cfg4 rows  %-8s %s: %.3f us per sample = %.3g samples/s" % (solver, "window" if win == "2" else "one workgroup", dt / nn * 1e6, nn / dt), flush=True)
