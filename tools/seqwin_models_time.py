"""samples/s of NFM_MODE_SEQUENTIAL on BASELINE configs[3] / [4]'s models -- cfg4: field-aware, 16 fields x one entry, k = 8;
cfg5: degree 3 with explicit lower orders, 32 entries, k = 8 -- the one-workgroup kernel (NFM_SEQ_WIN=0) against the
dependency window (win_worker_ffm / win_worker_fmx).  usage: python tools/seqwin_models_time.py [n] [cfg4,cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import nimfm_amd as nf
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg4", "cfg5"]
dev = torch.device("cuda", 0)
ctx = nf.default_context()
for name in names:
  wl = bench.WORKLOADS[name]
  X, *_keep = bench.make_dataset(torch, nf, ctx, dev, wl, n, 0)
  y = np.random.default_rng(0).standard_normal(n)
  X.set_targets(y)
  for solver in ("sgd", "adagrad"):
    for win, nn in (("0", min(n, 20_000)), ("2", n)):
        os.environ["NFM_SEQ_WIN"] = win
        if wl.get("fields"):
            fm = nf.newFieldAwareFactorizationMachine("regression", nComponents=wl["k"], randomState=1, warmStart=True)
        else:
            fm = nf.newFactorizationMachine("regression", degree=wl["degree"], nComponents=wl["k"], randomState=1, warmStart=True)
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
        print("%s rows  %-8s %s: %.3f us per sample = %.3g samples/s" % (name, solver, "window" if win == "2" else "one workgroup", dt / nn * 1e6, nn / dt), flush=True)
