"""Where a shuffled epoch's extra time goes (headline shape by default): wall clock of fixed-order epochs, of epochs over
a fresh device-drawn order with the next plan built beside the epoch, and -- NFM_PLAN_PREFETCH=0 -- with the plan built in
line and every launch timed (per-family HIP events).   usage: python tools/shuffle_cost.py [workload] [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import nimfm_amd as nf  # noqa: E402
from nimfm_amd import _capi as capi  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
wl = bench.WORKLOADS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
dev = torch.device("cuda:0")
ctx = nf.default_context()
X, *_keep = bench.make_dataset(torch, nf, ctx, dev, wl, n, 0)
y = np.sign(np.random.default_rng(1).standard_normal(n))
X.set_targets(y)
fm = nf.newFactorizationMachine("classification", degree=wl["degree"], nComponents=wl["k"], warmStart=True, randomState=1)
fm.init(X)
kw = dict(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch", batch=wl["batch"])
opt = nf.newSGD(touchCap=16.0, **kw) if wl["solver"] == "sgd" else nf.newAdaGrad(**kw)
opt._handle(fm, ctx, "minibatch")


def epochs(k):
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        opt._epoch(X, None, 0, n)
        opt.it += n
    ctx.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


epochs(2)
print("fixed order           %8.2f ms per epoch" % epochs(4), flush=True)
capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, 777))
epochs(2)
print("fresh order per epoch %8.2f ms per epoch (NFM_PLAN_PREFETCH=%s)" % (epochs(4), os.environ.get("NFM_PLAN_PREFETCH", "1")), flush=True)
ctx.timing_enable(True)
ctx.timing_reset()
k = 3
ms = epochs(k)
print("  with every launch timed: %.2f ms per epoch" % ms)
for fam in ("plan_build", "plan_seg", "row_phase", "col_phase", "singles", "schedule", "heavy_partial", "heavy_apply"):
    c, t = ctx.timing_get(fam)
    if c:
        print("  %-14s %7d launches %9.2f ms per epoch" % (fam, c // k, t / k))
