"""time of nfm_opt_predict_all_with_grad (pgd.predictAllWithGrad) on cfg2's shape; prints per-kernel-family times"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
n, d, m, k = 1_000_000, 100_000, 32, 16
indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
y = np.sign(np.random.default_rng(0).standard_normal(n))
fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1)
fm.init(X)
nf.predictAllWithGrad(X, y, fm, loss="logistic")
ctx.timing_reset(); ctx.timing_enable(True)
t0 = time.perf_counter()
for _ in range(3):
    nf.predictAllWithGrad(X, y, fm, loss="logistic")
dt = (time.perf_counter() - t0) / 3
fam = {f: ctx.timing_get(f) for f in ("plan_build", "row_phase", "col_phase", "heavy_partial", "heavy_apply")}
print("predictAllWithGrad n=%d d=%d m=%d k=%d: %.1f ms per call (host wall, incl. 24 MB of outputs back);" % (n, d, m, k, dt * 1e3),
      {f: round(v[1] / max(v[0], 1), 3) for f, v in fam.items()}, "ms per launch")
