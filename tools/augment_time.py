import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
n, d, m, k, B = 300_000, 100_000, 32, 8, 32768
indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
y = np.sign(np.random.default_rng(0).standard_normal(n))
for degree, lower in [(2, "explicit"), (2, "augment"), (3, "explicit"), (3, "augment"), (4, "augment")]:
    fm = nf.newFactorizationMachine("classification", degree=degree, nComponents=k, fitLower=lower, fitLinear=(lower != "augment" or degree > 2), randomState=1)
    if lower == "augment" and degree == 2:
        fm = nf.newFactorizationMachine("classification", degree=2, nComponents=k, fitLower="augment", fitLinear=False, randomState=1)
    fm.init(X)
    opt = nf.newSGD(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    X.set_targets(y); opt._handle(fm, ctx, "minibatch")
    for _ in range(2): opt._epoch(X, None, 0, n); opt.it += n
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(3): opt._epoch(X, None, 0, n); opt.it += n
    ctx.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("degree %d fitLower %-8s nAug %d nOrders %d: %.2f ms/epoch, %.3g samples/s" % (degree, lower, fm.nAugments, fm.nOrders, dt * 1e3, n / dt), flush=True)
