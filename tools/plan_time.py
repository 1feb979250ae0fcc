import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
for (n, d, m, k, B) in [(1_000_000, 100_000, 32, 16, 32768), (2_000_000, 1_000_000, 64, 64, 8192)]:
    indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
    y = np.sign(np.random.default_rng(0).standard_normal(n))
    fm = nf.newFactorizationMachine("classification", nComponents=k, warmStart=True, randomState=1); fm.init(X)
    opt = nf.newSGD(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
    X.set_targets(y); opt._handle(fm, ctx, "minibatch")
    for _ in range(3): opt._epoch(X, None, 0, n); opt.it += n
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): opt._epoch(X, None, 0, n); opt.it += n
    ctx.synchronize(); t_plain = (time.perf_counter() - t0) / 5
    rng = np.random.default_rng(1)
    perms = [rng.permutation(n).astype(np.int64) for _ in range(4)]
    opt._epoch(X, perms[0], 0, n); opt.it += n
    ctx.synchronize(); t0 = time.perf_counter()
    for p in perms[1:]: opt._epoch(X, p, 0, n); opt.it += n
    ctx.synchronize(); t_perm = (time.perf_counter() - t0) / 3
    print("n=%d m=%d k=%d B=%d: epoch %.2f ms (plan reused), %.2f ms with a fresh permutation" % (n, m, k, B, t_plain * 1e3, t_perm * 1e3), flush=True)
