import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
n, d, m, k, B = 300_000, 100_000, 32, 16, 8192
indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
rng = np.random.default_rng(0)
planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
planted.set_params(rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.1, 0.0)
y = np.sign(planted.decisionFunction(X))
free0 = torch.cuda.mem_get_info()[0]
for solver in ("sgd", "adagrad"):
    fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1)
    opt = (nf.newSGD(maxIter=60, loss="logistic", verbose=0, tol=0, shuffle=True, mode="minibatch", batch=B, eta0=0.05)
           if solver == "sgd" else nf.newAdaGrad(maxIter=60, loss="logistic", verbose=0, tol=0, shuffle=True, mode="minibatch", batch=B))
    t0 = time.perf_counter()
    opt.fit(X, y, fm)
    dt = time.perf_counter() - t0
    losses = [h[1] for h in opt.history]
    free1 = torch.cuda.mem_get_info()[0]
    print("%s: 60 shuffled epochs in %.2f s, loss %.4f -> %.4f, accuracy %.4f, device memory delta %.1f MB" % (
        solver, dt, losses[0], losses[-1], fm.score(X, y), (free0 - free1) / 1e6), flush=True)
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
# MBPSGD: shuffled stream with wrap-around, both step paths (d > 16384: row-parallel passes)
fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1)
opt = nf.newMBPSGD(maxIter=20, eta0=0.5, gamma=1e-6, loss="logistic", miniBatchSize=4096, verbose=0, tol=-1.0)
t0 = time.perf_counter()
opt.fit(X, y, fm)
dt = time.perf_counter() - t0
losses = [h[1] for h in opt.history]
free1 = torch.cuda.mem_get_info()[0]
print("mbpsgd: 20 outer iterations (74 mini-batches of 4096 each) in %.2f s, loss %.4f -> %.4f, accuracy %.4f, zeros in P %.1f %%, "
      "device memory delta %.1f MB" % (dt, losses[0], losses[-1], fm.score(X, y), 100.0 * (fm.P == 0).mean(), (free0 - free1) / 1e6), flush=True)
assert np.isfinite(losses).all() and losses[-1] < losses[0]
