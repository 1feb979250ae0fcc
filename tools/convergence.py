#!/usr/bin/env python3
"""What the deterministic mini-batch rule costs statistically: held-out accuracy after the same number of epochs,
reference-order sequential SGD (NFM_MODE_SEQUENTIAL) vs NFM_MODE_MINIBATCH at several batch sizes, on a
learnable synthetic problem (labels = sign of a planted FM + noise).  Prints one line per run."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (initialises HIP before libnimfm_hip is loaded)

import nimfm_amd as nf  # noqa: E402
from bench import gen_shard  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = nf.Context(0)
    nf.set_default_context(ctx)
    n, nt, d, m, k = 100_000, 50_000, 2_000, 16, 8
    ip, ix, dv = gen_shard(torch, dev, n + nt, d, m, 7)
    Xall = nf.CSRDataset.from_device(ctx, n + nt, d, (n + nt) * m, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), keep=(ip, ix, dv))
    rng = np.random.default_rng(3)
    planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    planted.set_params(rng.standard_normal((1, k, d)) * 0.3, rng.standard_normal(d) * 0.3, 0.0)
    f = planted.decisionFunction(Xall)
    y = np.sign(f + 0.3 * f.std() * rng.standard_normal(n + nt))
    ipt = ip[n:] - ip[n]
    Xtr = nf.CSRDataset.from_device(ctx, n, d, n * m, ip.data_ptr(), ix.data_ptr(), dv.data_ptr(), keep=(ip, ix, dv))
    ixt, dvt = ix[n * m:].contiguous(), dv[n * m:].contiguous()
    Xte = nf.CSRDataset.from_device(ctx, nt, d, nt * m, ipt.data_ptr(), ixt.data_ptr(), dvt.data_ptr(), keep=(ipt, ixt, dvt))
    ytr, yte = y[:n], y[n:]
    runs = [("sequential", 1, 5, 0.05), ("minibatch", 256, 5, 0.05), ("minibatch", 2048, 5, 0.05), ("minibatch", 8192, 5, 0.05),
            ("minibatch", 2048, 100, 0.05), ("minibatch", 8192, 200, 0.05),
            # the per-coordinate mean makes one step per batch: larger step sizes, constant schedule
            ("minibatch", 2048, 50, 1.0), ("minibatch", 8192, 100, 1.0), ("minibatch", 8192, 100, 4.0), ("minibatch", 32768, 200, 4.0)]
    for mode, batch, epochs, eta0 in runs:
        fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1, scale=0.01)
        opt = nf.newSGD(maxIter=epochs, eta0=eta0, alpha0=1e-6, alpha=1e-5, beta=1e-5, loss="logistic", verbose=0, tol=0,
                        shuffle=False, mode=mode, batch=batch, scheduling="optimal" if eta0 < 0.5 else "constant")
        t0 = time.perf_counter()
        opt.fit(Xtr, ytr, fm)
        dt = time.perf_counter() - t0
        print("%-10s batch %6d eta0 %5.2f: %3d epochs in %7.3f s, train loss %.4f, held-out accuracy %.4f" % (
            mode, batch, eta0, epochs, dt, opt.history[-1][1], fm.score(Xte, yte)), flush=True)


if __name__ == "__main__":
    main()
