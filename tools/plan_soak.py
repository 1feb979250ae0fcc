"""Soak: batch plans by bucketing against plans by the device-wide sort (NFM_PLAN_SEG=0) on random shapes -- sparse and
dense batches, ragged rows, skewed feature popularity, host permutations -- training runs compared bit for bit.
usage: python tools/plan_soak.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nimfm_amd as nf

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = nf.default_context()
bad = used = 0
for c in range(cases):
    n = int(rng.integers(3000, 40000))
    d = int(10 ** rng.uniform(2.5, 5.5))
    mmax = int(rng.integers(4, 70))
    zipf = rng.random() < 0.3
    rows, indptr = [], [0]
    for i in range(n):
        m = int(rng.integers(0 if i % 7 == 0 else 1, min(mmax, d) + 1))
        if zipf:
            idx = np.unique(np.minimum((rng.zipf(1.3, size=m) - 1), d - 1))
        else:
            idx = rng.choice(d, size=m, replace=False)
        if i % 2:
            idx = np.sort(idx)
        rows.append(idx)
        indptr.append(indptr[-1] + len(idx))
    indices = np.concatenate(rows).astype(np.int64)
    data = rng.uniform(-1, 1, len(indices))
    y = rng.standard_normal(n)
    batch = int(rng.choice([512, 1000, 2048, 4096, 8192]))
    k = int(rng.choice([2, 4, 8, 16, 33, 64]))
    kind = str(rng.choice(["sgd", "adagrad"]))
    order = str(rng.choice(["fixed", "host"]))
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    perms = [rng.permutation(n).astype(np.int64) for _ in range(2)] if order == "host" else None
    res = {}
    for seg in ("1", "0"):
        os.environ["NFM_PLAN_SEG"] = seg
        X = nf.newCSRDataset(data, indices, np.array(indptr, dtype=np.int64), n, d)
        fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
        fm.set_params(P0, w0, 0.0)
        kw = dict(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch, shuffle=False)
        ctx.timing_enable(True)
        ctx.timing_reset()
        if kind == "sgd":
            opt = nf.newSGD(eta0=1e-3, touchCap=4.0, **kw)
        else:
            opt = nf.newAdaGrad(**kw)
        try:
            opt.fit(X, y, fm, perms=perms)
        except ValueError as e:  # (a repeated id after the zipf clamp cannot happen: np.unique)
            res[seg] = ("error", str(e))
            continue
        nseg = ctx.timing_get("plan_seg")[0]
        ctx.timing_enable(False)
        res[seg] = (np.array(fm.P).copy(), np.array(fm.w).copy(), fm.intercept, list(opt.history), nseg)
    a, b = res["1"], res["0"]
    if isinstance(a[0], str) or isinstance(b[0], str):
        same = isinstance(a[0], str) and isinstance(b[0], str)
        print("case %d: %s / %s" % (c, a, b), flush=True)
    else:
        same = np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)) and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)) and a[2] == b[2] and a[3] == b[3]
        used += 1 if a[4] else 0
        print("case %2d: n=%d d=%d m<=%d %s B=%d k=%d %s %s: %s (bucketing used: %s)" % (
            c, n, d, mmax, "zipf" if zipf else "uniform", batch, k, kind, order, "bit-equal" if same else "DIFFERENT", bool(a[4])), flush=True)
    bad += 0 if same else 1
print("%d cases, %d different, bucketing used in %d" % (cases, bad, used))
sys.exit(1 if bad else 0)
