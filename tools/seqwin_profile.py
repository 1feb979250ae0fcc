"""where the one-term window spends its time on REAL row shapes (bench data, not the dependency-free / chained toy rows):
per-stage stamps (NFM_SEQ_WIN_TRACE=1) of the register-resident worker on the headline shape, by worker count.
usage: python tools/seqwin_profile.py [n] [W,W] [sgd,adagrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
from bench import gen_shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
Ws = [int(w) for w in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128]
solvers = sys.argv[3].split(",") if len(sys.argv) > 3 else ["sgd", "adagrad"]
d, m, k = (100_000, 32, 16) if os.environ.get("PROFILE_CFG2") else (1_000_000, 64, 64)  # cfg2's row shape: the LDS-resident worker
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
WL = os.environ.get("PROFILE_WL")  # cfg4 / cfg5: bench.py's field-aware / degree-3 workloads (the four-wavefront workers)
if WL:
    import bench
    wl_ = bench.WORKLOADS[WL]
    d, m, k = wl_["d"], wl_["m"], wl_["k"]
    X, indptr, indices, data, _kf = bench.make_dataset(torch, nf, ctx, dev, wl_, n, 0)
elif os.environ.get("PROFILE_NODEP"):  # no two samples share a feature: the conductor + hand-offs alone
    d = n * m
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
    indices = torch.arange(n * m, device=dev, dtype=torch.int32)
    data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
    torch.cuda.synchronize()
else:
    indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
if not WL:
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
y = np.sign(np.random.default_rng(0).standard_normal(n))
path = "/tmp/seqwin_trace.bin"
for solver in solvers:
    for W in Ws:
        os.environ["NFM_SEQ_WIN"] = "2"; os.environ["NFM_SEQ_WIN_W"] = str(W)
        os.environ.pop("NFM_SEQ_WIN_TRACE", None)
        if WL and wl_.get("fields"):
            fm = nf.newFieldAwareFactorizationMachine("classification", nComponents=k, randomState=1, warmStart=True)
        else:
            fm = nf.newFactorizationMachine("classification", degree=wl_["degree"] if WL else 2, nComponents=k, randomState=1, warmStart=True)
        fm.init(X)
        mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
        opt = mk(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential")
        opt.fit(X, y, fm)
        os.environ["NFM_SEQ_WIN_TRACE"] = "1"; os.environ["NFM_SEQ_WIN_TRACE_FILE"] = path
        opt.fit(X, y, fm)
        os.environ.pop("NFM_SEQ_WIN_TRACE", None)
        raw = np.fromfile(path, dtype=np.int64).reshape(-1, 8)
        print("  shader clock over the launch: %.0f MHz" % (100.0 * raw[-1, 0] / max(raw[-1, 1], 1)))
        kind = (raw[:, 1] & 15).copy()
        raw[:, 1] >>= 4
        t = raw.astype(np.float64) / 100.0  # us
        lo, hi = 4 * W, len(t) - 4 * W
        T, K = t[lo:hi], kind[lo:hi]
        gap = np.diff(t[lo - 1:hi, 7])  # answer to answer
        nxt = t[lo + W:hi + W, 0] - T[:, 4]  # rows written -> the worker's next sample taken up (only plain samples stamp [0] at take-up)
        print("%s W=%d: %.3f us per sample (answer to answer) = %.3g samples/s" % (solver, W, gap.mean(), 1e6 / gap.mean()))
        print("  conductor: gaps > 1 us: %.2f %% of samples, %.1f %% of the time; median gap %.3f us" %
              (100 * (gap > 1).mean(), 100 * gap[gap > 1].sum() / gap.sum(), np.median(gap)))
        big = gap > 1.0
        print("  of the samples the conductor waited > 1 us for: %s" % ", ".join("%s %.0f %%" % (nm, 100 * (sel_[big]).mean()) for nm, sel_ in
              (("plain", K == 0), ("far only", K == 1), ("waited for a dL", (K & 4) != 0), ("affine only", (K & 12) == 8), ("near, exact", (K & 14) == 2))))
        late = T[big]
        print("    their worker: taken up -> resolved %.2f, resolved -> posted %.2f, posted -> fetched %.2f, fetched -> answered %.2f; taken up %.2f us before the answer"
              % ((late[:, 1] - late[:, 0]).mean(), (late[:, 2] - late[:, 1]).mean(), (late[:, 5] - late[:, 2]).mean(), (late[:, 7] - late[:, 5]).mean(), (late[:, 7] - late[:, 0]).mean()))
        # what made their workers late: the same worker's PREVIOUS sample (v - W)
        idx_late = np.nonzero(big)[0] + lo
        prev = t[idx_late - W]
        kprev = kind[idx_late - W]
        cur = t[idx_late]
        print("    the same worker's previous sample: answered %.2f us before this answer; answered -> dL seen %.2f, dL seen -> written %.2f, "
              "written -> this one taken up %.2f; kinds: plain %.0f %%, far %.0f %%, waited for a dL %.0f %%, affine only %.0f %%" %
              ((cur[:, 7] - prev[:, 7]).mean(), (prev[:, 3] - prev[:, 7]).mean(), (prev[:, 4] - prev[:, 3]).mean(), (cur[:, 0] - prev[:, 4]).mean(),
               100 * (kprev == 0).mean(), 100 * (kprev == 1).mean(), 100 * ((kprev & 4) != 0).mean(), 100 * ((kprev & 12) == 8).mean()))
        print("    this sample: taken up -> resolved %.2f (median %.2f), resolved -> posted %.2f (median %.2f)" %
              ((cur[:, 1] - cur[:, 0]).mean(), np.median(cur[:, 1] - cur[:, 0]), (cur[:, 2] - cur[:, 1]).mean(), np.median(cur[:, 2] - cur[:, 1])))
        for name, sel in (("plain", K == 0), ("far wait only", K == 1), ("near, exact rows only", (K & 14) == 2),
                          ("waited for a writer's dL", (K & 4) != 0), ("affine only", (K & 12) == 8)):
            if sel.sum() == 0:
                continue
            S = T[sel]
            print("  %-28s %5.1f %%: resolved->posted %.2f, posted->fetched %.2f, fetched->answered %.2f, answered->dL seen %.2f, dL seen->written %.2f; "
                  "taken/dL-of-writer seen->resolved %.2f; stall the conductor took before it %.2f" %
                  (name, 100 * sel.mean(), (S[:, 2] - S[:, 1]).mean(), (S[:, 5] - S[:, 2]).mean(), (S[:, 7] - S[:, 5]).mean(),
                   (S[:, 3] - S[:, 7]).mean(), (S[:, 4] - S[:, 3]).mean(), (S[:, 1] - S[:, 0]).mean(), gap[sel].mean()))
        plain = K == 0
        print("  plain samples: taken up -> written %.2f us; written -> next taken up %.2f us (worker cycle %.2f = W x %.3f)" %
              ((T[plain, 4] - T[plain, 0]).mean(), np.nanmean(nxt[plain & (kind[lo + W:hi + W] == 0)]),
               (t[lo + W:hi + W, 4] - T[:, 4]).mean(), (t[lo + W:hi + W, 4] - T[:, 4]).mean() / W), flush=True)
