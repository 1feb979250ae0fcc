"""where a dependency's turnaround goes in the window kernel (seqwin.hip): the per-sample stamps (NFM_SEQ_WIN_TRACE=1) of a run
over data in which sample t shares one feature with sample t - 1, averaged.  usage: python tools/seqwin_trace.py [sgd|adagrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nimfm_amd as nf
solver = sys.argv[1] if len(sys.argv) > 1 else "sgd"
n, m, k, delta = 20000, 64, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
ctx = nf.Context(0); nf.set_default_context(ctx)
d = n * m + delta
indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
idx = torch.arange(n * m, device=dev, dtype=torch.int64).reshape(n, m)
if delta > 0:
    idx[delta:, 0] = idx[:-delta, 1]
indices = idx.reshape(-1).to(torch.int32)
data = torch.rand(n * m, device=dev, dtype=torch.float64) * 2 - 1
X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), keep=(indptr, indices, data))
y = np.sign(np.random.default_rng(0).standard_normal(n))
fm = nf.newFactorizationMachine("classification", nComponents=k, randomState=1, warmStart=True)
fm.init(X)
os.environ["NFM_SEQ_WIN"] = "2"
mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
opt = mk(maxIter=1, loss="logistic", verbose=0, tol=0, shuffle=False, mode="sequential")
opt.fit(X, y, fm)
path = "/tmp/seqwin_trace.bin"
os.environ["NFM_SEQ_WIN_TRACE"] = "1"; os.environ["NFM_SEQ_WIN_TRACE_FILE"] = path
opt.fit(X, y, fm)
t = np.fromfile(path, dtype=np.int64).reshape(-1, 8).astype(np.float64) / 100.0  # us
t = t[200:-200]
names = ["taken up", "deps resolved", "posted", "dL received", "rows written", "fetched", "chain starts", "answer posted"]
print("%s, sample t shares a feature with t-%d: per sample %.2f us (answer to answer)" % (solver, delta, np.mean(np.diff(t[:, 7]))))
ans_prev = np.roll(t[:, 7], delta)[delta:]
cur = t[delta:]
print("  answer(t-%d) -> deps resolved(t)   %.2f" % (delta, np.mean(cur[:, 1] - ans_prev)))
print("    of which answer(t-%d) -> its dL seen by t   %.2f" % (delta, np.mean(cur[:, 0] - ans_prev)))
print("  deps resolved -> posted            %.2f" % np.mean(cur[:, 2] - cur[:, 1]))
print("  posted -> fetched                  %.2f" % np.mean(cur[:, 5] - cur[:, 2]))
print("  fetched -> chain starts            %.2f" % np.mean(cur[:, 6] - cur[:, 5]))
print("  chain starts -> answer posted      %.2f" % np.mean(cur[:, 7] - cur[:, 6]))
print("  answer posted -> dL received (own) %.2f" % np.mean(cur[:, 3] - cur[:, 7]))
print("  dL received -> rows written        %.2f" % np.mean(cur[:, 4] - cur[:, 3]))
print("  taken up -> deps resolved          %.2f" % np.mean(cur[:, 1] - cur[:, 0]))
