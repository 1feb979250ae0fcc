import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
torch.zeros(4, device="cuda").sum().item()
import numpy as np
import nimfm_amd as nf
from nimfm_amd import dp
ctx = nf.Context(0)
g = dp.Group.rccl(ctx, dp.Group.unique_id(), 0, 1)
print("group", g.info())
from common import random_csr
Xo = random_csr(2000, 100, 8, seed=1)
X = nf.CSRDataset(Xo.data, Xo.indices, Xo.indptr, 2000, 100, ctx=ctx)
fm = nf.newFactorizationMachine("regression", nComponents=8)
opt = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=128)
opt.setDataParallel(g, 2, True)
opt.fit(X, np.random.default_rng(0).standard_normal(2000), fm)
print("ok", opt.history, g.info())
os.system("cat /proc/%d/maps | grep -E 'rccl|amdhip64' | awk '{print $6}' | sort -u" % os.getpid())
