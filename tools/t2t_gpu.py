"""bench.py's time_to_target leg alone, over a grid of its knobs (planted model, step size, regularisation, training
samples): which setting lets 1 / 3 / 10 epochs in the reference's order close a real share of the gap to the planted
model's loss, and whether the mini-batch rule gets there.
usage: python tools/t2t_gpu.py WORKLOAD 'json-overrides' ['json-overrides' ...]
  e.g. python tools/t2t_gpu.py headline '{"n_t": 2000000, "planted_P": 0.05}' '{"n_t": 2000000, "sgd": {"eta0": 0.04, "alpha0": 1e-6, "alpha": 1e-5, "beta": 1e-5}}'"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import nimfm_amd as nf

name = sys.argv[1]
grids = [json.loads(a) for a in sys.argv[2:]] or [{}]
dev = torch.device("cuda", 0)
ctx = nf.Context(0)
nf.set_default_context(ctx)
wl = dict(bench.WORKLOADS[name])
n = max(int(g.get("n_t", 0)) for g in grids)
n = (n or (1_000_000 if not wl.get("fields") and wl["degree"] == 2 else 400_000)) + 200_000
X, indptr, indices, data, fields = bench.make_dataset(torch, nf, ctx, dev, wl, n, 0)
task = "classification" if wl["loss"] in ("logistic", "squared_hinge") else "regression"
for g in grids:
    g = dict(g)
    batches = g.pop("batches", None)
    batch = g.pop("batch", wl["batch"])
    cap = float(g.pop("cap", 16.0))
    t = bench.time_to_target_leg(torch, nf, ctx, dev, wl, name, n, batch, cap, indices, data, fields, task, cfg=g, batches=batches)
    print(json.dumps({"cfg": g, "batch": batch, "cap": cap, "start": t["held_out_loss_at_start"], "planted": t["planted_model_held_out_loss"],
                      "seq": [(s["epochs"], s["seconds"], s["held_out_loss"], s["gap_closed"]) for s in t["sequential"]],
                      "mb": [(r["batch"], r["epochs_run"], r["held_out_loss"], r["seconds_per_epoch"],
                              [(h["epochs"] if h["reached"] else None, h["speedup"]) for h in r["targets"]]) for r in t["minibatch"]]}), flush=True)
