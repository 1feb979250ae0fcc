#!/usr/bin/env python3
"""bench.py -- SGD training samples/s per epoch on synthetic CSR (BASELINE.json metric).

One "step" = one pass of the hot path (nfm_opt_epoch, mini-batch mode) over the rank's whole
synthetic shard, inputs already resident in HBM.  N = 1 runs the north-star headline workload of
BASELINE.json (synthetic CSR 1e7 x 1e6, 64 nnz/row, k = 64, SGD, Logistic loss, mini-batch 8192, 1 MI355X)
and, as `extra.cfg2` ... `extra.cfg5`, the other BASELINE.json configs on one GPU: configs[1] (1e6 x 1e5, 32 nnz/row,
k = 16), [2] (the headline's data with AdaGrad, a 4e6-sample shard), [3] (field-aware, 16 fields, k = 8), [4] (degree 3,
k = 8) -- each with its own roofline, shuffled rate, exact-order rate (NFM_MODE_SEQUENTIAL), time_to_target and a bounded
cpu_baseline; about two minutes in all.
N > 1: one process per GPU (torch.distributed for the bootstrap and the clock; the exchange is the library's own RCCL
communicator), every rank trains on its own shard of the same size (weak scaling), the replicas are reconciled every
sync_period mini-batches on a second stream and exactly at the end of every epoch; the line then
carries `extra.cfg3` (configs[2] as written: AdaGrad, data-parallel).  `--gpus N` without
torchrun's environment starts the N ranks itself (a torch.distributed.run child; this parent process
never touches a GPU).

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
  roofline     achieved algorithmic bytes/s of the per-batch kernel pair vs the 8 TB/s HBM peak,
               durations from HIP events recorded by the library on its own stream
  cpu_baseline the reference-faithful CPU port (oracle/, single thread) on the same workload
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: n, d, m, k, degree, solver, loss   (SURVEY.md 8d / BASELINE.md)
    # cfg2 names no batch size; 32768 keeps the two dependent launches per batch off the critical path
    # (DESIGN.md section 7).  cfg3 uses the 8192 that BASELINE.json states.  The north-star headline (SGD, one GPU) names no batch:
    # 262144 with touch cap 32 since the end of round 5 (131072 / cap 16 before: profiles/r05a_headline_batch_sweep.txt,
    # r05f_headline_batch_sweep_big.txt).  The sweeps at cap 16 found 262144 needing 2 / 4 / 17 epochs to the three targets where
    # 131072 needs 1 / 3 / 8; the cap, not the batch, was the reason: at lambda = B m / d = 16.8 touches per coordinate and batch a
    # cap of 16 already averages half the coordinates' steps.  With cap 32 (about twice lambda, as 16 is for lambda = 8.4) 262144
    # reaches the targets in 1 / 3 / 7 epochs at 8.37e7 samples/s (profiles/r05h_touch_cap_sweep.txt); 524288 / cap 64 adds 1 %.
    # `value_batch_8192` on the line is the figure of rounds 1-4.  cfg2: 65536 / cap 32 (1 / 3 / 10 epochs as at 32768 / 16, +7 %).
    "cfg2": dict(n=1_000_000, d=100_000, m=32, k=16, degree=2, solver="sgd", loss="logistic", batch=65536, touch_cap=32.0),
    "headline": dict(n=10_000_000, d=1_000_000, m=64, k=64, degree=2, solver="sgd", loss="logistic", batch=262144, touch_cap=32.0),
    "cfg3": dict(n=10_000_000, d=1_000_000, m=64, k=64, degree=2, solver="adagrad", loss="squared", batch=8192),
    "cfg3c": dict(n=10_000_000, d=1_000_000, m=64, k=64, degree=2, solver="adagrad", loss="squared", batch=8192, ada_cross=0.1),
    "cfg3l": dict(n=10_000_000, d=1_000_000, m=64, k=64, degree=2, solver="adagrad", loss="logistic", batch=8192),
    # cfg5: higher-order FM, degree 3 with fitLower=explicit -> two parameter blocks (ANOVA degree 3 and 2)
    "cfg5": dict(n=1_000_000, d=100_000, m=32, k=8, degree=3, solver="sgd", loss="squared", batch=32768),
    # cfg4: field-aware FM, 16 fields, one nnz per field (field f owns the indices [f d/F, (f+1) d/F),
    # tests/utils.nim:66-68), AdaGrad
    # batch 2048: field-aware AdaGrad has no touch cap to turn -- a batch is ONE step per coordinate -- and at 32768 it does not
    # reach the held-out loss of ten sequential epochs in 40 of its own (time_to_target, measured in round 4: 32768: 13 / 28 /
    # never; 8192: 4 / 8 / 20 epochs; 2048: 2 / 4 / 11 epochs and the best speed-up); "cfg4big" keeps the bandwidth figure
    # Round 5: batch 32768 again, with the batch's gradient cross products in g_norm (nfm_opt_set_ada_cross, gamma 0.1): the same
    # 2 / 4 / 11 epochs to the three targets as batch 2048 needs, at twice the samples per second ("cfg4b2048": rounds 4 / early 5);
    # 65536 (the end of round 5): 2 / 4 / 12 epochs, every target in fewer seconds than at 32768 (speed-ups 31 / 48 / 55 against
    # 25 / 39 / 48), 6.3e7 samples/s, frac 0.53 (32768: 5.0e7, 0.43; 131072: 7.2e7, 0.60, but 2 / 5 / 14 epochs and no faster to the targets)
    "cfg4": dict(n=1_000_000, d=100_000, m=16, k=8, degree=2, solver="adagrad", loss="squared", batch=65536, fields=16, ada_cross=0.1),
    "cfg4b2048": dict(n=1_000_000, d=100_000, m=16, k=8, degree=2, solver="adagrad", loss="squared", batch=2048, fields=16),
    "cfg4big": dict(n=1_000_000, d=100_000, m=16, k=8, degree=2, solver="adagrad", loss="squared", batch=32768, fields=16),
    # cfg4 with three low-cardinality fields (2, 7 and 50 distinct features): their features are touched by a
    # large share of every batch
    "cfg4lc": dict(n=1_000_000, d=100_000, m=16, k=8, degree=2, solver="adagrad", loss="squared", batch=32768, fields=16,
                   low_card=[2, 7, 50]),
    # the shape of click-through data: 39 fields, one entry per field, k = 4, 13 low-cardinality fields
    "ffm39": dict(n=500_000, d=39 * 25641, m=39, k=4, degree=2, solver="adagrad", loss="logistic", batch=32768, fields=39,
                  low_card=[16] * 13),
    "tiny": dict(n=50_000, d=5_000, m=16, k=8, degree=2, solver="sgd", loss="logistic", batch=4096),
    # more than 128 factors (round 5: the factors of an order as device blocks of at most 128, csrc/common.h ModelView::kc):
    # the headline's data with k = 256 (two blocks of 128: 1 KB rows, 2 GB of parameters) and k = 200 (two blocks of 100)
    # (batch 32768 / cap 16: 1.40e7 samples/s, 0.46 of the roofline; 131072 / 16: 2.22e7, 0.74; 262144 / 32, the headline's setting: 2.41e7, 0.80)
    "wide256": dict(n=2_000_000, d=1_000_000, m=64, k=256, degree=2, solver="sgd", loss="logistic", batch=262144, touch_cap=32.0),
    "wide200a": dict(n=2_000_000, d=1_000_000, m=64, k=200, degree=2, solver="adagrad", loss="squared", batch=8192),
    # cfg2 with Zipf(1.1) feature popularity: a few features are touched by most samples of a batch
    "cfg2z": dict(n=1_000_000, d=100_000, m=32, k=16, degree=2, solver="sgd", loss="logistic", batch=32768, zipf=1.1),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


EXTRA_N = {"cfg3": 4_000_000}  # samples of the extra.cfg3 leg of the default run (the workload line says so)


def algorithmic_bytes_per_sample(solver, m, k, n_orders=1, fields=0):
    """int32 idx, fp64 val, fp64 params."""
    if fields:  # FFM, the reference's touch set (all F rows of every feature of the sample)
        per = 32 if solver == "adagrad" else 16
        return 16 * m + 16 + per * fields * m * k + (32 if solver == "adagrad" else 24) * m
    if solver == "sgd":
        return 12 * m + 16 + n_orders * 16 * m * k + 16 * m + 8 * m
    return 12 * m + 16 + n_orders * 32 * m * k + 32 * m


def gen_shard(torch, dev, n, d, m, seed, zipf=0.0):
    """m distinct column indices per row (sorted), values U(-1,1).  zipf = 0: uniform popularity;
    zipf = s > 0: feature j drawn with probability ~ 1/(j+1)^s (SURVEY.md 8d secondary distribution)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cdf = None
    if zipf > 0:
        pj = 1.0 / torch.arange(1, d + 1, device=dev, dtype=torch.float64) ** zipf
        cdf = torch.cumsum(pj / pj.sum(), 0)

    def draw(rows):
        if cdf is None:
            return torch.randint(0, d, (rows, m), device=dev, generator=g, dtype=torch.int64)
        u = torch.rand((rows, m), device=dev, generator=g, dtype=torch.float64)
        return torch.searchsorted(cdf, u).clamp_(max=d - 1)

    idx, _ = torch.sort(draw(n), dim=1)
    for _ in range(1000):  # re-draw the repeated entries (only those) until every row is distinct
        dup = torch.zeros_like(idx, dtype=torch.bool)
        dup[:, 1:] = idx[:, 1:] == idx[:, :-1]
        k_dup = int(dup.sum())
        if k_dup == 0:
            break
        idx[dup] = draw((k_dup + m - 1) // m).reshape(-1)[:k_dup]
        idx, _ = torch.sort(idx, dim=1)
    else:
        raise SystemExit("could not draw distinct indices")
    val = torch.rand((n, m), device=dev, generator=g, dtype=torch.float64) * 2.0 - 1.0
    indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * m
    out = indptr, idx.to(torch.int32).reshape(-1).contiguous(), val.reshape(-1).contiguous()
    torch.cuda.synchronize(dev)  # the library adopts these arrays on ITS stream (nfm_dataset_create_csr_device): they must be complete
    return out


def write_svmlight_file(path, n, d, m, seed):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for lo in range(0, n, 50_000):
            hi = min(n, lo + 50_000)
            idx = 1 + np.arange(m) * (d // m) + rng.integers(0, d // m, size=(hi - lo, m))  # distinct inside a row
            val = rng.uniform(-1, 1, size=(hi - lo, m))
            y = np.sign(rng.standard_normal(hi - lo))
            f.write("".join(
                repr(float(y[i])) + " " + " ".join("%d:%r" % (idx[i, q], float(val[i, q])) for q in range(m)) + "\n"
                for i in range(hi - lo)))



def ingest_leg(args):
    """--workload ingest: cfg2 as an svmlight text file (n lines x 32 "index:value" entries) loaded by
    nfm_dataset_load_svmlight (bytes -> HBM, parsed on the GPU); cpu_baseline = the C restatement of the
    reference's two-pass loader (oracle/nimfm_ingest.c <- dataset.nim:562-613) on one host core, over a
    bounded prefix of the same file."""
    import tempfile

    import nimfm_amd as nf

    n, m, d = args.n or 1_000_000, 32, 100_000
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(tmp, "cfg2.svm")
    try:
        write_svmlight_file(path, n, d, m, 42)
        nbytes = os.path.getsize(path)
        best = None
        for _ in range(max(1, min(args.steps, 3))):
            t0 = time.perf_counter()
            ds, y = nf.loadSVMLightFile(path)
            wall = time.perf_counter() - t0
            b, up, pa = ds.ingest_stats()
            if best is None or wall < best[0]:
                best = (wall, up, pa)
            del ds
        cpu = None
        if not args.no_cpu_baseline:
            import oracle as O

            with open(path, "rb") as f:
                head = f.read(int(nbytes * min(1.0, 200_000 / n)))
            head = head[: head.rfind(b"\n") + 1]
            t0 = time.perf_counter()
            r = O.svmlight_load_c(head)
            tc = time.perf_counter() - t0
            cpu = {"value": round(len(head) / tc / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                   "sample": "first %d lines of the same file, oracle/nimfm_ingest.c (two passes, strtod)" % len(r["y"])}
        out = {"metric": "svmlight text ingest to CSR in HBM", "value": round(nbytes / best[0] / 1e9, 3), "unit": "GB/s",
               "n_gpus": 1, "steps": args.steps, "warmup": 0, "ms_per_step": round(best[0] * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "ingest: cfg2 as svmlight text, %d lines x %d entries, %d bytes" % (n, m, nbytes)},
               "gpu": {"read_upload_ms": round(best[1], 2), "parse_ms": round(best[2], 2),
                       "parse_only_GBps": round(nbytes / best[2] / 1e6, 2)},
               "cpu_baseline": cpu}
        print(json.dumps(out))
    finally:
        if os.path.exists(path):
            os.remove(path)
        os.rmdir(tmp)


def psgd_leg(args):
    """--workload psgd: mini-batch proximal SGD on a matrix of the shape of the reference's own
    benchmark for this solver (benchmarks/ml100k/sparse_fm_mbpsgd.nim: ml-100k user x item x side features, d = 2703,
    squared loss, SquaredL12, nComponents = 30, default mini-batch = d n / nnz).  One step = one outer iteration
    (maxIterInner mini-batches: gradient, step on all parameters, prox).  cpu_baseline = oracle/nimfm_psgd.c."""
    import torch

    import nimfm_amd as nf

    n, d, m, k = args.n or 90_570, 2_703, 24, 30
    if args.psgd_shape:  # "d,m,k": other model sizes (the one-launch step holds up to 16384 features)
        d, m, k = (int(v) for v in args.psgd_shape.split(","))
    dev = torch.device("cuda", 0)
    ctx = nf.Context(0)
    nf.set_default_context(ctx)
    indptr, indices, data = gen_shard(torch, dev, n, d, m, 42)
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                  keep=(indptr, indices, data))
    rng = np.random.default_rng(1234)
    planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    planted.set_params(rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.1, 0.0)
    y = planted.decisionFunction(X)
    del planted
    B = args.batch or max((d * n) // (n * m), 1)
    inner = (n - 1) // B + 1
    need = B * inner
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True, randomState=1)
    fm.init(X)
    opt = nf.newMBPSGD(maxIter=1, beta=1e-5, alpha0=1e-10, alpha=1e-10, gamma=1e-5, miniBatchSize=B, verbose=0)
    opt.batch, opt.it = B, 1
    X.set_targets(y)
    opt._handle(fm, ctx, "minibatch")
    stream = np.concatenate([np.arange(n, dtype=np.int64), np.arange(need - n, dtype=np.int64)])

    def step():
        ls, _ = opt._epoch(X, stream, 0, need)
        opt.it += inner
        return ls

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    fams = ("plan_build", "row_phase", "col_phase", "heavy_partial", "heavy_apply", "psgd_step")
    ctx.timing_reset()
    ctx.timing_enable(True)
    step()
    ctx.synchronize()
    fam = {f: ctx.timing_get(f) for f in fams}
    ctx.timing_enable(False)
    da = d
    Kp = 2 * max(1, 1 << (((k + 1) // 2) - 1).bit_length())
    dense_bytes = 16 * da * Kp  # every parameter read and written once per mini-batch (shrink + prox)
    step_ms = fam["psgd_step"][1] / max(fam["psgd_step"][0], 1)
    roof = {"bound": "hbm", "kernel": "psgd_step: k_psgd_step_columns (one launch, models of <= 16384 features) or k_psgd_dense + k_prox_pass_* -- "
                      "all parameters, once per mini-batch",
            "achieved": round(dense_bytes / (step_ms * 1e-3) / 1e9, 2) if step_ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(dense_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if step_ms else None, "traffic": None,
            "algorithmic_bytes_per_minibatch": dense_bytes,
            "note": "%d x %d parameters = %.2f MB: the whole model sits in L2 and every kernel is launch-latency bound" % (da, Kp, 8e-6 * da * Kp),
            "avg_ms": {f: round(fam[f][1] / fam[f][0], 5) if fam[f][0] else 0.0 for f in fams},
            "launches_per_step": {f: fam[f][0] for f in fams}}
    cpu = None
    if not args.no_cpu_baseline:
        import oracle as O

        Xo = O.Dataset(indptr.cpu().numpy(), indices.cpu().numpy().astype(np.int64), data.cpu().numpy(), n, d)
        P0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01
        inner_c = max(1, min(inner, 250))
        cfg = O.psgd_cfg(beta=1e-5, alpha0=1e-10, alpha=1e-10, gamma=1e-5)
        tc = time.perf_counter()
        O.fm_mbpsgd_epoch(Xo, y, 2, P0, np.zeros(d), 0.0, cfg, stream[:B * inner_c], B)
        tc = time.perf_counter() - tc
        cpu = {"value": round(B * inner_c / tc, 1), "unit": "samples/s", "cores": 1, "kind": "port",
               "sample": "the first %d mini-batches of the same outer iteration, oracle/nimfm_psgd.c (minibatch_psgd.nim:87-122), "
                         "-O2, 1 thread" % inner_c}
    print(json.dumps({"metric": "MBPSGD training samples/sec/outer iteration", "value": round(need / dt, 1), "unit": "samples/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 3),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "psgd: synthetic CSR %dx%d, %d nnz/row, k=%d, MBPSGD squared loss, SquaredL12, "
                                             "mini-batch %d x %d per outer iteration" % (n, d, m, k, B, inner)},
                      "last_step": {"mean_loss": last / need}, "roofline": roof, "cpu_baseline": cpu}))


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a torch.distributed.run child and leave
    with its exit code.  This parent imports neither torch nor the library and makes no HIP call (a process that has
    initialised the GPU must not fork / exec workers on this pool)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    raise SystemExit(subprocess.call(cmd, env=env))


def make_dataset(torch, nf, ctx, dev, wl, n, rank):
    """the rank's synthetic shard, generated on the device (data seed 42 + rank), adopted without a copy"""
    d, m = wl["d"], wl["m"]
    F = wl.get("fields", 0)
    fields = None
    if F:
        g = torch.Generator(device=dev)
        g.manual_seed(42 + rank)
        per = d // F
        idx = torch.randint(0, per, (n, F), device=dev, generator=g, dtype=torch.int64)
        if wl.get("low_card"):  # the first fields have only a handful of distinct features (gender, weekday ...)
            for f_, card in enumerate(wl["low_card"]):
                idx[:, f_] = torch.randint(0, card, (n,), device=dev, generator=g, dtype=torch.int64)
        idx = idx + torch.arange(F, device=dev) * per
        indices = idx.reshape(-1).to(torch.int32)
        fields = torch.arange(F, device=dev, dtype=torch.int32).repeat(n)
        data = torch.rand((n * F,), device=dev, generator=g, dtype=torch.float64) * 2 - 1
        indptr = torch.arange(n + 1, device=dev, dtype=torch.int64) * F
    else:
        indptr, indices, data = gen_shard(torch, dev, n, d, m, 42 + rank, wl.get("zipf", 0.0))
    torch.cuda.synchronize()
    X = nf.CSRDataset.from_device(ctx, n, d, n * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                  fields_ptr=fields.data_ptr() if F else None, nFields=F,
                                  keep=(indptr, indices, data, fields))
    return X, indptr, indices, data, fields


def cpu_baseline_leg(args, wl, n, d, m, k, n_orders, indptr, indices, data, y, cheap=False, fields=None):
    """the reference-faithful CPU port (oracle/, test infrastructure used here as the reported baseline) on a bounded
    prefix of the same shard: built -O3 -march=native on this host (oracle/Makefile `timing`), one warm-up epoch, the
    median of the next three; Hogwild (optimizer/sgd_multi.nim semantics) at T = 4 (the reference
    benchmarks' value), this GPU's share of the host's cores, and twice that (the reference's default maxThreads)."""
    import oracle as O

    epochs = 4  # one warm-up epoch, then the median of three
    F = wl.get("fields", 0)
    work = (F * m * k) if F else (n_orders * m * k)  # parameter values a sample's step touches
    nc = min(n, args.cpu_samples, max(5_000, int((0.8e9 if cheap else 1.6e9) / (work * epochs))))
    ip = indptr[: nc + 1].cpu().numpy()
    ix = indices[: nc * m].cpu().numpy().astype(np.int64)
    dv = data[: nc * m].cpu().numpy()
    sgd = wl["solver"] == "sgd"
    cfg = O.sgd_cfg(loss=wl["loss"]) if sgd else O.adagrad_cfg(loss=wl["loss"])
    if F:
        fl = fields[: nc * m].cpu().numpy().astype(np.int64)
        Xo = O.Dataset(ip, ix, dv, nc, d, fl, F)
        P0 = np.random.default_rng(1).standard_normal((F, d, k)) * 0.01
    else:
        Xo = O.Dataset(ip, ix, dv, nc, d)
        P0 = np.random.default_rng(1).standard_normal((n_orders, k, d)) * 0.01
    w0 = np.zeros(d)

    check = {}  # the one-thread run's parameters: what the GPU's exact order is compared with (run_training)

    def med(threads):
        if F:
            res_ = (O.ffm_sgd_fit if sgd else O.ffm_adagrad_fit)(Xo, y[:nc], P0, w0, 0.0, cfg, epochs)
        elif sgd:
            res_ = O.fm_sgd_fit(Xo, y[:nc], wl["degree"], P0, w0, 0.0, cfg, epochs, hogwild_threads=threads)
        else:
            res_ = O.fm_adagrad_fit(Xo, y[:nc], wl["degree"], P0, w0, 0.0, cfg, epochs)
        if not threads:
            check.update(P=res_[0], w=res_[1], b=res_[2], P0=P0, nc=nc, epochs=epochs)
        return float(np.median(O.epoch_seconds(epochs)[1:]))  # epoch 0 is the warm-up

    def med_jagged(threads_):
        O.fm_sgd_fit_jagged(Xo, y[:nc], P0, w0, 0.0, cfg, epochs, threads=threads_)
        return float(np.median(O.epoch_seconds(epochs)[1:]))

    jag = None
    with O.variant("timing"):
        if sgd and wl["degree"] == 2 and not cheap and not F:  # the reference's own storage: one heap block per parameter row
            jag = {"value": round(nc / med_jagged(1), 1), "cores": 1,
                   "hogwild_4_threads": round(nc / med_jagged(min(4, os.cpu_count() or 1)), 1),
                   "note": "the same epoch on jagged storage (seq-of-seq as tensor/tensor.nim:8-17: a malloc'd row with a seq header "
                           "per feature behind a pointer table), oracle/nimfm_jagged.c, bit-identical results"}
        t1 = med(0)
        # Hogwild (optimizer/sgd_multi.nim: contiguous slices, shared unsynchronised state) at T = 1, 4 (the reference
        # benchmarks' value), this process's usable CPUs, all physical cores, and the reference's default
        # min(2 x countProcessors(), MaxThreadPoolSize = 256) (sgd_multi.nim:13-18) -- SURVEY 8(d); the line carries the BEST
        logical = os.cpu_count() or 1
        try:
            usable = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            usable = logical
        physical = logical
        try:
            cores_ = set()
            phys_, core_ = None, None
            with open("/proc/cpuinfo") as f:
                for ln in f:
                    if ln.startswith("physical id"):
                        phys_ = ln.split(":", 1)[1].strip()
                    elif ln.startswith("core id"):
                        core_ = ln.split(":", 1)[1].strip()
                    elif not ln.strip() and phys_ is not None and core_ is not None:
                        cores_.add((phys_, core_))
                        phys_, core_ = None, None
            if cores_:
                physical = len(cores_)
        except OSError:
            pass
        ref_default = min(2 * logical, 256)
        threads = min(16, usable)  # the GPU box's CPU share for one GPU
        sweep = []
        if sgd and not F:
            cand = {1, min(4, usable), threads} if cheap else {1, min(4, usable), threads, min(usable, physical), physical, ref_default}
            for T_ in sorted(cand):
                if nc // T_ < 8:  # (a slice per thread: nothing to time below a few samples each)
                    continue
                tt_ = med(T_)
                sweep.append({"threads": T_, "value": round(nc / tt_, 1)})
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"_check": check, "value": round(nc / t1, 1), "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "the first %d samples of the same shard (same d, nnz/row, k); C restatement of optimizer/%s semantics, "
                      "flat arrays, gcc -O3 -march=native on this host, 1 thread, 1 warm-up epoch then the median of %d; "
                      "epoch loop only (the per-fit layout transposes, sgd.nim:292,328, are outside)"
                      % (nc, ("sgd_ffm.nim:49-106" if sgd else "adagrad_ffm.nim:11-66") if F else ("sgd.nim:261-328" if sgd else "adagrad.nim:137-203"), epochs - 1),
            "jagged": jag,
            "hogwild": None if not sweep else dict(max(sweep, key=lambda e_: e_["value"]),
                                                   note="optimizer/sgd_multi.nim semantics (racy), same port and build; the BEST of the "
                                                        "sweep over T (1, 4, usable CPUs, physical cores, the reference's default "
                                                        "min(2 x logical, 256))", sweep=sweep),
            "host": {"cpu": cpu_model, "logical_cpus": os.cpu_count(), "usable_cpus": usable, "physical_cores": physical}}


def c_bar_of(lam_, cap_):
    """mean over the touched coordinates of max(1, c / cap), c ~ Poisson(lam_) given c >= 1: how many times fewer steps a
    coordinate makes per epoch than in the reference's order (cap = 1: lam / (1 - exp(-lam)))"""
    if lam_ < 1e-12:
        return 1.0
    tot, p = 0.0, math.exp(-lam_)
    for c_ in range(1, int(lam_ + 12 * math.sqrt(lam_) + 40)):
        p = p * lam_ / c_
        tot += p * max(1.0, c_ / cap_)
    return tot / (1.0 - math.exp(-lam_))


# time_to_target: the problem BOTH modes are timed on.  The throughput legs keep SURVEY 8(d)'s labels and the reference's
# default step sizes; with those, one epoch in the reference's order moves the held-out loss of the headline shape by 0.1 %
# (nothing is learnt: the `optimal` schedule with beta = 1e-3 has shrunk the step 11-fold after 1e6 samples and the planted
# signal is mostly second-order).  Here: labels from a planted degree-2 FM whose linear part carries most of the signal
# (w ~ N(0, 0.3^2), P ~ N(0, 0.1^2)), regularisation 1e-5, SGD eta0 = 0.02 -- on the CPU restatement at the same samples
# per feature (tests/t2t_explore.py) 1 / 3 / 10 epochs in the reference's order then close 38 / 76 / 115 % of the gap between the
# starting loss and the planted model's own held-out loss.
T2T = {"planted_w": 0.3, "planted_P": 0.1, "seq_epochs": (1, 3, 10), "mb_epoch_cap": 40,
       "sgd": dict(eta0=0.02, alpha0=1e-6, alpha=1e-5, beta=1e-5),
       "adagrad": dict(eta0=0.1, alpha0=1e-6, alpha=1e-5, beta=1e-5)}


# per workload: the shapes with d = 1e6 need 2e6 training samples (128 per feature) and a planted model whose second-order
# part is smaller still before ten sequential epochs close a third of the gap; AdaGrad at eta0 = 0.1 over-fits these
# sample counts with regularisation 1e-5 (held-out loss RISES above its start, measured): eta0 = 0.05 with 3e-5 closes
# 27 / 36 / 44 % of the gap in 1 / 3 / 10 sequential epochs (tools/t2t_gpu.py, gpurun_out of round 4)
T2T_WL = {"headline": dict(n_t=2_000_000, planted_P=0.05, sgd=dict(eta0=0.04, alpha0=1e-6, alpha=1e-5, beta=1e-5)),
          "cfg3": dict(n_t=2_000_000, planted_P=0.05, adagrad=dict(eta0=0.05, alpha0=1e-6, alpha=3e-5, beta=3e-5))}
T2T_WL["cfg3c"] = T2T_WL["cfg3"]


def time_to_target_leg(torch, nf, ctx, dev, wl, name, n, batch, cap, indices, data, fields_t, task, cfg=None, batches=None):
    """Speed to a given held-out loss, mini-batch rule against the reference's order (optimizer/sgd.nim:294-321 run by
    NFM_MODE_SEQUENTIAL in the window kernel): both start from the same parameters with the same step size and schedule on
    the first n_t samples of the shard; the sequential run makes 1, 3 and 10 epochs, its held-out loss after each is a
    target; the mini-batch rule runs epochs (held-out loss after every one) until it is at or below the last target.
    Seconds are training time only (the evaluations are outside both clocks)."""
    from nimfm_amd import _capi as capi

    cfg = dict(dict(T2T, **T2T_WL.get(name, {})), **(cfg or {}))
    d, m, k = wl["d"], wl["m"], wl["k"]
    F = wl.get("fields", 0)
    sgd = wl["solver"] == "sgd"
    hp = dict(cfg["sgd" if sgd else "adagrad"])
    n_h = 200_000
    n_t = min(int(cfg.get("n_t") or (1_000_000 if not F and wl["degree"] == 2 else 400_000)), n - n_h)
    ip_t = torch.arange(n_t + 1, device=dev, dtype=torch.int64) * m
    ip_h = torch.arange(n_h + 1, device=dev, dtype=torch.int64) * m
    off = n_t * m
    Xt = nf.CSRDataset.from_device(ctx, n_t, d, n_t * m, ip_t.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                   fields_ptr=fields_t.data_ptr() if F else None, nFields=F, keep=(ip_t, indices, data, fields_t))
    Xh = nf.CSRDataset.from_device(ctx, n_h, d, n_h * m, ip_h.data_ptr(), indices.data_ptr() + 4 * off, data.data_ptr() + 8 * off,
                                   fields_ptr=fields_t.data_ptr() + 4 * off if F else None, nFields=F,
                                   keep=(ip_h, indices, data, fields_t))
    rng = np.random.default_rng(4321)
    planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    planted.set_params(rng.standard_normal((1, k, d)) * cfg["planted_P"], rng.standard_normal(d) * cfg["planted_w"], 0.0)
    f_t, f_h = planted.decisionFunction(Xt), planted.decisionFunction(Xh)
    del planted
    classify = task == "classification"
    y_t, y_h = (np.sign(f_t), np.sign(f_h)) if classify else (f_t, f_h)
    Xt.set_targets(np.ascontiguousarray(y_t))

    def loss_of(p):
        if wl["loss"] == "logistic":
            z = p * y_h
            return float(np.mean(np.log1p(np.exp(-np.abs(z))) - np.minimum(z, 0.0)))
        if wl["loss"] == "squared_hinge":
            return float(np.mean(np.maximum(0.0, 1.0 - p * y_h) ** 2))
        return float(np.mean(0.5 * (p - y_h) ** 2))

    def new_model():
        if F:
            f_ = nf.newFieldAwareFactorizationMachine(task, nComponents=k, warmStart=True, randomState=1)
        else:
            f_ = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, warmStart=True, randomState=1)
        f_.init(Xt)
        return f_

    mk_ = nf.newSGD if sgd else nf.newAdaGrad
    l_planted = loss_of(f_h)  # the planted model's own held-out loss (0 for a regression target)
    f_seq = new_model()
    l_init = loss_of(f_seq.decisionFunction(Xh))
    o_seq = mk_(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="sequential", **hp)
    o_seq._handle(f_seq, ctx, "sequential")
    seq, t_seq = [], 0.0
    for e in range(1, max(cfg["seq_epochs"]) + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        o_seq._epoch(Xt, None, 0, n_t)
        o_seq.it += n_t
        ctx.synchronize()
        t_seq += time.perf_counter() - t0
        if e in cfg["seq_epochs"]:
            o_seq._finalize_into(f_seq)
            f_seq._dirty = False
            l_e = loss_of(f_seq.decisionFunction(Xh))
            seq.append({"epochs": e, "seconds": round(t_seq, 4), "held_out_loss": round(l_e, 6),
                        "gap_closed": round((l_init - l_e) / (l_init - l_planted), 4) if l_init > l_planted else None})
    del o_seq, f_seq

    def run_mb(batch_):
        f_mb = new_model()
        P_i, w_i = np.array(f_mb.P), np.array(f_mb.w)
        o_mb = mk_(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch", batch=batch_,
                   **({"touchCap": cap} if sgd else {"adaCross": float(wl.get("ada_cross", 0.0))}), **hp)
        o_mb._handle(f_mb, ctx, "minibatch")
        o_mb._epoch(Xt, None, 0, n_t)  # plan + graph built outside the clock, then start again from the same point
        f_mb.set_params(P_i, w_i, 0.0)
        f_mb._push(ctx)
        o_mb.it = 1
        capi.check(capi.lib().nfm_opt_set_it(o_mb._h, 1))  # (AdaGrad: the state starts over at it == 1)
        hits = [None] * len(seq)
        t_mb, e_mb, l_mb = 0.0, 0, l_init
        while e_mb < cfg["mb_epoch_cap"] and any(h is None for h in hits):
            ctx.synchronize()
            t0_ = time.perf_counter()
            o_mb._epoch(Xt, None, 0, n_t)
            o_mb.it += n_t
            ctx.synchronize()
            t_mb += time.perf_counter() - t0_
            e_mb += 1
            o_mb._finalize_into(f_mb)
            f_mb._dirty = False
            l_mb = loss_of(f_mb.decisionFunction(Xh))
            if not math.isfinite(l_mb):
                break
            for i_, s_ in enumerate(seq):
                if hits[i_] is None and l_mb <= s_["held_out_loss"]:
                    hits[i_] = {"seq_epochs": s_["epochs"], "target": s_["held_out_loss"], "reached": True, "epochs": e_mb,
                                "seconds": round(t_mb, 4), "speedup": round(s_["seconds"] / t_mb, 2)}
        for i_, s_ in enumerate(seq):
            if hits[i_] is None:
                hits[i_] = {"seq_epochs": s_["epochs"], "target": s_["held_out_loss"], "reached": False, "epochs": e_mb,
                            "seconds": round(t_mb, 4), "speedup": None}
        return {"batch": batch_, "touch_cap": cap if sgd else None, "epochs_run": e_mb, "held_out_loss": round(l_mb, 6) if math.isfinite(l_mb) else None,
                "seconds_per_epoch": round(t_mb / max(e_mb, 1), 5), "targets": hits}

    def score(r_):  # targets reached, then the speed-up on the hardest one reached
        got = [h for h in r_["targets"] if h["reached"]]
        return (len(got), got[-1]["speedup"] if got else 0.0)

    runs = [run_mb(batch)]
    if batches:  # an explicit sweep (--t2t-batches): every batch asked for, smaller or LARGER than the bench batch
        runs += [run_mb(b_) for b_ in batches if b_ != batch]
    else:
        for b_ in (8192, 2048):
            if score(runs[0])[0] == len(seq):
                break  # the bench batch reaches every target: nothing to look for
            if b_ < batch:
                runs.append(run_mb(b_))
    best = max(runs, key=score)
    return {"train_samples": n_t, "held_out_samples": n_h,
            "problem": "labels from a planted degree-2 FM (w ~ N(0, %g^2), P ~ N(0, %g^2), k = %d)%s; %s %s; start P ~ N(0, 0.01^2), w = 0"
                       % (cfg["planted_w"], cfg["planted_P"], k, ", sign" if classify else "", wl["solver"], json.dumps(hp)),
            "held_out_loss_at_start": round(l_init, 6), "planted_model_held_out_loss": round(l_planted, 6),
            "targets_are": "held-out mean loss after 1, 3 and 10 epochs in the reference's order (mode=sequential, the window kernel), same "
                           "start, step size and schedule; gap_closed = (start - loss) / (start - planted model's own loss)",
            "sequential": seq, "minibatch": runs, "best_batch": best["batch"],
            "note": "mini-batch epochs are capped at %d; seconds = training only (evaluation outside both clocks)" % cfg["mb_epoch_cap"]}


def run_training(args, name, torch, nf, dist, rank, world, dev, ctx, primary):
    """one workload on this rank's GPU -> (dict of the JSON fields that depend on the workload)"""
    import ctypes as C

    from nimfm_amd import _capi as capi

    wl = dict(WORKLOADS[name])
    if not primary and name in EXTRA_N:  # an extra leg of the default run: a bounded shard
        wl["n"] = EXTRA_N[name]
    if args.n and primary:
        wl["n"] = args.n
    n, d, m, k = wl["n"], wl["d"], wl["m"], wl["k"]
    n_orders = wl["degree"] - 1  # fitLower=explicit (model/factorization_machine.nim:81-97)
    batch = args.batch if (args.batch and primary) else wl["batch"]
    steps, warmup = (args.steps, args.warmup) if primary else (min(args.steps, 5 if name != "cfg2" else 10), min(args.warmup, 2))
    F = wl.get("fields", 0)
    X, indptr, indices, data, _keep_fields = make_dataset(torch, nf, ctx, dev, wl, n, rank)
    # labels from a planted FM (k, scale 0.1), like tests/utils.nim:29-47; classification -> sign
    rng = np.random.default_rng(1234)
    planted = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    planted.set_params(rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.1, 0.0)
    y = planted.decisionFunction(X)
    task = "classification" if wl["loss"] in ("logistic", "squared_hinge") else "regression"
    if task == "classification":
        y = np.sign(y)
    del planted
    if F:
        fm = nf.newFieldAwareFactorizationMachine(task, nComponents=k, warmStart=True, randomState=1)
    else:
        fm = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, warmStart=True, randomState=1)
    fm.init(X)  # w = 0, P ~ N(0, 0.01^2) (Box-Muller pairs in fill order), intercept = 0 (factorization_machine.nim:125-139)
    cap = float(args.touch_cap) if args.touch_cap > 0 else float(wl.get("touch_cap", 16.0))  # (per workload unless --touch-cap says otherwise)
    if wl["solver"] == "sgd":
        opt = nf.newSGD(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch", batch=batch, touchCap=cap)
    else:
        opt = nf.newAdaGrad(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch",
                            batch=batch, trackViol=not args.no_viol, adaCross=float(wl.get("ada_cross", 0.0)))
    X.set_targets(y)
    opt._handle(fm, ctx, "minibatch")
    sync_period = 0
    use_dp = world > 1 or getattr(run_training, "group", None) is not None
    if use_dp:
        # the exchange lives in the library (csrc/dp.hip): one RCCL communicator per rank, the replicas reconciled every
        # sync_period mini-batches on a second stream beside the next period's mini-batches, exactly at the end of the
        # epoch; torch.distributed only carried the group id (dp.Group.from_torch, in main)
        # an exchange about every 1e6 samples per rank (128 mini-batches of 8192, 4 of 262144) when the epoch has at least four
        # such stretches (half that for two); shorter epochs: the closing exchange only
        sp_ = max(1, 1_048_576 // batch)
        sync_period = args.sync_period if args.sync_period >= 0 else (sp_ if n // batch >= 4 * sp_ else (max(1, sp_ // 2) if n // batch >= 2 * sp_ else 0))
        opt.setDataParallel(run_training.group, sync_period, True, args.combine)

    def step(perm=None):
        ls, vs = opt._epoch(X, perm, 0, n)  # with a group: the sums over all ranks
        if use_dp:
            opt._sync_it()  # advanced by the samples of all ranks (the reference's threads share one counter)
        else:
            opt.it += n
        return ls, vs

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k_steps, perms=None):
        fence()
        t0 = time.perf_counter()
        last = None
        for e in range(k_steps):
            last = step(None if perms is None else perms[e])
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, last

    for _ in range(warmup):
        step()
    dp_info0 = run_training.group.info() if use_dp else None
    dt, last = timed(steps)
    ms_per_step = dt / steps * 1e3
    value = n * world / (dt / steps)
    dp_stats = None
    if use_dp:  # what the exchange moved during the timed steps (nfm_dp_info: this rank's collectives and bytes)
        i1 = run_training.group.info()
        comb = args.combine if args.combine != "auto" else ("mean" if wl["solver"] == "sgd" else ("sum" if sync_period == 1 else "state_cross"))
        dp_stats = {"combine": comb if wl["solver"] == "sgd" else {"sum": "state increments summed", "mean": "state increments summed",
                                                                    "state_mean": "state increments averaged",
                                                                    "state_cross": "g_sum increments summed, g_norm + the ranks' agreement (NFM_DP_STATE_CROSS)"}[comb],
                    "sync_period": sync_period, "world": i1["world"],
                    "collectives_per_step": (i1["collectives"] - dp_info0["collectives"]) / steps,
                    "bytes_per_step_per_rank": (i1["bytes"] - dp_info0["bytes"]) / steps,
                    "note": "mid-epoch all-reduces of the whole parameter (SGD) / state (AdaGrad) arena on the group's own stream, "
                            "delayed by one period, plus the exact closing exchange of every epoch call"}
        # what N ranks' epoch is worth against ONE rank's epoch over all samples (1.0: throughput IS speed-up), measured with
        # local groups on one GPU on a planted FM (tools/dp_convergence.py); quoted, not measured in this run
        prog = {"mean": {2: 1.15, 4: 1.20, 8: 1.12}, "state_mean": {2: 1.03, 4: 0.93, 8: 0.75}, "sum": {2: 0.96, 4: 0.98, 8: 1.01},
                "state_cross": {2: 1.00, 4: 0.74, 8: 1.01}}  # (ranks a whole epoch apart, 4 epochs; 16 mini-batches apart: 0.96 / 1.02 / 1.06)
        # the same runs' last-epoch TRAINING loss against one rank's (the optimisation progress itself; on that small problem one rank
        # over-fits from epoch 4 on, which the held-out figure above mixes in): below 1 = further along than one rank over all samples
        train = {"state_cross": {2: 0.90, 4: 0.94, 8: 1.04}, "state_mean": {2: 1.18, 4: 1.45, 8: 1.64}}
        if i1["world"] in (2, 4, 8) and comb in prog and (wl["solver"] == "adagrad" or comb == "mean"):
            dp_stats["progress_per_epoch"] = prog[comb][i1["world"]]
            dp_stats["progress_source"] = ("profiles/r05g_dp_convergence.txt" if comb == "state_cross" else "profiles/r03_dp_convergence.txt") + \
                                          " (planted FM, local groups of this size on one GPU; not this run)"
            if comb in train:
                dp_stats["train_loss_vs_one_rank"] = train[comb][i1["world"]]

    # ---- the reference's default shuffle = true (optimizer/sgd.nim:297): every epoch gets a FRESH permutation, so the
    # batch plan is rebuilt for every epoch inside the timed region.
    # value_shuffled: the order is drawn on the device (nfm_opt_set_shuffle) and the NEXT epoch's plan is built on a second
    # stream beside the current epoch.  value_shuffled_host_perm: the host hands over a permutation per epoch, as the
    # Nim host does after its own shuffle (upload + validation + plan build + un-graphed launches all counted; drawing
    # the permutation is the host's job and is not). ----
    value_shuffled = value_shuffled_host = float("nan")
    ks = kh = 0
    if args.no_shuffled:  # profiling runs: the shuffled epochs build plans beside the epoch and would mix into the per-kernel averages
        pass
    else:
        ks = max(2, min(5, steps))
        capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, 12345 + rank))
        step()  # warm-up: the first plan is built in line, the second one already beside this epoch
        dts, _ = timed(ks)
        value_shuffled = n * world / (dts / ks)
        capi.check(capi.lib().nfm_opt_set_shuffle(opt._h, -1))
        kh = max(1, min(3, steps))
        perms = [np.random.default_rng(7 + rank * 131 + e).permutation(n).astype(np.int64) for e in range(kh + 1)]
        step(perms[kh])  # warm-up of the un-cached path (allocator pools sized)
        _step_plain = step

        def step(perm=None, _seq=perms):  # noqa: F811 -- announces the next epoch's permutation, as the host loops do
            k_ = next(i for i, p_ in enumerate(_seq) if p_ is perm)
            if k_ + 1 < kh:
                capi.check(capi.lib().nfm_opt_announce_perm(opt._h, _seq[k_ + 1].ctypes.data, 0, n))
            return _step_plain(perm)

        dth, _ = timed(kh, perms)
        step = _step_plain
        value_shuffled_host = n * world / (dth / kh)
        del perms

    # ---- the headline at the batch of rounds 1-4 (8192), for continuity: a second optimizer on the same model, 3 epochs ----
    value_b8192 = None
    if primary and name == "headline" and batch != 8192 and world == 1 and not args.batch and not args.no_shuffled:
        opt8 = nf.newSGD(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch", batch=8192, touchCap=cap)
        opt8._handle(fm, ctx, "minibatch")
        opt8.it = opt.it
        for e8 in range(4):
            if e8 == 1:
                fence()
                t8 = time.perf_counter()
            opt8._epoch(X, None, 0, n)
            opt8.it += n
        fence()
        value_b8192 = round(n * 3 / (time.perf_counter() - t8), 1)
        del opt8

    # ---- AdaGrad without the stopping criterion's sum (trackViol = false): adagrad.nim:99 adds |P_old - P_new| per coordinate, which is
    # what makes a single-touch row SIX streams (g_sum, g_norm and the parameters as last stored, read and written) instead of four;
    # a caller that fits with tol <= 0 never reads it.  A secondary figure -- `value` always tracks it, as the reference does. ----
    value_no_viol = None
    if wl["solver"] == "adagrad" and rank == 0 and world == 1 and not args.no_viol and not args.no_shuffled:
        optv = nf.newAdaGrad(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="minibatch",
                             batch=batch, trackViol=False, adaCross=float(wl.get("ada_cross", 0.0)))
        optv._handle(fm, ctx, "minibatch")
        optv.it = opt.it
        for ev in range(4):
            if ev == 1:
                fence()
                tv = time.perf_counter()
            optv._epoch(X, None, 0, n)
            optv.it += n
        fence()
        value_no_viol = round(n * 3 / (time.perf_counter() - tv), 1)
        del optv

    # ---- predict samples/s (the metric's second half): decisionFunction over the shard, output on device ----
    pred = None
    if rank == 0:
        out_dev = torch.empty(n, dtype=torch.float64, device=dev)
        opt._finalize_into(fm)
        mh = fm._push(ctx)
        for _ in range(2):
            capi.check(capi.lib().nfm_decision_function_device(mh, X.h, out_dev.data_ptr()))
        ctx.synchronize()
        tp = time.perf_counter()
        reps_p = 10
        for _ in range(reps_p):
            capi.check(capi.lib().nfm_decision_function_device(mh, X.h, out_dev.data_ptr()))
        ctx.synchronize()
        tp = (time.perf_counter() - tp) / reps_p
        # score on the device (decisionFunction + reduction, only the scalar comes back)
        sc = C.c_double()
        capi.check(capi.lib().nfm_score(mh, X.h, C.byref(sc)))
        ctx.synchronize()
        ts = time.perf_counter()
        for _ in range(5):
            capi.check(capi.lib().nfm_score(mh, X.h, C.byref(sc)))
        ts = (time.perf_counter() - ts) / 5
        pbytes = 12 * m + 8 + (F * 8 * m * k if F else n_orders * 8 * m * k) + 8 * m + 8  # SURVEY.md 8(d) predict bytes per sample
        pred = {"value": round(n / tp, 1), "unit": "samples/s", "ms": round(tp * 1e3, 4),
                "roofline_frac": round(pbytes * n / tp / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_sample": pbytes,
                "score": {"value": round(n / ts, 1), "unit": "samples/s", "ms": round(ts * 1e3, 4), "result": sc.value}}
        del out_dev

    # ---- the reference's own sample-by-sample order (NFM_MODE_SEQUENTIAL: one sample in flight, results equal to the
    # reference-faithful CPU restatement) on a bounded prefix of the shard: the speed of the mode that reproduces the
    # reference exactly, reported beside the mini-batch rule's ----
    exact = None
    windowed_any = k <= 64 and wl["degree"] <= 6  # (the shapes whose exact order runs at speed: seqwin.hip)
    if rank == 0 and world == 1 and not args.no_exact:
        # seqwin.hip: the order as a dependency window over the chip (degree-2 FMs, several orders / degree <= 6, field-aware
        # models whose chain terms -- one per entry and per pair of entries -- fit a mailbox)
        windowed = k <= 64 and wl["degree"] <= 6 and (not F or m + m * (m - 1) // 2 <= 252)
        ns_ = min(n, (2_000_000 if not F and wl["degree"] == 2 else 500_000) if windowed else 20_000)
        Xs = nf.CSRDataset.from_device(ctx, ns_, d, ns_ * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                       fields_ptr=_keep_fields.data_ptr() if F else None, nFields=F,
                                       keep=(indptr, indices, data, _keep_fields))
        Xs.set_targets(np.ascontiguousarray(y[:ns_]))
        if F:
            fm_s = nf.newFieldAwareFactorizationMachine(task, nComponents=k, warmStart=True, randomState=1)
        else:
            fm_s = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, warmStart=True, randomState=1)
        fm_s.init(Xs)
        mk_ = nf.newSGD if wl["solver"] == "sgd" else nf.newAdaGrad
        opt_s = mk_(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="sequential")
        opt_s._handle(fm_s, ctx, "sequential")
        ctx.timing_reset()
        ctx.timing_enable(True)
        opt_s._epoch(Xs, None, 0, ns_)  # first epoch: builds the dependency table of this order (kept while the order stays)
        opt_s.it += ns_
        ctx.synchronize()
        deps_ms = ctx.timing_get("seq_window_deps")[1]
        ctx.timing_reset()
        t_s = time.perf_counter()
        opt_s._epoch(Xs, None, 0, ns_)
        opt_s.it += ns_
        ctx.synchronize()
        t_s = time.perf_counter() - t_s
        kern_ms = ctx.timing_get("sequential")[1]
        # the reference's default: a fresh order every epoch (sgd.nim:297) -- permutation from the host, table rebuilt
        perm_s = np.random.default_rng(5).permutation(ns_).astype(np.int64)
        t_p = time.perf_counter()
        opt_s._epoch(Xs, perm_s, 0, ns_)
        ctx.synchronize()
        t_p = time.perf_counter() - t_p
        ctx.timing_enable(False)
        exact = {"value": round(ns_ / t_s, 1), "unit": "samples/s", "us_per_step": round(t_s / ns_ * 1e6, 3),
                 "kernel_only": round(ns_ / (kern_ms * 1e-3), 1) if kern_ms > 0 else None,
                 "value_fresh_order": round(ns_ / t_p, 1),
                 "dependency_table_ms": round(deps_ms, 2),
                 "flavour": "one-term chain (the worker sums its sample's prediction but the intercept, the conductor's chain is "
                            "b + S -> dloss -> b'; NFM_SEQ_WIN_EXACT=1 is the term-by-term chain, bit-equal to the one-workgroup kernel)"
                            if windowed else None,
                 "sample": "the first %d samples of the shard, mode=sequential: %s; value = one epoch call (wall clock) over a fixed "
                           "order whose dependency table exists, value_fresh_order = one epoch call with a new permutation from "
                           "the host (upload + table build inside)" %
                           (ns_, "the reference's order as a dependency window over the chip (csrc/seqwin.hip), same sample order and "
                            "dependencies as the one-workgroup kernel; yhat rounded as b + (sum) (max_rel_diff: what that changes)"
                            if windowed else "one workgroup (csrc/seq.hip)")}
        if windowed:  # the bit-exact flavour of the window (the reference's term-by-term rounding of the prediction), for comparison
            old_ex = os.environ.get("NFM_SEQ_WIN_EXACT")
            os.environ["NFM_SEQ_WIN_EXACT"] = "1"
            try:
                opt_s._epoch(Xs, None, 0, ns_)  # (its mailboxes are sized on the first call)
                opt_s.it += ns_
                ctx.synchronize()
                t_e = time.perf_counter()
                opt_s._epoch(Xs, None, 0, ns_)
                opt_s.it += ns_
                ctx.synchronize()
                exact["value_term_by_term"] = round(ns_ / (time.perf_counter() - t_e), 1)
            finally:
                if old_ex is None:
                    os.environ.pop("NFM_SEQ_WIN_EXACT", None)
                else:
                    os.environ["NFM_SEQ_WIN_EXACT"] = old_ex
        if windowed:
            # fitIntercept = false: no scalar chain ties the samples (the intercept is what serialises the reference's order),
            # the window runs without its conductor on twice the workers -- the same order, bit-equal results
            if F:
                fm_n = nf.newFieldAwareFactorizationMachine(task, nComponents=k, fitIntercept=False, warmStart=True, randomState=1)
            else:
                fm_n = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, fitIntercept=False, warmStart=True, randomState=1)
            fm_n.init(Xs)
            opt_n = mk_(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="sequential")
            opt_n._handle(fm_n, ctx, "sequential")
            opt_n._epoch(Xs, None, 0, ns_)
            opt_n.it += ns_
            ctx.synchronize()
            t_n = time.perf_counter()
            opt_n._epoch(Xs, None, 0, ns_)
            opt_n.it += ns_
            ctx.synchronize()
            t_n = time.perf_counter() - t_n
            exact["no_intercept"] = {"value": round(ns_ / t_n, 1), "unit": "samples/s",
                                     "note": "the same order with fitIntercept=false: the window without a conductor (csrc/seqwin.hip)"}
            del opt_n, fm_n
        # the same order through the one-workgroup kernel (NFM_SEQ_WIN=0, csrc/seq.hip, which the parity tests hold to the
        # oracle) on the first 4096 samples: parameters, linear weights and intercept must agree BIT FOR BIT
        nb_ = min(ns_, 4096)
        Xb = nf.CSRDataset.from_device(ctx, nb_, d, nb_ * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                       fields_ptr=_keep_fields.data_ptr() if F else None, nFields=F,
                                       keep=(indptr, indices, data, _keep_fields))
        Xb.set_targets(np.ascontiguousarray(y[:nb_]))
        got_ = {}
        old_env, old_ex = os.environ.get("NFM_SEQ_WIN"), os.environ.get("NFM_SEQ_WIN_EXACT")
        for win_ in ("0", "2", "2x"):  # the one-workgroup kernel, the window (one-term chain), the window's term-by-term flavour
            os.environ["NFM_SEQ_WIN"] = win_[0]
            os.environ["NFM_SEQ_WIN_EXACT"] = "1" if win_ == "2x" else "0"
            if F:
                fb_ = nf.newFieldAwareFactorizationMachine(task, nComponents=k, warmStart=True, randomState=1)
            else:
                fb_ = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, warmStart=True, randomState=1)
            fb_.init(Xb)
            ob_ = mk_(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="sequential")
            ob_._handle(fb_, ctx, "sequential")
            ob_._epoch(Xb, None, 0, nb_)
            ob_.it += nb_
            ob_._finalize_into(fb_)
            got_[win_] = (np.array(fb_.P).copy(), np.array(fb_.w).copy(), float(fb_.intercept))
            del ob_, fb_
        for key_, old_ in (("NFM_SEQ_WIN", old_env), ("NFM_SEQ_WIN_EXACT", old_ex)):
            if old_ is None:
                os.environ.pop(key_, None)
            else:
                os.environ[key_] = old_
        exact["window_fallbacks"] = ctx.timing_get("seq_window_fallback")[0]  # window launches that aborted (must be 0)
        exact["bit_equal"] = bool(windowed and exact["window_fallbacks"] == 0 and np.array_equal(got_["0"][0].view(np.uint64), got_["2x"][0].view(np.uint64))
                                  and np.array_equal(got_["0"][1].view(np.uint64), got_["2x"][1].view(np.uint64))
                                  and got_["0"][2] == got_["2x"][2] and np.isfinite(got_["2x"][0]).all())

        def rel_(a_, b_):  # max |a - b| / (|b| + 1e-12 max|b|) over the tensor
            b_ = np.asarray(b_, dtype=np.float64)
            return float(np.max(np.abs(np.asarray(a_) - b_) / (np.abs(b_) + 1e-12 * max(float(np.max(np.abs(b_))), 1e-300)))) if b_.size else 0.0

        exact["max_rel_diff"] = max(rel_(got_["2"][0], got_["0"][0]), rel_(got_["2"][1], got_["0"][1]),
                                    abs(got_["2"][2] - got_["0"][2]) / max(abs(got_["0"][2]), 1e-300)) if np.isfinite(got_["2"][0]).all() else None
        exact["bit_equal_sample"] = ("first %d samples of the shard, P / w / intercept after one epoch: bit_equal = the window's term-by-term "
                                     "flavour (NFM_SEQ_WIN_EXACT=1) against the one-workgroup kernel, as bits; max_rel_diff = the one-term "
                                     "flavour (the one `value` is measured with) against the one-workgroup kernel (north_star: 1e-6)" % nb_)
        del opt_s, fm_s, Xs, Xb, got_

    # ---- what the mini-batch rule costs statistically: c_bar (a formula) and time_to_target (the measurement) ----
    lam = batch * m / d
    cap_eff = cap if wl["solver"] == "sgd" else float("inf")  # AdaGrad's state sums every step
    c_bar = c_bar_of(lam, cap_eff)
    t2t = None
    t2t_window = k <= 64 and wl["degree"] <= 6 and (not F or m + m * (m - 1) // 2 <= 252)  # the exact order at speed (seqwin.hip)
    if rank == 0 and world == 1 and t2t_window and n >= 400_000 and not args.no_t2t:
        t2t = time_to_target_leg(torch, nf, ctx, dev, wl, name, n, batch, cap, indices, data, _keep_fields, task,
                                 batches=[int(v_) for v_ in args.t2t_batches.split(",")] if args.t2t_batches else None)

    # ---- roofline leg: per-kernel durations from HIP events on the library's stream (one replica, no exchange) ----
    roof = None
    if use_dp:
        if world > 1:
            dist.barrier()
        opt.setDataParallel(None)
    if rank == 0:
        # five epochs, each timed on its own; the figures are those of the MEDIAN epoch (by the time of a mini-batch's launches), so that one
        # stall of the box -- seen once: a row phase average of 1.70 ms in two epochs whose timed region ran it at 1.41 -- does not
        # stand for the kernel; every epoch's figure is kept beside it ("pair_ms_epochs")
        fams_ = ("row_phase", "singles", "col_phase", "heavy_partial", "heavy_apply", "refresh", "schedule")
        ctx.timing_enable(True)
        reps, epochs_ = 5, []
        for _ in range(reps):
            ctx.timing_reset()
            opt._epoch(X, None, 0, n)
            opt.it += n
            ctx.synchronize()
            fam_ = {f: ctx.timing_get(f) for f in fams_}
            ms_ = {f: (fam_[f][1] / fam_[f][0] if fam_[f][0] else 0.0) for f in fam_}
            pair_ = (ms_["row_phase"] + ms_["singles"] + ms_["col_phase"]
                     + (fam_["heavy_partial"][1] + fam_["heavy_apply"][1] + fam_["refresh"][1]) / max(fam_["row_phase"][0], 1))
            epochs_.append((pair_, fam_, ms_))
        ctx.timing_enable(False)
        pair_ms, fam, per_batch_ms = sorted(epochs_, key=lambda e_: e_[0])[reps // 2]
        n_batches = fam["row_phase"][0]
        reps = 1  # (fam holds ONE epoch's counts)
        bps = algorithmic_bytes_per_sample(wl["solver"], m, k, n_orders, F)
        units = n / n_batches
        achieved = bps * units / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
        # HBM-side bytes per mini-batch are NOT measured in this run (PMC counters need their own rocprofv3 --pmc
        # passes): the figure of the newest committed PMC run of this workload / batch is quoted and labelled, else null
        traffic, traffic_src = None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
            t = json.load(open(path))
            if t.get("workload") == name and t.get("batch") == batch:
                traffic, traffic_src = t["hbm_bytes_per_minibatch"], "profiles/" + os.path.basename(path)
        roof = {"bound": "hbm", "kernel": "k_row_phase (+ k_singles in the sparse regime) + k_col_phase: one mini-batch = one launch of each",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": (traffic_src + " (a separate rocprofv3 --pmc run of this workload, not this run)") if traffic_src else None,
                "algorithmic_bytes_per_minibatch": bps * units,
                "bytes_per_sample": bps, "samples_per_launch": units,
                "avg_ms": {f: round(per_batch_ms[f], 5) for f in per_batch_ms},
                "avg_over": "the median of 5 epochs, each timed on its own", "pair_ms_epochs": [round(e_[0], 5) for e_ in epochs_],
                "launches_per_step": {f: fam[f][0] / reps for f in fam}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_leg(args, wl, n, d, m, k, n_orders, indptr, indices, data, y, cheap=not primary,
                               fields=X._keep[3] if F else None)

    chk = cpu.pop("_check", None) if cpu is not None else None
    if exact is not None and chk and windowed_any:
        # PARITY inside the bench: the same prefix, start and hyper-parameters through the GPU's exact order (mode=sequential,
        # the default window flavour) and through the one-thread CPU port (the oracle's restatement of optimizer/sgd.nim:261-328 /
        # adagrad.nim:137-203 -- run above as the reported baseline, here only its result is read): parameters after the same epochs
        nc_, ep_ = chk["nc"], chk["epochs"]
        Xc = nf.CSRDataset.from_device(ctx, nc_, d, nc_ * m, indptr.data_ptr(), indices.data_ptr(), data.data_ptr(),
                                       fields_ptr=_keep_fields.data_ptr() if F else None, nFields=F,
                                       keep=(indptr, indices, data, _keep_fields))
        Xc.set_targets(np.ascontiguousarray(y[:nc_]))
        if F:
            fc_ = nf.newFieldAwareFactorizationMachine(task, nComponents=k, warmStart=True, randomState=1)
        else:
            fc_ = nf.newFactorizationMachine(task, degree=wl["degree"], nComponents=k, warmStart=True, randomState=1)
        fc_.set_params(chk["P0"], np.zeros(d), 0.0)
        oc_ = (nf.newSGD if wl["solver"] == "sgd" else nf.newAdaGrad)(maxIter=1, loss=wl["loss"], verbose=0, tol=0, shuffle=False, mode="sequential")
        oc_._handle(fc_, ctx, "sequential")
        for _ in range(ep_):
            oc_._epoch(Xc, None, 0, nc_)
            oc_.it += nc_
        oc_._finalize_into(fc_)

        def rel2_(a_, b_):
            b_ = np.asarray(b_, dtype=np.float64)
            return float(np.max(np.abs(np.asarray(a_) - b_) / (np.abs(b_) + 1e-12 * max(float(np.max(np.abs(b_))), 1e-300))))

        exact["vs_cpu_port_max_rel_diff"] = max(rel2_(np.array(fc_.P), chk["P"]), rel2_(np.array(fc_.w), chk["w"]),
                                                abs(float(fc_.intercept) - chk["b"]) / max(abs(chk["b"]), 1e-300))
        exact["vs_cpu_port_sample"] = ("parameters after %d epochs over the first %d samples: GPU exact order (default window flavour) against "
                                       "the one-thread CPU port (built -O3 -march=native), same start and hyper-parameters" % (ep_, nc_))
        del oc_, fc_, Xc
    if exact is not None and cpu is not None and cpu.get("value"):
        # both run the reference's sample order and give the same parameters: the like-for-like ratio
        exact["vs_cpu_port_1_thread"] = round(exact["value"] / cpu["value"], 2)
        if exact.get("no_intercept"):  # (the CPU port's step costs the same with or without the intercept's three flops)
            exact["no_intercept"]["vs_cpu_port_1_thread"] = round(exact["no_intercept"]["value"] / cpu["value"], 2)
    return {"value": round(value, 1), "ms_per_step": round(ms_per_step, 4), "steps": steps, "warmup": warmup,
            "value_batch_8192": value_b8192, "value_no_viol": value_no_viol,
            "value_shuffled": None if math.isnan(value_shuffled) else round(value_shuffled, 1),
            "value_shuffled_host_perm": None if math.isnan(value_shuffled_host) else round(value_shuffled_host, 1),
            "shuffled_note": "%d epochs, each over a fresh random order drawn on the device, the next epoch's batch plan built on "
                             "a second stream beside the current epoch; host_perm: %d epochs, each with a permutation handed "
                             "over by the host (upload, validation, plan build, un-graphed launches all in the timed region)"
                             % (ks, kh),
            "config": {"workload": "%s: synthetic CSR %dx%d, %d nnz/row, k=%d, %s %s loss, mini-batch %d, "
                                   "mode=minibatch, fixed order (value_shuffled: a fresh order per epoch)" % (name, n, d, m, k, wl["solver"].upper(), wl["loss"], batch),
                       "update_rule": "this library's deterministic mini-batch rule (all samples of a batch see the batch-start "
                                      "parameters; per coordinate the batch's per-sample steps are %s, DESIGN.md section 4) -- NOT the "
                                      "reference's sample-by-sample order, which NFM_MODE_SEQUENTIAL reproduces (exact_order: its "
                                      "samples/s on this shape; time_to_target: what the rule costs statistically against it)"
                                      % (("summed up to touch cap %g, scaled by cap / c beyond it" % cap) if wl["solver"] == "sgd"
                                         else "summed into AdaGrad's additive state"),
                       "update_rule_short": ("library's mini-batch rule (DESIGN.md 4): %s; not the reference's order (see exact_order)"
                                             % (("steps summed up to touch cap %g" % cap) if wl["solver"] == "sgd" else "AdaGrad state summed per batch")),
                       "touch_cap": cap if wl["solver"] == "sgd" else None,
                       "samples_per_gpu": n, "batch": batch,
                       "c_bar": round(c_bar, 4),
                       "c_bar_note": "mean over the touched coordinates of max(1, c / touch_cap), c = touches of the coordinate in a "
                                     "mini-batch ~ Poisson(batch * nnz_per_row / d): one epoch of this rule makes about 1 / c_bar of the "
                                     "reference order's steps per coordinate (DESIGN.md section 4); `effective` = value / c_bar, "
                                     "`time_to_target` measures the real thing",
                       "parallelism": ("%d ranks, one process per GPU, contiguous sample shards; replicas %s in the library over "
                                       "RCCL every %s on a second stream + exactly at the end of every epoch"
                                       % (world, ("increments %s" % ("summed" if args.combine == "sum" else "averaged")) if wl["solver"] == "sgd"
                                          else ("state increments %s" % ("averaged" if args.combine == "state_mean" and sync_period != 1 else
                                                                         "summed, the squared norm taking the ranks' agreement" if args.combine in ("auto", "state_cross") and sync_period != 1
                                                                         else "summed")),
                                          ("%d mini-batches" % sync_period) if sync_period else "epoch (no mid-epoch exchange)"))
                       if use_dp else "1 GPU"},
            "last_step": {"mean_loss": last[0] / (n * world), "viol": last[1]}, "predict": pred, "exact_order": exact,
            "c_bar": round(c_bar, 4), "effective": round(value / c_bar, 1), "time_to_target": t2t, "dp": dp_stats,
            "roofline": roof, "cpu_baseline": cpu}


def _short(text, limit):
    text = "" if text is None else str(text)
    return text if len(text) <= limit else text[: limit - 3] + "..."


def _t2t_compact(t):
    """time_to_target -> {batch, targets: [{seq_epochs, target, epochs, seconds, speedup}] x 3} (best batch)"""
    if not t:
        return None
    best = next((r for r in t["minibatch"] if r["batch"] == t["best_batch"]), t["minibatch"][0])
    return {"batch": best["batch"], "gap_closed": [s_["gap_closed"] for s_ in t["sequential"]],
            "seq_seconds": [s_["seconds"] for s_ in t["sequential"]],
            "targets": [{"seq_epochs": h["seq_epochs"], "target": h["target"], "epochs": h["epochs"] if h["reached"] else None,
                         "seconds": h["seconds"] if h["reached"] else None, "speedup": h["speedup"]} for h in best["targets"]]}


def _roof_compact(r):
    if not r:
        return None
    return {"bound": r["bound"], "kernel": _short(r.get("kernel"), 100), "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
            "frac": r["frac"], "traffic": r.get("traffic"), "traffic_source": _short(r.get("traffic_source"), 60) if r.get("traffic_source") else None,
            "algorithmic_bytes_per_launch_pair": r.get("algorithmic_bytes_per_minibatch"), "avg_over": r.get("avg_over"),
            "avg_ms": {k_: v_ for k_, v_ in (r.get("avg_ms") or {}).items() if v_}}


def _cpu_compact(c):
    if not c:
        return None
    out = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": _short(c.get("sample"), 150)}
    if c.get("hogwild"):
        out["hogwild"] = {"value": c["hogwild"]["value"], "threads": c["hogwild"]["threads"]}
    if c.get("jagged"):
        out["jagged"] = {"value": c["jagged"]["value"], "cores": c["jagged"]["cores"]}
    if c.get("host"):
        out["host"] = c["host"]
    return out


def _extra_compact(e):
    """one BASELINE config of the default run as <= a dozen scalars (everything else: gpurun_out/bench_detail.json)"""
    roof, t = e.get("roofline") or {}, _t2t_compact(e.get("time_to_target"))
    traffic = roof.get("traffic")
    alg = roof.get("algorithmic_bytes_per_minibatch")
    return {"workload": _short(e["config"]["workload"], 170), "value": e["value"], "ms_per_step": e["ms_per_step"], "steps": e["steps"],
            "batch": e["config"]["batch"], "frac": roof.get("frac"),
            "traffic_ratio": round(traffic / alg, 3) if traffic and alg else None,
            "value_shuffled": e.get("value_shuffled"), **({"value_no_viol": e["value_no_viol"]} if e.get("value_no_viol") else {}),
            "predict": (e.get("predict") or {}).get("value"), "predict_frac": (e.get("predict") or {}).get("roofline_frac"),
            "exact_order": (e.get("exact_order") or {}).get("value"), "exact_bit_equal": (e.get("exact_order") or {}).get("bit_equal"),
            "exact_max_rel_diff": (e.get("exact_order") or {}).get("max_rel_diff"),
            "exact_vs_cpu_port_max_rel_diff": (e.get("exact_order") or {}).get("vs_cpu_port_max_rel_diff"),
            "exact_vs_cpu_1_thread": (e.get("exact_order") or {}).get("vs_cpu_port_1_thread"),
            "exact_no_intercept": ((e.get("exact_order") or {}).get("no_intercept") or {}).get("value"),
            "t2t_batch": t["batch"] if t else None, "t2t_speedup": [h["speedup"] for h in t["targets"]] if t else None,
            "reached": all(h["speedup"] is not None for h in t["targets"]) if t else None,
            "cpu_baseline": (e.get("cpu_baseline") or {}).get("value")}


def contract_line(full):
    """The ONE stdout line of the contract, kept under 6 KB (the driver holds about 8 KB of stdout: round 3's 24 KB line
    was cut in the middle and left the round without a parsed record).  Everything else -- notes, sweeps, per-kernel
    averages, every mini-batch run of time_to_target -- goes to gpurun_out/bench_detail.json and to stderr."""
    cfg = full["config"]
    out = {k_: full[k_] for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": _short(cfg["workload"], 180), "batch": cfg.get("batch"), "touch_cap": cfg.get("touch_cap"),
                     "update_rule": _short(cfg.get("update_rule_short") or cfg.get("update_rule"), 120), "parallelism": _short(cfg.get("parallelism"), 120)}
    out["value_shuffled"] = full.get("value_shuffled")
    out["value_batch_8192"] = full.get("value_batch_8192")
    p_ = full.get("predict")
    out["predict"] = {"value": p_["value"], "unit": p_["unit"], "roofline_frac": p_["roofline_frac"]} if p_ else None
    x_ = full.get("exact_order")
    out["exact_order"] = {"value": x_["value"], "unit": x_["unit"], "vs_cpu_port_1_thread": x_.get("vs_cpu_port_1_thread"),
                          "max_rel_diff": x_.get("max_rel_diff"), "vs_cpu_port_max_rel_diff": x_.get("vs_cpu_port_max_rel_diff"),
                          "term_by_term": x_.get("value_term_by_term"),
                          "bit_equal": x_.get("bit_equal"), "no_intercept": (x_.get("no_intercept") or {}).get("value"),
                          "no_intercept_vs_cpu_port_1_thread": (x_.get("no_intercept") or {}).get("vs_cpu_port_1_thread")} if x_ else None
    out["time_to_target"] = _t2t_compact(full.get("time_to_target"))
    if out["time_to_target"] and x_ and x_.get("vs_cpu_port_1_thread"):
        # the speed-ups are against the GPU's OWN exact order; against the one-thread CPU port in that order (which reaches the same
        # losses: it computes the same parameters) they are that many times exact_order.vs_cpu_port_1_thread
        out["time_to_target"]["speedup_is_vs"] = "gpu exact order"
        out["time_to_target"]["vs_cpu_port_1_thread"] = [round(t_["speedup"] * x_["vs_cpu_port_1_thread"], 0) if t_.get("speedup") else None
                                                         for t_ in out["time_to_target"]["targets"]]
    d_ = full.get("dp")
    out["dp"] = {k_: d_[k_] for k_ in ("combine", "sync_period", "world", "collectives_per_step", "bytes_per_step_per_rank",
                                         "progress_per_epoch", "train_loss_vs_one_rank") if k_ in d_} if d_ else None
    out["roofline"] = _roof_compact(full.get("roofline"))
    out["cpu_baseline"] = _cpu_compact(full.get("cpu_baseline"))
    c_ = full.get("cpu_baseline") or {}
    best_cpu = max([v_ for v_ in (c_.get("value"), (c_.get("hogwild") or {}).get("value")) if v_] or [0])
    out["vs_best_cpu"] = round(full["value"] / best_cpu, 1) if best_cpu else None  # (the faster of 1 thread and the best Hogwild setting)
    ex = full.get("extra")
    out["extra"] = {k_: _extra_compact(v_) for k_, v_ in ex.items()} if ex else None
    out["detail"] = "gpurun_out/bench_detail.json"
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= 6000:  # never again: drop the optional parts, most verbose first
        for victim in ("extra", "time_to_target", "dp", "exact_order", "predict"):
            out[victim] = None if victim != "extra" else {k_: {"value": v_["value"], "frac": v_["frac"]} for k_, v_ in (out["extra"] or {}).items()}
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < 6000:
                break
    return line


def emit(full):
    """detail -> gpurun_out/bench_detail.json + stderr; the contract line -> stdout, last"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        print("[bench.py] could not write gpurun_out/bench_detail.json: %s" % e, file=sys.stderr)
    print("[bench.py] detail: " + json.dumps(full), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(contract_line(full), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS) + ["ingest", "psgd"])
    ap.add_argument("--batch", type=int, default=0, help="mini-batch size (default: per workload)")
    ap.add_argument("--t2t-batches", default="", help="time_to_target: also run these mini-batch sizes (comma separated), smaller or larger")
    ap.add_argument("--n", "--samples", dest="n", type=int, default=0, help="override samples per GPU (--samples under torchrun, whose parser claims --n)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.cfg2 ... extra.cfg5 legs of the default (headline) run")
    ap.add_argument("--no-shuffled", action="store_true", help="skip the shuffled-epoch legs (profiling runs: their plan builds beside the epochs would mix into the per-kernel averages)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-order (mode=sequential) leg (batch sweeps)")
    ap.add_argument("--no-t2t", action="store_true", help="skip the time_to_target leg (profiling runs: its other batch sizes would mix into the per-kernel averages)")
    ap.add_argument("--no-viol", action="store_true",
                    help="AdaGrad without the reference's viol = sum|P_old - P_new| (no stored copy of P is read or "
                         "written; the stopping criterion is then unavailable) -- an information run, not the metric")
    ap.add_argument("--cpu-samples", type=int, default=1_000_000)
    ap.add_argument("--sync-period", type=int, default=-1,
                    help="N > 1: mini-batches between exchanges (0 = only at the end of every epoch; default: about every 1e6 samples "
                         "per rank -- 128 mini-batches of 8192, 4 of 262144 -- for epochs of at least four such stretches, else 0)")
    ap.add_argument("--combine", default="auto", choices=["auto", "mean", "sum", "state_mean", "state_cross"],
                    help="N > 1: how the ranks' increments are combined at an exchange (DESIGN.md section 6); auto = SGD: the mean, "
                         "AdaGrad: the state increments averaged (summed when the ranks exchange after every mini-batch)")
    ap.add_argument("--touch-cap", type=float, default=0.0,
                    help="SGD mini-batch rule: steps of a batch on one coordinate that are summed before averaging sets in "
                         "(nfm_opt_set_touch_cap; 1 = the per-coordinate mean, the library's default).  Default: per workload -- "
                         "about twice the touches a coordinate gets per batch: 16 at lambda = B m / d ~ 10 (cfg5), 32 at ~ 17-21 "
                         "(headline at 262144, cfg2 at 65536); time_to_target measures what it buys")
    ap.add_argument("--psgd-shape", default="", help="--workload psgd: d,m,k instead of the ml-100k shape")
    args = ap.parse_args()

    if args.workload == "ingest":
        return ingest_leg(args)
    if args.workload == "psgd":
        return psgd_leg(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)  # before anything that could touch a GPU
    if "WORLD_SIZE" in os.environ:  # a rank says so before the slow imports (tests/test_bench_spawn.py counts these lines)
        print("[bench.py] rank %s of %s started" % (os.environ.get("RANK", "0"), os.environ["WORLD_SIZE"]), file=sys.stderr, flush=True)
    import torch

    import nimfm_amd as nf

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libnimfm_hip has no CPU fallback")
    # one process per GPU; NIMFM_BENCH_BACKEND=gloo rehearses the multi-process path on fewer GPUs than
    # ranks (ranks then share devices; RCCL needs one GPU per rank)
    backend = os.environ.get("NIMFM_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU over RCCL); "
                         "NIMFM_BENCH_BACKEND=gloo rehearses the multi-process path on fewer GPUs" % (world, torch.cuda.device_count()))
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    ctx = nf.Context(dev_index)
    nf.set_default_context(ctx)
    if world > 1 or os.environ.get("NIMFM_BENCH_FORCE_DP") == "1":  # FORCE_DP: the N > 1 plumbing with one rank (rehearsal)
        import torch.distributed as dist

        from nimfm_amd import dp

        # torch.distributed (gloo, host side) is only the bootstrap and the clock: it carries the group id and the
        # barriers / MAX-over-ranks of the timing.  The exchange itself is the library's own RCCL communicator.
        # gloo and RCCL print banners ("[Gloo] Rank 0 is connected ...", "RCCL version : ...") on STDOUT when they come
        # up; stdout carries exactly one JSON line, so file descriptor 1 points at stderr while the group is created
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            if backend == "nccl":
                run_training.group = dp.Group.from_torch(ctx, dist)
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if backend != "nccl":
            raise SystemExit("NIMFM_BENCH_BACKEND=%s: the exchange is RCCL inside the library; rehearse the rule on the CPU "
                             "with tests/test_dp_gloo.py or on one GPU with tests/test_gpu_dp.py" % backend)

    res = run_training(args, args.workload, torch, nf, dist, rank, world, dev, ctx, primary=True)
    extra = None
    forced_dp = os.environ.get("NIMFM_BENCH_FORCE_DP") == "1"
    if args.workload == "headline" and world == 1 and not forced_dp and not args.no_extra and not args.n:
        # the other BASELINE.json configs on the same line: configs[1] (cfg2), [2] on one GPU (cfg3, a 4e6-sample shard),
        # [3] (cfg4, field-aware) and [4] (cfg5, degree 3) -- each with its own roofline, shuffled rate, exact-order rate,
        # time_to_target and a bounded cpu_baseline
        import gc

        extra = {}
        for ename in ("cfg2", "cfg3", "cfg4", "cfg5"):
            gc.collect()
            torch.cuda.empty_cache()
            ce = run_training(args, ename, torch, nf, dist, rank, world, dev, ctx, primary=False)
            extra[ename] = {"metric": "%s training samples/sec/epoch" % ("SGD" if WORKLOADS[ename]["solver"] == "sgd" else "AdaGrad"),
                            "unit": "samples/s", **ce}
    if args.workload == "headline" and (world > 1 or forced_dp) and not args.no_extra:
        # BASELINE.json configs[2] as written: AdaGrad, mini-batch 8192, data-parallel -- the replicas' state increments summed
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        c3 = run_training(args, "cfg3", torch, nf, dist, rank, world, dev, ctx, primary=False)
        extra = {"cfg3": {"metric": "AdaGrad training samples/sec/epoch", "unit": "samples/s", **c3}}
    if rank == 0:
        out = {"metric": "SGD training samples/sec/epoch", "value": res["value"], "unit": "samples/s",
               "n_gpus": world, "steps": res["steps"], "warmup": res["warmup"], "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": res["config"], "value_batch_8192": res.get("value_batch_8192"), "value_shuffled": res["value_shuffled"],
               "value_shuffled_host_perm": res["value_shuffled_host_perm"], "shuffled_note": res["shuffled_note"],
               "last_step": res["last_step"], "predict": res["predict"], "exact_order": res["exact_order"],
               "effective": res["effective"], "time_to_target": res["time_to_target"], "dp": res["dp"],
               "roofline": res["roofline"],
               "cpu_baseline": res["cpu_baseline"], "extra": extra}
        emit(out)
    if world > 1:
        dist.barrier()  # rank 0 ran the predict / roofline legs after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
