"""world_size-2 gloo test of the data-parallel exchange (nimfm_amd/dp.py) on CPU tensors.

Each rank trains its shard with the CPU oracle standing in for the GPU engine (the oracle is only
the test's engine here, never the product's), the replicas are reconciled with dp.exchange over
gloo, and the result is checked against a single-process restatement of the same rule."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle as O
    from common import random_csr
    from nimfm_amd import dp

    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, m, k, B = 400, 50, 6, 4, 32
    full = random_csr(n, d, m, seed=11)
    rng = np.random.default_rng(3)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.1, np.zeros(d)
    lo, hi = rank * n // world, (rank + 1) * n // world  # contiguous shards, sgd_multi.nim:85-88
    shard = O.Dataset(full.indptr[lo:hi + 1] - full.indptr[lo], full.indices[full.indptr[lo]:full.indptr[hi]],
                      full.data[full.indptr[lo]:full.indptr[hi]], hi - lo, d)
    # ---- SGD: replicas averaged after every epoch ----
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    for _ in range(3):
        b, it, _, _ = O.fm_sgd_epoch_mb(shard, y[lo:hi], 2, P, w, b, O.sgd_cfg(), B, it=it)
        tb = torch.tensor([b], dtype=torch.float64)
        tensors = [torch.from_numpy(P.reshape(-1)), torch.from_numpy(w), tb]
        dp.exchange(tensors, dist, world, "average")
        b = float(tb[0])
    # ---- AdaGrad: state increments summed ----
    cfg = O.adagrad_cfg()
    Pa, wa, ba, ita = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(1, d, k, d)
    views = [torch.from_numpy(st.gsum_P.reshape(-1)), torch.from_numpy(st.gnorm_P.reshape(-1)),
             torch.from_numpy(st.gsum_w), torch.from_numpy(st.gnorm_w)]
    sb = torch.zeros(2, dtype=torch.float64)
    ba, ita, _, _ = O.fm_adagrad_epoch_mb(shard, y[lo:hi], 2, Pa, wa, ba, cfg, B, st, it=ita)
    sb[0], sb[1] = st.gsum_b.value, st.gnorm_b.value
    prevs = [torch.zeros_like(v) for v in views] + [torch.zeros(2, dtype=torch.float64)]
    prevs[1].fill_(cfg.eps); prevs[3].fill_(cfg.eps); prevs[4][1] = cfg.eps  # state before the epoch
    dp.exchange(views + [sb], dist, world, "sum_deltas", prevs)
    q.put((rank, P.copy(), w.copy(), b, st.gsum_P.copy(), st.gnorm_P.copy(), st.gsum_w.copy(), sb.numpy().copy()))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_exchange_world2():
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from common import random_csr

    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=150) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # both ranks hold the same replicas after the exchange
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # single-process restatement
    n, d, m, k, B, world = 400, 50, 6, 4, 32, 2
    full = random_csr(n, d, m, seed=11)
    rng = np.random.default_rng(3)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.1, np.zeros(d)
    shards = []
    for r in range(world):
        lo, hi = r * n // world, (r + 1) * n // world
        shards.append((O.Dataset(full.indptr[lo:hi + 1] - full.indptr[lo], full.indices[full.indptr[lo]:full.indptr[hi]],
                                 full.data[full.indptr[lo]:full.indptr[hi]], hi - lo, d), y[lo:hi]))
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    for _ in range(3):
        reps = []
        for sh, ys in shards:
            Pr, wr = P.copy(), w.copy()
            br, itr, _, _ = O.fm_sgd_epoch_mb(sh, ys, 2, Pr, wr, b, O.sgd_cfg(), B, it=it)
            reps.append((Pr, wr, br))
        it = itr
        P = (reps[0][0] + reps[1][0]) / world
        w = (reps[0][1] + reps[1][1]) / world
        b = (reps[0][2] + reps[1][2]) / world
    assert np.allclose(res[0][1], P, rtol=1e-13, atol=1e-15) and np.allclose(res[0][2], w, rtol=1e-13, atol=1e-15)
    assert abs(res[0][3] - b) < 1e-14
    cfg = O.adagrad_cfg()
    gs, gn, gw = np.zeros((1, d, k)), np.full((1, d, k), cfg.eps), np.zeros(d)
    gb = np.array([0.0, cfg.eps])
    for sh, ys in shards:
        st = O.AdaState(1, d, k, d)
        O.fm_adagrad_epoch_mb(sh, ys, 2, P0.copy(), w0.copy(), 0.0, cfg, B, st, it=1)
        gs += st.gsum_P
        gn += st.gnorm_P - cfg.eps
        gw += st.gsum_w
        gb += [st.gsum_b.value, st.gnorm_b.value - cfg.eps]
    assert np.allclose(res[0][4], gs, rtol=1e-12, atol=1e-15) and np.allclose(res[0][5], gn, rtol=1e-12, atol=1e-18)
    assert np.allclose(res[0][6], gw, rtol=1e-12, atol=1e-15) and np.allclose(res[0][7], gb, rtol=1e-12, atol=1e-15)
