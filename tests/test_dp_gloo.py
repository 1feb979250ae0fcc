"""world_size-2 gloo test of the data-parallel exchange RULE (csrc/dp.hip, restated in tests/dp_rule.py) on CPU.

Each process is one rank: it trains its contiguous shard (the reference's thread partition, optimizer/sgd_multi.nim:
85-88) with the CPU oracle of the mini-batch rule standing in for the GPU engine (the oracle is only the test's engine
here, never the product's) and reconciles with gloo all-reduces at the points the library would.  The result is held
to the single-process lockstep simulation of the same rule: sync_period 0 (closing exchange only) and > 1, with the
exchange delayed by one period (overlap) and immediate, shards of unequal size, SGD and AdaGrad."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(0, True), (2, True), (3, False)]  # (sync_period, overlap)
N, D, M, K, B = 403, 50, 6, 4, 32  # 403 samples over 2 ranks: 201 + 202 (13 batches each, tails of 9 and 10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from common import random_csr

    full = random_csr(N, D, M, seed=11)
    rng = np.random.default_rng(3)
    y = rng.standard_normal(N)
    P0, w0 = rng.standard_normal((1, K, D)) * 0.1, rng.standard_normal(D) * 0.01
    return O, full, y, P0, w0


def _shard(O, full, y, rank, world):
    from nimfm_amd.dp import shard_bounds

    lo, hi = shard_bounds(N, rank, world)
    a, b = full.indptr[lo], full.indptr[hi]
    return O.Dataset(full.indptr[lo:hi + 1] - a, full.indices[a:b], full.data[a:b], hi - lo, D), y[lo:hi]


def _rank_gens(O, shard, ys, P0, w0, S, overlap, world):
    """the SGD and the AdaGrad generator of one rank (two epochs each: the second starts from reconciled replicas)"""
    import dp_rule as R

    def sgd():
        P, w, b, it = P0.copy(), w0.copy(), 0.25, 1
        cfg = O.sgd_cfg(eta0=0.05)
        out = []
        for _ in range(2):
            def ep(P_, w_, b_, lo, hi, it_):
                b2, _, ls, vs = O.fm_sgd_epoch_mb(shard, ys, 2, P_, w_, b_, cfg, B, begin=lo, end=hi, it=it_)
                return b2, ls, vs
            P, w, b, ls, vs, it = yield from R.rank_sgd(ep, P, w, b, cfg, shard.n, B, S, it, overlap, world)
            out.append((ls, vs, it))
        return P, w, b, out

    def ada():
        cfg = O.adagrad_cfg()
        P, w, it = P0.copy(), w0.copy(), 1
        hold = [0.0]
        st = O.AdaState(1, D, K, D)
        st.gnorm_P[...] = cfg.eps
        st.gnorm_w[...] = cfg.eps
        st.gnorm_b.value = cfg.eps
        out = []
        for _ in range(2):
            def ep(lo, hi, it_):
                hold[0], _, ls, vs = O.fm_adagrad_epoch_mb(shard, ys, 2, P, w, hold[0], cfg, B, st, begin=lo, end=hi, it=it_)
                return ls, vs
            # AdaGrad between exchanges of more than one mini-batch: the library's NFM_DP_AUTO is the cross rule (csrc/dp.h,
            # round 5: g_sum increments summed, g_norm taking the ranks' agreement); one mini-batch per exchange: the plain sum
            st, ls, vs, it = yield from R.rank_adagrad(ep, st, shard.n, B, S, it, overlap, world, cross_gamma=None if S == 1 else 0.1)
            out.append((ls, vs, it))
        return R._ada_flat(st), out

    return sgd(), ada()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import dp_rule as R

    dist.init_process_group("gloo", rank=rank, world_size=world)
    O, full, y, P0, w0 = _problem()
    shard, ys = _shard(O, full, y, rank, world)
    res = []
    for S, overlap in CASES:
        g_sgd, g_ada = _rank_gens(O, shard, ys, P0, w0, S, overlap, world)
        res.append((R.drive_with_dist(g_sgd, dist, torch), R.drive_with_dist(g_ada, dist, torch)))
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_exchange_rule_world2():
    import torch.multiprocessing as mp

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_rule as R

    O, full, y, P0, w0 = _problem()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for ci, (S, overlap) in enumerate(CASES):
        gens = [_rank_gens(O, *_shard(O, full, y, r, 2), P0, w0, S, overlap, 2) for r in range(2)]
        want_sgd = R.simulate([g[0] for g in gens])
        want_ada = R.simulate([g[1] for g in gens])
        for r in range(2):
            (P, w, b, hist), (st, hist_a) = got[r][ci]
            Pw, ww, bw, histw = want_sgd[r]
            # gloo adds in its own order: two ranks, so a + b == b + a bit for bit
            assert np.array_equal(P, Pw) and np.array_equal(w, ww) and b == bw, (S, overlap, r)
            assert hist == histw
            assert np.array_equal(st, want_ada[r][0]) and hist_a == want_ada[r][1]
        # all replicas leave an epoch identical, the step counter covers the samples of all ranks
        assert np.array_equal(got[0][ci][0][0], got[1][ci][0][0]) and np.array_equal(got[0][ci][1][0], got[1][ci][1][0])
        assert got[0][ci][0][3][-1][2] == 1 + 2 * N and got[0][ci][1][1][-1][2] == 1 + 2 * N
    # the delayed exchange really differs from the immediate one and from no mid-epoch exchange (the rule is exercised)
    assert not np.array_equal(got[0][0][0][0], got[0][1][0][0])


def test_sync_point_agreement():
    """sync points lie after regular mini-batches only and strictly before the last one"""
    import dp_rule as R

    assert R.n_sync_mine(R.batch_bounds(201, 32, False), 32, 2) == 3   # 7 batches (6 regular + tail of 9)
    assert R.n_sync_mine(R.batch_bounds(192, 32, False), 32, 2) == 2   # 6 regular batches: not after the last
    assert R.n_sync_mine(R.batch_bounds(192, 32, False), 32, 0) == 0
    assert R.n_sync_mine(R.batch_bounds(20, 32, False), 32, 1) == 0
    assert R.batch_bounds(70, 32, True) == [0, 1, 33, 65, 70]
