"""-m gpu: the data-parallel exchange inside the library (csrc/dp.hip, nfm_dp_* / nfm_opt_set_dp) on one MI355X.

* groups of 2 and 3 ranks made by nfm_dp_create_local -- every rank has its own context, stream, dataset shard, model
  replica, optimizer and host thread, all on the one GPU -- run nfm_opt_epoch together; the result is held to the CPU
  restatement of the exchange rule (tests/dp_rule.py, itself checked across real processes in tests/test_dp_gloo.py):
  sync_period 0 / 2 / 3, delayed (overlapped) and immediate exchange, shards of unequal size, SGD and AdaGrad, FM and
  FFM; every replica ends bitwise identical; loss / viol / `it` cover all ranks.
* the RCCL transport (nfm_dp_unique_id / nfm_dp_create, what one-process-per-GPU runs use) with world_size 1: the
  collective really goes through ncclAllReduce and must leave a single replica's training unchanged."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import pytest  # noqa: E402

import dp_rule as R  # noqa: E402
import nimfm_amd as nf  # noqa: E402
import oracle as O  # noqa: E402
from common import assert_close, random_csr  # noqa: E402
from gpu_common import gpu_ffm, gpu_fm  # noqa: E402
from nimfm_amd import dp  # noqa: E402

pytestmark = pytest.mark.gpu
N, D, M, K, B = 1003, 120, 8, 8, 64


def _shards(full, y, world):
    out = []
    for r in range(world):
        lo, hi = dp.shard_bounds(full.n, r, world)
        a, b = full.indptr[lo], full.indptr[hi]
        out.append((O.Dataset(full.indptr[lo:hi + 1] - a, full.indices[a:b], full.data[a:b], hi - lo, full.d,
                              None if full.fields is None else full.fields[a:b], full.n_fields), y[lo:hi]))
    return out


def _run_ranks(world, make_rank):
    """make_rank(r, ctx, group) -> result; one host thread per rank"""
    ctxs = [nf.Context(0) for _ in range(world)]
    groups = dp.Group.local(ctxs)
    res, err = [None] * world, []

    def body(r):
        try:
            res[r] = make_rank(r, ctxs[r], groups[r])
        except BaseException as e:  # noqa: BLE001
            err.append((r, e))

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a rank hangs in the exchange"
    assert not err, err
    info = [g.info() for g in groups]
    for g in groups:
        g.close()
    return res, info


@pytest.mark.parametrize("world,S,overlap,combine", [(2, 0, True, "mean"), (2, 2, True, "mean"), (3, 3, True, "mean"), (2, 3, False, "mean"),
                                                       (3, 2, True, "sum")])
def test_local_group_sgd_and_adagrad_vs_rule(world, S, overlap, combine):
    _vs_rule(N, D, world, S, overlap, combine)


@pytest.mark.parametrize("world,S,combine", [(4, 4, "mean"), (8, 2, "mean"), (8, 0, "mean"), (4, 0, "mean"), (4, 3, "state_mean"),
                                              (8, 0, "state_mean"), (4, 3, "auto"), (4, 0, "auto"), (2, 1, "auto"), (4, 3, "state_rsqrt"),
                                              (4, 3, "state_cross"), (8, 0, "state_cross"), (8, 2, "state_cross")])
def test_local_group_many_ranks_vs_rule(world, S, combine):
    """4 and 8 ranks (tools/dp_convergence.py runs these group sizes), several mid-epoch exchanges per rank; state_mean:
    AdaGrad's state increments averaged instead of summed (NFM_DP_STATE_MEAN; for SGD it is the mean); auto: NFM_DP_AUTO,
    resolved inside the library -- what an optimizer does when its host only ever calls nfm_opt_set_dp (the Nim shim's
    attach(), nimfm.hpp): SGD the mean, AdaGrad summed at sync_period 1 and the cross rule otherwise; state_cross
    (NFM_DP_STATE_CROSS, round 5): AdaGrad's g_sum increments summed, g_norm inflated by the ranks' agreement (csrc/dp.h)"""
    _vs_rule(4003, 300, world, S, True, combine)


@pytest.mark.parametrize("world,S", [(2, 2), (4, 4), (8, 3)])
def test_local_group_sparse_regime_vs_rule(world, S):
    """batch x nnz / d < 1.4: most features of a batch are touched once and are updated by the ROW phase (plan.use_singles),
    the regime of the headline shape -- the exchange must not care where a row was written"""
    _vs_rule(4003, 2000, world, S, True, "mean")


def _vs_rule(N, D, world, S, overlap, combine):
    lib_combine = combine
    if combine == "auto":  # the rule the library is expected to pick by itself
        combine = "state_cross" if S != 1 else "sum"
    full = random_csr(N, D, M, seed=21)
    rng = np.random.default_rng(5)
    y = rng.standard_normal(N)
    P0, w0, b0 = rng.standard_normal((1, K, D)) * 0.1, rng.standard_normal(D) * 0.01, 0.25
    shards = _shards(full, y, world)
    epochs = 2

    # ---- the rule on the CPU ----
    def gens():
        gs, ga = [], []
        for shard, ys in shards:
            def sgd(shard=shard, ys=ys):
                P, w, b, it = P0.copy(), w0.copy(), b0, 1
                cfg = O.sgd_cfg(eta0=0.05)
                hist = []
                for _ in range(epochs):
                    def ep(P_, w_, b_, lo, hi, it_):
                        b2, _, ls, vs = O.fm_sgd_epoch_mb(shard, ys, 2, P_, w_, b_, cfg, B, begin=lo, end=hi, it=it_)
                        return b2, ls, vs
                    P, w, b, ls, vs, it = yield from R.rank_sgd(ep, P, w, b, cfg, shard.n, B, S, it, overlap, world, "mean" if lib_combine in ("auto", "state_rsqrt", "state_cross") else combine)
                    hist.append((vs, ls / N))
                return P, w, b, hist, it

            def ada(shard=shard, ys=ys):
                cfg = O.adagrad_cfg()
                P, w, it = P0.copy(), w0.copy(), 1
                hold = [b0]
                st = O.AdaState(1, D, K, D)
                st.gnorm_P[...] = cfg.eps
                st.gnorm_w[...] = cfg.eps
                st.gnorm_b.value = cfg.eps
                hist = []
                for _ in range(epochs):
                    def ep(lo, hi, it_):
                        hold[0], _, ls, vs = O.fm_adagrad_epoch_mb(shard, ys, 2, P, w, hold[0], cfg, B, st, begin=lo, end=hi, it=it_)
                        return ls, vs
                    st, ls, vs, it = yield from R.rank_adagrad(ep, st, shard.n, B, S, it, overlap, world, 1.0 / world if combine == "state_mean" else (1.0 / np.sqrt(world) if combine == "state_rsqrt" else 1.0),
                                                                 cross_gamma=0.1 if combine == "state_cross" else None)
                    hist.append((vs, ls / N))
                bb = O.fm_adagrad_finalize(2, P, w, hold[0], cfg, it, st)
                return P, w, bb, hist, it

            gs.append(sgd())
            ga.append(ada())
        return gs, ga

    gs, ga = gens()
    want = {"sgd": R.simulate(gs), "adagrad": R.simulate(ga)}

    # ---- the library: one thread per rank ----
    for solver in ("sgd", "adagrad"):
        def make_rank(r, ctx, group, solver=solver):
            shard, ys = shards[r]
            X = nf.CSRDataset(shard.data, shard.indices, shard.indptr, shard.n, D, ctx=ctx)
            fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
            if solver == "sgd":
                opt = nf.newSGD(maxIter=epochs, eta0=0.05, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
            else:
                opt = nf.newAdaGrad(maxIter=epochs, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)
            if lib_combine == "auto":  # only nfm_opt_set_dp, as the Nim / C++ hosts do: the library's default must be the rule
                opt.batch = B
                X.set_targets(ys)
                opt._handle(fm, ctx, "minibatch")
                from nimfm_amd import _capi as capi
                capi.check(capi.lib().nfm_opt_set_dp(opt._h, group.h, S, 1 if overlap else 0))
                for _ in range(epochs):
                    ls, vs = opt._epoch(X, None, 0, shard.n)
                    opt._sync_it()
                    opt.history.append((vs, ls / N))
                opt._finalize_into(fm)
                capi.check(capi.lib().nfm_opt_set_dp(opt._h, None, 0, 0))
            else:
                opt.setDataParallel(group, S, overlap, combine)
                opt.fit(X, ys, fm)
            return fm.P.copy(), fm.w.copy(), fm.intercept, list(opt.history), opt.it

        got, info = _run_ranks(world, make_rank)
        for r in range(world):
            Pw, ww, bw, histw, itw = want[solver][r]
            P, w, b, hist, it = got[r]
            assert it == itw == 1 + epochs * N
            assert abs(b - bw) < 1e-11
            assert_close(w, ww, 1e-9, 1e-13, "%s w rank %d" % (solver, r))
            assert_close(P, Pw, 1e-9, 1e-13, "%s P rank %d" % (solver, r))
            assert_close([h[1] for h in hist], [h[1] for h in histw], 1e-10, 0, "mean loss over all ranks")
            assert_close([h[0] for h in hist], [h[0] for h in histw], 1e-8, 0, "viol over all ranks")
            # every replica holds the same bits
            assert np.array_equal(P, got[0][0]) and np.array_equal(w, got[0][1]) and b == got[0][2]
        n_mid = min(R.n_sync_mine(R.batch_bounds(sh.n, B, False), B, S) for sh, _ in shards)
        assert info[0]["world"] == world and info[0]["collectives"] >= epochs * (1 + (n_mid if solver == "sgd" else 0))


def test_local_group_ffm():
    from common import init_ffm
    world, S, F = 2, 2, 4
    rng = np.random.default_rng(8)
    n, per = 600, 10
    d = F * per
    idx = np.stack([f * per + rng.integers(0, per, size=n) for f in range(F)], axis=1)
    val = rng.uniform(-1, 1, size=(n, F))
    full = O.Dataset(np.arange(n + 1) * F, idx.ravel(), val.ravel(), n, d, fields=np.tile(np.arange(F), n), n_fields=F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, 4, scale=0.05)
    shards = _shards(full, y, world)
    cfg = O.sgd_cfg(eta0=0.01)

    def gen(shard, ys):
        P, w = P0.copy(), w0.copy()

        def ep(P_, w_, b_, lo, hi, it_):
            b2, _, ls, vs = O.ffm_sgd_epoch_mb(shard, ys, P_, w_, b_, cfg, 32, begin=lo, end=hi, it=it_)
            return b2, ls, vs
        return (yield from R.rank_sgd(ep, P, w, b0, cfg, shard.n, 32, S, 1, True, world))

    want = R.simulate([gen(*s) for s in shards])

    def make_rank(r, ctx, group):
        shard, ys = shards[r]
        X = nf.CSRDataset(shard.data, shard.indices, shard.indptr, shard.n, d, fields=shard.fields, nFields=F, ctx=ctx)
        ffm = gpu_ffm("regression", 4, True, True, P0, w0, b0)
        opt = nf.newSGD(maxIter=1, eta0=0.01, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=32)
        opt.setDataParallel(group, S, True)
        opt.fit(X, ys, ffm)
        return ffm.P.copy(), ffm.w.copy(), ffm.intercept, opt.it

    got, _ = _run_ranks(world, make_rank)
    for r in range(world):
        assert got[r][3] == want[r][5] == 1 + n
        assert abs(got[r][2] - want[r][2]) < 1e-11
        assert_close(got[r][1], want[r][1], 1e-9, 1e-13, "w")
        assert_close(got[r][0], want[r][0], 1e-9, 1e-13, "P")
        assert np.array_equal(got[r][0], got[0][0])


def test_rccl_group_world1():
    """the RCCL transport end to end on one GPU: id -> communicator -> ncclAllReduce on the arena (mid-epoch and closing
    exchanges); a single replica's result must be what it is without a group, up to the closing rescale"""
    full = random_csr(N, D, M, seed=4)
    rng = np.random.default_rng(2)
    y = rng.standard_normal(N)
    P0, w0 = rng.standard_normal((1, K, D)) * 0.05, np.zeros(D)
    ctx = nf.default_context()
    grp = dp.Group.rccl(ctx, dp.Group.unique_id(), 0, 1)
    try:
        X = nf.CSRDataset(full.data, full.indices, full.indptr, N, D, ctx=ctx)
        for solver in ("sgd", "adagrad"):
            res = []
            for use_group in (False, True):
                fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, 0.0)
                opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=3, verbose=0, tol=0, shuffle=False, mode="minibatch",
                                                                         batch=B)
                if use_group:
                    opt.setDataParallel(grp, 2, True)
                opt.fit(X, y, fm)
                res.append((fm.P.copy(), fm.w.copy(), fm.intercept, list(opt.history), opt.it))
            assert res[0][4] == res[1][4] == 1 + 3 * N
            assert_close(res[1][0], res[0][0], 1e-12, 1e-15, solver + " P")
            assert_close(res[1][1], res[0][1], 1e-12, 1e-15, solver + " w")
            assert abs(res[1][2] - res[0][2]) < 1e-13
            assert_close([h[1] for h in res[1][3]], [h[1] for h in res[0][3]], 1e-12, 0, "loss")
        info = grp.info()
        assert info["world"] == 1 and info["collectives"] >= 6 and info["bytes"] > 0
    finally:
        grp.close()


@pytest.mark.parametrize("solver,overlap", [("sgd", True), ("sgd", False), ("adagrad", True), ("adagrad", False)])
def test_segment_graphs_between_exchange_points_equal_direct_launches(solver, overlap):
    """With a multi-process (RCCL) group attached, the mini-batches BETWEEN two exchange points are captured as one hipGraph each
    on the first epoch over a plan and replayed afterwards (mb_fm.hip: MbWork::seg_execs); the exchange stays outside the
    capture.  World size 1 through RCCL takes exactly that path on one GPU.  Four epochs, the exchange period changed after
    the second (the graphs must be dropped and cut again), against the same fit with NFM_DP_GRAPH=0 (every launch direct):
    bit for bit, history included."""
    n, b = 64 * 40 + 17, 64  # 40 full mini-batches and a tail: >= 8, the path's own threshold
    full = random_csr(n, D, M, seed=14)
    rng = np.random.default_rng(12)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, K, D)) * 0.05, np.zeros(D)
    ctx = nf.default_context()
    grp = dp.Group.rccl(ctx, dp.Group.unique_id(), 0, 1)
    try:
        X = nf.CSRDataset(full.data, full.indices, full.indptr, n, D, ctx=ctx)
        res = []
        for graph in ("0", "1"):
            old = os.environ.get("NFM_DP_GRAPH")
            os.environ["NFM_DP_GRAPH"] = graph
            try:
                fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, 0.0)
                opt = (nf.newSGD if solver == "sgd" else nf.newAdaGrad)(maxIter=2, verbose=0, tol=0, shuffle=False, mode="minibatch",
                                                                         batch=b)
                opt.setDataParallel(grp, 4, overlap)
                opt.fit(X, y, fm)
                hist = list(opt.history)
                opt.setDataParallel(grp, 7, overlap)  # another period: other cut points
                opt.fit(X, y, fm)
                hist += list(opt.history)
                res.append((fm.P.copy(), fm.w.copy(), fm.intercept, hist, opt.it))
            finally:
                if old is None:
                    os.environ.pop("NFM_DP_GRAPH", None)
                else:
                    os.environ["NFM_DP_GRAPH"] = old
        assert res[0][4] == res[1][4] == 1 + 4 * n
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
        assert res[0][3] == res[1][3], "loss / viol per epoch"
        assert np.isfinite(res[1][0]).all() and not np.array_equal(res[1][0], P0)
    finally:
        grp.close()


def test_data_parallel_training_and_held_out_quality():
    """Convergence evidence for the exchange: a planted degree-2 FM, 4 ranks (a quarter of the samples
    each, replicas reconciled every 8 mini-batches and at the end of every epoch) against ONE rank, the same number of
    epochs, the same step size; held-out RMSE.
      AdaGrad: the state increments are summed -- the state one process would hold: within 10 % of one rank over all samples.
      SGD, mean (default): as stable as one rank, but the model moves only as far as one rank's steps take it: clearly
        behind one rank over all samples (0.87 against 0.27 here), ahead of one rank that only has a quarter of the samples
        (1.17: it overfits them).
      SGD, sum: all ranks' steps land in the model: clearly better than the mean, short of the single rank (steps taken
        from the same stale point overshoot where the ranks' features overlap -- here every feature is shared)."""
    from test_gpu_configs import _planted
    from gpu_common import to_gpu
    Xtr, Xte, ytr, yte, n, d, m, k = _planted("regression")
    world, B, E, S = 4, 64, 12, 8
    P0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01
    w0 = np.zeros(d)
    shards = _shards(Xtr, ytr, world)
    Xte_gpu = to_gpu(Xte)

    def make_opt(solver):
        mk = nf.newSGD if solver == "sgd" else nf.newAdaGrad
        return mk(maxIter=E, eta0=0.05, alpha=1e-5, beta=1e-5, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=B)

    def rmse(P, w, b):
        return float(np.sqrt(np.mean((O.fm_decision_function(Xte, 2, P, w, b) - yte) ** 2)))

    def single(solver, X, y):
        fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
        make_opt(solver).fit(to_gpu(X), y, fm)
        return rmse(fm.P, fm.w, fm.intercept)

    def ranks(solver, combine):
        def make_rank(r, ctx, group):
            shard, ys = shards[r]
            X = nf.CSRDataset(shard.data, shard.indices, shard.indptr, shard.n, d, ctx=ctx)
            fm = gpu_fm("regression", 2, k, "explicit", True, True, P0, w0, 0.0)
            opt = make_opt(solver)
            opt.setDataParallel(group, S, True, combine)
            opt.fit(X, ys, fm)
            return fm.P.copy(), fm.w.copy(), fm.intercept
        got, _ = _run_ranks(world, make_rank)
        return rmse(*got[0])

    sgd_all, sgd_quarter = single("sgd", Xtr, ytr), single("sgd", *shards[0])
    sgd_mean, sgd_sum = ranks("sgd", "mean"), ranks("sgd", "sum")
    ada_all, ada_dp = single("adagrad", Xtr, ytr), ranks("adagrad", "mean")
    print("held-out RMSE (target spread %.3f): SGD one rank, all samples %.4f; one rank, a quarter %.4f; 4 ranks mean %.4f, "
          "sum %.4f.  AdaGrad one rank %.4f, 4 ranks %.4f" % (yte.std(), sgd_all, sgd_quarter, sgd_mean, sgd_sum, ada_all, ada_dp))
    assert sgd_all < 0.5 * yte.std()
    assert sgd_mean <= 1.1 * sgd_quarter
    assert sgd_sum <= 0.7 * sgd_mean and np.isfinite(sgd_sum)
    assert ada_dp <= 1.1 * ada_all


@pytest.mark.parametrize("solver", ["sgd", "adagrad"])
def test_fit_devices_keyword(solver):
    """fit(X, y, fm, maxThreads, miniBatchSize=..., syncPeriod=..., devices=[0, 0]): the explicit knobs of the maxThreads
    overloads (the thread count itself selects the mode and nothing else).  One process, two ranks on the one GPU: the
    result equals the same two ranks driven by hand (a thread, a context, a shard, a replica, a group handle each)."""
    full = random_csr(N, D, M, seed=33)
    rng = np.random.default_rng(34)
    y = rng.standard_normal(N)
    P0, w0, b0 = rng.standard_normal((1, K, D)) * 0.1, rng.standard_normal(D) * 0.01, 0.1
    world, S = 2, 2
    mk = (lambda: nf.newSGD(maxIter=2, eta0=0.05, verbose=0, tol=0, shuffle=False, batch=999)) if solver == "sgd" else \
         (lambda: nf.newAdaGrad(maxIter=2, verbose=0, tol=0, shuffle=False, batch=999))
    shards = _shards(full, y, world)

    def make_rank(r, ctx, group):
        shard, ys = shards[r]
        X = nf.newCSRDataset(shard.data, shard.indices, shard.indptr, shard.n, shard.d, ctx=ctx)
        fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
        opt = mk()
        opt.setDataParallel(group, syncPeriod=S)
        opt.fit(X, ys, fm, maxThreads=4, miniBatchSize=B)
        return fm.P.copy(), fm.w.copy(), fm.intercept, opt.it, list(opt.history)

    res, _ = _run_ranks(world, make_rank)
    X = nf.newCSRDataset(full.data, full.indices, full.indptr, full.n, full.d)
    fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
    opt = mk()
    opt.fit(X, y, fm, maxThreads=4, miniBatchSize=B, syncPeriod=S, devices=[0, 0])
    assert opt.batch == 999  # the optimizer's own default is untouched; maxThreads = 4 did not become a batch size either
    assert np.array_equal(fm.P, res[0][0]) and np.array_equal(fm.w, res[0][1]) and fm.intercept == res[0][2]
    assert opt.it == res[0][3] and opt.history == res[0][4]
    assert np.array_equal(res[0][0], res[1][0])  # the replicas agree


def test_rank_with_an_empty_range_still_joins_the_exchange():
    """dp.shard_bounds hands ranks 0 .. W-2 nothing when there are fewer samples than ranks; such a rank's
    nfm_opt_epoch(begin == end) must issue the same collectives as its peers (they would wait for it forever) and leave
    with the group's sums, step counter and model"""
    full = random_csr(40, D, M, seed=9)
    rng = np.random.default_rng(10)
    y = rng.standard_normal(full.n)
    P0, w0, b0 = rng.standard_normal((1, K, D)) * 0.1, np.zeros(D), 0.0
    world = 3

    def make_rank(r, ctx, group):
        X = nf.newCSRDataset(full.data, full.indices, full.indptr, full.n, full.d, ctx=ctx)
        X.set_targets(y)
        fm = gpu_fm("regression", 2, K, "explicit", True, True, P0, w0, b0)
        opt = nf.newSGD(maxIter=1, eta0=0.05, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=16)
        opt.setDataParallel(group, syncPeriod=1)
        opt._handle(fm, ctx, "minibatch")
        ls, vs = opt._epoch(X, None, 0, full.n if r == world - 1 else 0)  # only the last rank has samples
        opt._sync_it()
        opt._finalize_into(fm)
        return fm.P.copy(), fm.w.copy(), fm.intercept, opt.it, ls, vs

    res, info = _run_ranks(world, make_rank)
    for r in range(world):
        assert np.array_equal(res[r][0], res[world - 1][0]) and np.array_equal(res[r][1], res[world - 1][1])
        assert res[r][3] == 1 + full.n and res[r][4] == res[world - 1][4]
    assert not np.array_equal(res[0][0], P0)  # and the model did move
