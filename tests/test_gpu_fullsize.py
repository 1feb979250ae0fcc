"""-m gpu: BASELINE.json configs[1] at FULL size (synthetic CSR 1e6 x 1e5, 32 nnz/row, k = 16, SGD,
Logistic) -- the configuration bench.py reports -- against the CPU restatement of the mini-batch
rule, plus size-independent properties of the path:
  * determinism: two runs give bitwise identical parameters (no atomics anywhere),
  * composition: one epoch call == two calls split at a batch boundary (nCalls callbacks),
  * permutation equivariance of decisionFunction, linearity of the linear term,
  * idempotence of finalize (warm-start invariant, tests/test_sgd.nim:58-89)."""
import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close
from gpu_common import gpu_fm, to_gpu

pytestmark = pytest.mark.gpu
N, D, M, K, B = 1_000_000, 100_000, 32, 16, 32768


def big_csr(n, d, m, seed):
    rng = np.random.default_rng(seed)
    idx = np.sort(rng.integers(0, d, size=(n, m)), axis=1)
    while True:
        bad = np.nonzero((idx[:, 1:] == idx[:, :-1]).any(axis=1))[0]
        if len(bad) == 0:
            break
        idx[bad] = np.sort(rng.integers(0, d, size=(len(bad), m)), axis=1)
    val = rng.uniform(-1.0, 1.0, size=(n, m))
    return O.Dataset(np.arange(n + 1, dtype=np.int64) * m, idx.ravel(), val.ravel(), n, d)


@pytest.fixture(scope="module")
def problem():
    Xo = big_csr(N, D, M, 42)
    rng = np.random.default_rng(1)
    Pt = rng.standard_normal((1, K, D)) * 0.1
    y = np.sign(O.fm_decision_function(Xo, 2, Pt, rng.standard_normal(D) * 0.1, 0.0))
    P0, w0 = rng.standard_normal((1, K, D)) * 0.01, np.zeros(D)
    return Xo, to_gpu(Xo), y, P0, w0


@pytest.mark.parametrize("cap,B", [(1.0, 32768), (16.0, 32768), (32.0, 65536)])
def test_cfg2_fullsize_vs_mb_oracle(problem, cap, B):
    """cap = 16 is the rule bench.py trains cfg2 / the headline / cfg5 with (nfm_opt_set_touch_cap: up to 16 of a batch's
    steps on a coordinate summed, as that many Hogwild threads of optimizer/sgd_multi.nim:83-101 would; every feature is
    touched ~10 times per batch here); cap = 1 the library's default (the per-coordinate mean); (32, 65536): what bench.py quotes cfg2 at
    since the end of round 5 (~21 touches per feature and batch, up to 32 summed)"""
    Xo, X, y, P0, w0 = problem
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    hist = []
    for _ in range(2):
        b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, O.sgd_cfg(loss="logistic"), B, it=it, touch_cap=cap)
        hist.append((vs, ls / N))
    fm = gpu_fm("classification", 2, K, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=B, touchCap=cap)
    sgd.fit(X, y, fm)
    assert sgd.it == it == 2 * N + 1
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close([h[1] for h in sgd.history], [h[1] for h in hist], 1e-11, 0, "mean loss")
    assert_close([h[0] for h in sgd.history], [h[0] for h in hist], 1e-9, 0, "viol")
    # predictions on the full set: north_star tolerance 1e-6 relative, asserted at 1e-10
    assert_close(fm.decisionFunction(X), O.fm_decision_function(Xo, 2, P, w, b), 1e-10, 1e-13, "decision")
    assert sgd.history[1][1] < sgd.history[0][1] < np.log(2.0)  # the loss goes down


def test_cfg2_fullsize_mbpsgd_vs_oracle(problem):
    """SURVEY 8(f) rank 3 at cfg2's size: one outer iteration of MBPSGD (default mini-batch d n / nnz = 3125 -> 320
    mini-batches, SquaredL12: the row-parallel threshold passes, d > 16384) against the C restatement"""
    Xo, X, y, P0, w0 = problem
    Bp = (D * N) // (N * M)
    inner = (N - 1) // Bp + 1
    stream = np.concatenate([np.arange(N, dtype=np.int64), np.arange(Bp * inner - N, dtype=np.int64)])
    cfg = O.psgd_cfg(eta0=0.5, gamma=1e-4, loss="logistic")
    P, w = P0.copy(), w0.copy()
    b, it, ls = O.fm_mbpsgd_epoch(Xo, y, 2, P, w, 0.0, cfg, stream, Bp, it=1)
    fm = gpu_fm("classification", 2, K, "explicit", True, True, P0, w0, 0.0)
    opt = nf.newMBPSGD(maxIter=1, eta0=0.5, gamma=1e-4, loss="logistic", verbose=0, tol=-1.0)
    opt.it = 1
    opt.fit(X, y, fm, stream=stream)
    assert opt.batch == Bp and opt.it == it == 1 + inner
    assert abs(opt.history[0][1] - ls / (Bp * inner)) < 1e-11
    assert abs(fm.intercept - b) < 1e-10
    assert_close(fm.w, w, 1e-8, 1e-13, "w")
    assert_close(fm.P, P, 1e-8, 1e-13, "P")
    assert np.array_equal(fm.P == 0.0, P == 0.0) and (P == 0.0).any() and (P != 0.0).any()


def test_cfg2_fullsize_predict_all_with_grad(problem):
    """pgd.predictAllWithGrad over the whole shard as ONE batch (every feature is touched ~320 times: the segment path)"""
    Xo, X, y, P0, w0 = problem
    wq = np.random.default_rng(5).standard_normal(D) * 0.05
    yp, dL, gP, gw, gb = O.fm_predict_all_with_grad(Xo, y, 2, P0 * 10, wq, 0.1, "logistic")
    fm = gpu_fm("classification", 2, K, "explicit", True, True, P0 * 10, wq, 0.1)
    yp_g, dL_g, g = nf.predictAllWithGrad(X, y, fm, loss="logistic")
    assert_close(yp_g, yp, 1e-10, 1e-13, "yPred")
    assert_close(dL_g, dL, 1e-9, 1e-13, "dL")
    assert_close(g["P"], gP, 1e-8, 1e-15, "grad P")
    assert_close(g["w"], gw, 1e-8, 1e-15, "grad w")
    assert abs(g["intercept"] - gb) < 1e-12


def test_determinism_and_composition(problem):
    Xo, X, y, P0, w0 = problem
    runs = []
    for split in (None, None, 20 * B):
        fm = gpu_fm("classification", 2, K, "explicit", True, True, P0, w0, 0.0)
        sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=B)
        X.set_targets(y)
        sgd._handle(fm, X.ctx, "minibatch")
        if split is None:
            sgd._epoch(X, None, 0, N)
        else:
            sgd._epoch(X, None, 0, split)
            sgd._epoch(X, None, split, N)
        sgd._finalize_into(fm)
        P1, w1, b1 = fm.P.copy(), fm.w.copy(), fm.intercept
        sgd._finalize_into(fm)  # finalize is idempotent
        assert np.array_equal(P1, fm.P) and np.array_equal(w1, fm.w) and b1 == fm.intercept
        runs.append((P1, w1, b1))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]
    assert_close(runs[2][0], runs[0][0], 1e-12, 1e-15, "split epoch P")  # scales are re-based at the split
    assert_close(runs[2][1], runs[0][1], 1e-12, 1e-15, "split epoch w")


def test_predict_properties(problem):
    Xo, X, y, P0, w0 = problem
    rng = np.random.default_rng(5)
    w1, w2 = rng.standard_normal(D), rng.standard_normal(D)
    dec = {}
    for name, (w, b) in {"w1": (w1, 0.5), "w2": (w2, -1.0), "sum": (w1 + w2, 0.0), "zero": (np.zeros(D), 0.0)}.items():
        dec[name] = gpu_fm("regression", 2, K, "explicit", True, True, P0, w, b).decisionFunction(X)
    # linear term + intercept are additive on top of the pairwise term
    assert_close(dec["w1"] + dec["w2"] - dec["zero"], dec["sum"] - 0.5, 1e-9, 1e-9, "linearity")
    # row permutation equivariance (a fresh dataset with permuted rows)
    sub = rng.permutation(200_000)
    Xp = O.Dataset(np.arange(len(sub) + 1, dtype=np.int64) * M, Xo.indices.reshape(N, M)[sub].ravel(),
                   Xo.data.reshape(N, M)[sub].ravel(), len(sub), D)
    got = gpu_fm("regression", 2, K, "explicit", True, True, P0, w1, 0.5).decisionFunction(to_gpu(Xp))
    assert np.array_equal(got, dec["w1"][sub])


# ---- the north-star shape: d = 1e6, 64 nnz/row, k = 64, mini-batch 8192 (BASELINE.json configs[2] /
# north_star).  n is cut to what the CPU restatement finishes in seconds; d, m, k and the batch size --
# everything that decides which kernels run and how a batch's touches collide (about 60 % of a batch's
# touches are singles, most other features are touched twice) -- are the full-size values.
HN, HD, HM, HK, HB = 120_000, 1_000_000, 64, 64, 8192


@pytest.fixture(scope="module")
def headline_problem():
    Xo = big_csr(HN, HD, HM, 43)
    rng = np.random.default_rng(2)
    y = np.sign(rng.standard_normal(HN))
    P0, w0 = (rng.standard_normal((1, HK, HD)) * 0.01), np.zeros(HD)
    return Xo, to_gpu(Xo), y, P0, w0


@pytest.mark.parametrize("cap", [1.0, 16.0])
def test_headline_shape_sgd_vs_mb_oracle(headline_problem, cap):
    """(cap = 16: bench.py's rule for this shape; about 40 % of a batch's touches share their feature with another)"""
    Xo, X, y, P0, w0 = headline_problem
    P, w = P0.copy(), w0.copy()
    b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, 0.0, O.sgd_cfg(loss="logistic"), HB, it=1, touch_cap=cap)
    runs = []
    for _ in range(2):
        fm = gpu_fm("classification", 2, HK, "explicit", True, True, P0, w0, 0.0)
        sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=HB, touchCap=cap)
        sgd.fit(X, y, fm)
        runs.append((fm.P.copy(), fm.w.copy(), fm.intercept, sgd.history[0]))
    Pg, wg, bg, h = runs[0]
    assert abs(bg - b) < 1e-11
    assert_close(wg, w, 1e-9, 1e-13, "w")
    assert_close(Pg, P, 1e-9, 1e-13, "P")
    assert_close(h[1], ls / HN, 1e-11, 0, "mean loss")
    assert_close(h[0], vs, 1e-9, 0, "viol")
    # bitwise reproducible
    assert np.array_equal(Pg, runs[1][0]) and np.array_equal(wg, runs[1][1]) and bg == runs[1][2]


def test_headline_shape_adagrad_vs_mb_oracle(headline_problem):
    Xo, X, y, P0, w0 = headline_problem
    n = 4 * HB + 1  # the it == 1 singleton batch + four full batches
    Xs = O.Dataset(Xo.indptr[: n + 1], Xo.indices[: n * HM], Xo.data[: n * HM], n, HD)
    cfg = O.adagrad_cfg(loss="squared")
    P, w = P0.copy(), w0.copy()
    st = O.AdaState(1, HD, HK, HD)
    b, it, ls, vs = O.fm_adagrad_epoch_mb(Xs, y[:n], 2, P, w, 0.0, cfg, HB, st, it=1)
    b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    fm = gpu_fm("regression", 2, HK, "explicit", True, True, P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, loss="squared", mode="minibatch", batch=HB)
    ada.fit(to_gpu(Xs), y[:n], fm)
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close(ada.history[0][0], vs, 1e-9, 0, "viol")


# ---- the headline at the batches bench.py quotes since round 5 (touch rate lambda = B m / d = 8.4 / 16.8: every feature of
# the model is touched in every batch, a row is written once per ~8 / ~17 touches, the touch cap bites on the tail of the
# Poisson touch counts).  Two full batches + a ragged one; the bench's cap.
@pytest.mark.parametrize("HB2,cap", [(131072, 16.0), (262144, 32.0)])
def test_headline_shape_at_the_bench_batch(HB2, cap):
    """(131072, cap 16): the batch of most of round 5; (262144, cap 32): what bench.py quotes since its end -- lambda = 16.8 touches per
    coordinate and batch, up to 32 of them summed"""
    n = 2 * HB2 + 4001
    Xo = big_csr(n, HD, HM, 47)
    rng = np.random.default_rng(5)
    y = np.sign(rng.standard_normal(n))
    P0, w0 = (rng.standard_normal((1, HK, HD)) * 0.01), np.zeros(HD)
    cfg = O.sgd_cfg(loss="logistic")
    P, w = P0.copy(), w0.copy()
    b, it, ls, vs = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, 0.0, cfg, HB2, it=1, touch_cap=cap)
    # a second epoch over a permuted order (the plan of a shuffled epoch: the bucketing path)
    perm = np.random.default_rng(6).permutation(n).astype(np.int64)
    b, it, ls2, vs2 = O.fm_sgd_epoch_mb(Xo, y, 2, P, w, b, cfg, HB2, it=it, touch_cap=cap, perm=perm)
    X = to_gpu(Xo)
    fm = gpu_fm("classification", 2, HK, "explicit", True, True, P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, loss="logistic", mode="minibatch", batch=HB2, touchCap=cap)
    sgd._handle(fm, X.ctx, "minibatch")
    X.set_targets(y)
    l1, v1 = sgd._epoch(X, None, 0, n)
    sgd.it += n
    l2, v2 = sgd._epoch(X, perm, 0, n)
    sgd.it += n
    sgd._finalize_into(fm)
    assert sgd.it == it
    assert abs(fm.intercept - b) < 1e-11
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close([l1, l2], [ls, ls2], 1e-11, 0, "loss sums")
    assert_close([v1, v2], [vs, vs2], 1e-9, 0, "viol")
