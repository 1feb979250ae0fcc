"""-m gpu, run by tests/test_gpu_seqwin.py::test_abort_paths_in_the_test_hooks_build in ONE child process against
libnimfm_hip_testhooks.so (the library built with -DNFM_TEST_HOOKS): the paths of the dependency window that only a fault can
reach.  Not collected by the default run (no test_ prefix): the product library has no hooks to provoke them with."""
import os

import numpy as np
import pytest

import nimfm_amd as nf
from common import assert_close, random_csr
from gpu_common import to_gpu
from test_gpu_seqwin import _fallbacks, _tol, env, fit, same_b, same_bits, window_flavour  # noqa: F401  (the flavour fixture is autouse)

pytestmark = pytest.mark.gpu


def test_this_is_the_hooks_build():
    assert os.environ.get("NFM_TEST_HOOKS_CHILD") == "1" and "testhooks" in os.environ.get("NIMFM_HIP_LIB", "")


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_aborted_window_is_put_back_and_rerun_by_the_one_workgroup_kernel(kind, capfd):
    """The window's workgroups wait for each other.  NFM_SEQ_WIN_TEST_DEAD_SLOT makes worker 3 leave at once, as a workgroup
    that never became resident would (CUs held by another tenant): the conductor's wait for that worker's mailbox runs
    into its 4 s wall-clock limit, the launch aborts with samples 0 ... 2 applied and others half-way.  nfm_opt_epoch
    must put the parameters (and AdaGrad's state) back to what they were when the call began and run the call through
    the one-workgroup kernel: the fit equals the NFM_SEQ_WIN=0 fit bit for bit, nothing is reported as an error."""
    n, d, k = 3000, 500, 16
    Xo = random_csr(n, d, 8, seed=71)
    y = np.random.default_rng(72).standard_normal(n)
    rng = np.random.default_rng(73)
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
    ref = fit(kind, 0, 16, Xo, y, "regression", k, P0, w0, b0, 1)
    capfd.readouterr()
    with env(NFM_SEQ_WIN_TEST_DEAD_SLOT=3):
        got = fit(kind, 2, 16, Xo, y, "regression", k, P0, w0, b0, 1, expect_fallbacks=1)
    err = capfd.readouterr().err
    assert "falling back to the one-workgroup kernel" in err, err[-500:]
    same_bits(got[0], ref[0], "P")
    same_bits(got[1], ref[1], "w")
    assert same_b(got[2], ref[2]) and got[3] == ref[3]
    assert_close([h[1] for h in got[4]], [h[1] for h in ref[4]], 0, 0, "loss per epoch (the fallback's own sums)")
    if kind == "adagrad":
        for g, h, name in zip(got[5], ref[5], ["g_sum.P", "g_norm.P", "g_sum.w", "g_norm.w", "g_sum.b", "g_norm.b"]):
            same_bits(np.atleast_1d(g), np.atleast_1d(h), name)
    # and the optimizer keeps working: the next fit of the same shape goes through the window again
    win = fit(kind, 2, 16, Xo, y, "regression", k, P0, w0, b0, 1)
    same_bits(win[0], ref[0], "P after the hook is gone")


def test_an_optimizer_stops_asking_for_the_window_after_two_aborted_launches():
    """A launch that cannot finish costs its 4 s limit.  One abort: the next launch of that optimizer uses 128 workers instead
    of a worker on every CU; two: the optimizer is no longer offered the window (nfm_opt_epoch, SeqWin::fallbacks) -- a tenant
    holding CUs must not cost every call 4 s.  Results stay those of the one-workgroup kernel throughout."""
    n, d, k = 600, 300, 64
    Xo = random_csr(n, d, 8, seed=91)
    y = np.random.default_rng(92).standard_normal(n)
    rng = np.random.default_rng(93)
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
    X = to_gpu(Xo)
    ctx = nf.default_context()

    def three_fits(win, dead):
        with env(NFM_SEQ_WIN=win):
            os.environ.pop("NFM_SEQ_WIN_W", None)
            fm = nf.newFactorizationMachine("regression", nComponents=k, fitIntercept=False, warmStart=True)
            fm.set_params(P0, w0, b0)
            opt = nf.newSGD(maxIter=1, verbose=0, tol=0, shuffle=False, mode="sequential")
            seen = []
            for r in range(3):
                ctx.timing_enable(True)
                ctx.timing_reset()
                if dead and r < 2:
                    with env(NFM_SEQ_WIN_TEST_DEAD_SLOT=5):
                        opt.fit(X, y, fm)
                else:
                    opt.fit(X, y, fm)
                seen.append((ctx.timing_get("seq_window_launch")[0], _fallbacks()))
                ctx.timing_enable(False)
            return fm.P.copy(), fm.w.copy(), seen

    ref = three_fits(0, False)
    got = three_fits(2, True)
    same_bits(got[0], ref[0], "P")
    same_bits(got[1], ref[1], "w")
    assert [s_[1] for s_ in got[2]] == [1, 1, 0], got[2]  # (the counter is reset per fit here: one fallback each in fits 1 and 2)
    assert got[2][0][0] > 0 and got[2][1][0] > 0 and got[2][2][0] == 0, "the third fit must not have tried the window"


@pytest.mark.parametrize("kind", ["sgd", "adagrad"])
def test_no_memory_for_the_snapshot_means_the_one_workgroup_kernel(kind):
    """nfm_opt_epoch snapshots the model (+ AdaGrad's state) before a window launch.  A model beyond half of the free memory
    has no room for that copy: the call must run in the one-workgroup kernel (which needs none), not fail
    (NFM_TEST_NO_SNAPSHOT=1 makes the allocation fail)."""
    n, d, k = 3000, 500, 16
    Xo = random_csr(n, d, 8, seed=75)
    y = np.random.default_rng(76).standard_normal(n)
    rng = np.random.default_rng(77)
    P0, w0, b0 = rng.standard_normal((1, k, d)) * 0.05, rng.standard_normal(d) * 0.01, 0.1
    ref = fit(kind, 0, 16, Xo, y, "regression", k, P0, w0, b0, 1)
    ctx = nf.default_context()
    before = ctx.timing_get("seq_window_no_snapshot")[0], ctx.timing_get("seq_window_launch")[0]
    with env(NFM_TEST_NO_SNAPSHOT=1):
        got = fit(kind, 2, 16, Xo, y, "regression", k, P0, w0, b0, 1)
    assert ctx.timing_get("seq_window_no_snapshot")[0] == before[0] + 1 and ctx.timing_get("seq_window_launch")[0] == before[1]
    for g, h in zip(got[:2], ref[:2]):
        assert np.array_equal(g, h)  # (both ran the one-workgroup kernel: bit for bit whatever the flavour)
    assert got[2] == ref[2] and got[3] == ref[3]


@pytest.mark.parametrize("solver,mode", [("sgd", "minibatch"), ("adagrad", "minibatch"), ("sgd", "sequential"), ("adagrad", "sequential")])
def test_an_epoch_of_more_entries_than_one_call_holds_is_walked_in_pieces(solver, mode):
    """nfm_opt_epoch cuts a range of more than 2^31 - 1 entries into consecutive pieces (whole mini-batches; AdaGrad's first
    step stays a mini-batch of its own) -- 288 GB hold such datasets.  NFM_TEST_MAX_EPOCH_NNZ (this build only) lowers the bound
    so that a 5000-sample epoch becomes several pieces: the fit must equal the uncut one bit for bit (mini-batch mode: cuts at
    batch boundaries; sequential mode: the one-workgroup kernel on any cut), with and without a permutation."""
    n, d, m, k, B = 5000, 400, 8, 8, 256
    Xo = random_csr(n, d, m, seed=81)
    rng = np.random.default_rng(82)
    y = rng.standard_normal(n)
    P0, w0 = rng.standard_normal((1, k, d)) * 0.05, np.zeros(d)
    perms = np.stack([np.arange(n), np.random.default_rng(83).permutation(n)]).astype(np.int64)

    def run():
        X = to_gpu(Xo)
        fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
        fm.set_params(P0, w0, 0.0)
        kw = dict(maxIter=2, verbose=0, tol=0, mode=mode, batch=B) if mode == "minibatch" else dict(maxIter=2, verbose=0, tol=0)
        opt = nf.newSGD(**kw) if solver == "sgd" else nf.newAdaGrad(**kw)
        with env(NFM_SEQ_WIN=0):
            opt.fit(X, y, fm, perms=perms)
        return np.array(fm.P), np.array(fm.w), fm.intercept, opt.it, list(opt.history)

    ref = run()
    # (8 + 8 guard entries per row) x 700 samples per piece -> eight pieces, none a whole number of mini-batches before rounding
    with env(NFM_TEST_MAX_EPOCH_NNZ=16 * 700):
        got = run()
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), "parameters of the cut epoch differ"
    if mode == "minibatch":  # (and the bound is really looked at: a piece must hold at least one whole mini-batch)
        from nimfm_amd._capi import NfmError
        with env(NFM_TEST_MAX_EPOCH_NNZ=16 * 100), pytest.raises(NfmError, match="one mini-batch of 256 samples"):
            run()
    assert got[2] == ref[2] and got[3] == ref[3]
    assert_close([h[0] for h in got[4]], [h[0] for h in ref[4]], 1e-12, 0, "viol (summed piece by piece)")
    assert_close([h[1] for h in got[4]], [h[1] for h in ref[4]], 1e-12, 0, "loss")
