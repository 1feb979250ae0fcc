"""FactorizationMachine.init / randomNormal (model/factorization_machine.nim:125-139,
tensor/tensor.nim:561-580) -- the Box-Muller pairing and the row-major fill order, restated over an injectable uniform
stream (oracle/rng.py), and the host procedures (nfm_rng_*, nimfm_amd/host.py) held to that restatement.  The
generator behind rand(1.0) is Nim's stdlib and stays unpinned (oracle/rng.py header)."""
import ctypes as C
import math

import numpy as np

import oracle.rng as R


def test_pairing_and_fill_order():
    u = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8]
    it = iter(u)
    got = R.random_normal([2, 1, 3], lambda: next(it), loc=1.0, scale=2.0)  # 6 values = 3 draws of (x, y)
    flat = [v for blk in got for row in blk for v in row]
    want = []
    for x, y in ((0.1, 0.2), (0.3, 0.4), (0.5, 0.6)):
        r = math.sqrt(-2 * math.log(1.0 - x))
        want += [1.0 + r * math.cos(2 * math.pi * y) * 2.0, 1.0 + r * math.sin(2 * math.pi * y) * 2.0]
    assert flat == want  # the sine twin of a draw lands on the NEXT element, across the block boundary too
    assert next(it) == 0.7  # exactly 2 uniforms per pair of elements
    # odd count: the last draw's sine half is dropped, both of its uniforms are consumed
    it = iter(u)
    got = R.random_normal([1, 1, 3], lambda: next(it))
    assert len(got[0][0]) == 3 and next(it) == 0.5


def test_host_random_normal_matches_restatement():
    from nimfm_amd import host

    rng = np.random.default_rng(0)
    for shape in ([2, 3, 5], [1, 4, 7], [3, 1, 1]):
        n = int(np.prod(shape))
        u = rng.uniform(0, 1, 2 * ((n + 1) // 2))
        it = iter(u.tolist())
        want = np.array(R.random_normal(shape, lambda: next(it), 0.5, 0.01))
        got = host.randomNormal(shape, 0.5, 0.01, uniform=u)
        assert got.shape == tuple(shape)
        np.testing.assert_allclose(got, want, rtol=1e-14, atol=0)


def test_library_rng_matches_restatement():
    """nfm_rng_* (host code of libnimfm_hip.so, no device work) against the pure-Python restatement: same words,
    same normals, same shuffle"""
    from nimfm_amd import host

    for seed in (1, 7, 123456789, 2 ** 40 + 17):
        g, o = host.NimRand(seed), R.NimRand(seed)
        assert (g.state[0], g.state[1]) == (o.a0, o.a1)
        P = g.randomNormal([2, 3, 5], scale=0.01)
        want = np.array(R.random_normal([2, 3, 5], o.rand1, 0.0, 0.01))
        np.testing.assert_allclose(P, want, rtol=1e-15, atol=0)
        assert (g.state[0], g.state[1]) == (o.a0, o.a1)
        x = np.arange(257, dtype=np.int64)
        g.shuffle(x)
        y = list(range(257))
        o.shuffle(y)
        assert x.tolist() == y and sorted(y) == list(range(257))


def test_fm_init_mirror():
    """host FactorizationMachine.init == restated init: w = 0, b = 0, P filled [o][s][j] after randomize(randomState)"""
    import nimfm_amd as nf

    class X:  # init only asks the dataset for its shape
        nFeatures, nFields = 6, 3

    fm = nf.newFactorizationMachine("regression", degree=3, nComponents=4, randomState=5, scale=0.1)
    fm.init(X)
    P, w, b, _ = R.fm_init(5, 2, 4, 6, 0, 0.1)
    np.testing.assert_allclose(fm.P, np.array(P), rtol=1e-15, atol=0)
    assert fm.P.shape == (2, 4, 6) and not fm.w.any() and fm.intercept == 0.0 and fm.isInitialized
    ffm = nf.newFieldAwareFactorizationMachine("regression", nComponents=2, randomState=5, scale=0.1)
    ffm.init(X)
    o = R.NimRand(5)
    np.testing.assert_allclose(ffm.P, np.array(R.random_normal([3, 6, 2], o.rand1, 0.0, 0.1)), rtol=1e-15, atol=0)
    # warmStart and isInitialized: init is skipped (factorization_machine.nim:129)
    fm.warmStart = True
    before = fm.P.copy()
    fm.randomState = 99
    fm.init(X)
    assert np.array_equal(fm.P, before)
