"""Shared fixtures for the parity tests.

Data follow the reference's planted-model generator (tests/utils.nim:7-79): dense
uniform(-1,1) entries, |x| < threshold zeroed, labels = brute-force model output.
numpy's RNG replaces Nim's (RNG parity is unpinned, so parameters
are always injected and permutations passed explicitly).
"""
import numpy as np

import oracle as O


def assert_close(actual, desired, rtol=1e-6, atol=1e-9, what=""):
    """tests/utils.nim:82-105 checkAlmostEqual."""
    actual, desired = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    assert actual.shape == desired.shape, (what, actual.shape, desired.shape)
    diff = np.abs(actual - desired)
    bound = atol + np.abs(desired) * rtol
    bad = ~(diff <= bound)
    assert not bad.any(), "%s: %d/%d off, max diff %.3e (bound %.3e)" % (
        what, bad.sum(), bad.size, diff[bad].max() if bad.any() else 0.0, bound[bad].min() if bad.any() else 0.0)


def random_normal(rng, shape, scale):
    return rng.standard_normal(shape) * scale


def make_fm_dataset(n, d, degree, k, seed, fit_lower="explicit", fit_linear=True, fit_intercept=True, scale=1.0,
                    threshold=0.0):
    """tests/utils.nim:29-47 createFMDataset (CSR flavour)."""
    rng = np.random.default_rng(seed)
    Xd = rng.uniform(-1.0, 1.0, size=(n, d))
    Xd[np.abs(Xd) < threshold] = 0.0
    n_ord = O.n_orders(degree, fit_lower)
    n_aug = O.n_augments(degree, fit_lower, fit_linear)
    P = random_normal(rng, (n_ord, k, d + n_aug), scale)
    y = O.slow_fm_decision_function(Xd, degree, P, np.zeros(d), 0.0, n_aug)
    return O.Dataset.from_dense(Xd), Xd, y


def make_ffm_dataset(n, d, n_fields, k, seed, scale=1.0, threshold=0.0):
    """tests/utils.nim:50-79 createFFMDataset."""
    rng = np.random.default_rng(seed)
    Xd = rng.uniform(-1.0, 1.0, size=(n, d))
    Xd[np.abs(Xd) < threshold] = 0.0
    field_of = np.arange(d) // (d // n_fields)
    P = random_normal(rng, (n_fields, d, k), scale)
    y = O.slow_ffm_decision_function(Xd, field_of, n_fields, P, np.zeros(d), 0.0)
    return O.Dataset.from_dense(Xd, field_of, n_fields), Xd, field_of, y


def init_fm(d, degree, k, fit_lower, fit_linear, seed=1, scale=0.01):
    """model/factorization_machine.nim:125-139 init: w=0, P~N(0,scale^2), b=0."""
    rng = np.random.default_rng(seed)
    n_ord = O.n_orders(degree, fit_lower)
    n_aug = O.n_augments(degree, fit_lower, fit_linear)
    return random_normal(rng, (n_ord, k, d + n_aug), scale), np.zeros(d), 0.0, n_aug


def init_ffm(d, n_fields, k, seed=1, scale=0.01):
    rng = np.random.default_rng(seed)
    return random_normal(rng, (n_fields, d, k), scale), np.zeros(d), 0.0


def make_perms(n, epochs, seed=7):
    rng = np.random.default_rng(seed)
    return np.stack([rng.permutation(n) for _ in range(epochs)]).astype(np.int64)


def random_csr(n, d, m, seed, sorted_idx=True):
    """Synthetic CSR with exactly m distinct indices per row, values U(-1,1)."""
    rng = np.random.default_rng(seed)
    idx = np.empty((n, m), dtype=np.int64)
    for i in range(n):
        idx[i] = rng.choice(d, size=m, replace=False)
    if sorted_idx:
        idx.sort(axis=1)
    val = rng.uniform(-1.0, 1.0, size=(n, m))
    indptr = np.arange(n + 1, dtype=np.int64) * m
    return O.Dataset(indptr, idx.ravel(), val.ravel(), n, d)
