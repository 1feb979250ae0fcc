"""-m gpu: decisionFunction on the MI355X vs the CPU oracle (through the C ABI).

north_star tolerance: 1e-6 relative on fp64 predictions; asserted here at rtol 1e-10 / atol 1e-12
(the only differences are summation order and fused multiply-adds)."""
import itertools

import numpy as np
import pytest

import oracle as O
from common import assert_close, init_fm, make_ffm_dataset, make_fm_dataset
from gpu_common import gpu_ffm, gpu_fm, ragged_csr, to_gpu

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-10, 1e-12


@pytest.mark.parametrize("degree,fit_lower,fit_linear", itertools.product([2, 3, 4, 5], ["explicit", "none", "augment"],
                                                                          [True, False]))
def test_fm_reference_grid(degree, fit_lower, fit_linear):
    n, d, k = 80, 8, 4
    Xo, Xd, _ = make_fm_dataset(n, d, degree, k, 5, fit_lower, fit_linear, threshold=0.3)
    P, _, _, n_aug = init_fm(d, degree, k, fit_lower, fit_linear, scale=0.5)
    w = np.random.default_rng(3).standard_normal(d)
    fm = gpu_fm("regression", degree, k, fit_lower, fit_linear, True, P, w, 0.25)
    got = fm.decisionFunction(to_gpu(Xo))
    assert_close(got, O.fm_decision_function(Xo, degree, P, w, 0.25, n_aug), RTOL, ATOL)
    # and against the brute-force definition (tests/model/fm_slow.nim:42-73), the reference's own tolerance
    assert_close(got, O.slow_fm_decision_function(Xd, degree, P, w, 0.25, n_aug), 1e-6, 1e-9)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 7, 8, 16, 30, 33, 64, 100, 128])
def test_fm_component_counts_and_ragged_rows(k):
    n, d = 257, 300
    Xo = ragged_csr(n, d, seed=k)
    rng = np.random.default_rng(k)
    for degree, fit_lower in [(2, "explicit"), (3, "explicit"), (3, "augment")]:
        n_ord, n_aug = O.n_orders(degree, fit_lower), O.n_augments(degree, fit_lower, True)
        P = rng.standard_normal((n_ord, k, d + n_aug)) * 0.2
        w = rng.standard_normal(d)
        fm = gpu_fm("regression", degree, k, fit_lower, True, True, P, w, -1.5)
        assert_close(fm.decisionFunction(to_gpu(Xo)), O.fm_decision_function(Xo, degree, P, w, -1.5, n_aug), RTOL, ATOL)


def test_fm_predict_proba_score():
    n, d, k = 200, 20, 8
    Xo, Xd, y = make_fm_dataset(n, d, 2, k, 9, threshold=0.5)
    P, w, b, _ = init_fm(d, 2, k, "explicit", True, scale=0.3)
    fm = gpu_fm("classification", 2, k, "explicit", True, True, P, w, 0.1)
    X = to_gpu(Xo)
    dec = O.fm_decision_function(Xo, 2, P, w, 0.1)
    assert (fm.predict(X) == np.sign(dec)).all()
    assert_close(fm.predictProba(X), [O.lib().orc_expit(v) for v in dec], 1e-12, 1e-15)
    import ctypes as C
    ys = np.sign(y)
    acc = O.lib().orc_accuracy_sign(ys.ctypes.data_as(C.c_void_p), dec.ctypes.data_as(C.c_void_p), C.c_int64(n))
    assert abs(fm.score(X, y) - acc) < 1e-15


def test_ffm():
    for n, d, F, k in [(80, 20, 5, 4), (120, 48, 16, 8), (50, 30, 3, 30)]:
        Xo, Xd, field_of, _ = make_ffm_dataset(n, d, F, k, 42, threshold=0.3)
        rng = np.random.default_rng(2)
        P, w = rng.standard_normal((F, d, k)) * 0.3, rng.standard_normal(d)
        ffm = gpu_ffm("regression", k, True, True, P, w, -0.5)
        got = ffm.decisionFunction(to_gpu(Xo))
        assert_close(got, O.ffm_decision_function(Xo, P, w, -0.5), RTOL, ATOL)
        assert_close(got, O.slow_ffm_decision_function(Xd, field_of, F, P, w, -0.5), 1e-6, 1e-9)


def test_larger_random_csr():
    from common import random_csr
    n, d, m, k = 20000, 5000, 32, 16
    Xo = random_csr(n, d, m, seed=42)
    rng = np.random.default_rng(1)
    P, w = rng.standard_normal((1, k, d)) * 0.1, rng.standard_normal(d) * 0.1
    fm = gpu_fm("regression", 2, k, "explicit", True, True, P, w, 0.3)
    assert_close(fm.decisionFunction(to_gpu(Xo)), O.fm_decision_function(Xo, 2, P, w, 0.3), RTOL, ATOL)


def test_errors():
    import nimfm_amd as nf
    Xo, _, _ = make_fm_dataset(10, 8, 2, 4, 1)
    X = to_gpu(Xo)
    fm = nf.newFactorizationMachine("regression", nComponents=4)
    with pytest.raises(nf.NotFittedError):  # model/fm_base.nim:13-15
        fm.decisionFunction(X)
    P, w, b, _ = init_fm(9, 2, 4, "explicit", True)
    fm = gpu_fm("regression", 2, 4, "explicit", True, True, P, w, b)
    with pytest.raises(ValueError, match="Invalid nFeatures"):  # model/factorization_machine.nim:114-115
        fm.decisionFunction(X)
    with pytest.raises(ValueError):  # :65-69
        nf.newFactorizationMachine("regression", degree=0)
    with pytest.raises(ValueError):
        nf.newFactorizationMachine("regression", nComponents=0)
