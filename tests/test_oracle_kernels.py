"""Oracle pin 1: ANOVA kernel, fast restatement vs brute force.

Re-runs the reference's own grid (tests/test_kernels.nim:26-46: n=20, d=10, k=10,
degree 2..5, 0..3 dummy features, |diff| < 1e-6) on the two restatements, plus
loss known-answers derived by hand from loss.nim.
"""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("m", [0, 1, 2, 3])
@pytest.mark.parametrize("degree", [2, 3, 4, 5])
def test_anova_csr_vs_bruteforce(m, degree):
    n, d, k = 20, 10, 10
    rng = np.random.default_rng(42)
    Xd = rng.uniform(0.0, 1.0, size=(n, d))
    P = rng.standard_normal((1, k, d + m))
    X = O.Dataset.from_dense(Xd)
    for s in range(k):
        lams = np.zeros(k)
        lams[s] = 1.0  # isolate component s, as anova(X, P, A, degree, s) does
        got = O.fm_decision_function(X, degree, P, np.zeros(d), 0.0, n_aug=m, lams=lams)
        for i in range(n):
            expect = O.slow_anova(Xd[i], P[0, s], m, degree)
            assert abs(got[i] - expect) < 1e-6


def test_loss_known_answers():
    L = O.lib()
    # loss.nim:18,21
    assert L.orc_loss(0, 0.0, 1.0, 3.0) == 2.0 and L.orc_dloss(0, 0.0, 1.0, 3.0) == 2.0
    # loss.nim:33-39: z = 1 - p*y
    assert L.orc_loss(1, 0.0, 1.0, 0.5) == 0.25 and L.orc_dloss(1, 0.0, 1.0, 0.5) == -1.0
    assert L.orc_loss(1, 0.0, 1.0, 2.0) == 0.0 and L.orc_dloss(1, 0.0, 1.0, 2.0) == 0.0
    # loss.nim:54-67 at z = 0: ln 2, -y/2
    assert abs(L.orc_loss(2, 0.0, 1.0, 0.0) - np.log(2.0)) < 1e-15
    assert L.orc_dloss(2, 0.0, -1.0, 0.0) == 0.5
    # both branches agree with the closed form
    for p, y in [(3.0, 1.0), (-3.0, 1.0), (40.0, -1.0), (-40.0, -1.0)]:
        assert abs(L.orc_loss(2, 0.0, y, p) - np.log1p(np.exp(-p * y))) < 1e-12
        assert abs(L.orc_dloss(2, 0.0, y, p) - (-y / (1.0 + np.exp(p * y)))) < 1e-12
    # loss.nim:84-93 incl. the sign quirk (dloss = y - p, +threshold outside)
    assert L.orc_loss(3, 1.0, 0.0, 0.5) == 0.125 and L.orc_dloss(3, 1.0, 0.0, 0.5) == -0.5
    assert L.orc_loss(3, 1.0, 0.0, 3.0) == 2.5 and L.orc_dloss(3, 1.0, 0.0, 3.0) == 1.0
    assert L.orc_dloss(3, 1.0, 0.0, -3.0) == 1.0


def test_eta_schedules():
    L = O.lib()
    # optimizer/sgd.nim:60-69
    assert L.orc_get_eta(0, 0.01, 1.0, 1e-3, 77) == 0.01
    assert abs(L.orc_get_eta(1, 0.01, 1.0, 1e-3, 100) - 0.01 / (1 + 0.01 * 1e-3 * 100)) < 1e-18
    assert abs(L.orc_get_eta(2, 0.01, 0.5, 1e-3, 100) - 0.001) < 1e-18
    assert abs(L.orc_get_eta(3, 0.01, 1.0, 1e-3, 100) - 10.0) < 1e-12


def test_orders_augments():
    # model/factorization_machine.nim:81-97
    assert O.n_orders(1, "explicit") == 0 and O.n_orders(4, "explicit") == 3
    assert O.n_orders(4, "none") == 1 and O.n_orders(4, "augment") == 1
    assert O.n_augments(4, "augment", True) == 2 and O.n_augments(4, "augment", False) == 3
    assert O.n_augments(4, "explicit", True) == 0


def test_metrics_known_answers():
    # tests/test_metrics.nim known answers for rmse / accuracy
    L = O.lib()
    import ctypes as C
    yt = np.array([1.0, 2.0, 3.0]); ys = np.array([1.0, 2.0, 5.0])
    r = L.orc_rmse(yt.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), C.c_int64(3))
    assert abs(r - np.sqrt(4.0 / 3.0)) < 1e-15
    yt = np.array([1.0, -1.0, 1.0, -1.0]); ys = np.array([0.3, 0.2, 2.0, -5.0])
    a = L.orc_accuracy_sign(yt.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), C.c_int64(4))
    assert a == 0.75
    assert abs(L.orc_expit(0.0) - 0.5) < 1e-16 and abs(L.orc_expit(-800.0)) < 1e-300
