"""CPU: oracle/metrics.py (restatement of metrics.nim) against the known-answer tests the reference
holds for it (/root/reference/tests/test_metrics.nim:5-46) -- these ARE golden vectors of the reference."""
import numpy as np

from oracle import metrics as M

Y_TRUE = [-1, -1, 1, 1]
Y_TRUE01 = [0, 0, 1, 1]
SCORE1 = [0.1, 0.4, 0.35, 0.8]
SCORE2 = [-0.1, 0.1, 0.9, -0.2]
ZEROS, ONES = [0.0] * 4, [1.0] * 4
INVERSE, INVERSE01 = [1, 1, -1, -1], [1, 1, 0, 0]
# (yTrue, yScore, expected) -- test_metrics.nim:21-36
ROCAUC_KAT = [(yt, ys, want) for yt in (Y_TRUE, Y_TRUE01) for ys, want in
              [(SCORE1, 0.75), (SCORE2, 0.5), ([float(v) for v in Y_TRUE], 1.0), (ZEROS, 0.5), (ONES, 0.5),
               ([float(v) for v in INVERSE], 0.0), ([float(v) for v in INVERSE01], 0.0)]]


def test_rocauc_known_answers():
    for yt, ys, want in ROCAUC_KAT:
        assert M.rocauc(yt, ys) == want, (yt, ys)


def test_accuracy_known_answers():  # test_metrics.nim:39-50
    sgn = lambda v: int(np.sign(v))
    assert M.accuracy(Y_TRUE, [sgn(x - 0.5) for x in SCORE1]) == 0.75
    assert M.accuracy(Y_TRUE, [sgn(x) for x in SCORE2]) == 0.5
    assert M.accuracy(Y_TRUE, Y_TRUE) == 1.0
    assert M.accuracy(Y_TRUE, [0] * 4) == 0
    assert M.accuracy(Y_TRUE, [1] * 4) == 0.5
    assert M.accuracy(Y_TRUE01, [int((sgn(x - 0.5) + 1) / 2) for x in SCORE1]) == 0.75


def test_rmse():
    assert M.rmse([1.0, 2.0, 3.0], [1.0, 2.0, 3.0]) == 0.0
    assert M.rmse([0.0, 0.0], [3.0, 4.0]) == np.sqrt(12.5)
