"""-m gpu: nfm_score / nfm_metrics (metrics.hip) against oracle/metrics.py: the reference's own
known-answer tests (tests/test_metrics.nim:5-46) pushed through the device path, random scores with ties,
and `score` after a fit."""
import numpy as np
import pytest

import nimfm_amd as nf
from oracle import metrics as M
from test_oracle_metrics import ROCAUC_KAT

pytestmark = pytest.mark.gpu


def model_with_scores(scores, task):
    """a linear-only model on the identity dataset: decisionFunction == scores exactly"""
    n = len(scores)
    X = nf.newCSRDataset(data=np.ones(n), indices=np.arange(n), indptr=np.arange(n + 1), nSamples=n, nFeatures=n)
    fm = nf.newFactorizationMachine(task, nComponents=1, warmStart=True)
    fm.set_params(np.zeros((1, 1, n)), np.asarray(scores, dtype=np.float64), 0.0)
    assert np.array_equal(fm.decisionFunction(X), np.asarray(scores, dtype=np.float64))
    return fm, X


def test_reference_known_answers_on_device():
    for yt, ys, want in ROCAUC_KAT:
        fm, X = model_with_scores(ys, "classification")
        # rocauc's positive class is yTrue == 1; the device path sees targets through sgn (pos <-> y > 0)
        got = fm.metrics(X, [1.0 if v == 1 else -1.0 for v in yt])["rocauc"]
        assert got == want, (yt, ys, got)
    fm, X = model_with_scores([-0.1, 0.1, 0.9, -0.2], "classification")
    assert fm.score(X, [-1.0, -1.0, 1.0, 1.0]) == 0.5  # accuracy(yTrue, sgn(yScore2)), test_metrics.nim:41


@pytest.mark.parametrize("n,ties", [(1000, False), (50_000, True), (300_001, True)])
def test_random_scores(n, ties):
    rng = np.random.default_rng(n)
    s = rng.standard_normal(n)
    if ties:
        s = np.round(s, 2)  # many equal scores: the trapezoid groups
    y = np.sign(s + rng.standard_normal(n))
    y[y == 0] = 1.0
    fm, X = model_with_scores(s, "classification")
    got = fm.metrics(X, y)
    assert got["accuracy"] == M.accuracy(np.sign(y).astype(np.int64), np.sign(s).astype(np.int64))
    assert got["rocauc"] == M.rocauc(y.astype(np.int64), s)  # integer trapezoid sums: bit for bit
    assert abs(got["rmse"] - M.rmse(y, s)) <= 1e-11 * M.rmse(y, s)  # the reference adds left to right, the device by a tree
    assert fm.score(X, y) == got["accuracy"]
    fm.task = "regression"
    fm2, _ = model_with_scores(s, "regression")
    assert abs(fm2.score(X, y) - M.rmse(y, s)) <= 1e-11 * M.rmse(y, s)  # the reference adds left to right, the device by a tree
    assert fm2.metrics(X, y) == fm2.metrics(X, y)  # reproducible


def test_degenerate_and_errors():
    fm, X = model_with_scores([0.3, 0.2, 0.1], "classification")
    assert np.isnan(fm.metrics(X, [1.0, 1.0, 1.0])["rocauc"])  # one class only: 0/0 as in the reference
    fm = nf.newFactorizationMachine("regression", nComponents=2)
    with pytest.raises(nf.NotFittedError):
        fm.score(X, [0.0, 0.0, 0.0])
