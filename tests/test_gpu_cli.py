"""-m gpu: `python -m nimfm_amd train / test` (nimfm.nim:72-134) end to end in a fresh interpreter: svmlight
files parsed on the GPU, SGD / AdaGrad fit, score reduced on the device, dump / load, --predict file; the
numbers it prints equal what the same calls give in-process."""
import os
import subprocess
import sys

import numpy as np
import pytest

import nimfm_amd as nf
from oracle import ingest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*argv):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nimfm_amd", *argv], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def make_files(tmp_path, task):
    rng = np.random.default_rng(21)
    n, d = 600, 40
    dense = rng.uniform(-1, 1, size=(n, d)) * (rng.random((n, d)) < 0.25)
    dense[0, 0], dense[1, d - 1] = 0.5, -0.5
    wt = rng.standard_normal(d)
    y = dense @ wt + 0.1 * rng.standard_normal(n)
    if task == "c":
        y = np.sign(y)
    rows, cols = np.nonzero(dense)
    indptr = np.concatenate([[0], np.cumsum((dense != 0).sum(1))])
    half = n // 2
    files = []
    for lo, hi in ((0, half), (half, n)):
        ip = indptr[lo:hi + 1] - indptr[lo]
        sl = slice(indptr[lo], indptr[hi])
        p = tmp_path / ("part%d.svm" % lo)
        p.write_text(ingest.dump_svmlight(ip, cols[sl], dense[rows[sl], cols[sl]], y[lo:hi]))
        files.append(str(p))
    return files, d


@pytest.mark.parametrize("task,solver", [("r", "sgd"), ("c", "adagrad")])
def test_train_dump_test(tmp_path, task, solver):
    (train, test), d = make_files(tmp_path, task)
    model, pred = str(tmp_path / "model.txt"), str(tmp_path / "pred.txt")
    loss = "squared" if task == "r" else "logistic"
    out = run("train", "--task", task, "--train", train, "--test", test, "--solver", solver, "--n-components", "4",
              "--maxIter", "5", "--eta0", "0.05", "--loss", loss, "--shuffle", "false", "--dump", model,
              "--nFeatures", str(d), "--verbose", "1", "--predict", pred)
    assert "Number of samples  : 300" in out and "Number of features : %d" % d in out
    key = "Test RMSE: " if task == "r" else "Test Accuracy: "
    cli_score = float([ln for ln in out.splitlines() if ln.startswith(key)][0][len(key):])
    # the same calls in-process
    X, y = nf.loadSVMLightFile(train, d)
    fm = nf.newFactorizationMachine("regression" if task == "r" else "classification", nComponents=4, scale=0.1)
    common = dict(maxIter=5, eta0=0.05, alpha0=1e-7, alpha=1e-5, beta=1e-3, loss=loss, verbose=0, tol=1e-5, shuffle=False,
                  lossParam=0.1)
    (nf.newSGD(**common) if solver == "sgd" else nf.newAdaGrad(**common)).fit(X, y, fm)
    Xt, yt = nf.loadSVMLightFile(test, d)
    assert fm.score(Xt, yt) == cli_score
    got = np.array([float(v) for v in open(pred).read().split()])
    assert np.array_equal(got, fm.decisionFunction(Xt))
    # the dumped model reloads to the same parameters, and `test` reproduces the score from it
    g = nf.load(model, False)
    assert np.array_equal(g.P, fm.P) and np.array_equal(g.w, fm.w) and g.intercept == fm.intercept
    out2 = run("test", "--task", task, "--test", test, "--load", model, "--nFeatures", str(d), "--verbose", "0")
    assert float(out2.strip().split(": ")[1]) == cli_score


def test_train_mbpsgd(tmp_path):
    """--solver mbpsgd (the reference's nimfm_sparsefm CLI offers it, src/nimfm_sparsefm.nim:58-63): ingest -> fit -> dump"""
    (train, test), d = make_files(tmp_path, "r")
    model = str(tmp_path / "sparse.txt")
    out = run("train", "--task", "r", "--train", train, "--test", test, "--solver", "mbpsgd", "--n-components", "4",
              "--maxIter", "5", "--eta0", "0.05", "--shuffle", "false", "--dump", model, "--nFeatures", str(d), "--verbose", "1",
              "--reg", "l1", "--gamma", "1e-3", "--miniBatchSize", "50", "--tol", "-1")
    assert "Minibatch size: 50" in out and "Number of inner iteration: 6" in out
    cli_score = float([ln for ln in out.splitlines() if ln.startswith("Test RMSE: ")][0][len("Test RMSE: "):])
    X, y = nf.loadSVMLightFile(train, d)
    fm = nf.newFactorizationMachine("regression", nComponents=4, scale=0.1)
    nf.newMBPSGD(maxIter=5, eta0=0.05, alpha0=1e-7, alpha=1e-5, beta=1e-3, gamma=1e-3, reg=nf.newL1(), miniBatchSize=50,
                 verbose=0, tol=-1.0, shuffle=False, lossParam=0.1).fit(X, y, fm)
    Xt, yt = nf.loadSVMLightFile(test, d)
    assert fm.score(Xt, yt) == cli_score
    g = nf.load(model, False)
    assert np.array_equal(g.P, fm.P) and np.array_equal(g.w, fm.w) and g.intercept == fm.intercept


def test_unsupported_solver(tmp_path):
    (train, _), d = make_files(tmp_path, "r")
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "nimfm_amd", "train", "--task", "r", "--train", train, "--solver", "cd"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "not supported" in r.stderr
