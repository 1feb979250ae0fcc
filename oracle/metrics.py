"""TEST INFRASTRUCTURE ONLY -- restatement of /root/reference/src/nimfm/metrics.nim: rmse (:5-13), accuracy
(:39-47), rocauc (:76-103), and of `score` (model/fm_base.nim:39-48).  Sequential, in the reference's
summation order.  Parity unpinned against reference-run outputs (no Nim toolchain); the reference's own
tests (tests/test_metrics.nim) compare with hand-computed values, repeated in tests/test_oracle_metrics.py."""
import math

import numpy as np


def rmse(y_true, y_score):
    if len(y_true) != len(y_score):
        raise ValueError("len(yScore)=%d, but len(yTrue)=%d" % (len(y_score), len(y_true)))
    acc = 0.0
    for a, b in zip(y_score, y_true):
        acc += (a - b) ** 2
    return math.sqrt(acc / float(len(y_true)))


def accuracy(y_true, y_pred):
    if len(y_true) != len(y_pred):
        raise ValueError("len(yPred)=%d, but len(yTrue)=%d" % (len(y_pred), len(y_true)))
    acc = 0.0
    for a, b in zip(y_pred, y_true):
        acc += float(a == b)
    return acc / float(len(y_pred))


def rocauc(y_true, y_score, pos=1):
    order = np.argsort(-np.asarray(y_score, dtype=np.float64), kind="stable")
    result = 0.0
    fp = tp = fp_prev = tp_prev = 0
    score_prev = -math.inf
    n_pos = n_neg = 0
    for i in order:
        if y_score[i] != score_prev:
            result += float((fp - fp_prev) * (tp + tp_prev)) / 2.0
            score_prev = y_score[i]
            fp_prev, tp_prev = fp, tp
        if y_true[i] == pos:
            n_pos += 1
            tp += 1
        else:
            n_neg += 1
            fp += 1
    result += float((fp - fp_prev) * (tp + tp_prev)) / 2.0
    return result / float(n_neg * n_pos) if n_neg * n_pos else float("nan")


def score(task, y, y_pred):
    if task == "regression":
        return rmse(y, y_pred)
    return accuracy(np.sign(y).astype(np.int64), np.sign(y_pred).astype(np.int64))
