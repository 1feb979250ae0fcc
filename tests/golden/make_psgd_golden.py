"""Generates tests/golden/psgd_golden.npz: small input/output vectors for mini-batch proximal SGD.

PROVENANCE: the reference is Nim-only and cannot be run in this image, and it has no test for MBPSGD itself.  These
vectors come from a DENSE NUMPY STATEMENT of optimizer/minibatch_psgd.nim:87-122 for a degree-2 model -- analytic
gradient of the sum-of-squares kernel, Params.step as `(theta - eta g) * (1 / (1 + eta reg))`, and the proximal operators
in their textbook (sort-based) form, i.e. tests/regularizer/squaredl12_slow.nim's construction -- not from the code
under test (oracle/nimfm_psgd.c's randomised-pivot operator, the HIP kernels' fixed-point threshold).  They pin
regressions and cross-implementation agreement; they are NOT outputs of the reference binary.

Run from the repo root:  python tests/golden/make_psgd_golden.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
N, D, K, B, OUTER = 60, 9, 4, 16, 2
ETA0, ALPHA0, ALPHA, BETA, GAMMA = 0.2, 1e-2, 1e-2, 1e-2, 0.03


def soft(x, a):
    return np.sign(x) * np.maximum(np.abs(x) - a, 0.0)


def prox_sql12_vec(p, lam):
    """squaredl12_slow.nim:10-25"""
    a = np.sort(np.abs(p))[::-1]
    S = 2.0 * lam * np.cumsum(a) / (1.0 + 2.0 * lam * (np.arange(len(a)) + 1.0))
    theta = 0
    for i in range(len(a)):
        if a[i] - S[i] < 0:
            break
        theta += 1
    if theta == 0:
        return np.zeros_like(p)
    return np.where(np.abs(p) < a[theta - 1], 0.0, soft(p, S[theta - 1]))


def prox(reg, Pt, lam):
    """Pt: [d][k]"""
    if reg == "l1":
        return soft(Pt, lam)
    nr = np.sqrt((Pt * Pt).sum(1))
    if reg == "l21":
        f = np.where(nr > lam, 1.0 - lam / np.where(nr > 0, nr, 1.0), 0.0)
        return Pt * f[:, None]
    if reg == "squaredl12":  # column-wise (transpose = true, the default)
        return np.stack([prox_sql12_vec(Pt[:, s], lam) for s in range(Pt.shape[1])], axis=1)
    nn = prox_sql12_vec(nr, lam)  # squaredl21
    return Pt / np.where(nr > 0, nr, 1.0)[:, None] * nn[:, None]


def main():
    rng = np.random.default_rng(77)
    X = rng.uniform(-1, 1, (N, D))
    X[np.abs(X) < 0.35] = 0.0
    Ptrue = rng.normal(size=(K, D)) * 0.5
    y = 0.5 * (((X @ Ptrue.T) ** 2).sum(1) - ((X ** 2) @ (Ptrue.T ** 2)).sum(1))
    P0 = rng.normal(size=(1, K, D)) * 0.3
    w0 = rng.normal(size=D) * 0.1
    inner = (N - 1) // B + 1
    stream = np.concatenate([rng.permutation(N) for _ in range(OUTER * 2)])[:OUTER * B * inner].astype(np.int64)
    out = {"X": X, "y": y, "P0": P0, "w0": w0, "b0": np.array(0.1), "stream": stream,
           "hyper": np.array([ETA0, ALPHA0, ALPHA, BETA, GAMMA]), "batch": np.array(B), "outer": np.array(OUTER)}
    for reg in ("l1", "l21", "squaredl12", "squaredl21"):
        P, w, b, it = P0[0].copy(), w0.copy(), 0.1, 1  # P: [k][d]
        losses = []
        for t in range(OUTER):
            ls = 0.0
            for r in range(inner):
                idx = stream[(t * inner + r) * B:(t * inner + r + 1) * B]
                Xb = X[idx]
                A = Xb @ P.T  # [B][k]
                yh = b + Xb @ w + 0.5 * ((A ** 2).sum(1) - ((Xb ** 2) @ (P.T ** 2)).sum(1))
                ls += 0.5 * ((y[idx] - yh) ** 2).sum()
                coef = (yh - y[idx]) / B
                gP = (coef[:, None] * A).T @ Xb - P * ((coef[:, None] * Xb ** 2).sum(0))[None, :]
                gw, gb = Xb.T @ coef, coef.sum()
                eta = lambda reg_: ETA0 / (1.0 + ETA0 * reg_ * it)  # optimal schedule, power 1
                eP, ew, e0 = eta(BETA), eta(ALPHA), eta(ALPHA0)
                P = (P - eP * gP) * (1.0 / (1.0 + eP * BETA))
                w = (w - ew * gw) * (1.0 / (1.0 + ew * ALPHA))
                b = (b - e0 * gb) * (1.0 / (1.0 + e0 * ALPHA0))
                P = prox(reg, P.T, GAMMA * eP / (1.0 + eP * BETA)).T
                it += 1
            losses.append(ls / (B * inner))
        out[reg + "_P"], out[reg + "_w"], out[reg + "_b"], out[reg + "_loss"] = P[None], w, np.array(b), np.array(losses)
    np.savez_compressed(os.path.join(HERE, "psgd_golden.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
