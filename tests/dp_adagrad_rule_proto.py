"""(Not a test: a research script that uses the oracle as its engine, which only tests/ may -- kept here for that reason.)
CPU prototype (numpy + the oracle's mini-batch AdaGrad, no GPU): how should the ranks' AdaGrad STATE increments be combined at an
exchange?  The library's rules -- sum (over-shoots beyond a few mini-batches per exchange), mean (stable, one rank's progress
x 0.75 at 8 ranks), 1 / sqrt(world) (good for short periods, diverges a whole epoch apart) -- against candidates, on the planted
problem of tools/dp_convergence.py.  progress = (L_start - L_run) / (L_start - L_one_rank_over_all_samples).
usage: python tests/dp_adagrad_rule_proto.py [epochs] [world] [period ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", ".."))
import numpy as np  # noqa: E402

import oracle as O  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
PERIODS = [int(v) for v in sys.argv[3:]] or [4, 16, 0]
n, nt, d, m, k, B = 160_000, 20_000, 20_000, 16, 8, 512
rng = np.random.default_rng(5)
idx = np.sort(rng.integers(0, d, size=(n + nt, m)), axis=1)
for _ in range(50):
    dup = np.zeros_like(idx, dtype=bool)
    dup[:, 1:] = idx[:, 1:] == idx[:, :-1]
    if not dup.any():
        break
    idx[dup] = rng.integers(0, d, size=int(dup.sum()))
    idx.sort(axis=1)
val = rng.uniform(-1.0, 1.0, size=(n + nt, m))
Xall = O.Dataset(np.arange(n + nt + 1, dtype=np.int64) * m, idx.ravel(), val.ravel(), n + nt, d)
rng = np.random.default_rng(9)
Pp, wp = rng.standard_normal((1, k, d)) * 0.3, rng.standard_normal(d) * 0.3
yfull = O.fm_decision_function(Xall, 2, Pp, wp, 0.1) + 0.1 * rng.standard_normal(n + nt)
Xtr = O.Dataset(np.arange(n + 1, dtype=np.int64) * m, idx[:n].ravel(), val[:n].ravel(), n, d)
Xte = O.Dataset(np.arange(nt + 1, dtype=np.int64) * m, idx[n:].ravel(), val[n:].ravel(), nt, d)
ytr, yte = yfull[:n], yfull[n:]
P0, w0 = np.random.default_rng(1).standard_normal((1, k, d)) * 0.01, np.zeros(d)
cfg = O.adagrad_cfg(eta0=0.1, alpha=1e-5, beta=1e-5)


def rmse(P, w, b):
    return float(np.sqrt(np.mean((O.fm_decision_function(Xte, 2, P, w, b) - yte) ** 2)))


class St:
    """the additive AdaGrad state as flat arrays"""

    def __init__(self, st=None):
        if st is None:
            self.G, self.N = np.zeros((1, d, k)), np.full((1, d, k), 1e-10)
            self.Gw, self.Nw = np.zeros(d), np.full(d, 1e-10)
            self.gb, self.nb = 0.0, 1e-10
        else:
            self.G, self.N, self.Gw, self.Nw, self.gb, self.nb = st.G.copy(), st.N.copy(), st.Gw.copy(), st.Nw.copy(), st.gb, st.nb

    def to_oracle(self):
        a = O.AdaState(1, d, k, d)
        a.gsum_P[:], a.gnorm_P[:], a.gsum_w[:], a.gnorm_w[:] = self.G, self.N, self.Gw, self.Nw
        a.gsum_b.value, a.gnorm_b.value = self.gb, self.nb
        return a

    @staticmethod
    def of(a):
        s = St.__new__(St)
        s.G, s.N, s.Gw, s.Nw = a.gsum_P.copy(), a.gnorm_P.copy(), a.gsum_w.copy(), a.gnorm_w.copy()
        s.gb, s.nb = a.gsum_b.value, a.gnorm_b.value
        return s


def params_of(s, it):
    P, w = np.zeros((1, k, d)), np.zeros(d)
    b = O.fm_adagrad_finalize(2, P, w, 0.0, cfg, it, s.to_oracle())
    return P, w, b


def run_range(s, it, lo, hi):
    """mini-batches over samples [lo, hi) from state s -> the state afterwards"""
    P, w, b = params_of(s, it)
    a = s.to_oracle()
    O.fm_adagrad_epoch_mb(Xtr, ytr, 2, P, w, b, cfg, B, a, begin=lo, end=hi, it=max(it, 2))  # (it >= 2: no singleton first step inside a fit)
    return St.of(a)


def combine(rule, s0, outs):
    w_ = len(outs)
    res = St(s0)
    for name in ("G", "Gw"):
        nn = "N" if name == "G" else "Nw"
        base_g, base_n = getattr(s0, name), getattr(s0, nn)
        dG = [getattr(o, name) - base_g for o in outs]
        dN = [getattr(o, nn) - base_n for o in outs]
        sG, sN = sum(dG), sum(dN)
        if rule == "sum":
            g, nrm = sG, sN
        elif rule == "mean":
            g, nrm = sG / w_, sN / w_
        elif rule == "rsqrt":
            g, nrm = sG / np.sqrt(w_), sN / np.sqrt(w_)
        elif rule.startswith("cross"):  # the sum, with the squared norm taking the ranks' AGREEMENT: + gamma ((sum dG)^2 - sum dG^2)
            gamma = float(rule[5:] or 1.0)
            cross = sG * sG - sum(x * x for x in dG)
            g, nrm = sG, sN + gamma * np.maximum(cross, 0.0)
        elif rule.startswith("xmono"):  # the same without the per-coordinate clamp of the cross term (it needs sum dG_r^2 on its own: a
            # third array in the exchange); instead the squared norm never DECREASES across an exchange (it does not in AdaGrad)
            gamma = float(rule[5:] or 1.0)
            cross = sG * sG - sum(x * x for x in dG)
            g, nrm = sG, np.maximum(sN + gamma * cross, 0.0)
        elif rule.startswith("agree"):  # weight between mean and sum by the ranks' agreement |sum dG| / sum |dG| per coordinate
            p_ = float(rule[5:] or 1.0)
            a_ = np.abs(sG) / np.maximum(sum(np.abs(x) for x in dG), 1e-300)  # 1: all ranks push the same way
            # agreeing ranks repeat one another's step (stale gradients): average them; disagreeing ones carry different information: add
            wgt = 1.0 / (1.0 + (w_ - 1.0) * a_ ** p_)
            g, nrm = sG * wgt, sN * wgt
        else:
            raise ValueError(rule)
        setattr(res, name, base_g + g)
        setattr(res, nn, base_n + nrm)
    dgb = [o.gb - s0.gb for o in outs]
    dnb = [o.nb - s0.nb for o in outs]
    f = {"sum": 1.0, "mean": 1.0 / w_, "rsqrt": 1.0 / np.sqrt(w_)}.get(rule, 1.0 / w_)
    res.gb, res.nb = s0.gb + f * sum(dgb), s0.nb + f * sum(dnb)
    return res


def fit(rule, world, period):
    s, it = St(), 1
    per = n // world
    nb = per // B
    P_ = period if period else nb
    for e in range(E):
        for b0 in range(0, nb, P_):
            b1 = min(nb, b0 + P_)
            outs = []
            for r in range(world):
                lo = r * per + b0 * B
                hi = r * per + (b1 * B if b1 < nb else per)
                outs.append(run_range(s, it + b0 * B, lo, hi))
            s = combine(rule, s, outs) if world > 1 else outs[0]
        it += n
        if not np.isfinite(s.G).all():
            return float("nan")
    return rmse(*params_of(s, it))


L0 = rmse(P0 * 0.0, w0, 0.0)
one = fit("sum", 1, 0)
print("planted FM as tools/dp_convergence.py: held-out RMSE at start %.4f, one rank over all samples after %d epochs %.4f" % (L0, E, one), flush=True)
for period in PERIODS:
    row = []
    for rule in (os.environ.get("RULES") or "sum,mean,rsqrt,cross1,cross0.25,agree1,agree2").split(","):
        v = fit(rule, W, period)
        row.append("%s %.4f (%.2f)" % (rule, v, (L0 - v) / (L0 - one) if np.isfinite(v) else float("nan")))
    print("  %d ranks, exchange every %s: %s" % (W, "%d mini-batches" % period if period else "epoch (%d)" % ((n // W) // B), "; ".join(row)), flush=True)
