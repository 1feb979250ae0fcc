"""CPU restatement of the data-parallel exchange rule of csrc/dp.hip -- TEST INFRASTRUCTURE.

A rank is a generator: it trains its shard with the CPU oracle of the mini-batch rule (oracle/nimfm_mb.c), yields
("sum" | "max", array) whenever the library issues a collective and is sent the reduced array back.  `simulate` drives
the ranks of a group in lockstep inside one process; tests/test_dp_gloo.py drives ONE rank per process with gloo
all-reduces.  The GPU tests hold the library (groups made by nfm_dp_create_local, one host thread per rank) to it.

Rule (true-value space; the library works on stored values = true / lazy-L2 scale, which is why its late fold-in needs
no explicit decay factor):
  sync points after mini-batches S, 2S, ... that are regular on every rank and lie before every rank's last batch
  SGD      at sync k a rank snapshots its parameters; the group's mean minus the snapshot is folded in at sync k+1
           (overlap: the collective runs beside period k+1), decayed by the L2 factors of that period; without overlap
           it is folded in at once.  Closing exchange: plain mean.
  AdaGrad  own = state - base is summed over the ranks; the others' share (sum - own) is folded in at sync k+1 (or at
           once); base tracks the agreed state.  Closing exchange: state = base + sum of the remaining increments.
"""
import numpy as np

import oracle as O


def batch_bounds(n, B, first_singleton):
    pos = [0]
    if first_singleton and n > 0:
        pos.append(1)
    while pos[-1] < n:
        pos.append(min(n, pos[-1] + B))
    return pos


def n_sync_mine(bounds, B, S):
    nb = len(bounds) - 1
    if S <= 0 or nb == 0:
        return 0
    regular = nb
    if bounds[-1] - bounds[-2] < B and not (nb == 1 and bounds[1] == 1):
        regular = nb - 1
    mine = regular // S
    if mine * S >= nb:
        mine = (nb - 1) // S
    return max(mine, 0)


def decay(cfg, reg, it_lo, it_hi):
    """prod_{it_lo <= t < it_hi} (1 - eta_t(reg) * reg): what the lazy L2 scale advances by over those steps"""
    d = 1.0
    for t in range(it_lo, it_hi):
        d *= 1.0 - O.lib().orc_get_eta(cfg.scheduling, cfg.eta0, cfg.power, reg, t) * reg
    return d


def rank_sgd(epoch_fn, P, w, b, cfg, n, B, S, it0, overlap, world):
    """epoch_fn(P, w, b, begin, end, it) -> (b, loss, viol): the oracle's mini-batch epoch over [begin, end) in place"""
    bounds = batch_bounds(n, B, False)
    nb = len(bounds) - 1
    n_sync = int(-(yield ("max", np.array([-float(n_sync_mine(bounds, B, S))])))[0])
    loss = viol = 0.0
    pending = None

    def run(b0, b1):
        nonlocal b, loss, viol
        if b1 > b0:
            b, ls, vs = epoch_fn(P, w, b, bounds[b0], bounds[b1], it0 + bounds[b0])
            loss += ls
            viol += vs

    def fold(b0, b1):
        nonlocal b, pending
        if pending is None:
            return
        dP, dw, db = pending
        P[...] += dP * decay(cfg, cfg.beta, it0 + bounds[b0], it0 + bounds[b1])
        if cfg.fit_linear:
            w[...] += dw * decay(cfg, cfg.alpha, it0 + bounds[b0], it0 + bounds[b1])
        else:
            w[...] += dw
        b += db  # the intercept is stored as a true value: no lazy scale on it
        pending = None

    for k in range(1, n_sync + 1):
        run((k - 1) * S, k * S)
        fold((k - 1) * S, k * S)
        snap = np.concatenate([P.ravel(), w, [b]])
        mean = (yield ("sum", snap.copy())) / world
        d = mean - snap
        pending = (d[:P.size].reshape(P.shape), d[P.size:P.size + w.size], d[-1])
        if not overlap:
            P[...] += pending[0]
            w[...] += pending[1]
            b += pending[2]
            pending = None
    run(n_sync * S, nb)
    fold(n_sync * S, nb)
    flat = (yield ("sum", np.concatenate([P.ravel(), w, [b]]))) / world
    P[...] = flat[:P.size].reshape(P.shape)
    w[...] = flat[P.size:P.size + w.size]
    b = flat[-1]
    sums = yield ("sum", np.array([loss, viol, float(n)]))
    return P, w, b, sums[0], sums[1], it0 + int(round(sums[2]))


def _ada_flat(st):
    return np.concatenate([st.gsum_P.ravel(), st.gnorm_P.ravel(), st.gsum_w, st.gnorm_w, [st.gsum_b.value, st.gnorm_b.value]])


def _ada_unflat(st, f):
    a = st.gsum_P.size
    st.gsum_P[...] = f[:a].reshape(st.gsum_P.shape)
    st.gnorm_P[...] = f[a:2 * a].reshape(st.gnorm_P.shape)
    d = st.gsum_w.size
    st.gsum_w[...] = f[2 * a:2 * a + d]
    st.gnorm_w[...] = f[2 * a + d:2 * a + 2 * d]
    st.gsum_b.value, st.gnorm_b.value = f[-2], f[-1]


def rank_adagrad(epoch_fn, st, n, B, S, it0, overlap, world):
    """epoch_fn(begin, end, it) -> (loss, viol): the oracle's AdaGrad mini-batch epoch over [begin, end), updating st"""
    bounds = batch_bounds(n, B, it0 == 1)
    nb = len(bounds) - 1
    n_sync = int(-(yield ("max", np.array([-float(n_sync_mine(bounds, B, S))])))[0])
    loss = viol = 0.0
    base = _ada_flat(st)
    pending = None

    def run(b0, b1):
        nonlocal loss, viol
        if b1 > b0:
            ls, vs = epoch_fn(bounds[b0], bounds[b1], it0 + bounds[b0])
            loss += ls
            viol += vs

    def fold():
        nonlocal pending, base
        if pending is None:
            return
        total, own = pending
        _ada_unflat(st, _ada_flat(st) + (total - own))
        base = base + total
        pending = None

    for k in range(1, n_sync + 1):
        run((k - 1) * S, k * S)
        fold()
        own = _ada_flat(st) - base
        total = yield ("sum", own.copy())
        pending = (total, own)
        if not overlap:
            fold()
    run(n_sync * S, nb)
    fold()
    own = _ada_flat(st) - base
    total = yield ("sum", own.copy())
    _ada_unflat(st, base + total)
    sums = yield ("sum", np.array([loss, viol, float(n)]))
    return st, sums[0], sums[1], it0 + int(round(sums[2]))


def simulate(gens):
    """drive the ranks of a group in lockstep; returns what every generator returned"""
    world = len(gens)
    results = [None] * world
    reqs = [next(g) for g in gens]
    live = list(range(world))
    while live:
        op = reqs[live[0]][0]
        assert all(reqs[r][0] == op for r in live) and len(live) == world, "ranks issue different collectives"
        stack = np.stack([reqs[r][1] for r in live])
        red = stack.max(0) if op == "max" else np.add.reduce(stack, 0)  # rank order, like the library's local transport
        nxt = []
        for r in live:
            try:
                reqs[r] = gens[r].send(red.copy())
                nxt.append(r)
            except StopIteration as e:
                results[r] = e.value
        live = nxt
    return results


def drive_with_dist(gen, dist, torch):
    """one rank per process: every yielded collective is a torch.distributed all-reduce"""
    req = next(gen)
    while True:
        t = torch.from_numpy(np.ascontiguousarray(req[1]).copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX if req[0] == "max" else dist.ReduceOp.SUM)
        try:
            req = gen.send(t.numpy())
        except StopIteration as e:
            return e.value
