"""CPU restatement of the data-parallel exchange rule of csrc/dp.hip -- TEST INFRASTRUCTURE.

A rank is a generator: it trains its shard with the CPU oracle of the mini-batch rule (oracle/nimfm_mb.c), yields
("sum" | "max", array) whenever the library issues a collective and is sent the reduced array back.  `simulate` drives
the ranks of a group in lockstep inside one process; tests/test_dp_gloo.py drives ONE rank per process with gloo
all-reduces.  The GPU tests hold the library (groups made by nfm_dp_create_local, one host thread per rank) to it.

Rule (true-value space; the library works on stored values = true / lazy-L2 scale, in which an untouched value does not
change while its true value decays -- the `dec` factors below are that decay):
  sync points after mini-batches S, 2S, ... that are regular on every rank and lie before every rank's last batch
  Both optimizers exchange INCREMENTS since the last agreed state (`base`) and add the ranks' increments up.
  SGD      own = parameters - base (base decayed to now); the ranks' increments are combined to total = cw * sum with
           cw = 1 / world (mean, the default) or 1 (sum); the others' share (total - own) is
           folded in at sync k+1 (overlap: the collective runs beside period k+1), decayed by that period's factors, or at
           once; base tracks the agreed state.  Closing exchange: increments in true values, the base under the smallest
           decay factor of the ranks' (possibly unequal) tails.
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


def rank_sgd(epoch_fn, P, w, b, cfg, n, B, S, it0, overlap, world, combine="mean"):
    """epoch_fn(P, w, b, begin, end, it) -> (b, loss, viol): the oracle's mini-batch epoch over [begin, end) in place"""
    bounds = batch_bounds(n, B, False)
    nb = len(bounds) - 1
    n_sync = int(-(yield ("max", np.array([-float(n_sync_mine(bounds, B, S))])))[0])
    loss = viol = 0.0
    pending = None
    cw = 1.0 if combine == "sum" else 1.0 / world  # the ranks' increments are averaged (default) or added up

    def flat():
        return np.concatenate([P.ravel(), w, [b]])

    def dec(it_lo, it_hi):  # what an untouched value shrinks by over those steps (the intercept has no lazy scale)
        dP = decay(cfg, cfg.beta, it_lo, it_hi)
        dw = decay(cfg, cfg.alpha, it_lo, it_hi) if cfg.fit_linear else 1.0
        return np.concatenate([np.full(P.size, dP), np.full(w.size, dw), [1.0]])

    def put(f):
        nonlocal b
        P[...] = f[:P.size].reshape(P.shape)
        w[...] = f[P.size:P.size + w.size]
        b = f[-1]

    def run(b0, b1):
        nonlocal b, loss, viol
        if b1 > b0:
            b, ls, vs = epoch_fn(P, w, b, bounds[b0], bounds[b1], it0 + bounds[b0])
            loss += ls
            viol += vs

    def fold(now_it):
        nonlocal pending
        if pending is None:
            return
        others, then_it = pending
        put(flat() + others * dec(then_it, now_it))
        pending = None

    base, base_it = flat(), it0 + bounds[0]
    for k in range(1, n_sync + 1):
        run((k - 1) * S, k * S)
        now_it = it0 + bounds[k * S]
        fold(now_it)
        base_now = base * dec(base_it, now_it)
        own = flat() - base_now
        total = cw * (yield ("sum", own.copy()))
        base, base_it = base_now + total, now_it  # what all ranks agree on at this point
        pending = (total - own, now_it)           # the other ranks' steps: folded in one period later (overlap) or at once
        if not overlap:
            fold(now_it)
    run(n_sync * S, nb)
    end_it = it0 + bounds[nb]
    fold(end_it)
    # closing exchange: the ranks' tails may differ in length; increments in true values, the base under the smallest
    # of the ranks' decay factors since the last agreed point
    d_mine = dec(base_it, end_it)
    own = flat() - base * d_mine
    d_min = -(yield ("max", -np.array([d_mine[0], d_mine[P.size] if w.size else 1.0])))
    total = cw * (yield ("sum", own.copy()))
    d_vec = np.concatenate([np.full(P.size, d_min[0]), np.full(w.size, d_min[1]), [1.0]])
    put(base * d_vec + total)
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


def rank_adagrad(epoch_fn, st, n, B, S, it0, overlap, world, w=1.0, cross_gamma=None):
    """epoch_fn(begin, end, it) -> (loss, viol): the oracle's AdaGrad mini-batch epoch over [begin, end), updating st.
    cross_gamma (NFM_DP_STATE_CROSS, csrc/dp.h): the g_sum increments are summed and the squared norm takes the ranks' agreement,
    g_norm += sum_r dN_r + gamma ((sum_r dG_r)^2 - sum_r dG_r^2), never less than before; a rank SENDS dN_r - gamma dG_r^2"""
    bounds = batch_bounds(n, B, it0 == 1)
    nb = len(bounds) - 1
    n_sync = int(-(yield ("max", np.array([-float(n_sync_mine(bounds, B, S))])))[0])
    loss = viol = 0.0
    base = _ada_flat(st)
    pending = None
    a_, d_ = st.gsum_P.size, st.gsum_w.size
    pairs = [(slice(0, a_), slice(a_, 2 * a_)), (slice(2 * a_, 2 * a_ + d_), slice(2 * a_ + d_, 2 * a_ + 2 * d_)),
             (slice(2 * a_ + 2 * d_, 2 * a_ + 2 * d_ + 1), slice(2 * a_ + 2 * d_ + 1, 2 * a_ + 2 * d_ + 2))]

    def sent_of(own):  # what travels
        if cross_gamma is None:
            return own.copy()
        out = own.copy()
        for g_, n_ in pairs:
            out[n_] = own[n_] - cross_gamma * (own[g_] * own[g_])
        return out

    def combined_of(total):  # what all ranks agree on
        if cross_gamma is None:
            return w * total
        out = total.copy()
        for g_, n_ in pairs:
            out[n_] = np.maximum(total[n_] + cross_gamma * (total[g_] * total[g_]), 0.0)
        return out

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
        comb = combined_of(total)
        _ada_unflat(st, _ada_flat(st) + (comb - own))
        base = base + comb
        pending = None

    for k in range(1, n_sync + 1):
        run((k - 1) * S, k * S)
        fold()
        own = _ada_flat(st) - base
        total = yield ("sum", sent_of(own))
        pending = (total, own)
        if not overlap:
            fold()
    run(n_sync * S, nb)
    fold()
    own = _ada_flat(st) - base
    total = yield ("sum", sent_of(own))
    _ada_unflat(st, base + combined_of(total))
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
