"""-m gpu: the AdaGrad mini-batch rule with the batch's gradient cross products in g_norm (nfm_opt_set_ada_cross, round 5):
a coordinate's squared norm grows by  sum_i g_i^2 + gamma max((sum_i g_i)^2 - sum_i g_i^2, 0)  over the samples of a batch that touch
it (include/nimfm_hip.h; the AdaGrad counterpart of SGD's touch cap).  HIP against the restatement of the rule (oracle/nimfm_mb.c:
orc_mb_ada_cross) on every path a coordinate's update takes -- the column phase of one order, several orders / degree 3, wide
models (the two-blocks-in-one-walk variant), heavy features (segment partial sums), field-aware units, the linear weights and the
intercept -- and what must NOT change: coordinates touched once, batch == 1."""
import numpy as np
import pytest

import nimfm_amd as nf
import oracle as O
from common import assert_close, init_ffm, make_perms, random_csr
from gpu_common import gpu_ffm, gpu_fm, to_gpu

pytestmark = pytest.mark.gpu
G = 0.1


def _fm_case(n, d, m, k, degree, B, seed, gamma=G, epochs=2, hot=0):
    Xo = random_csr(n, d, m, seed=seed)
    rng = np.random.default_rng(seed + 1)
    if hot:  # a few features that most samples of a batch have: the heavy-feature path (more than 128 touches per batch)
        idx = Xo.indices.reshape(n, m).copy()
        idx[:, 0] = rng.integers(0, hot, size=n)
        for i in range(n):  # keep ids distinct inside a row
            while len(set(idx[i])) < m:
                idx[i, 1:] = rng.choice(np.arange(hot, d), size=m - 1, replace=False)
        Xo = O.Dataset(Xo.indptr, idx.ravel(), Xo.data, n, d)
    y = rng.standard_normal(n)
    no = degree - 1
    P0, w0 = rng.standard_normal((no, k, d)) * 0.05, rng.standard_normal(d) * 0.01
    perms = make_perms(n, epochs)
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), 0.1, 1
    st = O.AdaState(no, d, k, d)
    hv = []
    for e in range(epochs):
        b, it, ls, vs = O.fm_adagrad_epoch_mb(Xo, y, degree, P, w, b, cfg, B, st, perm=perms[e], it=it, ada_cross=gamma)
        hv.append(vs)
    b = O.fm_adagrad_finalize(degree, P, w, b, cfg, it, st)
    fm = gpu_fm("regression", degree, k, "explicit", True, True, P0, w0, 0.1)
    ada = nf.newAdaGrad(maxIter=epochs, verbose=0, tol=0, mode="minibatch", batch=B, adaCross=gamma)
    ada.fit(to_gpu(Xo), y, fm, perms=perms)
    assert abs(fm.intercept - b) < 1e-10
    assert_close(fm.w, w, 1e-9, 1e-13, "w")
    assert_close(fm.P, P, 1e-9, 1e-13, "P")
    assert_close([h[0] for h in ada.history], hv, 1e-9, 1e-12, "viol")
    gs, gn, gsw, gnw, gsb, gnb = ada.get_state(fm)
    assert_close(gn, st.gnorm_P, 1e-9, 1e-13, "g_norm.P")
    assert_close(gnw, st.gnorm_w, 1e-9, 1e-13, "g_norm.w")
    assert abs(gnb - st.gnorm_b.value) <= 1e-9 * abs(st.gnorm_b.value)
    return (P, w, b), (Xo, y, P0, w0, perms)


@pytest.mark.parametrize("d,k,degree", [(300, 8, 2), (300, 64, 2), (40000, 16, 2), (300, 8, 3), (300, 150, 2), (300, 4, 4)])
def test_fm_adagrad_with_cross_products(d, k, degree):
    """dense batches (every feature touched ~14 times), the sparse regime (d = 40000: most touches are singles, updated by the
    row phase -- untouched by the rule), degree 3 (both blocks in one walk), degree 4, a wide model (k = 150: two blocks)"""
    _fm_case(5000, d, 8, k, degree, 512, seed=d + k + degree)


def test_the_rule_changes_what_it_should():
    """with gamma = 0.1 the fit differs from the plain rule where coordinates are touched several times per batch, and a
    mini-batch of ONE sample is the reference's step whatever gamma is"""
    (P1, w1, b1), (Xo, y, P0, w0, perms) = _fm_case(3000, 200, 8, 8, 2, 512, seed=7, gamma=G)
    (Pz, wz, bz), _ = _fm_case(3000, 200, 8, 8, 2, 512, seed=7, gamma=0.0)
    assert np.max(np.abs(P1 - Pz)) > 1e-6
    Xs = O.Dataset(Xo.indptr[:201], Xo.indices[:200 * 8], Xo.data[:200 * 8], 200, 200)
    fits = []
    for gamma in (0.0, 0.5):
        fm = gpu_fm("regression", 2, 8, "explicit", True, True, P0, w0, 0.1)
        nf.newAdaGrad(maxIter=1, verbose=0, tol=0, shuffle=False, mode="minibatch", batch=1, adaCross=gamma).fit(to_gpu(Xs), y[:200], fm)
        fits.append((np.array(fm.P), np.array(fm.w), fm.intercept))
    assert np.array_equal(fits[0][0], fits[1][0]) and np.array_equal(fits[0][1], fits[1][1]) and fits[0][2] == fits[1][2]


def test_heavy_features_take_the_cross_products_over_all_their_segments():
    """features touched by a third of every batch: segment partial sums (k_heavy_partial), then one apply per feature"""
    _fm_case(4000, 300, 6, 8, 2, 1024, seed=11, hot=3)


@pytest.mark.parametrize("k,batch", [(4, 700), (8, 300)])
def test_field_aware_units(k, batch):
    rng = np.random.default_rng(k)
    n, F, per = 2500, 4, 12
    cards = [2, 12, 12, 12]  # field 0: two features, each touched by half of every batch (the heavy path)
    d = F * per
    idx = np.stack([f * per + rng.integers(0, cards[f], size=n) for f in range(F)], axis=1)
    val = rng.uniform(-1, 1, size=(n, F))
    Xo = O.Dataset(np.arange(n + 1) * F, idx.ravel(), val.ravel(), n, d, fields=np.tile(np.arange(F), n), n_fields=F)
    y = rng.standard_normal(n)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.05)
    perms = make_perms(n, 2)
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), b0, 1
    st = O.AdaState(F, d, k, d)
    for e in range(2):
        b, it, ls, vs = O.ffm_adagrad_epoch_mb(Xo, y, P, w, b, cfg, batch, st, perm=perms[e], it=it, ada_cross=G)
    b = O.ffm_adagrad_finalize(P, w, b, cfg, it, st)
    ffm = gpu_ffm("regression", k, True, True, P0, w0, b0)
    nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=batch, adaCross=G).fit(to_gpu(Xo), y, ffm, perms=perms)
    assert abs(ffm.intercept - b) < 1e-10
    assert_close(ffm.w, w, 1e-9, 1e-13, "w")
    assert_close(ffm.P, P, 1e-9, 1e-13, "P")


def test_only_adagrad_in_minibatch_mode_takes_it():
    rng = np.random.default_rng(1)
    Xo = random_csr(100, 50, 4, seed=2)
    fm = gpu_fm("regression", 2, 4, "explicit", True, True, rng.standard_normal((1, 4, 50)) * 0.01, np.zeros(50), 0.0)
    with pytest.raises(ValueError):
        nf.newAdaGrad(adaCross=-0.1)
    from nimfm_amd import _capi as capi
    sgd = nf.newSGD(maxIter=1, verbose=0, tol=0, mode="minibatch", batch=16)
    X = to_gpu(Xo)
    sgd._handle(fm, X.ctx, "minibatch")
    assert capi.lib().nfm_opt_set_ada_cross(sgd._h, 0.1) == capi.ERR_UNSUPPORTED
