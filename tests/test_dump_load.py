"""CPU: the reference's text model format (dump / load, model/factorization_machine.nim:142-220) through
the host mirror -- no device needed: exact layout of a written file, round trip of every field."""
import numpy as np

import nimfm_amd as nf


def test_dump_layout_and_round_trip(tmp_path):
    fm = nf.newFactorizationMachine("classification", degree=3, nComponents=2, fitLower="augment", fitLinear=False,
                                    fitIntercept=True, randomState=7, scale=0.05)
    d = 3
    assert fm.nOrders == 1 and fm.nAugments == 2
    P = np.array([[[0.5, -1.25, 1e-05, 3.0, 0.1], [1.0, 2.0, 1e+20, -0.0, 0.30000000000000004]]])
    fm.set_params(P, np.array([0.25, 0.0, -2.0]), 0.125)
    p = tmp_path / "model.txt"
    fm.dump(str(p))
    want = ("task: classification\nnFeatures: 3\ndegree: 3\nnComponents: 2\nfitLower: augment\nfitIntercept: true\n"
            "fitLinear: false\nrandomState: 7\nscale: 0.05\nlams:\n1.0 1.0\nP[0]:\n0.5 -1.25 1e-05 3.0 0.1\n"
            "1.0 2.0 1e+20 -0.0 0.30000000000000004\nw:\n0.25 0.0 -2.0\nintercept: 0.125\n")
    assert p.read_text() == want
    g = nf.load(str(p), True)
    assert (g.task, g.degree, g.nComponents, g.fitLower, g.fitIntercept, g.fitLinear, g.randomState, g.scale, g.warmStart) == \
        ("classification", 3, 2, "augment", True, False, 7, 0.05, True)
    assert g.isInitialized and np.array_equal(g.P, P) and np.array_equal(g.w, fm.w) and g.intercept == 0.125
    assert np.array_equal(g.lams, np.ones(2)) and d == len(g.w)
    p2 = tmp_path / "again.txt"
    g.dump(str(p2))
    assert p2.read_text() == want


def test_round_trip_random(tmp_path):
    rng = np.random.default_rng(3)
    for degree, lower, lin in [(2, "explicit", True), (4, "explicit", True), (2, "none", False), (3, "augment", True)]:
        fm = nf.newFactorizationMachine("regression", degree=degree, nComponents=5, fitLower=lower, fitLinear=lin)
        d = 11
        P = rng.standard_normal((fm.nOrders, 5, d + fm.nAugments)) * 10.0 ** rng.integers(-6, 6)
        fm.set_params(P, rng.standard_normal(d), float(rng.standard_normal()))
        p = tmp_path / ("m%d.txt" % degree)
        fm.dump(str(p))
        g = nf.load(str(p), False)
        assert np.array_equal(g.P, fm.P) and np.array_equal(g.w, fm.w) and g.intercept == fm.intercept and not g.warmStart
