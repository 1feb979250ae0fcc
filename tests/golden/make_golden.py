"""Generates tests/golden/*.npz: small input/output vectors for the FM hot path.

PROVENANCE (read this): the reference is Nim-only and cannot be run in this image, and its tests
hold no literal vectors for this path.  These vectors are therefore produced by the
BRUTE-FORCE restatement of the reference's own test oracles (oracle/nimfm_slow.c: explicit subset
enumeration, dense per-step updates -- tests/kernels_slow.nim, tests/model/fm_slow.nim,
tests/optimizer/{sgd,adagrad}_slow.nim, tests/model/ffm_slow.nim), i.e. by the mathematical
definition, not by the code paths under test (oracle/nimfm_oracle.c, the HIP kernels).  They pin
regressions and cross-implementation agreement; they are NOT outputs of the reference binary.

Run from the repo root:  python tests/golden/make_golden.py
"""
import itertools
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle as O  # noqa: E402
from common import init_ffm, init_fm, make_ffm_dataset, make_fm_dataset, make_perms  # noqa: E402

N, D, K, EPOCHS = 40, 8, 4, 2


def main():
    out = {}
    for degree, fit_lower in itertools.product([2, 3], ["explicit", "augment", "none"]):
        tag = "fm_d%d_%s" % (degree, fit_lower)
        Xo, Xd, y = make_fm_dataset(N, D, degree, K, 42, fit_lower, threshold=0.3)
        P0, w0, b0, n_aug = init_fm(D, degree, K, fit_lower, True, scale=0.1)
        perms = make_perms(N, EPOCHS)
        rng = np.random.default_rng(9)
        wq = rng.standard_normal(D)
        out[tag + "_X"] = Xd
        out[tag + "_y"] = y
        out[tag + "_P0"] = P0
        out[tag + "_perms"] = perms
        out[tag + "_wq"] = wq
        out[tag + "_decision"] = O.slow_fm_decision_function(Xd, degree, P0, wq, 0.25, n_aug)
        Ps, ws, bs, _ = O.slow_fm_sgd_fit(Xd, y, degree, P0, w0, b0, O.sgd_cfg(), EPOCHS, n_aug, perms)
        out[tag + "_sgd_P"], out[tag + "_sgd_w"], out[tag + "_sgd_b"] = Ps, ws, np.array(bs)
        Pa, wa, ba, _ = O.slow_fm_adagrad_fit(Xd, y, degree, P0, w0, b0, O.adagrad_cfg(), EPOCHS, n_aug, perms)
        out[tag + "_ada_P"], out[tag + "_ada_w"], out[tag + "_ada_b"] = Pa, wa, np.array(ba)
    n, d, F, k = 40, 12, 3, 4
    Xo, Xd, field_of, y = make_ffm_dataset(n, d, F, k, 42, threshold=0.3)
    P0, w0, b0 = init_ffm(d, F, k, scale=0.1)
    out["ffm_X"], out["ffm_y"], out["ffm_P0"], out["ffm_field_of"] = Xd, y, P0, field_of
    out["ffm_decision"] = O.slow_ffm_decision_function(Xd, field_of, F, P0, np.linspace(-1, 1, d), -0.5)
    Ps, ws, bs, _ = O.slow_ffm_sgd_fit(Xd, field_of, F, y, P0, w0, b0, O.sgd_cfg(), EPOCHS)
    out["ffm_sgd_P"], out["ffm_sgd_w"], out["ffm_sgd_b"] = Ps, ws, np.array(bs)
    Pa, wa, ba, _ = O.slow_ffm_adagrad_fit(Xd, field_of, F, y, P0, w0, b0, O.adagrad_cfg(), EPOCHS)
    out["ffm_ada_P"], out["ffm_ada_w"], out["ffm_ada_b"] = Pa, wa, np.array(ba)
    np.savez_compressed(os.path.join(HERE, "fm_hotpath_golden.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
