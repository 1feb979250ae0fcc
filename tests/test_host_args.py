"""Host-side argument handling that needs no GPU: fit(devices=[...]) refuses the combinations it would otherwise ignore
(permutations, streamed datasets and nCalls callbacks were silently dropped)."""
import numpy as np
import pytest

import nimfm_amd as nf


def test_fit_devices_refuses_what_it_cannot_honour():
    opt = nf.newSGD(maxIter=1, verbose=0, mode="minibatch")
    with pytest.raises(ValueError, match="perms"):
        opt.fit(None, None, None, devices=[0, 1], perms=np.zeros((1, 4), dtype=np.int64))
    opt2 = nf.newSGD(maxIter=1, verbose=0, mode="minibatch", nCalls=10)
    with pytest.raises(ValueError, match="nCalls"):
        opt2.fit(None, None, None, devices=[0, 1], callback=lambda o, m: None)


def test_combine_names():
    opt = nf.newAdaGrad(maxIter=1, verbose=0, mode="minibatch")
    with pytest.raises(ValueError, match="combine"):
        opt.setDataParallel(object(), 0, True, "median")
    for name in ("auto", "mean", "sum", "state_mean", "state_rsqrt"):
        opt.setDataParallel(None, 0, True, name)  # (no group: nothing is attached, the name is accepted)
