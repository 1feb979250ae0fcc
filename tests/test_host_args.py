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


def test_suggested_touch_cap_is_what_the_bench_settings_were_measured_at():
    """nf.suggestTouchCap: about twice the touches per coordinate and batch, a power of two in [16, 64] -- and it gives the caps
    bench.py's SGD workloads run with (chosen there by time-to-target sweeps, profiles/r05h_touch_cap_sweep.txt)"""
    import types
    import bench
    import nimfm_amd as nf
    for name in ("headline", "cfg2", "cfg5", "wide256"):
        wl = bench.WORKLOADS[name]
        X = types.SimpleNamespace(nSamples=wl["n"], nFeatures=wl["d"], nnz=wl["n"] * wl["m"])
        assert nf.suggestTouchCap(X, wl["batch"]) == float(wl.get("touch_cap", 16.0)), name
    X = types.SimpleNamespace(nSamples=10_000_000, nFeatures=1_000_000, nnz=640_000_000)
    assert [nf.suggestTouchCap(X, b) for b in (8192, 65536, 131072, 262144, 524288, 4_000_000)] == [16.0, 16.0, 16.0, 32.0, 64.0, 64.0]
    small = types.SimpleNamespace(nSamples=100, nFeatures=50, nnz=400)
    assert nf.suggestTouchCap(small, 1_000_000) == 16.0  # (the batch cannot exceed the dataset)
    with pytest.raises(ValueError):
        nf.suggestTouchCap(types.SimpleNamespace(nSamples=0, nFeatures=5, nnz=0), 8)


def test_cli_takes_the_mini_batch_knobs():
    """`python -m nimfm_amd train ... --mode minibatch --batch B --touchCap auto|c --adaCross g` (both spellings of an option, as for
    the reference's own options); the defaults are the library's (the mean, no cross products)"""
    from nimfm_amd import cli
    base = ["train", "-t", "r", "--train", "a.svm"]
    a = cli._parser().parse_args(base)
    assert a.touchCap == "1" and a.adaCross == 0.0 and a.mode == "sequential"
    a = cli._parser().parse_args(base + ["--mode", "minibatch", "--batch", "65536", "--touch-cap", "auto", "--ada-cross", "0.1"])
    assert a.touchCap == "auto" and a.adaCross == 0.1 and a.batch == 65536
    assert cli._parser().parse_args(base + ["--touchCap", "32"]).touchCap == "32"
