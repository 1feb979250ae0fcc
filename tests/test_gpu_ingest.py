"""-m gpu: svmlight / libffm text parsed on the GPU (ingest.hip, through nfm_dataset_load_* /
nfm_dataset_parse_text) against the CPU restatement of the reference's loaders (oracle/ingest.py <-
dataset.nim:562-632, 696-790): committed fixtures, the reference's dump -> load round trip
(tests/test_dataset.nim), edge cases, error behaviour, and a file large enough to time."""
import os
import time

import numpy as np
import pytest

import nimfm_amd as nf
from oracle import ingest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = ["ingest_svm_1based.txt", "ingest_svm_0based.txt", "ingest_svm_digits.txt", "ingest_ffm_1based.txt",
         "ingest_ffm_0based.txt"]


def same(ds, y, r, with_fields):
    indptr, indices, data, fields = ds.to_host()
    assert ds.nSamples == len(r["y"]) and ds.nFeatures == r["n_features"]
    assert np.array_equal(indptr, r["indptr"]) and np.array_equal(indices, r["indices"])
    assert np.array_equal(data.view(np.uint64), r["data"].view(np.uint64))  # bit for bit
    assert np.array_equal(y.view(np.uint64), r["y"].view(np.uint64))
    if with_fields:
        assert ds.nFields == r["n_fields"] and np.array_equal(fields, r["fields"])


@pytest.mark.parametrize("name", FILES)
def test_fixtures_from_file(name):
    path = os.path.join(GOLD, name)
    with open(path, newline="") as f:
        text = f.read()
    ffm = "ffm" in name
    ds, y = nf.loadFFMFile(path) if ffm else nf.loadSVMLightFile(path)
    same(ds, y, ingest.load_ffm(text) if ffm else ingest.load_svmlight(text), ffm)
    g = np.load(os.path.join(GOLD, "ingest_golden.npz"))
    assert np.array_equal(ds.to_host()[1], g[name + ":indices"])


def random_csr_text(rng, n, d, density, ffm=False, n_fields=4, wide=True):
    mask = rng.random((n, d)) < density
    mask[0, 0] = mask[n - 1, d - 1] = True
    rows, cols = np.nonzero(mask)
    data = rng.uniform(-1, 1, size=len(rows))
    if wide:  # many decades, so that the exponent forms of repr() are exercised
        data = data * 10.0 ** rng.integers(-8, 8, size=len(rows))
    indptr = np.concatenate([[0], np.cumsum(mask.sum(1))])
    y = rng.standard_normal(n)
    if ffm:
        fields = cols % n_fields
        return ingest.dump_ffm(indptr, cols, fields, data, y)
    return ingest.dump_svmlight(indptr, cols, data, y)


@pytest.mark.parametrize("ffm", [False, True])
def test_dump_then_load_round_trip(ffm, tmp_path):
    rng = np.random.default_rng(3)
    text = random_csr_text(rng, 300, 120, 0.2, ffm)
    r = ingest.load_ffm(text) if ffm else ingest.load_svmlight(text)
    ds, y = nf.parseText(text, withFields=ffm)
    same(ds, y, r, ffm)
    p = tmp_path / "data.txt"
    p.write_text(text + "\n")  # with a final newline: same samples (Nim's `lines`)
    ds2, y2 = nf.loadFFMFile(str(p)) if ffm else nf.loadSVMLightFile(str(p))
    same(ds2, y2, r, ffm)


def test_edge_cases():
    for text in ["", "\n", "1.5", "1.5\n", "2 1:1\n\n\n3 2:2", "1 1:1\t2:2\r\n0 3:3\r\n", "-0.0 7:-0.0", "nan 1:inf 2:-inf"]:
        r = ingest.load_svmlight(text)
        ds, y = nf.parseText(text)
        assert ds.nSamples == len(r["y"]), repr(text)
        indptr, indices, data, _ = ds.to_host()
        assert np.array_equal(indptr, r["indptr"]) and np.array_equal(indices, r["indices"]), repr(text)
        assert np.array_equal(data.view(np.uint64), r["data"].view(np.uint64)), repr(text)
        assert np.array_equal(np.isnan(y), np.isnan(r["y"])) and np.array_equal(y[~np.isnan(y)], r["y"][~np.isnan(y)]), repr(text)
    # nFeatures given: larger wins, smaller is the reference's ValueError
    ds, _ = nf.parseText("1 1:0.5 5:1", nFeatures=9)
    assert ds.nFeatures == 9
    with pytest.raises(ValueError, match="nFeatures is 3"):
        nf.parseText("1 1:0.5 5:1", nFeatures=3)
    with pytest.raises(ValueError, match="Negative index"):
        nf.parseText("1 -1:0.5 2:1")
    with pytest.raises(ValueError, match="nFields is 1"):
        nf.parseText("1 1:1:0.5 3:5:1", withFields=True, nFields=1)
    # text the reference would mis-read silently is an error here
    for bad in ["1 1:0.5 2", "1 1:0.5junk", "1 a:0.5", "1 1:1:0.5", "1 1:", "x 1:2"]:
        with pytest.raises(ValueError):
            nf.parseText(bad)
    with pytest.raises(ValueError):
        nf.parseText("1 1:0.5", withFields=True)
    with pytest.raises(ValueError, match="cannot be read"):
        nf.loadSVMLightFile("/nonexistent/file.svm")


def test_loaded_dataset_trains_like_the_host_built_one(tmp_path):
    rng = np.random.default_rng(5)
    n, d, k = 400, 64, 8
    text = random_csr_text(rng, n, d, 0.15, wide=False)
    p = tmp_path / "train.svm"
    p.write_text(text)
    X, y = nf.loadSVMLightFile(str(p))
    r = ingest.load_svmlight(text)
    Xh = nf.newCSRDataset(data=r["data"], indices=r["indices"], indptr=r["indptr"], nSamples=n, nFeatures=d)
    outs = []
    for ds in (X, Xh):
        fm = nf.newFactorizationMachine("regression", nComponents=k, randomState=1)
        nf.newSGD(maxIter=3, verbose=0, tol=0, shuffle=False).fit(ds, y, fm)
        outs.append((fm.P.copy(), fm.w.copy(), fm.intercept))
    assert np.all(np.isfinite(outs[0][0]))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


def test_larger_file_and_throughput(tmp_path, capsys):
    rng = np.random.default_rng(9)
    n, d, m = 200_000, 100_000, 32
    idx = np.arange(m) * (d // m) + rng.integers(0, d // m, size=(n, m))  # one id per stratum: distinct inside a row
    idx[0, 0], idx[-1, -1] = 0, d - 1
    val = rng.uniform(-1, 1, size=(n, m))
    y = np.sign(rng.standard_normal(n))
    p = tmp_path / "big.svm"
    with open(p, "w") as f:
        for i in range(n):
            f.write(repr(float(y[i])) + " " + " ".join("%d:%r" % (idx[i, q] + 1, float(val[i, q])) for q in range(m)) + "\n")
    t0 = time.perf_counter()
    ds, yg = nf.loadSVMLightFile(str(p))
    wall = time.perf_counter() - t0
    indptr, indices, data, _ = ds.to_host()
    assert ds.nSamples == n and ds.nFeatures == d
    assert np.array_equal(indptr, np.arange(n + 1) * m) and np.array_equal(indices, idx.ravel())
    assert np.array_equal(data, val.ravel()) and np.array_equal(yg, y)
    nbytes, up_ms, parse_ms = ds.ingest_stats()
    with capsys.disabled():
        print("\n[ingest] %.1f MB text: read+upload %.1f ms, GPU parse %.1f ms (%.1f GB/s), wall %.2f s"
              % (nbytes / 1e6, up_ms, parse_ms, nbytes / parse_ms / 1e6, wall))


def test_stream_files(tmp_path):
    """STREAMCSR / STREAMCSRFIELD files (tensor/sparse_stream.nim:3-33): written by the restatement of
    convertSVMLightFile and by nfm_convert_svmlight, read back by nfm_dataset_load_stream"""
    rng = np.random.default_rng(11)
    text = random_csr_text(rng, 500, 200, 0.1)
    src = tmp_path / "in.svm"
    src.write_text(text)
    xb, yb = ingest.convert_svmlight(text)
    # the converter: byte-identical files
    nf.convertSVMLightFile(str(src), str(tmp_path / "x.bin"), str(tmp_path / "y.bin"))
    assert (tmp_path / "x.bin").read_bytes() == xb and (tmp_path / "y.bin").read_bytes() == yb
    # the reader
    ds, y = nf.newStreamCSRDataset(str(tmp_path / "x.bin"), str(tmp_path / "y.bin"))
    same(ds, y, ingest.read_stream(xb, yb), False)
    same(ds, y, ingest.load_svmlight(text), False)
    ds0, y0 = nf.newStreamCSRDataset(str(tmp_path / "x.bin"))
    assert np.array_equal(y0, np.zeros(ds0.nSamples))
    # field variant
    r = ingest.load_ffm(random_csr_text(rng, 300, 90, 0.15, ffm=True, n_fields=6))
    fb = ingest.write_stream_field(r["indptr"], r["indices"], r["fields"], r["data"], r["n_features"], r["n_fields"])
    (tmp_path / "f.bin").write_bytes(fb)
    dsf, _ = nf.newStreamCSRDataset(str(tmp_path / "f.bin"))
    indptr, indices, data, fields = dsf.to_host()
    assert dsf.nFields == r["n_fields"] and np.array_equal(fields, r["fields"]) and np.array_equal(indices, r["indices"])
    assert np.array_equal(indptr, r["indptr"]) and np.array_equal(data, r["data"])
    # errors
    (tmp_path / "bad.bin").write_bytes(b"STREAMCSC" + xb[9:])
    with pytest.raises(nf.NfmError):
        nf.newStreamCSRDataset(str(tmp_path / "bad.bin"))
    (tmp_path / "trunc.bin").write_bytes(xb[:-5])
    with pytest.raises(ValueError):
        nf.newStreamCSRDataset(str(tmp_path / "trunc.bin"))
    (tmp_path / "junk.bin").write_bytes(b"hello world, not a matrix")
    with pytest.raises(ValueError, match="not a StreamCSR"):
        nf.newStreamCSRDataset(str(tmp_path / "junk.bin"))
    # a header whose nCols is smaller than the ids the rows hold (they would index P / w out of bounds), whose nFields
    # is smaller than the fields, or whose nRows the file cannot hold
    import struct
    n_, d_, nnz_ = struct.unpack_from("<qqq", xb, 9)
    (tmp_path / "cols.bin").write_bytes(xb[:9] + struct.pack("<qqq", n_, max(1, d_ // 2), nnz_) + xb[33:])
    with pytest.raises(ValueError, match="outside"):
        nf.newStreamCSRDataset(str(tmp_path / "cols.bin"))
    (tmp_path / "rows.bin").write_bytes(xb[:9] + struct.pack("<qqq", 1 << 60, d_, nnz_) + xb[33:])
    with pytest.raises(ValueError, match="promises"):
        nf.newStreamCSRDataset(str(tmp_path / "rows.bin"))
    nF = struct.unpack_from("<q", fb, 14 + 24)[0]
    (tmp_path / "nf.bin").write_bytes(bytes(fb[:14 + 24]) + struct.pack("<q", nF - 1) + bytes(fb[14 + 32:]))
    with pytest.raises(ValueError, match="outside"):
        nf.newStreamCSRDataset(str(tmp_path / "nf.bin"))


def test_stream_file_in_row_blocks(tmp_path):
    """newStreamCSRDataset(f, cacheRows): the out-of-core path (tensor/sparse_stream.nim:232-270 readCache; epoch loop
    optimizer/sgd_multi.nim:83-97) -- at most cacheRows rows resident, blocks walked in file order, the optimizer's step
    counter / scales / state continuing across blocks.  In the reference's sample order this must equal the fit over the
    whole matrix; in mini-batch mode (cacheRows a multiple of the batch) likewise."""
    import oracle as O
    from common import assert_close
    rng = np.random.default_rng(13)
    n, d, k = 700, 120, 4
    text = random_csr_text(rng, n, d, 0.08, wide=False)
    xb, yb = ingest.convert_svmlight(text)
    (tmp_path / "x.bin").write_bytes(xb)
    (tmp_path / "y.bin").write_bytes(yb)
    ref = ingest.read_stream(xb, yb)
    Xo = O.Dataset(ref["indptr"], ref["indices"], ref["data"], len(ref["indptr"]) - 1, ref["n_features"])
    dd = ref["n_features"]
    y = ref["y"]
    P0, w0 = rng.standard_normal((1, k, dd)) * 0.05, np.zeros(dd)
    Xs, ys = nf.newStreamCSRDataset(str(tmp_path / "x.bin"), str(tmp_path / "y.bin"), cacheRows=256)
    assert isinstance(Xs, nf.StreamCSRDataset) and Xs.nSamples == n and Xs.nFeatures == dd and np.array_equal(ys, y)
    assert Xs.blocks() == [(0, 256), (256, 512), (512, 700)]
    blk = Xs.load(256, 512)
    ip, ix, dv, _ = blk.to_host()
    a, b = ref["indptr"][256], ref["indptr"][512]
    assert np.array_equal(ip, ref["indptr"][256:513] - a) and np.array_equal(ix, ref["indices"][a:b]) and np.array_equal(dv, ref["data"][a:b])
    assert np.array_equal(blk.targets(), y[256:512])
    # the reference's order (sequential mode), two epochs
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    fm.set_params(P0, w0, 0.0)
    sgd = nf.newSGD(maxIter=2, verbose=0, tol=0, shuffle=True, mode="sequential")  # shuffle is ignored for a streamed dataset
    sgd.fit(Xs, y, fm)
    P, w, b, it, el, ev, _ = O.fm_sgd_fit(Xo, y, 2, P0, w0, 0.0, O.sgd_cfg(), 2)
    assert sgd.it == it
    assert_close(fm.P, P, 1e-9, 1e-12, "P, sequential")
    assert_close(fm.w, w, 1e-9, 1e-12, "w, sequential")
    assert_close([h[1] for h in sgd.history], el, 1e-10, 1e-13, "mean loss per epoch")
    assert_close(fm.decisionFunction(Xs), O.fm_decision_function(Xo, 2, P, w, b), 1e-10, 1e-13, "decisionFunction over blocks")
    # mini-batch mode: blocks of 256 = 4 batches of 64
    fm = nf.newFactorizationMachine("regression", nComponents=k, warmStart=True)
    fm.set_params(P0, w0, 0.0)
    ada = nf.newAdaGrad(maxIter=2, verbose=0, tol=0, mode="minibatch", batch=64)
    ada.fit(Xs, y, fm)
    cfg = O.adagrad_cfg()
    P, w, b, it = P0.copy(), w0.copy(), 0.0, 1
    st = O.AdaState(1, dd, k, dd)
    for _ in range(2):
        for r0, r1 in Xs.blocks():
            b, it, _, _ = O.fm_adagrad_epoch_mb(Xo, y, 2, P, w, b, cfg, 64, st, begin=r0, end=r1, it=it)
    b = O.fm_adagrad_finalize(2, P, w, b, cfg, it, st)
    assert_close(fm.P, P, 1e-9, 1e-12, "P, mini-batch")
    assert abs(fm.intercept - b) < 1e-11


def test_stream_block_loaded_ahead(tmp_path):
    """nfm_stream_prefetch_rows: the block asked for ahead is the block a plain load returns; asking for another range
    drops it; an unclaimed one is released with the stream object."""
    rng = np.random.default_rng(4)
    n, d = 5000, 300
    text = random_csr_text(rng, n, d, 0.05, wide=False)
    xb, yb = ingest.convert_svmlight(text)
    (tmp_path / "x.bin").write_bytes(xb)
    (tmp_path / "y.bin").write_bytes(yb)
    Xs, _ = nf.newStreamCSRDataset(str(tmp_path / "x.bin"), str(tmp_path / "y.bin"), cacheRows=1024)
    plain = [Xs.load(r0, r1).to_host() for r0, r1 in Xs.blocks()]
    for bi, (r0, r1) in enumerate(Xs.blocks()):  # each block ahead of its load, the next one requested before this one is used
        if bi == 0:
            Xs.prefetch(r0, r1)
        blk = Xs.load(r0, r1)
        nxt = Xs.blocks()[(bi + 1) % len(Xs.blocks())]
        Xs.prefetch(*nxt)
        got = blk.to_host()
        for a, b in zip(got[:3], plain[bi][:3]):
            assert np.array_equal(a, b)
    other = Xs.load(1024, 2048)  # not the range asked for ahead: loaded in line, the other block dropped
    assert np.array_equal(other.to_host()[1], plain[1][1])
    Xs.prefetch(0, 1024)
    with pytest.raises(ValueError):
        Xs.prefetch(0, n + 1)
    del Xs  # an unclaimed block goes with the stream object


def test_reference_literal_dataset_through_the_device(tmp_path):
    """The literal 4 x 6 matrix and targets of the reference's dataset suite (tests/test_dataset.nim:124-128, committed as
    tests/golden/ref_dataset_literal.json) pushed through every loader of this path, as its tests do (:130-156):
    dump -> GPU loadSVMLightFile -> compare with the literal; convertSVMLightFile on the GPU -> STREAMCSR + label files ->
    nfm_dataset_load_stream -> compare; the same file in row blocks (cacheRows = 2: an empty row starts the second block)."""
    from test_oracle_ingest import dense_of, literal_text, ref_literal
    dense, y = ref_literal()
    src = tmp_path / "testsample.svm"
    src.write_text(literal_text(dense, y))

    def as_dict(ds, yy):
        indptr, indices, data, _ = ds.to_host()
        return {"indptr": indptr, "indices": indices, "data": data, "n_features": ds.nFeatures, "y": yy}

    ds, yg = nf.loadSVMLightFile(str(src))
    assert ds.nSamples == 4 and ds.nFeatures == 6 and ds.nnz == 5
    assert np.array_equal(dense_of(as_dict(ds, yg)), dense) and np.array_equal(yg, y)
    nf.convertSVMLightFile(str(src), str(tmp_path / "testsample"), str(tmp_path / "testlabel"))
    xb, yb = ingest.convert_svmlight(src.read_text())
    assert (tmp_path / "testsample").read_bytes() == xb and (tmp_path / "testlabel").read_bytes() == yb
    dss, ys = nf.newStreamCSRDataset(str(tmp_path / "testsample"), str(tmp_path / "testlabel"))
    assert np.array_equal(dense_of(as_dict(dss, ys)), dense) and np.array_equal(ys, y)
    Xs, ys2 = nf.newStreamCSRDataset(str(tmp_path / "testsample"), str(tmp_path / "testlabel"), cacheRows=2)
    assert Xs.blocks() == [(0, 2), (2, 4)] and np.array_equal(ys2, y)
    got = np.vstack([dense_of(as_dict(Xs.load(r0, r1), None)) if Xs.load(r0, r1).nnz else np.zeros((r1 - r0, 6)) for r0, r1 in Xs.blocks()])
    assert np.array_equal(got, dense)
    # and the loaded matrix scores like the literal: decisionFunction of a fixed model on both
    import oracle as O
    rng = np.random.default_rng(5)
    P0, w0 = rng.standard_normal((1, 3, 6)), rng.standard_normal(6)
    fm = nf.newFactorizationMachine("regression", nComponents=3, warmStart=True)
    fm.set_params(P0, w0, 0.25)
    want = O.slow_fm_decision_function(dense, 2, P0, w0, 0.25, 0)
    assert np.allclose(fm.decisionFunction(ds), want, rtol=1e-12, atol=1e-12)
    assert np.allclose(fm.decisionFunction(Xs), want, rtol=1e-12, atol=1e-12)
