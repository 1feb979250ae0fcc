"""CPU: the restatement of the reference's text loaders (oracle/ingest.py <- dataset.nim:562-632, 696-790)
against the committed fixtures (tests/golden/ingest_*.txt + ingest_golden.npz) and the reference's own
test shape -- dump, load, compare (tests/test_dataset.nim) -- plus its error behaviour."""
import os

import numpy as np
import pytest

from oracle import ingest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = ["ingest_svm_1based.txt", "ingest_svm_0based.txt", "ingest_svm_digits.txt", "ingest_ffm_1based.txt",
         "ingest_ffm_0based.txt"]


def read(name):
    with open(os.path.join(GOLD, name), newline="") as f:
        return f.read()


@pytest.mark.parametrize("name", FILES)
def test_oracle_matches_fixtures(name):
    g = np.load(os.path.join(GOLD, "ingest_golden.npz"))
    r = ingest.load_ffm(read(name)) if "ffm" in name else ingest.load_svmlight(read(name))
    for k, v in r.items():
        want = g[name + ":" + k]
        assert np.array_equal(np.asarray(v), want, equal_nan=True), (name, k)


def test_hand_checked_values():
    r = ingest.load_svmlight(read("ingest_svm_0based.txt"))
    assert r["offset"] == 0 and r["n_features"] == 10
    assert r["indptr"].tolist() == [0, 2, 5, 5, 6] and r["indices"].tolist() == [0, 4, 1, 2, 3, 9]
    assert r["y"].tolist() == [3.0, -2.0, -2.0, 7.0]  # the empty line keeps the previous target
    r = ingest.load_ffm(read("ingest_ffm_1based.txt"))
    assert (r["offset"], r["offset_field"], r["n_features"], r["n_fields"]) == (1, 1, 8, 4)
    assert r["fields"].tolist() == [0, 1, 3, 2, 0, 3] and r["indices"].tolist() == [0, 2, 6, 1, 0, 7]


def test_round_trip_like_the_reference_test():
    rng = np.random.default_rng(0)
    n, d = 50, 30
    dense = rng.uniform(-1, 1, size=(n, d)) * (rng.random((n, d)) < 0.3)
    dense[:, 0][0] = 0.7  # column 0 present: 1-based dump starts at index 1
    dense[-1, d - 1] = -0.3
    indptr = np.concatenate([[0], np.cumsum((dense != 0).sum(1))])
    rows, cols = np.nonzero(dense)
    y = rng.standard_normal(n)
    r = ingest.load_svmlight(ingest.dump_svmlight(indptr, cols, dense[rows, cols], y))
    assert r["n_features"] == d and np.array_equal(r["indptr"], indptr) and np.array_equal(r["indices"], cols)
    assert np.array_equal(r["data"], dense[rows, cols]) and np.array_equal(r["y"], y)
    fields = cols // 10
    r = ingest.load_ffm(ingest.dump_ffm(indptr, cols, fields, dense[rows, cols], y))
    assert np.array_equal(r["fields"], fields) and r["n_fields"] == 3 and np.array_equal(r["data"], dense[rows, cols])


def test_errors():
    with pytest.raises(ValueError, match="Negative index"):
        ingest.load_svmlight("1 -1:0.5 2:1")
    with pytest.raises(ValueError, match="nFeatures is 3"):
        ingest.load_svmlight("1 1:0.5 5:1", n_features=3)
    assert ingest.load_svmlight("1 1:0.5 5:1", n_features=9)["n_features"] == 9
    with pytest.raises(ValueError, match="nFields is 1"):
        ingest.load_ffm("1 1:1:0.5 3:5:1", n_fields=1)


def test_c_restatement_matches_python_restatement():
    """oracle/nimfm_ingest.c (the CPU baseline of bench.py --workload ingest) == oracle/ingest.py"""
    import oracle as O

    rng = np.random.default_rng(4)
    texts = [read(n) for n in FILES if "svm" in n]
    n, d = 200, 90
    dense = rng.uniform(-1, 1, size=(n, d)) * 10.0 ** rng.integers(-6, 6, size=(n, d)) * (rng.random((n, d)) < 0.2)
    rows, cols = np.nonzero(dense)
    indptr = np.concatenate([[0], np.cumsum((dense != 0).sum(1))])
    texts.append(ingest.dump_svmlight(indptr, cols, dense[rows, cols], rng.standard_normal(n)))
    texts += ["", "1.5\n", "2 1:1\n\n3 2:2"]
    for t in texts:
        a, b = ingest.load_svmlight(t), O.svmlight_load_c(t)
        for k in ("indptr", "indices", "n_features", "offset"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (k, t[:40])
        assert np.array_equal(a["data"].view(np.uint64), b["data"].view(np.uint64))
        assert np.array_equal(a["y"].view(np.uint64), b["y"].view(np.uint64))


def test_stream_format_round_trip():
    """convertSVMLightFile -> newStreamCSRDataset == loadSVMLightFile on the same text (ids shifted by the
    smallest index = the loaders' 0/1 base), header fields as the reference writes them"""
    for name in ("ingest_svm_1based.txt", "ingest_svm_0based.txt"):
        text = read(name)
        xb, yb = ingest.convert_svmlight(text)
        assert xb[:9] == b"STREAMCSR" and len(yb) == 8 * len(ingest.load_svmlight(text)["y"])
        a, b = ingest.load_svmlight(text), ingest.read_stream(xb, yb)
        for k in ("indptr", "indices", "data", "y", "n_features"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (name, k)
        assert b["max"] == a["data"].max() and b["min"] == a["data"].min() and b["nnz"] == len(a["data"])


def ref_literal():
    """the one literal dataset the reference's tests hold near this path (tests/test_dataset.nim:124-128)"""
    import json
    with open(os.path.join(GOLD, "ref_dataset_literal.json")) as f:
        lit = json.load(f)
    return np.array(lit["data"], dtype=np.float64), np.array(lit["yTrue"], dtype=np.float64)


def dense_of(r):
    n = len(r["indptr"]) - 1
    out = np.zeros((n, r["n_features"]))
    for i in range(n):
        for q in range(r["indptr"][i], r["indptr"][i + 1]):
            out[i, r["indices"][q]] = r["data"][q]
    return out


def literal_text(dense, y):
    """dumpSVMLightFile(f, X: seq[seq[float64]], y) (dataset.nim:808-822): zeros skipped, 1-based, no final newline"""
    rows, cols = np.nonzero(dense)
    indptr = np.concatenate([[0], np.cumsum((dense != 0).sum(1))])
    return ingest.dump_svmlight(indptr, cols, dense[rows, cols], y)


def test_reference_literal_dataset():
    """tests/test_dataset.nim:130-156 ('Test CSRDataset', 'Test streamLabel') on the restatement: dump -> load -> compare
    with the literal (checkDenseCSR, :7-28), dump the loaded data again -> load -> compare, convert -> stream files."""
    dense, y = ref_literal()
    text = literal_text(dense, y)
    assert text == "-1.0 3:1.0 5:-4.2\n2.0 1:-3.0\n-10.0\n5.2 4:-5.0 6:103.2"
    r = ingest.load_svmlight(text)
    assert r["n_features"] == 6 and len(r["y"]) == 4 and len(r["data"]) == int((dense != 0).sum())
    assert np.array_equal(dense_of(r), dense) and np.array_equal(r["y"], y)
    r2 = ingest.load_svmlight(ingest.dump_svmlight(r["indptr"], r["indices"], r["data"], r["y"]))
    assert np.array_equal(dense_of(r2), dense)
    xb, yb = ingest.convert_svmlight(text)
    s = ingest.read_stream(xb, yb)
    assert np.array_equal(dense_of(s), dense) and np.array_equal(s["y"], y)
