"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's text loaders, character by
character as the Nim code walks a line (never imported by the product path):

    loadSVMLightFile   /root/reference/src/nimfm/dataset.nim:562-613 (+ the checks of :616-632)
    loadFFMFile        /root/reference/src/nimfm/dataset.nim:696-765 (+ :768-790)

parse_float / parse_int stand for Nim's parseutils.parseFloat / parseInt: they return the number of
characters consumed (0 = nothing parsed, the variable keeps its previous value), which is what makes the
reference's "previous target on an empty line" behaviour.  Python's float() is correctly rounded, as
Nim's parseFloat is (exact fast path, C strtod otherwise).  Parity status: pinned only by the
reference's own round-trip test shape (tests/test_dataset.nim: dump -> load -> compare with the dense
matrix); no Nim toolchain here, so no reference-run outputs -- "parity unpinned" like the rest of oracle/.
"""
import re

import numpy as np

_FLOAT = re.compile(r"[+-]?(nan|inf(inity)?|(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?)", re.I)
_INT = re.compile(r"[+-]?\d+")


def parse_float(line, k, prev):
    m = _FLOAT.match(line, k)
    if not m:
        return 0, prev
    return len(m.group(0)), float(m.group(0))


def parse_int(line, k, prev):
    m = _INT.match(line, k)
    if not m:
        return 0, prev
    return len(m.group(0)), int(m.group(0))


def _lines(text):
    # Nim's `lines`: splits at \n, \r\n (and \r); no empty last line after a final terminator
    out = text.split("\n")
    if out and out[-1] == "":
        out.pop()
    return [ln[:-1] if ln.endswith("\r") else ln for ln in out]


def load_svmlight(text, n_features=-1):
    """-> dict(indptr, indices, data, y, n_features, offset); raises ValueError like the reference."""
    lines = _lines(text)
    min_index, max_index = 1, 0
    j, val, target = 0, 0.0, 0.0
    for line in lines:  # first pass: index range (dataset.nim:571-584)
        k = 0
        c, target = parse_float(line, k, target)
        k += c + 1
        while k < len(line):
            c, j = parse_int(line, k, j)
            k += c
            min_index, max_index = min(j, min_index), max(j, max_index)
            k += 1
            c, val = parse_float(line, k, val)
            k += c + 1
    if min_index < 0:
        raise ValueError("Negative index is included.")
    offset = 0 if min_index == 0 else 1
    nfeat = max_index + 1 - offset
    indptr, indices, data, y = [0], [], [], []
    target = 0.0
    for line in lines:  # second pass (dataset.nim:598-612)
        k = 0
        c, target = parse_float(line, k, target)
        y.append(target)
        k += c + 1
        while k < len(line):
            c, j = parse_int(line, k, j)
            k += c
            indices.append(j - offset)
            k += 1
            c, val = parse_float(line, k, val)
            k += c
            data.append(val)
            k += 1
        indptr.append(len(indices))
    if n_features > 0 and nfeat > n_features:
        raise ValueError("nFeatures is %d but dataset has at least %d features." % (n_features, nfeat))
    return dict(indptr=np.array(indptr, dtype=np.int64), indices=np.array(indices, dtype=np.int64),
                data=np.array(data, dtype=np.float64), y=np.array(y, dtype=np.float64),
                n_features=max(nfeat, n_features), offset=offset)


def load_ffm(text, n_features=-1, n_fields=-1):
    lines = _lines(text)
    min_index, max_index, min_field, max_field = 1, 0, 1, 1
    j, fld, val, target = 0, 0, 0.0, 0.0
    for line in lines:  # dataset.nim:706-724
        k = 0
        c, target = parse_float(line, k, target)
        k += c + 1
        while k < len(line):
            c, fld = parse_int(line, k, fld)
            k += c
            min_field, max_field = min(fld, min_field), max(fld, max_field)
            k += 1
            c, j = parse_int(line, k, j)
            k += c
            min_index, max_index = min(j, min_index), max(j, max_index)
            k += 1
            c, val = parse_float(line, k, val)
            k += c + 1
    if min_index < 0:
        raise ValueError("Negative index is included.")
    offset = 0 if min_index == 0 else 1
    nfeat = max_index + 1 - offset
    offset_field = 0 if min_field == 0 else 1
    nfld = max_field + 1 - offset_field
    indptr, indices, fields, data, y = [0], [], [], [], []
    target = 0.0
    for line in lines:  # dataset.nim:743-763
        k = 0
        c, target = parse_float(line, k, target)
        y.append(target)
        k += c + 1
        while k < len(line):
            c, fld = parse_int(line, k, fld)
            k += c
            fields.append(fld - offset_field)
            k += 1
            c, j = parse_int(line, k, j)
            k += c
            indices.append(j - offset)
            k += 1
            c, val = parse_float(line, k, val)
            k += c
            data.append(val)
            k += 1
        indptr.append(len(indices))
    if n_fields > 0 and nfld > n_fields:
        raise ValueError("nFields is %d but dataset has at least %d fields." % (n_fields, nfld))
    if n_features > 0 and nfeat > n_features:
        raise ValueError("nFeatures is %d but dataset has at least %d features." % (n_features, nfeat))
    return dict(indptr=np.array(indptr, dtype=np.int64), indices=np.array(indices, dtype=np.int64),
                fields=np.array(fields, dtype=np.int64), data=np.array(data, dtype=np.float64),
                y=np.array(y, dtype=np.float64), n_features=max(nfeat, n_features), n_fields=max(nfld, n_fields),
                offset=offset, offset_field=offset_field)


def dump_svmlight(indptr, indices, data, y):
    """dumpSVMLightFile (dataset.nim:793-805): 1-based, no newline after the last line; Nim's `$float`
    prints the shortest round-trip form like Python's repr (except integral values: "1.0" in both)."""
    rows = []
    for i in range(len(y)):
        s = repr(float(y[i]))
        for q in range(indptr[i], indptr[i + 1]):
            s += " %d:%s" % (indices[q] + 1, repr(float(data[q])))
        rows.append(s)
    return "\n".join(rows)


def dump_ffm(indptr, indices, fields, data, y):
    """dumpFFMFile (dataset.nim:825-837)"""
    rows = []
    for i in range(len(y)):
        s = repr(float(y[i]))
        for q in range(indptr[i], indptr[i + 1]):
            s += " %d:%d:%s" % (fields[q] + 1, indices[q] + 1, repr(float(data[q])))
        rows.append(s)
    return "\n".join(rows)


# ---- the reference's binary out-of-core format (tensor/sparse_stream.nim:3-33) ----
def convert_svmlight(text):
    """convertSVMLightFile (dataset.nim:1017-1097) -> (bytes of the STREAMCSR file, bytes of the label file).
    Ids are shifted by minIndex (initial 1, dataset.nim:1025), nCols = maxIndex - minIndex + 1, the
    header carries max / min of the values (initial low/high(float64) = -Inf/+Inf)."""
    import struct

    lines = _lines(text)
    min_index, max_index = 1, 0
    min_val, max_val = float("inf"), float("-inf")
    j, val, target = 0, 0.0, 0.0
    n_samples = nnz = 0
    for line in lines:
        pos = 0
        n_samples += 1
        c, target = parse_float(line, pos, target)
        pos += c + 1
        while pos < len(line):
            c, j = parse_int(line, pos, j)
            pos += c
            min_index, max_index = min(j, min_index), max(j, max_index)
            pos += 1
            c, val = parse_float(line, pos, val)
            pos += c
            min_val, max_val = min(min_val, val), max(max_val, val)
            pos += 1
            nnz += 1
    if min_index < 0:
        raise ValueError("Negative index is included.")
    x = bytearray(b"STREAMCSR")
    x += struct.pack("<qqqdd", n_samples, max_index - min_index + 1, nnz, max_val, min_val)
    y = bytearray()
    for line in lines:
        pos = 0
        c, target = parse_float(line, pos, target)
        pos += c + 1
        y += struct.pack("<d", target)
        row = []
        while pos < len(line):
            c, j = parse_int(line, pos, j)
            pos += c + 1
            c, val = parse_float(line, pos, val)
            pos += c + 1
            row.append((val, j - min_index))
        x += struct.pack("<q", len(row))
        for v, i in row:
            x += struct.pack("<dq", v, i)
    return bytes(x), bytes(y)


def read_stream(xbytes, ybytes=None):
    """what newStreamCSRMatrix / newStreamCSRFieldMatrix + readCache deliver when the cache holds every row
    (tensor/sparse_stream.nim:95-170, 200-260): header, then per row nnz and the elements."""
    import struct

    if xbytes[:14] == b"STREAMCSRFIELD":
        n, d, nnz, nf, mx, mn = struct.unpack_from("<qqqqdd", xbytes, 14)
        pos, fielded = 14 + 48, True
    elif xbytes[:9] == b"STREAMCSR":
        n, d, nnz, mx, mn = struct.unpack_from("<qqqdd", xbytes, 9)
        nf, pos, fielded = 0, 9 + 40, False
    else:
        raise IOError("not a StreamCSR file.")
    indptr, indices, data, fields = [0], [], [], []
    for _ in range(n):
        (r,) = struct.unpack_from("<q", xbytes, pos)
        pos += 8
        for _ in range(r):
            if fielded:
                f, v, i = struct.unpack_from("<qdq", xbytes, pos)
                pos += 24
                fields.append(f)
            else:
                v, i = struct.unpack_from("<dq", xbytes, pos)
                pos += 16
            data.append(v)
            indices.append(i)
        indptr.append(len(indices))
    y = np.frombuffer(ybytes, dtype="<f8").copy() if ybytes is not None else np.zeros(n)
    return dict(indptr=np.array(indptr, dtype=np.int64), indices=np.array(indices, dtype=np.int64),
                data=np.array(data, dtype=np.float64), fields=np.array(fields, dtype=np.int64), y=y,
                n_features=d, n_fields=nf, nnz=nnz, max=mx, min=mn)


def write_stream_field(indptr, indices, fields, data, n_cols, n_fields):
    """a STREAMCSRFIELD file as convertFFMFile lays it out (dataset.nim:1202-1299: magic, header
    {nRows, nCols, nnz, nFields, max, min}, rows of {field, val, id})"""
    import struct

    n = len(indptr) - 1
    x = bytearray(b"STREAMCSRFIELD")
    x += struct.pack("<qqqqdd", n, n_cols, len(data), n_fields, float(np.max(data)) if len(data) else float("-inf"),
                     float(np.min(data)) if len(data) else float("inf"))
    for i in range(n):
        x += struct.pack("<q", indptr[i + 1] - indptr[i])
        for q in range(indptr[i], indptr[i + 1]):
            x += struct.pack("<qdq", int(fields[q]), float(data[q]), int(indices[q]))
    return bytes(x)
