"""Host-side check of nimfm_amd/csrc/parse_num.h (the number parser the ingest kernels run on the GPU,
compiled here for the CPU from the same source): parse_float must be correctly rounded -- Python's
float() is -- on hand-picked hard cases, random decimal strings and repr() of random doubles (what
the reference's dumpSVMLightFile writes, dataset.nim:793-805); parse_int must follow Nim's
parseutils.parseInt (dataset.nim:577)."""
import os
import random
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "parse_num_test")


@pytest.fixture(scope="module")
def exe():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "parse_num_test.cpp")])
    return EXE


HARD = ["0", "-0.0", "1", "1.5", "0.1", "0.30000000000000004", "1e22", "1e23", "9007199254740993", "9007199254740992.5",
        "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324", "1.7976931348623157e308",
        "1.7976931348623159e308", "1e309", "1e-400", "123456789012345678901234567890", "0.000001", "5e-324",
        "2.2250738585072014e-308", "2.2250738585072011e-308",
        "1.00000000000000011102230246251565404236316680908203125",
        "1.00000000000000011102230246251565404236316680908203124",
        "1.00000000000000011102230246251565404236316680908203126", ".5", "5.", "+3.25", "1E5", "1e+5", "1e-5",
        "8.41e21", "nan", "inf", "-inf", "Infinity", "abc", "", "-", ".", "e5", "1e", "1e+", "12ab", "3:4", "7 8"]


def reference(t):
    m = re.match(r"[+-]?(nan|inf(inity)?)", t, re.I)
    if m:
        return len(m.group(0)), float(m.group(0))
    m = re.match(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", t)
    if not m:
        return 0, 0.0
    return len(m.group(0)), float(m.group(0))


def test_parse_float_correctly_rounded(exe):
    random.seed(1)
    toks = list(HARD)
    for _ in range(120000):
        nd = random.randint(1, 22)
        digs = "".join(random.choice("0123456789") for _ in range(nd))
        kind = random.random()
        p = random.randint(0, nd)
        t = digs if kind < 0.3 else digs[:p] + "." + digs[p:]
        if kind >= 0.7:
            t += "e" + str(random.randint(-330, 310))
        toks.append(("-" if random.random() < 0.3 else "") + t)
    for _ in range(80000):
        x = struct.unpack("<d", struct.pack("<Q", random.getrandbits(64)))[0]
        if x == x and abs(x) != float("inf"):
            toks.append(repr(x))
    for _ in range(50000):
        toks.append(repr(random.uniform(-1, 1)))
    out = subprocess.run([exe], input="\n".join(toks) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    flagged = 0
    for t, o in zip(toks, out):
        c, fl, bits = o.split()
        rc, rv = reference(t)
        assert int(c) == rc, (t, o)
        if int(fl):  # > 19 significant digits where the truncation matters: the loader re-reads these with strtod
            flagged += 1
            assert len(re.sub(r"[^0-9]", "", t.split("e")[0]).lstrip("0")) > 19, t
            continue
        if rc:
            got = struct.unpack("<d", struct.pack("<Q", int(bits, 16)))[0]
            assert struct.pack("<d", got) == struct.pack("<d", rv) or (got != got and rv != rv), (t, o, rv.hex())
    assert flagged < 0.01 * len(toks)


def test_parse_int(exe):
    toks = ["0", "12", "-7", "+5", "007", "12:3", "x", "", "-", "9223372036854775807", "3.5"]
    out = subprocess.run([exe, "int"], input="\n".join(toks) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
    for t, o in zip(toks, out):
        m = re.match(r"[+-]?\d+", t)
        c, v = o.split()
        assert int(c) == (len(m.group(0)) if m else 0), (t, o)
        if m:
            assert int(v) == int(m.group(0)), (t, o)
