#!/usr/bin/env python3
"""Writes tests/golden/ingest_*.txt (hand-written inputs in the two text formats) and
tests/golden/ingest_golden.npz (what oracle/ingest.py -- the restatement of dataset.nim:562-632,
696-790 -- reads from them).  The expected arrays were also checked by hand for the small files."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ingest  # noqa: E402

FILES = {
    # 1-based, no trailing newline (what dumpSVMLightFile writes)
    "ingest_svm_1based.txt": "1.0 1:0.5 3:-1.25 7:2\n-1.0 2:0.30000000000000004 7:1e-3\n0.0\n2.5 5:1.7976931348623157e308",
    # an index 0 makes the file 0-based; trailing newline; CRLF on one line; an empty line keeps the previous target
    "ingest_svm_0based.txt": "3 0:1 4:2.5\r\n-2 1:0.1 2:0.2 3:0.3\n\n7 9:9.007199254740993e15\n",
    # more than 19 significant digits (strtod on the host), exponents, signs
    "ingest_svm_digits.txt": "1e0 1:1.00000000000000011102230246251565404236316680908203125 2:+4.9e-324 3:-.5\n"
                             "-1E0 1:123456789012345678901234567890 4:2.2250738585072011e-308",
    "ingest_ffm_1based.txt": "1.0 1:1:0.5 2:3:-1.25 4:7:2\n0.0 3:2:0.25\n-1.0 1:1:1 4:8:1",
    "ingest_ffm_0based.txt": "1 0:0:0.5 1:3:1.5\n0 2:1:0.25 0:4:-4\n",
}


def main():
    out = {}
    for name, text in FILES.items():
        with open(os.path.join(HERE, name), "w", newline="") as f:
            f.write(text)
        r = ingest.load_ffm(text) if "ffm" in name else ingest.load_svmlight(text)
        for k, v in r.items():
            out[name + ":" + k] = np.asarray(v)
    np.savez(os.path.join(HERE, "ingest_golden.npz"), **out)
    print("wrote", len(FILES), "files and ingest_golden.npz")


if __name__ == "__main__":
    main()
