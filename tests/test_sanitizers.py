"""Sanitizer builds of the host-side C / C++ that can run without a GPU (GPU AddressSanitizer is not available on the
MI355X boxes): the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer runs its own test files (every fit,
the mini-batch rule, MBPSGD, the loaders' restatement), and parse_num.h -- the number parser shared by the ingest
kernels and the host -- runs its token corpus under the same sanitizers.  A finding aborts the process (non-zero
exit), -fno-sanitize-recover makes undefined behaviour fatal too."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc has no libasan here")
    env = dict(os.environ, NIMFM_ORACLE_VARIANT="asan", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = ["test_oracle_kernels.py", "test_oracle_sgd.py", "test_oracle_adagrad.py", "test_oracle_ffm.py", "test_oracle_mb.py",
             "test_oracle_psgd.py", "test_oracle_ingest.py", "test_oracle_metrics.py", "test_golden.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] +
                       [os.path.join(ROOT, "tests", f) for f in files], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the sanitized build really was the one loaded
    chk = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import oracle; oracle.lib(); "
                          "print(any('libnimfm_oracle_asan' in l for l in open('/proc/self/maps')))" % ROOT],
                         capture_output=True, text=True, env=env, timeout=300)
    assert chk.stdout.strip().endswith("True"), chk.stdout + chk.stderr


def test_parse_num_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "parse_num_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", exe, os.path.join(ROOT, "tests", "cpp", "parse_num_test.cpp")])
    random.seed(3)
    toks = ["0", "-0.0", "1e309", "1e-400", "4.9e-324", "123456789012345678901234567890", "nan", "inf", "", "-", ".", "e5", "1e+",
            "9007199254740993", "1.7976931348623159e308", "3:4", "7 8", "+", "-.", "1" * 400, "0." + "0" * 400 + "1", "1e" + "9" * 30]
    for _ in range(20000):
        nd = random.randint(1, 30)
        t = "".join(random.choice("0123456789.eE+-") for _ in range(nd))
        toks.append(t)
    data = ("\n".join(toks) + "\n").encode()
    for mode in ([], ["int"]):
        r = subprocess.run([exe] + mode, input=data, capture_output=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert len(r.stdout.splitlines()) == len(toks)
