"""The C++ host mirror (nimfm_amd/host/nimfm.hpp) builds against the C ABI with plain g++ and, on a GPU,
passes the reference-style checks in tests/cpp/host_mirror_test.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_host_mirror_test")


def _build():
    import __graft_entry__ as g
    g.build()
    lib = os.path.join(ROOT, "nimfm_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L", lib, "-lnimfm_hip", "-Wl,-rpath," + lib, "-o", EXE])


def test_cpp_host_mirror_builds():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_mirror_runs():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror ok" in out.stdout
