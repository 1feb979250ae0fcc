"""CPU-side checks of the drop-in boundary: libnimfm_hip.so loads without a GPU, exports exactly the
symbols include/nimfm_hip.h declares, and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nimfm_hip.h")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    import nimfm_amd
    return nimfm_amd


def header_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nfm_[a-z0-9_]+)\s*\(", txt)))


def test_header_library_binding_agree(built):
    from nimfm_amd import _capi
    hdr = header_symbols()
    assert hdr == sorted(_capi.SYMBOLS), set(hdr) ^ set(_capi.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("nfm_"))
    assert exported == hdr, set(exported) ^ set(hdr)


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "nimfm_hip.h"\nint main(void){nfm_model_cfg c; nfm_sgd_cfg s; nfm_adagrad_cfg a; '
                   '(void)c;(void)s;(void)a; return sizeof(c) == 48 && sizeof(s) == 72 && sizeof(a) == 72 ? 0 : 1;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_struct_layouts_match_ctypes(built):
    from nimfm_amd import _capi
    assert C.sizeof(_capi.ModelCfg) == 48 and C.sizeof(_capi.SGDCfg) == 72 and C.sizeof(_capi.AdaGradCfg) == 72


def test_no_cpu_fallback(built):
    """Without a usable HIP device every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nimfm_amd import _capi
    L = _capi.lib()
    n = C.c_int32(-1)
    assert L.nfm_device_count(C.byref(n)) == 0 and n.value == 0
    h = C.c_void_p()
    assert L.nfm_ctx_create(0, None, C.byref(h)) == _capi.ERR_HIP
    assert b"no CPU fallback" in L.nfm_last_error()
    with pytest.raises(_capi.NfmError):
        built.Context(0)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under nimfm_amd/ (the product) or tools/ may reference it."""
    import itertools
    for dirpath, _, files in itertools.chain(os.walk(os.path.join(ROOT, "nimfm_amd")), os.walk(os.path.join(ROOT, "tools"))):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b|^\s*import\s[^\n#]*\boracle\b", txt, flags=re.M), os.path.join(dirpath, f)
                assert "nimfm_oracle" not in txt, os.path.join(dirpath, f)


def test_library_resolves_every_symbol_eagerly():
    """dlopen with RTLD_NOW: a kernel's host stub left undefined by a split translation unit shows up here, on CPU"""
    import ctypes

    from nimfm_amd import _capi as capi

    path = os.environ.get("NIMFM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(capi.__file__)), "lib", "libnimfm_hip.so")
    ctypes.CDLL(path, mode=os.RTLD_NOW)
