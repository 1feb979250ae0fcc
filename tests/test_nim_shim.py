"""The Nim side of the boundary (nim/*.nim) cannot be compiled here (no Nim toolchain in the image), so its FFI
declarations are held to include/nimfm_hip.h mechanically: every `proc nfm_*` must exist in the header with the same
arity, the same parameter widths in the same order and the same return type; the {.bycopy.} config objects must match
the header's structs field for field; every nfm_* call in the include files must be declared; and the overloads
SURVEY 8(b) lists must be present under the reference's own names."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NIM = os.path.join(ROOT, "nim")

C2N = {  # C parameter type (const dropped, spaces normalised) -> Nim spelling
    "int32_t": "int32", "int64_t": "int64", "double": "float64", "char*": "cstring", "void*": "pointer",
    "int32_t*": "ptr int32", "int64_t*": "ptr int64", "double*": "ptr float64", "uint64_t*": "ptr uint64",
    "nfm_ctx*": "NfmCtx", "nfm_dataset*": "NfmDataset", "nfm_model*": "NfmModel", "nfm_opt*": "NfmOpt",
    "nfm_ctx**": "ptr NfmCtx", "nfm_dataset**": "ptr NfmDataset", "nfm_model**": "ptr NfmModel", "nfm_opt**": "ptr NfmOpt",
    "nfm_model_cfg*": "ptr NfmModelCfg", "nfm_sgd_cfg*": "ptr NfmSgdCfg", "nfm_adagrad_cfg*": "ptr NfmAdaGradCfg",
    "nfm_mbpsgd_cfg*": "ptr NfmMbpsgdCfg", "double**": "ptr ptr float64", "nfm_dp*": "NfmDp", "nfm_dp**": "ptr NfmDp", "nfm_stream*": "NfmStream", "nfm_stream**": "ptr NfmStream",
}


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def header_protos():
    h = _strip_c_comments(open(os.path.join(ROOT, "include", "nimfm_hip.h")).read())
    out = {}
    for ret, name, args in re.findall(r"\b(int32_t|const char\s*\*)\s*(nfm_\w+)\s*\(([^;{}]*?)\)\s*;", h):
        params = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"\bconst\b", "", a).strip()
                m = re.match(r"(.*?)(\w+)$", a)  # type, then the parameter's name
                t = m.group(1).replace(" ", "")
                params.append(C2N[t])
        out[name] = ("cstring" if "char" in ret else "int32", params)
    return out, h


def nim_protos(path):
    src = open(path).read()
    block = src[src.index("{.push importc"):src.index("{.pop.}")]
    block = re.sub(r"##.*", "", block)
    out = {}
    for name, args, ret in re.findall(r"proc (nfm_\w+)\*?\(([^)]*)\)\s*:\s*(\w+)", block, flags=re.S):
        params, pending = [], 0
        for piece in [p.strip() for p in args.replace("\n", " ").split(",") if p.strip()]:
            if ":" in piece:
                t = " ".join(piece.split(":", 1)[1].split())
                params += [t] * (pending + 1)
                pending = 0
            else:
                pending += 1
        assert pending == 0, (name, args)
        out[name] = (ret, params)
    return out, src


def test_ffi_declarations_match_the_header():
    hdr, _ = header_protos()
    nim, _ = nim_protos(os.path.join(NIM, "nimfm_hip.nim"))
    assert len(nim) >= 35
    for name, (ret, params) in nim.items():
        assert name in hdr, "%s is not declared in include/nimfm_hip.h" % name
        assert ret == hdr[name][0], (name, ret, hdr[name][0])
        assert params == hdr[name][1], "%s: nim %s vs header %s" % (name, params, hdr[name][1])


def test_config_objects_match_the_header_structs():
    _, h = header_protos()
    _, src = nim_protos(os.path.join(NIM, "nimfm_hip.nim"))
    wid = {"int32_t": "int32", "int64_t": "int64", "double": "float64"}
    for cname, nname in (("nfm_model_cfg", "NfmModelCfg"), ("nfm_sgd_cfg", "NfmSgdCfg"), ("nfm_adagrad_cfg", "NfmAdaGradCfg"),
                         ("nfm_mbpsgd_cfg", "NfmMbpsgdCfg")):
        body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s;" % (cname, cname), h, flags=re.S).group(1)
        cfields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                t, names = decl.split(None, 1)
                cfields += [wid[t]] * len(names.split(","))
        nbody = re.search(r"%s\* \{\.bycopy\.\} = object\n((?:    .*\n)+)" % nname, src).group(1)
        nfields = []
        for line in nbody.splitlines():
            names, t = line.strip().split(":")
            nfields += [t.strip()] * len(names.split(","))
        assert nfields == cfields, (nname, nfields, cfields)


def test_include_files_call_only_declared_entry_points():
    nim, _ = nim_protos(os.path.join(NIM, "nimfm_hip.nim"))
    for f in ("hip_sgd.nim", "hip_adagrad.nim", "hip_sgd_ffm.nim", "hip_adagrad_ffm.nim", "hip_mbpsgd.nim", "nimfm_hip.nim"):
        src = re.sub(r"##.*|#.*", "", open(os.path.join(NIM, f)).read())
        for call in set(re.findall(r"\b(nfm_\w+)\(", src)):
            assert call in nim, "%s calls %s, which nimfm_hip.nim does not declare" % (f, call)


def test_reference_surface_is_overloaded_under_its_own_names():
    """fit (+ maxThreads, + callback), decisionFunction, the loaders -- under the reference's names, on
    the device dataset types; AdaGrad state through self.g_sum / self.g_norm; eps and the Huber threshold plumbed."""
    sgd = open(os.path.join(NIM, "hip_sgd.nim")).read()
    ada = open(os.path.join(NIM, "hip_adagrad.nim")).read()
    sffm = open(os.path.join(NIM, "hip_sgd_ffm.nim")).read()
    affm = open(os.path.join(NIM, "hip_adagrad_ffm.nim")).read()
    core = open(os.path.join(NIM, "nimfm_hip.nim")).read()
    for src, opt, model, ds in ((sgd, "SGD", "FactorizationMachine", "HipCSRDataset"),
                                (ada, "AdaGrad", "FactorizationMachine", "HipCSRDataset"),
                                (sffm, "SGD", "FieldAwareFactorizationMachine", "HipCSRFieldDataset"),
                                (affm, "AdaGrad", "FieldAwareFactorizationMachine", "HipCSRFieldDataset")):
        flat = " ".join(src.split())
        assert re.search(r"proc fit\*\[L\]\(self: %s\[L\], X: %s, y: seq\[float64\], \w+: %s, callback:" % (opt, ds, model), flat)
        assert re.search(r"proc fit\*\[L\]\(self: %s\[L\], X: %s, y: seq\[float64\], \w+: %s, maxThreads: int, callback:"
                         % (opt, ds, model), flat)
    assert "proc decisionFunction*(self: FactorizationMachine, X: HipCSRDataset)" in core
    assert "proc decisionFunction*(self: FieldAwareFactorizationMachine, X: HipCSRFieldDataset)" in core
    assert "proc loadSVMLightFile*(f: string, dataset: var HipCSRDataset" in core and "proc loadFFMFile*(" in core
    assert "eps: self.eps" in ada and "nfm_opt_set_state" in ada and "nfm_opt_get_state" in ada and "self.g_sum" in ada
    assert "lossParam(self.loss)" in sgd and "loss.dloss(0.0, Inf)" in core
    assert "callback(self, fm)" in sgd and "nCalls" in core


def test_max_threads_is_never_read_as_a_batch_size():
    """the maxThreads overloads take the mini-batch size from an explicit, defaulted `miniBatchSize` (and
    `syncPeriod`, `group` for the multi-GPU form); the reference's thread count only selects the mode"""
    for name in ("hip_sgd.nim", "hip_adagrad.nim", "hip_sgd_ffm.nim", "hip_adagrad_ffm.nim"):
        src = open(os.path.join(NIM, name)).read()
        assert "maxThreads >=" not in src and "maxThreads else" not in src, name
        # (round 5: followed by the rule's own defaulted knob -- touchCap for SGD, adaCross for AdaGrad)
        knob = "touchCap: float64 = 1.0" if "sgd" in name else "adaCross: float64 = 0.0"
        sig = re.search(r"maxThreads: int,\s*callback:[^=]*= nil,\s*miniBatchSize: int = defaultBatch\(\),\s*syncPeriod: int = 0, group: HipGroup = nil, "
                        + re.escape(knob) + r"\)", src)
        assert sig, name
        assert ("nfm_opt_set_touch_cap(o, touchCap)" if "sgd" in name else "adaCross)") in src, name
        assert "discard maxThreads" in src, name
