"""ctypes binding of libnimfm_hip.so (include/nimfm_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent the import of the
compute layer fails loudly.  Nothing here touches oracle/.
"""
import atexit
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NIMFM_HIP_LIB: an alternative build of the same library (kernel tuning A/B runs)
LIB_PATH = os.environ.get("NIMFM_HIP_LIB") or os.path.join(_HERE, "lib", "libnimfm_hip.so")
CSRC = os.path.join(_HERE, "csrc")

NFM_OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_FITTED, ERR_NOMEM, ERR_UNSUPPORTED = -1, -2, -3, -4, -5

TASK = {"regression": 0, "r": 0, "classification": 1, "c": 1}
KIND_FM, KIND_FFM = 0, 1
LOWER = {"explicit": 0, "augment": 1, "none": 2}
LOSS = {"squared": 0, "squared_hinge": 1, "logistic": 2, "huber": 3}
SCHED = {"constant": 0, "optimal": 1, "invscaling": 2, "pegasos": 3}
MODE = {"sequential": 0, "minibatch": 1}
REG = {"l1": 0, "l21": 1, "squaredl12": 2, "squaredl21": 3}

# every symbol include/nimfm_hip.h declares (tests/test_abi.py checks header <-> library <-> this list)
SYMBOLS = [
    "nfm_last_error", "nfm_version", "nfm_device_count",
    "nfm_ctx_create", "nfm_ctx_destroy", "nfm_ctx_synchronize",
    "nfm_ctx_timing_enable", "nfm_ctx_timing_reset", "nfm_ctx_timing_get",
    "nfm_dataset_create_csr", "nfm_dataset_create_csr_device", "nfm_dataset_set_targets", "nfm_dataset_destroy",
    "nfm_dataset_load_svmlight", "nfm_dataset_load_ffm", "nfm_dataset_parse_text", "nfm_dataset_load_stream",
    "nfm_convert_svmlight", "nfm_dataset_shape",
    "nfm_dataset_ingest_stats", "nfm_dataset_get_targets", "nfm_dataset_get_csr",
    "nfm_model_create", "nfm_model_shape", "nfm_model_set_params", "nfm_model_get_params",
    "nfm_decision_function", "nfm_decision_function_device", "nfm_score", "nfm_metrics", "nfm_model_sqnorms", "nfm_model_device_buffers",
    "nfm_model_destroy",
    "nfm_sgd_create", "nfm_adagrad_create", "nfm_mbpsgd_create", "nfm_opt_predict_all_with_grad", "nfm_opt_set_it", "nfm_opt_get_it", "nfm_opt_get_state",
    "nfm_opt_set_state", "nfm_opt_epoch", "nfm_opt_finalize", "nfm_opt_device_state", "nfm_opt_destroy",
    "nfm_rng_randomize", "nfm_rng_random_normal", "nfm_rng_shuffle",
    "nfm_dp_unique_id", "nfm_dp_create", "nfm_dp_create_local", "nfm_dp_info", "nfm_dp_destroy", "nfm_opt_set_dp", "nfm_opt_set_dp_combine", "nfm_opt_set_touch_cap", "nfm_opt_set_ada_cross",
    "nfm_opt_set_shuffle", "nfm_opt_get_perm", "nfm_opt_announce_perm",
    "nfm_stream_open", "nfm_stream_shape", "nfm_stream_load_rows", "nfm_stream_prefetch_rows", "nfm_stream_close",
]


class ModelCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("task", C.c_int32), ("degree", C.c_int32), ("n_components", C.c_int32),
                ("fit_lower", C.c_int32), ("fit_intercept", C.c_int32), ("fit_linear", C.c_int32),
                ("reserved", C.c_int32), ("n_features", C.c_int64), ("n_fields", C.c_int64)]


class SGDCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("power", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32), ("scheduling", C.c_int32),
                ("mode", C.c_int32), ("reserved", C.c_int32), ("batch", C.c_int64)]


class MBPSGDCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("gamma", C.c_double), ("power", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32),
                ("scheduling", C.c_int32), ("reg", C.c_int32), ("reg_transpose", C.c_int32), ("batch", C.c_int64)]


class AdaGradCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("eps", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32), ("mode", C.c_int32),
                ("track_viol", C.c_int32), ("reserved", C.c_int32), ("batch", C.c_int64)]


class NfmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libnimfm_hip error %d: %s" % (code, msg))
        self.code = code


class NotFittedError(NfmError):
    """model/fm_base.nim:10-15"""


def build(force=False):
    """Compile libnimfm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


_lib = None
alive = True  # cleared at interpreter exit: object finalizers then leave teardown to the OS


def _at_exit():
    global alive
    alive = False


atexit.register(_at_exit)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libnimfm_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "-- there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError("libnimfm_hip.so lacks symbols: %s" % missing)
    L.nfm_last_error.restype = C.c_char_p
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    pp = C.POINTER(C.c_void_p)
    sig = {
        "nfm_version": [],
        "nfm_device_count": [C.POINTER(i32)],
        "nfm_ctx_create": [i32, vp, pp],
        "nfm_ctx_destroy": [vp],
        "nfm_ctx_synchronize": [vp],
        "nfm_ctx_timing_enable": [vp, i32],
        "nfm_ctx_timing_reset": [vp],
        "nfm_ctx_timing_get": [vp, C.c_char_p, C.POINTER(i64), C.POINTER(dbl)],
        "nfm_dataset_create_csr": [vp, i64, i64, vp, vp, vp, vp, i64, vp, pp],
        "nfm_dataset_create_csr_device": [vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, pp],
        "nfm_dataset_set_targets": [vp, vp],
        "nfm_dataset_destroy": [vp],
        "nfm_dataset_load_svmlight": [vp, C.c_char_p, i64, pp],
        "nfm_dataset_load_ffm": [vp, C.c_char_p, i64, i64, pp],
        "nfm_dataset_parse_text": [vp, C.c_char_p, i64, i32, i64, i64, pp],
        "nfm_dataset_load_stream": [vp, C.c_char_p, C.c_char_p, pp],
        "nfm_convert_svmlight": [vp, C.c_char_p, C.c_char_p, C.c_char_p],
        "nfm_dataset_shape": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
        "nfm_dataset_ingest_stats": [vp, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_dataset_get_targets": [vp, vp],
        "nfm_dataset_get_csr": [vp, vp, vp, vp, vp],
        "nfm_model_create": [vp, C.POINTER(ModelCfg), pp],
        "nfm_model_shape": [vp, C.POINTER(i32), C.POINTER(i32)],
        "nfm_model_set_params": [vp, vp, vp, dbl, vp],
        "nfm_model_get_params": [vp, vp, vp, C.POINTER(dbl)],
        "nfm_decision_function": [vp, vp, vp],
        "nfm_decision_function_device": [vp, vp, vp],
        "nfm_model_sqnorms": [vp, C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_score": [vp, vp, C.POINTER(dbl)],
        "nfm_metrics": [vp, vp, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_model_device_buffers": [vp, pp, C.POINTER(i64), pp, C.POINTER(i64), pp, C.POINTER(i64)],
        "nfm_model_destroy": [vp],
        "nfm_sgd_create": [vp, C.POINTER(SGDCfg), pp],
        "nfm_adagrad_create": [vp, C.POINTER(AdaGradCfg), pp],
        "nfm_mbpsgd_create": [vp, C.POINTER(MBPSGDCfg), pp],
        "nfm_opt_predict_all_with_grad": [vp, vp, vp, vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_opt_set_it": [vp, i64],
        "nfm_opt_get_it": [vp, C.POINTER(i64)],
        "nfm_opt_get_state": [vp, vp, vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_opt_set_state": [vp, vp, vp, vp, vp, dbl, dbl],
        "nfm_opt_epoch": [vp, vp, vp, i64, i64, C.POINTER(dbl), C.POINTER(dbl)],
        "nfm_opt_finalize": [vp],
        "nfm_opt_device_state": [vp, pp, pp, C.POINTER(i64), pp, pp, C.POINTER(i64), pp],
        "nfm_opt_destroy": [vp],
        "nfm_rng_randomize": [i64, vp],
        "nfm_rng_random_normal": [vp, i64, dbl, dbl, vp],
        "nfm_rng_shuffle": [vp, vp, i64],
        "nfm_dp_unique_id": [vp],
        "nfm_dp_create": [vp, vp, i32, i32, pp],
        "nfm_dp_create_local": [vp, i32, vp],
        "nfm_dp_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)],
        "nfm_dp_destroy": [vp],
        "nfm_opt_set_dp": [vp, vp, i64, i32],
        "nfm_opt_set_dp_combine": [vp, i32],
        "nfm_opt_set_touch_cap": [vp, C.c_double],
        "nfm_opt_set_ada_cross": [vp, C.c_double],
        "nfm_opt_set_shuffle": [vp, i64],
        "nfm_opt_get_perm": [vp, vp, i64],
        "nfm_opt_announce_perm": [vp, vp, i64, i64],
        "nfm_stream_open": [vp, C.c_char_p, C.c_char_p, pp],
        "nfm_stream_shape": [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)],
        "nfm_stream_load_rows": [vp, i64, i64, pp],
        "nfm_stream_prefetch_rows": [vp, i64, i64],
        "nfm_stream_close": [vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int32
    _lib = L
    return L


def check(rc):
    if rc == NFM_OK:
        return
    msg = lib().nfm_last_error().decode("utf-8", "replace")
    if rc == ERR_NOT_FITTED:
        raise NotFittedError(rc, msg)
    if rc == ERR_INVALID:
        raise ValueError(msg)  # the reference raises ValueError for these
    raise NfmError(rc, msg)
