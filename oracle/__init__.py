"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  "parity unpinned" against reference-run outputs (no Nim
toolchain; the reference's tests hold no literal vectors): see nimfm_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnimfm_oracle.so")

LOSS = {"squared": 0, "squared_hinge": 1, "logistic": 2, "huber": 3}
SCHED = {"constant": 0, "optimal": 1, "invscaling": 2, "pegasos": 3}
LOWER = {"explicit": 0, "augment": 1, "none": 2}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("nimfm_oracle.c", "nimfm_slow.c", "nimfm_mb.c", "nimfm_psgd.c", "nimfm_ingest.c",
                                             "nimfm_jagged.c", "nimfm_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class CSR(C.Structure):
    _fields_ = [("n", C.c_int64), ("d", C.c_int64), ("indptr", C.c_void_p), ("indices", C.c_void_p),
                ("data", C.c_void_p), ("fields", C.c_void_p), ("n_fields", C.c_int64)]


class SGDCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("power", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32),
                ("scheduling", C.c_int32), ("fit_linear", C.c_int32), ("fit_intercept", C.c_int32)]


class AdaCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("eps", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32),
                ("fit_linear", C.c_int32), ("fit_intercept", C.c_int32), ("pad_", C.c_int32)]


class PSGDCfg(C.Structure):
    _fields_ = [("eta0", C.c_double), ("alpha0", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("gamma", C.c_double), ("power", C.c_double), ("loss_param", C.c_double), ("loss", C.c_int32),
                ("scheduling", C.c_int32), ("fit_linear", C.c_int32), ("fit_intercept", C.c_int32),
                ("reg", C.c_int32), ("reg_transpose", C.c_int32)]


_lib = None
_libs = {}


def _load(path):
    L = C.CDLL(path)
    for name in ("orc_loss", "orc_dloss", "orc_get_eta", "orc_expit", "orc_rmse", "orc_accuracy_sign",
                 "orc_regularization", "slow_anova", "orc_reg_eval"):
        getattr(L, name).restype = C.c_double
    L.orc_loss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_dloss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_get_eta.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int64]
    L.orc_expit.argtypes = [C.c_double]
    return L


class variant:
    """`with oracle.variant("timing"):` -- the same C sources built -O3 -march=native on THIS machine (oracle/Makefile
    target `timing`), for bench.py's cpu_baseline leg only; parity tests use the default -O2 -ffp-contract=off build."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _lib
        self.prev = _lib
        if self.name in ("timing", "fma"):  # ("fma": contraction on -- tests/fuzz_mb.py's conditioning yardstick, never a parity build)
            if self.name not in _libs:
                so = os.path.join(_HERE, "_build", "libnimfm_oracle_%s.so" % self.name)
                if os.path.exists(so):
                    os.remove(so)  # -march=native: never reuse a build made on another host
                subprocess.check_call(["make", "-C", _HERE, "-s", self.name])
                _libs[self.name] = _load(so)
            _lib = _libs[self.name]
        else:
            _lib = None
            lib()
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


def epoch_seconds(n):
    """wall-clock seconds of the first n epochs of the last *_fit call of the active build"""
    arr = (C.c_double * 64).in_dll(lib(), "orc_epoch_seconds")
    return [arr[i] for i in range(min(n, 64))]


def lib():
    global _lib
    if _lib is None:
        if os.environ.get("NIMFM_ORACLE_VARIANT") == "asan":
            # the same sources under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile target `asan`); the process
            # must have been started with LD_PRELOAD=libasan.so (tests/test_sanitizers.py does that for the oracle tests)
            so = os.path.join(_HERE, "_build", "libnimfm_oracle_asan.so")
            subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
            _lib = _load(so)
            return _lib
        build()
        _lib = C.CDLL(_SO)
        for name in ("orc_loss", "orc_dloss", "orc_get_eta", "orc_expit", "orc_rmse", "orc_accuracy_sign",
                     "orc_regularization", "slow_anova", "orc_reg_eval"):
            getattr(_lib, name).restype = C.c_double
        _lib.orc_loss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.orc_dloss.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        _lib.orc_get_eta.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int64]
        _lib.orc_expit.argtypes = [C.c_double]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Dataset:
    """Holds a CSR matrix in the reference's widths (int64 indices, fp64 values)."""

    def __init__(self, indptr, indices, data, n, d, fields=None, n_fields=0):
        self.indptr, self.indices, self.data = i64(indptr), i64(indices), f64(data)
        self.fields = None if fields is None else i64(fields)
        self.n, self.d, self.n_fields = int(n), int(d), int(n_fields)
        self.c = CSR(self.n, self.d, _p(self.indptr), _p(self.indices), _p(self.data), _p(self.fields),
                     self.n_fields)

    @staticmethod
    def from_dense(Xd, field_of=None, n_fields=0):
        Xd = f64(Xd)
        n, d = Xd.shape
        mask = Xd != 0.0
        indptr = np.concatenate([[0], np.cumsum(mask.sum(1))])
        rows, cols = np.nonzero(mask)
        fields = None if field_of is None else np.asarray(field_of)[cols]
        return Dataset(indptr, cols, Xd[rows, cols], n, d, fields, n_fields)

    def dense(self):
        Xd = np.zeros((self.n, self.d))
        for i in range(self.n):
            s, e = self.indptr[i], self.indptr[i + 1]
            Xd[i, self.indices[s:e]] = self.data[s:e]
        return Xd


def sgd_cfg(eta0=0.01, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared", scheduling="optimal", power=1.0,
            fit_linear=True, fit_intercept=True, loss_param=1.0):
    return SGDCfg(eta0, alpha0, alpha, beta, power, loss_param, LOSS[loss], SCHED[scheduling],
                  int(fit_linear), int(fit_intercept))


def adagrad_cfg(eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared", eps=1e-10, fit_linear=True,
                fit_intercept=True, loss_param=1.0):
    return AdaCfg(eta0, alpha0, alpha, beta, eps, loss_param, LOSS[loss], int(fit_linear), int(fit_intercept), 0)


def n_orders(degree, fit_lower):
    return lib().orc_n_orders(degree, LOWER[fit_lower])


def n_augments(degree, fit_lower, fit_linear):
    return lib().orc_n_augments(degree, LOWER[fit_lower], int(fit_linear))


class AdaState:
    def __init__(self, n_blocks, da, k, d):
        self.gsum_P = np.zeros((n_blocks, da, k))
        self.gnorm_P = np.zeros((n_blocks, da, k))
        self.gsum_w = np.zeros(d)
        self.gnorm_w = np.zeros(d)
        self.gsum_b = C.c_double(0.0)
        self.gnorm_b = C.c_double(0.0)

    def args(self):
        return (_p(self.gsum_P), _p(self.gnorm_P), _p(self.gsum_w), _p(self.gnorm_w), C.byref(self.gsum_b),
                C.byref(self.gnorm_b))


def _perms(perms):
    return None if perms is None else i64(perms)


def fm_decision_function(X, degree, P, w, intercept, n_aug=0, lams=None):
    O, k, da = P.shape
    out = np.zeros(X.n)
    lams = np.ones(k) if lams is None else f64(lams)
    P, w = f64(P), f64(w)
    rc = lib().orc_fm_decision_function(C.byref(X.c), degree, k, O, n_aug, _p(P), _p(lams), _p(w),
                                        C.c_double(intercept), _p(out))
    assert rc == 0
    return out


def fm_sgd_fit(X, y, degree, P, w, intercept, cfg, max_iter, n_aug=0, tol=0.0, perms=None, it=1,
               hogwild_threads=0):
    """Runs in place on copies; returns (P, w, intercept, it, epoch_loss, epoch_viol, n_run)."""
    P, w, y = f64(P).copy(), f64(w).copy(), f64(y)
    O, k, da = P.shape
    b, itc, nrun = C.c_double(intercept), C.c_int64(it), C.c_int(0)
    el, ev = np.zeros(max_iter), np.zeros(max_iter)
    pm = _perms(perms)
    if hogwild_threads:
        rc = lib().orc_fm_sgd_fit_hogwild(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                                          C.byref(cfg), max_iter, C.c_double(tol), _p(pm), C.byref(itc),
                                          hogwild_threads, _p(el), _p(ev), C.byref(nrun))
    else:
        rc = lib().orc_fm_sgd_fit(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                                  C.byref(cfg), max_iter, C.c_double(tol), _p(pm), C.byref(itc), _p(el),
                                  _p(ev), C.byref(nrun))
    assert rc == 0
    return P, w, b.value, itc.value, el, ev, nrun.value


def fm_sgd_fit_jagged(X, y, P, w, intercept, cfg, max_iter, perms=None, it=1, threads=1):
    """nimfm_jagged.c: the degree-2, one-order SGD fit on the reference's jagged (seq-of-seq) storage; threads > 1 = the
    Hogwild driver.  Returns (P, w, intercept, it, epoch_loss, epoch_viol)."""
    P, w, y = f64(P).copy(), f64(w).copy(), f64(y)
    O, k, da = P.shape
    assert O == 1 and da == X.d
    b, itc = C.c_double(intercept), C.c_int64(it)
    el, ev = np.zeros(max_iter), np.zeros(max_iter)
    pm = _perms(perms)
    rc = lib().orc_fm_sgd_fit_jagged(C.byref(X.c), _p(y), k, _p(P), _p(w), C.byref(b), C.byref(cfg), max_iter, _p(pm),
                                     C.byref(itc), int(threads), _p(el), _p(ev))
    assert rc == 0
    return P, w, b.value, itc.value, el, ev


def fm_adagrad_fit(X, y, degree, P, w, intercept, cfg, max_iter, n_aug=0, tol=0.0, perms=None, it=1, state=None):
    P, w, y = f64(P).copy(), f64(w).copy(), f64(y)
    O, k, da = P.shape
    st = state if state is not None else AdaState(O, da, k, X.d)
    b, itc, nrun = C.c_double(intercept), C.c_int64(it), C.c_int(0)
    el, ev = np.zeros(max_iter), np.zeros(max_iter)
    pm = _perms(perms)
    rc = lib().orc_fm_adagrad_fit(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                                  C.byref(cfg), max_iter, C.c_double(tol), _p(pm), C.byref(itc), *st.args(),
                                  _p(el), _p(ev), C.byref(nrun))
    assert rc == 0
    return P, w, b.value, itc.value, el, ev, nrun.value, st


def ffm_decision_function(X, P, w, intercept):
    F, d, k = P.shape
    out = np.zeros(X.n)
    P, w = f64(P), f64(w)
    rc = lib().orc_ffm_decision_function(C.byref(X.c), k, _p(P), _p(w), C.c_double(intercept), _p(out))
    assert rc == 0
    return out


def ffm_sgd_fit(X, y, P, w, intercept, cfg, max_iter, tol=0.0, perms=None, it=1):
    P, w, y = f64(P).copy(), f64(w).copy(), f64(y)
    F, d, k = P.shape
    b, itc, nrun = C.c_double(intercept), C.c_int64(it), C.c_int(0)
    el, ev = np.zeros(max_iter), np.zeros(max_iter)
    pm = _perms(perms)
    rc = lib().orc_ffm_sgd_fit(C.byref(X.c), _p(y), k, _p(P), _p(w), C.byref(b), C.byref(cfg), max_iter,
                               C.c_double(tol), _p(pm), C.byref(itc), _p(el), _p(ev), C.byref(nrun))
    assert rc == 0
    return P, w, b.value, itc.value, el, ev, nrun.value


def ffm_adagrad_fit(X, y, P, w, intercept, cfg, max_iter, tol=0.0, perms=None, it=1, state=None):
    P, w, y = f64(P).copy(), f64(w).copy(), f64(y)
    F, d, k = P.shape
    st = state if state is not None else AdaState(F, d, k, d)
    b, itc, nrun = C.c_double(intercept), C.c_int64(it), C.c_int(0)
    el, ev = np.zeros(max_iter), np.zeros(max_iter)
    pm = _perms(perms)
    rc = lib().orc_ffm_adagrad_fit(C.byref(X.c), _p(y), k, _p(P), _p(w), C.byref(b), C.byref(cfg), max_iter,
                                   C.c_double(tol), _p(pm), C.byref(itc), *st.args(), _p(el), _p(ev),
                                   C.byref(nrun))
    assert rc == 0
    return P, w, b.value, itc.value, el, ev, nrun.value, st


# ---- brute force (reference tests' slow models) ----
def slow_anova(xrow, prow, m, degree):
    xrow, prow = f64(xrow), f64(prow)
    return lib().slow_anova(_p(xrow), _p(prow), len(xrow), m, degree)


def slow_fm_decision_function(Xd, degree, P, w, intercept, n_aug=0):
    Xd, P, w = f64(Xd), f64(P), f64(w)
    n, d = Xd.shape
    O, k, da = P.shape
    out = np.zeros(n)
    lib().slow_fm_decision_function(_p(Xd), C.c_int64(n), d, degree, k, O, n_aug, _p(P), _p(w),
                                    C.c_double(intercept), _p(out))
    return out


def slow_fm_sgd_fit(Xd, y, degree, P, w, intercept, cfg, max_iter, n_aug=0, perms=None, it=1):
    Xd, y, P, w = f64(Xd), f64(y), f64(P).copy(), f64(w).copy()
    n, d = Xd.shape
    O, k, da = P.shape
    b, itc = C.c_double(intercept), C.c_int64(it)
    pm = _perms(perms)
    lib().slow_fm_sgd_fit(_p(Xd), C.c_int64(n), d, _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                          C.byref(cfg), max_iter, _p(pm), C.byref(itc))
    return P, w, b.value, itc.value


def slow_fm_adagrad_fit(Xd, y, degree, P, w, intercept, cfg, max_iter, n_aug=0, perms=None, it=1):
    Xd, y, P, w = f64(Xd), f64(y), f64(P).copy(), f64(w).copy()
    n, d = Xd.shape
    O, k, da = P.shape
    b, itc = C.c_double(intercept), C.c_int64(it)
    pm = _perms(perms)
    lib().slow_fm_adagrad_fit(_p(Xd), C.c_int64(n), d, _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                              C.byref(cfg), max_iter, _p(pm), C.byref(itc))
    return P, w, b.value, itc.value


def slow_ffm_decision_function(Xd, field_of, n_fields, P, w, intercept):
    Xd, P, w, field_of = f64(Xd), f64(P), f64(w), i64(field_of)
    n, d = Xd.shape
    k = P.shape[2]
    out = np.zeros(n)
    lib().slow_ffm_decision_function(_p(Xd), C.c_int64(n), d, _p(field_of), n_fields, k, _p(P), _p(w),
                                     C.c_double(intercept), _p(out))
    return out


def slow_ffm_sgd_fit(Xd, field_of, n_fields, y, P, w, intercept, cfg, max_iter, perms=None, it=1):
    Xd, y, P, w, field_of = f64(Xd), f64(y), f64(P).copy(), f64(w).copy(), i64(field_of)
    n, d = Xd.shape
    k = P.shape[2]
    b, itc = C.c_double(intercept), C.c_int64(it)
    pm = _perms(perms)
    lib().slow_ffm_sgd_fit(_p(Xd), C.c_int64(n), d, _p(field_of), n_fields, _p(y), k, _p(P), _p(w),
                           C.byref(b), C.byref(cfg), max_iter, _p(pm), C.byref(itc))
    return P, w, b.value, itc.value


def slow_ffm_adagrad_fit(Xd, field_of, n_fields, y, P, w, intercept, cfg, max_iter, perms=None, it=1):
    Xd, y, P, w, field_of = f64(Xd), f64(y), f64(P).copy(), f64(w).copy(), i64(field_of)
    n, d = Xd.shape
    k = P.shape[2]
    b, itc = C.c_double(intercept), C.c_int64(it)
    pm = _perms(perms)
    lib().slow_ffm_adagrad_fit(_p(Xd), C.c_int64(n), d, _p(field_of), n_fields, _p(y), k, _p(P), _p(w),
                               C.byref(b), C.byref(cfg), max_iter, _p(pm), C.byref(itc))
    return P, w, b.value, itc.value


# ---- this repository's mini-batch rule (see nimfm_mb.c) ----
class _touch_cap:
    """the SGD mini-batch rule's touch cap for the calls inside (oracle/nimfm_mb.c: orc_mb_touch_cap; 1 = the mean)"""

    def __init__(self, cap):
        self.cap = float(cap)

    def __enter__(self):
        self.var = C.c_double.in_dll(lib(), "orc_mb_touch_cap")
        self.old, self.var.value = self.var.value, self.cap

    def __exit__(self, *a):
        self.var.value = self.old


class _ada_cross:
    """the AdaGrad mini-batch rule's cross-product weight for the calls inside (oracle/nimfm_mb.c: orc_mb_ada_cross; 0 = off)"""

    def __init__(self, g):
        self.g = float(g)

    def __enter__(self):
        self.var = C.c_double.in_dll(lib(), "orc_mb_ada_cross")
        self.old, self.var.value = self.var.value, self.g

    def __exit__(self, *a):
        self.var.value = self.old


def fm_sgd_epoch_mb(X, y, degree, P, w, intercept, cfg, batch, n_aug=0, perm=None, begin=0, end=None, it=1, touch_cap=1.0):
    """In place on P (model layout) and w; returns (intercept, it, loss_sum, viol_sum)."""
    if touch_cap != 1.0:
        with _touch_cap(touch_cap):
            return fm_sgd_epoch_mb(X, y, degree, P, w, intercept, cfg, batch, n_aug, perm, begin, end, it)
    O, k, da = P.shape
    assert P.dtype == np.float64 and P.flags.c_contiguous and w.flags.c_contiguous
    y = f64(y)
    end = X.n if end is None else end
    b, itc, ls, vs = C.c_double(intercept), C.c_int64(it), C.c_double(0), C.c_double(0)
    pm = _perms(perm)
    rc = lib().orc_fm_sgd_epoch_mb(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                                   C.byref(cfg), _p(pm), C.c_int64(begin), C.c_int64(end), C.c_int64(batch),
                                   C.byref(itc), C.byref(ls), C.byref(vs))
    assert rc == 0
    return b.value, itc.value, ls.value, vs.value


def fm_adagrad_epoch_mb(X, y, degree, P, w, intercept, cfg, batch, state, n_aug=0, perm=None, begin=0,
                        end=None, it=1, ada_cross=0.0):
    if ada_cross != 0.0:
        with _ada_cross(ada_cross):
            return fm_adagrad_epoch_mb(X, y, degree, P, w, intercept, cfg, batch, state, n_aug, perm, begin, end, it)
    O, k, da = P.shape
    assert P.dtype == np.float64 and P.flags.c_contiguous and w.flags.c_contiguous
    y = f64(y)
    end = X.n if end is None else end
    b, itc, ls, vs = C.c_double(intercept), C.c_int64(it), C.c_double(0), C.c_double(0)
    pm = _perms(perm)
    rc = lib().orc_fm_adagrad_epoch_mb(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b),
                                       C.byref(cfg), _p(pm), C.c_int64(begin), C.c_int64(end),
                                       C.c_int64(batch), C.byref(itc), *state.args(), C.byref(ls),
                                       C.byref(vs))
    assert rc == 0
    return b.value, itc.value, ls.value, vs.value


def fm_adagrad_finalize(degree, P, w, intercept, cfg, it, state, n_aug=0):
    O, k, da = P.shape
    b = C.c_double(intercept)
    rc = lib().orc_fm_adagrad_finalize(degree, k, O, n_aug, C.c_int64(da - n_aug), _p(P), _p(w), C.byref(b),
                                       C.byref(cfg), C.c_int64(it), *state.args())
    assert rc == 0
    return b.value


def ffm_sgd_epoch_mb(X, y, P, w, intercept, cfg, batch, perm=None, begin=0, end=None, it=1, touch_cap=1.0):
    if touch_cap != 1.0:
        with _touch_cap(touch_cap):
            return ffm_sgd_epoch_mb(X, y, P, w, intercept, cfg, batch, perm, begin, end, it)
    F, d, k = P.shape
    y = f64(y)
    end = X.n if end is None else end
    b, itc, ls, vs = C.c_double(intercept), C.c_int64(it), C.c_double(0), C.c_double(0)
    pm = _perms(perm)
    rc = lib().orc_ffm_sgd_epoch_mb(C.byref(X.c), _p(y), k, _p(P), _p(w), C.byref(b), C.byref(cfg), _p(pm),
                                    C.c_int64(begin), C.c_int64(end), C.c_int64(batch), C.byref(itc),
                                    C.byref(ls), C.byref(vs))
    assert rc == 0
    return b.value, itc.value, ls.value, vs.value


def ffm_adagrad_epoch_mb(X, y, P, w, intercept, cfg, batch, state, perm=None, begin=0, end=None, it=1, ada_cross=0.0):
    if ada_cross != 0.0:
        with _ada_cross(ada_cross):
            return ffm_adagrad_epoch_mb(X, y, P, w, intercept, cfg, batch, state, perm, begin, end, it)
    F, d, k = P.shape
    y = f64(y)
    end = X.n if end is None else end
    b, itc, ls, vs = C.c_double(intercept), C.c_int64(it), C.c_double(0), C.c_double(0)
    pm = _perms(perm)
    rc = lib().orc_ffm_adagrad_epoch_mb(C.byref(X.c), _p(y), k, _p(P), _p(w), C.byref(b), C.byref(cfg),
                                        _p(pm), C.c_int64(begin), C.c_int64(end), C.c_int64(batch),
                                        C.byref(itc), *state.args(), C.byref(ls), C.byref(vs))
    assert rc == 0
    return b.value, itc.value, ls.value, vs.value


def ffm_adagrad_finalize(P, w, intercept, cfg, it, state):
    F, d, k = P.shape
    b = C.c_double(intercept)
    rc = lib().orc_adagrad_finalize(F, k, C.c_int64(d), C.c_int64(d), _p(P), _p(w), C.byref(b), C.byref(cfg),
                                    C.c_int64(it), *state.args())
    assert rc == 0
    return b.value


def svmlight_load_c(text):
    """nimfm_ingest.c (the C restatement of dataset.nim:562-613): -> dict like oracle.ingest.load_svmlight."""
    if isinstance(text, str):
        text = text.encode()
    L = lib()
    n, nnz, d, off = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    L.orc_svmlight_scan.restype = C.c_int
    rc = L.orc_svmlight_scan(text, C.c_int64(len(text)), C.byref(n), C.byref(nnz), C.byref(d), C.byref(off))
    if rc != 0:
        raise ValueError("Negative index is included.")
    indptr = np.zeros(n.value + 1, dtype=np.int64)
    indices = np.zeros(nnz.value, dtype=np.int64)
    data = np.zeros(nnz.value)
    y = np.zeros(n.value)
    L.orc_svmlight_fill.restype = None
    L.orc_svmlight_fill(text, C.c_int64(len(text)), off, _p(indptr), _p(indices), _p(data), _p(y))
    return dict(indptr=indptr, indices=indices, data=data, y=y, n_features=d.value, offset=off.value)


# ---- mini-batch proximal SGD, SURVEY.md 8(f) rank 3 (nimfm_psgd.c) ----
REG = {"l1": 0, "l21": 1, "squaredl12": 2, "squaredl21": 3}


def psgd_cfg(eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-4, gamma=1e-4, loss="squared", reg="squaredl12",
             transpose=None, scheduling="optimal", power=1.0, fit_linear=True, fit_intercept=True, loss_param=1.0):
    """newMBPSGD's defaults (optimizer/minibatch_psgd.nim:24-29); transpose defaults as the regularizers'
    constructors do (squaredl12.nim:85: true, squaredl21.nim:15: false)"""
    if transpose is None:
        transpose = reg == "squaredl12"
    return PSGDCfg(eta0, alpha0, alpha, beta, gamma, power, loss_param, LOSS[loss], SCHED[scheduling], int(fit_linear),
                   int(fit_intercept), REG[reg], int(transpose))


def prox_squaredl12(p, lam, seed=1):
    p = np.array(p, dtype=np.float64)
    rng = C.c_uint64(seed)
    lib().orc_prox_squaredl12(_p(p), C.c_int64(len(p)), C.c_double(lam), C.byref(rng))
    return p


def prox_squaredl12_slow(p, lam):
    p = np.array(p, dtype=np.float64)
    lib().orc_prox_squaredl12_slow(_p(p), C.c_int64(len(p)), C.c_double(lam))
    return p


def prox(reg, Pt, lam, transpose=None, seed=1):
    """matrix prox on one order in the training layout [da][k]; returns a new array"""
    if transpose is None:
        transpose = reg == "squaredl12"
    Pt = np.array(Pt, dtype=np.float64, order="C")
    rng = C.c_uint64(seed)
    lib().orc_prox(REG[reg], int(transpose), _p(Pt), C.c_int64(Pt.shape[0]), int(Pt.shape[1]), C.c_double(lam),
                   C.byref(rng))
    return Pt


def reg_eval(reg, Pt, transpose=None):
    if transpose is None:
        transpose = reg == "squaredl12"
    Pt = np.ascontiguousarray(Pt, dtype=np.float64)
    return lib().orc_reg_eval(REG[reg], int(transpose), _p(Pt), C.c_int64(Pt.shape[0]), int(Pt.shape[1]))


def fm_mbpsgd_epoch(X, y, degree, P, w, intercept, cfg, stream, batch, n_aug=0, it=1, seed=1):
    """one outer iteration (minibatch_psgd.nim:87-122) in place on P (model layout) and w;
    returns (intercept, it, loss_sum)"""
    O, k, da = P.shape
    assert P.dtype == np.float64 and P.flags.c_contiguous and w.flags.c_contiguous
    y = f64(y)
    stream = i64(stream)
    b, itc, ls, rng = C.c_double(intercept), C.c_int64(it), C.c_double(0), C.c_uint64(seed)
    rc = lib().orc_fm_mbpsgd_epoch(C.byref(X.c), _p(y), degree, k, O, n_aug, _p(P), _p(w), C.byref(b), C.byref(cfg),
                                   _p(stream), C.c_int64(len(stream)), C.c_int64(batch), C.byref(itc), C.byref(rng),
                                   C.byref(ls))
    assert rc == 0
    return b.value, itc.value, ls.value


def fm_predict_all_with_grad(X, y, degree, P, w, intercept, loss="squared", n_aug=0, fit_linear=True, fit_intercept=True,
                             loss_param=1.0):
    """pgd.nim:70-103 -> (yPred, dL, gP [O][d+a][k], gw, gb)"""
    O_, k, da = P.shape
    y = f64(y)
    yp, dL = np.zeros(X.n), np.zeros(X.n)
    gP, gw, gb = np.zeros((O_, da, k)), np.zeros(X.d), C.c_double(0)
    rc = lib().orc_fm_predict_all_with_grad(C.byref(X.c), _p(y), degree, k, O_, n_aug, _p(np.ascontiguousarray(P)),
                                            _p(np.ascontiguousarray(w)), C.c_double(intercept), LOSS[loss],
                                            C.c_double(loss_param), int(fit_linear), int(fit_intercept), _p(yp), _p(dL),
                                            _p(gP), _p(gw), C.byref(gb))
    assert rc == 0
    return yp, dL, gP, gw, gb.value
