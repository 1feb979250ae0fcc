"""Host-side mirror of nimfm's FM surface over the C ABI (include/nimfm_hip.h).

Same names, argument meaning, defaults and error behaviour as the reference procs it stands in
for (citations relative to /root/reference/src/nimfm/), so that parity tests read like the
reference's own tests:

    newCSRDataset / newCSRFieldDataset       dataset.nim:116-123, tensor/sparse.nim:9-24
    newFactorizationMachine, init, decisionFunction, predict, predictProba, score
                                             model/factorization_machine.nim:43-139, model/fm_base.nim:13-48
    newFieldAwareFactorizationMachine        model/field_aware_factorization_machine.nim:24-92
    newSGD(...).fit(X, y, fm, callback)      optimizer/sgd.nim:23-52,261-328 (FFM: sgd_ffm.nim:49-106)
    newAdaGrad(...).fit(X, y, fm, callback)  optimizer/adagrad.nim:20-44,137-203 (FFM: adagrad_ffm.nim:11-66)
    fit(..., maxThreads=...)                 optimizer/sgd_multi.nim:40-42, adagrad_multi.nim:39-41

The epoch loop, shuffle, stopping criterion, verbose lines and callbacks run here exactly where
the Nim shim (nim/nimfm_hip.nim) runs them; everything per-sample runs in libnimfm_hip.so.
The reference is Nim and no Nim toolchain exists in the build image, so this Python module is the
executable stand-in for that shim (and what bench.py / torch.distributed drive).
"""
import ctypes as C
import os
import math

import numpy as np

from . import _capi as capi


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class Context:
    """One GPU + one HIP stream (one process per GPU)."""

    def __init__(self, device_id=0, stream=None):
        self.h = C.c_void_p()
        capi.check(capi.lib().nfm_ctx_create(device_id, stream, C.byref(self.h)))
        self.device_id = device_id

    def synchronize(self):
        capi.check(capi.lib().nfm_ctx_synchronize(self.h))

    def timing_enable(self, on=True):
        capi.check(capi.lib().nfm_ctx_timing_enable(self.h, int(on)))

    def timing_reset(self):
        capi.check(capi.lib().nfm_ctx_timing_reset(self.h))

    def timing_get(self, family):
        n, ms = C.c_int64(0), C.c_double(0.0)
        capi.check(capi.lib().nfm_ctx_timing_get(self.h, family.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def close(self):
        if self.h and capi.alive:
            capi.lib().nfm_ctx_destroy(self.h)
        self.h = C.c_void_p()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx


class CSRDataset:
    """dataset.nim:10-16 BaseDataset[CSRMatrix]; rows resident in HBM."""

    def __init__(self, data=None, indices=None, indptr=None, nSamples=0, nFeatures=0, fields=None, nFields=0,
                 ctx=None, _handle=None, _keep=None):
        self.ctx = ctx or default_context()
        self.nSamples, self._nFeatures, self.nFields = int(nSamples), int(nFeatures), int(nFields)
        self._keep = _keep
        if _handle is not None:
            self.h = _handle
            return
        data, indices, indptr = _f64(data), _i64(indices), _i64(indptr)
        if len(indptr) != self.nSamples + 1:
            raise ValueError("len(indptr) != nSamples + 1")
        if len(indices) != len(data) or (len(indptr) and indptr[-1] != len(data)):
            raise ValueError("indices/data/indptr sizes are inconsistent")
        fl = None if fields is None else _i64(fields)
        self.nnz = len(data)
        self.h = C.c_void_p()
        capi.check(capi.lib().nfm_dataset_create_csr(self.ctx.h, self.nSamples, self._nFeatures, _vp(indptr),
                                                     _vp(indices), _vp(data), _vp(fl), self.nFields, None,
                                                     C.byref(self.h)))

    @classmethod
    def from_device(cls, ctx, n, d, nnz, indptr_ptr, indices_ptr, data_ptr, y_ptr=None, fields_ptr=None, nFields=0,
                    keep=None):
        """Adopt device-resident arrays (raw device pointers, e.g. torch tensors' data_ptr())."""
        h = C.c_void_p()
        capi.check(capi.lib().nfm_dataset_create_csr_device(ctx.h, n, d, nnz, indptr_ptr, indices_ptr, data_ptr,
                                                            fields_ptr, nFields, y_ptr, C.byref(h)))
        ds = cls(nSamples=n, nFeatures=d, nFields=nFields, ctx=ctx, _handle=h, _keep=keep)
        ds.nnz = nnz
        return ds

    @classmethod
    def _from_loader(cls, ctx, h):
        n, d, nnz, nf = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().nfm_dataset_shape(h, C.byref(n), C.byref(d), C.byref(nnz), C.byref(nf)))
        ds = cls(nSamples=n.value, nFeatures=d.value, nFields=nf.value, ctx=ctx, _handle=h)
        ds.nnz = nnz.value
        return ds

    def targets(self):
        """the targets stored with the dataset (the loaders' y)"""
        y = np.zeros(self.nSamples)
        capi.check(capi.lib().nfm_dataset_get_targets(self.h, _vp(y)))
        return y

    def to_host(self):
        """(indptr, indices, data, fields or None) in the reference's widths (tensor/sparse.nim:9-12, 19-24)"""
        indptr = np.zeros(self.nSamples + 1, dtype=np.int64)
        indices = np.zeros(self.nnz, dtype=np.int64)
        data = np.zeros(self.nnz)
        fields = np.zeros(self.nnz, dtype=np.int64) if self.nFields else None
        capi.check(capi.lib().nfm_dataset_get_csr(self.h, _vp(indptr), _vp(indices), _vp(data), _vp(fields)))
        return indptr, indices, data, fields

    def ingest_stats(self):
        """(bytes of text, upload ms, parse ms) of the loader that built this dataset"""
        b, u, p = C.c_int64(), C.c_double(), C.c_double()
        capi.check(capi.lib().nfm_dataset_ingest_stats(self.h, C.byref(b), C.byref(u), C.byref(p)))
        return b.value, u.value, p.value

    @property
    def nFeatures(self):
        return self._nFeatures

    @property
    def shape(self):
        return [self.nSamples, self._nFeatures]

    def set_targets(self, y):
        y = _f64(y)
        if len(y) != self.nSamples:
            raise ValueError("len(y) != nSamples")
        capi.check(capi.lib().nfm_dataset_set_targets(self.h, _vp(y)))

    def __del__(self):
        try:
            if capi.alive and getattr(self, "h", None):
                capi.lib().nfm_dataset_destroy(self.h)
                self.h = None
        except Exception:
            pass


def newCSRDataset(data, indices, indptr, nSamples, nFeatures, ctx=None):
    return CSRDataset(data, indices, indptr, nSamples, nFeatures, ctx=ctx)


def newCSRFieldDataset(data, indices, indptr, fields, nSamples, nFeatures, nFields, ctx=None):
    return CSRDataset(data, indices, indptr, nSamples, nFeatures, fields=fields, nFields=nFields, ctx=ctx)


# ------------------------------------------------------------------------------------------------
# Nim's global random number generator, as the reference uses it: randomize(seed) in init
# (model/factorization_machine.nim:131), randomNormal for P (tensor/tensor.nim:561-580), shuffle(indices) in fit
# (optimizer/sgd.nim:297).  The procedures live in the library (nfm_rng_*, host code); the generator itself is
# outside the reference tree and restated from memory (unverified, include/nimfm_hip.h).
# ------------------------------------------------------------------------------------------------
class NimRand:
    def __init__(self, seed=None):
        self.state = (C.c_uint64 * 2)(0x69B4C98CB8530805, 0xFED1DD3004688D68)  # Nim's default global state
        if seed is not None:
            self.randomize(seed)

    def randomize(self, seed):
        capi.check(capi.lib().nfm_rng_randomize(int(seed), self.state))

    def randomNormal(self, shape, loc=0.0, scale=1.0):
        """tensor/tensor.nim:561-580: row-major fill, the cosine / sine halves of one draw on consecutive elements"""
        out = np.empty(int(np.prod(shape)), dtype=np.float64)
        capi.check(capi.lib().nfm_rng_random_normal(self.state, out.size, float(loc), float(scale), _vp(out)))
        return out.reshape(shape)

    def shuffle(self, x):
        """in place, Nim's shuffle: for i in countdown(high, 1): swap(x[i], x[rand(i)])"""
        assert x.dtype == np.int64 and x.flags["C_CONTIGUOUS"]
        capi.check(capi.lib().nfm_rng_shuffle(self.state, _vp(x), len(x)))


_global_rng = None


def globalRand():
    global _global_rng
    if _global_rng is None:
        _global_rng = NimRand()
    return _global_rng


def randomize(seed):
    globalRand().randomize(seed)


def randomNormal(shape, loc=0.0, scale=1.0, uniform=None):
    """tensor/tensor.nim:561-580.  uniform: an explicit stream of rand(1.0) draws (array, 2 per pair of elements)
    instead of the global generator -- the same pairing and fill order, vectorised."""
    if uniform is None:
        return globalRand().randomNormal(shape, loc, scale)
    n = int(np.prod(shape))
    u = _f64(uniform)[: 2 * ((n + 1) // 2)]
    if len(u) < 2 * ((n + 1) // 2):
        raise ValueError("randomNormal needs %d uniform draws" % (2 * ((n + 1) // 2)))
    x, y = u[0::2], u[1::2]
    r = np.sqrt(-2 * np.log(1.0 - x))
    z = np.empty(2 * len(x))
    z[0::2] = r * np.cos(2 * math.pi * y)
    z[1::2] = r * np.sin(2 * math.pi * y)
    return (loc + z[:n] * scale).reshape(shape)


def expit(x):
    """utils.nim:33"""
    x = np.asarray(x, dtype=np.float64)
    return np.exp(np.minimum(0.0, x)) / (1.0 + np.exp(-np.abs(x)))


def rmse(yTrue, yScore):
    """metrics.nim:5-13"""
    yTrue, yScore = _f64(yTrue), _f64(yScore)
    if len(yTrue) != len(yScore):
        raise ValueError("len(yScore)=%d, but len(yTrue)=%d" % (len(yScore), len(yTrue)))
    return math.sqrt(float(np.sum((yScore - yTrue) ** 2)) / len(yTrue))


def accuracy(yTrue, yPred):
    """metrics.nim:39-47"""
    yTrue, yPred = np.asarray(yTrue), np.asarray(yPred)
    if len(yTrue) != len(yPred):
        raise ValueError("len(yPred)=%d, but len(yTrue)=%d" % (len(yPred), len(yTrue)))
    return float(np.mean(yTrue == yPred))


class _ModelBase:
    """Fields/procs shared by FM and FFM (model/fm_base.nim)."""

    def __init__(self):
        self._h = None
        self._gen = 0  # bumps whenever the device model is released: optimizers built on it are stale then
        self._ctx = None
        self._dirty = True
        self._P = None
        self._w = None
        self._intercept = 0.0
        self.isInitialized = False

    # host copies are the public truth (fm.P / fm.w / fm.intercept); assigning marks the device copy stale
    @property
    def P(self):
        return self._P

    @P.setter
    def P(self, v):
        self._P = _f64(v)
        self._dirty = True

    @property
    def w(self):
        return self._w

    @w.setter
    def w(self, v):
        self._w = _f64(v)
        self._dirty = True

    @property
    def intercept(self):
        return self._intercept

    @intercept.setter
    def intercept(self, v):
        self._intercept = float(v)
        self._dirty = True

    def checkInitialized(self):
        if not self.isInitialized:
            raise capi.NotFittedError(capi.ERR_NOT_FITTED, "Factorization machines is not fitted.")

    def _handle(self, ctx):
        if self._h is None or self._ctx is not ctx:
            self._release()
            cfg = self._cfg()
            self._h = C.c_void_p()
            capi.check(capi.lib().nfm_model_create(ctx.h, C.byref(cfg), C.byref(self._h)))
            self._ctx = ctx
            self._dirty = True
        return self._h

    def _push(self, ctx):
        h = self._handle(ctx)
        if self._dirty:
            lams = getattr(self, "lams", None)
            capi.check(capi.lib().nfm_model_set_params(h, _vp(self._P), _vp(self._w), self._intercept,
                                                       None if lams is None else _vp(_f64(lams))))
            self._dirty = False
        return h

    def _pull(self):
        b = C.c_double(0.0)
        capi.check(capi.lib().nfm_model_get_params(self._h, _vp(self._P), _vp(self._w), C.byref(b)))
        self._intercept = b.value
        self._dirty = False

    def _release(self):
        if self._h is not None:
            self._gen += 1
            if capi.alive:
                capi.lib().nfm_model_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def decisionFunction(self, X):
        """model/factorization_machine.nim:100-122 / field_aware_factorization_machine.nim:52-76"""
        self.checkInitialized()
        self._check_shapes(X)
        if isinstance(X, StreamCSRDataset):  # block by block
            return np.concatenate([self.decisionFunction(X.load(r0, r1)) for r0, r1 in X.blocks()] or [np.zeros(0)])
        h = self._push(X.ctx)
        out = np.empty(X.nSamples, dtype=np.float64)
        capi.check(capi.lib().nfm_decision_function(h, X.h, _vp(out)))
        return out

    def predict(self, X):
        """model/fm_base.nim:18-20"""
        return np.sign(self.decisionFunction(X)).astype(np.int64)

    def predictProba(self, X):
        """model/fm_base.nim:23-26"""
        return expit(self.decisionFunction(X))

    def checkTarget(self, y):
        """model/fm_base.nim:29-36"""
        y = _f64(y)
        return np.sign(y) if self.task == "classification" else y

    def score(self, X, y):
        """model/fm_base.nim:39-48: rmse (regression) or accuracy of the signs (classification) of
        decisionFunction(X) against y -- reduced on the device, only the scalar comes back."""
        self.checkInitialized()
        self._check_shapes(X)
        X.set_targets(_f64(y))
        out = C.c_double()
        capi.check(capi.lib().nfm_score(self._push(X.ctx), X.h, C.byref(out)))
        return out.value

    def metrics(self, X, y):
        """rmse, accuracy (of signs) and rocauc (pos = 1) of decisionFunction(X) against y
        (metrics.nim:5-13, 39-47, 76-103), computed on the device."""
        self.checkInitialized()
        self._check_shapes(X)
        X.set_targets(_f64(y))
        r, a, u = C.c_double(), C.c_double(), C.c_double()
        capi.check(capi.lib().nfm_metrics(self._push(X.ctx), X.h, C.byref(r), C.byref(a), C.byref(u)))
        return {"rmse": r.value, "accuracy": a.value, "rocauc": u.value}


def _task_name(task):
    t = {"r": "regression", "c": "classification"}.get(task, task)
    if t not in ("regression", "classification"):
        raise ValueError("unknown task %r" % (task,))
    return t


class FactorizationMachine(_ModelBase):
    def __init__(self, task, degree=2, nComponents=30, fitLower="explicit", fitIntercept=True, fitLinear=True,
                 warmStart=False, randomState=1, scale=0.01):
        super().__init__()
        self.task = _task_name(task)
        if degree < 1:
            raise ValueError("degree < 1.")
        if nComponents < 1:
            raise ValueError("nComponents < 1.")
        if fitLower not in capi.LOWER:
            raise ValueError("unknown fitLower %r" % (fitLower,))
        self.degree, self.nComponents, self.fitLower = int(degree), int(nComponents), fitLower
        self.fitIntercept, self.fitLinear, self.warmStart = bool(fitIntercept), bool(fitLinear), bool(warmStart)
        self.randomState, self.scale = int(randomState), float(scale)
        self.lams = np.ones(self.nComponents)
        self._d = None

    @property
    def nAugments(self):
        """model/factorization_machine.nim:81-86"""
        if self.fitLower == "augment":
            return self.degree - 2 if self.fitLinear else self.degree - 1
        return 0

    @property
    def nOrders(self):
        """model/factorization_machine.nim:89-97"""
        if self.degree == 1:
            return 0
        return self.degree - 1 if self.fitLower == "explicit" else 1

    def _cfg(self):
        return capi.ModelCfg(capi.KIND_FM, capi.TASK[self.task], self.degree, self.nComponents,
                             capi.LOWER[self.fitLower], int(self.fitIntercept), int(self.fitLinear), 0,
                             int(self._d), 0)

    def _check_shapes(self, X):
        if X.nFeatures + self.nAugments != self._P.shape[2]:
            raise ValueError("Invalid nFeatures.")

    def init(self, X, force=False):
        """model/factorization_machine.nim:125-139: randomize(randomState); w = 0; P = randomNormal([nOrders,
        nComponents, nFeatures + nAugments], scale) -- Box-Muller pairs along the row-major fill -- intercept = 0.
        (The generator behind randomize / rand is Nim's stdlib, restated unverified: see NimRand.)"""
        if force or not (self.warmStart and self.isInitialized):
            d = X.nFeatures
            randomize(self.randomState)
            if self._d != d:
                self._release()
            self._d = d
            self.w = np.zeros(d)
            self.P = randomNormal((self.nOrders, self.nComponents, d + self.nAugments), scale=self.scale)
            self.intercept = 0.0
        self.isInitialized = True

    def set_params(self, P, w, intercept):
        """Inject parameters (the reference's warm-start path: init is skipped when
        warmStart and isInitialized, factorization_machine.nim:129)."""
        P = _f64(P)
        if P.ndim != 3 or P.shape[0] != self.nOrders or P.shape[1] != self.nComponents:
            raise ValueError("P must have shape [nOrders, nComponents, nFeatures+nAugments]")
        d = P.shape[2] - self.nAugments
        if self._d != d:
            self._release()
        self._d = d
        self.P, self.w, self.intercept = P.copy(), _f64(w).copy(), intercept
        if len(self._w) != d:
            raise ValueError("len(w) != nFeatures")
        self.isInitialized = True


def _nim_float(v):
    """Nim's `$float64`: the shortest form that reads back exactly (repr), "inf"/"nan" spelled Nim's way"""
    v = float(v)
    if v != v:
        return "nan"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    return repr(v)


def _dump_fm(self, fname):
    """model/factorization_machine.nim:142-165: the reference's text model format"""
    self.checkInitialized()
    P, w = self.P, self.w
    d = P.shape[2] - self.nAugments
    with open(os.path.expanduser(fname), "w") as f:
        f.write("task: %s\n" % self.task)
        f.write("nFeatures: %d\n" % d)
        f.write("degree: %d\n" % self.degree)
        f.write("nComponents: %d\n" % self.nComponents)
        f.write("fitLower: %s\n" % self.fitLower)
        f.write("fitIntercept: %s\n" % ("true" if self.fitIntercept else "false"))
        f.write("fitLinear: %s\n" % ("true" if self.fitLinear else "false"))
        f.write("randomState: %d\n" % self.randomState)
        f.write("scale: %s\n" % _nim_float(self.scale))
        f.write("lams:\n")
        f.write(" ".join(_nim_float(v) for v in self.lams) + "\n")
        for order in range(P.shape[0]):
            f.write("P[%d]:\n" % order)
            for s_ in range(self.nComponents):
                f.write(" ".join(_nim_float(v) for v in P[order, s_]) + "\n")
        f.write("w:\n")
        f.write(" ".join(_nim_float(v) for v in w) + "\n")
        f.write("intercept: %s\n" % _nim_float(self.intercept))


FactorizationMachine.dump = _dump_fm


def load(fname, warmStart):
    """model/factorization_machine.nim:168-220 `load(fm, fname, warmStart)`: -> FactorizationMachine"""
    with open(os.path.expanduser(fname)) as f:
        lines = f.read().split("\n")
    it = iter(lines)

    def field():
        return next(it).split(" ")[1]

    task = field()
    d = int(field())
    degree = int(field())
    k = int(field())
    fit_lower = field()
    fit_intercept = field().lower() in ("true", "y", "yes", "1", "on")  # Nim's parseBool
    fit_linear = field().lower() in ("true", "y", "yes", "1", "on")
    random_state = int(field())
    scale = float(field())
    fm = FactorizationMachine(task, degree, k, fit_lower, fit_intercept, fit_linear, bool(warmStart), random_state, scale)
    next(it)  # "lams:"
    lams = np.array([float(v) for v in next(it).split(" ") if v][:k])
    P = np.zeros((fm.nOrders, k, d + fm.nAugments))
    for order in range(fm.nOrders):
        next(it)  # "P[order]:"
        for s_ in range(k):
            row = [float(v) for v in next(it).split(" ") if v]
            P[order, s_] = row[: d + fm.nAugments]
    next(it)  # "w:"
    w = np.array([float(v) for v in next(it).split(" ") if v][:d])
    if len(w) != d:
        w = np.zeros(d)
    intercept = float(next(it).split(" ")[1])
    fm.set_params(P, w, intercept)
    fm.lams = lams
    return fm


class FieldAwareFactorizationMachine(_ModelBase):
    def __init__(self, task, nComponents=10, fitIntercept=True, fitLinear=True, warmStart=False, randomState=1,
                 scale=0.01):
        super().__init__()
        self.task = _task_name(task)
        if nComponents < 1:
            raise ValueError("nComponents < 1.")
        self.nComponents = int(nComponents)
        self.fitIntercept, self.fitLinear, self.warmStart = bool(fitIntercept), bool(fitLinear), bool(warmStart)
        self.randomState, self.scale = int(randomState), float(scale)
        self._d = None
        self._F = None

    nAugments = 0  # model/field_aware_factorization_machine.nim:49

    def _cfg(self):
        return capi.ModelCfg(capi.KIND_FFM, capi.TASK[self.task], 2, self.nComponents, 0, int(self.fitIntercept),
                             int(self.fitLinear), 0, int(self._d), int(self._F))

    def _check_shapes(self, X):
        if X.nFeatures != self._P.shape[1]:
            raise ValueError("Invalid nFeatures.")
        if X.nFields != self._P.shape[0]:
            raise ValueError("Invalid nFields.")

    def init(self, X, force=False):
        """model/field_aware_factorization_machine.nim:79-92"""
        if force or not (self.warmStart and self.isInitialized):
            d, F = X.nFeatures, X.nFields
            randomize(self.randomState)
            if (self._d, self._F) != (d, F):
                self._release()
            self._d, self._F = d, F
            self.w = np.zeros(d)
            self.P = randomNormal((F, d, self.nComponents), scale=self.scale)
            self.intercept = 0.0
        self.isInitialized = True

    def set_params(self, P, w, intercept):
        P = _f64(P)
        if P.ndim != 3 or P.shape[2] != self.nComponents:
            raise ValueError("P must have shape [nFields, nFeatures, nComponents]")
        if (self._d, self._F) != (P.shape[1], P.shape[0]):
            self._release()
        self._F, self._d = P.shape[0], P.shape[1]
        self.P, self.w, self.intercept = P.copy(), _f64(w).copy(), intercept
        self.isInitialized = True


def newFactorizationMachine(task, degree=2, nComponents=30, fitLower="explicit", fitIntercept=True, fitLinear=True,
                            warmStart=False, randomState=1, scale=0.01):
    return FactorizationMachine(task, degree, nComponents, fitLower, fitIntercept, fitLinear, warmStart, randomState,
                                scale)


def newFieldAwareFactorizationMachine(task, nComponents=10, fitIntercept=True, fitLinear=True, warmStart=False,
                                      randomState=1, scale=0.01):
    return FieldAwareFactorizationMachine(task, nComponents, fitIntercept, fitLinear, warmStart, randomState, scale)


# ------------------------------------------------------------------------------------------------
# loaders (dataset.nim:616-632, 768-790): text parsed on the GPU, the dataset stays in HBM
# ------------------------------------------------------------------------------------------------
def loadSVMLightFile(f, nFeatures=-1, ctx=None):
    """-> (CSRDataset, y).  The reference fills `var dataset` / `var y` (dataset.nim:616-617)."""
    ctx = ctx or default_context()
    h = C.c_void_p()
    capi.check(capi.lib().nfm_dataset_load_svmlight(ctx.h, os.path.expanduser(f).encode(), int(nFeatures), C.byref(h)))
    ds = CSRDataset._from_loader(ctx, h)
    return ds, ds.targets()


def loadFFMFile(f, nFeatures=-1, nFields=-1, ctx=None):
    """-> (CSRFieldDataset, y) (dataset.nim:768-790)."""
    ctx = ctx or default_context()
    h = C.c_void_p()
    capi.check(capi.lib().nfm_dataset_load_ffm(ctx.h, os.path.expanduser(f).encode(), int(nFeatures), int(nFields),
                                               C.byref(h)))
    ds = CSRDataset._from_loader(ctx, h)
    return ds, ds.targets()


class StreamCSRDataset:
    """A STREAMCSR / STREAMCSRFIELD file used in row blocks (dataset.nim:170-174 newStreamCSRDataset(f, cacheSize);
    tensor/sparse_stream.nim:232-270 readCache): at most cacheRows rows are resident in HBM at a time.  fit walks the
    blocks in file order without shuffling, as the reference does for a dataset that is not fully cached
    (optimizer/sgd.nim:297, sgd_multi.nim:83-97)."""

    def __init__(self, f, cacheRows, ctx=None, fY=None):
        self.ctx = ctx or default_context()
        self.h = C.c_void_p()
        capi.check(capi.lib().nfm_stream_open(self.ctx.h, os.path.expanduser(f).encode(),
                                              None if fY is None else os.path.expanduser(fY).encode(), C.byref(self.h)))
        n, d, nnz, nf = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().nfm_stream_shape(self.h, C.byref(n), C.byref(d), C.byref(nnz), C.byref(nf)))
        self.nSamples, self._nFeatures, self.nnz, self.nFields = n.value, d.value, nnz.value, nf.value
        self.cacheRows = max(1, int(cacheRows))

    @property
    def nFeatures(self):
        return self._nFeatures

    @property
    def shape(self):
        return [self.nSamples, self._nFeatures]

    def blocks(self):
        return [(r0, min(self.nSamples, r0 + self.cacheRows)) for r0 in range(0, self.nSamples, self.cacheRows)]

    def prefetch(self, r0, r1):
        """starts loading rows [r0, r1) beside whatever runs on the context's stream (nfm_stream_prefetch_rows); the next
        load(r0, r1) hands the block over"""
        capi.check(capi.lib().nfm_stream_prefetch_rows(self.h, r0, r1))

    def load(self, r0, r1):
        """rows [r0, r1) as a resident dataset"""
        h = C.c_void_p()
        capi.check(capi.lib().nfm_stream_load_rows(self.h, r0, r1, C.byref(h)))
        return CSRDataset._from_loader(self.ctx, h)

    def __del__(self):
        try:
            if capi.alive and getattr(self, "h", None):
                capi.lib().nfm_stream_close(self.h)
                self.h = None
        except Exception:
            pass


def newStreamCSRDataset(f, fY=None, ctx=None, cacheRows=None):
    """dataset.nim:170-174 newStreamCSRDataset (+ loadStreamLabel, :1007-1014, when fY is given).  cacheRows = None:
    the STREAMCSR / STREAMCSRFIELD file is made resident in HBM as a whole; cacheRows = r: a StreamCSRDataset that
    keeps at most r rows resident per block (the reference's cacheSize, in rows; fit loads the next block while the
    current one trains, so two blocks are resident at a time).  -> (dataset, y)"""
    ctx = ctx or default_context()
    if cacheRows is not None:
        ds = StreamCSRDataset(f, cacheRows, ctx, fY)
        y = np.fromfile(os.path.expanduser(fY), dtype=np.float64) if fY is not None else np.zeros(ds.nSamples)
        if len(y) != ds.nSamples:
            raise ValueError("%s holds %d labels, the matrix has %d rows" % (fY, len(y), ds.nSamples))
        return ds, y
    h = C.c_void_p()
    capi.check(capi.lib().nfm_dataset_load_stream(ctx.h, os.path.expanduser(f).encode(),
                                                  None if fY is None else os.path.expanduser(fY).encode(), C.byref(h)))
    ds = CSRDataset._from_loader(ctx, h)
    return ds, ds.targets()


def convertSVMLightFile(fIn, fOutX, fOutY, ctx=None):
    """dataset.nim:1017-1097: svmlight text -> STREAMCSR binary + raw float64 labels"""
    ctx = ctx or default_context()
    capi.check(capi.lib().nfm_convert_svmlight(ctx.h, os.path.expanduser(fIn).encode(), os.path.expanduser(fOutX).encode(),
                                               os.path.expanduser(fOutY).encode()))


def parseText(text, withFields=False, nFeatures=-1, nFields=-1, ctx=None):
    """the loaders on an in-memory buffer (bytes)"""
    ctx = ctx or default_context()
    if isinstance(text, str):
        text = text.encode()
    h = C.c_void_p()
    capi.check(capi.lib().nfm_dataset_parse_text(ctx.h, text, len(text), int(bool(withFields)), int(nFeatures),
                                                 int(nFields), C.byref(h)))
    ds = CSRDataset._from_loader(ctx, h)
    return ds, ds.targets()


# ------------------------------------------------------------------------------------------------
# optimizers
# ------------------------------------------------------------------------------------------------
def _echo_header(maxIter):
    """optimizer/utils.nim:26-40"""
    print("%s   %s   %s   Regularization" % ("Epoch".ljust(len(str(maxIter))), "Violation".ljust(10), "Loss".ljust(10)),
          flush=True)


def _echo_info(it, maxIter, viol, loss, regul):
    """optimizer/utils.nim:43-53"""
    print("%s   %-10.4e   %-10.4e   %-10.4e" % (str(it).ljust(max(5, len(str(maxIter)))), viol, loss, regul), flush=True)


class _OptimizerBase:
    """optimizer/optimizer_base.nim:2-8 + the fit driver shared by SGD and AdaGrad."""

    def __init__(self, maxIter, alpha0, alpha, beta, loss, verbose, tol, shuffle, nCalls, mode, batch, lossParam,
                 deviceShuffle=False):
        if loss not in capi.LOSS:
            raise ValueError("unknown loss %r" % (loss,))
        if mode not in capi.MODE:
            raise ValueError("unknown mode %r" % (mode,))
        self.maxIter, self.alpha0, self.alpha, self.beta = int(maxIter), float(alpha0), float(alpha), float(beta)
        self.loss, self.lossParam = loss, float(lossParam)
        self.verbose, self.tol, self.shuffle, self.nCalls = int(verbose), float(tol), bool(shuffle), int(nCalls)
        self.mode, self.batch = mode, int(batch)
        # deviceShuffle (mini-batch mode): shuffle = true draws each epoch's order on the device (nfm_opt_set_shuffle) and
        # builds the next epoch's batch plan beside the current epoch, instead of shuffling `indices` on the host
        self.deviceShuffle = bool(deviceShuffle)
        self.it = 1
        self._h = None
        self._model = None
        self._mode_built = None
        self._mh = None
        self.history = []  # (viol, mean loss) per epoch, what echoInfo prints
        self._dp = None  # (dp.Group, sync_period, overlap): see setDataParallel

    def _release(self):
        if self._h is not None and capi.alive:
            capi.lib().nfm_opt_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, fm, ctx, mode):
        mh = fm._push(ctx)
        key = (mode, self.batch) + self._cfg_key()  # a changed hyper-parameter needs a new device optimizer
        # the device optimizer belongs to ONE device model: a model that was released and created again (init /
        # set_params with another nFeatures) may sit at the same address -- compared by generation, not by pointer
        if self._h is None or self._model is not fm or self._mode_built != key or self._mh != (mh.value, fm._gen):
            self._release()
            self._h = C.c_void_p()
            self._create(mh, mode)
            self._model, self._mode_built, self._mh = fm, key, (mh.value, fm._gen)
            if self._dp is not None:
                self._attach_dp()
        return self._h

    def setDataParallel(self, group, syncPeriod=0, overlap=True, combine="auto"):
        """The maxThreads overloads across GPUs (optimizer/sgd_multi.nim:40-120 and twins): `group` is this rank's
        dp.Group; X handed to fit is then this rank's contiguous slice of the samples (dp.shard_bounds), every rank
        calls fit together, and the library reconciles the replicas every syncPeriod mini-batches (0: only at the end of
        every epoch).  AdaGrad's state increments are summed.  SGD: combine = "mean" (default) averages the ranks' increments
        (local SGD: as stable as one rank, the model moves as far as ONE rank's steps take it), "sum" adds them up (every
        rank's steps land in the model, as every Hogwild thread's steps do in the reference; acts like a step size times
        the number of ranks where features overlap -- keep syncPeriod small).  The epoch's loss / viol and the step
        counter `it` then cover the samples of ALL ranks."""
        if combine not in ("auto", "mean", "sum", "state_mean", "state_rsqrt", "state_cross"):
            raise ValueError("combine must be 'auto', 'mean', 'sum', 'state_mean', 'state_rsqrt' or 'state_cross'")
        # "auto" is resolved IN THE LIBRARY (NFM_DP_AUTO, the default of every optimizer -- the Nim and C++ hosts only call
        # nfm_opt_set_dp): what tools/dp_convergence.py measured: SGD -- the mean at any period; AdaGrad
        # -- the summed state is synchronous data-parallel AdaGrad when the ranks exchange after EVERY mini-batch and
        # over-shoots with longer periods, where the averaged state stays as stable as one rank
        self._dp = None if group is None else (group, int(syncPeriod), bool(overlap), combine)
        if self._h is not None:
            self._attach_dp()

    def _attach_dp(self):
        g = self._dp
        capi.check(capi.lib().nfm_opt_set_dp(self._h, None if g is None else g[0].h, 0 if g is None else g[1],
                                             1 if g is None or g[2] else 0))
        capi.check(capi.lib().nfm_opt_set_dp_combine(self._h, -1 if g is None else {"auto": -1, "mean": 0, "sum": 1, "state_mean": 2, "state_rsqrt": 3, "state_cross": 4}[g[3]]))

    def _sync_it(self):
        """with a group attached the library advances `it` by the samples of all ranks"""
        it = C.c_int64()
        capi.check(capi.lib().nfm_opt_get_it(self._h, C.byref(it)))
        self.it = it.value

    def _cfg_key(self):
        """what _create bakes into the device optimizer (the common part; subclasses extend it)"""
        return (self.alpha0, self.alpha, self.beta, self.loss, self.lossParam)

    def _epoch(self, X, perm, begin, end):
        ls, vs = C.c_double(0.0), C.c_double(0.0)
        capi.check(capi.lib().nfm_opt_epoch(self._h, X.h, _vp(perm), begin, end, C.byref(ls), C.byref(vs)))
        return ls.value, vs.value

    def last_permutation(self, n):
        """the sample order of the most recent permuted epoch call (nfm_opt_get_perm)"""
        out = np.zeros(n, dtype=np.int64)
        capi.check(capi.lib().nfm_opt_get_perm(self._h, _vp(out), n))
        return out

    def _finalize_into(self, fm):
        capi.check(capi.lib().nfm_opt_finalize(self._h))
        fm._pull()

    def _regularization(self, fm):
        """optimizer/utils.nim:56-59"""
        psq, wsq = C.c_double(0.0), C.c_double(0.0)
        capi.check(capi.lib().nfm_model_sqnorms(fm._h, C.byref(psq), C.byref(wsq)))
        b = C.c_double(0.0)
        return 0.5 * self.alpha * wsq.value + 0.5 * self.beta * psq.value  # intercept term added by caller

    def fit(self, X, y, fm, maxThreads=None, callback=None, perms=None, miniBatchSize=None, syncPeriod=None, devices=None):
        """optimizer/sgd.nim:261-328, adagrad.nim:137-203 (and the FFM / *_multi overloads).
        maxThreads (the reference's Hogwild overload) only SELECTS the mini-batch mode: a thread count is not a batch
        size.  The mode's knobs are explicit and defaulted: miniBatchSize (else the optimizer's `batch`, 8192),
        syncPeriod (mini-batches between exchanges of a data-parallel fit: setDataParallel, or `devices`), and
        devices = [ids]: ONE process trains over several GPUs (or one GPU listed several times) -- contiguous slices of
        the samples per device (optimizer/sgd_multi.nim:85-88), one replica + host thread each, reconciled by the library
        (nfm_dp_create_local).
        perms ([maxIter][n], optional) replaces the internal shuffle with explicit permutations: the
        reference shuffles with Nim's global RNG (sgd.nim:297), which a Nim host passes in here."""
        if devices is not None and len(devices) > 1:
            if perms is not None:
                raise ValueError("fit(devices=[...]) draws every rank's order itself (the device shuffle of its shard): perms is not supported")
            if isinstance(X, StreamCSRDataset):
                raise ValueError("fit(devices=[...]) needs a resident dataset (its rows are split over the devices), not a StreamCSRDataset")
            if callback is not None and self.nCalls > 0:
                raise ValueError("fit(devices=[...]) calls back once, after the fit: nCalls is not supported")
            return self._fit_devices(X, y, fm, list(devices), miniBatchSize, syncPeriod or 0, callback)
        if miniBatchSize is not None:  # for this fit only
            if int(miniBatchSize) < 1:
                raise ValueError("miniBatchSize < 1.")
            keep, self.batch = self.batch, int(miniBatchSize)
            try:
                return self.fit(X, y, fm, maxThreads, callback, perms, None, syncPeriod, None)
            finally:
                self.batch = keep
        if syncPeriod is not None and self._dp is not None:  # for this fit only, like miniBatchSize
            keep_dp, self._dp = self._dp, (self._dp[0], int(syncPeriod)) + tuple(self._dp[2:])
            if self._h is not None:
                self._attach_dp()
            try:
                return self.fit(X, y, fm, maxThreads, callback, perms, None, None, None)
            finally:
                self._dp = keep_dp
                if self._h is not None:
                    self._attach_dp()
        if isinstance(X, StreamCSRDataset):
            return self._fit_stream(X, y, fm, maxThreads, callback)
        fm.init(X)
        y = _f64(y)
        if len(y) != X.nSamples:
            raise ValueError("len(y) != nSamples")
        X.set_targets(y)  # checkTarget (fm_base.nim:29-36) is applied on the device from the model's task
        mode = self.mode if (maxThreads is None and self._dp is None) else "minibatch"
        if not fm.warmStart:
            self.it = 1  # sgd.nim:288-289, adagrad.nim:49-50
        self._handle(fm, X.ctx, mode)
        if fm._dirty:
            fm._push(X.ctx)
        capi.check(capi.lib().nfm_opt_set_it(self._h, self.it))
        dev_shuffle = self.shuffle and self.deviceShuffle and mode == "minibatch" and perms is None
        capi.check(capi.lib().nfm_opt_set_shuffle(self._h, int(getattr(fm, "randomState", 1)) if dev_shuffle else -1))
        if self.verbose > 0:
            _echo_header(self.maxIter)
        n = X.nSamples
        rng = globalRand()  # the reference shuffles with Nim's global generator, seeded by fm.init (sgd.nim:297)
        indices = np.arange(n, dtype=np.int64)
        isConverged = False
        per_epoch_cb = self._per_epoch_callback(callback)
        self.history = []
        # Host-side orders: epoch e+1's permutation is drawn while epoch e is still to run (the same sequence of shuffles
        # of the same array as the reference's, one epoch early) and announced to the library, which builds its batch
        # plan beside the running epoch (nfm_opt_announce_perm); two index arrays alternate.
        host_shuffle = self.shuffle and not dev_shuffle and perms is None
        order = [indices, None]
        if host_shuffle:
            rng.shuffle(order[0])  # sgd.nim:297 (Nim's global RNG there)
        whole_epochs = not (callback is not None and self.nCalls > 0 and mode == "sequential")
        if perms is not None:
            perms = [_i64(p_) for p_ in perms]
        for epoch in range(self.maxIter):
            viol = runningLoss = 0.0
            perm = nxt = None
            if perms is not None:
                perm = perms[epoch]
                nxt = perms[epoch + 1] if epoch + 1 < len(perms) and epoch + 1 < self.maxIter else None
            elif host_shuffle:
                perm = order[epoch % 2]
                if epoch + 1 < self.maxIter:
                    order[(epoch + 1) % 2] = perm.copy()
                    rng.shuffle(order[(epoch + 1) % 2])
                    nxt = order[(epoch + 1) % 2]
            if nxt is not None and whole_epochs and mode == "minibatch":
                capi.check(capi.lib().nfm_opt_announce_perm(self._h, _vp(nxt), 0, n))
            if callback is not None and self.nCalls > 0 and mode == "sequential":
                pos = 0
                while pos < n:  # sgd.nim:303-308: callback whenever it mod nCalls == 0
                    to_next = (self.nCalls - self.it % self.nCalls) % self.nCalls + 1
                    end = min(n, pos + to_next)
                    ls, vs = self._epoch(X, perm, pos, end)
                    runningLoss += ls
                    viol += vs
                    self.it += end - pos
                    pos = end
                    if (self.it - 1) % self.nCalls == 0:
                        self._finalize_into(fm)
                        self.it -= 1  # the reference calls back before inc(self.it)
                        callback(self, fm)
                        self.it += 1
            else:
                runningLoss, viol = self._epoch(X, perm, 0, n)
                self.it += n
            n_all = n
            if self._dp is not None:  # sums and step counter cover all ranks' samples
                it0 = self.it - n
                self._sync_it()
                n_all = self.it - it0
            runningLoss /= float(n_all)
            if per_epoch_cb:
                self._finalize_into(fm)
                callback(self, fm)
            self.history.append((viol, runningLoss))
            # stoppingCriterion, sgd.nim:72-89
            isContinue = True
            if math.isnan(runningLoss):
                print("Loss is NaN. Use smaller learning rate.")
                isContinue = False
            if self.verbose > 0:
                b = C.c_double(0.0)
                reg = self._regularization(fm)
                capi.check(capi.lib().nfm_model_get_params(fm._h, None, None, C.byref(b)))
                _echo_info(epoch + 1, self.maxIter, viol, runningLoss, reg + 0.5 * self.alpha0 * b.value ** 2)
            if viol < self.tol:
                if self.verbose > 0:
                    print("Converged at epoch %d." % epoch)
                isConverged = True
                isContinue = False
            if not isContinue:
                break
        if not isConverged and self.verbose > 0:
            print("Objective did not converge. Increase maxIter.")
        self._finalize_into(fm)
        return self


def _fit_devices(self, X, y, fm, devices, miniBatchSize, syncPeriod, callback):
    """fit(..., devices=[...]): the maxThreads overloads over several GPUs from ONE process.  Rank r gets the contiguous
    slice dp.shard_bounds(n, r, world) of the samples on devices[r] (the reference's thread partition,
    optimizer/sgd_multi.nim:85-88), its own replica of `fm` and of this optimizer, and a host thread; the library
    reconciles the replicas every syncPeriod mini-batches and exactly at the end of every epoch (nfm_dp_create_local).
    All replicas end bitwise identical; rank 0's comes back in `fm`, its history / step counter in `self`."""
    import copy
    import threading

    from . import dp

    fm.init(X)
    y = _f64(y)
    n, world = X.nSamples, len(devices)
    if len(y) != n:
        raise ValueError("len(y) != nSamples")
    indptr, indices, data, fields = X.to_host()
    ctxs = [Context(int(d_)) for d_ in devices]
    groups = dp.Group.local(ctxs)
    P0, w0, b0 = np.array(fm.P), np.array(fm.w), float(fm.intercept)
    models, opts, err = [None] * world, [None] * world, []

    def body(r):
        try:
            lo, hi = dp.shard_bounds(n, r, world)
            a, b = int(indptr[lo]), int(indptr[hi])
            if fields is not None:
                Xr = newCSRFieldDataset(data[a:b], indices[a:b], indptr[lo:hi + 1] - a, fields[a:b], hi - lo, X.nFeatures, X.nFields, ctx=ctxs[r])
            else:
                Xr = newCSRDataset(data[a:b], indices[a:b], indptr[lo:hi + 1] - a, hi - lo, X.nFeatures, ctx=ctxs[r])
            fr = copy.copy(fm)
            fr._h, fr._ctx, fr._dirty = None, None, True  # a replica of its own on this rank's device
            fr.warmStart = True
            fr.set_params(P0, w0, b0)
            orr = copy.copy(self)
            orr._h, orr._model, orr._mode_built, orr._mh, orr.history = None, None, None, None, []
            if miniBatchSize is not None:
                orr.batch = int(miniBatchSize)
            orr.setDataParallel(groups[r], syncPeriod=syncPeriod)
            orr.fit(Xr, y[lo:hi], fr, maxThreads=world, callback=None)
            models[r], opts[r] = fr, orr
        except BaseException as e:  # noqa: BLE001
            err.append((r, e))

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g in groups:
        g.close()
    if err:
        raise err[0][1]
    fm.set_params(models[0].P, models[0].w, models[0].intercept)
    self.it, self.history = opts[0].it, list(opts[0].history)
    if callback is not None:  # ONCE, with the finished model (the ranks train without a host in their epoch loops)
        callback(self, fm)
    return self


def _fit_stream(self, X, y, fm, maxThreads=None, callback=None):
    """fit over a dataset that is resident one row block at a time (optimizer/sgd_multi.nim:83-97: `while nRest > 0:
    X.readCache(...)`): blocks in file order, no shuffling (sgd.nim:297 shuffles only a fully cached dataset); the
    optimizer's step counter, scales and state continue from block to block."""
    fm.init(X)
    y = _f64(y)
    if len(y) != X.nSamples:
        raise ValueError("len(y) != nSamples")
    mode = self.mode if (maxThreads is None and self._dp is None) else "minibatch"
    if not fm.warmStart:
        self.it = 1
    self._handle(fm, X.ctx, mode)
    if fm._dirty:
        fm._push(X.ctx)
    capi.check(capi.lib().nfm_opt_set_it(self._h, self.it))
    capi.check(capi.lib().nfm_opt_set_shuffle(self._h, -1))
    if self.verbose > 0:
        _echo_header(self.maxIter)
    n = X.nSamples
    isConverged = False
    self.history = []
    blocks = X.blocks()
    for epoch in range(self.maxIter):
        viol = runningLoss = 0.0
        for bi, (r0, r1) in enumerate(blocks):
            blk = X.load(r0, r1)
            if len(blocks) > 1 and os.environ.get("NIMFM_STREAM_PREFETCH", "1") != "0":  # the next block (the next epoch's first one after the last) is read, uploaded and split
                X.prefetch(*blocks[(bi + 1) % len(blocks)])  # while this one trains: two blocks resident at a time
            blk.set_targets(y[r0:r1])
            ls, vs = self._epoch(blk, None, 0, r1 - r0)
            runningLoss += ls
            viol += vs
            self.it += r1 - r0
            del blk
        runningLoss /= float(n)
        if callback is not None:
            self._finalize_into(fm)
            callback(self, fm)
        self.history.append((viol, runningLoss))
        isContinue = True
        if math.isnan(runningLoss):
            print("Loss is NaN. Use smaller learning rate.")
            isContinue = False
        if self.verbose > 0:
            b = C.c_double(0.0)
            reg = self._regularization(fm)
            capi.check(capi.lib().nfm_model_get_params(fm._h, None, None, C.byref(b)))
            _echo_info(epoch + 1, self.maxIter, viol, runningLoss, reg + 0.5 * self.alpha0 * b.value ** 2)
        if viol < self.tol:
            if self.verbose > 0:
                print("Converged at epoch %d." % epoch)
            isConverged = True
            isContinue = False
        if not isContinue:
            break
    if not isConverged and self.verbose > 0:
        print("Objective did not converge. Increase maxIter.")
    self._finalize_into(fm)
    return self


_OptimizerBase._fit_stream = _fit_stream
_OptimizerBase._fit_devices = _fit_devices


class SGD(_OptimizerBase):
    def __init__(self, maxIter=100, eta0=0.01, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared",
                 scheduling="optimal", power=1.0, verbose=1, tol=1e-3, shuffle=True, nCalls=-1, mode="sequential",
                 batch=8192, lossParam=1.0, deviceShuffle=False, touchCap=1.0):
        super().__init__(maxIter, alpha0, alpha, beta, loss, verbose, tol, shuffle, nCalls, mode, batch, lossParam, deviceShuffle)
        if scheduling not in capi.SCHED:
            raise ValueError("unknown scheduling %r" % (scheduling,))
        self.eta0, self.scheduling, self.power = float(eta0), scheduling, float(power)
        # mini-batch mode: how many of a batch's per-sample steps on one coordinate are summed before averaging sets in
        # (nfm_opt_set_touch_cap; 1 = the per-coordinate mean; the reference's Hogwild with T threads ~ T)
        if not float(touchCap) >= 1.0:
            raise ValueError("touchCap < 1.")
        self.touchCap = float(touchCap)

    def _create(self, mh, mode):
        cfg = capi.SGDCfg(self.eta0, self.alpha0, self.alpha, self.beta, self.power, self.lossParam,
                          capi.LOSS[self.loss], capi.SCHED[self.scheduling], capi.MODE[mode], 0, self.batch)
        capi.check(capi.lib().nfm_sgd_create(mh, C.byref(cfg), C.byref(self._h)))
        if mode == "minibatch" and self.touchCap != 1.0:
            capi.check(capi.lib().nfm_opt_set_touch_cap(self._h, self.touchCap))

    def _cfg_key(self):
        return super()._cfg_key() + (self.eta0, self.scheduling, self.power, self.touchCap)

    def _per_epoch_callback(self, callback):
        return callback is not None and (self.nCalls <= 0 or self.mode != "sequential")  # sgd.nim:312


class AdaGrad(_OptimizerBase):
    def __init__(self, maxIter=100, eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared", eps=1e-10,
                 verbose=1, tol=1e-3, shuffle=True, nCalls=-1, mode="sequential", batch=8192, lossParam=1.0,
                 trackViol=True, deviceShuffle=False, adaCross=0.0):
        super().__init__(maxIter, alpha0, alpha, beta, loss, verbose, tol, shuffle, nCalls, mode, batch, lossParam, deviceShuffle)
        self.eta0, self.eps, self.trackViol = float(eta0), float(eps), bool(trackViol)
        # mini-batch mode: the weight of the batch's gradient cross products in g_norm (nfm_opt_set_ada_cross; 0 = the samples'
        # squares alone; 0.1 lets field-aware AdaGrad run at batch 32768 with the epochs-to-target of batch 2048)
        if not float(adaCross) >= 0.0:
            raise ValueError("adaCross < 0.")
        self.adaCross = float(adaCross)

    def _create(self, mh, mode):
        cfg = capi.AdaGradCfg(self.eta0, self.alpha0, self.alpha, self.beta, self.eps, self.lossParam,
                              capi.LOSS[self.loss], capi.MODE[mode], int(self.trackViol), 0, self.batch)
        capi.check(capi.lib().nfm_adagrad_create(mh, C.byref(cfg), C.byref(self._h)))
        if mode == "minibatch" and self.adaCross != 0.0:
            capi.check(capi.lib().nfm_opt_set_ada_cross(self._h, self.adaCross))

    def _cfg_key(self):
        return super()._cfg_key() + (self.eta0, self.eps, self.trackViol, self.adaCross)

    def _per_epoch_callback(self, callback):
        return callback is not None  # adagrad.nim:188-191

    def get_state(self, fm):
        """g_sum / g_norm (adagrad.nim:15-16) in the reference layout."""
        nb, da, k = (fm._P.shape[0], fm._P.shape[2], fm._P.shape[1]) if isinstance(fm, FactorizationMachine) else (
            fm._P.shape[0], fm._P.shape[1], fm._P.shape[2])
        gs, gn = np.zeros((nb, da, k)), np.zeros((nb, da, k))
        gsw, gnw = np.zeros(len(fm._w)), np.zeros(len(fm._w))
        gsb, gnb = C.c_double(0.0), C.c_double(0.0)
        capi.check(capi.lib().nfm_opt_get_state(self._h, _vp(gs), _vp(gn), _vp(gsw), _vp(gnw), C.byref(gsb),
                                                C.byref(gnb)))
        return gs, gn, gsw, gnw, gsb.value, gnb.value


def suggestTouchCap(X, batch):
    """A touch cap for `newSGD(mode="minibatch", batch=batch, touchCap=...)` on dataset X (anything with nSamples, nFeatures,
    nnz): about twice the mean number of a batch's samples that touch one coordinate, lambda = batch * (entries per row) /
    nFeatures, as a power of two between 16 and 64 -- the settings measured to keep the epochs to a held-out loss of the
    reference's sample order (profiles/r05h_touch_cap_sweep.txt: 16 up to lambda ~ 10, 32 at 17-21,
    64 at 34).  Not a default: the library's own default stays 1, the per-coordinate mean (include/nimfm_hip.h)."""
    n, d, nnz = int(X.nSamples), int(X.nFeatures), int(X.nnz)
    if n <= 0 or d <= 0 or int(batch) < 1:
        raise ValueError("suggestTouchCap: empty dataset or batch < 1.")
    lam = min(int(batch), n) * (nnz / n) / d
    cap = 16.0
    while cap < 64.0 and 2.0 * lam > cap * 2.0 ** 0.5:  # (to the nearest power of two in the logarithm)
        cap *= 2.0
    return cap


def newSGD(maxIter=100, eta0=0.01, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared", scheduling="optimal",
           power=1.0, verbose=1, tol=1e-3, shuffle=True, nCalls=-1, **gpu):
    return SGD(maxIter, eta0, alpha0, alpha, beta, loss, scheduling, power, verbose, tol, shuffle, nCalls, **gpu)


def newAdaGrad(maxIter=100, eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-3, loss="squared", eps=1e-10, verbose=1,
               tol=1e-3, shuffle=True, nCalls=-1, **gpu):
    return AdaGrad(maxIter, eta0, alpha0, alpha, beta, loss, eps, verbose, tol, shuffle, nCalls, **gpu)


# ------------------------------------------------------------------------------------------------
# mini-batch proximal SGD
# ------------------------------------------------------------------------------------------------
class _Regularizer:
    """regularizer/*.nim: the sparsity-inducing penalties that have a matrix proximal operator.
    eval (for the verbose line, minibatch_psgd.nim:196-199) runs on the host copy of one order, [d+a][k]."""
    name = None

    def __init__(self, transpose=False):
        self.transpose = bool(transpose)


class L1(_Regularizer):
    name = "l1"

    def eval(self, Pt, degree=2):  # l1.nim:19-22
        return float(np.abs(Pt).sum())


class L21(_Regularizer):
    name = "l21"

    def eval(self, Pt, degree=2):  # l21.nim:17-20
        return float(np.sqrt((Pt * Pt).sum(1)).sum())


class SquaredL12(_Regularizer):
    name = "squaredl12"

    def __init__(self, transpose=True):  # squaredl12.nim:85-88
        super().__init__(transpose)

    def eval(self, Pt, degree=2):  # squaredl12.nim:72-82
        if degree > 2:
            raise ValueError("SquaredL12 supports only degree=2.")
        return float((np.abs(Pt).sum(0 if self.transpose else 1) ** 2).sum())


class SquaredL21(_Regularizer):
    name = "squaredl21"

    def __init__(self, transpose=False):  # squaredl21.nim:15-17
        super().__init__(transpose)

    def eval(self, Pt, degree=2):  # squaredl21.nim:21-29
        if degree != 2:
            raise ValueError("SquaredL21 supports only degree=2.")
        return float(np.sqrt((Pt * Pt).sum(0 if self.transpose else 1)).sum() ** 2)


def newL1():
    return L1()


def newL21():
    return L21()


def newSquaredL12(transpose=True):
    return SquaredL12(transpose)


def newSquaredL21(transpose=False):
    return SquaredL21(transpose)


class MBPSGD(_OptimizerBase):
    """optimizer/minibatch_psgd.nim:11-65,125-210: newMBPSGD(...).fit(X, y, sfm).  The gradient of a mini-batch,
    the step on all parameters and the proximal operator run on the device (nfm_mbpsgd_create / nfm_opt_epoch);
    the outer loop, the index stream (indices[ii] with wrap-around and reshuffle, :98-108), the stopping rule
    (:201-204) and the verbose lines run here where the reference has them."""

    def __init__(self, maxIter=100, eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-4, gamma=1e-4, loss="squared", reg=None,
                 miniBatchSize=-1, maxIterInner=-1, scheduling="optimal", power=1.0, verbose=1, tol=1e-6, shuffle=True,
                 nCalls=-1, lossParam=1.0):
        super().__init__(maxIter, alpha0, alpha, beta, loss, verbose, tol, shuffle, nCalls, "minibatch", 1, lossParam)
        if scheduling not in capi.SCHED:
            raise ValueError("unknown scheduling %r" % (scheduling,))
        self.reg = reg if reg is not None else newSquaredL12()
        if not isinstance(self.reg, _Regularizer):
            raise ValueError("reg must be one of newL1(), newL21(), newSquaredL12(), newSquaredL21()")
        self.gamma, self.eta0, self.scheduling, self.power = float(gamma), float(eta0), scheduling, float(power)
        self.miniBatchSize, self.maxIterInner = int(miniBatchSize), int(maxIterInner)
        self.it = 0  # :63

    def _create(self, mh, mode):
        cfg = capi.MBPSGDCfg(self.eta0, self.alpha0, self.alpha, self.beta, self.gamma, self.power, self.lossParam,
                             capi.LOSS[self.loss], capi.SCHED[self.scheduling], capi.REG[self.reg.name],
                             int(self.reg.transpose), self.batch)
        capi.check(capi.lib().nfm_mbpsgd_create(mh, C.byref(cfg), C.byref(self._h)))

    def _cfg_key(self):
        return super()._cfg_key() + (self.eta0, self.gamma, self.scheduling, self.power, self.reg.name, self.reg.transpose)

    def fit(self, X, y, sfm, callback=None, stream=None):
        """stream (optional): the sample indices in the order the inner loops consume them, at least
        maxIter * miniBatchSize * maxIterInner of them -- replaces the internal shuffle (Nim's global RNG in the
        reference, :107,170), which a Nim host passes in here."""
        if not isinstance(sfm, FactorizationMachine):
            raise ValueError("MBPSGD fits a FactorizationMachine")
        sfm.init(X)
        y = _f64(y)
        if len(y) != X.nSamples:
            raise ValueError("len(y) != nSamples")
        X.set_targets(y)
        n, d = X.nSamples, X.nFeatures
        if not sfm.warmStart:
            self.it = 1  # :153-154
        B = self.miniBatchSize
        if B <= 0:  # :160-163
            B = max((d * n) // max(X.nnz, 1), 1)
        inner = self.maxIterInner
        if inner <= 0:  # :164-167
            inner = max((n - 1) // B + 1, 1)
        if self.reg.name in ("squaredl12", "squaredl21") and sfm.degree != 2:  # initSGD, squaredl12.nim:103-105
            raise ValueError("%s supports only degree=2." % type(self.reg).__name__)
        self.batch = B
        self._handle(sfm, X.ctx, "minibatch")
        if sfm._dirty:
            sfm._push(X.ctx)
        capi.check(capi.lib().nfm_opt_set_it(self._h, self.it))
        rng = globalRand()
        indices = np.arange(n, dtype=np.int64)
        ii = 0
        if stream is None and self.shuffle:
            rng.shuffle(indices)  # :169-170
        if stream is not None:
            stream = _i64(stream)
        if self.verbose > 0:
            print("Minibatch size: %d" % B)
            print("Number of inner iteration: %d" % inner)
            print("%s   %s   Regularization" % ("Epoch".ljust(len(str(self.maxIter))), "Loss".ljust(10)), flush=True)
        oldLossVal = float("inf")
        isConverged = False
        self.history = []
        need = B * inner
        for it in range(self.maxIter):
            if stream is not None:
                chunk = stream[it * need:(it + 1) * need]
                if len(chunk) != need:
                    raise ValueError("stream holds fewer than maxIter * miniBatchSize * maxIterInner indices")
            else:  # :98-108: indices[ii], ii wraps and reshuffles
                chunk = np.empty(need, dtype=np.int64)
                got = 0
                while got < need:
                    take = min(need - got, n - ii)
                    chunk[got:got + take] = indices[ii:ii + take]
                    got += take
                    ii += take
                    if ii >= n:
                        ii = 0
                        if self.shuffle:
                            rng.shuffle(indices)
            ls, _ = self._epoch(X, chunk, 0, need)
            self.it += inner
            runningLoss = ls / float(B * inner)  # :122
            self.history.append((0.0, runningLoss))
            if callback is not None:  # :185-187
                self._finalize_into(sfm)
                callback(self, sfm)
            if math.isnan(runningLoss):  # :189-191
                print("Loss is NaN. Use smaller learning rate.")
                break
            if self.verbose > 0:  # :193-199
                self._finalize_into(sfm)
                regVal = 0.5 * self.alpha0 * sfm.intercept ** 2 + 0.5 * self.alpha * float((sfm.w ** 2).sum()) \
                    + 0.5 * self.beta * float((sfm.P ** 2).sum())
                for order in range(sfm.P.shape[0]):
                    regVal += self.gamma * self.reg.eval(np.ascontiguousarray(sfm.P[order].T), sfm.degree - order)
                print("%s   %-10.4e   %-10.4e" % (str(it + 1).ljust(max(5, len(str(self.maxIter)))), runningLoss, regVal),
                      flush=True)
            if abs(oldLossVal - runningLoss) < self.tol:  # :201-204
                if self.verbose > 0:
                    print("Converged at epoch %d." % (it + 1))
                isConverged = True
                break
            oldLossVal = runningLoss
        if not isConverged and self.verbose > 0:
            print("Objective did not converge. Increase maxIter.")
        self._finalize_into(sfm)
        return self


def predictAllWithGrad(X, y, sfm, loss="squared", lossParam=1.0):
    """optimizer/pgd.nim:70-103 on the device: -> (yPred, dL, grads) with grads = {"P": [nOrders][d+a][k] (the training
    layout of the reference's grads.P), "w": [d], "intercept": float, "loss": mean loss} -- the gradient of the mean
    loss at sfm's current parameters (sfm must be initialised; classification targets are sign()-ed)."""
    sfm.checkInitialized()
    y = _f64(y)
    if len(y) != X.nSamples:
        raise ValueError("len(y) != nSamples")
    X.set_targets(y)
    # the device optimizer (and with it the one-batch plan of X, nfm_opt_predict_all_with_grad) is kept on the model:
    # PGD-style solvers ask for the full gradient of the same data once per iteration
    opt = getattr(sfm, "_grad_opt", None)
    if opt is None or opt._grad_key != (loss, float(lossParam)):
        opt = MBPSGD(maxIter=1, loss=loss, reg=newL1(), miniBatchSize=1, verbose=0, lossParam=lossParam)
        opt._grad_key = (loss, float(lossParam))
        sfm._grad_opt = opt
    opt._handle(sfm, X.ctx, "minibatch")
    if sfm._dirty:
        sfm._push(X.ctx)
    n, d = X.nSamples, X.nFeatures
    nb, k, da = sfm._P.shape
    yp, dL = np.zeros(n), np.zeros(n)
    gP, gw = np.zeros((nb, da, k)), np.zeros(d)
    gb, ls = C.c_double(0.0), C.c_double(0.0)
    capi.check(capi.lib().nfm_opt_predict_all_with_grad(opt._h, X.h, _vp(yp), _vp(dL), _vp(gP), _vp(gw), C.byref(gb),
                                                        C.byref(ls)))
    return yp, dL, {"P": gP, "w": gw, "intercept": gb.value, "loss": ls.value / max(n, 1)}


def newMBPSGD(maxIter=100, eta0=0.1, alpha0=1e-6, alpha=1e-3, beta=1e-4, gamma=1e-4, loss="squared", reg=None,
              miniBatchSize=-1, maxIterInner=-1, scheduling="optimal", power=1.0, verbose=1, tol=1e-6, shuffle=True,
              nCalls=-1, **gpu):
    return MBPSGD(maxIter, eta0, alpha0, alpha, beta, gamma, loss, reg, miniBatchSize, maxIterInner, scheduling, power,
                  verbose, tol, shuffle, nCalls, **gpu)
