"""Helpers for the -m gpu parity tests: build nimfm_amd objects from the same arrays the oracle gets."""
import numpy as np

import nimfm_amd as nf
import oracle as O


def to_gpu(Xo):
    """oracle.Dataset -> nimfm_amd.CSRDataset (same arrays, through nfm_dataset_create_csr)."""
    if Xo.fields is not None:
        return nf.newCSRFieldDataset(Xo.data, Xo.indices, Xo.indptr, Xo.fields, Xo.n, Xo.d, Xo.n_fields)
    return nf.newCSRDataset(Xo.data, Xo.indices, Xo.indptr, Xo.n, Xo.d)


def gpu_fm(task, degree, k, fit_lower, fit_linear, fit_intercept, P0, w0, b0):
    fm = nf.newFactorizationMachine(task, degree=degree, nComponents=k, fitLower=fit_lower, fitLinear=fit_linear,
                                    fitIntercept=fit_intercept, warmStart=True)
    fm.set_params(P0, w0, b0)
    return fm


def gpu_ffm(task, k, fit_linear, fit_intercept, P0, w0, b0):
    ffm = nf.newFieldAwareFactorizationMachine(task, nComponents=k, fitLinear=fit_linear, fitIntercept=fit_intercept,
                                               warmStart=True)
    ffm.set_params(P0, w0, b0)
    return ffm


def ragged_csr(n, d, seed, max_m=100, empty_every=7):
    """Rows of very different lengths: empty rows, 1 nnz, > 64 nnz (more than one wavefront chunk)."""
    rng = np.random.default_rng(seed)
    rows, vals, indptr = [], [], [0]
    for i in range(n):
        if empty_every and i % empty_every == 3:
            m = 0
        else:
            m = int(rng.integers(1, min(max_m, d) + 1))
        idx = rng.choice(d, size=m, replace=False)
        if i % 2 == 0:
            idx = np.sort(idx)  # storage order is not required to be sorted (dataset.nim:597-612)
        rows.append(idx)
        vals.append(rng.uniform(-1, 1, size=m))
        indptr.append(indptr[-1] + m)
    return O.Dataset(np.array(indptr), np.concatenate(rows) if rows else np.zeros(0), np.concatenate(vals), n, d)
