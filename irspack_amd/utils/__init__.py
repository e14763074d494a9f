"""The pieces of the reference's ``irspack.utils._util_cpp`` that the kNN path calls
(/root/reference/cpp_source/util.hpp:158-226, util.cpp:14,29-32).

``remove_diagonal`` goes through the C ABI; the two pre-weighting helpers are
plain element-wise host transforms (float64 like the reference) written with
numpy — they are data preparation, not part of the accelerated product.
"""

import ctypes as C

import numpy as np
import scipy.sparse as sps

from .. import _lib
from .._lib import check, lib, ptr


def remove_diagonal(X) -> sps.csr_matrix:
    """util.hpp:211-226: stored diagonal entries are set to 0.0 and kept (explicit zeros)."""
    Xc, indptr, indices, data = _lib.csr_arrays(X, np.float64)
    data = data.copy()
    if data.size == 0:
        data_arg = np.zeros(1, dtype=np.float64)
        idx_arg = np.zeros(1, dtype=np.int32)
    else:
        data_arg, idx_arg = data, indices
    check(lib().irs_remove_diagonal(C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]),
                                    ptr(indptr, C.c_int64), ptr(idx_arg, C.c_int32),
                                    ptr(data_arg, C.c_double)))
    out = sps.csr_matrix((data, indices.copy(), indptr.copy()), shape=Xc.shape)
    out.has_sorted_indices = True
    return out


def tf_idf_weight(X, smooth: bool = True) -> sps.csr_matrix:
    """util.hpp:190-209."""
    Xc = sps.csr_matrix(X, dtype=np.float64)
    Xc.sort_indices()
    N = Xc.shape[0]
    df = np.bincount(Xc.indices, minlength=Xc.shape[1]).astype(np.float64)
    with np.errstate(divide="ignore"):
        idf = np.log(N / (df + float(bool(smooth))))
    out = Xc.copy()
    out.data = out.data * idf[out.indices]
    return out


def okapi_BM_25_weight(X, k1: float = 1.2, b: float = 0.75) -> sps.csr_matrix:
    """util.hpp:158-188."""
    Xc = sps.csr_matrix(X, dtype=np.float64)
    Xc.sort_indices()
    N = Xc.shape[0]
    df = np.bincount(Xc.indices, minlength=Xc.shape[1]).astype(np.float64)
    doc_length = np.asarray(Xc.sum(axis=1)).ravel()
    avgdl = doc_length.sum() / N
    idf = np.log(N / (df + 1.0) + 1.0)
    rows = np.repeat(np.arange(N), np.diff(Xc.indptr))
    regularizer = k1 * (1 - b + b * doc_length[rows] / avgdl)
    out = Xc.copy()
    out.data = idf[out.indices] * (out.data * (k1 + 1)) / (out.data + regularizer)
    return out
