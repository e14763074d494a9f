"""The pieces of the reference's ``irspack.utils._util_cpp`` that the kNN path calls
(/root/reference/cpp_source/util.hpp:158-226, util.cpp:14,29-32).

Everything goes through the C ABI: ``remove_diagonal``, the serving top-k
``retrieve_recommend_from_score`` and the two feature weightings (``irs_knn_weight``; the kNN
recommenders do not call them - they hand the weighting to the computer's constructor, which
applies it on the device without a host copy of the weighted matrix).
"""

from typing import List, Optional, Sequence, Tuple

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


def _weight(X, scheme: int, k1: float, b: float, smooth: bool, device: Optional[int]) -> sps.csr_matrix:
    """X's pattern with the weighted values: column counts, row sums and the idf table on host threads,
    the per-entry pass on the device (``irs_knn_weight``; the values are the host loop's bit for bit)."""
    Xc, indptr, indices, data = _lib.csr_arrays(X, np.float64)
    out = np.empty(data.shape[0], dtype=np.float64)
    if data.shape[0]:
        check(lib().irs_knn_weight(
            C.c_int32(scheme), C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]), ptr(indptr, C.c_int64),
            ptr(indices, C.c_int32), ptr(data, C.c_double), C.c_double(k1), C.c_double(b),
            C.c_int32(1 if smooth else 0), C.c_int32(_lib.default_device() if device is None else device),
            ptr(out, C.c_double)))
    res = sps.csr_matrix((out, Xc.indices.copy(), Xc.indptr.copy()), shape=Xc.shape)
    res.has_sorted_indices = True
    return res


def tf_idf_weight(X, smooth: bool = True, *, device: Optional[int] = None) -> sps.csr_matrix:
    """util.hpp:190-209 (bound at util.cpp:29-32): ``x * log(N / (df + smooth))``."""
    return _weight(X, _lib.WEIGHT_TF_IDF, 0.0, 0.0, bool(smooth), device)


def okapi_BM_25_weight(X, k1: float = 1.2, b: float = 0.75, *, device: Optional[int] = None) -> sps.csr_matrix:
    """util.hpp:158-188: ``idf * x (k1 + 1) / (x + k1 (1 - b + b dl / avgdl))``."""
    return _weight(X, _lib.WEIGHT_BM25, float(k1), float(b), True, device)


def _retrieve(score: np.ndarray, allowed_item_indices: Sequence[Sequence[int]], cutoff: int,
              n_threads: int, device: Optional[int]) -> List[List[Tuple[int, float]]]:
    if cutoff < 0 or n_threads < 0:
        raise TypeError("cutoff and n_threads must be non-negative (size_t).")
    score = np.ascontiguousarray(score)
    if score.ndim != 2:
        raise ValueError("score must be a 2-d array.")
    rows, n_items = score.shape
    lptr = np.zeros(len(allowed_item_indices) + 1, dtype=np.int64)
    for i, l in enumerate(allowed_item_indices):
        lptr[i + 1] = lptr[i] + len(l)
    litems = np.zeros(max(int(lptr[-1]), 1), dtype=np.int64)
    for i, l in enumerate(allowed_item_indices):
        litems[lptr[i]:lptr[i + 1]] = np.asarray(l, dtype=np.int64)
    # the reference clamps the list length to the candidate count (util.hpp:476-481): a cutoff
    # of n_items ("rank everything") needs n_items slots at most (above 2048 the device keeps
    # the selected lists in global scratch instead of LDS: rank_rows_kernel<..., BIG>)
    width = max(min(int(cutoff), int(n_items)), 0)
    out = np.full((rows, max(width, 1)), -1, dtype=np.int32)
    check(lib().irs_retrieve_recommend(
        C.c_int32(1 if score.dtype == np.float64 else 0), score.ctypes.data_as(C.c_void_p),
        C.c_int64(rows), C.c_int64(n_items), C.c_int64(len(allowed_item_indices)),
        ptr(lptr, C.c_int64), ptr(litems, C.c_int64), C.c_int64(width), C.c_int64(n_threads),
        C.c_int32(_lib.default_device() if device is None else device), ptr(out, C.c_int32)))
    result: List[List[Tuple[int, float]]] = []
    for r in range(rows):
        idx = out[r][out[r] >= 0] if width > 0 else out[r][:0]
        # the reference returns std::pair<int64_t, float>: scores are narrowed to float32
        vals = score[r, idx].astype(np.float32)
        result.append([(int(i), float(v)) for i, v in zip(idx, vals)])
    return result


def retrieve_recommend_from_score_f32(score, allowed_indices, cutoff: int, n_threads: int = 1,
                                      *, device: Optional[int] = None):
    """util.hpp:426-504 for float32 scores (bound as ``retrieve_recommend_from_score_f32``)."""
    return _retrieve(np.asarray(score, dtype=np.float32), allowed_indices, cutoff, n_threads,
                     device)


def retrieve_recommend_from_score_f64(score, allowed_indices, cutoff: int, n_threads: int = 1,
                                      *, device: Optional[int] = None):
    """util.hpp:426-504 for float64 scores (bound as ``retrieve_recommend_from_score_f64``)."""
    return _retrieve(np.asarray(score, dtype=np.float64), allowed_indices, cutoff, n_threads,
                     device)


def retrieve_recommend_from_score(score, allowed_item_indices, cutoff: int, n_threads: int = 1,
                                  *, device: Optional[int] = None):
    """irspack/utils/id_mapping.py:29-44: dispatch on the score dtype."""
    score = np.asarray(score)
    if score.dtype == np.float32:
        return retrieve_recommend_from_score_f32(score, allowed_item_indices, cutoff, n_threads,
                                                 device=device)
    if score.dtype == np.float64:
        return retrieve_recommend_from_score_f64(score, allowed_item_indices, cutoff, n_threads,
                                                 device=device)
    raise ValueError("Only float32 or float64 are allowed.")
