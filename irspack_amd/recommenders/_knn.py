"""Counterpart of the reference's nanobind module ``irspack.recommenders._knn``
(/root/reference/cpp_source/knn/wrapper.cpp:11-66), backed by ``libirspack_amd.so``
(irspack_amd/csrc/knn.hip).  Inputs are scipy sparse float64, CSR or CSC - a CSC matrix
(``X.T`` of the recommenders, knn.py:77-79) is handed over as it is with a layout flag, the
library regroups what it needs itself; results are CSR float64 with sorted indices
(``compute_similarity``) or CSC (``compute_W``).

Not in the reference: the keyword-only ``weighting=`` of the cosine / asymmetric-cosine computers
(``("TF_IDF", smooth)`` / ``("BM_25", k1, b)``): the feature weighting of knn.py:67-75 /
user_knn.py:62-72 applied on the device on the way in - to the matrix as stored, i.e.
``Computer(X.T, ..., weighting=w)`` is ``Computer(weight(X).T, ...)`` and ``Computer(X, ...,
weighting=w)`` is ``Computer(weight(X), ...)`` - so that ``learn()`` never builds the weighted
matrix on the host.
"""

import ctypes as C
from typing import Any, Optional, Tuple

import numpy as np
import scipy.sparse as sps

from .. import _lib
from .._lib import check, lib, ptr

_COSINE, _ASYMMETRIC, _JACCARD, _TVERSKY, _P3ALPHA, _RP3BETA = range(6)


class _Computer:
    _sim_type = _COSINE

    def _create(self, X: Any, shrinkage: float, alpha: float, beta: float, normalize: bool,
                n_threads: int, max_chunk_size: int, device: Optional[int],
                weighting: Optional[Tuple] = None) -> None:
        if n_threads < 0 or max_chunk_size < 0:
            raise TypeError("n_threads / max_chunk_size must be non-negative (size_t).")
        Xc, layout, indptr, indices, data = _lib.sparse_arrays(X, np.float64)
        self._N, self._n_features = int(Xc.shape[0]), int(Xc.shape[1])
        self._device = _lib.default_device() if device is None else int(device)
        # the stored matrix is the TRANSPOSE of X for a CSC input: the weighting's documents are its rows
        spec = _lib.KnnInputStruct(layout=layout, weighting=_lib.WEIGHT_NONE, smooth=1, reserved=0, k1=1.2, b=0.75)
        if weighting is not None:
            kind = str(weighting[0]).upper()
            if kind in ("TF_IDF", "TFIDF"):
                spec.weighting = _lib.WEIGHT_TF_IDF
                spec.smooth = 1 if (len(weighting) < 2 or weighting[1]) else 0
            elif kind in ("BM_25", "BM25"):
                spec.weighting = _lib.WEIGHT_BM25
                spec.k1 = float(weighting[1]) if len(weighting) > 1 else 1.2
                spec.b = float(weighting[2]) if len(weighting) > 2 else 0.75
            elif kind != "NONE":
                raise ValueError("weighting must be ('TF_IDF', smooth), ('BM_25', k1, b) or None.")
        h = C.c_void_p()
        check(
            lib().irs_knn_create(
                C.c_int32(self._sim_type), C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]),
                ptr(indptr, C.c_int64), ptr(indices, C.c_int32), ptr(data, C.c_double), C.byref(spec),
                C.c_double(shrinkage), C.c_double(alpha), C.c_double(beta),
                C.c_int32(1 if normalize else 0), C.c_int64(n_threads),
                C.c_int64(max_chunk_size), C.c_int32(self._device), C.byref(h),
            )
        )
        self._h: Optional[C.c_void_p] = h
        self.last_kernel_ms = 0.0
        self.last_macs = 0
        self.last_walked_macs = 0
        self.dense_block_rows = 0

    def __del__(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().irs_knn_destroy(h)
            except Exception:
                pass
            self._h = None

    # candidate slots (target rows x column tiles x top_k) one device call may hold: 64 M
    # (~0.8 GB of scratch); larger requests are cut into row batches
    _MAX_SLOT_ENTRIES = 1 << 26

    def _compute(self, X: Any, top_k: int, as_w: bool,
                 rows: Optional[Tuple[int, int]] = None, csc_zero_diagonal: bool = False):
        if top_k < 0:
            raise TypeError("top_k must be non-negative (size_t).")
        Xc, layout, indptr, indices, data = _lib.sparse_arrays(X, np.float64)
        rb, re = (0, Xc.shape[0]) if rows is None else rows
        n_tiles = max(1, -(-self._N // 16384))
        per_row = max(1, n_tiles * min(max(int(top_k), 1), self._N))
        batch = max(1, self._MAX_SLOT_ENTRIES // per_row)
        if re - rb > batch and csc_zero_diagonal:  # (huge catalogues: the host conversion of the stitched result)
            from ..utils import remove_diagonal

            return remove_diagonal(self._compute(X, top_k, as_w, rows)).tocsc()
        if re - rb > batch:  # rows are independent: stitch the batches
            if layout == _lib.LAYOUT_CSC:  # (regrouped once here, not once per batch inside the library)
                Xc = sps.csr_matrix(Xc)
                Xc.sort_indices()
            parts, ms, macs = [], 0.0, 0
            for b in range(rb, re, batch):
                parts.append(self._compute(Xc, top_k, as_w, (b, min(b + batch, re))))
                ms += self.last_kernel_ms
                macs += self.last_macs
            self.last_kernel_ms, self.last_macs = ms, macs
            res = sps.vstack(parts, format="csr")
            res.has_sorted_indices = True
            return res
        nnz = C.c_int64(0)
        check(
            lib().irs_knn_compute(
                self._h, C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]), ptr(indptr, C.c_int64),
                ptr(indices, C.c_int32), ptr(data, C.c_double), C.c_int32(layout), C.c_int64(top_k),
                C.c_int32(1 if as_w else 0), C.c_int64(rb), C.c_int64(re), C.byref(nnz),
            )
        )
        o_idx = np.empty(max(nnz.value, 1), dtype=np.int32)
        o_val = np.empty(max(nnz.value, 1), dtype=np.float64)
        if csc_zero_diagonal:
            o_ptr = np.empty(self._N + 1, dtype=np.int64)
            check(lib().irs_knn_fetch_csc(self._h, C.c_int64(rb), ptr(o_ptr, C.c_int64), ptr(o_idx, C.c_int32),
                                          ptr(o_val, C.c_double)))
        else:
            o_ptr = np.empty(re - rb + 1, dtype=np.int64)
            check(lib().irs_knn_fetch(self._h, ptr(o_ptr, C.c_int64), ptr(o_idx, C.c_int32),
                                      ptr(o_val, C.c_double)))
        ms = C.c_double(0)
        macs = C.c_int64(0)
        check(lib().irs_knn_last_stats(self._h, C.byref(ms), C.byref(macs)))
        self.last_kernel_ms, self.last_macs = float(ms.value), int(macs.value)
        walked, dense_rows = C.c_int64(0), C.c_int32(0)
        check(lib().irs_knn_last_walked(self._h, C.byref(walked), C.byref(dense_rows)))
        #: (measurement) multiply-adds added one by one / rows of the dense popular block
        self.last_walked_macs, self.dense_block_rows = int(walked.value), int(dense_rows.value)
        if csc_zero_diagonal:
            res = sps.csc_matrix((o_val[: nnz.value], o_idx[: nnz.value], o_ptr.astype(np.int32)),
                                 shape=(re - rb, self._N))
        else:
            res = sps.csr_matrix((o_val[: nnz.value], o_idx[: nnz.value], o_ptr),
                                 shape=(re - rb, self._N))
        res.has_sorted_indices = True
        return res


class _SimilarityComputer(_Computer):
    def compute_similarity(self, X: Any, top_k: int, *, rows: Optional[Tuple[int, int]] = None
                           ) -> sps.csr_matrix:
        """knn.hpp:43-83.  ``rows`` (keyword-only, not in the reference) restricts the
        call to a shard of target rows — rows are independent, so a multi-GPU run
        concatenates the shards."""
        return self._compute(X, top_k, False, rows)

    def compute_similarity_without_diagonal_csc(self, X: Any, top_k: int) -> sps.csc_matrix:
        """``remove_diagonal(self.compute_similarity(X, top_k)).tocsc()`` - the last two steps of the kNN
        recommenders' ``_learn`` (knn.py:78-80) - with the regrouping into columns and the zeroed
        (kept) diagonal done on the device before the result travels (``irs_knn_fetch_csc``).  The same
        matrix entry for entry.  Not in the reference."""
        if X.shape[0] != self._N:
            raise ValueError("X must be square")  # (remove_diagonal's check, util.hpp:213)
        return self._compute(X, top_k, False, None, csc_zero_diagonal=True)


class CosineSimilarityComputer(_SimilarityComputer):  # wrapper.cpp:12-19, similarities.hpp:6-47
    _sim_type = _COSINE

    def __init__(self, X: Any, shrinkage: float, normalize: bool, n_threads: int = 1,
                 max_chunk_size: int = 128, *, device: Optional[int] = None,
                 weighting: Optional[Tuple] = None) -> None:
        self._create(X, shrinkage, 0.0, 0.0, bool(normalize), n_threads, max_chunk_size, device, weighting)


class JaccardSimilarityComputer(_SimilarityComputer):  # wrapper.cpp:21-29
    _sim_type = _JACCARD

    def __init__(self, X: Any, shrinkage: float, n_threads: int = 1, max_chunk_size: int = 128,
                 *, device: Optional[int] = None) -> None:
        self._create(X, shrinkage, 0.0, 0.0, False, n_threads, max_chunk_size, device)


class TverskyIndexComputer(_SimilarityComputer):  # wrapper.cpp:31-40
    _sim_type = _TVERSKY

    def __init__(self, X: Any, shrinkage: float, alpha: float, beta: float, n_threads: int = 1,
                 max_chunk_size: int = 128, *, device: Optional[int] = None) -> None:
        self._create(X, shrinkage, alpha, beta, False, n_threads, max_chunk_size, device)


class AsymmetricSimilarityComputer(_SimilarityComputer):  # wrapper.cpp:42-51
    _sim_type = _ASYMMETRIC

    def __init__(self, X: Any, shrinkage: float, alpha: float, n_threads: int = 1,
                 max_chunk_size: int = 128, *, device: Optional[int] = None,
                 weighting: Optional[Tuple] = None) -> None:
        self._create(X, shrinkage, alpha, 0.0, False, n_threads, max_chunk_size, device, weighting)


class P3alphaComputer(_Computer):  # wrapper.cpp:53-58, similarities.hpp:186-251
    _sim_type = _P3ALPHA

    def __init__(self, X: Any, alpha: float = 0, n_threads: int = 1, max_chunk_size: int = 128,
                 *, device: Optional[int] = None) -> None:
        self._create(X, 0.0, alpha, 0.0, False, n_threads, max_chunk_size, device)

    def compute_W(self, X: Any, top_k: int) -> sps.csc_matrix:
        return self._compute(X, top_k, True).T.tocsc()


class RP3betaComputer(_Computer):  # wrapper.cpp:60-65, similarities.hpp:253-336
    _sim_type = _RP3BETA

    def __init__(self, X: Any, alpha: float = 0, beta: float = 0, n_threads: int = 1,
                 max_chunk_size: int = 128, *, device: Optional[int] = None) -> None:
        self._create(X, 0.0, alpha, beta, False, n_threads, max_chunk_size, device)

    def compute_W(self, X: Any, top_k: int) -> sps.csc_matrix:
        return self._compute(X, top_k, True).T.tocsc()
