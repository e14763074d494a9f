"""Counterpart of the reference's nanobind module ``irspack.evaluation._core_evaluator``
(/root/reference/cpp_source/evaluator.cpp:441-484), backed by ``libirspack_amd.so``.

``EvaluatorCore.get_metrics_f32/_f64`` rank the score block on the GPU
(irspack_amd/csrc/evaluator.hip).  ``Metrics`` is the plain accumulator of
evaluator.cpp:49-179: merging and the ``as_dict`` summary (entropy / gini over the
item histogram) are host bookkeeping here as they are in the reference.
"""

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sps

from .. import _lib
from .._lib import MetricsStruct, check, lib, ptr


class Metrics:
    """evaluator.cpp:49-179."""

    def __init__(self, n_item: int) -> None:
        if int(n_item) < 0:
            raise TypeError("n_item must be non-negative (size_t).")
        self.n_item = int(n_item)
        self.valid_user = 0
        self.total_user = 0
        self.hit = 0.0
        self.recall = 0.0
        self.ndcg = 0.0
        self.precision = 0.0
        self.map = 0.0
        self.item_cnt = np.zeros(self.n_item, dtype=np.int64)

    @classmethod
    def _from_struct(cls, n_item: int, st: MetricsStruct, item_cnt: np.ndarray) -> "Metrics":
        m = cls(n_item)
        m.valid_user = int(st.valid_user)
        m.total_user = int(st.total_user)
        m.hit, m.recall, m.ndcg = float(st.hit), float(st.recall), float(st.ndcg)
        m.precision, m.map = float(st.precision), float(st.map)
        m.item_cnt = item_cnt
        return m

    def merge(self, other: "Metrics") -> None:  # :76-85
        self.hit += other.hit
        self.recall += other.recall
        self.ndcg += other.ndcg
        self.total_user += other.total_user
        self.valid_user += other.valid_user
        self.item_cnt = self.item_cnt + other.item_cnt
        self.precision += other.precision
        self.map += other.map

    def as_dict(self) -> Dict[str, float]:  # :87-123
        cnt = np.sort(self.item_cnt)
        n = cnt.shape[0]
        total_item = float(self.item_cnt.sum())
        nz = cnt > 0
        appeared = float(nz.sum())
        entropy = 0.0
        gini = 0.0
        if appeared > 0:
            c = cnt[nz].astype(np.float64)
            p = c / total_item
            # sequential accumulation in ascending count order, like the reference loop
            for pi in p:
                entropy += -np.log(pi) * pi
            idx = np.flatnonzero(nz).astype(np.int64)
            for i, ci in zip(idx, cnt[nz]):
                gini += float((2 * int(i) - n + 1) * int(ci))
        if total_item > 0:
            gini /= n * total_item
        denom = self.valid_user if self.valid_user > 0 else 1
        return {
            "total_user": float(self.total_user),
            "valid_user": float(self.valid_user),
            "n_items": float(self.n_item),
            "hit": self.hit / denom,
            "ndcg": self.ndcg / denom,
            "recall": self.recall / denom,
            "map": self.map / denom,
            "precision": self.precision / denom,
            "appeared_item": appeared,
            "entropy": float(entropy),
            "gini_index": float(gini),
        }

    def _update(self, rec: Sequence[int], gt: set, recall_with_cutoff: bool) -> None:
        """Metrics::update (:127-166); used by evaluate_list_vs_list only."""
        n_gt, n_rec = len(gt), len(rec)
        self.valid_user += 1
        if n_rec == 0:
            return
        disc = 1.0 / np.log2(2.0 + np.arange(max(n_rec, 1)))
        dcg = 0.0
        idcg = 0.0
        for i in range(min(n_gt, n_rec)):
            idcg += disc[i]
        ap = 0.0
        cum_hit = 0
        for i, r in enumerate(rec):
            self.item_cnt[r] += 1
            if r in gt:
                dcg += disc[i]
                cum_hit += 1
                ap += cum_hit / (i + 1)
        if cum_hit > 0:
            self.hit += 1
        self.precision += cum_hit / n_rec
        with np.errstate(divide="ignore", invalid="ignore"):
            self.recall += float(np.float64(cum_hit) / np.float64(
                (n_rec if n_gt > n_rec else n_gt) if recall_with_cutoff else n_gt))
            self.ndcg += float(np.float64(dcg) / np.float64(idcg))
            self.map += float(np.float64(ap) / np.float64(n_gt))


def _ragged(lists: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    ptr_ = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        ptr_[i + 1] = ptr_[i] + len(l)
    flat = np.empty(max(int(ptr_[-1]), 1), dtype=np.int64)
    pos = 0
    for l in lists:
        for x in l:
            if int(x) < 0:
                raise TypeError("recommendable item indices must be non-negative (size_t).")
            flat[pos] = int(x)
            pos += 1
    return ptr_, flat


class MaskRows:
    """The NONZERO pattern of a mask matrix (``scores[mask.nonzero()] = -inf``,
    evaluator.py:389 / :432: stored zeros do not mask) prepared once for row-range slicing:
    ``rows(b, e)`` hands ``irs_eval_get_metrics_masked`` views of the row pointers and the
    column indices of rows ``[b, e)`` without copying either."""

    def __init__(self, mask: Any, n_cols: int) -> None:
        m = mask if sps.isspmatrix_csr(mask) else sps.csr_matrix(mask)
        if m.shape[1] != n_cols:
            raise ValueError(f"mask must have {n_cols} columns, got {m.shape[1]}.")
        indptr, indices = m.indptr, m.indices
        if m.nnz and not np.all(m.data != 0):
            keep = m.data != 0
            row_of = np.repeat(np.arange(m.shape[0]), np.diff(indptr))[keep]
            indices = indices[keep]
            indptr = np.zeros(m.shape[0] + 1, dtype=np.int64)
            np.cumsum(np.bincount(row_of, minlength=m.shape[0]), out=indptr[1:])
        self.n_rows = int(m.shape[0])
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        if self.indices.size and (self.indices.min() < 0 or self.indices.max() >= n_cols):
            raise ValueError("mask column index out of range.")

    def rows(self, begin: int, end: int) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        if begin < 0 or end > self.n_rows or end < begin:
            raise ValueError("mask rows out of range.")
        p = self.indptr[begin:end + 1]
        if p[-1] == p[0]:
            return None, None
        return p, self.indices[p[0]:p[-1]]


class EvaluatorCore:
    """``EvaluatorCore(ground_truth, recommendable)`` — evaluator.cpp:181-374, :449-480."""

    def __init__(self, ground_truth: Any, recommendable: List[List[int]], *,
                 device: Optional[int] = None) -> None:
        X, indptr, indices, _ = _lib.csr_arrays(ground_truth, np.float64)
        self._X = sps.csr_matrix(X, dtype=np.float64)
        self._recommendable = [list(map(int, l)) for l in recommendable]
        self.n_users, self.n_items = int(X.shape[0]), int(X.shape[1])
        self._device = _lib.default_device() if device is None else int(device)
        rp, rf = _ragged(self._recommendable)
        h = C.c_void_p()
        check(
            lib().irs_eval_create(
                C.c_int64(self.n_users), C.c_int64(self.n_items), ptr(indptr, C.c_int64),
                ptr(indices, C.c_int32), C.c_int64(len(self._recommendable)),
                ptr(rp, C.c_int64), ptr(rf, C.c_int64), C.c_int32(self._device), C.byref(h),
            )
        )
        self._h: Optional[C.c_void_p] = h
        # the reference sorts the stored lists (:193-194)
        self._recommendable = [sorted(l) for l in self._recommendable]

    def __del__(self) -> None:
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().irs_eval_destroy(h)
            except Exception:
                pass
            self._h = None

    def _get(self, scores: np.ndarray, is_f64: bool, cutoff: int, offset: int, n_threads: int,
             recall_with_cutoff: bool) -> Metrics:
        want = np.float64 if is_f64 else np.float32
        if not isinstance(scores, np.ndarray) or scores.dtype != want or scores.ndim != 2:
            raise TypeError(f"score_array must be a 2-D {np.dtype(want).name} ndarray.")
        if scores.shape[1] != self.n_items:
            raise ValueError("score_array.shape[1] must equal n_items.")
        if cutoff < 0 or offset < 0 or n_threads < 0:
            raise TypeError("cutoff / offset / n_threads must be non-negative (size_t).")
        scores = np.ascontiguousarray(scores)
        st = MetricsStruct()
        cnt = np.zeros(self.n_items, dtype=np.int64)
        check(
            lib().irs_eval_get_metrics(
                self._h, C.c_int32(1 if is_f64 else 0), scores.ctypes.data_as(C.c_void_p),
                C.c_int64(scores.shape[0]), C.c_int64(cutoff), C.c_int64(offset),
                C.c_int64(n_threads), C.c_int32(1 if recall_with_cutoff else 0), C.byref(st),
                ptr(cnt, C.c_int64),
            )
        )
        return Metrics._from_struct(self.n_items, st, cnt)

    def get_metrics_f64(self, score_array, cutoff, offset, n_threads, recall_with_cutoff=False):
        return self._get(score_array, True, cutoff, offset, n_threads, recall_with_cutoff)

    def get_metrics_f32(self, score_array, cutoff, offset, n_threads, recall_with_cutoff=False):
        return self._get(score_array, False, cutoff, offset, n_threads, recall_with_cutoff)

    def get_metrics_masked(self, score_array: np.ndarray, mask: Optional["MaskRows"],
                           mask_begin: int, cutoffs: Sequence[int], offset: int, n_threads: int,
                           recall_with_cutoff: bool = False) -> List[Metrics]:
        """One block of the caller loops of evaluation/evaluator.py:371-393 / :417-438 in one
        device call (``irs_eval_get_metrics_masked``, not in the reference): the block is
        uploaded once, rows ``mask_begin ..`` of ``mask`` are set to ``-inf`` on the device
        (``score_array`` itself is never written) and the block is ranked once per cutoff.
        Returns one ``Metrics`` per cutoff, equal to masking on the host and calling
        ``get_metrics_f32/_f64`` per cutoff."""
        if not isinstance(score_array, np.ndarray) or score_array.ndim != 2 or \
                score_array.dtype not in (np.dtype("float32"), np.dtype("float64")):
            raise TypeError("score_array must be a 2-D float32 or float64 ndarray.")
        if score_array.shape[1] != self.n_items:
            raise ValueError("score_array.shape[1] must equal n_items.")
        if offset < 0 or n_threads < 0 or any(int(c) < 0 for c in cutoffs):
            raise TypeError("cutoff / offset / n_threads must be non-negative (size_t).")
        scores = np.ascontiguousarray(score_array)
        rows, nc = scores.shape[0], len(cutoffs)
        cut = np.asarray([int(c) for c in cutoffs], dtype=np.int64)
        sts = (MetricsStruct * max(nc, 1))()
        cnt = np.zeros((max(nc, 1), self.n_items), dtype=np.int64)
        mp, mi = (None, None) if mask is None else mask.rows(mask_begin, mask_begin + rows)
        check(
            lib().irs_eval_get_metrics_masked(
                self._h, C.c_int32(1 if scores.dtype == np.float64 else 0),
                scores.ctypes.data_as(C.c_void_p), C.c_int64(rows),
                None if mp is None else ptr(mp, C.c_int64),
                None if mi is None else ptr(mi, C.c_int32), C.c_int32(nc), ptr(cut, C.c_int64),
                C.c_int64(offset), C.c_int64(n_threads), C.c_int32(1 if recall_with_cutoff else 0),
                sts, ptr(cnt, C.c_int64),
            )
        )
        return [Metrics._from_struct(self.n_items, sts[i], cnt[i]) for i in range(nc)]

    def get_metrics_ials(self, trainer, begin: int, end: int, mask: Optional[sps.csr_matrix],
                         cutoff: int, offset: int, recall_with_cutoff: bool = False) -> Metrics:
        """Fused device path (not in the reference): score, mask and rank users
        [begin, end) of an ``irspack_amd`` IALSTrainer without leaving HBM."""
        st = MetricsStruct()
        cnt = np.zeros(self.n_items, dtype=np.int64)
        # The mask (usually the training interactions) is converted and uploaded once and stays
        # on the device while calls keep passing the same matrix object over the same users
        # (a tuning loop evaluates hundreds of times against one mask).  The object is held
        # here so that its identity stays valid; a mask edited in place needs a new object.
        key = None if mask is None else (id(mask), mask.shape, mask.nnz, begin, end,
                                         self._mask_fingerprint(mask))
        if key != getattr(self, "_mask_key", None):
            if mask is not None:
                M, mp, mi, md = _lib.csr_arrays(mask, np.float32)
                if M.shape != (end - begin, self.n_items):
                    raise ValueError("mask must have shape (end - begin, n_items).")
                if md.size and not md.all():
                    # the reference masks ``mask.nonzero()`` (evaluator.py:426-432): stored
                    # zeros are not masked
                    keep = md != 0
                    rows = np.repeat(np.arange(M.shape[0]), np.diff(mp))[keep]
                    mi = np.ascontiguousarray(mi[keep])
                    mp = np.zeros(M.shape[0] + 1, dtype=np.int64)
                    np.cumsum(np.bincount(rows, minlength=M.shape[0]), out=mp[1:])
                if mi.size == 0:
                    mi = np.zeros(1, dtype=np.int32)
                check(lib().irs_eval_cache_mask(self._h, C.c_int64(end - begin),
                                                ptr(mp, C.c_int64), ptr(mi, C.c_int32)))
            else:
                check(lib().irs_eval_cache_mask(self._h, C.c_int64(0), None, None))
            self._mask_key, self._mask_ref = key, mask
        mp_arg, mi_arg = None, None
        check(
            lib().irs_eval_get_metrics_ials(
                self._h, trainer._h, C.c_int64(begin), C.c_int64(end), mp_arg, mi_arg,
                C.c_int64(cutoff), C.c_int64(offset), C.c_int32(1 if recall_with_cutoff else 0),
                C.byref(st), ptr(cnt, C.c_int64),
            )
        )
        return Metrics._from_struct(self.n_items, st, cnt)

    def get_metrics_similarity(self, X: sps.csr_matrix, W: Any, begin: int, end: int,
                               mask: Optional["MaskRows"], mask_begin: int, cutoffs: Sequence[int],
                               offset: int, recall_with_cutoff: bool = False) -> List[Metrics]:
        """Device path for SIMILARITY models (not in the reference): users ``[begin, end)`` of the
        profile matrix ``X`` (the training matrix for item-kNN / P3alpha / RP3beta with ``W`` item x item;
        the user-user weights for user-kNN with ``W`` = the training matrix) are scored as ``X[u] @ W`` on the device - entry by entry in the order
        and with the rounding of scipy's sparse product, so the block is ``X[begin:end].dot(W)`` bit
        for bit -, rows ``mask_begin ..`` of ``mask`` are set to ``-inf`` and the block is ranked
        once per cutoff: what ``Evaluator`` does per 128-user block with
        ``model.get_score_block`` (evaluator.py:417-438), without the scores crossing PCIe.
        ``offset``: ground-truth row of user ``begin``.  One ``Metrics`` per cutoff."""
        if offset < 0 or any(int(c) < 0 for c in cutoffs):
            raise TypeError("cutoff / offset must be non-negative (size_t).")
        Xc, xp, xi, xd = _lib.csr_arrays(X, np.float64)
        Wc, wp, wi, wd = _lib.csr_arrays(W, np.float64)  # (a CSC matrix is regrouped into rows here)
        if Wc.shape != (Xc.shape[1], self.n_items):
            raise ValueError("W must be X.shape[1] x n_items.")
        rows, nc = end - begin, len(cutoffs)
        cut = np.asarray([int(c) for c in cutoffs], dtype=np.int64)
        sts = (MetricsStruct * max(nc, 1))()
        cnt = np.zeros((max(nc, 1), self.n_items), dtype=np.int64)
        mp, mi = (None, None) if mask is None else mask.rows(mask_begin, mask_begin + rows)
        one_i, one_d = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.float64)
        check(
            lib().irs_eval_get_metrics_similarity(
                self._h, C.c_int64(begin), C.c_int64(end), C.c_int64(Xc.shape[0]), C.c_int64(Xc.shape[1]),
                ptr(xp, C.c_int64),
                ptr(xi if xi.size else one_i, C.c_int32), ptr(xd if xd.size else one_d, C.c_double),
                ptr(wp, C.c_int64), ptr(wi if wi.size else one_i, C.c_int32),
                ptr(wd if wd.size else one_d, C.c_double),
                None if mp is None else ptr(mp, C.c_int64), None if mi is None else ptr(mi, C.c_int32),
                C.c_int32(nc), ptr(cut, C.c_int64), C.c_int64(offset),
                C.c_int32(1 if recall_with_cutoff else 0), sts, ptr(cnt, C.c_int64),
            )
        )
        return [Metrics._from_struct(self.n_items, sts[i], cnt[i]) for i in range(nc)]

    def last_call_stats(self) -> dict:
        """What the last ``get_metrics_ials`` call did on the device (measurement only):
        which path ran, how many 64 x 64 score tiles it computed of how many, and the rows it
        had to rank from their full score row."""
        from .._lib import EvalStatsStruct

        st = EvalStatsStruct()
        check(lib().irs_eval_last_stats(self._h, C.byref(st)))
        names = {0: "two_pass", 1: "emit", 2: "emit_bounded"}
        return {"path": names.get(st.path, str(st.path)), "hard_rows": int(st.hard_rows),
                "tiles_total": int(st.tiles_total), "tiles_scored": int(st.tiles_scored),
                "sample_items": int(st.sample_items), "call_ms": float(st.call_ms),
                "device_span_ms": float(st.device_span_ms)}

    #: ``True`` (the default): EVERY byte of the mask is hashed on every ``get_metrics_ials``
    #: call (xxhash when installed, else CRC-32: ~15 ms for 20 M entries, next to a 2 ms device
    #: pass), so a mask edited in place is noticed and uploaded again.  ``False`` trades that for a
    #: sampled fingerprint (row-pointer sum + CRC of 1024 strided samples: microseconds) for callers
    #: that treat their masks as immutable, e.g. a tuning loop; ``invalidate_mask()`` then forces a
    #: re-upload after an in-place edit.
    strict_mask_fingerprint = True

    def invalidate_mask(self) -> None:
        """Drop the device-resident mask: the next ``get_metrics_ials`` call converts and uploads
        its mask again (call this after editing a mask matrix in place)."""
        self._mask_key = None
        self._mask_ref = None

    @classmethod
    def _mask_fingerprint(cls, mask: sps.spmatrix) -> int:
        """Content check for the device-resident mask, run on every call.  Default: a hash of all
        of its row pointers, column indices and values.  With ``strict_mask_fingerprint = False``:
        the sum of the row pointers and a CRC of 1024 strided samples of each array - an in-place
        edit that keeps nnz, the pointer sum and every sampled entry then goes unnoticed."""
        import zlib

        m = mask if sps.isspmatrix_csr(mask) else sps.csr_matrix(mask)
        if cls.strict_mask_fingerprint:
            # every byte of the three arrays, hashed by the library on several host threads
            # (irs_fingerprint: ~1 ms for the 16 M-entry mask of the ML-20M shape; one thread of
            # xxh3 took 4.5 ms of a 6.1 ms call)
            h = 0
            for a in (m.indptr, m.indices, m.data):
                a = np.ascontiguousarray(a)
                out = C.c_uint64(0)
                check(lib().irs_fingerprint(a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes),
                                            C.c_uint64(h), C.byref(out)))
                h = int(out.value)
            return h
        h = int(m.indptr.sum(dtype=np.int64)) & 0xFFFFFFFF
        for a in (m.indptr, m.indices, m.data):
            h = zlib.crc32(np.ascontiguousarray(a[:: max(1, a.size // 1024)]).tobytes(), h)
        return h

    def get_ground_truth(self) -> sps.csr_matrix:
        return self._X.copy()

    def cache_X_as_set(self, n_threads: int) -> None:
        if n_threads <= 0:
            raise ValueError("n_threads must be strictly positive.")

    def __getstate__(self) -> tuple:
        return (self._X, self._recommendable)

    def __setstate__(self, state: tuple) -> None:
        self.__init__(state[0], state[1])  # type: ignore[misc]


def evaluate_list_vs_list(recommendations: List[List[int]], ground_truths: List[List[int]],
                          n_items: int, n_threads: int) -> Metrics:
    """evaluator.cpp:376-431 — list bookkeeping, no scoring or ranking involved."""
    if len(recommendations) != len(ground_truths):
        raise ValueError("recommendation array and ground_truth array has different size.")
    for rec in recommendations:
        for r in rec:
            if not (0 <= int(r) < n_items):
                raise ValueError("found recommendation index larger than n_items.")
    for gt in ground_truths:
        for g in gt:
            if not (0 <= int(g) < n_items):
                raise ValueError("found ground truth index larger than n_items.")
    overall = Metrics(n_items)
    for rec, gt in zip(recommendations, ground_truths):
        overall.total_user += 1
        overall._update([int(r) for r in rec], set(int(g) for g in gt), False)
    return overall
