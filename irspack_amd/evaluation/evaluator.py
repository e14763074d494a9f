"""``Evaluator`` / ``EvaluatorWithColdUser`` counterparts: the caller contract of the
reference's ``irspack/evaluation/evaluator.py:98-183, 196-441, 444-657`` on top of the GPU
``EvaluatorCore``: block loop over users (``mb_size``), ``get_score_block`` with a
``get_score`` fallback, seen-item masking with ``-inf``, one ranking pass per cutoff,
``Metrics.merge``, the ``catalog_coverage`` post-processing, the score-matrix / score-chunk
entry points (:229-398) and the cold-user evaluator with feature-only items (:444-657).

Two things are done differently from the reference's Python, with the same results:

* A host score block is masked ON THE DEVICE: ``EvaluatorCore.get_metrics_masked`` uploads the
  block once (the upload is the copy the reference makes before masking, :387), sets the mask's
  nonzero entries to ``-inf`` there and ranks once per cutoff.  Blocks returned by a model are
  therefore not written either (the reference's ``scores[mask.nonzero()] = -inf`` at :432 writes
  into them).
* When the model is an ``irspack_amd`` ``IALSRecommender`` living on the evaluator's device,
  ``fused=True`` (the default) scores, masks and ranks on the GPU without materialising the
  dense block at all (``irs_eval_get_metrics_ials``); tests/test_gpu_evaluator.py.
"""

import enum
import warnings
from typing import Any, Dict, Iterable, List, Optional, Union

import numpy as np
import scipy.sparse as sps

from .._threading import get_n_threads
from ._core_evaluator import EvaluatorCore, MaskRows, Metrics

METRIC_NAMES = ["hit", "recall", "ndcg", "map", "precision", "gini_index", "entropy",
                "appeared_item", "catalog_coverage"]

_SCORE_DTYPES = (np.dtype("float32"), np.dtype("float64"))


class TargetMetric(enum.Enum):
    ndcg = enum.auto()
    recall = enum.auto()
    hit = enum.auto()
    map = enum.auto()
    precision = enum.auto()


class Evaluator:
    """evaluator.py:37-441."""

    n_users: int
    n_items: int
    masked_interactions: Optional[sps.csr_matrix]

    def __init__(self, ground_truth: Any, offset: int = 0, cutoff: int = 10,
                 target_metric: str = "ndcg", recommendable_items: Optional[List[int]] = None,
                 per_user_recommendable_items: Union[None, List[List[int]], Any] = None,
                 masked_interactions: Optional[Any] = None, n_threads: Optional[int] = None,
                 recall_with_cutoff: bool = False, mb_size: int = 128, fused: bool = True,
                 device: Optional[int] = None) -> None:
        ground_truth = sps.csr_matrix(ground_truth).astype(np.float64)  # evaluator.py:114-115
        ground_truth.sort_indices()
        if recommendable_items is None:
            if per_user_recommendable_items is None:
                rec_arg: List[List[int]] = []
            else:
                if sps.issparse(per_user_recommendable_items):
                    m = sps.csr_matrix(per_user_recommendable_items)
                    rec_arg = [[int(j) for j in m[i].nonzero()[1]] for i in range(m.shape[0])]
                else:
                    rec_arg = per_user_recommendable_items
                if len(rec_arg) != ground_truth.shape[0]:
                    raise ValueError(
                        "ground_truth and per_user_recommendable_items have inconsistent shapes.")
        else:
            rec_arg = [recommendable_items]
        self.core = EvaluatorCore(ground_truth, rec_arg, device=device)
        if not rec_arg:
            self.n_recommendable_items = ground_truth.shape[1]
        elif len(rec_arg) == 1:
            self.n_recommendable_items = len(rec_arg[0])
        else:
            self.n_recommendable_items = len({i for l in rec_arg for i in l})
        self.offset = offset
        self.n_users = ground_truth.shape[0]
        self.n_items = ground_truth.shape[1]
        self.n_cold_items = 0
        self.target_metric = TargetMetric[target_metric]
        self.cutoff = cutoff
        self.target_metric_name = f"{self.target_metric.name}@{self.cutoff}"
        self.n_threads = get_n_threads(n_threads)
        self.mb_size = mb_size
        self.fused = fused
        if masked_interactions is None:
            self.masked_interactions = None
        else:
            if masked_interactions.shape != ground_truth.shape:
                raise ValueError("ground_truth and masked_interactions have different shapes. ")
            self.masked_interactions = sps.csr_matrix(masked_interactions)
        self.recall_with_cutoff = recall_with_cutoff

    def _get_metrics(self, scores: np.ndarray, cutoff: int, ground_truth_begin: int) -> Metrics:
        # evaluator.py:163-183
        if scores.dtype == np.float64:
            return self.core.get_metrics_f64(scores, cutoff, ground_truth_begin, self.n_threads,
                                             self.recall_with_cutoff)
        elif scores.dtype == np.float32:
            return self.core.get_metrics_f32(scores, cutoff, ground_truth_begin, self.n_threads,
                                             self.recall_with_cutoff)
        raise ValueError("score must be either float32 or float64.")

    def _merge_block(self, metrics: List[Metrics], scores: np.ndarray, mask: Optional[MaskRows],
                     mask_begin: int, cutoffs: List[int], ground_truth_begin: int) -> None:
        """Mask + rank one host block for every cutoff (evaluator.py:389-391 / :432-439) and
        merge into the running accumulators."""
        if scores.dtype not in _SCORE_DTYPES:
            raise ValueError("score must be either float32 or float64.")
        for acc, m in zip(metrics, self.core.get_metrics_masked(
                scores, mask, mask_begin, cutoffs, ground_truth_begin, self.n_threads,
                self.recall_with_cutoff)):
            acc.merge(m)

    def get_target_score(self, model: Any) -> float:
        return self.get_score(model)[self.target_metric.name]

    def get_score(self, model: Any) -> Dict[str, float]:
        return self._get_scores_as_list(model, [self.cutoff])[0]

    def _named(self, cutoffs: List[int], scores: List[Dict[str, float]]) -> Dict[str, float]:
        result: Dict[str, float] = {}
        for cutoff, score in zip(cutoffs, scores):
            for name in METRIC_NAMES:
                result[f"{name}@{cutoff}"] = score[name]
        return result

    def get_scores(self, model: Any, cutoffs: List[int]) -> Dict[str, float]:
        return self._named(cutoffs, self._get_scores_as_list(model, cutoffs))

    # -- score matrix / score chunks (evaluator.py:229-324) ---------------------------------
    def get_score_from_score_matrix(self, scores: np.ndarray) -> Dict[str, float]:
        return self._get_scores_from_score_matrix_as_list(scores, [self.cutoff])[0]

    def get_scores_from_score_matrix(self, scores: np.ndarray, cutoffs: List[int]) -> Dict[str, float]:
        return self._named(cutoffs, self._get_scores_from_score_matrix_as_list(scores, cutoffs))

    def get_score_from_score_chunks(self, score_chunks: Iterable[np.ndarray]) -> Dict[str, float]:
        return self._get_scores_from_score_chunks_as_list(score_chunks, [self.cutoff])[0]

    def get_scores_from_score_chunks(self, score_chunks: Iterable[np.ndarray],
                                     cutoffs: List[int]) -> Dict[str, float]:
        return self._named(cutoffs, self._get_scores_from_score_chunks_as_list(score_chunks, cutoffs))

    def _metrics_as_dict(self, metrics: Metrics) -> Dict[str, float]:
        # evaluator.py:326-334
        result = metrics.as_dict()
        if self.n_recommendable_items:
            result["catalog_coverage"] = result["appeared_item"] / self.n_recommendable_items
        else:
            result["catalog_coverage"] = float("nan")
        return result

    def _get_score_matrix_mask(self) -> Optional[sps.csr_matrix]:
        return self.masked_interactions

    def _mask_rows(self, mask: Optional[Any]) -> Optional[MaskRows]:
        """``MaskRows`` of a mask matrix, kept while calls pass the same object (the nonzero
        pattern of a 20 M-entry mask takes longer to prepare than a block to rank)."""
        if mask is None:
            return None
        held = getattr(self, "_mask_rows_held", None)
        key = (id(mask), mask.shape, mask.nnz, EvaluatorCore._mask_fingerprint(mask))
        if held is None or held[0] != key:
            held = (key, mask, MaskRows(mask, self.n_items))
            self._mask_rows_held = held
        return held[2]

    #: rows per device call of the score-MATRIX entry points: at least this many (the reference
    #: walks the matrix ``mb_size`` = 128 rows at a time, :363-367; 128 waves do not fill 256 compute
    #: units, and the chunking changes nothing but the order of the float64 sums)
    score_matrix_rows_per_call = 4096

    def _get_scores_from_score_matrix_as_list(self, scores: np.ndarray,
                                              cutoffs: List[int]) -> List[Dict[str, float]]:
        # evaluator.py:339-369
        if not isinstance(scores, np.ndarray) or scores.ndim != 2 or \
                scores.shape != (self.n_users, self.n_items):
            raise ValueError(f"score matrix must have shape ({self.n_users}, {self.n_items}), "
                             f"but got {getattr(scores, 'shape', None)}.")
        if scores.dtype not in _SCORE_DTYPES:
            raise ValueError("score matrix must have dtype float32 or float64.")
        per_row = max(self.n_items, 1) * scores.dtype.itemsize
        step = max(int(self.mb_size), min(self.score_matrix_rows_per_call,
                                          max(1, (1 << 30) // per_row)))
        chunks = (scores[b:b + step] for b in range(0, self.n_users, step))
        return self._get_scores_from_score_chunks_as_list(chunks, cutoffs)

    def _get_scores_from_score_chunks_as_list(self, score_chunks: Iterable[np.ndarray],
                                              cutoffs: List[int]) -> List[Dict[str, float]]:
        # evaluator.py:371-398
        mask = self._mask_rows(self._get_score_matrix_mask())
        metrics = [Metrics(self.n_items) for _ in cutoffs]
        chunk_start = 0
        for score_chunk in score_chunks:
            if not isinstance(score_chunk, np.ndarray) or score_chunk.ndim != 2:
                raise ValueError("each score chunk must be a 2-D ndarray, got "
                                 f"{type(score_chunk).__name__}.")
            if score_chunk.shape[1] != self.n_items:
                raise ValueError(f"score chunk must have n_items={self.n_items} columns, "
                                 f"got {score_chunk.shape[1]}.")
            if score_chunk.dtype not in _SCORE_DTYPES:
                raise ValueError("score chunk must have dtype float32 or float64.")
            chunk_end = chunk_start + score_chunk.shape[0]
            if chunk_end > self.n_users:
                raise ValueError("score chunks supplied more rows than the evaluator's "
                                 f"n_users={self.n_users}: processed {chunk_end} rows.")
            if score_chunk.shape[0] == 0:
                continue
            # the caller's array is only read: the mask is applied to the device copy
            self._merge_block(metrics, score_chunk, mask, chunk_start, cutoffs, chunk_start)
            chunk_start = chunk_end
        if chunk_start != self.n_users:
            raise ValueError("score chunks did not cover the evaluator's "
                             f"n_users={self.n_users} rows: processed {chunk_start} rows.")
        return [self._metrics_as_dict(m) for m in metrics]

    # -- model evaluation (evaluator.py:400-441) --------------------------------------------
    def _fused_trainer(self, model: Any):
        if not self.fused:
            return None
        trainer = getattr(getattr(model, "trainer", None), "core_trainer", None)
        if trainer is None or not hasattr(trainer, "_h"):
            return None
        if getattr(trainer, "_device", None) != self.core._device:
            return None
        return trainer

    def _get_scores_as_list(self, model: Any, cutoffs: List[int]) -> List[Dict[str, float]]:
        if self.offset + self.n_users > model.n_users:
            raise ValueError("evaluator offset + n_users exceeds the model's n_users.")
        if self.n_items != model.n_items:
            raise ValueError("The model and evaluator assume different n_items.")
        metrics = [Metrics(self.n_items) for _ in cutoffs]
        block_start, block_end = self.offset, self.offset + self.n_users
        trainer = self._fused_trainer(model)
        if trainer is not None:
            if self.masked_interactions is not None:
                mask = self.masked_interactions
            elif block_start == 0 and block_end == model.X_train_all.shape[0]:
                mask = model.X_train_all  # the whole matrix: no copy, and the device copy is reused
            else:
                # the slice is kept: a new object per call would re-upload the mask every time
                # (keyed on the content fingerprint too: an edited training matrix gets a new slice)
                key = (id(model.X_train_all), block_start, block_end,
                       self.core._mask_fingerprint(model.X_train_all))
                if getattr(self, "_mask_slice_key", None) != key:
                    self._mask_slice_key = key
                    self._mask_slice = (model.X_train_all, model.X_train_all[block_start:block_end])
                mask = self._mask_slice[1]
            for i, c in enumerate(cutoffs):
                metrics[i].merge(self.core.get_metrics_ials(trainer, block_start, block_end, mask, c,
                                                            0, self.recall_with_cutoff))
            return [self._metrics_as_dict(m) for m in metrics]
        if self.masked_interactions is None:
            mask, mask_shift = self._mask_rows(model.X_train_all), 0
        else:
            mask, mask_shift = self._mask_rows(self.masked_interactions), -self.offset
        for chunk_start in range(block_start, block_end, self.mb_size):
            chunk_end = min(chunk_start + self.mb_size, block_end)
            try:
                scores = model.get_score_block(chunk_start, chunk_end)
            except NotImplementedError:
                scores = model.get_score(np.arange(chunk_start, chunk_end))
            self._merge_block(metrics, np.asarray(scores), mask, chunk_start + mask_shift, cutoffs,
                              chunk_start - self.offset)
        return [self._metrics_as_dict(m) for m in metrics]


class EvaluatorWithColdUser(Evaluator):
    """evaluator.py:444-657: evaluates against users the model has not seen, whose known
    interactions are ``input_interaction``; ``cold_item_features`` adds feature-only items as
    extra columns behind the training items."""

    def __init__(self, input_interaction: Any, ground_truth: Any, cutoff: int = 10,
                 target_metric: str = "ndcg", recommendable_items: Optional[List[int]] = None,
                 per_user_recommendable_items: Union[None, List[List[int]], Any] = None,
                 masked_interactions: Optional[Any] = None, n_threads: Optional[int] = None,
                 recall_with_cutoff: bool = False, mb_size: int = 1024,
                 cold_item_features: Optional[Any] = None, device: Optional[int] = None) -> None:
        if input_interaction.shape[0] != ground_truth.shape[0]:
            raise ValueError("input_interaction and ground_truth must have the same number of rows.")
        n_cold_items = 0 if cold_item_features is None else cold_item_features.shape[0]
        n_warm_items = input_interaction.shape[1]
        if cold_item_features is not None:
            expected_n_items = n_warm_items + n_cold_items
            if ground_truth.shape[1] != expected_n_items:
                raise ValueError("ground_truth must have input_interaction.shape[1] + "
                                 "cold_item_features.shape[0] columns, but got "
                                 f"{ground_truth.shape[1]} instead of {expected_n_items}.")
            if masked_interactions is not None and masked_interactions.shape == input_interaction.shape:
                masked_interactions = self._widen(masked_interactions, n_cold_items)
        super().__init__(ground_truth, offset=0, cutoff=cutoff, target_metric=target_metric,
                         recommendable_items=recommendable_items,
                         per_user_recommendable_items=per_user_recommendable_items,
                         masked_interactions=masked_interactions, n_threads=n_threads,
                         recall_with_cutoff=recall_with_cutoff, mb_size=mb_size, fused=False,
                         device=device)
        self.input_interaction = input_interaction
        self.n_warm_items = n_warm_items
        self.n_cold_items = n_cold_items
        self.cold_item_features = cold_item_features
        if n_cold_items:
            self._input_interaction_mask = self._widen(input_interaction, n_cold_items)
        else:
            self._input_interaction_mask = sps.csr_matrix(input_interaction)

    @staticmethod
    def _widen(X: Any, n_extra: int) -> sps.csr_matrix:
        """``X`` with ``n_extra`` empty columns appended (evaluator.py:543-553, :572-582)."""
        X = sps.csr_matrix(X)
        return sps.csr_matrix((X.data, X.indices, X.indptr), shape=(X.shape[0], X.shape[1] + n_extra))

    def _get_score_matrix_mask(self) -> Optional[sps.csr_matrix]:
        if self.masked_interactions is None:
            return self._input_interaction_mask
        return self.masked_interactions

    def _get_scores_as_list(self, model: Any, cutoffs: List[int]) -> List[Dict[str, float]]:
        # evaluator.py:591-657
        if model.n_items != self.n_warm_items:
            raise ValueError("The model and input_interaction assume different numbers of "
                             "training items.")
        metrics = [Metrics(self.n_items) for _ in cutoffs]
        block_start, block_end = self.offset, self.offset + self.n_users
        score_with_item_features = None
        if self.cold_item_features is not None:
            try:
                score_with_item_features = model._create_cold_user_with_item_features_scorer(
                    self.cold_item_features)
            except NotImplementedError:
                pass
        mask = self._mask_rows(self._get_score_matrix_mask())
        assert mask is not None
        for chunk_start in range(block_start, block_end, self.mb_size):
            chunk_end = min(chunk_start + self.mb_size, block_end)
            input_chunk = self.input_interaction[chunk_start:chunk_end]
            if score_with_item_features is not None:
                try:
                    scores = score_with_item_features(input_chunk)
                except NotImplementedError:
                    score_with_item_features = None
                    scores = model.get_score_cold_user(input_chunk)
            else:
                scores = model.get_score_cold_user(input_chunk)
            scores = np.asarray(scores)
            if scores.shape[1] == self.n_warm_items and self.n_cold_items:
                # a model that cannot score feature-only items leaves them unrankable (:623-634)
                scores = np.concatenate(
                    [scores, np.full((scores.shape[0], self.n_cold_items), -np.inf, dtype=scores.dtype)],
                    axis=1)
            if not scores.flags.c_contiguous:
                warnings.warn("Found col-major(fortran-style) score values.\n"
                              "Transforming it to row-major score matrix.")
                scores = np.ascontiguousarray(scores, dtype=np.float64)
            self._merge_block(metrics, scores, mask, chunk_start, cutoffs, chunk_start)
        return [self._metrics_as_dict(m) for m in metrics]
