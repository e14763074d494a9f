"""``Evaluator`` / ``EvaluatorWithColdUser`` counterparts: the caller contract of the
reference's ``irspack/evaluation/evaluator.py:98-183, 196-441, 444-657`` (public names,
arguments, defaults and error messages - its tests ``match=`` on them) on top of the GPU
``EvaluatorCore``.

Structure (this module's own): every way of evaluating - a score matrix, an iterable of score
chunks, a model scored block by block, a cold-user model - is a SOURCE of consecutive
``(first_row, scores)`` blocks, and one ``_BlockAccumulator`` consumes them: it validates a
block, hands it to the device with the rows of the seen-item mask that belong to it, ranks it
once per cutoff and merges the ``Metrics``; at the end it checks that the blocks covered every
user and adds ``catalog_coverage``.  The fused path (an ``irspack_amd`` iALS model on the
evaluator's device) bypasses blocks altogether.

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


class _BlockAccumulator:
    """Running ``Metrics`` per cutoff over consecutive row blocks of scores.

    ``mask`` / ``mask_shift``: the seen-item mask (``MaskRows``) and the offset between a block's
    first row as the SOURCE counts it and the mask's row numbering; ``truth_shift`` likewise for the
    ground truth (both zero unless the evaluator looks at a window of the model's users)."""

    def __init__(self, owner: "Evaluator", cutoffs: List[int], mask: Optional[MaskRows],
                 mask_shift: int = 0, truth_shift: int = 0, what: str = "score chunk") -> None:
        self.owner, self.cutoffs, self.mask = owner, list(cutoffs), mask
        self.mask_shift, self.truth_shift, self.what = mask_shift, truth_shift, what
        self.totals = [Metrics(owner.n_items) for _ in self.cutoffs]
        self.rows_seen = 0

    def check(self, block: Any) -> np.ndarray:
        """the contract of a block handed in from outside (evaluator.py:371-386)"""
        n_items, n_users = self.owner.n_items, self.owner.n_users
        if not isinstance(block, np.ndarray) or block.ndim != 2:
            raise ValueError(f"each {self.what} must be a 2-D ndarray, got {type(block).__name__}.")
        if block.shape[1] != n_items:
            raise ValueError(f"{self.what} must have n_items={n_items} columns, got {block.shape[1]}.")
        if block.dtype not in _SCORE_DTYPES:
            raise ValueError(f"{self.what} must have dtype float32 or float64.")
        if self.rows_seen + block.shape[0] > n_users:
            raise ValueError("score chunks supplied more rows than the evaluator's "
                             f"n_users={n_users}: processed {self.rows_seen + block.shape[0]} rows.")
        return block

    def feed(self, first_row: int, block: np.ndarray) -> None:
        """rank ``block`` (rows ``first_row ..`` of the source) for every cutoff; the caller's array is
        only read - the mask is applied to the device copy"""
        if block.shape[0] == 0:
            return
        if block.dtype not in _SCORE_DTYPES:
            raise ValueError("score must be either float32 or float64.")
        own = self.owner
        ranked = own.core.get_metrics_masked(block, self.mask, first_row + self.mask_shift, self.cutoffs,
                                             first_row + self.truth_shift, own.n_threads,
                                             own.recall_with_cutoff)
        for total, part in zip(self.totals, ranked):
            total.merge(part)
        self.rows_seen += block.shape[0]

    def feed_checked_stream(self, blocks: Iterable[Any]) -> None:
        for block in blocks:
            self.feed(self.rows_seen, self.check(block))
        if self.rows_seen != self.owner.n_users:
            raise ValueError("score chunks did not cover the evaluator's "
                             f"n_users={self.owner.n_users} rows: processed {self.rows_seen} rows.")

    def results(self) -> List[Dict[str, float]]:
        return [self.owner._with_coverage(total) for total in self.totals]


class Evaluator:
    """evaluator.py:37-441."""

    n_users: int
    n_items: int
    masked_interactions: Optional[sps.csr_matrix]

    #: rows per device call of the score-MATRIX entry points: at least this many (the reference
    #: walks the matrix ``mb_size`` = 128 rows at a time, :363-367; 128 waves do not fill 256 compute
    #: units, and the chunking changes nothing but the order of the float64 sums)
    score_matrix_rows_per_call = 4096

    def __init__(self, ground_truth: Any, offset: int = 0, cutoff: int = 10,
                 target_metric: str = "ndcg", recommendable_items: Optional[List[int]] = None,
                 per_user_recommendable_items: Union[None, List[List[int]], Any] = None,
                 masked_interactions: Optional[Any] = None, n_threads: Optional[int] = None,
                 recall_with_cutoff: bool = False, mb_size: int = 128, fused: bool = True,
                 device: Optional[int] = None) -> None:
        ground_truth = sps.csr_matrix(ground_truth).astype(np.float64)  # evaluator.py:114-115
        ground_truth.sort_indices()
        candidates = self._candidate_lists(ground_truth.shape[0], recommendable_items,
                                           per_user_recommendable_items)
        self.core = EvaluatorCore(ground_truth, candidates, device=device)
        distinct = {i for one in candidates for i in one} if len(candidates) > 1 else None
        self.n_recommendable_items = (ground_truth.shape[1] if not candidates else
                                      len(candidates[0]) if distinct is None else len(distinct))
        self.n_users, self.n_items = ground_truth.shape
        self.n_cold_items = 0
        self.offset = offset
        self.cutoff = cutoff
        self.target_metric = TargetMetric[target_metric]
        self.target_metric_name = f"{self.target_metric.name}@{self.cutoff}"
        self.n_threads = get_n_threads(n_threads)
        self.mb_size = mb_size
        self.fused = fused
        self.recall_with_cutoff = recall_with_cutoff
        self.masked_interactions = None
        if masked_interactions is not None:
            if masked_interactions.shape != ground_truth.shape:
                raise ValueError("ground_truth and masked_interactions have different shapes. ")
            self.masked_interactions = sps.csr_matrix(masked_interactions)

    @staticmethod
    def _candidate_lists(n_users: int, shared: Optional[List[int]], per_user: Any) -> List[List[int]]:
        """EvaluatorCore's ``recommendable`` argument: [] (every item), one shared list, or one list
        per user (evaluator.py:117-140)."""
        if shared is not None:
            return [shared]
        if per_user is None:
            return []
        if sps.issparse(per_user):
            rows = sps.csr_matrix(per_user, copy=True)
            rows.eliminate_zeros()  # (the candidates are the NONZERO entries of a row, evaluator.py:124-127)
            per_user = [rows.indices[rows.indptr[u]:rows.indptr[u + 1]].tolist() for u in range(rows.shape[0])]
        if len(per_user) != n_users:
            raise ValueError("ground_truth and per_user_recommendable_items have inconsistent shapes.")
        return per_user

    # -- public entry points: one cutoff or several, flat ``name@cutoff`` keys for several -----
    def get_target_score(self, model: Any) -> float:
        return self.get_score(model)[self.target_metric.name]

    def get_score(self, model: Any) -> Dict[str, float]:
        return self._evaluate_model(model, [self.cutoff])[0]

    def get_scores(self, model: Any, cutoffs: List[int]) -> Dict[str, float]:
        return self._flat(cutoffs, self._evaluate_model(model, cutoffs))

    def get_score_from_score_matrix(self, scores: np.ndarray) -> Dict[str, float]:
        return self._evaluate_matrix(scores, [self.cutoff])[0]

    def get_scores_from_score_matrix(self, scores: np.ndarray, cutoffs: List[int]) -> Dict[str, float]:
        return self._flat(cutoffs, self._evaluate_matrix(scores, cutoffs))

    def get_score_from_score_chunks(self, score_chunks: Iterable[np.ndarray]) -> Dict[str, float]:
        return self._evaluate_chunks(score_chunks, [self.cutoff])[0]

    def get_scores_from_score_chunks(self, score_chunks: Iterable[np.ndarray],
                                     cutoffs: List[int]) -> Dict[str, float]:
        return self._flat(cutoffs, self._evaluate_chunks(score_chunks, cutoffs))

    @staticmethod
    def _flat(cutoffs: List[int], per_cutoff: List[Dict[str, float]]) -> Dict[str, float]:
        return {f"{name}@{c}": one[name] for c, one in zip(cutoffs, per_cutoff) for name in METRIC_NAMES}

    def _with_coverage(self, metrics: Metrics) -> Dict[str, float]:
        out = metrics.as_dict()  # evaluator.py:326-334
        n = self.n_recommendable_items
        out["catalog_coverage"] = out["appeared_item"] / n if n else float("nan")
        return out

    def _get_metrics(self, scores: np.ndarray, cutoff: int, ground_truth_begin: int) -> Metrics:
        """one unmasked block through the reference's own core call (evaluator.py:163-183)"""
        rank = {np.dtype("float64"): self.core.get_metrics_f64,
                np.dtype("float32"): self.core.get_metrics_f32}.get(scores.dtype)
        if rank is None:
            raise ValueError("score must be either float32 or float64.")
        return rank(scores, cutoff, ground_truth_begin, self.n_threads, self.recall_with_cutoff)

    # -- the seen-item mask -------------------------------------------------------------------
    def _score_matrix_mask(self) -> Optional[sps.csr_matrix]:
        """what the score-matrix / score-chunk entry points hide (nothing unless asked to)"""
        return self.masked_interactions

    def _mask_rows(self, mask: Optional[Any]) -> Optional[MaskRows]:
        """``MaskRows`` of a mask matrix, kept while calls pass the same object (the nonzero
        pattern of a 20 M-entry mask takes longer to prepare than a block to rank)."""
        if mask is None:
            return None
        key = (id(mask), mask.shape, mask.nnz, EvaluatorCore._mask_fingerprint(mask))
        held = getattr(self, "_mask_rows_held", None)
        if held is None or held[0] != key:
            held = self._mask_rows_held = (key, mask, MaskRows(mask, self.n_items))
        return held[2]

    # -- sources of blocks ----------------------------------------------------------------------
    def _evaluate_chunks(self, score_chunks: Iterable[np.ndarray], cutoffs: List[int]) -> List[Dict[str, float]]:
        acc = _BlockAccumulator(self, cutoffs, self._mask_rows(self._score_matrix_mask()))
        acc.feed_checked_stream(score_chunks)
        return acc.results()

    def _evaluate_matrix(self, scores: np.ndarray, cutoffs: List[int]) -> List[Dict[str, float]]:
        if not isinstance(scores, np.ndarray) or scores.ndim != 2 or \
                scores.shape != (self.n_users, self.n_items):
            raise ValueError(f"score matrix must have shape ({self.n_users}, {self.n_items}), "
                             f"but got {getattr(scores, 'shape', None)}.")
        if scores.dtype not in _SCORE_DTYPES:
            raise ValueError("score matrix must have dtype float32 or float64.")
        row_bytes = max(self.n_items, 1) * scores.dtype.itemsize
        rows = max(int(self.mb_size), min(self.score_matrix_rows_per_call, max(1, (1 << 30) // row_bytes)))
        return self._evaluate_chunks((scores[b:b + rows] for b in range(0, self.n_users, rows)), cutoffs)

    def _fusable_trainer(self, model: Any):
        """the model's device trainer when scoring, masking and ranking can stay on this evaluator's GPU"""
        trainer = getattr(getattr(model, "trainer", None), "core_trainer", None) if self.fused else None
        if trainer is None or not hasattr(trainer, "_h"):
            return None
        return trainer if getattr(trainer, "_device", None) == self.core._device else None

    def _similarity_weights(self, model: Any) -> Optional[Any]:
        """``(profiles, W by rows)`` when the model scores as ``profiles[u] @ W`` with sparse float64
        operands and nothing overrides its block scores: item-similarity models (``X_train[u] @ W``,
        base.py:406-429) and user-similarity models (``U[u] @ X_train``, base.py:432-453).  The row form of
        a CSC operand is kept per model."""
        from ..recommenders.base import BaseSimilarityRecommender, BaseUserSimilarityRecommender

        if not self.fused:
            return None
        if isinstance(model, BaseSimilarityRecommender):
            if type(model).get_score_block is not BaseSimilarityRecommender.get_score_block:
                return None  # (a subclass that scores differently goes through its own get_score_block)
            profiles, W = model.X_train_all, getattr(model, "_W", None)
        elif isinstance(model, BaseUserSimilarityRecommender):
            if type(model).get_score_block is not BaseUserSimilarityRecommender.get_score_block:
                return None
            profiles, W = getattr(model, "U_", None), model.X_train_all
        else:
            return None
        if W is None or profiles is None or not sps.issparse(W) or not sps.issparse(profiles) or \
                W.dtype != np.float64 or profiles.dtype != np.float64:
            return None
        held = getattr(self, "_sim_rows_held", None)
        if held is None or held[0] is not W or held[1] is not profiles:
            rows = []
            for M in (profiles, W):
                Mr = M if sps.isspmatrix_csr(M) else sps.csr_matrix(M)
                Mr.sort_indices()
                rows.append(Mr)
            held = self._sim_rows_held = (W, profiles, rows[0], rows[1])
        return held[2], held[3]

    def _window_of_training_matrix(self, model: Any, first: int, last: int) -> Any:
        """rows [first, last) of the model's training matrix as the fused path's mask: the whole
        matrix as it is (no copy, its device copy is reused), a slice kept between calls otherwise
        (keyed on the content fingerprint too: an edited training matrix gets a new slice)"""
        X = model.X_train_all
        if first == 0 and last == X.shape[0]:
            return X
        key = (id(X), first, last, self.core._mask_fingerprint(X))
        if getattr(self, "_mask_slice_key", None) != key:
            self._mask_slice_key, self._mask_slice = key, (X, X[first:last])
        return self._mask_slice[1]

    def _evaluate_model(self, model: Any, cutoffs: List[int]) -> List[Dict[str, float]]:
        first, last = self.offset, self.offset + self.n_users
        if last > model.n_users:
            raise ValueError("evaluator offset + n_users exceeds the model's n_users.")
        if self.n_items != model.n_items:
            raise ValueError("The model and evaluator assume different n_items.")
        trainer = self._fusable_trainer(model)
        if trainer is not None:
            mask = (self.masked_interactions if self.masked_interactions is not None
                    else self._window_of_training_matrix(model, first, last))
            return [self._with_coverage(self.core.get_metrics_ials(trainer, first, last, mask, c, 0,
                                                                   self.recall_with_cutoff))
                    for c in cutoffs]
        sim = self._similarity_weights(model)
        if sim is not None:
            profiles, W = sim
            # score = X_train[u] @ W on the device (the host product bit for bit), masked and ranked there:
            # what the block loop below does per `mb_size` users through scipy and PCIe
            if self.masked_interactions is None:
                mask, mask_begin = self._mask_rows(model.X_train_all), first
            else:
                mask, mask_begin = self._mask_rows(self.masked_interactions), 0
            ranked = self.core.get_metrics_similarity(profiles, W, first, last, mask, mask_begin,
                                                      cutoffs, 0, self.recall_with_cutoff)
            return [self._with_coverage(m) for m in ranked]
        # blocks are numbered by MODEL user: the ground truth starts at `offset`; an explicit mask is
        # indexed like the ground truth, the training matrix like the model
        if self.masked_interactions is None:
            acc = _BlockAccumulator(self, cutoffs, self._mask_rows(model.X_train_all), 0, -self.offset)
        else:
            acc = _BlockAccumulator(self, cutoffs, self._mask_rows(self.masked_interactions),
                                    -self.offset, -self.offset)
        for b in range(first, last, self.mb_size):
            e = min(b + self.mb_size, last)
            try:
                block = model.get_score_block(b, e)
            except NotImplementedError:
                block = model.get_score(np.arange(b, e))
            acc.feed(b, np.asarray(block))
        return acc.results()


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
        n_warm = input_interaction.shape[1]
        n_cold = 0 if cold_item_features is None else cold_item_features.shape[0]
        if cold_item_features is not None and ground_truth.shape[1] != n_warm + n_cold:
            raise ValueError("ground_truth must have input_interaction.shape[1] + "
                             "cold_item_features.shape[0] columns, but got "
                             f"{ground_truth.shape[1]} instead of {n_warm + n_cold}.")
        if (n_cold and masked_interactions is not None
                and masked_interactions.shape == input_interaction.shape):
            masked_interactions = self._append_empty_columns(masked_interactions, n_cold)
        super().__init__(ground_truth, offset=0, cutoff=cutoff, target_metric=target_metric,
                         recommendable_items=recommendable_items,
                         per_user_recommendable_items=per_user_recommendable_items,
                         masked_interactions=masked_interactions, n_threads=n_threads,
                         recall_with_cutoff=recall_with_cutoff, mb_size=mb_size, fused=False,
                         device=device)
        self.input_interaction = input_interaction
        self.cold_item_features = cold_item_features
        self.n_warm_items, self.n_cold_items = n_warm, n_cold
        # what a cold user has already seen, in the evaluator's (warm + cold) column space
        self._input_interaction_mask = self._append_empty_columns(input_interaction, n_cold)

    @staticmethod
    def _append_empty_columns(X: Any, n_extra: int) -> sps.csr_matrix:
        X = sps.csr_matrix(X)
        if not n_extra:
            return X
        return sps.csr_matrix((X.data, X.indices, X.indptr), shape=(X.shape[0], X.shape[1] + n_extra))

    def _score_matrix_mask(self) -> Optional[sps.csr_matrix]:
        return self._input_interaction_mask if self.masked_interactions is None else self.masked_interactions

    def _cold_user_scorer(self, model: Any):
        """``history rows -> scores``: through the item features when the model offers that
        (evaluator.py:600-620), with the plain cold-user scores as the way back from a
        ``NotImplementedError`` raised at creation or at the first block"""
        state = {"with_features": None}
        if self.cold_item_features is not None:
            try:
                state["with_features"] = model._create_cold_user_with_item_features_scorer(
                    self.cold_item_features)
            except NotImplementedError:
                pass

        def score(history: Any) -> np.ndarray:
            if state["with_features"] is not None:
                try:
                    return np.asarray(state["with_features"](history))
                except NotImplementedError:
                    state["with_features"] = None
            return np.asarray(model.get_score_cold_user(history))

        return score

    def _in_evaluator_columns(self, block: np.ndarray) -> np.ndarray:
        """a block over the warm items only gets -inf columns for the feature-only items (a model that
        cannot score them leaves them unrankable, evaluator.py:623-634); row-major for the device"""
        if self.n_cold_items and block.shape[1] == self.n_warm_items:
            pad = np.full((block.shape[0], self.n_cold_items), -np.inf, dtype=block.dtype)
            block = np.concatenate([block, pad], axis=1)
        if not block.flags.c_contiguous:
            warnings.warn("Found col-major(fortran-style) score values.\n"
                          "Transforming it to row-major score matrix.")
            block = np.ascontiguousarray(block, dtype=np.float64)
        return block

    def _evaluate_model(self, model: Any, cutoffs: List[int]) -> List[Dict[str, float]]:
        if model.n_items != self.n_warm_items:
            raise ValueError("The model and input_interaction assume different numbers of "
                             "training items.")
        acc = _BlockAccumulator(self, cutoffs, self._mask_rows(self._score_matrix_mask()))
        score = self._cold_user_scorer(model)
        for b in range(0, self.n_users, self.mb_size):
            history = self.input_interaction[b:min(b + self.mb_size, self.n_users)]
            acc.feed(b, self._in_evaluator_columns(score(history)))
        return acc.results()
