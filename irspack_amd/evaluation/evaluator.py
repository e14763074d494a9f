"""``Evaluator`` counterpart: the caller contract of the reference's
``irspack/evaluation/evaluator.py:98-183, 196-205, 326-441`` on top of the GPU
``EvaluatorCore``: block loop over users (``mb_size``), ``get_score_block`` with a
``get_score`` fallback, seen-item masking with ``-inf``, one ranking pass per cutoff,
``Metrics.merge`` and the ``catalog_coverage`` post-processing.

When the model is an ``irspack_amd`` ``IALSRecommender`` living on the evaluator's
device, ``fused=True`` (the default) scores, masks and ranks on the GPU without
materialising the dense block on the host (``irs_eval_get_metrics_ials``); the
results are the same (tests/test_gpu_evaluator.py).
"""

import enum
from typing import Any, Dict, List, Optional, Union

import numpy as np
import scipy.sparse as sps

from .._threading import get_n_threads
from ._core_evaluator import EvaluatorCore, Metrics

METRIC_NAMES = ["hit", "recall", "ndcg", "map", "precision", "gini_index", "entropy",
                "appeared_item", "catalog_coverage"]


class TargetMetric(enum.Enum):
    ndcg = "ndcg"
    recall = "recall"
    hit = "hit"
    map = "map"
    precision = "precision"


class Evaluator:
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

    def get_target_score(self, model: Any) -> float:
        return self.get_score(model)[self.target_metric.name]

    def get_score(self, model: Any) -> Dict[str, float]:
        return self._get_scores_as_list(model, [self.cutoff])[0]

    def get_scores(self, model: Any, cutoffs: List[int]) -> Dict[str, float]:
        result: Dict[str, float] = {}
        for cutoff, score in zip(cutoffs, self._get_scores_as_list(model, cutoffs)):
            for name in METRIC_NAMES:
                result[f"{name}@{cutoff}"] = score[name]
        return result

    def _metrics_as_dict(self, metrics: Metrics) -> Dict[str, float]:
        # evaluator.py:326-334
        result = metrics.as_dict()
        if self.n_recommendable_items:
            result["catalog_coverage"] = result["appeared_item"] / self.n_recommendable_items
        else:
            result["catalog_coverage"] = float("nan")
        return result

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
        # evaluator.py:400-441
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
        for chunk_start in range(block_start, block_end, self.mb_size):
            chunk_end = min(chunk_start + self.mb_size, block_end)
            try:
                scores = model.get_score_block(chunk_start, chunk_end)
            except NotImplementedError:
                scores = model.get_score(np.arange(chunk_start, chunk_end))
            if self.masked_interactions is None:
                mask = model.X_train_all[chunk_start:chunk_end]
            else:
                mask = self.masked_interactions[chunk_start - self.offset:chunk_end - self.offset]
            scores[mask.nonzero()] = -np.inf
            for i, c in enumerate(cutoffs):
                metrics[i].merge(self._get_metrics(scores, c, chunk_start - self.offset))
        return [self._metrics_as_dict(m) for m in metrics]
