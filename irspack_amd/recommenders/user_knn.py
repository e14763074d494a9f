"""User-kNN recommenders: counterpart of the reference's ``irspack/recommenders/user_knn.py``
(:31-76 ``BaseUserKNNRecommender._learn``, :81-148 cosine, :155-218 asymmetric cosine) on the
GPU similarity computers of ``_knn``.

``_learn`` keeps the reference's order (user_knn.py:62-76): optional feature weighting ->
computer built on ``X_weighted`` (users are the rows: no transpose, unlike item-kNN) ->
``compute_similarity`` on the *unweighted* ``X`` -> ``remove_diagonal``.  The result ``U`` is
[n_users, n_users]; scores are ``U[u] @ X`` (base.py:449-453).
"""

from typing import Any, Optional

from .._threading import get_n_threads
from ..utils import remove_diagonal
from ._knn import AsymmetricSimilarityComputer, CosineSimilarityComputer
from .base import BaseUserSimilarityRecommender
from .knn import FeatureWeightingScheme


class BaseUserKNNRecommender(BaseUserSimilarityRecommender):
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, top_k: int = 100,
                 n_threads: Optional[int] = None, feature_weighting: str = "NONE",
                 bm25_k1: float = 1.2, bm25_b: float = 0.75) -> None:
        super().__init__(X_train_all)
        self.shrinkage = shrinkage
        self.top_k = top_k
        self.feature_weighting = FeatureWeightingScheme(feature_weighting)
        self.bm25_k1 = bm25_k1
        self.bm25_b = bm25_b
        self.n_threads = get_n_threads(n_threads)

    def _create_computer(self, X: Any, weighting=None):
        raise NotImplementedError("")

    def _weighting(self):
        """user_knn.py:62-72 as the computer constructors' ``weighting=`` (applied on the device)."""
        scheme = self.feature_weighting
        if scheme == FeatureWeightingScheme.NONE:
            return None
        if scheme == FeatureWeightingScheme.TF_IDF:
            return ("TF_IDF", True)
        if scheme == FeatureWeightingScheme.BM_25:
            return ("BM_25", self.bm25_k1, self.bm25_b)
        raise RuntimeError("Unknown weighting scheme.")

    def _learn(self) -> None:
        computer = self._create_computer(self.X_train_all, self._weighting())
        self.U_ = remove_diagonal(computer.compute_similarity(self.X_train_all, self.top_k))


class CosineUserKNNRecommender(BaseUserKNNRecommender):  # user_knn.py:81-148 (normalize defaults to True)
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, normalize: bool = True,
                 top_k: int = 100, feature_weighting: str = "NONE", bm25_k1: float = 1.2,
                 bm25_b: float = 0.75, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads, feature_weighting, bm25_k1, bm25_b)
        self.normalize = normalize

    def _create_computer(self, X: Any, weighting=None) -> CosineSimilarityComputer:
        return CosineSimilarityComputer(X, self.shrinkage, self.normalize, self.n_threads, weighting=weighting)


class AsymmetricCosineUserKNNRecommender(BaseUserKNNRecommender):  # user_knn.py:155-218
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, alpha: float = 0.5,
                 top_k: int = 100, feature_weighting: str = "NONE", bm25_k1: float = 1.2,
                 bm25_b: float = 0.75, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads, feature_weighting, bm25_k1, bm25_b)
        self.alpha = alpha

    def _create_computer(self, X: Any, weighting=None) -> AsymmetricSimilarityComputer:
        return AsymmetricSimilarityComputer(X, self.shrinkage, self.alpha, self.n_threads, weighting=weighting)
