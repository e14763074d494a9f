"""Item-kNN recommenders: counterpart of the reference's ``irspack/recommenders/knn.py``
(:35-80 ``BaseKNNRecommender._learn`` and the four concrete classes), ``p3.py`` and
``rp3.py``, on top of the GPU similarity computers in ``_knn``.

``_learn`` keeps the reference's order of operations (knn.py:67-80): optional
feature weighting -> computer built on ``X_weighted.T`` -> ``compute_similarity``
on the *unweighted* ``X.T`` -> ``remove_diagonal`` -> CSC.  Note that the diagonal
competes for a top-k slot before it is zeroed.
"""

import enum
from typing import Any, Optional

from .._threading import get_n_threads
from ._knn import (AsymmetricSimilarityComputer, CosineSimilarityComputer,
                   JaccardSimilarityComputer, P3alphaComputer, RP3betaComputer,
                   TverskyIndexComputer)
from .base import BaseSimilarityRecommender


class FeatureWeightingScheme(str, enum.Enum):
    NONE = "NONE"
    TF_IDF = "TF_IDF"
    BM_25 = "BM_25"


class BaseKNNRecommender(BaseSimilarityRecommender):
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
        """knn.py:68-75 as the computer constructors' ``weighting=`` (applied on the device)."""
        if self.feature_weighting == FeatureWeightingScheme.NONE:
            return None
        if self.feature_weighting == FeatureWeightingScheme.TF_IDF:
            return ("TF_IDF", True)
        if self.feature_weighting == FeatureWeightingScheme.BM_25:
            return ("BM_25", self.bm25_k1, self.bm25_b)
        raise RuntimeError("Unknown weighting scheme.")

    def _learn(self) -> None:
        # knn.py:67-80 - weighting -> computer on X_weighted.T -> compute_similarity(X.T) -> remove_diagonal
        # -> CSC - with the first two steps as one device construction: X.T is a CSC view of the training
        # matrix's own arrays (no host transpose), the weighting happens on the way in
        Xt = self.X_train_all.T
        if self.X_train_all.has_sorted_indices:  # (scipy's transposed view does not inherit the flag: a 20 M-entry rescan)
            Xt.has_sorted_indices = True
        computer = self._create_computer(Xt, self._weighting())
        # (= remove_diagonal(computer.compute_similarity(Xt, top_k)).tocsc(), regrouped on the device)
        self._W = computer.compute_similarity_without_diagonal_csc(Xt, self.top_k)


class CosineKNNRecommender(BaseKNNRecommender):  # knn.py:94-161
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, normalize: bool = False,
                 top_k: int = 100, feature_weighting: str = "NONE", bm25_k1: float = 1.2,
                 bm25_b: float = 0.75, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads, feature_weighting, bm25_k1, bm25_b)
        self.normalize = normalize

    def _create_computer(self, X: Any, weighting=None) -> CosineSimilarityComputer:
        return CosineSimilarityComputer(X, self.shrinkage, self.normalize, self.n_threads, weighting=weighting)


class AsymmetricCosineKNNRecommender(BaseKNNRecommender):  # knn.py:168-235
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, alpha: float = 0.5,
                 top_k: int = 100, feature_weighting: str = "NONE", bm25_k1: float = 1.2,
                 bm25_b: float = 0.75, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads, feature_weighting, bm25_k1, bm25_b)
        self.alpha = alpha

    def _create_computer(self, X: Any, weighting=None) -> AsymmetricSimilarityComputer:
        return AsymmetricSimilarityComputer(X, self.shrinkage, self.alpha, self.n_threads, weighting=weighting)


class JaccardKNNRecommender(BaseKNNRecommender):  # knn.py:238-270
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, top_k: int = 100,
                 n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads)

    def _create_computer(self, X: Any, weighting=None) -> JaccardSimilarityComputer:
        return JaccardSimilarityComputer(X, self.shrinkage, self.n_threads)  # (binarises: nothing to weight)


class TverskyIndexKNNRecommender(BaseKNNRecommender):  # knn.py:273-326
    def __init__(self, X_train_all: Any, shrinkage: float = 0.0, alpha: float = 0.5,
                 beta: float = 0.5, top_k: int = 100, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all, shrinkage, top_k, n_threads)
        self.alpha = alpha
        self.beta = beta

    def _create_computer(self, X: Any, weighting=None) -> TverskyIndexComputer:
        return TverskyIndexComputer(X, self.shrinkage, self.alpha, self.beta, self.n_threads)


class P3alphaRecommender(BaseSimilarityRecommender):  # p3.py:44-76
    def __init__(self, X_train_all: Any, alpha: float = 1, top_k: Optional[int] = None,
                 normalize_weight: bool = False, n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all)
        self.alpha = alpha
        self.top_k = top_k
        self.normalize_weight = normalize_weight
        self.n_threads = get_n_threads(n_threads)

    def _learn(self) -> None:
        import numpy as np
        import scipy.sparse as sps

        computer = P3alphaComputer(self.X_train_all.T, alpha=self.alpha, n_threads=self.n_threads)
        top_k = self.X_train_all.shape[1] if self.top_k is None else self.top_k
        W = computer.compute_W(self.X_train_all.T, top_k)
        if self.normalize_weight:  # p3.py:70-75 (sklearn.preprocessing.normalize(norm="l1", axis=1))
            W = sps.csr_matrix(W)
            s = np.asarray(np.abs(W).sum(axis=1)).ravel()
            s[s == 0] = 1.0
            W = sps.diags(1.0 / s) @ W
        self._W = W


class RP3betaRecommender(BaseSimilarityRecommender):  # rp3.py:49-84
    def __init__(self, X_train_all: Any, alpha: float = 1, beta: float = 0.6,
                 top_k: Optional[int] = None, normalize_weight: bool = False,
                 n_threads: Optional[int] = None) -> None:
        super().__init__(X_train_all)
        self.alpha = alpha
        self.beta = beta
        self.top_k = top_k
        self.normalize_weight = normalize_weight
        self.n_threads = get_n_threads(n_threads)

    def _learn(self) -> None:
        import numpy as np
        import scipy.sparse as sps

        computer = RP3betaComputer(self.X_train_all.T, alpha=self.alpha, beta=self.beta,
                                   n_threads=self.n_threads)
        top_k = self.X_train_all.shape[1] if self.top_k is None else self.top_k
        W = computer.compute_W(self.X_train_all.T, top_k)
        if self.normalize_weight:
            W = sps.csr_matrix(W)
            s = np.asarray(np.abs(W).sum(axis=1)).ravel()
            s[s == 0] = 1.0
            W = sps.diags(1.0 / s) @ W
        self._W = W
