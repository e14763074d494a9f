"""Minimal recommender base: the caller contract of the reference's
``irspack/recommenders/base.py:79-126, 290-337, 406-429`` that the hot path relies on
(matrix normalisation, ``learn``, seen-item masking, similarity scoring).  Tuning,
the registry metaclass and config classes are orchestration and out of scope.
"""

from typing import Any, Callable, Optional, Union

import numpy as np
import scipy.sparse as sps


def _sparse_to_array(U: Any) -> np.ndarray:
    if sps.issparse(U):
        return np.asarray(U.toarray())
    return np.asarray(U)


class BaseRecommender:
    def __init__(self, X_train_all: Any, **kwargs: Any) -> None:
        # base.py:94-101
        self.X_train_all: sps.csr_matrix = sps.csr_matrix(X_train_all).astype(np.float64)
        self.n_users: int = self.X_train_all.shape[0]
        self.n_items: int = self.X_train_all.shape[1]
        self.X_train_all.sort_indices()

    def learn(self):
        self._learn()
        return self

    def _learn(self) -> None:
        raise NotImplementedError("_learn must be implemented.")

    def get_score(self, user_indices: np.ndarray) -> np.ndarray:
        raise NotImplementedError("get_score must be implemented")

    def get_score_block(self, begin: int, end: int) -> np.ndarray:
        raise NotImplementedError("get_score_block not implemented!")

    def get_score_remove_seen(self, user_indices: np.ndarray) -> np.ndarray:
        # base.py:308-322
        scores = _sparse_to_array(self.get_score(user_indices))
        m = self.X_train_all[user_indices].tocsr()
        scores[m.nonzero()] = -np.inf
        return scores

    def get_score_remove_seen_block(self, begin: int, end: int) -> np.ndarray:
        # base.py:324-337
        scores = _sparse_to_array(self.get_score_block(begin, end))
        m = self.X_train_all[begin:end]
        scores[m.nonzero()] = -np.inf
        return scores

    def get_score_cold_user(self, X: Any) -> np.ndarray:
        raise NotImplementedError(
            f"get_score_cold_user is not implemented for {self.__class__.__name__}!")

    def _create_cold_user_with_item_features_scorer(self, item_features: Any) -> Callable[[Any], np.ndarray]:
        """base.py:353-362: lets an evaluator prepare feature-derived item state once and score
        many user blocks against it."""
        return lambda X: self.get_score_cold_user_with_item_features(X, item_features)

    def get_score_cold_user_with_item_features(self, X: Any, item_features: Any) -> np.ndarray:
        # base.py:364-389
        raise NotImplementedError("Scoring additional items from features is not implemented for "
                                  f"{self.__class__.__name__}.")

    def get_score_cold_user_remove_seen(self, X: Any) -> np.ndarray:
        score = self.get_score_cold_user(X)
        score[sps.csr_matrix(X).nonzero()] = -np.inf
        return score


class BaseSimilarityRecommender(BaseRecommender):
    """base.py:406-429: score = X[u] @ W with the learnt item-item weights."""

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._W: Optional[Union[sps.csr_matrix, sps.csc_matrix, np.ndarray]] = None

    @property
    def W(self):
        if self._W is None:
            raise RuntimeError("W fetched before fit.")
        return self._W

    def get_score(self, user_indices: np.ndarray) -> np.ndarray:
        return _sparse_to_array(self.X_train_all[user_indices].dot(self.W))

    def get_score_cold_user(self, X: Any) -> np.ndarray:
        return _sparse_to_array(sps.csr_matrix(X).dot(self.W))

    def get_score_block(self, begin: int, end: int) -> np.ndarray:
        return _sparse_to_array(self.X_train_all[begin:end].dot(self.W))


class BaseUserSimilarityRecommender(BaseRecommender):
    """base.py:432-453: score = U[u] @ X with the learnt user-user weights."""

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._X_csc: sps.csc_matrix = self.X_train_all.tocsc()
        self.U_: Optional[Union[sps.csr_matrix, sps.csc_matrix, np.ndarray]] = None

    @property
    def U(self):
        if self.U_ is None:
            raise RuntimeError("W fetched before fit.")  # (the reference's wording, base.py:446)
        return self.U_

    def get_score(self, user_indices: np.ndarray) -> np.ndarray:
        return _sparse_to_array(self.U[user_indices].dot(self._X_csc).toarray())

    def get_score_block(self, begin: int, end: int) -> np.ndarray:
        return _sparse_to_array(self.U[begin:end].dot(self._X_csc))
