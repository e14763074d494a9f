"""Mirror of ``irspack.recommenders`` for the hot path (irspack/recommenders/__init__.py):
the recommenders whose compiled core is rebuilt here."""

from .base import BaseRecommender, BaseSimilarityRecommender, BaseUserSimilarityRecommender
from .ials import IALSRecommender
from .knn import (AsymmetricCosineKNNRecommender, CosineKNNRecommender, JaccardKNNRecommender,
                  P3alphaRecommender, RP3betaRecommender, TverskyIndexKNNRecommender)
from .user_knn import AsymmetricCosineUserKNNRecommender, CosineUserKNNRecommender

__all__ = ["BaseRecommender", "BaseSimilarityRecommender", "IALSRecommender",
           "CosineKNNRecommender", "AsymmetricCosineKNNRecommender", "JaccardKNNRecommender",
           "TverskyIndexKNNRecommender", "P3alphaRecommender", "RP3betaRecommender",
           "BaseUserSimilarityRecommender", "CosineUserKNNRecommender",
           "AsymmetricCosineUserKNNRecommender"]
