"""Mirror of ``irspack.recommenders`` for the hot path (irspack/recommenders/__init__.py):
the recommenders whose compiled core is rebuilt here."""

from .base import BaseRecommender, BaseSimilarityRecommender
from .ials import IALSRecommender
from .knn import (AsymmetricCosineKNNRecommender, CosineKNNRecommender, JaccardKNNRecommender,
                  P3alphaRecommender, RP3betaRecommender, TverskyIndexKNNRecommender)

__all__ = ["BaseRecommender", "BaseSimilarityRecommender", "IALSRecommender",
           "CosineKNNRecommender", "AsymmetricCosineKNNRecommender", "JaccardKNNRecommender",
           "TverskyIndexKNNRecommender", "P3alphaRecommender", "RP3betaRecommender"]
