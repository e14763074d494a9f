"""Mirror of ``irspack.evaluation`` for the hot path: the compiled ``_core_evaluator``
surface and the ``Evaluator`` host class that drives it (irspack/evaluation/__init__.py)."""

from ._core_evaluator import EvaluatorCore, Metrics, evaluate_list_vs_list
from .evaluator import METRIC_NAMES, Evaluator, EvaluatorWithColdUser, TargetMetric

__all__ = ["Evaluator", "EvaluatorWithColdUser", "EvaluatorCore", "Metrics", "METRIC_NAMES", "TargetMetric",
           "evaluate_list_vs_list"]
