"""Development probe: the fused iALS evaluator at the ML-20M shape (K from argv)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import holdout, model_config, solver_config  # noqa: E402
from irspack_amd.evaluation._core_evaluator import EvaluatorCore  # noqa: E402
from irspack_amd.recommenders._ials_core import IALSTrainer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = make_interactions("ml20m")
tr = IALSTrainer(model_config(K), X)
tr.step(solver_config("CG"))
gt, mask = holdout(X)
ev = EvaluatorCore(gt, [])
for rep in range(4):
    t0 = time.perf_counter()
    m = ev.get_metrics_ials(tr, 0, X.shape[0], mask, 20, 0, False)
    print("K", K, "wall", round(time.perf_counter() - t0, 4), "ndcg", m.as_dict()["ndcg"], flush=True)
