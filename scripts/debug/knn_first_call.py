"""Development probe: construction + FIRST compute call of fresh cosine computers (what learn() pays)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.recommenders._knn import CosineSimilarityComputer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402

X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64)
Xt.data[:] = 1.0
for rep in range(4):
    if rep == 3:
        os.environ["IRSPACK_AMD_KNN_TIMING"] = "1"
    t0 = time.perf_counter()
    c = CosineSimilarityComputer(Xt, 0.0, True)
    t1 = time.perf_counter()
    S = c.compute_similarity(Xt, 100)
    t2 = time.perf_counter()
    S2 = c.compute_similarity(Xt, 100)
    t3 = time.perf_counter()
    print(f"fresh computer #{rep}: create {(t1 - t0) * 1e3:.1f} ms, first call {(t2 - t1) * 1e3:.1f} ms, second call {(t3 - t2) * 1e3:.1f} ms",
          file=sys.stderr, flush=True)
    del c, S, S2
