"""Development probe: RP3beta through the boundary at the ML-20M shape: construction + compute_W(top_k = 100)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["IRSPACK_AMD_KNN_TIMING"] = "1"
from irspack_amd.recommenders._knn import RP3betaComputer  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402

X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64)
Xt.data[:] = 1.0
for rep in range(3):
    t0 = time.perf_counter()
    c = RP3betaComputer(Xt, 0.6, 0.4)
    t1 = time.perf_counter()
    W = c.compute_W(Xt, 100)
    t2 = time.perf_counter()
    print(f"rp3beta #{rep}: create {(t1 - t0) * 1e3:.1f} ms, compute_W {(t2 - t1) * 1e3:.1f} ms (kernel {c.last_kernel_ms:.1f}), nnz {W.nnz}",
          file=sys.stderr, flush=True)
