"""Development probe: the tuned CPU build's Cholesky half-steps at 1 .. all host threads (what limits
bench.py's cpu_baseline on the GPU box: the kernel, the memory system or the thread count)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402

X = make_interactions("ml20m")
K = 64
U, I = X.shape
rng = np.random.default_rng(0)
user = (rng.standard_normal((U, K)) * 0.1).astype(np.float32)
item = (rng.standard_normal((I, K)) * 0.1).astype(np.float32)
mc = O.model_config(K, alpha0=0.1, reg=1e-3, nu=1.0, init_stdev=0.1, random_seed=42)
O.use_fast_build()
Pu = O.ials_gramian(item, 0.1, 8)
cores = os.cpu_count()
print(json.dumps({"cpu_count": cores}))
for thr in (1, 8, 32, 64, 128, cores):
    sc = O.solver_config(thr, "CHOLESKY", 3)
    n = min(U, max(2000, 3000 * thr))
    O.ials_solver_step(user, X, item, Pu, mc, sc, 0, min(n, 4000))
    t0 = time.perf_counter()
    O.ials_solver_step(user, X, item, Pu, mc, sc, 0, n)
    dt = time.perf_counter() - t0
    nnz = float(X.indptr[n])
    fl = nnz * (K * (K + 1) + 2 * K) + n * (K ** 3 / 3 + 2 * K * K)
    print(json.dumps({"threads": thr, "rows": n, "s": round(dt, 4), "gflops": round(fl / dt / 1e9, 1),
                      "gflops_per_thread": round(fl / dt / 1e9 / thr, 2)}), flush=True)
