"""Development probe: feature-aware iALS epoch time at the ML-20M shape."""
import json, os, sys, time
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,
                                                  IALSTrainer, SolverType)
from irspack_amd.synthetic import make_interactions

X = make_interactions("ml20m")
rng = np.random.default_rng(0)
UF = rng.standard_normal((X.shape[0], 32)).astype(np.float32)            # dense user features
IF = sps.random(X.shape[1], 200, density=0.05, random_state=1, format="csr", dtype=np.float32)
mc = (IALSModelConfigBuilder().set_K(64).set_alpha0(0.1).set_reg(1e-3)
      .set_lambda_user_feature(1.0).set_lambda_item_feature(1.0).build())
sc = IALSSolverConfigBuilder().set_solver_type(SolverType.CHOLESKY).build()
t0 = time.time()
t = IALSTrainer(mc, X, UF, IF)
print("create", round(time.time() - t0, 2), "s")
t.step(sc)
t.profile(True)
times = []
for _ in range(5):
    t1 = time.perf_counter(); t.step(sc); times.append(time.perf_counter() - t1)
prof = t.profile_read()
print(json.dumps({"epoch_ms": [round(x * 1e3, 2) for x in times],
                  "kernels": {k: round(v["ms"] / v["launches"], 3) for k, v in prof.items()}}))
