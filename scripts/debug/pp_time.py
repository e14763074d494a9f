import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config
from irspack_amd.recommenders._ials_core import IALSSolverConfigBuilder, IALSTrainer, SolverType
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
for K in (64, 32):
    tr = IALSTrainer(model_config(K), X)
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType.IALSPP).set_ialspp_subspace_dimension(64).set_ialspp_iteration(1).build()
    for _ in range(2): tr.step(sc)
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): tr.step(sc)
    tr.synchronize()
    print("K", K, "IALSPP D=64 ms/epoch", (time.perf_counter() - t0) / 10 * 1e3, "direct =", os.environ.get("IRSPACK_AMD_IALSPP_DIRECT", "1"), flush=True)
