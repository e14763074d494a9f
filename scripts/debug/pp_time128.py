import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config
from irspack_amd.recommenders._ials_core import IALSSolverConfigBuilder, IALSTrainer, SolverType
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tr = IALSTrainer(model_config(K), X)
sc = IALSSolverConfigBuilder().set_solver_type(SolverType.IALSPP).set_ialspp_subspace_dimension(64).set_ialspp_iteration(1).build()
tr.step(sc); tr.synchronize()
tr.profile(True)
t0 = time.perf_counter()
for _ in range(3): tr.step(sc)
tr.synchronize()
dt = (time.perf_counter() - t0) / 3 * 1e3
prof = tr.profile_read()
print(os.environ.get("IRSPACK_AMD_LIB", "default")[-22:], "K", K, "ms/epoch %.2f" % dt, {k: round(v["ms"] / v["launches"], 2) for k, v in prof.items() if "pp" in k}, flush=True)
