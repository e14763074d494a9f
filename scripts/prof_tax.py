import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer, SolverType
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
mc = IALSModelConfigBuilder().set_K(64).set_alpha0(0.1).set_reg(1e-3).set_init_stdev(0.1).build()
sc = IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType.CHOLESKY).build()
tr = IALSTrainer(mc, X)
for _ in range(3): tr.step(sc)
tr.synchronize()
for rep in range(3):
    for mode in (False, True, 2):
        tr.profile(mode)
        tr.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): tr.step(sc)
        tr.synchronize()
        dt = (time.perf_counter() - t0) / 20
        if mode: tr.profile_read()
        tr.profile(False)
        print("profile", mode, "epoch ms", round(dt * 1e3, 4), flush=True)
