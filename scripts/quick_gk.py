"""Development probe: epoch times of the general-size path (K > 256, wide iALS++ blocks) on the
ML-20M shape.  python scripts/quick_gk.py [K ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,  # noqa: E402
                                                  IALSTrainer, SolverType)
from irspack_amd.synthetic import make_interactions  # noqa: E402

X = make_interactions("ml20m")
Ks = [int(a) for a in sys.argv[1:]] or [320, 512]
for K in Ks:
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-3).set_init_stdev(0.1).build()
    tr = IALSTrainer(mc, X)
    for kind, sub in (("CG", 64), ("CHOLESKY", 64), ("IALSPP", 64), ("IALSPP", 128)):
        if kind == "IALSPP" and sub == 128 and K > 320:
            continue
        sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind])
              .set_max_cg_steps(3).set_ialspp_subspace_dimension(sub).build())
        tr.step(sc)
        tr.profile(True)
        times = []
        for _ in range(2):
            t1 = time.perf_counter()
            tr.step(sc)
            times.append(time.perf_counter() - t1)
        prof = tr.profile_read()
        tr.profile(False)
        print(json.dumps({"K": K, "solver": kind, "sub": sub, "epoch_ms": [round(t * 1e3, 1) for t in times],
                          "kernels": {k: round(v["ms"] / v["launches"], 2) for k, v in prof.items()},
                          "finite": bool(np.isfinite(tr.user).all())}), flush=True)
