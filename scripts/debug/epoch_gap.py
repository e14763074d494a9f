import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config, solver_config
from irspack_amd.recommenders._ials_core import IALSTrainer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
tr = IALSTrainer(model_config(64), X)
for kind in ("CHOLESKY", "CG"):
    sc = solver_config(kind)
    for prof in (False, True, False):
        for _ in range(3): tr.step(sc)
        tr.synchronize()
        tr.profile(prof)
        t0 = time.perf_counter()
        for _ in range(20): tr.step(sc)
        tr.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        if prof: tr.profile_read()
        tr.profile(False)
        print(kind, "profile", prof, "ms/epoch %.3f" % dt, flush=True)
