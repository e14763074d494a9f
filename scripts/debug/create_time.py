import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import model_config
from irspack_amd.recommenders._ials_core import IALSTrainer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
for i in range(3):
    t0 = time.perf_counter()
    tr = IALSTrainer(model_config(64), X)
    print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    del tr
