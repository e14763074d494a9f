import os, sys, time
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.recommenders._knn import CosineSimilarityComputer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64); Xt.data[:] = 1.0
for i in range(3):
    t0 = time.perf_counter()
    comp = CosineSimilarityComputer(Xt, 0.0, True)
    print("create %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    del comp
