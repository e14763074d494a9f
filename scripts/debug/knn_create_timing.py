import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["IRSPACK_AMD_KNN_TIMING"] = "1"
import numpy as np, scipy.sparse as sps
from irspack_amd.recommenders._knn import CosineSimilarityComputer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64); Xt.data[:] = 1.0
for i in range(3):
    t0 = time.perf_counter()
    c = CosineSimilarityComputer(Xt, 0.0, True)
    print(f"knn create #{i}: {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr, flush=True)
    del c
Xw = Xt.copy(); Xw.data = np.random.default_rng(0).uniform(0.5, 2.0, Xw.nnz)
for i in range(3):
    t0 = time.perf_counter()
    c = CosineSimilarityComputer(Xw, 0.0, True)
    print(f"knn create (weighted) #{i}: {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr, flush=True)
    del c
os.environ["IRSPACK_AMD_KNN_DEVICE_CREATE"] = "0"
t0 = time.perf_counter()
c = CosineSimilarityComputer(Xw, 0.0, True)
print(f"knn create (weighted, host path): {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr, flush=True)
