import os, sys, time
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.recommenders._knn import CosineSimilarityComputer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64); Xt.data[:] = 1.0
comp = CosineSimilarityComputer(Xt, 0.0, True)
comp.compute_similarity(Xt, 100, rows=(0, 64))
import cProfile, pstats
for rep in range(2):
    t0 = time.perf_counter(); S = comp.compute_similarity(Xt, 100); print("wall", round(time.perf_counter() - t0, 4), "kernel_ms", comp.last_kernel_ms, flush=True)
pr = cProfile.Profile(); pr.enable(); S = comp.compute_similarity(Xt, 100); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
