import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
from irspack_amd.recommenders import _knn as K

def run(U, N, dens, seed, top):
    rng = np.random.default_rng(seed)
    X = sps.random(U, N, density=dens, random_state=rng, format="csr", dtype=np.float64)
    X.data[:] = 1.0
    Xt = sps.csr_matrix(X.T)
    comp = K.CosineSimilarityComputer(Xt, 0.0, False)
    ref = O.KNNComputer("cosine", Xt, 0.0, normalize=False, n_threads=4)
    got = comp.compute_similarity(Xt, top).toarray()
    want = ref.compute_similarity(Xt, top).toarray()
    bad = np.argwhere(got != want)
    print(U, N, dens, "user row lens", np.diff(X.indptr)[:6], "mismatches", len(bad))
    if len(bad):
        for (i, j) in bad[:10]:
            print("  row", i, "col", j, "got", got[i, j], "want", want[i, j])
        cols = np.unique(bad[:, 1])
        print("  bad col range", cols.min(), cols.max(), "n", len(cols), "mod128 hist", np.bincount((cols % 128) // 16, minlength=8))

run(2, 1000, 0.1, 0, 1000)
run(2, 1000, 0.3, 1, 1000)
run(3, 1000, 0.6, 2, 1000)
run(1, 20000, 0.002, 3, 2000)
run(1, 20000, 0.01, 3, 2000)
run(2, 20000, 0.02, 4, 2000)
run(5, 40000, 0.01, 5, 2000)
