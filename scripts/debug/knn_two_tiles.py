import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
from irspack_amd.recommenders import _knn as K
rng2 = np.random.default_rng(5)
N, U = 20000, 300
rows = rng2.integers(0, U, size=60000)
cols = rng2.integers(0, N, size=60000)
X = sps.csr_matrix((np.ones(60000), (rows, cols)), shape=(U, N))
X.data[:] = 1.0
Xt = sps.csr_matrix(X.T)
comp = K.CosineSimilarityComputer(Xt, 0.0, False)
ref = O.KNNComputer("cosine", Xt, 0.0, normalize=False, n_threads=8)
for top in (2000,):
    for sel in ((458, 459),):
        got = comp.compute_similarity(Xt, top, rows=sel).toarray()
        want = ref.compute_similarity(Xt, top)[sel[0]:sel[1]].toarray()
        bad = np.argwhere(got != want)
        print("top", top, "sel", sel, "mismatches", len(bad), "rows affected", len(np.unique(bad[:, 0])) if len(bad) else 0)
        for (i, j) in bad[:6]:
            print("  row", i, "users", Xt[sel[0] + i].indices, "col", j, "got", got[i, j], "want", want[i, j])
