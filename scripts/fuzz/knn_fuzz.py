"""Randomised check of the kNN computers against the oracle (indices exact, values 1e-12)."""
import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
from irspack_amd.recommenders import _knn as K
n_fail = 0
N_RUN = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(N_RUN):
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.choice([50, 700, 5000, 17000, 34000])); F = int(rng.integers(20, 1500))
    dens = float(rng.choice([0.002, 0.01, 0.05])) if N > 5000 else float(rng.choice([0.01, 0.1, 0.4]))
    binary = rng.random() < 0.5
    X = sps.random(N, F, density=dens, format="csr", random_state=rng, dtype=np.float64)
    X.data[:] = 1.0 if binary else np.round(0.5 + 3 * rng.random(X.nnz), 3)
    kind = str(rng.choice(["cosine", "cosine_raw", "jaccard", "asymmetric", "tversky", "p3alpha", "rp3beta"]))
    top_k = int(rng.choice([1, 7, 100, 3000, 10 ** 6]))
    same_target = rng.random() < 0.7
    T = X if same_target else sps.csr_matrix(X[rng.choice(N, size=max(1, N // 7), replace=False)])
    try:
        if kind == "cosine":
            g, o = K.CosineSimilarityComputer(X, 1.5, True), O.KNNComputer("cosine", X, 1.5, normalize=True, n_threads=8)
        elif kind == "cosine_raw":
            g, o = K.CosineSimilarityComputer(X, 0.0, False), O.KNNComputer("cosine", X, 0.0, normalize=False, n_threads=8)
        elif kind == "jaccard":
            g, o = K.JaccardSimilarityComputer(X, 0.7), O.KNNComputer("jaccard", X, 0.7, n_threads=8)
        elif kind == "asymmetric":
            g, o = K.AsymmetricSimilarityComputer(X, 1.0, 0.3), O.KNNComputer("asymmetric", X, 1.0, alpha=0.3, n_threads=8)
        elif kind == "tversky":
            g, o = K.TverskyIndexComputer(X, 0.2, 0.6, 1.7), O.KNNComputer("tversky", X, 0.2, alpha=0.6, beta=1.7, n_threads=8)
        elif kind == "p3alpha":
            g, o = K.P3alphaComputer(X, alpha=0.8), O.KNNComputer("p3alpha", X, alpha=0.8, n_threads=8)
        else:
            g, o = K.RP3betaComputer(X, alpha=0.8, beta=0.4), O.KNNComputer("rp3beta", X, alpha=0.8, beta=0.4, n_threads=8)
        if kind in ("p3alpha", "rp3beta"):
            if top_k > 3000 and N > 5000: top_k = 3000
            a, b = sps.csr_matrix(g.compute_W(T, top_k).T), sps.csr_matrix(o.compute_W(T, top_k).T)
        else:
            if top_k > 3000 and N > 5000: top_k = 3000
            a, b = g.compute_similarity(T, top_k), o.compute_similarity(T, top_k)
        a.sort_indices(); b.sort_indices()
        ok = a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) \
            and np.allclose(a.data, b.data, rtol=1e-12, atol=0)
    except Exception as ex:  # noqa: BLE001
        ok = False
        print("exception", repr(ex))
    if not ok: n_fail += 1
    print("seed", seed, kind, "N", N, "F", F, "dens", dens, "binary", binary, "top_k", top_k, "same", same_target, "ok" if ok else "FAIL", flush=True)
print("failures:", n_fail)
