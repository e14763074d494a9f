"""Development probe: kernel time of single target rows (heaviest, median) of the ML-20M shape."""
import os, sys, json
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irspack_amd.recommenders._knn import CosineSimilarityComputer
from irspack_amd.synthetic import make_interactions
X = make_interactions("ml20m")
Xt = sps.csr_matrix(X.T, dtype=np.float64); Xt.data[:] = 1.0
udeg = np.diff(X.indptr)
work = np.array([udeg[Xt.indices[Xt.indptr[i]:Xt.indptr[i+1]]].sum() for i in range(Xt.shape[0])])
order = np.argsort(-work)
comp = CosineSimilarityComputer(Xt, 0.0, True)
comp.compute_similarity(Xt, 100, rows=(0, 8))
for name, i in [("heaviest", order[0]), ("2nd", order[1]), ("10th", order[9]), ("100th", order[99]), ("1000th", order[999]), ("median", order[len(order)//2])]:
    comp.compute_similarity(Xt, 100, rows=(int(i), int(i)+1))
    print(json.dumps({"row": name, "users": int(Xt.indptr[i+1]-Xt.indptr[i]), "macs": int(work[i]), "kernel_ms": round(comp.last_kernel_ms, 3),
                      "gmacs_per_s": round(work[i]/comp.last_kernel_ms/1e6, 2)}))
print("total macs", int(work.sum()), "top10 share", float(work[order[:10]].sum()/work.sum()), "top100 share", float(work[order[:100]].sum()/work.sum()), "top1000", float(work[order[:1000]].sum()/work.sum()))

# kernel time by work quantile: target rows sorted heaviest first, timed in contiguous ranges
Xs = sps.csr_matrix(Xt[order])
edges = [0, 100, 1000, 3000, 8000, 16000, Xs.shape[0]]
for a, b in zip(edges[:-1], edges[1:]):
    comp.compute_similarity(Xs, 100, rows=(a, b))
    w = int(work[order[a:b]].sum())
    print(json.dumps({"rows": [a, b], "macs": w, "share": round(w / work.sum(), 4),
                      "kernel_ms": round(comp.last_kernel_ms, 3),
                      "us_per_row": round(1e3 * comp.last_kernel_ms / (b - a), 2),
                      "gmacs_per_s": round(w / comp.last_kernel_ms / 1e6, 1)}))
