"""Development probe: item-kNN (cosine, top_k=100) and the fused iALS evaluator at the
ML-20M shape: wall time, kernel time and work rates."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irspack_amd.evaluation import Evaluator  # noqa: E402
from irspack_amd.recommenders import IALSRecommender  # noqa: E402
from irspack_amd.recommenders._knn import CosineSimilarityComputer  # noqa: E402
from irspack_amd.synthetic import holdout_split, make_interactions  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml20m")
    ap.add_argument("--top-k", type=int, default=100)
    ap.add_argument("--skip-knn", action="store_true")
    ap.add_argument("--skip-eval", action="store_true")
    ap.add_argument("--K", type=int, default=64)
    args = ap.parse_args()
    X = make_interactions(args.shape)
    U, I = X.shape
    if not args.skip_knn:
        Xt = sps.csr_matrix(X.T, dtype=np.float64)
        Xt.data[:] = 1.0
        t0 = time.perf_counter()
        comp = CosineSimilarityComputer(Xt, 0.0, True)
        t1 = time.perf_counter()
        for rep in range(2):
            t2 = time.perf_counter()
            S = comp.compute_similarity(Xt, args.top_k)
            t3 = time.perf_counter()
            print(json.dumps({"knn": "cosine", "rows": I, "top_k": args.top_k, "rep": rep,
                              "create_s": round(t1 - t0, 3), "wall_s": round(t3 - t2, 3),
                              "kernel_ms": round(comp.last_kernel_ms, 2),
                              "macs": comp.last_macs,
                              "gmacs_per_s": round(comp.last_macs / comp.last_kernel_ms / 1e6, 2),
                              "item_pairs_per_s_dense_equiv": round(I * I / (comp.last_kernel_ms * 1e-3), 1),
                              "out_nnz": int(S.nnz)}), flush=True)
    if not args.skip_eval:
        train, test = holdout_split(X, 0.2, seed=1)
        rec = IALSRecommender(train, n_components=args.K, alpha0=0.1, reg=1e-3, train_epochs=2,
                              solver_type="CHOLESKY")
        t0 = time.perf_counter()
        rec.learn()
        t1 = time.perf_counter()
        ev = Evaluator(test, cutoff=20, target_metric="ndcg")
        for rep in range(2):
            t2 = time.perf_counter()
            m = ev.get_scores(rec, [20])
            t3 = time.perf_counter()
            print(json.dumps({"eval": "ndcg@20 fused iALS", "users": U, "items": I, "rep": rep,
                              "learn_s": round(t1 - t0, 3), "wall_s": round(t3 - t2, 3),
                              "users_per_s": round(U / (t3 - t2), 1),
                              "ndcg@20": m.get("ndcg@20")}), flush=True)


if __name__ == "__main__":
    main()
