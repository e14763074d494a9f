"""Development probe: a larger synthetic shape (vectorised generator) through iALS at K = 128,
to exercise 32-bit offsets / task lists / memory at scale on one GPU."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,  # noqa: E402
                                                  IALSTrainer, SolverType)


def big_matrix(U, I, nnz, seed=0):
    rng = np.random.default_rng(seed)
    deg = rng.lognormal(3.0, 1.0, size=U)
    deg = np.maximum(1, np.round(deg * nnz / deg.sum())).astype(np.int64)
    rows = np.repeat(np.arange(U, dtype=np.int64), deg)
    w = (np.arange(I, dtype=np.float64) + 1.0) ** -0.9
    cdf = np.cumsum(w / w.sum())
    cols = np.minimum(np.searchsorted(cdf, rng.random(rows.shape[0])), I - 1)
    cols = rng.permutation(I)[cols]
    key = np.unique(rows * I + cols)
    X = sps.csr_matrix((np.ones(key.shape[0], np.float32), (key // I, key % I)), shape=(U, I))
    X.sort_indices()
    return X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=2_000_000)
    ap.add_argument("--items", type=int, default=300_000)
    ap.add_argument("--nnz", type=int, default=60_000_000)
    ap.add_argument("--K", type=int, default=128)
    ap.add_argument("--solver", default="CG")
    args = ap.parse_args()
    t0 = time.time()
    X = big_matrix(args.users, args.items, args.nnz)
    print("matrix", X.shape, X.nnz, "max row", int(np.diff(X.indptr).max()),
          "max col", int(np.bincount(X.indices, minlength=X.shape[1]).max()), f"{time.time() - t0:.1f}s", flush=True)
    mc = IALSModelConfigBuilder().set_K(args.K).set_alpha0(0.1).set_reg(1e-3).build()
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType[args.solver]).set_max_cg_steps(3).build())
    t0 = time.time()
    tr = IALSTrainer(mc, X)
    print("create", f"{time.time() - t0:.1f}s", flush=True)
    times = []
    tr.step(sc)
    tr.profile(True)
    for _ in range(3):
        t1 = time.perf_counter()
        tr.step(sc)
        times.append(time.perf_counter() - t1)
    u = tr.user
    prof = tr.profile_read()
    print(json.dumps({k: round(v["ms"] / v["launches"], 3) for k, v in prof.items()}))
    print(json.dumps({"shape": list(X.shape), "nnz": int(X.nnz), "K": args.K, "solver": args.solver,
                      "epoch_ms": [round(t * 1e3, 1) for t in times],
                      "updates_per_s": round(sum(X.shape) / min(times), 1),
                      "finite": bool(np.isfinite(u).all()), "absmax": float(np.abs(u).max())}))


if __name__ == "__main__":
    main()
