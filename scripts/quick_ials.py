"""Development probe: time iALS epochs on a synthetic shape and print per-kernel times."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from irspack_amd.recommenders._ials_core import (  # noqa: E402
    IALSModelConfigBuilder,
    IALSSolverConfigBuilder,
    IALSTrainer,
    SolverType,
)
from irspack_amd.synthetic import describe, make_interactions  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml20m")
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--solvers", default="CHOLESKY,CG")
    args = ap.parse_args()
    t0 = time.time()
    X = make_interactions(args.shape)
    print("matrix", describe(X), f"{time.time() - t0:.1f}s", flush=True)
    mc = (IALSModelConfigBuilder().set_K(args.K).set_alpha0(0.1).set_reg(1e-3)
          .set_init_stdev(0.1).build())
    for kind in args.solvers.split(","):
        sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind])
              .set_max_cg_steps(3).build())
        t0 = time.time()
        tr = IALSTrainer(mc, X)
        print(kind, "create", f"{time.time() - t0:.2f}s", flush=True)
        tr.step(sc)  # warm-up
        tr.profile(True)
        times = []
        for _ in range(args.epochs):
            t1 = time.perf_counter()
            tr.step(sc)
            times.append(time.perf_counter() - t1)
        prof = tr.profile_read()
        tr.profile(False)
        med = float(np.median(times))
        n = X.shape[0] + X.shape[1]
        print(json.dumps({"solver": kind, "epoch_ms": [round(t * 1e3, 3) for t in times],
                          "median_ms": round(med * 1e3, 3),
                          "updates_per_s": round(n / med, 1),
                          "kernels": {k: {"ms_per_launch": round(v["ms"] / v["launches"], 4),
                                          "launches": v["launches"]} for k, v in prof.items()}}),
              flush=True)
        u = tr.user
        print(kind, "finite", bool(np.isfinite(u).all()), "absmax", float(np.abs(u).max()), flush=True)


if __name__ == "__main__":
    main()
