"""Sweeps the reference's operating points (IALSRecommender defaults and the corners of
default_tune_range, /root/reference/src/irspack/recommenders/ials.py:357-379) over one matrix and
prints the achieved GPU / oracle / float64 distributions: the exploratory form of
tests/test_gpu_operating_point.py (same machinery, tests/_operating_point.py).

    python scripts/operating_point_probe.py [shape] [K,K,...] [kinds] [epochs_before] [alpha0,...] [reg,...]
        > gpurun_out/op_probe.jsonl
(without the last two arguments: the defaults point and the four corners of the tune range)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import _operating_point as OP  # noqa: E402
from irspack_amd.synthetic import make_interactions  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "ml100k"
    Ks = [int(k) for k in (sys.argv[2] if len(sys.argv) > 2 else "4,20,64,300").split(",")]
    kinds = (sys.argv[3] if len(sys.argv) > 3 else "CG,CHOLESKY,IALSPP").split(",")
    X = make_interactions(shape)
    Xt = OP.transpose_csr(X)
    epochs_before = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    points = [(0.0, 1e-3)] + [(a, r) for a in (3e-3, 1.0) for r in (1e-4, 1e-1)]
    if len(sys.argv) > 6:
        points = [(float(a), float(r)) for a in sys.argv[5].split(",") for r in sys.argv[6].split(",")]
    for K in Ks:
        for kind in kinds:
            for alpha0, reg in points:
                t0 = time.time()
                res = OP.run_point(X, Xt, K, kind, alpha0, reg, epochs_before=epochs_before,
                                   kappa_rows=64 if (kind == "CHOLESKY" and X.shape[0] < 5000) else 0)
                rec = dict(shape=shape, K=K, kind=kind, alpha0=alpha0, reg=reg, epochs_before=epochs_before,
                           train_exc=res["train_exc"], wall_s=round(time.time() - t0, 2))
                for k in ("train_fac_user", "train_fac_item"):
                    if k in res:
                        rec[k] = res[k]
                for side, m in enumerate(res["sides"]):
                    s = OP.summary(m)
                    s["exc"] = (m["gpu_exc"], m["orc_exc"])
                    if "fac_gpu" in m:
                        w = int(np.argmax(m["fac_gpu"]))
                        s["worst_row"], s["worst_row_nnz"] = w, int(m["nnz"][w])
                        s["finite"] = m["finite"]
                    rec["user" if side == 0 else "item"] = s
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
