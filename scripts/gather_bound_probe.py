"""Development probe: is the item half of the K = 64 epoch bound by the gather of the user table?
The same matrix with its user ids folded into a small range (u mod M): the same entry counts per item,
the same matrix instructions, but the gathered rows come out of an M-row table that stays in L2."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder, IALSSolverConfigBuilder,  # noqa: E402
                                                  IALSTrainer, SolverType)
from irspack_amd.synthetic import make_interactions  # noqa: E402


def kernels(tr, sc, n=6):
    for _ in range(2):
        tr.step(sc)
    tr.synchronize()
    tr.profile(True)
    for _ in range(n):
        tr.step(sc)
    tr.synchronize()
    out = {k: round(v["ms"] / v["launches"], 4) for k, v in tr.profile_read().items()}
    tr.profile(False)
    return out


def main():
    X = make_interactions("ml20m")
    mc = IALSModelConfigBuilder().set_K(64).set_alpha0(0.1).set_reg(1e-3).build()
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType.CHOLESKY).build()
    print(json.dumps({"full": kernels(IALSTrainer(mc, X), sc)}), flush=True)
    for M in (1 << 17, 1 << 15, 1 << 13, 1 << 11):
        Xc = X.tocsc()
        folded = sps.csc_matrix((Xc.data, (Xc.indices % M).astype(np.int32), Xc.indptr), shape=(M, X.shape[1]))
        Xf = folded.tocsr()  # duplicates stay separate entries: the same entry count per item
        print(json.dumps({"users_folded_to": M, "nnz": int(Xf.nnz), "kernels": kernels(IALSTrainer(mc, Xf), sc)}), flush=True)


if __name__ == "__main__":
    main()
