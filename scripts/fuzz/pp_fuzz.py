"""Randomised check of iALS++ (64-dim blocks): chained passes vs three-pass form vs oracle."""
import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import oracle as O
from conftest import row_rel_err
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer, SolverType, LossType
n_fail = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for seed in range(N):
    rng = np.random.default_rng(500 + seed)
    K = int(rng.integers(65, 257)); U = int(rng.integers(50, 400)); I = int(rng.integers(60, 3500))
    dens = float(rng.choice([0.02, 0.1, 0.5])); binary = rng.random() < 0.5
    X = sps.random(U, I, density=dens, format="csr", random_state=rng, dtype=np.float32)
    X.data[:] = 1.0 if binary else (1.0 + 3 * rng.random(X.nnz)).astype(np.float32)
    if I > 2200 and rng.random() < 0.7:  # rows above the workgroup threshold
        extra = sps.csr_matrix((rng.random((3, I)) < 0.85).astype(np.float32))
        X = sps.vstack([X, extra]).tocsr(); U += 3
    loss = "IALSPP" if rng.random() < 0.5 else "ORIGINAL"
    iters = int(rng.integers(1, 3))
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-2).set_nu(1.0).set_loss_type(getattr(LossType, loss)).build())
    omc = O.model_config(K, alpha0=0.1, reg=1e-2, nu=1.0, loss_type=loss)
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType.IALSPP).set_ialspp_subspace_dimension(64).set_ialspp_iteration(iters).build())
    osc = O.solver_config(8, "IALSPP", 3, ialspp_subspace_dimension=64, ialspp_iteration=iters)
    out = {}
    for flag in ("1", "0"):
        os.environ["IRSPACK_AMD_IALSPP_CHAIN"] = flag
        t = IALSTrainer(mc, X)
        t.step(sc)
        out[flag] = (t.user, t.item)
    o = O.IALSTrainer(omc, X); o.step(osc)
    e1 = max(row_rel_err(out["1"][0], out["0"][0]), row_rel_err(out["1"][1], out["0"][1]))
    e2 = max(row_rel_err(out["1"][0], o.user), row_rel_err(out["1"][1], o.item))
    ok = e1 < 1e-4 and e2 < 2e-3
    if not ok:
        n_fail += 1
    print("seed", seed, "K", K, "U", U, "I", I, "dens", dens, "binary", binary, loss, "iters", iters, "chain-vs-3pass %.2e chain-vs-oracle %.2e" % (e1, e2), "" if ok else "FAIL", flush=True)
print("failures:", n_fail)
