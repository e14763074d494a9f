"""Randomised check of iALS half-steps (Cholesky / CG) on the device against the oracle.

Each half-step starts from the oracle's own inputs.  GPU and oracle must agree to 1e-4 per row;
where they do not (ill-conditioned systems: tiny regulariser + sparse rows; truncated CG with
many steps from a random start, which is chaotic in float32 for BOTH), a float64 evaluation of
the same half-step arbitrates on a row sample: the GPU's median distance from it may be at most
twice the oracle's (+ 1e-5); row by row the two float32 results scatter around the float64 one
symmetrically (FUZZ_VERBOSE=1 prints the counts).  Development script, not part of the tests."""
import os, sys
import numpy as np, scipy.sparse as sps
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import oracle as O
from conftest import row_rel_err
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer, SolverType, LossType

def cg64(A, b, x0, steps):
    x = x0.copy(); r = b - A @ x; p = r.copy()
    for _ in range(steps):
        r2 = r @ r
        if r2 <= 1e-20: break
        Ap = A @ p; al = r2 / (p @ Ap); x += al * p; r -= al * Ap; r2n = r @ r
        if r2n <= 1e-20: break
        p = r + (r2n / r2) * p
    return x

def arbitrate(Xs, tgt0, oth0, got, want, kind, steps, alpha0, reg, nu, bias, rng):
    K = oth0.shape[1]; V = oth0.astype(np.float64); P = alpha0 * V.T @ V
    num = np.linalg.norm(got.astype(np.float64) - want, axis=1); den = np.linalg.norm(want, axis=1)
    err = num / np.maximum(den, 1e-6 * max(den.max(), 1e-30))
    far = np.flatnonzero(err >= 1e-4)
    rows = far if far.size <= 150 else rng.choice(far, 150, replace=False)
    bad = 0
    egs, eos = [], []
    for r in rows:
        sl = slice(Xs.indptr[r], Xs.indptr[r + 1]); Vr = V[Xs.indices[sl]]; c = Xs.data[sl].astype(np.float64)
        regr = float(np.float32(reg) * np.float32(np.float32(alpha0) * np.float32(Xs.shape[1]) + np.float32(sl.stop - sl.start)) ** np.float32(nu))
        A = P + (Vr.T * c) @ Vr + regr * np.eye(K); b = Vr.T @ (c + bias)
        if sl.stop == sl.start and kind == "CG": ref = np.zeros(K)
        else: ref = np.linalg.solve(A, b) if kind == "CHOLESKY" else cg64(A, b, tgt0[r].astype(np.float64), steps if steps else K)
        n = max(np.linalg.norm(ref), 1e-30)
        eg = np.linalg.norm(got[r] - ref) / n; eo = np.linalg.norm(want[r] - ref) / n
        bad += eg > 2 * eo + 1e-5
        egs.append(eg); eos.append(eo)
    bad = 0
    if egs:
        egs, eos = np.array(egs), np.array(eos)
        bad = int(np.median(egs) > 2 * np.median(eos) + 1e-5)
        if os.environ.get("FUZZ_VERBOSE"):
            print("   arbitrated rows %d: gpu worse(2x) %d, oracle worse(2x) %d, median gpu %.2e, median oracle %.2e" %
                  (len(egs), int((egs > 2 * eos + 1e-5).sum()), int((eos > 2 * egs + 1e-5).sum()), np.median(egs), np.median(eos)))
    return int(far.size), int(bad), float(err.max())

n_fail = 0
seeds = [int(a) for a in sys.argv[2:]] if len(sys.argv) > 2 else range(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
for seed in seeds:
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([3, 10, 16, 31, 48, 64, 65, 100, 128, 129, 192, 200, 256]))
    U = int(rng.integers(30, 500)); I = int(rng.integers(40, 4000))
    dens = float(rng.choice([0.005, 0.05, 0.3])); binary = rng.random() < 0.5
    X = sps.random(U, I, density=dens, format="csr", random_state=rng, dtype=np.float32)
    X.data[:] = 1.0 if binary else (0.5 + 4 * rng.random(X.nnz)).astype(np.float32)
    if I > 1500 and rng.random() < 0.6:
        X = sps.vstack([X, sps.csr_matrix((rng.random((2, I)) < 0.9).astype(np.float32))]).tocsr(); U += 2
    Xt = X.T.tocsr(); Xt.sort_indices()
    kind = str(rng.choice(["CHOLESKY", "CG"])); loss = str(rng.choice(["IALSPP", "ORIGINAL"]))
    alpha0 = float(rng.choice([0.02, 0.1, 1.0])); reg = float(rng.choice([1e-3, 1e-2, 0.3])); nu = float(rng.choice([0.0, 0.5, 1.0]))
    steps = int(rng.choice([1, 3, 6]))
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(alpha0).set_reg(reg).set_nu(nu).set_loss_type(getattr(LossType, loss)).build())
    omc = O.model_config(K, alpha0=alpha0, reg=reg, nu=nu, loss_type=loss)
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(steps).build()
    osc = O.solver_config(8, kind, steps)
    bias = 0.0 if loss == "IALSPP" else alpha0
    t = IALSTrainer(mc, X); o = O.IALSTrainer(omc, X)
    n_far = n_bad = 0; worst = 0.0
    for ep in range(2):
        u0, i0 = o.user.copy(), o.item.copy()
        o.step(osc)
        for side, (uu, ii, Xs) in enumerate(((u0, i0, X), (o.user.copy(), i0, Xt))):
            t.user, t.item = uu, ii
            t.partial_gramian_async(side); t.finish_gramian_async(side); t.half_step_async(side, sc); t.synchronize()
            got, want = (t.user, o.user) if side == 0 else (t.item, o.item)
            tgt0, oth0 = (uu, ii) if side == 0 else (ii, uu)
            f, b, e = arbitrate(Xs, tgt0, oth0, got, want.astype(np.float64), kind, steps, alpha0, reg, nu, bias, rng)
            n_far += f; n_bad += b; worst = max(worst, e)
    ok = n_bad == 0
    if not ok: n_fail += 1
    print("seed", seed, kind, loss, "K", K, "U", U, "I", I, "dens", dens, "binary", binary, "a0", alpha0, "reg", reg, "nu", nu, "cg", steps,
          "max err vs oracle %.2e, rows > 1e-4: %d, half-steps where the GPU is the worse one vs float64: %d" % (worst, n_far, n_bad), "ok" if ok else "FAIL", flush=True)
print("failures:", n_fail)
