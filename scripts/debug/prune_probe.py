"""How much of the user x item score block could a norm bound skip? (development probe)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import holdout, model_config, solver_config
from irspack_amd.recommenders._ials_core import IALSTrainer
from irspack_amd.synthetic import make_interactions
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
X = make_interactions("ml20m")
tr = IALSTrainer(model_config(K), X)
rng = np.random.default_rng(0)
for ep in range(1, 11):
    tr.step(solver_config("CG"))
    if ep not in (1, 3, 10):
        continue
    Uf, If = tr.user.astype(np.float64), tr.item.astype(np.float64)
    inorm = np.linalg.norm(If, axis=1)
    order = np.argsort(-inorm)
    rows = np.sort(rng.choice(X.shape[0], 8192, replace=False))
    S = Uf[rows] @ If.T
    # seen items are masked in the evaluator; ignore here (upper bound on tau anyway)
    samp = S[:, order[:2048]]
    tau = np.partition(samp, -20, axis=1)[:, -20]
    tau0 = np.partition(S[:, :2048], -20, axis=1)[:, -20]
    true20 = np.partition(S, -20, axis=1)[:, -20]
    unorm = np.linalg.norm(Uf[rows], axis=1)
    r = tau / np.maximum(unorm, 1e-30)
    sn = inorm[order]
    # per user: items that survive the bound
    keep_user = np.searchsorted(-sn, -r, side="right") / len(sn)
    # tiles of 64 users sorted by r
    rs = np.sort(r)
    keep_tile = np.array([np.searchsorted(-sn, -rs[i:i + 64].min(), side="right") for i in range(0, len(rs), 64)]) / len(sn)
    # round the item count up to a 64-item tile
    cand = (S >= tau[:, None]).sum(1)
    cand0 = (S >= tau0[:, None]).sum(1)
    print("K", K, "epoch", ep, "item norm pct 10/50/90/99", np.percentile(inorm, [10, 50, 90, 99]).round(3),
          "user norm median", np.median(unorm).round(3), "tau median", np.median(tau).round(3), "true20 median", np.median(true20).round(3))
    print("   work kept: per-user bound %.3f, 64-user tiles %.3f; candidates/user norm-sorted sample %.0f (first-2048 sample %.0f)"
          % (keep_user.mean(), keep_tile.mean(), cand.mean(), cand0.mean()), flush=True)
