"""Randomised check of the fused iALS evaluator (all paths) against the oracle."""
import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as O
from irspack_amd.evaluation._core_evaluator import EvaluatorCore
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer

def same(m, om):
    d, od = m.as_dict(), om.as_dict()
    assert np.array_equal(m.item_cnt, om.item_cnt()), "item histogram"
    for k in ("valid_user", "total_user"):
        assert d[k] == od[k], k
    for k in ("hit", "recall", "ndcg", "map", "precision"):
        assert abs(d[k] - od[k]) <= 1e-12 * max(1.0, abs(od[k])), (k, d[k], od[k])

n_fail = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(1000 + seed)
    U = int(rng.integers(1, 2500)); I = int(rng.integers(8192, 14000)); K = int(rng.choice([8, 16, 24, 32, 48, 64, 100, 128, 192, 256]))
    cutoff = int(rng.integers(1, 33)); integer = rng.random() < 0.35; skew = rng.choice([0.0, 0.3, 0.8, 1.5])
    mc = IALSModelConfigBuilder().set_K(K).build(); sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
    scale = ((1.0 + rng.permutation(I)) ** -skew).astype(np.float32)
    if integer:
        user = rng.integers(-2, 3, size=(U, K)).astype(np.float32); item = rng.integers(-2, 3, size=(I, K)).astype(np.float32)
        item[rng.random(I) < 0.3] *= 2
    else:
        user = rng.standard_normal((U, K)).astype(np.float32); item = rng.standard_normal((I, K)).astype(np.float32) * scale[:, None]
    if rng.random() < 0.3: user[rng.integers(0, U, max(1, U // 20))] = 0.0
    t.user, t.item = user, item
    dens = float(rng.choice([0.0, 0.001, 0.02, 0.2]))
    mask = None
    if dens > 0:
        mask = sps.random(U, I, density=dens, format="csr", random_state=rng, dtype=np.float32); mask.data[:] = 1.0
        if rng.random() < 0.4:  # a few users have seen nearly everything popular
            top = np.argsort(-np.linalg.norm(item, axis=1))[: int(rng.integers(400, 6000))]
            m2 = mask.tolil()
            for u in rng.integers(0, U, 3): m2[u, top] = 1.0
            mask = sps.csr_matrix(m2); mask.data[:] = 1.0
    gt = sps.random(U, I, density=float(rng.choice([0.0005, 0.003])), format="csr", random_state=rng, dtype=np.float64); gt.data[:] = 1.0
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    b = int(rng.integers(0, max(1, U // 3))); e = int(rng.integers(b + 1, U + 1)); rwc = bool(rng.random() < 0.5)
    scores = t.user_scores(b, e, sc)
    msub = None if mask is None else sps.csr_matrix(mask[b:e])
    if msub is not None: scores[msub.nonzero()] = -np.inf
    want = ocore.get_metrics_f32(scores, cutoff, b, 4, rwc)
    for env in ({}, {"IRSPACK_AMD_EVAL_BOUND": "0"}, {"IRSPACK_AMD_EVAL_EMIT": "0"}, {"IRSPACK_AMD_EVAL_PASS_ROWS": "128"}, {"IRSPACK_AMD_EVAL_SAMPLE": "64"}):
        for k in ("IRSPACK_AMD_EVAL_BOUND", "IRSPACK_AMD_EVAL_EMIT", "IRSPACK_AMD_EVAL_PASS_ROWS", "IRSPACK_AMD_EVAL_SAMPLE"): os.environ.pop(k, None)
        os.environ.update(env)
        try:
            got = core.get_metrics_ials(t, b, e, msub, cutoff, b, rwc)
            same(got, want)
        except AssertionError as ex:
            n_fail += 1
            print("FAIL seed", seed, "U", U, "I", I, "K", K, "cutoff", cutoff, "integer", integer, "skew", skew, "dens", dens, "rows", (b, e), "env", env, core.last_call_stats(), ex, flush=True)
    if seed % 10 == 9: print("seed", seed, "done, failures", n_fail, flush=True)
print("failures:", n_fail)
