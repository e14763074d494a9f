"""Fused evaluator on a million-item catalogue: multi-pass bounded path vs the two-pass path."""
import os, sys, time
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.evaluation._core_evaluator import EvaluatorCore
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSTrainer
rng = np.random.default_rng(0)
U, I, K, cutoff = 40000, 1_000_000, 32, 20
mc = IALSModelConfigBuilder().set_K(K).build()
t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
scale = ((1.0 + np.arange(I)) ** -0.6).astype(np.float32)
rng.shuffle(scale)
t.user = rng.standard_normal((U, K)).astype(np.float32)
t.item = rng.standard_normal((I, K)).astype(np.float32) * scale[:, None]
nm = 30
mask = sps.csr_matrix((np.ones(U * nm, np.float32), rng.integers(0, I, U * nm).astype(np.int32), np.arange(U + 1) * nm), shape=(U, I))
mask.sum_duplicates()
gt = sps.csr_matrix((np.ones(U), (np.arange(U), rng.integers(0, I, U))), shape=(U, I))
core = EvaluatorCore(gt, [])
res = {}
for emit in ("1", "0"):
    os.environ["IRSPACK_AMD_EVAL_EMIT"] = emit
    core.get_metrics_ials(t, 0, 2048, sps.csr_matrix(mask[:2048]), cutoff, 0, False)
    t0 = time.perf_counter()
    m = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    w1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    m = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    w2 = time.perf_counter() - t0
    res[emit] = m
    print("emit", emit, "wall first %.3f s second %.3f s" % (w1, w2), core.last_call_stats(), "ndcg", m.as_dict()["ndcg"], flush=True)
print("item_cnt equal:", np.array_equal(res["1"].item_cnt, res["0"].item_cnt), "valid", res["1"].valid_user, res["0"].valid_user)
