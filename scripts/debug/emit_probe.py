import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.evaluation._core_evaluator import EvaluatorCore
from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder, IALSSolverConfigBuilder, IALSTrainer
rng2 = np.random.default_rng(11)
U, I, K, cutoff = 700, 9000, 24, 20
mc = IALSModelConfigBuilder().set_K(K).build(); sc = IALSSolverConfigBuilder().build()
t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
t.user = rng2.integers(-2, 3, size=(U, K)).astype(np.float32)
t.item = rng2.integers(-2, 3, size=(I, K)).astype(np.float32)
mask = sps.random(U, I, density=0.05, format="csr", random_state=rng2, dtype=np.float32); mask.data[:] = 1.0
gt = sps.random(U, I, density=0.002, format="csr", random_state=rng2, dtype=np.float64); gt.data[:] = 1.0
core = EvaluatorCore(gt, [])
scores = t.user_scores(0, U, sc); scores[mask.nonzero()] = -np.inf
bad = []
for u in range(U):
    m1 = sps.csr_matrix(mask[u:u + 1])
    os.environ["IRSPACK_AMD_EVAL_EMIT"] = "1"; a = core.get_metrics_ials(t, u, u + 1, m1, cutoff, u, True)
    os.environ["IRSPACK_AMD_EVAL_EMIT"] = "0"; b = core.get_metrics_ials(t, u, u + 1, m1, cutoff, u, True)
    if not np.array_equal(a.item_cnt, b.item_cnt):
        bad.append(u)
        if len(bad) <= 5:
            s = scores[u]; order = np.lexsort((np.arange(I), -s)); top = order[:cutoff]
            samp = np.sort(s[:2048][np.isfinite(s[:2048])])[::-1]
            tau = samp[cutoff - 1]
            print("user", u, "n_gt", gt[u].nnz, "tau(sample)", tau, "cands>=tau", int((s >= tau).sum()), "20th", s[top[-1]],
                  "emit items", np.flatnonzero(a.item_cnt)[:25], "ref", np.sort(top)[:25], "two-pass", np.flatnonzero(b.item_cnt)[:25])
print("differing users", len(bad), bad[:20])
import oracle as O
ocore = O.EvaluatorCore(gt, [])
for u in bad[:4]:
    s = scores[u]
    order = np.lexsort((np.arange(I), -s)); top = order[:cutoff]
    m = core.get_metrics_f32(scores[u:u + 1], cutoff, u, 1, True)
    om = ocore.get_metrics_f32(scores[u:u + 1], cutoff, u, 1, True)
    got = np.flatnonzero(m.item_cnt); want = np.flatnonzero(om.item_cnt())
    extra = sorted(set(got) - set(want)); miss = sorted(set(want) - set(got))
    print("user", u, "get_metrics_f32 == oracle:", np.array_equal(got, want), "extra", extra, s[extra], "missing", miss, s[miss],
          "20th", s[top[-1]], "#>=20th", int((s >= s[top[-1]]).sum()), "#>20th", int((s > s[top[-1]]).sum()), "lane of missing", [int(x) % 64 for x in miss],
          "same-lane better-or-equal before missing:", [int(((s[(x % 64)::64] >= s[x]) & (np.arange(x % 64, I, 64) < x)).sum()) for x in miss])
