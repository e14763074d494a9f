import os, sys
import numpy as np, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from irspack_amd.evaluation._core_evaluator import EvaluatorCore
rng = np.random.default_rng(11)
U, I, K, cutoff = 700, 9000, 24, 20
user = rng.integers(-2, 3, size=(U, K)).astype(np.float32)
item = rng.integers(-2, 3, size=(I, K)).astype(np.float32)
mask = sps.random(U, I, density=0.05, format="csr", random_state=rng, dtype=np.float32)
gt = sps.random(U, I, density=0.002, format="csr", random_state=rng, dtype=np.float64); gt.data[:] = 1.0
scores = user @ item.T
scores[mask.nonzero()] = -np.inf
core = EvaluatorCore(gt, [])
nbad = 0
for u in range(U):
    s = scores[u]
    top = np.lexsort((np.arange(I), -s))[:cutoff]
    m = core.get_metrics_f32(scores[u:u + 1], cutoff, u, 1, True)
    got = np.flatnonzero(m.item_cnt)
    if not np.array_equal(got, np.sort(top)):
        nbad += 1
        if nbad <= 6:
            extra = sorted(set(got) - set(top)); miss = sorted(set(top) - set(got))
            print("user", u, "extra", extra, s[extra], "missing", miss, s[miss], "20th", s[top[-1]],
                  "#>=", int((s >= s[top[-1]]).sum()), "#>", int((s > s[top[-1]]).sum()), flush=True)
print("wave=%s differing rows %d of %d" % (os.environ.get("IRSPACK_AMD_EVAL_WAVE", "1"), nbad, U))
