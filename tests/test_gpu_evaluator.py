"""GPU parity tests of the evaluator path vs the CPU oracle and the sklearn / hand-loop
checks of the reference (tests/evaluation/test_evaluator.py:19-152, 358-398;
tests/evaluation/test_restricted_evaluator.py:25-108).  Bar: counters and item
histogram bit-exact; fp64 metric sums within 1e-12 relative (sum order).
"""
import pickle

import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from irspack_amd.evaluation._core_evaluator import EvaluatorCore, Metrics, evaluate_list_vs_list

pytestmark = pytest.mark.gpu

KEYS = O.METRIC_KEYS


def compare(m: Metrics, om: "O.Metrics"):
    np.testing.assert_array_equal(m.item_cnt, om.item_cnt())  # bit-exact histogram
    raw = om.raw()
    assert m.valid_user == int(raw[0]) and m.total_user == int(raw[1])  # bit-exact counters
    np.testing.assert_allclose([m.hit, m.recall, m.ndcg, m.precision, m.map], raw[2:], rtol=1e-12)
    d, od = m.as_dict(), om.as_dict()
    for k in KEYS:
        assert d[k] == pytest.approx(od[k], rel=1e-12, abs=1e-15), k


@pytest.mark.parametrize("U,I,dtype", [(10, 5, "float32"), (10, 30, "float64"), (3000, 5, "float32"),
                                      (200, 1000, "float32"), (64, 5000, "float64")])
@pytest.mark.parametrize("cutoff_kind", ["full", "small"])
def test_random_blocks_match_oracle(U, I, dtype, cutoff_kind):
    rns = np.random.RandomState(42)
    scores = rns.randn(U, I).astype(dtype)
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.7).astype(np.float64))
    cutoff = I if cutoff_kind == "full" else max(1, min(I // 2, 20))  # 5000 > 2048: the BIG lists
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    f = "get_metrics_f64" if dtype == "float64" else "get_metrics_f32"
    for rwc in (False, True):
        compare(getattr(core, f)(scores, cutoff, 0, 4, rwc), getattr(ocore, f)(scores, cutoff, 0, 4, rwc))


def test_vs_sklearn():
    # tests/evaluation/test_evaluator.py:19-49
    from sklearn.metrics import average_precision_score, ndcg_score

    for U, I, dtype in [(10, 5, "float32"), (10, 30, "float64"), (300, 5, "float32")]:
        rns = np.random.RandomState(42)
        scores = rns.randn(U, I).astype(dtype)
        X_gt = (rns.rand(U, I) >= 0.7).astype(np.float64)
        core = EvaluatorCore(sps.csr_matrix(X_gt), [])
        f = core.get_metrics_f64 if dtype == "float64" else core.get_metrics_f32
        d = f(scores, I, 0, 4).as_dict()
        maps, ndcgs = [], []
        for i in range(U):
            if X_gt[i].sum() == 0:
                continue
            maps.append(average_precision_score(X_gt[i], scores[i]))
            ndcgs.append(ndcg_score(X_gt[i][None, :], scores[i][None, :]))
        assert d["map"] == pytest.approx(np.mean(maps), abs=1e-8)
        assert d["ndcg"] == pytest.approx(np.mean(ndcgs), abs=1e-8)


def test_ties_and_neg_inf():
    # ties -> lower index first (evaluator.cpp:329,353-355); -inf never ranked (:328)
    I = 300
    scores = np.zeros((6, I), dtype=np.float32)
    scores[0, :] = 1.0                       # all tied
    scores[1, ::2] = 2.0                     # ties in two groups
    scores[2, :] = -np.inf                   # nothing rankable
    scores[2, 7] = 3.0
    scores[3, :] = -np.inf                   # fully masked
    scores[4, :] = np.linspace(1, 0, I)
    scores[5, :] = -0.0
    scores[5, 10:20] = 0.0                   # -0.0 == +0.0
    gt = sps.csr_matrix((np.random.RandomState(1).rand(6, I) > 0.9).astype(np.float64))
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    for cutoff in (1, 5, 20, 299, 300):
        compare(core.get_metrics_f32(scores, cutoff, 0, 1), ocore.get_metrics_f32(scores, cutoff, 0, 1))
    # tests/evaluation/test_evaluator.py:358-368
    one = EvaluatorCore(sps.csr_matrix(np.asarray([[1.0, 1.0, 0.0]])), [])
    d = one.get_metrics_f64(np.asarray([[1.0, -np.inf, -np.inf]]), 3, 0, 1).as_dict()
    assert d["precision"] == 1.0 and d["recall"] == 0.5 and d["hit"] == 1.0


def test_offset_and_chunking():
    # mb_size invariance (tests/evaluation/test_evaluator.py:98-106) and offset semantics
    rns = np.random.RandomState(3)
    U, I = 57, 400
    scores = rns.randn(U, I)
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.6).astype(np.float64))
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    whole = core.get_metrics_f64(scores, 10, 0, 2)
    acc, oacc = Metrics(I), O.Metrics(I)
    for b in range(0, U, 13):
        e = min(b + 13, U)
        acc.merge(core.get_metrics_f64(scores[b:e], 10, b, 2))
        oacc.merge(ocore.get_metrics_f64(scores[b:e], 10, b, 2))
    compare(acc, oacc)
    np.testing.assert_array_equal(acc.item_cnt, whole.item_cnt)
    assert acc.as_dict()["ndcg"] == pytest.approx(whole.as_dict()["ndcg"], rel=1e-12)


@pytest.mark.parametrize("mode", ["global", "per_user"])
def test_restricted_items(mode):
    # tests/evaluation/test_restricted_evaluator.py:25-108
    rns = np.random.RandomState(5)
    U, I = 40, 120
    scores = rns.randn(U, I).astype(np.float32)
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.8).astype(np.float64))
    if mode == "global":
        rec = [sorted(rns.choice(I, size=50, replace=False).tolist())[::-1]]  # unsorted input
    else:
        rec = [rns.choice(I, size=rns.randint(0, 40), replace=False).tolist() for _ in range(U)]
    core, ocore = EvaluatorCore(gt, rec), O.EvaluatorCore(gt, rec)
    for cutoff in (3, 20):
        compare(core.get_metrics_f32(scores, cutoff, 0, 3), ocore.get_metrics_f32(scores, cutoff, 0, 3))


def test_argument_validation():
    # evaluator.cpp:187-205, 263-268; tests/evaluation/test_restricted_evaluator.py:130-164
    gt = sps.csr_matrix(np.eye(4))
    with pytest.raises(ValueError):
        EvaluatorCore(gt, [[0], [1]])          # size not in {0, 1, U}
    with pytest.raises(ValueError):
        EvaluatorCore(gt, [[0, 0]])            # duplicates
    with pytest.raises(ValueError):
        EvaluatorCore(gt, [[4]])               # index >= n_items
    core = EvaluatorCore(gt, [])
    s = np.zeros((4, 4), dtype=np.float32)
    with pytest.raises(ValueError):
        core.get_metrics_f32(s, 0, 0, 1)       # cutoff == 0
    with pytest.raises(ValueError):
        core.get_metrics_f32(s, 5, 0, 1)       # cutoff > n_items
    with pytest.raises(ValueError):
        core.get_metrics_f32(s, 2, 4, 1)       # offset >= n_users
    with pytest.raises(ValueError):
        core.get_metrics_f32(s, 2, 1, 1)       # offset + rows > n_users
    with pytest.raises(ValueError):
        core.get_metrics_f32(s, 2, 0, 0)       # n_threads == 0


def test_pickle_and_list_vs_list():
    rns = np.random.RandomState(9)
    U, I = 30, 50
    scores = rns.randn(U, I)
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.7).astype(np.float64))
    core = EvaluatorCore(gt, [])
    core2 = pickle.loads(pickle.dumps(core))
    a, b = core.get_metrics_f64(scores, 7, 0, 1), core2.get_metrics_f64(scores, 7, 0, 1)
    assert a.as_dict() == b.as_dict()
    # tests/evaluation/test_df_vs_df.py:50-64: list-vs-list equals the evaluator on the same ranking
    recs = [list(np.argsort(-scores[u], kind="stable")[:7]) for u in range(U)]
    gts = [gt[u].indices.tolist() for u in range(U)]
    keep = [u for u in range(U) if len(gts[u]) > 0]
    m = evaluate_list_vs_list([recs[u] for u in keep], [gts[u] for u in keep], I, 2)
    om = O.evaluate_list_vs_list([recs[u] for u in keep], [gts[u] for u in keep], I, 2)
    for k in ("hit", "ndcg", "recall", "map", "precision", "entropy", "gini_index"):
        assert m.as_dict()[k] == pytest.approx(om.as_dict()[k], rel=1e-12)
        assert m.as_dict()[k] == pytest.approx(a.as_dict()[k], rel=1e-9)


@pytest.mark.parametrize("emit", ["1", "0"])
def test_fused_ials_path_matches_block_path(emit, monkeypatch):
    # score + mask + rank on the device == user_scores -> mask -> get_metrics_f32
    # (both device implementations: the threshold-filtered default and the two-pass one)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_EMIT", emit)
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer)
    from irspack_amd.synthetic import holdout_split, make_interactions

    X = make_interactions("small")
    tr, te = holdout_split(X, 0.2, 7)
    mc = IALSModelConfigBuilder().set_K(32).set_alpha0(0.1).set_reg(1e-2).build()
    sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, tr.astype(np.float32))
    for _ in range(2):
        t.step(sc)
    core = EvaluatorCore(te, [])
    U = X.shape[0]
    fused = core.get_metrics_ials(t, 0, U, tr, 20, 0, False)
    acc = Metrics(X.shape[1])
    for b in range(0, U, 500):
        e = min(b + 500, U)
        s = t.user_scores(b, e, sc)
        s[tr[b:e].nonzero()] = -np.inf
        acc.merge(core.get_metrics_f32(s, 20, b, 1))
    np.testing.assert_array_equal(fused.item_cnt, acc.item_cnt)
    assert fused.valid_user == acc.valid_user and fused.total_user == acc.total_user
    for k in ("hit", "ndcg", "recall", "map", "precision"):
        assert getattr(fused, k) == pytest.approx(getattr(acc, k), rel=1e-12)
    # the device copy of the mask is reused for the same object, replaced for another one and
    # dropped for None
    again = core.get_metrics_ials(t, 0, U, tr, 20, 0, False)
    np.testing.assert_array_equal(again.item_cnt, fused.item_cnt)
    assert again.ndcg == fused.ndcg
    other = sps.csr_matrix(tr.shape, dtype=np.float32)  # empty mask: training items are ranked too
    unmasked = core.get_metrics_ials(t, 0, U, other, 20, 0, False)
    nomask = core.get_metrics_ials(t, 0, U, None, 20, 0, False)
    np.testing.assert_array_equal(unmasked.item_cnt, nomask.item_cnt)
    assert unmasked.ndcg == nomask.ndcg and (nomask.item_cnt != fused.item_cnt).any()
    back = core.get_metrics_ials(t, 0, U, tr, 20, 0, False)
    np.testing.assert_array_equal(back.item_cnt, fused.item_cnt)
    # stored zeros of the mask are not masked (the reference uses mask.nonzero(),
    # evaluator.py:426-432), and an in-place edit of the resident mask is noticed
    holes = tr.copy().astype(np.float32)
    holes.data[::3] = 0.0
    pruned = holes.copy()
    pruned.eliminate_zeros()
    with_holes = core.get_metrics_ials(t, 0, U, holes, 20, 0, False)
    expect = core.get_metrics_ials(t, 0, U, pruned, 20, 0, False)
    np.testing.assert_array_equal(with_holes.item_cnt, expect.item_cnt)
    assert with_holes.ndcg == expect.ndcg and (expect.item_cnt != fused.item_cnt).any()
    edited = tr.copy().astype(np.float32)
    first = core.get_metrics_ials(t, 0, U, edited, 20, 0, False)
    edited.data[:] = 0.0  # same object, same nnz: now nothing is masked
    second = core.get_metrics_ials(t, 0, U, edited, 20, 0, False)
    np.testing.assert_array_equal(first.item_cnt, fused.item_cnt)
    np.testing.assert_array_equal(second.item_cnt, nomask.item_cnt)
    # an edit of one unsampled value (same nnz and pointers): the default, whole-array fingerprint
    # notices it; the opt-in SAMPLED fingerprint cannot see it - invalidate_mask() picks it up
    step = max(1, tr.data.size // 1024)
    assert type(core).strict_mask_fingerprint is True
    sneaky2 = tr.copy().astype(np.float32)
    core.get_metrics_ials(t, 0, U, sneaky2, 20, 0, False)
    sneaky2.data[np.arange(sneaky2.data.size) % step != 0] = 0.0  # every unsampled entry: no longer masked
    strict = core.get_metrics_ials(t, 0, U, sneaky2, 20, 0, False)
    assert (strict.item_cnt != fused.item_cnt).any()
    type(core).strict_mask_fingerprint = False
    try:
        sneaky = tr.copy().astype(np.float32)
        core.get_metrics_ials(t, 0, U, sneaky, 20, 0, False)
        sneaky.data[np.arange(sneaky.data.size) % step != 0] = 0.0
        stale = core.get_metrics_ials(t, 0, U, sneaky, 20, 0, False)
        np.testing.assert_array_equal(stale.item_cnt, fused.item_cnt)  # (the documented blind spot)
        core.invalidate_mask()
        fresh = core.get_metrics_ials(t, 0, U, sneaky, 20, 0, False)
        np.testing.assert_array_equal(fresh.item_cnt, strict.item_cnt)
    finally:
        type(core).strict_mask_fingerprint = True
    # cutoffs on the wave-per-row kernel with 8 entries per lane, and on the general kernel
    for cutoff in (50, 64, 100):
        f2 = core.get_metrics_ials(t, 0, U, tr, cutoff, 0, True)
        a2 = Metrics(X.shape[1])
        for b in range(0, U, 500):
            e = min(b + 500, U)
            s = t.user_scores(b, e, sc)
            s[tr[b:e].nonzero()] = -np.inf
            a2.merge(core.get_metrics_f32(s, cutoff, b, 1, True))
        np.testing.assert_array_equal(f2.item_cnt, a2.item_cnt)
        assert f2.ndcg == pytest.approx(a2.ndcg, rel=1e-12) and f2.recall == pytest.approx(a2.recall, rel=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("levels,cutoff", [(1, 10), (3, 20), (40, 64), (3, 700), (100000, 1500)])
def test_many_ties_short_list_and_general_selection(dtype, levels, cutoff):
    """Rows whose scores take few distinct values: thousands of keys tie at the cutoff, so the
    short-list ranking overflows and the general selection (ties by lowest index) runs; with
    many levels the short list is used, also for cutoffs above one wave."""
    rns = np.random.RandomState(7)
    U, I = 48, 6000
    scores = rns.randint(0, levels, size=(U, I)).astype(dtype) / 7.0
    scores[rns.rand(U, I) < 0.05] = -np.inf
    scores[1, :] = -np.inf  # nothing rankable
    scores[2, :5990] = -np.inf  # fewer rankable items than the cutoff
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.97).astype(np.float64))
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    f = "get_metrics_f64" if dtype == "float64" else "get_metrics_f32"
    compare(getattr(core, f)(scores, cutoff, 0, 4, False), getattr(ocore, f)(scores, cutoff, 0, 4, False))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("cutoff", [10, 20, 24, 32, 64])
def test_moderately_tied_rows_keep_index_order(dtype, cutoff):
    """Integer-valued scores with a bell-shaped distribution: a handful of items tie at the
    cutoff-th score, few enough that the one-wave kernel keeps the row.  An entry pushed down
    its lane's list by a better late arrival has to stay ahead of the equal entries behind it
    (lowest index first, evaluator.cpp:329, 353-355); round 1's lists dropped it instead."""
    rns = np.random.default_rng(11)
    U, I, K = 700, 9000, 24
    user = rns.integers(-2, 3, size=(U, K)).astype(np.float64)
    item = rns.integers(-2, 3, size=(I, K)).astype(np.float64)
    scores = (user @ item.T).astype(dtype)
    scores[rns.random((U, I)) < 0.05] = -np.inf
    gt = sps.random(U, I, density=0.002, format="csr", random_state=rns, dtype=np.float64)
    gt.data[:] = 1.0
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    f = "get_metrics_f64" if dtype == "float64" else "get_metrics_f32"
    m = getattr(core, f)(scores, cutoff, 0, 4, False)
    compare(m, getattr(ocore, f)(scores, cutoff, 0, 4, False))
    want = np.zeros(I, dtype=np.int64)
    for u in np.flatnonzero(np.diff(gt.indptr)):
        np.add.at(want, np.lexsort((np.arange(I), -scores[u]))[:cutoff], 1)
    np.testing.assert_array_equal(m.item_cnt, want)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_nan_scores_rank_last_and_signed_zeros_tie(dtype):
    """The reference leaves NaN ordering undefined; this build ranks NaN after every other
    rankable score (ties by index), so replacing NaN by a value below all others must not
    change anything.  -0.0 and +0.0 tie (the reference compares floats)."""
    rns = np.random.RandomState(11)
    U, I = 40, 900
    scores = rns.randn(U, I).astype(dtype)
    scores[rns.rand(U, I) < 0.02] = np.nan
    scores[3, :] = np.nan
    scores[4, 10:] = -np.inf
    scores[4, :10] = np.nan
    scores[5, :] = 0.0
    scores[5, ::2] = -0.0
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.9).astype(np.float64))
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    f = "get_metrics_f64" if dtype == "float64" else "get_metrics_f32"
    replaced = np.where(np.isnan(scores), np.asarray(-1e30, dtype=dtype), scores)
    for cutoff in (5, 20, 300):
        compare(getattr(core, f)(scores, cutoff, 0, 2), getattr(ocore, f)(replaced, cutoff, 0, 2))


@pytest.mark.parametrize("cutoff", [5, 20, 32])
def test_emit_path_on_a_wide_catalogue(cutoff, monkeypatch):
    """The threshold-filtered path (catalogues of >= 8,192 items, cutoff <= 32): sample
    thresholds -> candidates -> exact ranking must give the two-pass path's metrics bit for
    bit (counters, histogram) on a model with many exactly tied scores (integer factors), a
    dense mask and users without ground truth; and both must equal the oracle."""
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer)

    rng2 = np.random.default_rng(11)
    U, I, K = 700, 9000, 24
    mc = IALSModelConfigBuilder().set_K(K).build()
    sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
    t.user = rng2.integers(-2, 3, size=(U, K)).astype(np.float32)  # small integers: many ties
    t.item = rng2.integers(-2, 3, size=(I, K)).astype(np.float32)
    mask = sps.random(U, I, density=0.05, format="csr", random_state=rng2, dtype=np.float32)
    mask.data[:] = 1.0
    gt = sps.random(U, I, density=0.002, format="csr", random_state=rng2, dtype=np.float64)
    gt.data[:] = 1.0
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    monkeypatch.setenv("IRSPACK_AMD_EVAL_EMIT", "1")
    a = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, True)
    part = core.get_metrics_ials(t, 100, 571, sps.csr_matrix(mask[100:571]), cutoff, 100, False)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_EMIT", "0")
    b = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, True)
    np.testing.assert_array_equal(a.item_cnt, b.item_cnt)
    assert (a.valid_user, a.total_user) == (b.valid_user, b.total_user)
    for k in ("hit", "ndcg", "recall", "map", "precision"):
        assert getattr(a, k) == pytest.approx(getattr(b, k), rel=1e-12)
    scores = t.user_scores(0, U, sc)
    scores[mask.nonzero()] = -np.inf
    compare(a, ocore.get_metrics_f32(scores, cutoff, 0, 4, True))
    compare(part, ocore.get_metrics_f32(scores[100:571], cutoff, 100, 4, False))


@pytest.mark.parametrize("K", [16, 48])
def test_norm_bound_pruning_keeps_the_lists(K, monkeypatch):
    """Bounded emit path on a catalogue with a popularity skew: item norms fall off like a
    power law, so most score tiles are pruned (Cauchy-Schwarz against the per-user threshold).
    The metrics must be those of the oracle on the full masked score block, including for the
    rows the filtered pass cannot finish: users who have seen every popular item (no threshold
    in the sample), users with an all-zero factor (every score ties at 0: list overflow), and
    users without ground truth (skipped)."""
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer)

    rng2 = np.random.default_rng(3)
    U, I, cutoff = 1500, 9000, 20
    mc = IALSModelConfigBuilder().set_K(K).build()
    sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
    pop = (1.0 + np.arange(I)) ** -0.7
    rng2.shuffle(pop)
    item = rng2.standard_normal((I, K)).astype(np.float32) * pop[:, None].astype(np.float32)
    user = rng2.standard_normal((U, K)).astype(np.float32)
    user[5] = 0.0  # all scores tie at 0
    user[6] = 0.0
    t.user, t.item = user, item
    mask = sps.random(U, I, density=0.01, format="lil", random_state=rng2, dtype=np.float32)
    top = np.argsort(-np.linalg.norm(item.astype(np.float64), axis=1))  # the path's item order
    for u in (10, 11, 700):  # these users have seen the 5000 items of largest norm
        mask[u, top[:5000]] = 1.0
    mask[12, top[:600]] = 1.0  # and this one the first sample
    mask = sps.csr_matrix(mask)
    mask.data[:] = 1.0
    gt = sps.random(U, I, density=0.003, format="lil", random_state=rng2, dtype=np.float64)
    gt[5, 17] = 1.0
    gt[6, 18] = 1.0
    gt[10, top[6000]] = 1.0
    gt[11, top[5500]] = 1.0
    gt[700, 3] = 1.0
    gt[12, top[700]] = 1.0
    gt[20] = 0  # no ground truth
    gt = sps.csr_matrix(gt)
    gt.data[:] = 1.0
    gt.eliminate_zeros()
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    scores = t.user_scores(0, U, sc)
    scores[mask.nonzero()] = -np.inf
    want = ocore.get_metrics_f32(scores, cutoff, 0, 4, False)
    a = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    st = core.last_call_stats()
    assert st["path"] == "emit_bounded" and st["tiles_scored"] < 0.5 * st["tiles_total"], st
    assert st["hard_rows"] >= 5, st  # 5, 6 (overflow), 10, 11, 700 (no threshold)
    compare(a, want)
    # the sample pass as three launches per block of users (round 5) instead of the one fused launch: the
    # thresholds are the same numbers, so the pruning statistics are too
    monkeypatch.setenv("IRSPACK_AMD_EVAL_SAMPLE_FUSED", "0")
    unfused = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    stu = core.last_call_stats()
    assert (stu["path"], stu["hard_rows"], stu["tiles_scored"]) == (st["path"], st["hard_rows"], st["tiles_scored"])
    compare(unfused, want)
    np.testing.assert_array_equal(a.item_cnt, unfused.item_cnt)
    monkeypatch.delenv("IRSPACK_AMD_EVAL_SAMPLE_FUSED")
    monkeypatch.setenv("IRSPACK_AMD_EVAL_BOUND", "0")
    b = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    assert core.last_call_stats()["path"] == "emit"
    compare(b, want)
    np.testing.assert_array_equal(a.item_cnt, b.item_cnt)
    # the same call in passes of 256 users (what a call over millions of users, or a catalogue
    # of a million items, does to keep its scratch bounded)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_BOUND", "1")
    monkeypatch.setenv("IRSPACK_AMD_EVAL_PASS_ROWS", "256")
    c = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    stc = core.last_call_stats()
    assert stc["path"] == "emit_bounded" and stc["tiles_total"] == st["tiles_total"], stc
    assert stc["hard_rows"] == st["hard_rows"]
    compare(c, want)
    np.testing.assert_array_equal(a.item_cnt, c.item_cnt)
    monkeypatch.delenv("IRSPACK_AMD_EVAL_PASS_ROWS")
    again = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)  # the bitmap cache is rebuilt
    np.testing.assert_array_equal(a.item_cnt, again.item_cnt)
    # a sub-block with an offset, and the tiny first sample (more second chances)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_BOUND", "1")
    monkeypatch.setenv("IRSPACK_AMD_EVAL_SAMPLE", "64")
    part = core.get_metrics_ials(t, 3, 1203, sps.csr_matrix(mask[3:1203]), cutoff, 3, True)
    compare(part, ocore.get_metrics_f32(scores[3:1203], cutoff, 3, 4, True))


@pytest.mark.parametrize("seed", range(8))
def test_one_launch_sample_pass_against_the_three_launch_form_and_the_oracle(seed, monkeypatch):
    """Random shapes for the sample pass in one launch (sample_tau_fused_kernel): user counts that are not
    multiples of its 16-user tiles, K = 16 .. 128, cutoffs 1 .. 32, masks from empty to covering most of the
    sample (users without a threshold), duplicated scores (ties at the cutoff).  The thresholds must be the
    numbers of the three-launch form - same pruning statistics, same hard rows - and the metrics the
    oracle's on the full masked score block."""
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer)

    rng2 = np.random.default_rng(100 + seed)
    U = int(rng2.integers(1, 700))
    I = int(rng2.integers(8192, 9500))
    K = int(rng2.choice([16, 32, 64, 128]))
    cutoff = int(rng2.choice([1, 5, 20, 32]))
    mc = IALSModelConfigBuilder().set_K(K).build()
    sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
    pop = ((1.0 + np.arange(I)) ** -0.6).astype(np.float32)
    rng2.shuffle(pop)
    item = rng2.standard_normal((I, K)).astype(np.float32) * pop[:, None]
    user = rng2.standard_normal((U, K)).astype(np.float32)
    if seed % 2:  # few distinct user rows and duplicated item rows: scores tie, also at the cutoff
        user = user[rng2.integers(0, max(1, U // 7 + 1), U)]
        item[rng2.integers(0, I, I // 3)] = item[rng2.integers(0, I, I // 3)]
    t.user, t.item = user, item
    density = [0.0, 0.002, 0.02, 0.1][seed % 4]
    mask = sps.random(U, I, density=density, format="lil", random_state=rng2, dtype=np.float32)
    top = np.argsort(-np.linalg.norm(item.astype(np.float64), axis=1))
    for u in rng2.integers(0, U, 3):  # these users have seen most of the sample
        mask[int(u), top[:int(rng2.integers(490, 4000))]] = 1.0
    mask = sps.csr_matrix(mask)
    mask.data[:] = 1.0
    gt = sps.csr_matrix(sps.random(U, I, density=0.004, format="csr", random_state=rng2, dtype=np.float64))
    gt.data[:] = 1.0
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    scores = t.user_scores(0, U, sc)
    scores[mask.nonzero()] = -np.inf
    want = ocore.get_metrics_f32(scores, cutoff, 0, 4, False)
    a = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    st = core.last_call_stats()
    assert st["path"] == "emit_bounded", st
    compare(a, want)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_SAMPLE_FUSED", "0")
    b = core.get_metrics_ials(t, 0, U, mask, cutoff, 0, False)
    stb = core.last_call_stats()
    assert (stb["path"], stb["hard_rows"], stb["tiles_scored"]) == (st["path"], st["hard_rows"], st["tiles_scored"])
    compare(b, want)


@pytest.mark.parametrize("rows", [1, 5, 64, 65, 130])
def test_bounded_path_on_a_handful_of_users(rows):
    """User blocks smaller than, equal to and just above one 64-user tile (the second-chance
    buffer, the work list and the sorts are all sized by the block)."""
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer)

    rng2 = np.random.default_rng(rows)
    U, I, K, cutoff = 200, 8300, 32, 10
    mc = IALSModelConfigBuilder().set_K(K).build()
    sc = IALSSolverConfigBuilder().build()
    t = IALSTrainer(mc, sps.csr_matrix((U, I), dtype=np.float32))
    scale = ((1.0 + np.arange(I)) ** -0.5).astype(np.float32)
    t.user = rng2.standard_normal((U, K)).astype(np.float32)
    t.item = rng2.standard_normal((I, K)).astype(np.float32) * scale[:, None]
    mask = sps.random(U, I, density=0.02, format="csr", random_state=rng2, dtype=np.float32)
    mask.data[:] = 1.0
    gt = sps.random(U, I, density=0.004, format="csr", random_state=rng2, dtype=np.float64)
    gt.data[:] = 1.0
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    b = 17
    m = core.get_metrics_ials(t, b, b + rows, sps.csr_matrix(mask[b:b + rows]), cutoff, b, False)
    assert core.last_call_stats()["path"] == "emit_bounded"
    scores = t.user_scores(b, b + rows, sc)
    scores[mask[b:b + rows].nonzero()] = -np.inf
    compare(m, ocore.get_metrics_f32(scores, cutoff, b, 4, False))
    # no mask at all
    m2 = core.get_metrics_ials(t, b, b + rows, None, cutoff, b, True)
    compare(m2, ocore.get_metrics_f32(t.user_scores(b, b + rows, sc), cutoff, b, 4, True))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_cutoff_above_2048_with_ties_masks_and_lists(dtype):
    """evaluator.cpp:263-268 accepts any 0 < cutoff <= n_items.  Above 2048 the selected lists
    are kept in global scratch (rank_rows_kernel<..., BIG>): counters and histogram bit-exact
    against the oracle with heavy ties (integer scores), -inf masks, a global candidate list and
    per-user lists, offsets, both recall conventions."""
    rns = np.random.RandomState(7)
    U, I = 40, 7000
    scores = rns.randint(0, 50, size=(U, I)).astype(dtype)  # ~140 ties per value
    scores[rns.rand(U, I) < 0.05] = -np.inf
    scores[3, :] = -np.inf
    scores[4, 100:] = -np.inf  # fewer rankable items than the cutoff
    gt = sps.csr_matrix((rns.rand(U + 5, I) >= 0.98).astype(np.float64))
    f = "get_metrics_f64" if dtype == "float64" else "get_metrics_f32"
    glob = [sorted(rns.choice(I, size=5500, replace=False).tolist())]
    per = [sorted(rns.choice(I, size=int(n), replace=False).tolist())
           for n in rns.randint(0, 6000, size=U + 5)]
    for lists in ([], glob, per):
        core, ocore = EvaluatorCore(gt, lists), O.EvaluatorCore(gt, lists)
        for cutoff in (2049, 4096, 5000, I):
            for rwc in (False, True):
                compare(getattr(core, f)(scores, cutoff, 5, 2, rwc),
                        getattr(ocore, f)(scores, cutoff, 5, 2, rwc))


def test_cutoff_5000_at_ml20m_width():
    """The shape named by the round-2 review: cutoff = 5,000 at I = 26,744 (float keys in
    registers do not apply: the BIG kernel keeps them in LDS), counters bit-exact."""
    rns = np.random.RandomState(3)
    U, I = 96, 26_744
    scores = rns.randn(U, I).astype(np.float32)
    scores[:, ::11] = np.round(scores[:, ::11], 1)  # ties
    scores[rns.rand(U, I) < 0.01] = -np.inf
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.995).astype(np.float64))
    core, ocore = EvaluatorCore(gt, []), O.EvaluatorCore(gt, [])
    for cutoff in (5000, 20_000, I):
        compare(core.get_metrics_f32(scores, cutoff, 0, 1), ocore.get_metrics_f32(scores, cutoff, 0, 4))
