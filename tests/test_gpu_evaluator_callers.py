"""GPU tests of the ``Evaluator`` / ``EvaluatorWithColdUser`` caller contract
(evaluation/evaluator.py:229-398, 444-657): the score-matrix and score-chunk entry points and
the cold-user evaluator, restating the reference's own tests
(tests/evaluation/test_evaluator.py:182-229 cold users vs hot users, :246-276 score matrix,
:279-345 feature-only items, :348-368 shape validation / -inf, :371-441 chunks and their errors)
and checking the device-side masking call against the host loop of the reference
(copy -> ``scores[mask.nonzero()] = -inf`` -> ``get_metrics`` per cutoff) run on the oracle.
Bar: counters and histogram bit-exact, fp64 sums 1e-12.
"""
import warnings

import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from irspack_amd.evaluation import Evaluator, EvaluatorWithColdUser
from irspack_amd.evaluation._core_evaluator import EvaluatorCore, MaskRows
from irspack_amd.recommenders.base import BaseRecommender

pytestmark = pytest.mark.gpu


def oracle_host_loop(gt, scores, mask, cutoffs, mb_size, recommendable=(), rwc=False):
    """The reference's chunk loop (evaluator.py:371-398) on the CPU oracle: one merged raw
    accumulator per cutoff."""
    core = O.EvaluatorCore(sps.csr_matrix(gt, dtype=np.float64), list(recommendable))
    out = []
    for c in cutoffs:
        raw, cnt = np.zeros(7), np.zeros(gt.shape[1], dtype=np.int64)
        for b in range(0, gt.shape[0], mb_size):
            blk = scores[b:b + mb_size].copy(order="C")
            if mask is not None:
                blk[mask[b:b + mb_size].nonzero()] = -np.inf
            f = core.get_metrics_f64 if blk.dtype == np.float64 else core.get_metrics_f32
            m = f(blk, c, b, 1, rwc)
            raw += m.raw()
            cnt += m.item_cnt()
        out.append((raw, cnt))
    return out


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("with_zeros", [False, True])
def test_masked_block_call_equals_host_masking(dtype, with_zeros):
    rns = np.random.RandomState(3)
    U, I = 300, 517
    scores = rns.randn(U, I).astype(dtype)
    scores[rns.rand(U, I) > 0.97] = 1.5  # ties
    keep = scores.copy()
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.9).astype(np.float64))
    dense_mask = (rns.rand(U, I) >= 0.6).astype(np.float64)
    dense_mask[47] = 0  # a row without mask entries inside the block
    mask = sps.csr_matrix(dense_mask)
    if with_zeros:  # stored zeros do not mask (``mask.nonzero()``)
        mask.data[rns.rand(mask.nnz) > 0.5] = 0.0
    cutoffs = [1, 10, 64, 65, I]
    core = EvaluatorCore(gt, [])
    got = core.get_metrics_masked(scores[40:260], MaskRows(mask, I), 40, cutoffs, 40, 2, True)
    ocore = O.EvaluatorCore(gt, [])
    blk = scores[40:260].copy()
    blk[mask[40:260].nonzero()] = -np.inf
    for c, m in zip(cutoffs, got):
        f = ocore.get_metrics_f64 if dtype == "float64" else ocore.get_metrics_f32
        om = f(blk, c, 40, 1, True)
        np.testing.assert_array_equal(m.item_cnt, om.item_cnt())
        raw = om.raw()
        assert (m.valid_user, m.total_user) == (int(raw[0]), int(raw[1]))
        np.testing.assert_allclose([m.hit, m.recall, m.ndcg, m.precision, m.map], raw[2:], rtol=1e-12)
    np.testing.assert_array_equal(scores, keep)  # the caller's block is only read
    # no mask rows in the block / no mask at all: the plain call
    for mk in (None, MaskRows(sps.csr_matrix((U, I)), I)):
        m = core.get_metrics_masked(scores[:50], mk, 0, [5], 0, 1)[0]
        p = (core.get_metrics_f64 if dtype == "float64" else core.get_metrics_f32)(scores[:50], 5, 0, 1)
        np.testing.assert_array_equal(m.item_cnt, p.item_cnt)
        assert m.ndcg == p.ndcg and m.valid_user == p.valid_user
    with pytest.raises(ValueError):
        core.get_metrics_masked(scores[:50], None, 0, [0], 0, 1)  # cutoff == 0
    with pytest.raises(ValueError):
        core.get_metrics_masked(scores[:50], None, 0, [5], U - 10, 1)  # offset + rows > n_users


def test_score_from_score_matrix():
    # test_evaluator.py:246-262
    scores = np.array([[0.1, 0.9], [0.8, 0.2]], dtype=np.float32)
    original_scores = scores.copy()
    ground_truth = sps.csr_matrix([[0, 1], [1, 0]])
    mask = sps.csr_matrix([[1, 0], [0, 0]])
    evaluator = Evaluator(ground_truth, cutoff=1, masked_interactions=mask, mb_size=1)
    assert evaluator.get_score_from_score_matrix(scores)["recall"] == 1.0
    assert evaluator.get_scores_from_score_matrix(scores, [1])["recall@1"] == 1.0
    np.testing.assert_array_equal(scores, original_scores)
    with pytest.raises(ValueError, match="shape"):
        evaluator.get_score_from_score_matrix(scores[:, :1])
    with pytest.raises(ValueError, match="dtype"):
        evaluator.get_score_from_score_matrix(scores.astype(np.float16))


def test_score_from_score_matrix_cold_user_masks_input():
    # test_evaluator.py:265-276: the seen item has the highest raw score and must be excluded
    evaluator = EvaluatorWithColdUser(sps.csr_matrix([[1, 0]]), sps.csr_matrix([[0, 1]]), cutoff=1)
    assert evaluator.get_score_from_score_matrix(np.array([[1.0, 0.0]], dtype=np.float64))["recall"] == 1.0
    assert evaluator.get_score_from_score_chunks(iter([np.array([[1.0, 0.0]], dtype=np.float64)]))["recall"] == 1.0


def test_negative_infinity_scores_are_not_recommendations():
    # test_evaluator.py:358-368
    evaluator = Evaluator(sps.csr_matrix([[1, 0, 0]]), cutoff=3)
    score = evaluator.get_score_from_score_matrix(np.array([[1.0, -np.inf, -np.inf]], dtype=np.float64))
    assert score["recall"] == 1.0
    assert score["precision"] == 1.0
    assert score["appeared_item"] == 1.0
    assert score["catalog_coverage"] == pytest.approx(1 / 3)


def test_score_from_score_chunks_matches_matrix():
    # test_evaluator.py:371-399
    rns = np.random.RandomState(0)
    U, I = 11, 7
    scores = rns.randn(U, I).astype(np.float64)
    original_scores = scores.copy()
    X_gt = sps.csr_matrix((rns.rand(U, I) >= 0.5).astype(np.float64))
    mask = sps.csr_matrix((rns.rand(U, I) >= 0.5).astype(np.float64))
    evaluator = Evaluator(X_gt, cutoff=3, masked_interactions=mask, mb_size=2)
    expected = evaluator.get_scores_from_score_matrix(scores, [1, 3])
    split_points = [0, 1, 1, 3, 3, 6, 10, 11]
    chunks = [scores[split_points[i]:split_points[i + 1]] for i in range(len(split_points) - 1)
              if split_points[i] != split_points[i + 1]]
    got = evaluator.get_scores_from_score_chunks(iter(chunks), [1, 3])
    for key, value in expected.items():
        assert got[key] == pytest.approx(value, abs=1e-12), key
    np.testing.assert_array_equal(scores, original_scores)
    seen = [c.copy() for c in chunks]
    evaluator.get_scores_from_score_chunks(iter(chunks), [3])
    for original, mutated in zip(seen, chunks):
        np.testing.assert_array_equal(original, mutated)
    # and both equal the reference's host loop on the oracle
    for (raw, cnt), c in zip(oracle_host_loop(X_gt, scores, mask, [1, 3], 2), [1, 3]):
        denom = max(raw[0], 1)
        assert got[f"ndcg@{c}"] == pytest.approx(raw[4] / denom, rel=1e-12)
        assert got[f"map@{c}"] == pytest.approx(raw[6] / denom, rel=1e-12)
        assert got[f"appeared_item@{c}"] == float((cnt > 0).sum())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_score_from_score_chunks_errors(dtype):
    # test_evaluator.py:414-441
    U, I = 3, 4
    evaluator = Evaluator(sps.csr_matrix(np.eye(U, I, dtype=np.float64)), cutoff=2)
    with pytest.raises(ValueError, match="n_items"):
        evaluator.get_score_from_score_chunks(iter([np.zeros((U, I - 1), dtype=dtype)]))
    with pytest.raises(ValueError, match="dtype"):
        evaluator.get_score_from_score_chunks(iter([np.zeros((U, I), dtype=np.float16)]))
    with pytest.raises(ValueError, match="did not cover"):
        evaluator.get_score_from_score_chunks(iter([np.zeros((U - 1, I), dtype=dtype)]))
    with pytest.raises(ValueError, match="more rows"):
        evaluator.get_score_from_score_chunks(iter([np.zeros((U + 1, I), dtype=dtype)]))
    with pytest.raises(ValueError, match="2-D ndarray"):
        evaluator.get_score_from_score_chunks(iter([np.zeros(I, dtype=dtype)]))
    # empty chunks are skipped, not an error (evaluator.py:384-385)
    z = np.zeros((0, I), dtype=dtype)
    evaluator.get_score_from_score_chunks(iter([z, np.zeros((U, I), dtype=dtype), z]))


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_score_matrix_at_size_with_restrictions(dtype):
    """Larger than one device call (rows per call forced down), per-user recommendable lists,
    several cutoffs, recall_with_cutoff: equal to the reference's 128-row host loop."""
    rns = np.random.RandomState(11)
    U, I = 1500, 700
    scores = np.round(rns.randn(U, I), 1).astype(dtype)  # many ties
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.95).astype(np.float64))
    mask = sps.csr_matrix((rns.rand(U, I) >= 0.8).astype(np.float64))
    rec = [sorted(rns.choice(I, size=rns.randint(0, 60), replace=False).tolist()) for _ in range(U)]
    cutoffs = [3, 20, 100]
    ev = Evaluator(gt, cutoff=20, per_user_recommendable_items=rec, masked_interactions=mask,
                   recall_with_cutoff=True, mb_size=128)
    ev.score_matrix_rows_per_call = 333
    got = ev.get_scores_from_score_matrix(scores, cutoffs)
    want = oracle_host_loop(gt, scores, mask, cutoffs, 128, rec, True)
    for (raw, cnt), c in zip(want, cutoffs):
        denom = max(raw[0], 1)
        for name, k in (("hit", 2), ("recall", 3), ("ndcg", 4), ("precision", 5), ("map", 6)):
            assert got[f"{name}@{c}"] == pytest.approx(raw[k] / denom, rel=1e-12), (name, c)
        assert got[f"appeared_item@{c}"] == float((cnt > 0).sum())
        assert got[f"catalog_coverage@{c}"] == pytest.approx(
            (cnt > 0).sum() / len({i for l in rec for i in l}))


class MockRecommender(BaseRecommender):
    def __init__(self, X, scores):
        super().__init__(X)
        self.scores = scores

    def get_score(self, user_indices):
        return self.scores[user_indices]

    def _learn(self):
        pass


def test_model_blocks_are_not_written_and_match_host_loop():
    # evaluator.py:400-441 with a model that only has get_score (the NotImplementedError branch)
    rns = np.random.RandomState(5)
    U, I, off = 230, 90, 40
    X = sps.csr_matrix((rns.rand(U, I) >= 0.7).astype(np.float64))
    scores = rns.randn(U, I)
    keep = scores.copy()
    gt = sps.csr_matrix((rns.rand(U - off, I) >= 0.8).astype(np.float64))
    rec = MockRecommender(X, scores)
    ev = Evaluator(gt, offset=off, cutoff=7, mb_size=32, n_threads=2)
    got = ev.get_scores(rec, [7, 30])
    np.testing.assert_array_equal(scores, keep)
    want = oracle_host_loop(gt, scores[off:], X[off:], [7, 30], 32)
    for (raw, cnt), c in zip(want, [7, 30]):
        assert got[f"ndcg@{c}"] == pytest.approx(raw[4] / max(raw[0], 1), rel=1e-12)
        assert got[f"appeared_item@{c}"] == float((cnt > 0).sum())
    # an explicit mask is indexed relative to the evaluator's own rows (evaluator.py:427-430)
    m2 = sps.csr_matrix((rns.rand(U - off, I) >= 0.5).astype(np.float64))
    ev2 = Evaluator(gt, offset=off, cutoff=7, mb_size=50, masked_interactions=m2)
    want2 = oracle_host_loop(gt, scores[off:], m2, [7], 50)[0]
    assert ev2.get_score(rec)["map"] == pytest.approx(want2[0][6] / max(want2[0][0], 1), rel=1e-12)
    # shape checks (test_evaluator.py:232-243)
    with pytest.raises(ValueError):
        Evaluator(sps.csr_matrix((U, I)), cutoff=3).get_score(MockRecommender(sps.csr_matrix((U - 1, I)), scores[1:]))
    with pytest.raises(ValueError):
        Evaluator(sps.csr_matrix((U, I)), cutoff=3).get_score(MockRecommender(sps.csr_matrix((U, I - 1)), scores[:, 1:]))


class SimilarityMock(BaseRecommender):
    """score = X @ W (the shape of the reference's P3alpha in test_evaluator.py:182-229)."""

    def __init__(self, X, W):
        super().__init__(X)
        self.W = W

    def _learn(self):
        pass

    def get_score(self, user_indices):
        return np.asarray(self.X_train_all[user_indices].dot(self.W))

    def get_score_block(self, begin, end):
        return np.asarray(self.X_train_all[begin:end].dot(self.W))

    def get_score_cold_user(self, X):
        r = X.dot(self.W)
        return np.asarray(r.toarray() if sps.issparse(r) else r)


def test_cold_user_evaluator_equals_hot_evaluator():
    # test_evaluator.py:182-229: users appended to the training matrix, evaluated hot with an
    # offset, must score like the same users evaluated cold from their learn-half
    rns = np.random.RandomState(0)
    U, U_val, I = 60, 25, 40
    X_train = sps.csr_matrix((rns.rand(U, I) >= 0.7).astype(np.float64))
    X_val_learn = sps.csr_matrix((rns.rand(U_val, I) >= 0.7).astype(np.float64))
    X_val_target = sps.csr_matrix(((rns.rand(U_val, I) >= 0.7) & (X_val_learn.toarray() == 0)).astype(np.float64))
    X_all = sps.vstack([X_train, X_val_learn]).tocsr()
    W = rns.rand(I, I)
    rec = SimilarityMock(X_all, W)
    hot = Evaluator(X_val_target, offset=U, cutoff=I // 2, n_threads=2).get_score(rec)
    cold_ev = EvaluatorWithColdUser(X_val_learn, X_val_target, cutoff=I // 2, mb_size=5)
    cold = cold_ev.get_score(rec)
    for key in hot:
        assert hot[key] == pytest.approx(cold[key], abs=1e-8), key
    idx = np.arange(U_val)
    rns.shuffle(idx)
    shuffled = EvaluatorWithColdUser(X_val_learn[idx], X_val_target[idx], cutoff=I // 2).get_score(rec)
    for key in cold:
        assert shuffled[key] == pytest.approx(cold[key]), key
    # column-major scores: warned about and converted (evaluator.py:641-646)
    class FortranMock(SimilarityMock):
        def get_score_cold_user(self, X):
            return np.asfortranarray(super().get_score_cold_user(X))
    with pytest.warns(UserWarning):
        f = EvaluatorWithColdUser(X_val_learn, X_val_target, cutoff=I // 2, mb_size=7).get_score(FortranMock(X_all, W))
    for key in cold:
        assert f[key] == pytest.approx(cold[key]), key
    with pytest.raises(ValueError):  # rows of the two matrices differ
        EvaluatorWithColdUser(X_val_learn[:-1], X_val_target)
    with pytest.raises(ValueError):  # the model was trained on another item set
        EvaluatorWithColdUser(X_val_learn[:, :-1], X_val_target[:, :-1]).get_score(rec)


def test_cold_user_evaluator_with_cold_item_features():
    # test_evaluator.py:279-342
    class FeatureItemMock(BaseRecommender):
        def __init__(self):
            super().__init__(sps.csr_matrix((2, 2)))
            self.prepare_count = 0

        def _learn(self):
            pass

        def get_score(self, user_indices):
            return np.repeat(np.array([[0.2, 0.1]], dtype=np.float32), len(user_indices), axis=0)

        def get_score_cold_user(self, X):
            return np.repeat(np.array([[0.2, 0.1]], dtype=np.float32), X.shape[0], axis=0)

        def _create_cold_user_with_item_features_scorer(self, item_features):
            self.prepare_count += 1
            np.testing.assert_array_equal(item_features, np.array([[1.0], [2.0]], dtype=np.float32))
            return lambda X: np.repeat(np.array([[0.2, 0.1, 0.9, 0.8]], dtype=np.float32), X.shape[0], axis=0)

    class PlainMock(FeatureItemMock):
        _create_cold_user_with_item_features_scorer = BaseRecommender._create_cold_user_with_item_features_scorer

    input_interaction = sps.csr_matrix([[1, 0], [0, 1]], dtype=np.float32)
    ground_truth = sps.csr_matrix([[0, 0, 1, 0], [0, 0, 1, 0]])
    cold_item_features = np.array([[1.0], [2.0]], dtype=np.float32)
    evaluator = EvaluatorWithColdUser(input_interaction, ground_truth, cold_item_features=cold_item_features,
                                      cutoff=1, mb_size=1, n_threads=1)
    rec = FeatureItemMock()
    score = evaluator.get_score(rec)
    assert score["recall"] == 1.0
    assert score["catalog_coverage"] == 0.25
    assert rec.prepare_count == 1
    assert evaluator.get_scores(rec, [1])["catalog_coverage@1"] == 0.25
    # a model without feature scoring leaves the feature-only items unrankable; -inf cold-item
    # scores are not counted as recommendations
    fallback_score = evaluator.get_score(PlainMock())
    assert fallback_score["recall"] == 0.0
    assert fallback_score["appeared_item"] == 2.0
    # a mask with the training columns only is widened (evaluator.py:543-553)
    ev2 = EvaluatorWithColdUser(input_interaction, ground_truth, cold_item_features=cold_item_features,
                                masked_interactions=sps.csr_matrix([[0, 1], [1, 0]], dtype=np.float32), cutoff=1)
    assert ev2.masked_interactions.shape == (2, 4)
    assert ev2.get_score(rec)["recall"] == 1.0


def test_cold_user_evaluator_cold_item_shape_validation():
    # test_evaluator.py:345-355
    with pytest.raises(ValueError, match="ground_truth"):
        EvaluatorWithColdUser(sps.csr_matrix((2, 3)), sps.csr_matrix((2, 4)),
                              cold_item_features=np.ones((2, 1), dtype=np.float32))


def test_ials_recommender_cold_paths_through_the_evaluator():
    """``IALSRecommender`` through ``EvaluatorWithColdUser`` (fold-in on the device, the block
    masked and ranked on the device) equals scoring the same fold-in by hand."""
    from irspack_amd.recommenders.ials import IALSRecommender

    rns = np.random.RandomState(2)
    U, I = 300, 120
    X = sps.csr_matrix((rns.rand(U, I) >= 0.85).astype(np.float64))
    rec = IALSRecommender(X, n_components=16, alpha0=0.1, reg=1e-2, train_epochs=3,
                          solver_type="CHOLESKY").learn()
    Xc = sps.csr_matrix((rns.rand(50, I) >= 0.85).astype(np.float64))
    gt = sps.csr_matrix(((rns.rand(50, I) >= 0.8) & (Xc.toarray() == 0)).astype(np.float64))
    ev = EvaluatorWithColdUser(Xc, gt, cutoff=10, mb_size=16)
    got = ev.get_scores(rec, [5, 10])
    scores = rec.get_score_cold_user(Xc)
    want = oracle_host_loop(gt, scores, Xc, [5, 10], 16)
    for (raw, cnt), c in zip(want, [5, 10]):
        assert got[f"ndcg@{c}"] == pytest.approx(raw[4] / max(raw[0], 1), rel=1e-9)
        assert got[f"appeared_item@{c}"] == float((cnt > 0).sum())


# ---------------------------------------------------------------- similarity models on the device (round 6)
def _knn_problem(seed, U=1500, I=700, density=0.03, weighted=False):
    rns = np.random.RandomState(seed)
    X = sps.random(U, I, density=density, format="csr", random_state=rns, dtype=np.float64)
    X.data = rns.uniform(0.5, 3.0, X.nnz) if weighted else np.ones_like(X.data)
    lil = X.tolil()
    for r in (0, 7, U - 1):  # users without a profile: every score is 0, ties by item index
        lil.rows[r], lil.data[r] = [], []
    X = sps.csr_matrix(lil)
    X.sort_indices()
    gt = sps.csr_matrix((rns.rand(U, I) >= 0.985).astype(np.float64))
    return X, gt


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("kind", ["cosine", "p3alpha", "user_cosine"])
def test_similarity_models_are_scored_on_the_device(kind, weighted):
    """``Evaluator.get_scores(model)`` for a similarity model (score = X_train[u] @ W, base.py:406-429):
    the fused path computes the score block on the device - per column the updates `x * w` in the order of
    the user's stored profile, product and sum rounded separately: scipy's `csr_matmat` order, so the block is
    `X[b:e].dot(W)` bit for bit - masks and ranks it there.  Against (a) the block loop through the host
    (`fused=False`: the model's own `get_score_block`, uploaded and ranked per 128 users) and (b) the oracle's
    evaluator fed the model's `get_score_remove_seen_block`: counters and item histogram equal, float64
    sums to 1e-12; several cutoffs, an offset window, an explicit mask, users without a profile, stored zeros
    in the training matrix (not masked: `mask.nonzero()`)."""
    from irspack_amd.recommenders import knn as KN
    from irspack_amd.recommenders.user_knn import CosineUserKNNRecommender

    X, gt = _knn_problem(3 if weighted else 4, weighted=weighted)
    if weighted:
        X.data[::11] = 0.0  # stored zeros: part of the profile's pattern, not of the mask
    if kind == "cosine":
        model = KN.CosineKNNRecommender(X, shrinkage=1.0, normalize=True, top_k=20, feature_weighting="TF_IDF").learn()
    elif kind == "p3alpha":
        model = KN.P3alphaRecommender(X, alpha=0.8, top_k=15).learn()
    else:
        model = CosineUserKNNRecommender(X, shrinkage=0.5, top_k=25).learn()
    cutoffs = [1, 5, 20, 65]
    keys = ("hit", "recall", "ndcg", "map", "precision")
    for offset, n in ((0, X.shape[0]), (300, 700)):
        g = gt[offset:offset + n]
        for masked in (None, sps.csr_matrix((np.random.RandomState(9).rand(n, X.shape[1]) > 0.9).astype(np.float64))):
            fused = Evaluator(g, offset=offset, cutoff=10, masked_interactions=masked)
            plain = Evaluator(g, offset=offset, cutoff=10, masked_interactions=masked, fused=False)
            got, want = fused.get_scores(model, cutoffs), plain.get_scores(model, cutoffs)
            assert fused._similarity_weights(model) is not None and plain._similarity_weights(model) is None
            for c in cutoffs:
                for k in ("appeared_item", "catalog_coverage"):
                    assert got[f"{k}@{c}"] == want[f"{k}@{c}"], (k, c, offset)
                for k in keys + ("entropy", "gini_index"):
                    assert got[f"{k}@{c}"] == pytest.approx(want[f"{k}@{c}"], rel=1e-12, abs=1e-15), (k, c, offset)
    # and against the oracle's evaluator on the model's own host scores, all users, no explicit mask
    ocore = O.EvaluatorCore(gt, [])
    core = EvaluatorCore(gt, [])
    profiles, Wr = (model.U, model.X_train_all) if kind == "user_cosine" else (model.X_train_all, sps.csr_matrix(model.W))
    Wr.sort_indices()
    got = core.get_metrics_similarity(profiles, Wr, 0, X.shape[0], MaskRows(model.X_train_all, X.shape[1]), 0,
                                      cutoffs, 0, False)
    for c, m in zip(cutoffs, got):
        raw, cnt = np.zeros(7), np.zeros(X.shape[1], dtype=np.int64)
        for b in range(0, X.shape[0], 256):
            om = ocore.get_metrics_f64(np.ascontiguousarray(model.get_score_remove_seen_block(b, min(b + 256, X.shape[0]))),
                                       c, b, 1, False)
            raw += om.raw()
            cnt += om.item_cnt()
        np.testing.assert_array_equal(m.item_cnt, cnt)
        assert m.valid_user == int(raw[0]) and m.total_user == int(raw[1])
        for got_v, want_v in ((m.hit, raw[2]), (m.recall, raw[3]), (m.ndcg, raw[4]), (m.precision, raw[5]), (m.map, raw[6])):
            assert got_v == pytest.approx(want_v, rel=1e-12, abs=1e-15)


def test_similarity_path_in_several_score_blocks(monkeypatch):
    """The call walks the users in blocks of 4 GB of float64 scores (20,074 users at 26,744 items); here 37
    users per block: counters and histogram of the one-block call, float64 sums merged block by block
    (Metrics::merge order) to 1e-12."""
    X, gt = _knn_problem(13, U=700, I=1200, density=0.02, weighted=True)
    rns = np.random.RandomState(6)
    W = sps.random(1200, 1200, density=0.02, format="csr", random_state=rns, dtype=np.float64)
    W.sort_indices()
    core = EvaluatorCore(gt, [])
    want = core.get_metrics_similarity(X, W, 3, 650, None, 0, [5, 20], 3)
    monkeypatch.setenv("IRSPACK_AMD_EVAL_SIM_BLOCK_ROWS", "37")
    got = core.get_metrics_similarity(X, W, 3, 650, None, 0, [5, 20], 3)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a.item_cnt, b.item_cnt)
        assert (a.valid_user, a.total_user) == (b.valid_user, b.total_user)
        np.testing.assert_allclose([a.hit, a.recall, a.ndcg, a.precision, a.map],
                                   [b.hit, b.recall, b.ndcg, b.precision, b.map], rtol=1e-12)


def test_similarity_path_with_unsorted_rows_of_w():
    """W whose rows are NOT in column order (a C-ABI caller may hand that over; `_lib.csr_arrays` would sort
    it, so the flag scipy caches is set by hand): the library sees it, skips the per-tile ranges and every
    tile walks the whole rows - the same float64 block, so the same metrics as the sorted W."""
    X, gt = _knn_problem(11, U=900, I=2500, density=0.02, weighted=True)
    rns = np.random.RandomState(5)
    W = sps.random(2500, 2500, density=0.01, format="csr", random_state=rns, dtype=np.float64)
    W.sort_indices()
    core = EvaluatorCore(gt, [])
    want = core.get_metrics_similarity(X, W, 0, 900, None, 0, [5, 20], 0)
    Wu = W.copy()
    for r in range(Wu.shape[0]):  # reverse every row's entries
        b, e = Wu.indptr[r], Wu.indptr[r + 1]
        Wu.indices[b:e] = Wu.indices[b:e][::-1].copy()
        Wu.data[b:e] = Wu.data[b:e][::-1].copy()
    Wu.has_sorted_indices = True  # (a lie, on purpose: csr_arrays then passes the arrays as they are)
    got = core.get_metrics_similarity(X, Wu, 0, 900, None, 0, [5, 20], 0)
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a.item_cnt, b.item_cnt)
        assert (a.valid_user, a.total_user, a.hit, a.ndcg, a.map) == (b.valid_user, b.total_user, b.hit, b.ndcg, b.map)


def test_similarity_path_argument_errors():
    X, gt = _knn_problem(5, U=200, I=90)
    core = EvaluatorCore(gt, [])
    W = sps.identity(90, format="csr", dtype=np.float64)
    with pytest.raises(ValueError, match="n_items"):
        core.get_metrics_similarity(X, sps.identity(91, format="csr", dtype=np.float64), 0, 200, None, 0, [5], 0)
    with pytest.raises(ValueError, match="n_items"):
        core.get_metrics_similarity(X, sps.csr_matrix((90, 89), dtype=np.float64), 0, 200, None, 0, [5], 0)
    with pytest.raises(ValueError, match="out of bounds"):
        core.get_metrics_similarity(X, W, 0, 201, None, 0, [5], 0)
    with pytest.raises(ValueError, match="cutoff"):
        core.get_metrics_similarity(X, W, 0, 200, None, 0, [0], 0)
    # identity W: the scores are the profile itself; unmasked, every user's own items lead the list
    m = core.get_metrics_similarity(X, W, 0, 200, None, 0, [3], 0)[0]
    assert m.total_user == 200
