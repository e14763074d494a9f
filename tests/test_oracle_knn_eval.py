"""Pins the kNN and evaluator oracles against the reference tests' closed forms and the
committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

Restates /root/reference/tests/recommenders/test_knn.py:33-165 and
/root/reference/tests/evaluation/test_evaluator.py:19-152, 358-398,
tests/evaluation/test_restricted_evaluator.py:25-108.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------ kNN
@pytest.fixture(scope="module")
def knn_golden():
    return np.load(os.path.join(G, "knn_dense.npz"))


@pytest.mark.parametrize("name", ["small", "many", "dense"])
def test_knn_dense_formulas(knn_golden, name):
    X = sps.csr_matrix(knn_golden[f"X_{name}"])
    Xt = sps.csr_matrix(X.T)
    N = X.shape[1]
    for key, comp in (
        ("cos_raw", O.KNNComputer("cosine", Xt, normalize=False, n_threads=5)),
        ("cos_norm", O.KNNComputer("cosine", Xt, normalize=True, n_threads=5)),
        ("jaccard", O.KNNComputer("jaccard", Xt)),
    ):
        W = O.remove_diagonal(comp.compute_similarity(Xt, N)).toarray()
        np.testing.assert_allclose(W, knn_golden[f"{key}_{name}"], rtol=1e-7, atol=1e-12)


def test_knn_tie_known_answer(knn_golden):
    # test_knn.py:144-165
    Xt = sps.csr_matrix(knn_golden["tie_X"].T)
    c = O.KNNComputer("cosine", Xt, normalize=False, n_threads=1, max_chunk_size=128)
    np.testing.assert_array_equal(c.compute_similarity(Xt, 2).toarray(), knn_golden["tie_top2"])


@pytest.mark.parametrize("alpha,shrink", [(0.5, 0.0), (0.7, 1.0), (0.01, 3.0)])
def test_knn_asymmetric(knn_golden, alpha, shrink):
    # test_knn.py:73-95
    X = sps.csr_matrix(knn_golden["X_dense"])
    Xt = sps.csr_matrix(X.T)
    W = O.remove_diagonal(O.KNNComputer("asymmetric", Xt, shrinkage=shrink, alpha=alpha)
                          .compute_similarity(Xt, X.shape[1])).toarray()
    m = X.T.toarray()
    norm = (m ** 2).sum(axis=1)
    sim = m @ m.T / (np.power(norm, alpha)[:, None] * np.power(norm, 1 - alpha)[None, :] + 1e-6 + shrink)
    np.fill_diagonal(sim, 0)
    np.testing.assert_allclose(W, sim)


def test_knn_tversky(knn_golden):
    # test_knn.py:98-132
    X = sps.csr_matrix(knn_golden["X_many"])
    Xt = sps.csr_matrix(X.T)
    alpha, beta, shrink = 0.5, 0.5, 0.0
    W = O.KNNComputer("tversky", Xt, shrinkage=shrink, alpha=alpha, beta=beta).compute_similarity(
        Xt, X.shape[1]).toarray()
    Xc = X.tocsc()
    rns = np.random.RandomState(0)
    for i, j in zip(rns.randint(0, W.shape[0], 100), rns.randint(0, W.shape[0], 100)):
        if i == j:
            continue
        Ui, Uj = set(Xc[:, i].nonzero()[0]), set(Xc[:, j].nonzero()[0])
        inter = len(Ui & Uj)
        target = inter / (inter + alpha * len(Ui - Uj) + beta * len(Uj - Ui) + shrink + 1e-6)
        assert W[i, j] == pytest.approx(target)


def test_knn_topk_count_and_validation(knn_golden):
    X = sps.csr_matrix(knn_golden["X_many"])
    Xt = sps.csr_matrix(X.T)
    W = O.KNNComputer("cosine", Xt, n_threads=5).compute_similarity(Xt, 30)
    assert np.all(np.diff(W.indptr) <= 30)  # test_knn.py:135-141
    with pytest.raises(ValueError):
        O.KNNComputer("cosine", Xt, shrinkage=-1.0)
    with pytest.raises(ValueError):
        O.KNNComputer("cosine", Xt, n_threads=0)
    with pytest.raises(ValueError):
        O.KNNComputer("asymmetric", Xt, alpha=1.5)
    with pytest.raises(ValueError, match="illegal # of feature"):
        O.KNNComputer("cosine", Xt).compute_similarity(sps.csr_matrix(X), 3)
    a = O.KNNComputer("cosine", Xt, normalize=True, n_threads=1).compute_similarity(Xt, 9)
    b = O.KNNComputer("cosine", Xt, normalize=True, n_threads=7, max_chunk_size=3).compute_similarity(Xt, 9)
    assert (a != b).nnz == 0  # thread / chunk invariance


def test_p3alpha_matches_dense_random_walk(knn_golden):
    # tests/recommenders/test_knn.py:168-199 (P3alpha: W = P_iu^T-normalised two-step walk)
    X = sps.csr_matrix(knn_golden["X_dense"])
    alpha = 2.0
    Xt = sps.csr_matrix(X.T)
    W = O.KNNComputer("p3alpha", Xt, alpha=alpha).compute_W(Xt, X.shape[1]).toarray()
    Xd = X.toarray() ** alpha
    P_ui = Xd / Xd.sum(axis=1, keepdims=True)          # user -> item
    P_iu = (Xd / Xd.sum(axis=0, keepdims=True)).T      # item -> user
    np.testing.assert_allclose(W, (P_iu @ P_ui), rtol=1e-9, atol=1e-14)


# ------------------------------------------------------------------ evaluator
@pytest.fixture(scope="module")
def ev_golden():
    return np.load(os.path.join(G, "evaluator_rs42.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_eval_vs_sklearn_golden(ev_golden, tag):
    scores, gt = ev_golden[f"full_{tag}_scores"], ev_golden[f"full_{tag}_gt"]
    core = O.EvaluatorCore(sps.csr_matrix(gt), [])
    f = core.get_metrics_f64 if scores.dtype == np.float64 else core.get_metrics_f32
    d = f(scores, scores.shape[1], 0, 4).as_dict()
    exp_map, exp_ndcg = ev_golden[f"full_{tag}_expected"]
    assert d["map"] == pytest.approx(exp_map, abs=1e-8)
    assert d["ndcg"] == pytest.approx(exp_ndcg, abs=1e-8)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_eval_with_cutoff_golden(ev_golden, tag):
    scores, gt, C = ev_golden[f"cut_{tag}_scores"], ev_golden[f"cut_{tag}_gt"], int(ev_golden[f"cut_{tag}_C"])
    core = O.EvaluatorCore(sps.csr_matrix(gt), [])
    whole = core.get_metrics_f64(scores, C, 0, 2, True)
    finer = O.Metrics(gt.shape[1])
    for u in range(gt.shape[0]):  # mb_size = 1
        finer.merge(core.get_metrics_f64(scores[u:u + 1], C, u, 2, True))
    d, df = whole.as_dict(), finer.as_dict()
    for k in d:
        assert df[k] == pytest.approx(d[k])
    ndcg, mapv, prec, rec, entropy, gini = ev_golden[f"cut_{tag}_expected"]
    assert d["ndcg"] == pytest.approx(ndcg)
    assert d["map"] == pytest.approx(mapv, abs=1e-8)
    assert d["precision"] == pytest.approx(prec, abs=1e-8)
    assert d["recall"] == pytest.approx(rec, abs=1e-8)
    assert d["entropy"] == pytest.approx(entropy)
    assert d["gini_index"] == pytest.approx(gini)


def test_eval_neg_inf_and_validation():
    # test_evaluator.py:358-368
    core = O.EvaluatorCore(sps.csr_matrix(np.asarray([[1.0, 1.0, 0.0]])), [])
    d = core.get_metrics_f64(np.asarray([[1.0, -np.inf, -np.inf]]), 3, 0, 1).as_dict()
    assert d["precision"] == 1.0 and d["recall"] == 0.5
    gt = sps.csr_matrix(np.eye(4))
    for bad in ([[0], [1]], [[0, 0]], [[4]]):
        with pytest.raises(ValueError):
            O.EvaluatorCore(gt, bad)
    c = O.EvaluatorCore(gt, [])
    s = np.zeros((4, 4), dtype=np.float32)
    for args in ((0, 0, 1), (5, 0, 1), (2, 4, 1), (2, 1, 1), (2, 0, 0)):
        with pytest.raises(ValueError):
            c.get_metrics_f32(s, *args)


def test_eval_restricted_lists():
    # test_restricted_evaluator.py:25-108: metrics on a restricted list equal metrics on the
    # score matrix with everything else at -inf and the ground truth intersected
    rns = np.random.RandomState(5)
    U, I = 30, 60
    scores = rns.randn(U, I)
    gt = (rns.rand(U, I) >= 0.8).astype(np.float64)
    rec = sorted(rns.choice(I, size=25, replace=False).tolist())
    a = O.EvaluatorCore(sps.csr_matrix(gt), [rec]).get_metrics_f64(scores, 5, 0, 2)
    masked = np.full_like(scores, -np.inf)
    masked[:, rec] = scores[:, rec]
    gt2 = np.zeros_like(gt)
    gt2[:, rec] = gt[:, rec]
    b = O.EvaluatorCore(sps.csr_matrix(gt2), []).get_metrics_f64(masked, 5, 0, 2)
    da, db = a.as_dict(), b.as_dict()
    for k in da:
        assert da[k] == pytest.approx(db[k]), k
