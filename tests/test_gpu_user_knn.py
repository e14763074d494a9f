"""User-kNN callers (irspack_amd/recommenders/user_knn.py) on the GPU similarity kernels:
the checks of the reference's tests/recommenders/test_user_knn.py (dense numpy formulas with
the + 1e-6 denominators, top-k row bound, "fetched before fit") and bit-exact neighbour sets
against the CPU oracle, including a weighted (tf-idf / BM25) case where the computer is built
on the weighted matrix and queried with the unweighted one (user_knn.py:62-76).
"""
import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from irspack_amd.recommenders import AsymmetricCosineUserKNNRecommender, CosineUserKNNRecommender

pytestmark = pytest.mark.gpu

_rng = np.random.RandomState(0)
X_small = sps.csr_matrix(np.asarray(
    [[1, 1, 2, 3, 4], [0, 1, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]], dtype=float))
_d = _rng.rand(888, 512)
X_many = sps.csr_matrix((_d > 0.9).astype(float))
X_many.sort_indices()
X_many_dense = sps.csr_matrix(_rng.rand(133, 245))


@pytest.mark.parametrize("X,normalize", [(X_many, True), (X_small, False), (X_many_dense, True)])
def test_cosine(X, normalize):
    # test_user_knn.py:25-49
    rec = CosineUserKNNRecommender(X, shrinkage=0, n_threads=5, top_k=X.shape[0], normalize=normalize)
    with pytest.raises(RuntimeError):
        rec.U
    rec.learn()
    sim = rec.U.toarray()
    manual = X.toarray()
    norm = (manual ** 2).sum(axis=1) ** 0.5
    manual = manual.dot(manual.T)
    if normalize:
        manual /= norm[:, None] * norm[None, :] + 1e-6
    np.fill_diagonal(manual, 0)
    np.testing.assert_allclose(sim, manual)


@pytest.mark.parametrize("X,alpha,shrinkage", [(X_many, 0.5, 0.0), (X_small, 0.7, 1.0),
                                               (X_many_dense, 0.01, 3)])
def test_asymmetric_cosine(X, alpha, shrinkage):
    # test_user_knn.py:52-76
    rec = AsymmetricCosineUserKNNRecommender(X, shrinkage=shrinkage, alpha=alpha, n_threads=1,
                                             top_k=X.shape[0])
    rec.learn()
    sim = rec.U.toarray()
    manual = X.toarray()
    norm = (manual ** 2).sum(axis=1)
    manual_sim = manual.dot(manual.T)
    manual_sim /= np.power(norm, alpha)[:, None] * np.power(norm, 1 - alpha)[None, :] + 1e-6 + shrinkage
    np.fill_diagonal(manual_sim, 0)
    np.testing.assert_allclose(sim, manual_sim)


@pytest.mark.parametrize("X", [X_many, X_small])
def test_topk(X):
    # test_user_knn.py:79-85
    rec = AsymmetricCosineUserKNNRecommender(X, shrinkage=0, top_k=30, n_threads=5).learn()
    assert np.all((rec.U.toarray() > 0).sum(axis=1) <= 30)


@pytest.mark.parametrize("weighting", ["NONE", "TF_IDF", "BM_25"])
def test_learn_order_and_scores_vs_oracle(weighting):
    """_learn against the oracle restatement of the same steps; scores = U[u] @ X."""
    X = X_many[:300]
    rec = CosineUserKNNRecommender(X, shrinkage=0.5, top_k=17, feature_weighting=weighting).learn()
    Xw = {"NONE": lambda m: m, "TF_IDF": O.tf_idf_weight,
          "BM_25": lambda m: O.okapi_BM_25_weight(m, 1.2, 0.75)}[weighting](X)
    want = O.remove_diagonal(O.KNNComputer("cosine", Xw, 0.5, normalize=True).compute_similarity(X, 17))
    got = sps.csr_matrix(rec.U)
    got.sort_indices()
    want.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    np.testing.assert_allclose(got.data, want.data, rtol=1e-12, atol=0)
    users = np.asarray([0, 5, 299])
    np.testing.assert_allclose(rec.get_score(users), (want[users] @ X).toarray(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(rec.get_score_block(10, 20), (want[10:20] @ X).toarray(), rtol=1e-12, atol=1e-15)
    seen = rec.get_score_remove_seen(users)
    assert np.all(np.isneginf(seen[X[users].toarray() != 0]))
