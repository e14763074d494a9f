"""GPU parity tests of the kNN path vs the CPU oracle and the dense numpy formulas of
the reference's tests (tests/recommenders/test_knn.py:33-165).  Bar: top-k column
indices bit-exact; values within 1e-12 relative (fp64; exact for binary inputs).
"""
import numpy as np
import pytest
import scipy.sparse as sps

import oracle as O
from irspack_amd.recommenders import _knn as K
from irspack_amd.utils import okapi_BM_25_weight, remove_diagonal, tf_idf_weight

pytestmark = pytest.mark.gpu

rng = np.random.RandomState(0)
X_small = sps.csr_matrix(
    np.asarray([[1, 1, 2, 3, 4], [0, 1, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]], dtype=float))
_d = rng.rand(888, 512)
X_many = sps.csr_matrix((_d > 0.9).astype(float))
X_many.sort_indices()
X_many_dense = sps.csr_matrix(rng.rand(133, 245))


def assert_same_csr(a: sps.csr_matrix, b: sps.csr_matrix, rtol=1e-12):
    a, b = sps.csr_matrix(a), sps.csr_matrix(b)
    a.sort_indices()
    b.sort_indices()
    np.testing.assert_array_equal(a.indptr, b.indptr)
    np.testing.assert_array_equal(a.indices, b.indices)  # bit-exact top-k indices
    np.testing.assert_allclose(a.data, b.data, rtol=rtol, atol=0)


CASES = [
    ("cosine", dict(shrinkage=0.0, normalize=True)),
    ("cosine", dict(shrinkage=0.0, normalize=False)),
    ("cosine", dict(shrinkage=2.5, normalize=True)),
    ("jaccard", dict(shrinkage=0.0)),
    ("jaccard", dict(shrinkage=1.5)),
    ("asymmetric", dict(shrinkage=1.0, alpha=0.7)),
    ("tversky", dict(shrinkage=0.0, alpha=0.7, beta=2.0)),
]


def make(kind, X, **kw):
    n_threads = kw.pop("n_threads", 1)
    if kind == "cosine":
        g = K.CosineSimilarityComputer(X, kw["shrinkage"], kw["normalize"], n_threads)
    elif kind == "jaccard":
        g = K.JaccardSimilarityComputer(X, kw["shrinkage"], n_threads)
    elif kind == "asymmetric":
        g = K.AsymmetricSimilarityComputer(X, kw["shrinkage"], kw["alpha"], n_threads)
    elif kind == "tversky":
        g = K.TverskyIndexComputer(X, kw["shrinkage"], kw["alpha"], kw["beta"], n_threads)
    o = O.KNNComputer(kind, X, n_threads=n_threads, **kw)
    return g, o


@pytest.mark.parametrize("kind,kw", CASES)
@pytest.mark.parametrize("Xname", ["small", "many", "dense"])
@pytest.mark.parametrize("top_k", [2, 30, 100000])
def test_matches_oracle(kind, kw, Xname, top_k):
    X = {"small": X_small, "many": X_many, "dense": X_many_dense}[Xname]
    Xt = sps.csr_matrix(X.T)
    g, o = make(kind, Xt, **dict(kw))
    top_k = min(top_k, 1024)
    got = g.compute_similarity(Xt, top_k)
    exp = o.compute_similarity(Xt, top_k)
    # binary inputs give exact sums; float inputs differ by summation order only
    assert_same_csr(got, exp, rtol=1e-12)


@pytest.mark.parametrize("top_k", [700, 1024, 1025, 1500, 2048, 2500])
def test_top_k_around_the_candidate_list_size(top_k):
    """Binary data, 3,000 items that all co-occur: requests up to 1,024 take the approximate
    selection (its candidate list holds 2,048 columns), larger ones the exact kernel, above 2,048
    the threshold merge - the same rows as the oracle on every side of the two boundaries."""
    r = np.random.RandomState(11)
    Xt = sps.csr_matrix((r.rand(3000, 400) < 0.2).astype(float))
    g, o = make("cosine", Xt, shrinkage=0.0, normalize=True)
    assert_same_csr(g.compute_similarity(Xt, top_k), o.compute_similarity(Xt, top_k))


@pytest.mark.parametrize("X,normalize", [(X_many, True), (X_small, False), (X_many_dense, True)])
def test_cosine_dense_formula(X, normalize):
    # tests/recommenders/test_knn.py:33-51
    Xt = sps.csr_matrix(X.T)
    g = K.CosineSimilarityComputer(Xt, 0.0, normalize, 5)
    W = remove_diagonal(g.compute_similarity(Xt, X.shape[1])).toarray()
    manual = X.T.toarray()
    norm = (manual ** 2).sum(axis=1) ** 0.5
    manual = manual.dot(manual.T)
    if normalize:
        manual /= norm[:, None] * norm[None, :] + 1e-6
    np.fill_diagonal(manual, 0)
    np.testing.assert_allclose(W, manual)


@pytest.mark.parametrize("X", [X_many, X_small, X_many_dense])
def test_jaccard_dense_formula(X):
    # tests/recommenders/test_knn.py:54-70
    Xt = sps.csr_matrix(X.T)
    g = K.JaccardSimilarityComputer(Xt, 0.0, 1)
    W = remove_diagonal(g.compute_similarity(Xt, X.shape[1])).toarray()
    Xb = X.copy()
    Xb.data[:] = 1
    manual = Xb.T.toarray()
    norm = manual.sum(axis=1)
    manual = manual.dot(manual.T)
    denom = norm[:, None] + norm[None, :] - manual + 1e-6
    denom[denom <= 1e-10] = 1e-10
    manual = manual / denom
    np.fill_diagonal(manual, 0)
    np.testing.assert_allclose(W, manual)


def test_topk_ties_known_answer():
    # tests/recommenders/test_knn.py:144-165
    X = sps.csr_matrix(np.asarray(
        [[1, 1, 1, 0, 0], [1, 1, 0, 1, 0], [1, 0, 1, 1, 0], [0, 1, 1, 1, 1]], dtype=float))
    Xt = sps.csr_matrix(X.T)
    c = K.CosineSimilarityComputer(Xt, 0.0, False, 1, 128)
    full = c.compute_similarity(Xt, X.shape[1]).toarray()
    actual = c.compute_similarity(Xt, 2).toarray()
    expected = np.zeros_like(full)
    for row, scores in enumerate(full):
        cand = np.flatnonzero(scores)
        order = np.lexsort((cand, -scores[cand]))
        sel = cand[order[:2]]
        expected[row, sel] = scores[sel]
    np.testing.assert_array_equal(actual, expected)


@pytest.mark.parametrize("kind,kw", CASES)
def test_thousands_of_ties_at_the_threshold_go_to_the_exact_kernel(kind, kw, monkeypatch):
    """Binary data with a similarity that ends in a division selects on float32 approximations and
    ranks only the candidates exactly (knn_tile_kernel, FAST).  Here most of 3,000 items are
    identical columns: every one of them ties at the top_k-th value, the candidates do not fit the
    list (2,048) and the pair is handed to the exact kernel - the lowest columns must win
    (knn.hpp:119-136).  A few distinct items keep some rows on the approximate path; both settings
    of the switch must give the oracle's rows."""
    r = np.random.RandomState(5)
    n_users, n_items = 60, 3000
    base = (r.rand(n_users) < 0.5).astype(float)
    D = np.tile(base[:, None], (1, n_items))
    D[:, ::97] = (r.rand(n_users, len(range(0, n_items, 97))) < 0.4).astype(float)
    Xt = sps.csr_matrix(D.T)  # items x users
    g, o = make(kind, Xt, **dict(kw))
    want = o.compute_similarity(Xt, 10)
    assert_same_csr(g.compute_similarity(Xt, 10), want)
    monkeypatch.setenv("IRSPACK_AMD_KNN_FAST", "0")
    assert_same_csr(g.compute_similarity(Xt, 10), want)


@pytest.mark.parametrize("kind,kw", [("cosine", dict(shrinkage=1e31, normalize=True)),
                                     ("jaccard", dict(shrinkage=1e31)),
                                     ("tversky", dict(shrinkage=0.0, alpha=40.0, beta=0.5)),
                                     ("tversky", dict(shrinkage=2.0, alpha=0.5, beta=1e3))])
def test_parameters_outside_the_approximate_path_take_the_exact_kernel(kind, kw):
    """The float32 approximations bound their error only for non-negative denominators in
    float32 range (Tversky weights up to 16, shrinkage below 1e30; negative values are rejected at
    construction like the reference does): larger ones - which the reference accepts - run the
    exact fp64 kernel."""
    Xb = sps.csr_matrix((X_many > 0).astype(float))
    g, o = make(kind, Xb, **dict(kw))
    assert_same_csr(g.compute_similarity(Xb, 20), o.compute_similarity(Xb, 20), rtol=1e-11)


def test_explicit_zero_products_are_stored_entries():
    # knn.hpp:111-118: candidates are the stored entries of the product, exact zeros included
    X = sps.csr_matrix(np.asarray([[1.0, -1.0, 0.0], [1.0, 1.0, 2.0], [0.0, 0.0, 3.0]]))
    Xt = sps.csr_matrix(X.T)
    g, o = make("cosine", Xt, shrinkage=0.0, normalize=False)
    got, exp = g.compute_similarity(Xt, 3), o.compute_similarity(Xt, 3)
    assert_same_csr(got, exp)
    assert got.nnz == exp.nnz and (got.data == 0).any()


@pytest.mark.parametrize("kind,alpha,beta", [("p3alpha", 0.5, 0.0), ("p3alpha", 1.0, 0.0),
                                             ("rp3beta", 0.7, 0.4)])
def test_compute_w(kind, alpha, beta):
    X = X_many_dense
    Xt = sps.csr_matrix(X.T)
    if kind == "p3alpha":
        g = K.P3alphaComputer(Xt, alpha)
        o = O.KNNComputer("p3alpha", Xt, alpha=alpha)
    else:
        g = K.RP3betaComputer(Xt, alpha, beta)
        o = O.KNNComputer("rp3beta", Xt, alpha=alpha, beta=beta)
    got = g.compute_W(Xt, 40).tocsr()
    exp = o.compute_W(Xt, 40).tocsr()
    got.sort_indices()
    exp.sort_indices()
    np.testing.assert_array_equal(got.indptr, exp.indptr)
    np.testing.assert_array_equal(got.indices, exp.indices)
    np.testing.assert_allclose(got.data, exp.data, rtol=1e-11)


def test_thread_and_chunk_invariance_and_row_shards():
    Xt = sps.csr_matrix(X_many.T)
    a = K.CosineSimilarityComputer(Xt, 0.0, True, 1, 128).compute_similarity(Xt, 17)
    b = K.CosineSimilarityComputer(Xt, 0.0, True, 7, 3).compute_similarity(Xt, 17)
    assert_same_csr(a, b, rtol=0)
    c = K.CosineSimilarityComputer(Xt, 0.0, True)
    parts = [c.compute_similarity(Xt, 17, rows=(lo, hi)) for lo, hi in [(0, 100), (100, 101), (101, 512)]]
    assert_same_csr(sps.vstack(parts).tocsr(), a, rtol=0)


def test_argument_validation():
    # tests/recommenders/test_knn.py:239-251; knn.hpp:35-45; similarities.hpp:66,148-149
    Xt = sps.csr_matrix(X_small.T)
    with pytest.raises(ValueError):
        K.CosineSimilarityComputer(Xt, -1.0, False)
    with pytest.raises(ValueError):
        K.CosineSimilarityComputer(Xt, 0.0, False, 0)
    with pytest.raises(ValueError):
        K.CosineSimilarityComputer(Xt, 0.0, False, 1, 0)
    with pytest.raises(ValueError):
        K.AsymmetricSimilarityComputer(Xt, 0.0, 1.5)
    with pytest.raises(ValueError):
        K.TverskyIndexComputer(Xt, 0.0, -0.1, 0.0)
    c = K.CosineSimilarityComputer(Xt, 0.0, False)
    with pytest.raises(ValueError, match="illegal # of feature"):
        c.compute_similarity(sps.csr_matrix(X_small), 2)
    with pytest.raises(ValueError):
        remove_diagonal(sps.csr_matrix(np.ones((2, 3))))


def test_non_monotone_indptr_through_the_c_abi_is_an_error_not_a_wild_read():
    """irs_knn_compute validates the row pointers of the call before any host thread walks them
    (a C-ABI caller may hand over anything; scipy would never build such a matrix)."""
    import ctypes as C

    from irspack_amd import _lib

    rng2 = np.random.default_rng(0)
    X = sps.random(400, 300, density=0.05, format="csr", random_state=rng2, dtype=np.float64)
    X.data[:] = 1.0
    comp = K.CosineSimilarityComputer(sps.csr_matrix(X.T), 0.0, False)
    good = comp.compute_similarity(sps.csr_matrix(X.T), 5)
    assert good.shape == (300, 300)
    Xt = sps.csr_matrix(X.T)
    indptr = Xt.indptr.astype(np.int64).copy()
    indptr[100], indptr[101] = indptr[101] + 50, indptr[100]  # non-monotone in the middle
    bad = sps.csr_matrix(Xt.shape, dtype=np.float64)
    bad.indices, bad.data, bad.indptr = Xt.indices, Xt.data, indptr  # bypass scipy's own checks
    bad.has_sorted_indices = True  # (the wrapper would otherwise ask scipy to sort the garbage)
    with pytest.raises(ValueError, match="malformed indptr"):
        comp.compute_similarity(bad, 5)


def test_weighting_helpers_match_oracle():
    """irs_knn_weight (tables on host threads with libm's log, per-entry pass on the device: one IEEE
    operation per operation of util.hpp:183-184, :206, no contraction) against the oracle's sequential
    loops: BIT FOR BIT, values and pattern - binary and real-valued matrices, empty rows and columns,
    smooth on and off, a matrix large enough for several host threads and device blocks."""
    big = power_law_items(3000, 1200, 400000, 3)
    big_w = big.copy()
    big_w.data = np.random.RandomState(5).uniform(0.0, 4.0, big_w.nnz)
    holes = sps.csr_matrix(X_many).tolil()
    for r in (0, 5, 887):
        holes.rows[r], holes.data[r] = [], []
    holes = sps.csr_matrix(holes)
    for X in (X_small, X_many, X_many_dense, holes, big, big_w):
        pairs = [(tf_idf_weight(X), O.tf_idf_weight(X)), (tf_idf_weight(X, False), O.tf_idf_weight(X, False)),
                 (okapi_BM_25_weight(X, 1.3, 0.6), O.okapi_BM_25_weight(X, 1.3, 0.6)),
                 (okapi_BM_25_weight(X), O.okapi_BM_25_weight(X))]
        for mine, ref in pairs:
            mine, ref = sps.csr_matrix(mine), sps.csr_matrix(ref)
            ref.sort_indices()
            np.testing.assert_array_equal(mine.indptr, ref.indptr)
            np.testing.assert_array_equal(mine.indices, ref.indices)
            np.testing.assert_array_equal(mine.data, ref.data)  # (inf where smooth=False meets df = N... never stored)
    with pytest.raises(ValueError, match="column index out of range"):
        bad = sps.csr_matrix(X_many)
        bad = sps.csr_matrix((bad.data, bad.indices.copy(), bad.indptr), shape=bad.shape)
        bad.indices[7] = 512
        bad.has_sorted_indices = True
        tf_idf_weight(bad)


def test_two_column_tiles():
    # N > 16384 columns -> the product row is split over two LDS tiles and merged.
    rng2 = np.random.default_rng(5)
    N, U = 20000, 300
    rows = rng2.integers(0, U, size=60000)
    cols = rng2.integers(0, N, size=60000)
    X = sps.csr_matrix((np.ones(60000), (rows, cols)), shape=(U, N))
    X.data[:] = 1.0
    Xt = sps.csr_matrix(X.T)
    g, o = make("cosine", Xt, shrinkage=0.0, normalize=False)
    sel = (0, 3000)  # a shard of target rows keeps the CPU side quick
    got = g.compute_similarity(Xt, 50, rows=sel)
    exp = o.compute_similarity(Xt[sel[0]:sel[1]].tocsr() if False else Xt, 50)[sel[0]:sel[1]]
    assert_same_csr(got, exp, rtol=0)


def test_row_batching_is_transparent(monkeypatch):
    """Requests whose candidate scratch would be too large are cut into row batches."""
    from conftest import random_csr

    X = random_csr(300, 120, 0.1, 17, dtype=np.float64)
    comp = K.CosineSimilarityComputer(X, 0.5, True)
    whole = comp.compute_similarity(X, 7)
    monkeypatch.setattr(K._Computer, "_MAX_SLOT_ENTRIES", 7 * 37)  # 37 rows per call
    cut = comp.compute_similarity(X, 7)
    # indices are bit-exact; real-valued sums may differ in the last bit between two runs
    # (fp64 LDS atomics), batched or not
    assert np.array_equal(whole.indptr, cut.indptr) and np.array_equal(whole.indices, cut.indices)
    np.testing.assert_allclose(whole.data, cut.data, rtol=1e-12)
    assert comp.last_macs > 0


def test_many_tiles_and_large_top_k_merge_in_rounds():
    """3 column tiles x top_k = 1500 candidates exceed one merge buffer (4096): the row merge
    runs in rounds.  Indices bit-exact against the oracle."""
    from conftest import random_csr

    X = random_csr(33000, 40, 0.05, 23, dtype=np.float64, binary=True)
    comp = K.CosineSimilarityComputer(X, 0.0, True)
    ref = O.KNNComputer("cosine", X, 0.0, normalize=True, n_threads=8)
    rows = (100, 164)
    got = comp.compute_similarity(X, 1500, rows=rows)
    want = ref.compute_similarity(X, 1500)[rows[0]:rows[1]]
    assert np.diff(want.indptr).max() == 1500  # the cut is exercised
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    np.testing.assert_allclose(got.data, want.data, rtol=1e-12)


@pytest.mark.parametrize("values", ["ones", "weighted_target", "weighted_both"])
def test_long_slices_many_strips_per_block(values):
    """Target rows with ~600 stored users (several 16-user blocks per wave) whose slices span
    two 128-entry strips: more than 16 strips per block, i.e. several load batches per block,
    on each accumulator variant (32-bit counts, fp64 sums of y, fp64 sums of x * y)."""
    rng2 = np.random.default_rng(11)
    X = sps.random(1500, 600, density=0.4, random_state=rng2, format="csr", dtype=np.float64)
    X.data[:] = 1.0
    Xt = sps.csr_matrix(X.T)  # 600 target rows x 1500 features
    arg = Xt.copy()
    tgt = Xt.copy()
    if values != "ones":  # small integers: every sum is exact in fp64, so rtol = 0 holds
        tgt.data[:] = rng2.integers(1, 4, size=tgt.nnz).astype(np.float64)
    if values == "weighted_both":
        arg.data[:] = rng2.integers(1, 4, size=arg.nnz).astype(np.float64)
    comp = K.CosineSimilarityComputer(arg, 0.0, False)
    ref = O.KNNComputer("cosine", arg, 0.0, normalize=False, n_threads=8)
    got = comp.compute_similarity(tgt, 40)
    want = ref.compute_similarity(tgt, 40)
    assert_same_csr(got, want, rtol=0)


@pytest.mark.parametrize("n_items,top_k", [(3000, 2500), (17000, 3000), (17000, 17000)])
def test_top_k_above_the_lds_merge_cap(n_items, top_k):
    """top_k > 2048 (one and two column tiles; top_k = N keeps every stored product): the row
    merge selects a threshold key over the column-sorted tile lists instead of sorting in LDS.
    Binary data: massive ties at the cut, decided by column order (knn.hpp:119-136)."""
    from conftest import random_csr

    X = random_csr(n_items, 60, 0.25, 31, dtype=np.float64, binary=True)  # [items, users]
    comp = K.CosineSimilarityComputer(X, 0.0, False)  # raw co-occurrence counts
    ref = O.KNNComputer("cosine", X, 0.0, normalize=False, n_threads=8, max_chunk_size=8)
    rows = (40, 72)
    got = comp.compute_similarity(X, top_k, rows=rows)
    want = ref.compute_similarity(X[rows[0]:rows[1]], top_k)
    assert np.diff(want.indptr).max() > 2048
    if top_k < n_items:
        assert np.diff(want.indptr).max() == top_k  # the cut is exercised
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    np.testing.assert_array_equal(got.data, want.data)  # integer counts: exact


def test_p3alpha_default_top_k_none_on_a_catalogue_above_2048_items():
    """P3alphaRecommender / RP3betaRecommender keep the reference's default
    top_k = None (-> n_items, p3.py:62-66); with more than 2048 items that used to raise."""
    from conftest import random_csr
    from irspack_amd.recommenders import P3alphaRecommender, RP3betaRecommender

    X = random_csr(150, 2600, 0.02, 5, dtype=np.float64, binary=True)  # users x items
    rec = P3alphaRecommender(X, alpha=1.0).learn()
    want = O.KNNComputer("p3alpha", sps.csr_matrix(X.T), alpha=1.0).compute_W(sps.csr_matrix(X.T), 2600)
    got = sps.csc_matrix(rec.W)
    got.sort_indices()
    want.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    np.testing.assert_allclose(got.data, want.data, rtol=1e-12, atol=0)
    rec2 = RP3betaRecommender(X, alpha=1.0, beta=0.6).learn()
    assert rec2.W.shape == (2600, 2600) and rec2.W.nnz > 0


def test_weighted_sums_do_not_depend_on_arrival_order():
    """tf-idf weighted item-kNN (knn.py:67-80: computer on the weighted matrix, query with the
    unweighted one), 2,000 users x 5,000 items: the weighted products are summed in 64-bit
    fixed point, so two runs are bit-identical (the LDS atomics land in a different order
    every run) and the top-k sets equal the oracle's."""
    from conftest import random_csr

    X = random_csr(2000, 5000, 0.02, 77, dtype=np.float64, binary=True)
    Xw = O.tf_idf_weight(X)
    arg, tgt = sps.csr_matrix(Xw.T), sps.csr_matrix(X.T)
    comp = K.CosineSimilarityComputer(arg, 0.0, True)
    a = comp.compute_similarity(tgt, 50)
    b = comp.compute_similarity(tgt, 50)
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data.view(np.uint64), b.data.view(np.uint64))  # bit for bit
    want = O.KNNComputer("cosine", arg, 0.0, normalize=True, n_threads=8).compute_similarity(tgt, 50)
    a.sort_indices()
    want.sort_indices()
    assert np.array_equal(a.indptr, want.indptr) and np.array_equal(a.indices, want.indices)
    np.testing.assert_allclose(a.data, want.data, rtol=1e-12, atol=0)
    # both operands weighted (user-kNN with BM25, user_knn.py:62-76), both signs of rounding
    Xb = O.okapi_BM_25_weight(X[:600])
    cw = K.CosineSimilarityComputer(Xb, 0.5, True)
    c1, c2 = cw.compute_similarity(Xb, 30), cw.compute_similarity(Xb, 30)
    assert np.array_equal(c1.indices, c2.indices)
    assert np.array_equal(c1.data.view(np.uint64), c2.data.view(np.uint64))
    w2 = O.KNNComputer("cosine", Xb, 0.5, normalize=True, n_threads=8).compute_similarity(Xb, 30)
    c1.sort_indices()
    w2.sort_indices()
    assert np.array_equal(c1.indices, w2.indices)
    np.testing.assert_allclose(c1.data, w2.data, rtol=1e-12, atol=0)


def test_adversarial_exact_ties_with_weights():
    """Dyadic weights (every sum exact in fp64 and in fixed point) and duplicated columns:
    many exactly tied similarities at the cut, decided by column order like knn.hpp:119-125;
    values must be EQUAL to the oracle's, not just close."""
    rng2 = np.random.default_rng(5)
    U, I = 300, 400
    dense = (rng2.random((U, I)) < 0.1) * (rng2.integers(1, 257, size=(U, I)) / 64.0)
    dense[:, 200:260] = dense[:, 100:160]  # 60 duplicated items: exact ties everywhere
    X = sps.csr_matrix(dense)
    arg = sps.csr_matrix(X.T)
    for normalize in (False, True):
        comp = K.CosineSimilarityComputer(arg, 0.0, normalize)
        got = comp.compute_similarity(arg, 7)
        want = O.KNNComputer("cosine", arg, 0.0, normalize=normalize).compute_similarity(arg, 7)
        got.sort_indices()
        want.sort_indices()
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        if not normalize:
            np.testing.assert_array_equal(got.data, want.data)
        else:
            np.testing.assert_allclose(got.data, want.data, rtol=1e-14, atol=0)


@pytest.mark.parametrize("mode", ["by_range", "forced", "off"])
def test_weights_over_twelve_decades_keep_their_small_sums(mode, monkeypatch):
    """Weights from 1e-6 to 1e6 in one matrix: a column whose sum is 1e-12 of its row's largest
    possible sum has few significant bits left in a one-limb fixed-point accumulator (one scale per
    row).  Such rows are summed in TWO 64-bit limbs (two passes over the slices, the second one
    accumulating what the first rounding left out): values within 1e-12 of the oracle's float64
    sums for every stored column, bit-identical from run to run, top-k sets equal.  "off"
    (IRSPACK_AMD_KNN_WIDE=0, the one-limb sums) shows that the case is real: its small sums are
    NOT within 1e-12."""
    if mode != "by_range":
        monkeypatch.setenv("IRSPACK_AMD_KNN_WIDE", "1" if mode == "forced" else "0")
    rng2 = np.random.default_rng(9)
    U, I = 1500, 900
    mask = rng2.random((U, I)) < 0.03
    decades = rng2.integers(-6, 7, size=I)  # one magnitude per item: products span 1e-12 .. 1e12
    W = np.where(mask, rng2.uniform(1.0, 9.0, size=(U, I)) * 10.0 ** decades[None, :], 0.0)
    arg = sps.csr_matrix(W.T)            # items x users, weighted
    tgt = sps.csr_matrix(W.T)            # weighted targets too (user-kNN style: both operands)
    top_k = I                            # every stored column: small and large sums alike
    comp = K.CosineSimilarityComputer(arg, 0.0, False)  # normalize=False: the raw sums
    a = comp.compute_similarity(tgt, top_k)
    b = comp.compute_similarity(tgt, top_k)
    want = O.KNNComputer("cosine", arg, 0.0, normalize=False, n_threads=8).compute_similarity(tgt, top_k)
    for m in (a, b, want):
        m.sort_indices()
    assert np.array_equal(a.indptr, want.indptr) and np.array_equal(a.indices, want.indices)
    assert np.array_equal(a.data.view(np.uint64), b.data.view(np.uint64))  # order-independent sums
    rel = np.abs(a.data - want.data) / np.abs(want.data)
    if mode == "off":
        assert rel.max() > 1e-9  # (the one-limb sums lose the small columns: the reason for two limbs)
    else:
        assert rel.max() <= 1e-12, float(rel.max())
    # and the usual top-k request on the same data: same sets as the oracle
    if mode != "off":
        a5 = comp.compute_similarity(tgt, 20)
        w5 = O.KNNComputer("cosine", arg, 0.0, normalize=False, n_threads=8).compute_similarity(tgt, 20)
        a5.sort_indices()
        w5.sort_indices()
        assert np.array_equal(a5.indices, w5.indices)
        np.testing.assert_allclose(a5.data, w5.data, rtol=1e-12, atol=0)


def power_law_items(U, N, nnz, seed):
    """items x users, binary, item popularity ~ 1 / rank (the ML-20M-shaped generator's law)"""
    rng2 = np.random.default_rng(seed)
    w = 1.0 / (np.arange(N) + 1.0)
    cols = rng2.choice(N, size=nnz, p=w / w.sum())
    rows = rng2.integers(0, U, size=nnz)
    perm = rng2.permutation(N)  # popular items anywhere in the column range
    X = sps.csr_matrix((np.ones(nnz), (rows, perm[cols])), shape=(U, N))
    X.data[:] = 1.0
    return sps.csr_matrix(X.T)


# ---------------------------------------------------------------- row chunks of one call (round 5)
@pytest.mark.parametrize("chunks", ["1", "3", "7", "64"])
@pytest.mark.parametrize("kind,kw", CASES)
def test_row_chunks_of_a_call_are_transparent(kind, kw, chunks, monkeypatch):
    """A call's rows are walked and launched in chunks so that the host pass and the uploads of chunk
    k + 1 overlap the kernels of chunk k (IRSPACK_AMD_KNN_CHUNKS; 4 by default on large calls).  Every
    chunk chooses its accumulator from its own rows: the first third of this target is binary (counts),
    the rest weighted (fixed-point sums), three rows are empty, one chunk boundary falls on empty rows."""
    monkeypatch.setenv("IRSPACK_AMD_KNN_CHUNKS", chunks)
    Xt = sps.csr_matrix(X_many.T).tolil()
    for r in (0, 170, 171, 511):
        Xt.rows[r], Xt.data[r] = [], []
    Xt = sps.csr_matrix(Xt)
    g, o = make(kind, Xt, **dict(kw))
    assert_same_csr(g.compute_similarity(Xt, 25), o.compute_similarity(Xt, 25), rtol=1e-12)
    T = Xt.copy()
    w = np.random.RandomState(3).uniform(0.5, 2.0, T.nnz)
    w[: T.indptr[170]] = 1.0
    T.data = w if kind in ("cosine", "asymmetric") else T.data
    assert_same_csr(g.compute_similarity(T, 25), o.compute_similarity(T, 25), rtol=1e-12)
    sel = (100, 400)
    assert_same_csr(g.compute_similarity(T, 9, rows=sel), o.compute_similarity(T, 9)[sel[0]:sel[1]], rtol=1e-12)


@pytest.mark.parametrize("chunks", ["1", "5"])
def test_row_chunks_compute_w_and_errors(chunks, monkeypatch):
    monkeypatch.setenv("IRSPACK_AMD_KNN_CHUNKS", chunks)
    Xt = sps.csr_matrix(X_many.T)
    g = K.RP3betaComputer(Xt, 0.8, 0.4, 1)
    o = O.KNNComputer("rp3beta", Xt, alpha=0.8, beta=0.4, n_threads=1)
    assert_same_csr(g.compute_W(Xt, 30), o.compute_W(Xt, 30), rtol=1e-11)
    # an index out of range in the LAST chunk is an error although earlier chunks were already launched,
    # and the computer stays usable
    bad = Xt.copy()
    bad.indices = bad.indices.copy()
    bad.indices[-1] = Xt.shape[1] + 5
    c = K.CosineSimilarityComputer(Xt, 0.0, True)
    with pytest.raises(ValueError):
        c.compute_similarity(bad, 10)
    assert_same_csr(c.compute_similarity(Xt, 10), O.KNNComputer("cosine", Xt, 0.0, normalize=True, n_threads=1).compute_similarity(Xt, 10))


# ---------------------------------------------------------------- construction on the device (round 5)
@pytest.mark.parametrize("kind,kw", CASES)
@pytest.mark.parametrize("shape", ["one_tile", "two_tiles"])
def test_device_create_is_the_host_create(kind, kw, shape, monkeypatch):
    """X_arg^T, its slice pointers, packed offsets and per-row value ranges are built on the device from
    the caller's arrays (no host copy / transpose); IRSPACK_AMD_KNN_DEVICE_CREATE=0 keeps the host
    construction.  Same similarities bit for bit - empty rows and columns, one and two column tiles,
    values that are not ones under Jaccard / Tversky (binarised), weighted matrices."""
    if shape == "one_tile":
        Xt = sps.csr_matrix(X_many.T).tolil()
        for r in (0, 200, 511):
            Xt.rows[r], Xt.data[r] = [], []
        Xt = sps.csr_matrix(Xt)
    else:
        Xt = power_law_items(900, 17000, 120000, 7)
    if kind in ("jaccard", "tversky"):
        Xt = Xt.copy()
        Xt.data = np.random.RandomState(1).uniform(0.0, 3.0, Xt.nnz)  # (binarised by the similarity)
    g_dev, o = make(kind, Xt, **dict(kw))
    monkeypatch.setenv("IRSPACK_AMD_KNN_DEVICE_CREATE", "0")
    g_host, _ = make(kind, Xt, **dict(kw))
    monkeypatch.delenv("IRSPACK_AMD_KNN_DEVICE_CREATE")
    a, b = g_dev.compute_similarity(Xt, 30), g_host.compute_similarity(Xt, 30)
    assert_same_csr(a, b, rtol=0)
    assert_same_csr(a, o.compute_similarity(Xt, 30), rtol=1e-12)
    if kind in ("cosine", "asymmetric"):  # weighted: the values are transposed on the device too
        W = Xt.copy()
        W.data = np.random.RandomState(2).uniform(0.5, 2.0, W.nnz)
        W.data[::7] = 1.0
        gw, ow = make(kind, W, **dict(kw))
        monkeypatch.setenv("IRSPACK_AMD_KNN_DEVICE_CREATE", "0")
        gw_host, _ = make(kind, W, **dict(kw))
        monkeypatch.delenv("IRSPACK_AMD_KNN_DEVICE_CREATE")
        aw = gw.compute_similarity(W, 30)
        assert_same_csr(aw, gw_host.compute_similarity(W, 30), rtol=0)
        assert_same_csr(aw, ow.compute_similarity(W, 30), rtol=1e-12)


@pytest.mark.parametrize("kind,alpha,beta", [("p3alpha", 0.5, 0.0), ("p3alpha", 1.0, 0.0), ("rp3beta", 0.7, 0.4)])
def test_device_create_p3alpha_rp3beta(kind, alpha, beta, monkeypatch):
    """compute_W computers: the rows are pow-ed and normalised on the host (libm), everything else of the
    construction runs on the device - same W bit for bit as the host construction."""
    Xt = sps.csr_matrix(X_many.T).tolil()
    Xt.rows[3], Xt.data[3] = [], []
    Xt = sps.csr_matrix(Xt)
    Xt.data = np.random.RandomState(4).uniform(0.5, 3.0, Xt.nnz)

    def both():
        if kind == "p3alpha":
            return K.P3alphaComputer(Xt, alpha), O.KNNComputer("p3alpha", Xt, alpha=alpha)
        return K.RP3betaComputer(Xt, alpha, beta), O.KNNComputer("rp3beta", Xt, alpha=alpha, beta=beta)

    g_dev, o = both()
    monkeypatch.setenv("IRSPACK_AMD_KNN_DEVICE_CREATE", "0")
    g_host, _ = both()
    monkeypatch.delenv("IRSPACK_AMD_KNN_DEVICE_CREATE")
    a = g_dev.compute_W(Xt, 40)
    assert_same_csr(a, g_host.compute_W(Xt, 40), rtol=0)
    assert_same_csr(a, o.compute_W(Xt, 40), rtol=1e-11)


@pytest.mark.parametrize("kind,kw", [c for c in CASES if c[0] in ("cosine", "asymmetric")])
@pytest.mark.parametrize("shape", ["one_tile", "two_tiles"])
def test_row_constant_values_skip_the_value_stream(kind, kw, shape, monkeypatch):
    """TF-IDF of binary interactions: every feature row of X_arg^T holds one value (its idf), found at
    construction; the kernels then run their all-ones form on y' = x_u y and never read the float64 value
    stream.  The same products x y: bit for bit the general weighted kernel's result (the host
    construction keeps that one), one and two column tiles, a weighted AND a binary target."""
    Xb = sps.csr_matrix(X_many.T) if shape == "one_tile" else power_law_items(900, 17000, 120000, 11)
    Xw = tf_idf_weight(Xb)
    g, o = make(kind, Xw, **dict(kw))
    monkeypatch.setenv("IRSPACK_AMD_KNN_DEVICE_CREATE", "0")
    g_general, _ = make(kind, Xw, **dict(kw))
    monkeypatch.delenv("IRSPACK_AMD_KNN_DEVICE_CREATE")
    for T in (Xw, Xb):
        a = g.compute_similarity(T, 30)
        assert_same_csr(a, g_general.compute_similarity(T, 30), rtol=0)
        # against the oracle: the top-k VALUES of every row (idf weights of binary data produce sums that
        # are equal in exact arithmetic; the oracle's sequential float64 sums break such ties by their
        # rounding, the fixed-point sums by column - test_adversarial_exact_ties_with_weights)
        want = sps.csr_matrix(o.compute_similarity(T, 30))
        assert np.array_equal(a.indptr, want.indptr)
        for r in range(0, a.shape[0], 7):
            ga = np.sort(a.data[a.indptr[r]:a.indptr[r + 1]])
            wa = np.sort(want.data[want.indptr[r]:want.indptr[r + 1]])
            np.testing.assert_allclose(ga, wa, rtol=1e-12, atol=0)


# ---------------------------------------------------------------- CSC inputs and fused weighting (round 6)
def _weighted_copy(X, seed):
    W = X.copy()
    W.data = np.random.RandomState(seed).uniform(0.5, 2.0, W.nnz)
    W.data[::5] = 1.0
    return W


@pytest.mark.parametrize("kind,kw", CASES)
@pytest.mark.parametrize("shape", ["one_tile", "two_tiles"])
def test_csc_layout_is_the_csr_layout(kind, kw, shape, monkeypatch):
    """The reference's recommenders pass ``X.T`` - a CSC matrix - to the computers (knn.py:77-79).  The
    library takes the CSC arrays as they are (IRS_LAYOUT_CSC): for the constructor they ARE X_arg^T (no
    transpose at all; the norms of a weighted matrix from a device regrouping, their squares added in row
    order), for ``compute_similarity`` the target's columns are regrouped on host threads.  Same result
    BIT FOR BIT as the CSR input of the same matrix: binary and weighted values, every similarity, empty
    rows / columns, one and two column tiles, a row range, and the host construction
    (IRSPACK_AMD_KNN_DEVICE_CREATE=0)."""
    if shape == "one_tile":
        Xt = sps.csr_matrix(X_many.T).tolil()
        for r in (0, 200, 511):
            Xt.rows[r], Xt.data[r] = [], []
        Xt = sps.csr_matrix(Xt)
    else:
        Xt = power_law_items(900, 17000, 120000, 7)
    for M in (Xt, _weighted_copy(Xt, 3)):
        M_csc = sps.csc_matrix(M)
        M_csc.sort_indices()
        g_csr, o = make(kind, M, **dict(kw))
        g_csc, _ = make(kind, M_csc, **dict(kw))
        want = g_csr.compute_similarity(M, 30)
        assert_same_csr(want, o.compute_similarity(M, 30), rtol=1e-12)
        assert_same_csr(g_csc.compute_similarity(M, 30), want, rtol=0)       # CSC constructor, CSR target
        assert_same_csr(g_csr.compute_similarity(M_csc, 30), want, rtol=0)   # CSR constructor, CSC target
        assert_same_csr(g_csc.compute_similarity(M_csc, 30), want, rtol=0)
        part = g_csc.compute_similarity(M_csc, 30, rows=(100, 300))
        assert_same_csr(part, want[100:300], rtol=0)
        monkeypatch.setenv("IRSPACK_AMD_KNN_DEVICE_CREATE", "0")
        g_host, _ = make(kind, M_csc, **dict(kw))
        monkeypatch.delenv("IRSPACK_AMD_KNN_DEVICE_CREATE")
        assert_same_csr(g_host.compute_similarity(M_csc, 30), want, rtol=0)


@pytest.mark.parametrize("kind,alpha,beta", [("p3alpha", 0.5, 0.0), ("rp3beta", 0.7, 0.4)])
def test_csc_layout_compute_w(kind, alpha, beta):
    """P3alpha / RP3beta recommenders pass ``X.T`` (CSC) to the constructor and to compute_W
    (p3.py:61-68, rp3.py:68-76): regrouped on host threads with their values."""
    Xt = sps.csr_matrix(X_many.T).copy()
    Xt.data = np.random.RandomState(4).uniform(0.5, 3.0, Xt.nnz)
    Xc = sps.csc_matrix(Xt)
    Xc.sort_indices()
    mk = (lambda M: K.P3alphaComputer(M, alpha)) if kind == "p3alpha" else (lambda M: K.RP3betaComputer(M, alpha, beta))
    want = sps.csr_matrix(mk(Xt).compute_W(Xt, 40))
    for ctor_in, target in ((Xc, Xt), (Xt, Xc), (Xc, Xc)):
        assert_same_csr(sps.csr_matrix(mk(ctor_in).compute_W(target, 40)), want, rtol=0)


@pytest.mark.parametrize("kind,kw", [c for c in CASES if c[0] in ("cosine", "asymmetric")])
@pytest.mark.parametrize("scheme", [("TF_IDF", True), ("TF_IDF", False), ("BM_25", 1.2, 0.75), ("BM_25", 0.7, 0.3)])
@pytest.mark.parametrize("values", ["binary", "real"])
def test_fused_weighting_is_the_host_weighting(kind, kw, scheme, values, monkeypatch):
    """``Computer(X.T, weighting=w)`` (item-kNN: knn.py:67-77) and ``Computer(X, weighting=w)`` (user-kNN:
    user_knn.py:62-74) weight the stored matrix on the device while it is uploaded; the similarities are
    those of a computer built on the oracle's weighted matrix BIT FOR BIT (same weights - one IEEE
    operation each -, norms added in the same order), and match the oracle's computer to 1e-12."""
    X = power_law_items(700, 2500, 60000, 21).T.tocsr()  # [users, items]
    X.sort_indices()
    if values == "real":
        X = X.copy()
        X.data = np.random.RandomState(8).uniform(0.2, 5.0, X.nnz)
    Xw = O.tf_idf_weight(X, scheme[1]) if scheme[0] == "TF_IDF" else O.okapi_BM_25_weight(X, scheme[1], scheme[2])
    Xw.sort_indices()
    # item-kNN orientation: the computer sees X_w.T, the target is the unweighted X.T
    # (binary interactions would take the separable form - one factor per feature row, one per column, see
    # test_separable_weighting: same values to rounding, not bit for bit; here the general form)
    monkeypatch.setenv("IRSPACK_AMD_KNN_SEPARABLE", "0")
    plain, o = make(kind, sps.csr_matrix(Xw.T), **dict(kw))
    want = plain.compute_similarity(sps.csr_matrix(X.T), 25)
    assert_same_csr(want, o.compute_similarity(sps.csr_matrix(X.T), 25), rtol=1e-12)
    if kind == "cosine":
        fused = K.CosineSimilarityComputer(X.T, kw["shrinkage"], kw["normalize"], weighting=scheme)
    else:
        fused = K.AsymmetricSimilarityComputer(X.T, kw["shrinkage"], kw["alpha"], weighting=scheme)
    assert_same_csr(fused.compute_similarity(X.T, 25), want, rtol=0)
    # user-kNN orientation: the computer sees X_w, the target is X (CSR layout, weighting over its own rows)
    plain_u, _ = make(kind, Xw, **dict(kw))
    want_u = plain_u.compute_similarity(X, 25)
    if kind == "cosine":
        fused_u = K.CosineSimilarityComputer(X, kw["shrinkage"], kw["normalize"], weighting=scheme)
    else:
        fused_u = K.AsymmetricSimilarityComputer(X, kw["shrinkage"], kw["alpha"], weighting=scheme)
    assert_same_csr(fused_u.compute_similarity(X, 25), want_u, rtol=0)


def test_fused_weighting_argument_errors():
    X = sps.csr_matrix(X_many)
    with pytest.raises(ValueError, match="weighting"):
        K.CosineSimilarityComputer(X.T, 0.0, True, weighting=("LOG",))
    # Jaccard / Tversky binarise their input: they take no weighting argument at all
    with pytest.raises(TypeError):
        K.JaccardSimilarityComputer(X.T, 0.0, weighting=("TF_IDF", True))


@pytest.mark.parametrize("kind,kw", [CASES[0], CASES[3], CASES[5]])
@pytest.mark.parametrize("shape", ["one_tile", "two_tiles", "tiny"])
def test_result_as_csc_without_diagonal(kind, kw, shape):
    """``remove_diagonal(compute_similarity(X, k)).tocsc()`` (knn.py:78-80) regrouped on the device
    (irs_knn_fetch_csc): the same CSC matrix entry for entry - explicit zeros on the diagonal kept
    (util.hpp:211-226), row numbers ascending in every column, empty columns, an empty result."""
    if shape == "one_tile":
        Xt = sps.csr_matrix(X_many.T)
    elif shape == "two_tiles":
        Xt = power_law_items(900, 17000, 120000, 7)
    else:
        Xt = sps.csr_matrix(X_small.T)
    for M in (Xt, _weighted_copy(Xt, 9)):
        g, _ = make(kind, M, **dict(kw))
        for top_k in (30, 1, 0):
            want = remove_diagonal(g.compute_similarity(M, top_k)).tocsc()
            want.sort_indices()
            got = g.compute_similarity_without_diagonal_csc(M, top_k)
            assert sps.isspmatrix_csc(got) and got.shape == want.shape
            np.testing.assert_array_equal(got.indptr, want.indptr)
            np.testing.assert_array_equal(got.indices, want.indices)
            np.testing.assert_array_equal(got.data, want.data)
            if top_k == 30:
                assert (got.diagonal() == 0).all() and got.nnz == want.nnz
    with pytest.raises(ValueError, match="square"):
        g.compute_similarity_without_diagonal_csc(M[:2], 3)


def assert_same_topk_up_to_rounding(got, want, rtol=1e-12):
    """Two top-k results whose sums were rounded differently: per row the same values to `rtol` (sorted),
    and the same columns except where a column's value is within `rtol` of another candidate's (a tie in
    exact arithmetic may be broken either way by the last bit)."""
    got, want = sps.csr_matrix(got), sps.csr_matrix(want)
    np.testing.assert_array_equal(got.indptr, want.indptr)
    for r in range(got.shape[0]):
        sl = slice(got.indptr[r], got.indptr[r + 1])
        gv, wv = got.data[sl], want.data[sl]
        np.testing.assert_allclose(np.sort(gv), np.sort(wv), rtol=rtol, atol=0)
        gi, wi = got.indices[sl], want.indices[sl]
        if np.array_equal(gi, wi):
            continue
        allv = np.concatenate([gv, wv])
        for j in np.setxor1d(gi, wi):
            v = gv[gi == j][0] if j in gi else wv[wi == j][0]
            near = np.sum(np.abs(allv - v) <= rtol * max(abs(v), 1e-300))
            assert near >= 3, (r, int(j), float(v))  # itself (once or twice) + at least one rival within rounding


@pytest.mark.parametrize("scheme", [("TF_IDF", True), ("BM_25", 1.2, 0.75)])
@pytest.mark.parametrize("orient", ["item", "user"])
@pytest.mark.parametrize("kind,kw", [CASES[0], CASES[1], CASES[2], CASES[5]])
def test_separable_weighting(kind, kw, scheme, orient, monkeypatch):
    """tf-idf / BM25 of BINARY interactions factor into (a factor per document) x (a factor per term): the
    fused constructor keeps the two factors instead of a float64 value per entry - pure tf-idf of an item-kNN
    runs the exact COUNT kernel and scales the finished counts per column.  Against (a) the general weighted
    form of the same computer (IRSPACK_AMD_KNN_SEPARABLE=0) and (b) the oracle on the oracle-weighted matrix:
    values to 1e-12; columns identical except between candidates that agree to 1e-12 (ties in exact
    arithmetic - equal idf and equal counts - are EXACT ties in the separable form and come out by ascending
    column, knn.hpp:119-129; sums of individually rounded weights break them by their last bits)."""
    X = power_law_items(600, 1800, 40000, 33).T.tocsr()  # [users, items], binary
    X.sort_indices()
    Xw = O.tf_idf_weight(X, scheme[1]) if scheme[0] == "TF_IDF" else O.okapi_BM_25_weight(X, scheme[1], scheme[2])
    Xw.sort_indices()
    arg, target, ow = (X.T, X.T, sps.csr_matrix(Xw.T)) if orient == "item" else (X, X, Xw)

    def fused():
        if kind == "cosine":
            return K.CosineSimilarityComputer(arg, kw["shrinkage"], kw["normalize"], weighting=scheme)
        return K.AsymmetricSimilarityComputer(arg, kw["shrinkage"], kw["alpha"], weighting=scheme)

    sep = fused().compute_similarity(target, 20)
    monkeypatch.setenv("IRSPACK_AMD_KNN_SEPARABLE", "0")
    general = fused().compute_similarity(target, 20)
    monkeypatch.delenv("IRSPACK_AMD_KNN_SEPARABLE")
    assert_same_topk_up_to_rounding(sep, general)
    _, o = make(kind, ow, **dict(kw))
    assert_same_topk_up_to_rounding(sep, o.compute_similarity(sps.csr_matrix(target), 20))


def test_separable_weighting_tie_order_known_answer():
    """The 4 x 5 tie matrix of tests/recommenders/test_knn.py:144-165 with tf-idf weights, top_k = 2: the
    separable form's value of column j is fl(count_ij * idf_j) through the un-fused cosine epilogue - restated
    here in numpy operation by operation - and candidates of equal value come out by ascending column."""
    X = sps.csr_matrix(np.asarray([[1, 1, 1, 0, 0], [1, 1, 0, 1, 0], [1, 0, 1, 1, 0], [0, 1, 1, 1, 1]], dtype=float))
    Xw = O.tf_idf_weight(X, True)
    idf = np.log(X.shape[0] / (np.bincount(X.indices, minlength=5) + 1.0))
    Wd, Xd = Xw.toarray(), X.toarray()
    norms = np.empty(5)
    for j in range(5):
        ss = 0.0
        for u in range(4):  # stored entries of column j in row order
            if Xd[u, j]:
                ss += Wd[u, j] * Wd[u, j]
        norms[j] = np.sqrt(ss)
    tnorm = np.sqrt(Xd.sum(axis=0))
    counts = Xd.T @ Xd
    got = K.CosineSimilarityComputer(X.T, 0.0, True, weighting=("TF_IDF", True)).compute_similarity(X.T, 2)
    got.sort_indices()
    for i in range(5):
        cols = np.flatnonzero(counts[i] > 0)
        vals = (counts[i, cols] * idf[cols]) / (norms[cols] * tnorm[i] + 0.0 + 1e-6)
        order = np.lexsort((cols, -vals))[:2]
        sel = np.sort(cols[order])
        sl = slice(got.indptr[i], got.indptr[i + 1])
        np.testing.assert_array_equal(got.indices[sl], sel)
        np.testing.assert_array_equal(got.data[sl], vals[np.searchsorted(cols, sel)])
