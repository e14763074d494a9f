"""GPU parity tests of ``retrieve_recommend_from_score`` (cpp_source/util.hpp:426-504) against
a direct Python restatement: candidates in list order, best score first, stop at -inf, scores
narrowed to float32.  Ties are compared as sets of equal-score items (the reference's
comparator leaves their order unspecified, util.hpp:483-485).
"""
import numpy as np
import pytest

from irspack_amd.utils import (retrieve_recommend_from_score, retrieve_recommend_from_score_f32,
                               retrieve_recommend_from_score_f64)

pytestmark = pytest.mark.gpu


def restated(score, allowed, cutoff):
    rows, n_items = score.shape
    out = []
    for r in range(rows):
        if len(allowed) == 0:
            cand = list(range(n_items))
        else:
            lst = allowed[0] if len(allowed) == 1 else allowed[r]
            cand = [i for i in lst if 0 <= i < n_items]
        pairs = sorted(((i, score[r, i]) for i in cand), key=lambda t: -t[1])  # stable
        res = []
        for i, s in pairs[:cutoff]:
            if s == -np.inf:
                break
            res.append((int(i), float(np.float32(s))))
        out.append(res)
    return out


def assert_same(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert [s for _, s in g] == [s for _, s in w]
        # equal scores may come in any order: compare the item sets per score value
        for v in set(s for _, s in w):
            assert sorted(i for i, s in g if s == v) == sorted(i for i, s in w if s == v)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("cutoff", [1, 7, 64, 65, 300])
def test_all_items(dtype, cutoff):
    rng = np.random.default_rng(cutoff)
    score = rng.standard_normal((23, 257)).astype(dtype)
    score[3, :] = -np.inf
    score[5, 10:] = -np.inf
    score[7, ::2] = -np.inf
    got = retrieve_recommend_from_score(score, [], cutoff, 2)
    assert_same(got, restated(score, [], cutoff))
    assert got[3] == [] and len(got[5]) == min(cutoff, 10)


def test_global_and_per_row_lists():
    rng = np.random.default_rng(0)
    score = rng.standard_normal((9, 120)).astype(np.float32)
    glob = [[5, 3, 3, 119, 120, -1, 77, 0]]  # duplicates kept, out-of-range dropped
    assert_same(retrieve_recommend_from_score_f32(score, glob, 5, 1), restated(score, glob, 5))
    per = [list(rng.integers(-3, 125, size=rng.integers(0, 40))) for _ in range(9)]
    per[4] = []
    got = retrieve_recommend_from_score_f32(score, per, 10, 3)
    assert_same(got, restated(score, per, 10))
    assert got[4] == []


def test_f64_is_narrowed_and_ties():
    score = np.zeros((2, 50), dtype=np.float64)
    score[0, 7] = 1.0 + 1e-12  # not representable in float32
    score[1, :] = 2.5
    got = retrieve_recommend_from_score_f64(score, [], 3, 1)
    assert got[0][0] == (7, 1.0)
    assert len(got[1]) == 3 and all(s == 2.5 for _, s in got[1])
    assert_same(got, restated(score, [], 3))


def test_errors():
    score = np.zeros((4, 10), dtype=np.float32)
    with pytest.raises(ValueError, match="n_threads"):
        retrieve_recommend_from_score(score, [], 3, 0)
    with pytest.raises(ValueError, match="allowed_indices"):
        retrieve_recommend_from_score(score, [[1], [2]], 3, 1)
    with pytest.raises(ValueError, match="float32 or float64"):
        retrieve_recommend_from_score(score.astype(np.int32), [], 3, 1)
    assert retrieve_recommend_from_score(score, [], 0, 1) == [[], [], [], []]
    assert retrieve_recommend_from_score(score[:0], [], 3, 1) == []


def test_ml100k_width_rows_with_list_longer_than_items():
    """A candidate list may be longer than n_items (duplicates): the key cache is sized by it."""
    rng = np.random.default_rng(2)
    score = rng.standard_normal((3, 40)).astype(np.float32)
    lst = [list(rng.integers(0, 40, size=500))]
    assert_same(retrieve_recommend_from_score(score, lst, 20, 1), restated(score, lst, 20))


def test_cutoff_is_clamped_to_the_candidate_count():
    """cutoff = n_items ("rank everything") is clamped like util.hpp:476-481."""
    rng = np.random.default_rng(1)
    score = rng.standard_normal((6, 40)).astype(np.float32)
    got = retrieve_recommend_from_score(score, [], 10 ** 9, 1)
    assert_same(got, restated(score, [], 40))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cutoff_above_2048(dtype):
    """Any cutoff the reference accepts (util.hpp:426-504 has no limit): above 2048 the
    selected lists live in global scratch (rank_rows_kernel<..., BIG>).  Distinct scores, so
    the order is fully determined; all items, one global list and per-row lists."""
    rng = np.random.default_rng(5)
    rows, I = 5, 6000
    score = rng.permutation(rows * I).reshape(rows, I).astype(dtype)  # distinct, exact in float32
    score[1, ::7] = -np.inf
    for cutoff in (2049, 3000, 5000, I, I + 10):
        assert_same(retrieve_recommend_from_score(score, [], cutoff, 1), restated(score, [], cutoff))
    lst = [sorted(rng.choice(I, size=4500, replace=False).tolist())]
    assert_same(retrieve_recommend_from_score(score, lst, 4000, 1), restated(score, lst, 4000))
    per = [sorted(rng.choice(I, size=n, replace=False).tolist()) for n in (3000, 10, 5999, 2500, 0)]
    assert_same(retrieve_recommend_from_score(score, per, 2600, 2), restated(score, per, 2600))
