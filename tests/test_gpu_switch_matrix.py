"""ONE parametrised matrix for the kernel-selection switches (``IRSPACK_AMD_*``, read when a trainer /
computer / call is made; DESIGN.md section 7 lists every one with its status).

Round 4's review: "42 switches select kernel variants ... each is tested once, none in combination".
Here every iALS switch that selects a kernel family is flipped alone AND in pairs (the pairs are the
combinations that can meet inside one launch: a fallback of one family landing on the fallback of
another), across the K classes that route differently (<= 64 one wave per row, 128 the wave-128 kernel,
200 a workgroup per row / matrix-free CG; the general-size kernels above 256 read no switch), for the
three solvers, on a
matrix with split rows, short rows and empty rows.  Bar: the float64 arbiter of conftest
(``assert_float64_bar``), every row.  kNN and evaluator switches: singles and pairs against the oracle,
indices / counters bit-exact.
"""
import itertools

import numpy as np
import pytest
import scipy.sparse as sps

import _operating_point as OP
import oracle as O
from conftest import assert_float64_bar
from irspack_amd.recommenders._ials_core import IALSTrainer

pytestmark = pytest.mark.gpu

IALS_SWITCHES = {
    "wave128_off": {"IRSPACK_AMD_IALS_WAVE128": "0"},
    "unit_off": {"IRSPACK_AMD_IALS_UNIT": "0"},
    "short_off": {"IRSPACK_AMD_IALS_SHORT": "0"},
    "short2_off": {"IRSPACK_AMD_IALS_SHORT2": "0"},
    "wg16_off": {"IRSPACK_AMD_IALS_WG16": "0"},
    "eig_off": {"IRSPACK_AMD_IALS_EIG": "0"},
    "mf_off": {"IRSPACK_AMD_IALS_MF": "0"},
    "pp_direct_off": {"IRSPACK_AMD_IALSPP_DIRECT": "0"},
    "pp_chain_off": {"IRSPACK_AMD_IALSPP_CHAIN": "0"},
    "pp_fork_off": {"IRSPACK_AMD_IALSPP_FORK": "0"},
    "bf16x3_off": {"IRSPACK_AMD_IALS_BF16X3": "0"},
    "chunk_small": {"IRSPACK_AMD_IALS_CHUNK": "256", "IRSPACK_AMD_IALS_CHUNK_LONG": "512"},
}
IALS_PAIRS = [("unit_off", "short_off"), ("wave128_off", "wg16_off"), ("eig_off", "short_off"),
              ("mf_off", "wg16_off"), ("unit_off", "chunk_small"), ("pp_direct_off", "pp_chain_off"),
              ("pp_chain_off", "pp_fork_off"), ("short2_off", "eig_off"), ("unit_off", "bf16x3_off"),
              ("mf_off", "wave128_off")]
IALS_CASES = [("default",)] + [(k,) for k in IALS_SWITCHES] + IALS_PAIRS


@pytest.fixture(scope="module")
def matrices():
    """binary and weighted, 3000 x 900: rows of 0 .. 1500 entries (split above the lowered chunk size),
    a third of the rows with at most 32 entries"""
    rng = np.random.default_rng(12)
    n_u, n_i = 3000, 900
    deg = np.clip(np.round(rng.lognormal(3.2, 1.3, n_u)), 0, n_i).astype(int)
    deg[:5] = [0, 1, 2, 880, 600]
    rows = np.repeat(np.arange(n_u), deg)
    cols = np.concatenate([rng.choice(n_i, size=d, replace=False) for d in deg]) if deg.sum() else np.zeros(0, int)
    out = {}
    for name, data in (("binary", np.ones(rows.size, np.float32)),
                       ("weighted", rng.uniform(0.5, 3.0, rows.size).astype(np.float32))):
        X = sps.csr_matrix((data, (rows, cols)), shape=(n_u, n_i))
        X.sort_indices()
        out[name] = (X, OP.transpose_csr(X))
    return out


@pytest.mark.parametrize("case", IALS_CASES, ids=lambda c: "+".join(c))
@pytest.mark.parametrize("K,kind,data", [(64, "CHOLESKY", "binary"), (64, "CG", "weighted"), (48, "IALSPP", "binary"),
                                         (128, "CHOLESKY", "weighted"), (128, "CHOLESKY", "binary"), (128, "CG", "binary"),
                                         (128, "IALSPP", "weighted"),
                                         (200, "CHOLESKY", "binary"), (200, "CG", "weighted")])
def test_ials_switches_alone_and_in_pairs(matrices, monkeypatch, case, K, kind, data):
    for name in case:
        for key, value in IALS_SWITCHES.get(name, {}).items():
            monkeypatch.setenv(key, value)
    X, Xt = matrices[data]
    alpha0, reg = 0.1, 1e-2
    mc, sc, omc, osc = OP.configs(K, kind, alpha0, reg)
    t = IALSTrainer(mc, X)  # (the switches are read here)
    user0, item0 = t.user, t.item
    for side, (Xs, tgt0, oth0) in enumerate(((X, user0, item0), (Xt, item0, user0))):
        t.user, t.item = user0, item0
        OP.gpu_half_step(t, side, sc)
        got = t.user if side == 0 else t.item
        want = O.ials_solver_step(tgt0, Xs, oth0, O.ials_gramian(oth0, omc.alpha0, OP.CORES), omc, osc)
        ref64 = O.ials_solver_step_f64(tgt0, Xs, oth0, None, omc, osc, OP.CORES)
        assert_float64_bar(got, want, ref64, f"switch matrix {'+'.join(case)} K={K} {kind} {data} side {side}",
                           test="switch_matrix_ials", truncated=(kind != "CHOLESKY"))


# ---------------------------------------------------------------- kNN
KNN_SWITCHES = {
    "wide_on": {"IRSPACK_AMD_KNN_WIDE": "1"},
    "fast_off": {"IRSPACK_AMD_KNN_FAST": "0"},
    "threads_1": {"IRSPACK_AMD_KNN_THREADS": "1"},
    "chunks_3": {"IRSPACK_AMD_KNN_CHUNKS": "3"},
    "host_create": {"IRSPACK_AMD_KNN_DEVICE_CREATE": "0"},
    "stage_off": {"IRSPACK_AMD_KNN_STAGE": "0"},
}
KNN_CASES = ([("default",)] + [(k,) for k in KNN_SWITCHES]
             + [p for p in itertools.combinations(["wide_on", "fast_off"], 2)]
             + [("chunks_3", "fast_off"), ("chunks_3", "stage_off"), ("host_create", "wide_on")])


@pytest.mark.parametrize("case", KNN_CASES, ids=lambda c: "+".join(c))
@pytest.mark.parametrize("kind,binary", [("cosine", True), ("jaccard", True), ("cosine", False)])
def test_knn_switches_alone_and_in_pairs(monkeypatch, case, kind, binary):
    from irspack_amd.recommenders._knn import CosineSimilarityComputer, JaccardSimilarityComputer

    for name in case:
        for key, value in KNN_SWITCHES.get(name, {}).items():
            monkeypatch.setenv(key, value)
    rng = np.random.default_rng(5)
    X = sps.random(700, 1200, density=0.03, format="csr", random_state=rng, dtype=np.float64)
    X.data = np.ones_like(X.data) if binary else np.round(rng.uniform(0.5, 3.0, X.nnz), 3)
    X.sort_indices()
    Xt = sps.csr_matrix(X.T)
    Xt.sort_indices()
    if kind == "cosine":
        got = CosineSimilarityComputer(Xt, 0.5, True).compute_similarity(Xt, 20)
        want = O.KNNComputer("cosine", Xt, 0.5, normalize=True, n_threads=2).compute_similarity(Xt, 20)
    else:
        got = JaccardSimilarityComputer(Xt, 0.5).compute_similarity(Xt, 20)
        want = O.KNNComputer("jaccard", Xt, 0.5, n_threads=2).compute_similarity(Xt, 20)
    got.sort_indices()
    want.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices), case
    np.testing.assert_allclose(got.data, want.data, rtol=1e-12, atol=0)


# ---------------------------------------------------------------- evaluator (fused iALS call)
EVAL_SWITCHES = {
    "emit_off": {"IRSPACK_AMD_EVAL_EMIT": "0"},
    "bound_off": {"IRSPACK_AMD_EVAL_BOUND": "0"},
    "wave_off": {"IRSPACK_AMD_EVAL_WAVE": "0"},
    "sample_64": {"IRSPACK_AMD_EVAL_SAMPLE": "64"},
    "sample_unfused": {"IRSPACK_AMD_EVAL_SAMPLE_FUSED": "0"},
    "pass_rows_256": {"IRSPACK_AMD_EVAL_PASS_ROWS": "256"},
    "block_512": {"IRSPACK_AMD_EVAL_BLOCK": "512"},
}
EVAL_CASES = ([("default",)] + [(k,) for k in EVAL_SWITCHES]
              + [("emit_off", "block_512"), ("bound_off", "sample_64"),
                 ("bound_off", "pass_rows_256"), ("wave_off", "emit_off"), ("sample_64", "pass_rows_256")])


@pytest.mark.parametrize("case", EVAL_CASES, ids=lambda c: "+".join(c))
@pytest.mark.parametrize("K", [64, 200])
def test_evaluator_switches_alone_and_in_pairs(monkeypatch, case, K):
    from irspack_amd.evaluation._core_evaluator import EvaluatorCore
    from irspack_amd.synthetic import holdout_split, make_interactions

    for name in case:
        for key, value in EVAL_SWITCHES.get(name, {}).items():
            monkeypatch.setenv(key, value)
    X = make_interactions("small")
    train, test = holdout_split(X, 0.2, seed=4)
    mc, sc, _, _ = OP.configs(K, "CG", 0.1, 1e-2)
    t = IALSTrainer(mc, train)
    for _ in range(2):
        t.step(sc)
    gt = sps.csr_matrix(test, dtype=np.float64)
    got = EvaluatorCore(gt, []).get_metrics_ials(t, 0, X.shape[0], sps.csr_matrix(train, dtype=np.float32), 20, 0, False)
    scores = t.user_scores(0, X.shape[0], sc).astype(np.float32)
    scores[train.nonzero()] = -np.inf
    want = O.EvaluatorCore(gt, []).get_metrics_f32(scores, 20, 0, OP.CORES)
    raw = want.raw()
    assert np.array_equal(got.item_cnt, want.item_cnt()), case
    assert got.valid_user == int(raw[0]) and got.total_user == int(raw[1])
    assert abs(got.ndcg - raw[4]) <= 1e-12 * max(1.0, abs(raw[4]))
