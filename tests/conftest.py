import os
import sys

import numpy as np
import pytest
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built() -> None:
    """A source-only checkout (the built .so files are git-ignored) builds the HIP library and
    the CPU oracle once, with the same recipe as ``__graft_entry__.build()``; nothing happens
    when they are already there."""
    lib = os.path.join(ROOT, "irspack_amd", "libirspack_amd.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if os.path.exists(lib) and os.path.exists(orc):
        return
    import subprocess

    try:
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "irspack_amd", "csrc"), "-j4"])
        if not os.path.exists(orc):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    except Exception as exc:  # the tests that need the libraries then fail loudly themselves
        print(f"conftest: build failed: {exc}", file=sys.stderr)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def _gpu_available() -> bool:
    try:
        from irspack_amd import _lib

        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently;
    # plain runs (no -m) on a CPU box skip the gpu tests.
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture()
def X_small() -> sps.csr_matrix:
    # the reference's fixture, tests/conftest.py:9-16 (4 x 5 with an empty row)
    return sps.csr_matrix(
        np.asarray(
            [[1, 1, 2, 3, 4], [0, 1, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]],
            dtype=float,
        )
    )


def random_csr(n_rows, n_cols, density, seed, dtype=np.float32, binary=False, empty_rows=()):
    rng = np.random.default_rng(seed)
    M = sps.random(n_rows, n_cols, density=density, format="csr", random_state=rng, dtype=np.float64)
    if binary:
        M.data[:] = 1.0
    else:
        M.data = rng.uniform(0.5, 3.0, size=M.nnz)
    M = M.tolil()
    for r in empty_rows:
        M.rows[r] = []
        M.data[r] = []
    M = M.tocsr().astype(dtype)
    M.sort_indices()
    return M


def row_rel_err(a, b, floor: float = 1e-6) -> float:
    """Worst row of ``||a_r - b_r||_2 / ||b_r||_2`` — the reading of north_star's "1e-4 relative
    on factor matrices / scores" that a small row cannot hide behind a large one.

    ``floor``: a row whose norm is below ``floor * max_r ||b_r||`` is measured against that
    floor instead of its own norm.  Such rows are rounding noise of a mathematically zero row
    (iALS++ leaves 1e-7 .. 1e-15 in an empty row where the exact answer is 0; the oracle's
    noise and the GPU's are unrelated).  A row that is exactly zero in ``b`` must be exactly
    zero in ``a`` when ``floor == 0``."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    num = np.linalg.norm(a2 - b2, axis=1)
    den = np.maximum(np.linalg.norm(b2, axis=1), floor * np.linalg.norm(b2, axis=1).max())
    err = np.where(den > 0, num / np.where(den > 0, den, 1.0), np.where(num > 0, np.inf, 0.0))
    return float(err.max())


PARITY_LOG = os.environ.get("IRSPACK_AMD_PARITY_LOG",
                            os.path.join(ROOT, "gpurun_out", "parity_gpu.jsonl"))


def record_parity(test: str, config: str, **fields) -> None:
    """Appends one JSON line per parity comparison - the ACHIEVED errors, not just pass / fail -
    to ``gpurun_out/parity_gpu.jsonl`` (``IRSPACK_AMD_PARITY_LOG`` overrides), which a GPU run
    leaves behind; ``scripts/collect_parity.py`` folds it into the tracked
    ``profiles/parity_rNN.json``.  Keys used by the factor tests: ``n_rows``,
    ``worst_row_err`` (max over rows of ||gpu_r - oracle_r|| / ||oracle_r||),
    ``n_rows_over_1e-4``, ``worst_vs_float64`` (of the arbitrated rows: GPU against the float64
    evaluation; null when no row needed it) and ``oracle_vs_float64`` beside it."""
    import json

    rec = {"test": test, "config": config}
    for k, v in fields.items():
        if isinstance(v, (np.floating, np.integer)):
            v = v.item()
        rec[k] = v
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError as exc:  # a read-only checkout must not fail the parity test itself
        print(f"record_parity: {exc}", file=sys.stderr)
    print("PARITY", json.dumps(rec))


def rows_vs_float64(gpu, oracle32, ref64, floor: float = 1e-6):
    """Per-row distances of the GPU rows and of the float32 oracle's rows from the float64
    evaluation of the same algorithm (``oracle.ials_solver_step_f64``): ``||a_r - ref_r|| /
    max(||ref_r||, floor * max ||ref||)`` for both."""
    ref = np.asarray(ref64, dtype=np.float64)
    den = np.linalg.norm(ref, axis=1)
    den = np.maximum(den, floor * max(float(den.max()) if den.size else 0.0, 1e-300))
    e_gpu = np.linalg.norm(np.asarray(gpu, dtype=np.float64) - ref, axis=1) / den
    e_orc = np.linalg.norm(np.asarray(oracle32, dtype=np.float64) - ref, axis=1) / den
    return e_gpu, e_orc


def count_bar(n_oracle: int, n_rows: int) -> int:
    """How many rows of the GPU may lie beyond the tolerance when ``n_oracle`` rows of the float32 oracle
    do (truncated iterations: both counts sample the same exceedance rate): 0.01 % of the rows, or the
    oracle's count plus three standard deviations of a Poisson count of that size."""
    import math

    return max(math.ceil(1e-4 * n_rows), n_oracle + math.ceil(3.0 * math.sqrt(max(n_oracle, 1))))


def assert_float64_bar(gpu, oracle32, ref64, what: str, test: str = "", rtol: float = 1e-4,
                       truncated: bool = False, **extra) -> None:
    """The factor-parity bar with float64 as the arbiter of EVERY row.  The achieved distributions
    go to the parity log.

    ``truncated=False`` (Cholesky; anything whose exact answer is the solution of the row's system):
    the GPU's worst row is no farther from float64 than ``rtol`` - or, where the float32 oracle itself
    is farther (ill-conditioned rows), than the oracle's worst row.  No slack factor.

    ``truncated=True`` (CG after a fixed number of steps, one iALS++ sweep: the iteration has NOT
    converged on every row, and such a row amplifies any float32 rounding by its conditioning - the
    oracle's sequential sums as much as the GPU's tree sums, on different rows, run to run): a
    comparison of two maxima is a comparison of two extreme-value samples, so the bar is stated on the
    distribution, as COUNTS at two thresholds: the number of GPU rows beyond rtol, and the number beyond
    10 rtol (the heavy tail), are each <= max(0.01 % of the rows, the oracle's count + three standard
    deviations of a Poisson count of that size - two implementations with the SAME exceedance rate differ
    by that much, ``count_bar``); and no single row lies beyond 10 x max(rtol, the oracle's worst row)
    (an outright wrong row is O(1) away).  Quantiles are logged, not asserted: with 10^2 .. 10^5 rows the
    upper quantiles of two float32 evaluations are single rows again."""
    e_gpu, e_orc = rows_vs_float64(gpu, oracle32, ref64)
    g32 = np.linalg.norm(np.asarray(gpu, np.float64) - np.asarray(oracle32, np.float64), axis=1)
    d32 = np.linalg.norm(np.asarray(oracle32, np.float64), axis=1)
    e32 = g32 / np.maximum(d32, 1e-6 * max(float(d32.max()) if d32.size else 0.0, 1e-300))
    q = lambda e, p: float(np.quantile(e, p)) if e.size else 0.0  # noqa: E731
    record_parity(test or "float64_bar", what, n_rows=int(e_gpu.size), truncated=bool(truncated),
                  gpu_vs_f64_worst=float(e_gpu.max()) if e_gpu.size else 0.0,
                  gpu_vs_f64_p9999=q(e_gpu, 0.9999), gpu_vs_f64_p999=q(e_gpu, 0.999), gpu_vs_f64_median=q(e_gpu, 0.5),
                  oracle_f32_vs_f64_worst=float(e_orc.max()) if e_orc.size else 0.0,
                  oracle_f32_vs_f64_p9999=q(e_orc, 0.9999), oracle_f32_vs_f64_p999=q(e_orc, 0.999),
                  oracle_f32_vs_f64_median=q(e_orc, 0.5),
                  gpu_vs_oracle_f32_worst=float(e32.max()) if e32.size else 0.0,
                  n_rows_gpu_over_1e_4_vs_f64=int((e_gpu >= rtol).sum()),
                  n_rows_oracle_over_1e_4_vs_f64=int((e_orc >= rtol).sum()), **extra)
    if not e_gpu.size:
        return
    assert np.isfinite(e_gpu).all(), what
    if not truncated:
        assert e_gpu.max() <= max(rtol, e_orc.max()), (what, float(e_gpu.max()), float(e_orc.max()))
        return
    for thr in (rtol, 10.0 * rtol):
        assert int((e_gpu >= thr).sum()) <= count_bar(int((e_orc >= thr).sum()), e_gpu.size), \
            (what, thr, int((e_gpu >= thr).sum()), int((e_orc >= thr).sum()))
    assert e_gpu.max() <= 10.0 * max(rtol, e_orc.max()), (what, float(e_gpu.max()), float(e_orc.max()))
