"""Seeded synthetic interaction matrices of the benchmark shapes.

No dataset can be downloaded, so the MovieLens-shaped inputs are generated from
public summary statistics: log-normal user degrees, power-law item popularity,
all values 1.0 (float32), sorted unique int32 column indices.
"""

from typing import Dict, Tuple

import numpy as np
import scipy.sparse as sps

SHAPES: Dict[str, Dict] = {
    # C1  MovieLens-100K shape
    "ml100k": dict(n_users=943, n_items=1682, nnz=100_000, seed=100, mu=4.3, sigma=0.9,
                   dmin=20, dmax=737, pop_exp=0.8),
    # C2 / C3 / C5  MovieLens-20M shape
    "ml20m": dict(n_users=138_493, n_items=26_744, nnz=20_000_263, seed=2000, mu=4.25,
                  sigma=1.05, dmin=20, dmax=9_254, pop_exp=1.0),
    # reduced shapes for tests
    "tiny": dict(n_users=300, n_items=200, nnz=6_000, seed=7, mu=2.7, sigma=0.8, dmin=1,
                 dmax=150, pop_exp=0.8),
    "small": dict(n_users=4_000, n_items=1_500, nnz=200_000, seed=11, mu=3.6, sigma=1.0,
                  dmin=2, dmax=1_200, pop_exp=1.0),
    # C4  BASELINE configs[3]: 10 M users x 1 M items, ~1e8 stored entries, short rows
    # (degree = 1 + geometric(p = 1/9) clipped at 2,000; Zipf(1.0) items); SURVEY.md 8(d)
    "c4": dict(kind="geometric", n_users=10_000_000, n_items=1_000_000, p=1.0 / 9, dmax=2_000,
               seed=4000, pop_exp=1.0),
    # the same generator at 1/50 of the size (parity tests) and at 1/5 (a bench leg that fits
    # any host)
    "c4_small": dict(kind="geometric", n_users=200_000, n_items=20_000, p=1.0 / 9, dmax=2_000,
                     seed=4001, pop_exp=1.0),
    "c4_fifth": dict(kind="geometric", n_users=2_000_000, n_items=200_000, p=1.0 / 9, dmax=2_000,
                     seed=4002, pop_exp=1.0),
}


def _degrees(rng: np.random.Generator, n: int, target_nnz: int, mu: float, sigma: float,
             dmin: int, dmax: int) -> np.ndarray:
    raw = rng.lognormal(mean=mu, sigma=sigma, size=n)
    deg = np.clip(np.round(raw), dmin, dmax)
    # rescale towards the target mean, keeping the clip
    for _ in range(8):
        scale = target_nnz / max(deg.sum(), 1.0)
        if abs(scale - 1.0) < 1e-3:
            break
        deg = np.clip(np.round(deg * scale), dmin, dmax)
    return deg.astype(np.int64)


def make_interactions(name: str = "ml20m", dtype=np.float32) -> sps.csr_matrix:
    """CSR [n_users, n_items] for one of SHAPES (realised nnz is close to the target)."""
    cfg = SHAPES[name]
    if cfg.get("kind") == "geometric":
        return _make_geometric(cfg, dtype)
    rng = np.random.default_rng(cfg["seed"])
    U, I = cfg["n_users"], cfg["n_items"]
    deg = _degrees(rng, U, cfg["nnz"], cfg["mu"], cfg["sigma"], cfg["dmin"], min(cfg["dmax"], I))
    w = (np.arange(I, dtype=np.float64) + 1.0) ** (-cfg["pop_exp"])
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(I)  # popularity rank -> item id
    total = int(deg.sum())
    # Draw with replacement by popularity, dedupe per row, top up heavy rows uniformly.
    draws = np.searchsorted(cdf, rng.random(int(total * 1.6) + 16), side="right")
    draws = np.minimum(draws, I - 1)
    over = np.ceil(deg * 1.6).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(over)])
    indptr = np.zeros(U + 1, dtype=np.int64)
    cols = []
    for u in range(U):
        d = int(deg[u])
        cand = np.unique(draws[starts[u]:starts[u + 1]])
        if cand.size > d:
            cand = rng.choice(cand, size=d, replace=False)
        elif cand.size < d:
            # heavy users exhaust the head of the distribution: fill from unseen items
            mask = np.ones(I, dtype=bool)
            mask[cand] = False
            rest = np.flatnonzero(mask)
            extra = rng.choice(rest, size=min(d - cand.size, rest.size), replace=False)
            cand = np.concatenate([cand, extra])
        c = np.sort(perm[cand])
        cols.append(c.astype(np.int32))
        indptr[u + 1] = indptr[u] + c.size
    indices = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int32)
    data = np.ones(indices.size, dtype=dtype)
    X = sps.csr_matrix((data, indices, indptr), shape=(U, I))
    X.has_sorted_indices = True
    return X


def _make_geometric(cfg: Dict, dtype) -> sps.csr_matrix:
    """Vectorised generator of the short-row shapes: draws with replacement by popularity,
    duplicates inside a row removed (the realised nnz is a little below the draw count)."""
    rng = np.random.default_rng(cfg["seed"])
    U, I = cfg["n_users"], cfg["n_items"]
    deg = np.minimum(1 + rng.geometric(cfg["p"], size=U), min(cfg["dmax"], I)).astype(np.int64)
    rows = np.repeat(np.arange(U, dtype=np.int64), deg)
    w = (np.arange(I, dtype=np.float64) + 1.0) ** (-cfg["pop_exp"])
    cdf = np.cumsum(w / w.sum())
    cols = np.minimum(np.searchsorted(cdf, rng.random(rows.shape[0]), side="right"), I - 1)
    cols = rng.permutation(I)[cols]  # popularity rank -> item id
    key = np.unique(rows * I + cols)  # sorted by (row, column)
    del rows, cols
    r = key // I
    indices = (key - r * I).astype(np.int32)
    del key
    indptr = np.zeros(U + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=U), out=indptr[1:])
    del r
    X = sps.csr_matrix((np.ones(indices.shape[0], dtype=dtype), indices, indptr), shape=(U, I))
    X.has_sorted_indices = True
    return X


def describe(X: sps.csr_matrix) -> Dict[str, float]:
    X = X.tocsr()
    rd = np.diff(X.indptr)
    cd = np.bincount(X.indices, minlength=X.shape[1])
    return dict(n_users=int(X.shape[0]), n_items=int(X.shape[1]), nnz=int(X.nnz),
                max_user_degree=int(rd.max(initial=0)), max_item_degree=int(cd.max(initial=0)),
                mean_user_degree=float(rd.mean()) if rd.size else 0.0)


def holdout_split(X: sps.csr_matrix, ratio: float = 0.2, seed: int = 7
                  ) -> Tuple[sps.csr_matrix, sps.csr_matrix]:
    """Per-row random hold-out (own splitter; the reference's split/ is out of scope)."""
    X = X.tocsr()
    rng = np.random.default_rng(seed)
    keep = rng.random(X.nnz) >= ratio
    rows = np.repeat(np.arange(X.shape[0]), np.diff(X.indptr))
    tr = sps.csr_matrix((X.data[keep], (rows[keep], X.indices[keep])), shape=X.shape)
    te = sps.csr_matrix((X.data[~keep], (rows[~keep], X.indices[~keep])), shape=X.shape)
    tr.sort_indices()
    te.sort_indices()
    return tr, te
