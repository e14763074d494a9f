"""Generates the golden fixtures in this directory (run here, in the build container).

The reference can be neither compiled nor imported offline, so the expected
values are produced the way the reference's OWN tests produce theirs: float64
closed forms in numpy / sklearn on the inputs those tests use
(/root/reference/tests/recommenders/test_ials.py, test_knn.py,
tests/evaluation/test_evaluator.py).  The one exception is the factor-init
stream, which is restated here in Python from libstdc++'s mt19937 +
normal_distribution<float> (the classes IALSTrainer.hpp:64-76 calls), independently of the
oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

X_SMALL = np.asarray([[1, 1, 2, 3, 4], [0, 1, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]], dtype=float)


def libstdcxx_normal_stream(seed, n, stddev_f32):
    """Independent restatement of libstdc++'s ``std::mt19937(seed)`` +
    ``std::normal_distribution<float>(0, stddev)`` (bits/random.tcc: Marsaglia polar method on
    ``generate_canonical<float, 24>``, the second variate saved), the classes
    IALSTrainer.hpp:64-76 calls.  Raw 32-bit words come from numpy's legacy MT19937
    (``RandomState(seed)`` seeds with init_genrand like std::mt19937); the float arithmetic is
    done step by step in float32 with libm's logf / sqrtf (what libstdc++ calls)."""
    import ctypes

    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype = libm.sqrtf.restype = ctypes.c_float
    libm.logf.argtypes = libm.sqrtf.argtypes = [ctypes.c_float]
    f = np.float32
    rs = np.random.RandomState(seed)
    words = iter(())

    def canonical():
        nonlocal words
        try:
            w = next(words)
        except StopIteration:
            words = iter(np.frombuffer(rs.bytes(4 * 4096), dtype="<u4").tolist())
            w = next(words)
        r = f(w) / f(4294967296.0)  # (urng() - min) / 2^32 in float
        return np.nextafter(f(1), f(0)) if r >= f(1) else r

    out = np.empty(n, dtype=np.float32)
    saved = None
    for i in range(n):
        if saved is not None:
            ret, saved = saved, None
        else:
            while True:
                x = f(f(2) * canonical() - f(1))
                y = f(f(2) * canonical() - f(1))
                r2 = f(f(x * x) + f(y * y))
                if not (r2 > f(1) or r2 == f(0)):
                    break
            mult = f(libm.sqrtf(f(f(f(-2) * f(libm.logf(r2))) / r2)))
            saved = f(x * mult)
            ret = f(y * mult)
        out[i] = f(f(ret * stddev_f32) + f(0))
    return out


def golden_init():
    """Initial factors, hpp:64-76.  The stddev is init_stdev / std::sqrt(factor.cols()) with the
    INTEGRAL sqrt overload, i.e. a double quotient rounded to float once; K = 10 / 40 are
    dimensions where a float quotient would differ by one ulp."""
    out = {}
    for K in (10, 16, 40, 64):
        sd = np.float32(np.float64(np.float32(0.1)) / np.sqrt(np.float64(K)))
        out[f"K{K}"] = libstdcxx_normal_stream(42, 8 * K, sd).reshape(8, K)
    assert np.float32(np.float32(0.1) / np.sqrt(np.float32(40))) != np.float32(
        np.float64(np.float32(0.1)) / np.sqrt(40.0))
    np.savez(os.path.join(HERE, "ials_init_seed42.npz"), **out)


def golden_ials_halfstep():
    rng = np.random.default_rng(20260101)
    U, I, K = 48, 37, 16
    X = sps.random(U, I, density=0.18, format="csr", random_state=7, dtype=np.float64)
    X.data = rng.uniform(0.5, 3.0, X.nnz)
    X.sort_indices()
    # make one empty row and one dense row
    X = X.tolil()
    X.rows[5], X.data[5] = [], []
    X.rows[9], X.data[9] = list(range(I)), list(rng.uniform(0.5, 3.0, I))
    X = X.tocsr()
    item = (rng.standard_normal((I, K)) * 0.3).astype(np.float32)
    alpha0, reg, nu = 0.3, 0.05, 0.5
    out = dict(indptr=X.indptr.astype(np.int64), indices=X.indices.astype(np.int32),
               data=X.data.astype(np.float32), item=item, alpha0=alpha0, reg=reg, nu=nu)
    V64 = item.astype(np.float64)
    P = alpha0 * V64.T @ V64
    for loss, bias in (("IALSPP", 0.0), ("ORIGINAL", alpha0)):
        exp = np.zeros((U, K))
        for r in range(U):
            sl = slice(X.indptr[r], X.indptr[r + 1])
            Vr, c = V64[X.indices[sl]], X.data[sl].astype(np.float32).astype(np.float64)
            regr = np.float32(reg) * np.float32(np.float32(alpha0) * I + (sl.stop - sl.start)) ** np.float32(nu)
            A = P + (Vr * c[:, None]).T @ Vr + float(regr) * np.eye(K)
            b = ((c + bias)[:, None] * Vr).sum(axis=0)
            exp[r] = np.linalg.solve(A, b)
        out[f"user_{loss}"] = exp
    np.savez(os.path.join(HERE, "ials_halfstep.npz"), **out)


def golden_knn():
    out = {}
    rng = np.random.RandomState(0)
    mats = {"small": X_SMALL, "many": (rng.rand(200, 96) > 0.9).astype(float), "dense": rng.rand(40, 33)}
    for name, Xd in mats.items():
        out[f"X_{name}"] = Xd
        m = Xd.T.copy()  # I x U
        norm = (m ** 2).sum(axis=1) ** 0.5
        raw = m @ m.T
        cos = raw / (norm[:, None] * norm[None, :] + 1e-6)  # test_knn.py:33-51
        mb = (Xd.T != 0).astype(float)
        nb = mb.sum(axis=1)
        inter = mb @ mb.T
        jac = inter / (nb[:, None] + nb[None, :] - inter + 1e-6)  # test_knn.py:54-70
        jac[inter == 0] = 0
        for k, v in (("cos_raw", raw), ("cos_norm", cos), ("jaccard", jac)):
            v = v.copy()
            np.fill_diagonal(v, 0)
            out[f"{k}_{name}"] = v
    # deterministic tie-break known answer, test_knn.py:144-165
    T = np.asarray([[1, 1, 1, 0, 0], [1, 1, 0, 1, 0], [1, 0, 1, 1, 0], [0, 1, 1, 1, 1]], dtype=float)
    full = T.T @ T
    exp = np.zeros_like(full)
    for row, scores in enumerate(full):
        cand = np.flatnonzero(scores)
        order = np.lexsort((cand, -scores[cand]))
        sel = cand[order[:2]]
        exp[row, sel] = scores[sel]
    out["tie_X"], out["tie_top2"] = T, exp
    np.savez(os.path.join(HERE, "knn_dense.npz"), **out)


def golden_evaluator():
    from sklearn.metrics import average_precision_score, ndcg_score

    out = {}
    # test_evaluator.py:19-49
    for tag, (U, I, dtype) in {"a": (10, 5, "float32"), "b": (10, 30, "float64"), "c": (300, 5, "float32")}.items():
        rns = np.random.RandomState(42)
        scores = rns.randn(U, I).astype(dtype)
        gt = (rns.rand(U, I) >= 0.7).astype(np.float64)
        maps, ndcgs = [], []
        for i in range(U):
            if gt[i].sum() == 0:
                continue
            maps.append(average_precision_score(gt[i], scores[i]))
            ndcgs.append(ndcg_score(gt[i][None, :], scores[i][None, :]))
        out[f"full_{tag}_scores"], out[f"full_{tag}_gt"] = scores, gt
        out[f"full_{tag}_expected"] = np.asarray([np.mean(maps), np.mean(ndcgs)])
    # test_evaluator.py:87-152
    for tag, (U, I, C) in {"a": (10, 5, 5), "b": (10, 30, 29)}.items():
        rns = np.random.RandomState(42)
        scores = rns.randn(U, I)
        gt = (rns.rand(U, I) >= 0.3).astype(np.float64)
        ndcg = mapv = prec = rec = 0.0
        valid = 0
        cnt = np.zeros(I)
        for i in range(U):
            nzs = set(gt[i].nonzero()[0])
            if not nzs:
                continue
            valid += 1
            ndcg += ndcg_score(gt[[i]], scores[[i]], k=C)
            recommended = scores[i].argsort()[::-1][:C]
            denom = min(C, len(nzs))
            ap, hit = 0.0, 0
            for j, r in enumerate(recommended):
                cnt[r] += 1
                if r in nzs:
                    hit += 1
                    ap += hit / float(j + 1)
            mapv += ap / denom
            rec += hit / denom
            prec += hit / C
        p = cnt / cnt.sum()
        entropy = -(p[p > 0] * np.log(p[p > 0])).sum()
        lorentz = np.cumsum(np.sort(cnt) / cnt.sum())
        gini = sum((1 / I) * 2 * (((i + 1) / I) - lorentz[i]) for i in range(I))
        out[f"cut_{tag}_scores"], out[f"cut_{tag}_gt"], out[f"cut_{tag}_C"] = scores, gt, C
        out[f"cut_{tag}_expected"] = np.asarray(
            [ndcg / valid, mapv / valid, prec / valid, rec / valid, entropy, gini])
    np.savez(os.path.join(HERE, "evaluator_rs42.npz"), **out)


if __name__ == "__main__":
    golden_init()
    golden_ials_halfstep()
    golden_knn()
    golden_evaluator()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
