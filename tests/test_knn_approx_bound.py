"""The kNN count path selects on float32 approximations of the similarities and keeps every
column whose approximation is within 2^-17 of the threshold (csrc/knn.hip, approx_epilogue /
FAST).  That is exact as long as |approx - exact| <= eps * exact with 2 eps well below 2^-17.
This test restates approx_epilogue in numpy float32 (every operation rounded to float32, the
reciprocal off by one unit in the last place in the unfavourable direction) next to the float64
epilogue of similarities.hpp:20-159 over random and adversarial inputs inside the parameter
ranges the host admits to that path (shrinkage in [0, 1e30), Tversky weights in [0, 16] and - for
Tversky - counts below 2^24: with larger ones the differences norms(j) - v are no longer exact in
float32 and the error reaches 2.1e-6, which is why the host sends those to the exact kernel), and
checks eps < 2^-20."""
import numpy as np
import pytest

f32 = np.float32
BOUND = 2.0 ** -20


def exact(kind, v, nj, ts, shrink, alpha, beta):
    v, nj, ts = v.astype(np.float64), nj.astype(np.float64), ts.astype(np.float64)
    if kind == "cosine":
        d = nj * ts + shrink + 1e-6
    elif kind == "jaccard":
        d = nj + ts - v + shrink + 1e-6
    else:
        d = v + beta * (nj - v) + alpha * (ts - v) + shrink + 1e-6
    return v / d


def approx(kind, v, nj, ts, shrink, alpha, beta, rcp_ulp):
    v, nj, ts = v.astype(f32), nj.astype(f32), ts.astype(f32)  # (float64 norms rounded once)
    shrink, alpha, beta = f32(shrink), f32(alpha), f32(beta)
    if kind == "cosine":
        d = nj * ts
    elif kind == "jaccard":
        d = (nj + ts) - v
    else:
        d = (v + beta * (nj - v)) + alpha * (ts - v)
    d = (d + shrink) + f32(1e-6)
    r = (f32(1.0) / d).astype(f32)
    r = np.nextafter(r, f32(np.inf) if rcp_ulp > 0 else f32(-np.inf)).astype(f32)  # v_rcp_f32: 1 ulp
    return (v * r).astype(f32)


@pytest.mark.parametrize("kind", ["cosine", "jaccard", "tversky"])
@pytest.mark.parametrize("rcp_ulp", [1, -1])
def test_float32_approximation_stays_within_the_candidate_margin(kind, rcp_ulp):
    rng = np.random.default_rng(7)
    n = 400_000
    worst = 0.0
    scales = (1e1, 1e3, 1e5, 2.0 ** 24 - 1) if kind == "tversky" else (1e1, 1e3, 1e5, 2.0 ** 24, 2.0 ** 30)
    for scale in scales:
        a = np.floor(rng.uniform(1, scale, n))
        b = np.floor(rng.uniform(1, scale, n))
        v = np.floor(rng.uniform(1, np.minimum(a, b) + 1))          # a count: 1 <= v <= min(norms)
        v = np.minimum(v, np.minimum(a, b))
        if kind == "cosine":                                        # norms are square roots (or powers) of counts
            nj, ts = np.sqrt(a), np.sqrt(b)
        else:
            nj, ts = a, b
        for shrink in (0.0, 0.5, 1e4, 1e29):
            for alpha, beta in ((0.0, 0.0), (0.5, 2.0), (16.0, 16.0), (1e-3, 7.0)):
                e = exact(kind, v, nj, ts, shrink, alpha, beta)
                g = approx(kind, v, nj, ts, shrink, alpha, beta, rcp_ulp).astype(np.float64)
                ok = e >= 1.2e-38  # (smaller values leave the path through the device-side net)
                rel = np.abs(g[ok] - e[ok]) / e[ok]
                worst = max(worst, float(rel.max()))
    # adversarial: v == norm (the subtractions cancel completely) and v == 1 next to huge norms
    big = np.array([1.0, 3.0, 2.0 ** 24 - 1] + ([] if kind == "tversky" else [2.0 ** 24 + 1, 2.0 ** 31 - 1]))
    for x in big:
        for y in big:
            v = np.array([1.0, min(x, y)])
            nj = np.array([x, x]) if kind != "cosine" else np.sqrt(np.array([x, x]))
            ts = np.array([y, y]) if kind != "cosine" else np.sqrt(np.array([y, y]))
            e = exact(kind, v, nj, ts, 0.0, 16.0, 16.0)
            g = approx(kind, v, nj, ts, 0.0, 16.0, 16.0, rcp_ulp).astype(np.float64)
            worst = max(worst, float((np.abs(g - e) / e).max()))
    assert worst < BOUND, worst
