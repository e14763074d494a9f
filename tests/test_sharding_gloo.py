"""CPU tests of the multi-GPU host loop (irspack_amd/sharding.py): world_size-2 ``gloo``
processes run ``ShardedIALSTrainer`` with the CPU oracle standing in for the device
(tests may call the oracle; the product's ``HipLocalSolver`` is exercised on the GPU
box).  Checks the partition, the K x K all-reduce and the shard broadcasts: the
sharded run must reproduce the single-process result up to Gramian summation order.
"""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from irspack_amd.sharding import (LocalSolver, ShardedIALSTrainer, balanced_bounds,  # noqa: E402
                                  equal_bounds, equal_shard_bounds, even_bounds, row_cost,
                                  shard_bounds, sharded_metrics, sharded_similarity)


class OracleLocalSolver(LocalSolver):
    """LocalSolver backed by oracle/ (test double for HipLocalSolver)."""

    def __init__(self, omc, X, shard, K, pad=False):
        import oracle as O
        import torch

        self.O, self.torch = O, torch
        self.mc, self.K = omc, K
        self.X = [sps.csr_matrix(X, dtype=np.float32), sps.csr_matrix(X.T, dtype=np.float32)]
        for m in self.X:
            m.sort_indices()
        self.shard = shard
        U, I = X.shape
        init = [O.ials_init(U, K, omc.init_stdev, omc.random_seed),
                O.ials_init(I, K, omc.init_stdev, omc.random_seed)]
        # pad=True mirrors the device buffers: rows padded to a multiple of 8 (zeros)
        self._buf = [np.zeros((-(-n // 8) * 8 if pad else n, K), np.float32) for n in (U, I)]
        self.factor = [self._buf[0][:U], self._buf[1][:I]]
        for w in range(2):
            self.factor[w][...] = init[w]
        self.G = [np.zeros((K, K), np.float32), np.zeros((K, K), np.float32)]
        self.P = [np.zeros((K, K), np.float32), np.zeros((K, K), np.float32)]
        # torch views alias the numpy buffers, like the product's device views
        self._fv = [torch.from_numpy(f) for f in self._buf]
        self._gv = [torch.from_numpy(g) for g in self.G]

    def _range(self, which):
        return (self.shard[0], self.shard[1]) if which == 0 else (self.shard[2], self.shard[3])

    def factor_view(self, which):
        return self._fv[which]

    def gramian_view(self, side):
        return self._gv[side]

    def partial_gramian(self, side):
        b, e = self._range(1 - side)
        F = self.factor[1 - side][b:e]
        self.G[side][...] = self.O.ials_gramian(F, 1.0, 1) if e > b else 0.0

    def finish_gramian(self, side):
        self.P[side] = (np.float32(self.mc.alpha0) * self.G[side]).astype(np.float32)

    def half_step(self, side, sc):
        b, e = self._range(side)
        self.factor[side][...] = self.O.ials_solver_step(self.factor[side], self.X[side],
                                                         self.factor[1 - side], self.P[side],
                                                         self.mc, sc, b, e)

    def synchronize(self):
        pass


def _worker(rank, world, port, kind, out_dir, equal=False, overlap=True):
    import torch
    import torch.distributed as dist

    import oracle as O
    from conftest import random_csr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = random_csr(97, 61, 0.15, 11, empty_rows=(4,))
    K = 8
    omc = O.model_config(K, alpha0=0.2, reg=0.05)
    sc = O.solver_config(1, kind, 3)
    ub, ib = equal_shard_bounds(X, world) if equal else shard_bounds(X, K, kind, world)
    local = OracleLocalSolver(omc, X, (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1]), K,
                              pad=equal)
    # the default is the sequential exchange (overlap off until RCCL has run it at world >= 2)
    assert ShardedIALSTrainer(local, ub, ib).overlap is False
    tr = ShardedIALSTrainer(local, ub, ib, overlap=overlap, timing=True)
    for _ in range(2):
        tr.step(sc)
    # one collective per half-epoch either way: in place for equal blocks, through the
    # padded staging rows for the cost-balanced uneven shards
    assert tr.exchange == (["inplace", "inplace"] if equal else ["padded", "padded"])
    assert tr.overlap == overlap and (tr.gram_group is not None) == overlap
    assert tr._owns_gram_group == overlap
    tm = tr.last_timing()
    assert set(tm) == {"gramian_ms", "allreduce_ms", "solve_ms", "allgather_ms", "exposed_comm_ms", "total_ms"}
    assert tm["solve_ms"] > 0 and tm["allgather_ms"] > 0 and tm["total_ms"] >= tm["exposed_comm_ms"]
    tr.close()
    tr.close()  # idempotent
    assert not tr._owns_gram_group
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), user=local.factor[0], item=local.factor[1])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("kind", ["CHOLESKY", "CG"])
def test_world2_matches_single_process(tmp_path, kind, equal, overlap):
    import torch.multiprocessing as mp

    import oracle as O
    from conftest import random_csr

    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path), equal, overlap), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # replicas agree bit for bit (solved rows are broadcast; the reduced Gramian is shared)
    np.testing.assert_array_equal(r0["user"], r1["user"])
    np.testing.assert_array_equal(r0["item"], r1["item"])
    X = random_csr(97, 61, 0.15, 11, empty_rows=(4,))
    omc = O.model_config(8, alpha0=0.2, reg=0.05)
    ref = O.IALSTrainer(omc, X)
    for _ in range(2):
        ref.step(O.solver_config(1, kind, 3))
    np.testing.assert_allclose(r0["user"], ref.user, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r0["item"], ref.item, rtol=1e-4, atol=1e-6)


def test_equal_bounds():
    assert equal_bounds(97, 2) == [0, 52, 97]          # padded to 104 rows, blocks of 52
    assert equal_bounds(61, 4) == [0, 16, 32, 48, 61]  # padded to 64
    assert equal_bounds(5, 8) == [0, 1, 2, 3, 4, 5, 5, 5, 5]
    assert equal_bounds(16, 1) == [0, 16]
    with pytest.raises(ValueError):
        equal_bounds(10, 3)


def test_balanced_bounds_properties():
    rng = np.random.default_rng(0)
    nnz = rng.integers(0, 500, size=1000)
    for parts in (1, 2, 3, 8):
        c = row_cost(nnz, 64, "CHOLESKY")
        b = balanced_bounds(c, parts)
        assert b[0] == 0 and b[-1] == 1000 and len(b) == parts + 1
        assert all(b[i] <= b[i + 1] for i in range(parts))
        loads = [c[b[i]:b[i + 1]].sum() for i in range(parts)]
        assert max(loads) <= c.sum() / parts + c.max() + 1e-9
    # degenerate: more parts than rows
    assert balanced_bounds(np.ones(2), 4)[-1] == 2


def test_single_process_needs_no_group():
    import oracle as O
    from conftest import random_csr

    X = random_csr(30, 20, 0.2, 3)
    omc = O.model_config(4, alpha0=0.1, reg=0.1)
    local = OracleLocalSolver(omc, X, (0, 30, 0, 20), 4)
    tr = ShardedIALSTrainer(local, [0, 30], [0, 20])
    sc = O.solver_config(1, "CG", 3)
    tr.step(sc)
    ref = O.IALSTrainer(omc, X)
    ref.step(sc)
    np.testing.assert_allclose(local.factor[0], ref.user, rtol=1e-5, atol=1e-7)


def test_even_bounds():
    assert even_bounds(10, 3) == [0, 3, 6, 10]
    assert even_bounds(2, 4) == [0, 0, 1, 1, 2]
    assert even_bounds(7, 1) == [0, 7]


class _RawMetrics:
    """picklable stand-in with Metrics' merge contract (a plain sum of the raw terms)"""

    def __init__(self, n_items, raw=None, cnt=None):
        self.raw = np.zeros(7) if raw is None else raw
        self.cnt = np.zeros(n_items, np.int64) if cnt is None else cnt

    def merge(self, other):
        self.raw = self.raw + other.raw
        self.cnt = self.cnt + other.cnt


def _knn_eval_worker(rank, world, port, out_dir):
    """kNN target rows and evaluator users sharded over two ranks, the oracle standing in for
    the device computers (sharding.sharded_similarity / sharded_metrics are host logic)."""
    import pickle

    import torch.distributed as dist

    import oracle as O
    from conftest import random_csr

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = random_csr(83, 57, 0.2, 5, dtype=np.float64)
    Xt = sps.csr_matrix(X.T)
    comp = O.KNNComputer("cosine", Xt, 0.5, normalize=True)
    S = sharded_similarity(lambda b, e: comp.compute_similarity(Xt[b:e], 7), Xt.shape[0])
    rng = np.random.default_rng(3)
    scores = rng.standard_normal(X.shape)
    gt = sps.csr_matrix((rng.random(X.shape) > 0.8).astype(np.float64))
    core = O.EvaluatorCore(gt, [])

    def evaluate(b, e):
        if e == b:
            return _RawMetrics(X.shape[1])
        m = core.get_metrics_f64(scores[b:e], 5, b, 1)
        return _RawMetrics(X.shape[1], m.raw(), m.item_cnt())

    total = sharded_metrics(evaluate, X.shape[0], _RawMetrics(X.shape[1]))
    # the byte exchange under it: buffers of different lengths, one of them empty; and a rank without rows
    from irspack_amd.sharding import _all_gather_bytes
    mine = (np.arange(0 if rank == 1 else 1000 + rank) % 251).astype(np.uint8)
    parts = _all_gather_bytes(mine)
    assert [p.size for p in parts] == [1000, 0] and np.array_equal(parts[rank], mine)
    S_uneven = sharded_similarity(lambda b, e: comp.compute_similarity(Xt[b:e], 7), Xt.shape[0],
                                  bounds=[0, Xt.shape[0], Xt.shape[0]])
    assert (S_uneven != S).nnz == 0 and np.array_equal(S_uneven.indptr, S.indptr)
    with open(os.path.join(out_dir, f"knn_eval{rank}.pkl"), "wb") as fh:
        pickle.dump((S, total.raw, total.cnt), fh)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_knn_rows_and_evaluator_users(tmp_path):
    import pickle

    import torch.multiprocessing as mp

    import oracle as O
    from conftest import random_csr

    mp.spawn(_knn_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [pickle.load(open(tmp_path / f"knn_eval{r}.pkl", "rb")) for r in range(2)]
    X = random_csr(83, 57, 0.2, 5, dtype=np.float64)
    Xt = sps.csr_matrix(X.T)
    want = O.KNNComputer("cosine", Xt, 0.5, normalize=True).compute_similarity(Xt, 7)
    rng = np.random.default_rng(3)
    scores = rng.standard_normal(X.shape)
    gt = sps.csr_matrix((rng.random(X.shape) > 0.8).astype(np.float64))
    whole = O.EvaluatorCore(gt, []).get_metrics_f64(scores, 5, 0, 1)
    for S, raw, cnt in res:  # every rank ends with the complete result
        assert np.array_equal(S.indptr, want.indptr) and np.array_equal(S.indices, want.indices)
        np.testing.assert_array_equal(S.data, want.data)
        np.testing.assert_array_equal(cnt, whole.item_cnt())
        assert raw[0] == whole.raw()[0] and raw[1] == whole.raw()[1]  # valid / total users
        np.testing.assert_allclose(raw[2:], whole.raw()[2:], rtol=1e-12)
