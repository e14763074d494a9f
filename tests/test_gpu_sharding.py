"""GPU test of the product multi-rank path (``HipLocalSolver`` + ``ShardedIALSTrainer``).

The test box has ONE GPU, so both ranks drive device 0 and the collectives run over
``gloo`` (which stages device tensors through the host); RCCL refuses two ranks on one
device.  What is exercised is everything that is ours: the row shards inside the HIP
library, the partial Gramian + ``copy_rows_async`` staging, the half steps on a shard and
the shard broadcasts.  The 8-GPU RCCL run is the driver's.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from conftest import row_rel_err  # noqa: E402

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, kind, out_dir, equal=False, K=64, shape="small"):
    import torch
    import torch.distributed as dist

    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, SolverType)
    from irspack_amd.sharding import (HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds,
                                      shard_bounds)
    from irspack_amd.synthetic import make_interactions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = make_interactions(shape)
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-2).build()
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(3).build())
    ub, ib = equal_shard_bounds(X, world) if equal else shard_bounds(X, K, kind, world)
    local = HipLocalSolver(mc, X, (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1]), 0)
    tr = ShardedIALSTrainer(local, ub, ib)
    for _ in range(2):
        tr.step(sc)
    tr.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), user=local.trainer.user, item=local.trainer.item)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind,equal,K", [("CHOLESKY", False, 64), ("CHOLESKY", True, 64),
                                          ("CG", False, 64), ("CG", True, 64), ("IALSPP", True, 64),
                                          # BASELINE configs[3] is K = 128, sharded
                                          ("CG", True, 128), ("CHOLESKY", True, 128),
                                          # iALS++ with two 64-dim blocks (chained passes, the
                                          # second stream of the short-row launch) on a shard
                                          ("IALSPP", True, 128)])
def test_two_ranks_match_single_gpu(tmp_path, kind, equal, K):
    import torch.multiprocessing as mp

    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer,
                                                      SolverType)
    from irspack_amd.synthetic import make_interactions

    mp.spawn(_worker, args=(2, _free_port(), kind, str(tmp_path), equal, K), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["user"], r1["user"])  # replicas stay bit-identical
    np.testing.assert_array_equal(r0["item"], r1["item"])
    X = make_interactions("small")
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-2).build()
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(3).build()
    ref = IALSTrainer(mc, X)
    for _ in range(2):
        ref.step(sc)
    # only the Gramian's summation order differs between 1 and 2 ranks; two free-running
    # epochs (truncated CG amplifies the last-bit differences) stay within 5e-4 per row
    assert row_rel_err(r0["user"], ref.user) < 5e-4
    assert row_rel_err(r0["item"], ref.item) < 5e-4


@pytest.mark.parametrize("extra", [["--shape", "small", "--balance", "equal"],
                                   ["--shape", "small", "--balance", "cost", "--solver", "CG"],
                                   ["--shape", "c4_small", "--K", "128", "--solver", "CG"]])
def test_bench_two_ranks_control_flow(extra):
    """bench.py's N > 1 path (process group, shards, barrier + max-over-ranks timing, the
    per-phase compute / all-reduce / all-gather split, one JSON line from rank 0) with two ranks
    on one device over gloo: equal shards, cost-balanced shards, and the configs[3] invocation
    (`--shape c4 --K 128 --solver CG`, here its 1/50-scale matrix), whose longest row selects
    the cost balance by itself."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IRSPACK_AMD_BENCH_BACKEND="gloo", IRSPACK_AMD_BENCH_ONE_DEVICE="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"),
         "--gpus", "2", "--steps", "2", "--warmup", "1", *extra],
        env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "roofline" in d and d["roofline"]["frac"] > 0
    want_balance = "equal" if "equal" in extra else "cost"
    assert d["config"]["balance"] == want_balance
    c = d["comm"]
    assert c["compute_ms"] > 0 and c["allgather_ms"] > 0 and c["allreduce_ms"] >= 0
    assert 0 < c["exposed_comm_ms"] <= c["allreduce_ms"] + c["allgather_ms"] + 1e-6
    assert len(c["exchange"]) == 2 and c["overlap"] is False


def _knn_eval_worker(rank, world, port, out_dir):
    """Product kNN computer and evaluator sharded over two ranks on one device."""
    import pickle

    import scipy.sparse as sps
    import torch.distributed as dist

    from irspack_amd.evaluation._core_evaluator import EvaluatorCore, Metrics
    from irspack_amd.recommenders._knn import CosineSimilarityComputer
    from irspack_amd.sharding import sharded_metrics, sharded_similarity
    from irspack_amd.synthetic import make_interactions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = make_interactions("small")
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    comp = CosineSimilarityComputer(Xt, 0.0, True)
    S = sharded_similarity(lambda b, e: comp.compute_similarity(Xt, 20, rows=(b, e)), Xt.shape[0])
    rng = np.random.default_rng(1)
    scores = rng.standard_normal(X.shape).astype(np.float32)
    gt = sps.csr_matrix((rng.random(X.shape) > 0.97).astype(np.float64))
    core = EvaluatorCore(gt, [])
    total = sharded_metrics(lambda b, e: core.get_metrics_f32(scores[b:e], 20, b, 1), X.shape[0],
                            Metrics(X.shape[1]))
    with open(os.path.join(out_dir, f"ke{rank}.pkl"), "wb") as fh:
        pickle.dump((S, total), fh)
    dist.barrier()
    dist.destroy_process_group()


def test_knn_rows_and_evaluator_users_over_two_ranks(tmp_path):
    import pickle

    import scipy.sparse as sps
    import torch.multiprocessing as mp

    from irspack_amd.evaluation._core_evaluator import EvaluatorCore
    from irspack_amd.recommenders._knn import CosineSimilarityComputer
    from irspack_amd.synthetic import make_interactions

    mp.spawn(_knn_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    res = [pickle.load(open(tmp_path / f"ke{r}.pkl", "rb")) for r in range(2)]
    X = make_interactions("small")
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    want = CosineSimilarityComputer(Xt, 0.0, True).compute_similarity(Xt, 20)
    rng = np.random.default_rng(1)
    scores = rng.standard_normal(X.shape).astype(np.float32)
    gt = sps.csr_matrix((rng.random(X.shape) > 0.97).astype(np.float64))
    whole = EvaluatorCore(gt, []).get_metrics_f32(scores, 20, 0, 1)
    for S, m in res:
        assert np.array_equal(S.indptr, want.indptr) and np.array_equal(S.indices, want.indices)
        np.testing.assert_array_equal(S.data, want.data)
        np.testing.assert_array_equal(m.item_cnt, whole.item_cnt)
        assert m.valid_user == whole.valid_user and m.total_user == whole.total_user
        for k in ("hit", "ndcg", "recall", "map", "precision"):
            assert getattr(m, k) == pytest.approx(getattr(whole, k), rel=1e-12)


def _rccl_world1_worker(rank, port, out_dir):
    """The RCCL calls of the sharded loop on the library's own device buffers, world size 1
    (the only RCCL configuration a one-GPU box can run: RCCL refuses two ranks on a device)."""
    import torch
    import torch.distributed as dist

    from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder
    from irspack_amd.sharding import HipLocalSolver, equal_shard_bounds
    from irspack_amd.synthetic import make_interactions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    X = make_interactions("small")
    mc = IALSModelConfigBuilder().set_K(64).set_alpha0(0.1).set_reg(1e-2).build()
    ub, ib = equal_shard_bounds(X, 1)
    local = HipLocalSolver(mc, X, (ub[0], ub[1], ib[0], ib[1]), 0)
    second = dist.new_group(ranks=[0])  # the Gramian's communicator
    user = local.factor_view(0)
    before = user.clone()
    # in-place all-gather of the (whole) row block, as ShardedIALSTrainer._exchange_rows issues it
    w = dist.all_gather_into_tensor(user, user[0:user.shape[0]], async_op=True)
    local.partial_gramian(1)
    red = dist.all_reduce(local.gramian_view(1), op=dist.ReduceOp.SUM, group=second, async_op=True)
    w.wait()
    red.wait()
    local.finish_gramian(1)
    dist.broadcast(local.factor_view(1), src=0)
    local.synchronize()
    torch.cuda.synchronize()
    ok = bool(torch.equal(user, before)) and bool(torch.isfinite(local.gramian_view(1)).all())
    with open(os.path.join(out_dir, "rccl1.txt"), "w") as fh:
        fh.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def test_rccl_calls_on_library_buffers_world_size_one(tmp_path):
    """all_gather_into_tensor (in place), all_reduce on a second communicator and broadcast run
    through RCCL on zero-copy views of the library's device buffers."""
    import torch.multiprocessing as mp

    mp.spawn(_rccl_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "rccl1.txt").read_text() == "ok"


def _native_world1_worker(rank, kind, K, out_dir):
    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, IALSTrainer,
                                                      SolverType)
    from irspack_amd.sharding import HipLocalSolver, ShardedIALSTrainer
    from irspack_amd.synthetic import make_interactions

    X = make_interactions("small")
    U, I = X.shape
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-2).build()
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(3).build()
    local = HipLocalSolver(mc, X, (0, U, 0, I), 0)
    tr = ShardedIALSTrainer(local, [0, U], [0, I], native=True)
    ref = IALSTrainer(mc, X)
    for _ in range(3):
        tr.step(sc)
        ref.step(sc)
    np.testing.assert_array_equal(local.trainer.user, ref.user)
    np.testing.assert_array_equal(local.trainer.item, ref.item)
    item = ref.item * 0.5
    local.trainer.item = item  # (invalidates the prefetched Gramian of the user half)
    ref.item = item
    tr.step(sc)
    ref.step(sc)
    np.testing.assert_array_equal(local.trainer.user, ref.user)
    try:
        local.sharded_step(sc, [0, U - 1], [0, I])  # bounds that do not cover every row
    except ValueError:
        pass
    else:
        raise AssertionError("bad row bounds were accepted")
    tr.close()
    with open(os.path.join(out_dir, f"native_{kind}_{K}.txt"), "w") as f:
        f.write("ok")


@pytest.mark.parametrize("kind,K,exchange,chunks", [("CHOLESKY", 64, "allgather", 1), ("CG", 200, "allgather", 1),
                                                   ("CHOLESKY", 64, "broadcast", 1), ("CHOLESKY", 64, "allgather", 3),
                                                   ("CG", 200, "allgather", 2), ("CG", 128, "allgather", 4)])
def test_native_sharded_step_world_size_one(tmp_path, kind, K, exchange, chunks, monkeypatch):
    """``irs_ials_sharded_step`` (the epoch behind one C-ABI call, RCCL opened and called by the
    library itself) on the one GPU of the box: world size 1 - communicator creation, the stream /
    event plumbing and the Gramian prefetch run; the collectives are skipped.  The result must be
    the unsharded trainer's, bit for bit, over three epochs (the prefetched Gramian of the next
    epoch is used from the second one on), and a factor set from outside must invalidate it.
    The collectives themselves ARE issued (an all-reduce / all-gather / broadcast over one rank is
    the identity): the calls, buffers, counts, streams and events are those of a multi-GPU epoch.
    "broadcast": the grouped in-place broadcasts of uneven shards, forced for the equal ones.
    chunks > 1 (IRSPACK_AMD_SHARD_CHUNKS): the rows of the shard cut into chunks with task lists of
    their own, each exchanged behind its own solve: same bits (rows are independent).
    (In a spawned process like the other tests of this file: torch initialises the device there.)"""
    import torch.multiprocessing as mp

    monkeypatch.setenv("IRSPACK_AMD_SHARD_EXCHANGE", exchange)
    if chunks > 1:  # the shard's rows solved and exchanged in chunks (chunk k + 1 solved while chunk k travels)
        monkeypatch.setenv("IRSPACK_AMD_SHARD_CHUNKS", str(chunks))

    mp.spawn(_native_world1_worker, args=(kind, K, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / f"native_{kind}_{K}.txt").read_text() == "ok"
