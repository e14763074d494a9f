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


@pytest.mark.parametrize("extra,env_extra,timed", [
    # `python bench.py --gpus 2` with NO outer launcher: bench.py starts its two ranks itself (a child
    # torch.distributed.run, before the parent touches the GPU) and relays rank 0's line, which also
    # carries the row-sharded kNN leg and the user-sharded evaluator leg
    (["--shape", "small", "--balance", "equal", "--self-launch"], {}, "torch"),
    (["--shape", "small", "--balance", "cost", "--solver", "CG"], {}, "torch"),
    (["--shape", "c4_small", "--K", "128", "--solver", "CG"], {}, "torch"),
    # the native path as the timed one: the RCCL-free peer-store transport (two processes, one GPU)
    (["--shape", "small", "--balance", "equal"], {"IRSPACK_AMD_BENCH_COMM": "local"}, "native"),
    (["--shape", "small", "--balance", "cost", "--solver", "CG"], {"IRSPACK_AMD_BENCH_COMM": "local"}, "native"),
    # one rank cannot set the transport up -> BOTH ranks time the torch host loop, and say so
    (["--shape", "small", "--balance", "equal"],
     {"IRSPACK_AMD_BENCH_COMM": "local", "IRSPACK_AMD_TEST_FAIL_COMM_RANK": "1"}, "fallback"),
    # one rank's first native epoch fails; the other's waits time out (5 s) -> both fall back
    (["--shape", "small", "--balance", "equal"],
     {"IRSPACK_AMD_BENCH_COMM": "local", "IRSPACK_AMD_TEST_FAIL_STEP_RANK": "0",
      "IRSPACK_AMD_PEER_TIMEOUT_S": "5"}, "fallback")])
def test_bench_two_ranks_control_flow(extra, env_extra, timed):
    """bench.py's N > 1 path (process group, shards, barrier + max-over-ranks timing, the
    per-phase compute / all-reduce / all-gather split, one JSON line from rank 0) with two ranks
    on one device over gloo: equal shards, cost-balanced shards, and the configs[3] invocation
    (`--shape c4 --K 128 --solver CG`, here its 1/50-scale matrix), whose longest row selects
    the cost balance by itself; the native path (irs_ials_sharded_step) timed over the peer-store
    transport; and the collective fallback when one rank fails to create the transport or to run
    its first native epoch - every rank then times the host loop and the line records why."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IRSPACK_AMD_BENCH_BACKEND="gloo", IRSPACK_AMD_BENCH_ONE_DEVICE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    self_launch = "--self-launch" in extra
    extra = [a for a in extra if a != "--self-launch"]
    if self_launch:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        launcher = [sys.executable]
    else:
        extra = extra + ["--no-secondary"]
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    out = subprocess.run(
        [*launcher, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", *extra],
        env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    if self_launch:
        knn, ev = d["secondary"]["knn"], d["secondary"]["evaluator"]
        assert "error" not in knn and "error" not in ev, (knn, ev)
        assert knn["item_pairs_per_s"] > 0 and len(knn["row_bounds"]) == 3 and knn["out_nnz"] > 0
        assert ev["users_per_s"] > 0 and 0 < ev["valid_user"] <= ev["total_user"] == 4000
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "roofline" in d and d["roofline"]["frac"] > 0
    want_balance = "equal" if "equal" in extra else "cost"
    assert d["config"]["balance"] == want_balance
    c = d["comm"]
    assert c["compute_ms"] > 0 and c["allgather_ms"] > 0 and c["allreduce_ms"] >= 0
    assert 0 < c["exposed_comm_ms"] <= c["allreduce_ms"] + c["allgather_ms"] + 1e-6
    assert len(c["exchange"]) == 2 and c["overlap"] is False
    pf = c["native_preflight"]
    if timed == "torch":
        assert pf is None and c["timed_path"] == "torch.distributed host loop"
    elif timed == "native":
        assert pf["created"] and pf["first_epoch"] and "error" not in pf
        assert c["timed_path"].startswith("native") and "local" in c["timed_path"]
    else:
        assert not (pf["created"] and pf["first_epoch"]) and pf["error"]
        assert c["timed_path"] == "torch.distributed host loop"


def test_bench_rccl_path_with_one_rank():
    """bench.py's N > 1 path over RCCL cannot run with two ranks on the one GPU of the box; with ONE rank
    (IRSPACK_AMD_BENCH_FORCE_DIST=1: an RCCL process group of size 1) every line of it does run - the
    library-owned communicators, the collective set-up vote, the preflight epoch, the exchange A/B over
    `auto`, `mesh` and `peer` with its digests, the timed native region, the host-loop phase split and
    the `comm` object of the line.  (What one rank cannot show is data on the wire.)"""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IRSPACK_AMD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", IRSPACK_AMD_BENCH_AB_EPOCHS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--shape", "small", "--balance", "equal", "--no-secondary", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    c = d["comm"]
    assert c["native_preflight"] == {"requested": "rccl", "created": True, "first_epoch": True, "peers_mapped": False} \
        or c["native_preflight"]["created"] and c["native_preflight"]["first_epoch"]
    assert c["timed_path"].startswith("native") and "rccl" in c["timed_path"]
    ab = c["exchange_ab"]
    assert "error" not in ab, ab
    assert set(ab["modes"]) == {"auto", "mesh", "peer"} and ab["peers_mapped"] is True
    digests = {m: r["digest"] for m, r in ab["modes"].items()}
    assert all("error" not in r and r["replicas_identical"] and r["ms_per_epoch"] > 0 for r in ab["modes"].values()), ab
    assert len(set(digests.values())) == 1, digests  # the exchange moves rows, it does not compute
    assert sorted(ab["agree_with_auto"]) == ["auto", "mesh", "peer"] and ab["timed"] in ab["modes"]


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
    # the byte exchange of sharded_similarity on RCCL's device tensors (empty and odd-sized buffers)
    from irspack_amd.sharding import _all_gather_bytes
    for n in (0, 1, 12345):
        buf = (np.arange(n) % 251).astype(np.uint8)
        got = _all_gather_bytes(buf)
        ok = ok and len(got) == 1 and np.array_equal(got[0], buf)
    with open(os.path.join(out_dir, "rccl1.txt"), "w") as fh:
        fh.write("ok" if ok else "mismatch")
    dist.destroy_process_group()


def test_rccl_calls_on_library_buffers_world_size_one(tmp_path):
    """all_gather_into_tensor (in place), all_reduce on a second communicator and broadcast run
    through RCCL on zero-copy views of the library's device buffers."""
    import torch.multiprocessing as mp

    mp.spawn(_rccl_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "rccl1.txt").read_text() == "ok"


def _native_world1_worker(rank, kind, K, out_dir, exchange="auto", transport="rccl"):
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
    tr = ShardedIALSTrainer(local, [0, U], [0, I], native=transport, exchange=exchange)
    assert tr.native and tr.native_error is None, tr.native_error
    assert tr.native_transport == transport and tr.native_exchange == exchange
    ref = IALSTrainer(mc, X)
    assert tr.preflight_step(sc)
    ref.step(sc)
    for _ in range(2):
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
    if transport == "rccl":  # the exchange can be switched between steps
        for mode in ("mesh", "broadcast", "auto"):
            tr.set_exchange(mode)
            tr.step(sc)
            ref.step(sc)
            np.testing.assert_array_equal(local.trainer.item, ref.item)
        try:
            tr.set_exchange("peer")  # no peers mapped unless asked for at creation
        except ValueError:
            assert exchange != "peer"
        else:
            assert exchange == "peer"
    try:
        local.sharded_step(sc, [0, U - 1], [0, I])  # bounds that do not cover every row
    except ValueError:
        pass
    else:
        raise AssertionError("bad row bounds were accepted")
    tr.close()
    with open(os.path.join(out_dir, f"native_{kind}_{K}.txt"), "w") as f:
        f.write("ok")


@pytest.mark.parametrize("kind,K,exchange,chunks,transport", [
    ("CHOLESKY", 64, "auto", 1, "rccl"), ("CG", 200, "auto", 1, "rccl"),
    ("CHOLESKY", 64, "broadcast", 1, "rccl"), ("CHOLESKY", 64, "auto", 3, "rccl"),
    ("CG", 200, "auto", 2, "rccl"), ("CG", 128, "auto", 4, "rccl"),
    ("CHOLESKY", 64, "mesh", 1, "rccl"), ("CG", 128, "mesh", 3, "rccl"),
    ("CHOLESKY", 64, "peer", 1, "rccl"), ("CG", 128, "peer", 2, "rccl"),
    ("CHOLESKY", 64, "peer", 1, "local"), ("CG", 200, "peer", 3, "local")])
def test_native_sharded_step_world_size_one(tmp_path, kind, K, exchange, chunks, transport, monkeypatch):
    """``irs_ials_sharded_step`` (the epoch behind one C-ABI call, the transport opened and driven
    by the library itself) on the one GPU of the box: world size 1 - communicator creation, the
    collective set-up vote, the stream / event plumbing and the Gramian prefetch run.  The result
    must be the unsharded trainer's, bit for bit, over three epochs (the prefetched Gramian of the
    next epoch is used from the second one on), and a factor set from outside must invalidate it.
    The collectives themselves ARE issued (an all-reduce / all-gather / broadcast over one rank is
    the identity): the calls, buffers, counts, streams and events are those of a multi-GPU epoch.
    "broadcast": the grouped in-place broadcasts of uneven shards, forced for the equal ones;
    "mesh": the send / receive group (empty over one rank); "peer": the signal / wait kernels and,
    with transport "local", the mailbox all-reduce - no RCCL communicator at all.
    chunks > 1 (IRSPACK_AMD_SHARD_CHUNKS): the rows of the shard cut into chunks with task lists of
    their own, each exchanged behind its own solve: same bits (rows are independent).
    (In a spawned process like the other tests of this file: torch initialises the device there.)"""
    import torch.multiprocessing as mp

    if chunks > 1:  # the shard's rows solved and exchanged in chunks (chunk k + 1 solved while chunk k travels)
        monkeypatch.setenv("IRSPACK_AMD_SHARD_CHUNKS", str(chunks))

    mp.spawn(_native_world1_worker, args=(kind, K, str(tmp_path), exchange, transport), nprocs=1, join=True)
    assert (tmp_path / f"native_{kind}_{K}.txt").read_text() == "ok"


def _local_transport_worker(rank, world, port, kind, K, out_dir, chunks):
    import torch.distributed as dist

    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, SolverType)
    from irspack_amd.sharding import (HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds,
                                      shard_bounds)
    from irspack_amd.synthetic import make_interactions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["IRSPACK_AMD_PEER_TIMEOUT_S"] = "30"
    if chunks > 1:
        os.environ["IRSPACK_AMD_SHARD_CHUNKS"] = str(chunks)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = make_interactions("small")
    mc = IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-2).build()
    sc = (IALSSolverConfigBuilder().set_solver_type(SolverType[kind]).set_max_cg_steps(3).build())
    ub, ib = shard_bounds(X, K, kind, world) if chunks == 1 else equal_shard_bounds(X, world)
    shard = (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1])
    local = HipLocalSolver(mc, X, shard, 0)
    tr = ShardedIALSTrainer(local, ub, ib, native="local", exchange="peer", watchdog_s=240)
    assert tr.native, tr.native_error
    assert tr.preflight_step(sc)
    tr.step(sc)
    tr.synchronize()
    got_user, got_item = local.trainer.user, local.trainer.item
    tr.close()
    # the torch host loop (gloo) from the same start on a second solver: with two ranks the
    # Gramian is a + b either way, so the two transports must agree bit for bit
    os.environ.pop("IRSPACK_AMD_SHARD_CHUNKS", None)
    local2 = HipLocalSolver(mc, X, shard, 0)
    host = ShardedIALSTrainer(local2, ub, ib)
    for _ in range(2):
        host.step(sc)
    host.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), user=got_user, item=got_item,
             host_user=local2.trainer.user, host_item=local2.trainer.item)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,K,chunks", [("CHOLESKY", 64, 1), ("CG", 128, 1), ("CG", 64, 3), ("IALSPP", 128, 1)])
def test_peer_store_transport_two_processes_one_gpu(tmp_path, kind, K, chunks):
    """The RCCL-free transport with TWO ranks (two processes sharing device 0, which RCCL refuses):
    factor buffers, mailbox and flags exported with hipIpcGetMemHandle and mapped by the peer,
    the solved rows STORED into the peer's replica by the push kernel, arrival signalled by
    sequence numbers in mapped flag words, the K x K Gramian summed out of the mailbox slots in rank
    order, the solver's error flag reduced the same way.  Two epochs (the second uses the
    prefetched Gramian): replicas bit-identical, and bit-identical to the torch / gloo host loop
    from the same start."""
    import torch.multiprocessing as mp

    mp.spawn(_local_transport_worker, args=(2, _free_port(), kind, K, str(tmp_path), chunks), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for k in ("user", "item"):
        np.testing.assert_array_equal(r0[k], r1[k])
        np.testing.assert_array_equal(r0[k], r0["host_" + k])
    assert np.isfinite(r0["user"]).all() and np.abs(r0["user"]).max() > 0


def _error_parity_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, SolverType)
    from irspack_amd.sharding import HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds
    from irspack_amd.synthetic import make_interactions

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X = make_interactions("tiny").tolil()
    X.rows[7], X.data[7] = [], []  # an empty user row in rank 0's shard: alpha0 = 0 makes its system the zero matrix
    X = X.tocsr()
    mc = IALSModelConfigBuilder().set_K(16).set_alpha0(0.0).set_reg(1e-3).build()
    sc = IALSSolverConfigBuilder().set_solver_type(SolverType.CHOLESKY).build()
    ub, ib = equal_shard_bounds(X, world)
    local = HipLocalSolver(mc, X, (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1]), 0)
    tr = ShardedIALSTrainer(local, ub, ib, native="local", exchange="peer", watchdog_s=240)
    assert tr.native, tr.native_error
    msg = "none"
    try:
        tr.step(sc)
    except RuntimeError as exc:
        msg = str(exc)
    with open(os.path.join(out_dir, f"err{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_solver_error_is_raised_on_every_rank(tmp_path):
    """What the reference throws from inside a solve (hpp:316-318: LLT of a zero matrix, an empty
    row at alpha0 = 0) happens on ONE rank's rows; the sharded step all-reduces the device error
    flag, so both ranks raise the same RuntimeError instead of one raising and one waiting."""
    import torch.multiprocessing as mp

    mp.spawn(_error_parity_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = [(tmp_path / f"err{r}.txt").read_text() for r in range(2)]
    assert msgs[0] == msgs[1] == "Cholesky decomposition failed."
