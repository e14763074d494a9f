"""Multi-GPU host loop of the iALS step: one process per GPU, rows sharded.

The reference has no distributed layer; this is the natural
sharding of ``IALSTrainer::step`` (IALSTrainer.hpp:758-789): the per-row solves
are independent, so users (items on the alternate half-epoch) are split into
contiguous, cost-balanced ranges; every rank keeps full replicas of both
factor matrices and, per half-epoch, takes part in

  1. an all-reduce of the K x K partial Gramian of its own rows
     (``Solver::prepare_p``, hpp:78-115, summed over ranks), and
  2. an all-gather of the freshly solved factor rows, always ONE collective:
     in place when the shards are equal row blocks of the (row padded) factor buffer
     (``equal_bounds``; what ``bench.py`` uses - with RCCL over xGMI every shard leaves on
     its own links), through a staging buffer of ``world x max-shard`` rows when they are
     cost-balanced and uneven (``shard_bounds``).

Overlap: the Gramian of the NEXT half-epoch only needs the rows this rank has just solved,
so its partial sum and its K x K all-reduce (on a second process group = a second RCCL
communicator and stream) are issued while the all-gather is still in flight.

Which exchange the backend supports is probed ONCE at construction with a collective every
rank takes part in (``gloo`` cannot gather device tensors in one call and uses per-rank
broadcasts); nothing on the hot path catches exceptions.

Host side of a sharded run: every rank prepares only its own rows of X and X^T
(``irs_ials_create`` with a shard), and the initial factors - a sequential libstdc++ random
stream, 56 s for the 10 M x 1 M shape - are drawn by rank 0 only and broadcast.

kNN and the evaluator shard without a data-path collective: ``sharded_similarity`` (target
rows) and ``sharded_metrics`` (users) give every rank a contiguous range and combine the
results on the host (CSR blocks stacked in rank order; ``Metrics.merge``, a plain sum,
evaluator.cpp:76-85).

Collectives go through ``torch.distributed`` (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests) and run IN PLACE on torch views of the solver's own
buffers (zero-copy: the HIP library's device memory is wrapped through
``__cuda_array_interface__``), so no staging copies sit between the kernels and
the wire.  The arithmetic is behind a small ``LocalSolver`` interface whose
product implementation (``HipLocalSolver``) drives the HIP library.  Results equal
the single-GPU ones up to the summation order of the Gramian.
"""

import os
from typing import List, Sequence, Tuple

import numpy as np
import scipy.sparse as sps


def balanced_bounds(cost: np.ndarray, parts: int) -> List[int]:
    """Contiguous partition of ``len(cost)`` rows into ``parts`` ranges of ~equal cost."""
    n = int(cost.shape[0])
    csum = np.concatenate([[0.0], np.cumsum(cost.astype(np.float64))])
    total = csum[-1]
    bounds = [0]
    for p in range(1, parts):
        target = total * p / parts
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return bounds


def row_cost(nnz: np.ndarray, K: int, solver: str) -> np.ndarray:
    """Per-row work model: gather + rank update, plus the dense solve."""
    nnz = nnz.astype(np.float64)
    if solver == "CHOLESKY":
        return nnz * (K * K + 2.0 * K) + (K ** 3) / 6.0 + 2.0 * K * K
    return nnz * (K * K + 2.0 * K) + 8.0 * K * K


def equal_bounds(n: int, parts: int, multiple: int = 8) -> List[int]:
    """``parts`` equal row blocks of the buffer padded to a multiple of ``multiple`` rows,
    clipped to ``n``: shard r is rows [r*S, (r+1)*S) with S = padded / parts.  Row order of
    the benchmark matrix is random, so equal blocks are cost-balanced to a few per cent."""
    padded = -(-n // multiple) * multiple
    if padded % parts:
        raise ValueError("the padded row count must be divisible by the number of shards.")
    S = padded // parts
    return [min(r * S, n) for r in range(parts)] + [n]


def even_bounds(n: int, parts: int) -> List[int]:
    """``parts`` contiguous ranges of ``n`` units whose sizes differ by at most one."""
    return [n * r // parts for r in range(parts + 1)]


def equal_shard_bounds(X: sps.csr_matrix, world: int) -> Tuple[List[int], List[int]]:
    return equal_bounds(X.shape[0], world), equal_bounds(X.shape[1], world)


def shard_bounds(X: sps.csr_matrix, K: int, solver: str, world: int) -> Tuple[List[int], List[int]]:
    X = sps.csr_matrix(X)
    user_nnz = np.diff(X.indptr)
    item_nnz = np.bincount(X.indices, minlength=X.shape[1])
    return (balanced_bounds(row_cost(user_nnz, K, solver), world),
            balanced_bounds(row_cost(item_nnz, K, solver), world))


class LocalSolver:
    """What the host loop needs from one rank's device (or, in tests, from the oracle)."""

    def factor_view(self, which: int):
        """torch tensor [rows, ld] ALIASING the solver's factor matrix (0 user, 1 item)."""
        raise NotImplementedError

    def gramian_view(self, side: int):
        """torch tensor [ld, ld] aliasing the Gramian accumulator of the solve of `side`."""
        raise NotImplementedError

    def partial_gramian(self, side: int) -> None:
        """gramian_view(side) <- sum over this rank's rows of the *other* side of f f^T."""
        raise NotImplementedError

    def finish_gramian(self, side: int) -> None:
        """Scale the (all-reduced) gramian_view(side) by alpha0 and make it the solve's P."""
        raise NotImplementedError

    def half_step(self, side: int, solver_config) -> None:
        raise NotImplementedError

    def synchronize(self) -> None:
        raise NotImplementedError


class _DeviceArray:
    """Minimal ``__cuda_array_interface__`` carrier for a float32 [rows, ld] device buffer."""

    def __init__(self, ptr: int, rows: int, ld: int) -> None:
        self.__cuda_array_interface__ = {
            "shape": (rows, ld), "typestr": "<f4", "data": (ptr, False), "version": 2,
            "strides": None,
        }


def _group_info(group) -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


class HipLocalSolver(LocalSolver):
    """Product implementation: the HIP trainer of this rank's GPU, on torch's current stream.

    With more than one rank only group rank 0 draws the initial factors (a sequential host
    random stream); the others start from zeros and receive them by broadcast."""

    def __init__(self, model_config, X, shard: Tuple[int, int, int, int], device: int, group=None):
        import torch
        import torch.distributed as dist

        from .recommenders._ials_core import IALSModelConfig, IALSTrainer

        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        rank, world = _group_info(group)
        cfg = model_config
        if world > 1 and rank != 0:
            state = list(model_config.__getstate__())
            state[4] = 0.0  # init_stdev <= 0: no draw, zero factors (filled by the broadcast)
            cfg = IALSModelConfig(*state)
        self.trainer = IALSTrainer(cfg, X, device=device, shard=shard)
        self.trainer._config = model_config
        self.trainer.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._views = {}
        for which in range(4):
            ptr, rows, ld = self.trainer.device_buffer(which)
            if rows * ld == 0:
                self._views[which] = torch.empty((rows, ld), dtype=torch.float32, device=self.device)
            else:
                self._views[which] = torch.as_tensor(_DeviceArray(ptr, rows, ld), device=self.device)
        if world > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            for which in (0, 1):
                dist.broadcast(self._views[which], src=src, group=group)

    def factor_view(self, which: int):
        return self._views[which]

    def gramian_view(self, side: int):
        return self._views[2 + side]

    def partial_gramian(self, side: int) -> None:
        self.trainer.partial_gramian_async(side)

    def gramian(self, side: int) -> None:
        """the whole Gramian of `side` in one call: for a solver that holds every row (world size 1)"""
        self.trainer.gramian_async(side)

    def finish_gramian(self, side: int) -> None:
        self.trainer.finish_gramian_async(side)

    def half_step(self, side: int, solver_config) -> None:
        self.trainer.half_step_async(side, solver_config)

    def synchronize(self) -> None:
        self.trainer.synchronize()

    # -- the epoch behind the C ABI (irs_ials_sharded_step): the transport lives inside the library --
    def create_comm(self, group=None, transport: str = "rccl", peers: bool = True) -> None:
        """This rank's transport (irs_comm).

        ``transport="rccl"``: two RCCL communicators; the 256-byte id is made on group rank 0 and
        handed round through ``torch.distributed`` (an object broadcast: any backend).
        ``transport="local"``: no RCCL - rows and Gramians travel as stores into the peers' mapped
        memory (``irs_comm_create_local``).  ``peers``: map every rank's factor buffers, mailbox
        and flags (``irs_comm_export`` / ``irs_comm_attach``, blobs gathered through
        ``torch.distributed``) - a must for "local", and what ``set_exchange("peer")`` needs on
        an RCCL communicator; a mapping failure there only leaves ``peers_attached`` False.

        Collective-safe: whatever fails on this rank, it still takes part in EVERY exchange of the
        set-up in the same order as the others (a rank with nothing to offer sends ``None``), the
        ranks agree before the RCCL communicators are created (a rank missing from
        ``ncclCommInitRank`` would leave the others waiting in it), and the error is raised only
        at the end - on every rank that failed or depended on one that did."""
        import ctypes as C

        import torch.distributed as dist

        from ._lib import COMM_HANDLE_BYTES, check, lib

        if transport not in ("rccl", "local"):
            raise ValueError("transport must be 'rccl' or 'local'.")
        rank, world = _group_info(group)
        src = dist.get_global_rank(group, 0) if (group is not None and world > 1) else 0

        def gather(value):
            if world == 1:
                return [value]
            out = [None] * world
            dist.all_gather_object(out, value, group=group)
            return out

        err = None
        id_bytes = None
        if transport == "rccl":
            if rank == 0:
                try:
                    buf = (C.c_char * 256)()
                    check(lib().irs_comm_unique_id(buf))
                    id_bytes = bytes(buf)
                except (RuntimeError, ValueError) as exc:
                    err = repr(exc)
            if world > 1:
                box = [id_bytes]
                dist.broadcast_object_list(box, src=src, group=group)
                id_bytes = box[0]
            if id_bytes is None:
                err = err or "group rank 0 could not make the RCCL id"
        if os.environ.get("IRSPACK_AMD_TEST_FAIL_COMM_RANK") == str(rank):  # fault injection (tests)
            err = err or "RuntimeError('injected failure of the communicator set-up on this rank')"
        if not all(gather(err is None)):  # nobody enters ncclCommInitRank unless everybody will
            err = err or "another rank could not start the communicator set-up"
        h = C.c_void_p()
        if err is None:
            try:
                if transport == "local":
                    check(lib().irs_comm_create_local(C.c_int32(rank), C.c_int32(world),
                                                      C.c_int32(self.device.index), C.byref(h)))
                else:
                    check(lib().irs_comm_create((C.c_char * 256).from_buffer_copy(id_bytes), C.c_int32(rank),
                                                C.c_int32(world), C.c_int32(self.device.index), C.byref(h)))
            except (RuntimeError, ValueError) as exc:
                err = repr(exc)
        self._comm = h if err is None else None
        self.peers_attached, self.peers_error = False, None
        if peers or transport == "local":
            blob, perr = None, err
            if err is None:
                try:
                    mine = (C.c_char * COMM_HANDLE_BYTES)()
                    check(lib().irs_comm_export(h, self.trainer._h, mine))
                    blob = bytes(mine)
                except (RuntimeError, ValueError) as exc:
                    perr = repr(exc)
            blobs = gather(blob)
            ok = all(b is not None for b in blobs)
            if ok:
                try:
                    allb = (C.c_char * (COMM_HANDLE_BYTES * world)).from_buffer_copy(b"".join(blobs))
                    check(lib().irs_comm_attach(h, self.trainer._h, allb))
                except (RuntimeError, ValueError) as exc:
                    ok, perr = False, repr(exc)
            elif perr is None:
                perr = "another rank could not export its buffers"
            votes = gather(bool(ok))  # attached on every rank or on none
            if ok and not all(votes):
                ok, perr = False, "another rank could not map the peers' memory"
            self.peers_attached, self.peers_error = bool(ok), (None if ok else perr)
            if transport == "local" and not ok:
                err = err or f"peer stores are not available: {perr}"
        if err is not None:
            self.close_comm()
            raise RuntimeError(err)

    def set_exchange(self, mode: str) -> None:
        """How ``sharded_step`` moves the solved rows: "auto" (in-place all-gather / grouped
        broadcasts), "broadcast", "mesh" (one send / receive group) or "peer" (stores into the
        peers' mapped buffers; needs ``peers_attached``).  The same value on every rank."""
        import ctypes as C

        from ._lib import EXCHANGE_MODES, check, lib

        check(lib().irs_comm_set_exchange(self._comm, C.c_int32(EXCHANGE_MODES[mode])))

    def sharded_step(self, solver_config, user_bounds: Sequence[int], item_bounds: Sequence[int]) -> None:
        import ctypes as C

        from ._lib import check, lib, ptr

        if os.environ.get("IRSPACK_AMD_TEST_FAIL_STEP_RANK") == str(_group_info(None)[0]):  # fault injection (tests)
            raise RuntimeError("injected failure of the sharded step on this rank")
        ub = np.ascontiguousarray(user_bounds, dtype=np.int64)
        ib = np.ascontiguousarray(item_bounds, dtype=np.int64)
        sc = solver_config._struct()
        check(lib().irs_ials_sharded_step(self.trainer._h, C.byref(sc), self._comm,
                                          ptr(ub, C.c_int64), ptr(ib, C.c_int64)))

    def close_comm(self) -> None:
        from ._lib import lib

        h = getattr(self, "_comm", None)
        if h:
            lib().irs_comm_destroy(h)
            self._comm = None

    def __del__(self):
        try:
            self.close_comm()
        except Exception:
            pass


def run_with_watchdog(fn, timeout_s: float, what: str, device: int = None):
    """``fn()`` on a worker thread; a call that has not returned after ``timeout_s`` seconds ends
    the PROCESS with exit code 3 (a collective that one rank never entered blocks the others
    inside the driver or on the device, where no exception can reach them: the launcher then
    takes the other ranks down, which is the only way such a job ever ends).  Exceptions of
    ``fn`` are re-raised here.  ``device``: the HIP device the worker binds first - the current
    device is per THREAD and a new thread starts on device 0, where torch.distributed's object
    collectives of every rank would otherwise be staged under the nccl backend."""
    import os
    import sys
    import threading

    box = {}

    def target():
        try:
            if device is not None:
                import torch

                if torch.cuda.is_available():
                    torch.cuda.set_device(device)
            box["value"] = fn()
        except BaseException as exc:  # noqa: BLE001 - handed to the caller
            box["error"] = exc

    th = threading.Thread(target=target, daemon=True, name="irspack-amd-watchdog")
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        print(f"irspack_amd: {what} did not finish within {timeout_s:.0f} s; this rank gives up "
              "(exit code 3).", file=sys.stderr, flush=True)
        os._exit(3)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def all_ranks_ok(ok: bool, group=None) -> bool:
    """True when ``ok`` is true on every rank (an object all-gather: any backend, no device)."""
    import torch.distributed as dist

    rank, world = _group_info(group)
    if world == 1:
        return bool(ok)
    votes = [None] * world
    dist.all_gather_object(votes, bool(ok), group=group)
    return all(votes)


class ShardedIALSTrainer:
    """``IALSTrainer.step`` over ``world_size`` ranks (see the module docstring)."""

    def __init__(self, local: LocalSolver, user_bounds: Sequence[int], item_bounds: Sequence[int],
                 group=None, overlap: bool = False, gram_group=None, timing: bool = False,
                 native=False, exchange: str = "auto", watchdog_s: float = 600.0):
        """``native`` (``HipLocalSolver`` only): the whole epoch runs behind ONE C-ABI call,
        ``irs_ials_sharded_step`` - the library moves the Gramians and the solved rows itself on
        its own streams, the rows travel in place and the next half-epoch's Gramian overlaps
        their exchange.  ``True`` / ``"rccl"``: two RCCL communicators opened by the library
        (from an id that rank 0 makes and ``torch.distributed`` hands round); ``"local"``: no
        RCCL, every exchange is a store into the peers' mapped memory.  ``exchange``: how the
        rows travel ("auto" | "broadcast" | "mesh" | "peer", ``HipLocalSolver.set_exchange``).

        The native transport is set up COLLECTIVELY: every rank tries (under a watchdog of
        ``watchdog_s`` seconds: a rank stuck in communicator creation leaves with exit code 3
        instead of hanging the job), the ranks vote through the torch group, and unless all of
        them succeeded all of them fall back to the host loop below together
        (``self.native`` False, the reason in ``self.native_error``).  ``preflight_step`` does
        the same for the first epoch.  The host loop (ten torch / ctypes calls per epoch) also
        serves the per-phase ``timing``, backends other than RCCL and the CPU tests.

        ``overlap=True`` issues the NEXT half-epoch's K x K all-reduce on a second
        communicator while the all-gather of the solved rows is in flight.  It is OFF by
        default: the two concurrent collectives have only ever run on RCCL at world size 1
        (no multi-GPU hardware was available to this build; the gloo world-2 tests cover the
        logic), so the default is the plain sequential exchange until a world >= 2 RCCL run has
        passed ``tests/test_gpu_sharding.py``.

        ``gram_group``: the second communicator, built by the caller.  When it is None and
        ``overlap`` is set the constructor calls ``dist.new_group`` - a collective over the
        DEFAULT group: every rank of the default group must construct the trainer (passing a
        strict sub-group as ``group`` without a ready-made ``gram_group`` would hang).  A group
        created here is destroyed by ``close()``.

        ``timing=True`` records per half-epoch the time spent in the Gramian (partial sum +
        all-reduce), the solve and the row exchange (``last_timing()``; device events on the
        current stream for CUDA tensors, the host clock otherwise)."""
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.local = local
        self.rank, self.world = _group_info(group)
        assert len(user_bounds) == self.world + 1 and len(item_bounds) == self.world + 1
        self.bounds = (list(user_bounds), list(item_bounds))
        self.overlap = bool(overlap) and self.world > 1
        self._gram_ready = [False, False]  # Gramian of side s already reduced (prefetched)
        self.exchange = ["none", "none"]
        self._stage = [None, None]
        self.gram_group = group
        self._owns_gram_group = False
        self.timing = bool(timing)
        self._marks = []  # (label, event or host time) of the current epoch
        self.native = bool(native)
        self.native_transport = None
        self.native_exchange = None
        self.native_error = None
        self.watchdog_s = float(watchdog_s)
        if self.native:
            if not isinstance(local, HipLocalSolver):
                raise ValueError("native needs a HipLocalSolver.")
            transport = native if isinstance(native, str) else "rccl"
            err = None
            try:
                run_with_watchdog(lambda: local.create_comm(group, transport=transport,
                                                            peers=(transport == "local" or exchange == "peer")),
                                  self.watchdog_s, "creating the native communicators",
                                  device=getattr(local, "device", None))
                local.set_exchange(exchange)
            except (RuntimeError, ValueError) as exc:
                err = repr(exc)
            if all_ranks_ok(err is None, group):
                self.native_transport, self.native_exchange = transport, exchange
            else:  # every rank falls back together
                self.native = False
                self.native_error = err or "another rank could not set up the native transport"
                local.close_comm()
        if not self.native and self.world > 1:
            # a second communicator: the K x K all-reduce of the next half-epoch's Gramian must
            # not queue behind the all-gather of the solved rows
            if self.overlap:
                if gram_group is not None:
                    self.gram_group = gram_group
                else:
                    ranks = (list(range(self.world)) if group is None
                             else [dist.get_global_rank(group, r) for r in range(self.world)])
                    self.gram_group = dist.new_group(ranks=ranks)
                    self._owns_gram_group = True
            self._plan_host_exchange()

    def _plan_host_exchange(self) -> None:
        torch, local = self.torch, self.local
        gather_ok = self._probe_gather(local.factor_view(0))
        for side in (0, 1):
            view, b = local.factor_view(side), self.bounds[side]
            if not gather_ok:
                self.exchange[side] = "broadcast"
                continue
            S = view.shape[0] // self.world
            equal = (view.shape[0] % self.world == 0 and
                     all(b[r] == min(r * S, b[-1]) for r in range(self.world)))
            if equal:
                self.exchange[side] = "inplace"
            else:
                self.exchange[side] = "padded"
                smax = max(b[r + 1] - b[r] for r in range(self.world))
                self._stage[side] = torch.zeros((self.world * max(smax, 1), view.shape[1]),
                                                dtype=view.dtype, device=view.device)

    def set_exchange(self, mode: str) -> None:
        """Switches the native row exchange (every rank, the same mode; between steps)."""
        if not self.native:
            raise RuntimeError("set_exchange needs the native transport.")
        self.local.set_exchange(mode)
        self.native_exchange = mode

    def preflight_step(self, solver_config) -> bool:
        """One native epoch, collectively checked: a rank whose step raises votes no, and unless
        every rank's step succeeded every rank restores the factors it started from and falls
        back to the host loop (``self.native`` False; returns False).  A step that does not
        return within ``watchdog_s`` seconds ends the process with exit code 3 - a collective
        stuck on the device cannot be recovered from inside the process."""
        if not self.native:
            return False
        tr = self.local.trainer
        user0, item0 = tr.user, tr.item
        err = None
        try:
            run_with_watchdog(lambda: self.local.sharded_step(solver_config, self.bounds[0], self.bounds[1]),
                              self.watchdog_s, "the first native sharded epoch",
                              device=getattr(self.local, "device", None))
        except (RuntimeError, ValueError) as exc:
            err = repr(exc)
        if all_ranks_ok(err is None, self.group):
            return True
        self.native = False
        self.native_error = err or "the first native epoch failed on another rank"
        self.local.close_comm()
        tr.user, tr.item = user0, item0
        self._gram_ready = [False, False]
        if self.world > 1:
            self._plan_host_exchange()
        return False

    def close(self) -> None:
        """Destroys the communicator this trainer created (idempotent)."""
        if self.native:  # (a host-loop trainer on the same solver must not close another trainer's transport)
            self.local.close_comm()
            self.native = False
        if self._owns_gram_group and self.gram_group is not None:
            try:
                self.dist.destroy_process_group(self.gram_group)
            except Exception:  # the default group may already be gone at interpreter exit
                pass
        self._owns_gram_group = False
        self.gram_group = self.group

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- phase timing ------------------------------------------------------------------
    def _mark(self, label: str) -> None:
        if not self.timing:
            return
        view = self.local.factor_view(0)
        if view.device.type == "cuda":
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record(self.torch.cuda.current_stream(view.device))
            self._marks.append((label, ev))
        else:
            import time

            self._marks.append((label, time.perf_counter()))

    def last_timing(self) -> dict:
        """ms per phase summed over the marks since the last call: ``gramian_ms`` (own-row
        partial sum + finish), ``allreduce_ms``, ``solve_ms``, ``allgather_ms`` (the row
        exchange, with the overlapped all-reduce inside it when ``overlap``),
        ``exposed_comm_ms`` = all-reduce + exchange time on the critical path, ``total_ms``."""
        out = {"gramian_ms": 0.0, "allreduce_ms": 0.0, "solve_ms": 0.0, "allgather_ms": 0.0}
        marks, self._marks = self._marks, []
        if len(marks) < 2:
            out.update(exposed_comm_ms=0.0, total_ms=0.0)
            return out
        if not isinstance(marks[0][1], float):
            marks[-1][1].synchronize()
        for (_, a), (label, b) in zip(marks, marks[1:]):
            dt = (b - a) * 1e3 if isinstance(a, float) else a.elapsed_time(b)
            if label in out:
                out[label] += dt
        out["exposed_comm_ms"] = out["allreduce_ms"] + out["allgather_ms"]
        out["total_ms"] = sum(out[k] for k in ("gramian_ms", "allreduce_ms", "solve_ms", "allgather_ms"))
        return out

    def _probe_gather(self, like) -> bool:
        """Can this backend all-gather tensors on the factors' device in ONE call?  Decided
        once, by a collective every rank executes (so the ranks cannot disagree), not by
        catching errors on the hot path.  RCCL / NCCL and gloo-on-CPU can; gloo cannot for
        device tensors."""
        torch, dist = self.torch, self.dist
        backend = str(dist.get_backend(self.group)).lower()
        if "nccl" in backend:
            return True
        if like.device.type == "cpu":
            out = torch.zeros(self.world, dtype=torch.float32)
            dist.all_gather_into_tensor(out, out[self.rank:self.rank + 1].clone(), group=self.group)
            return True
        return False

    def _exchange_rows(self, side: int):
        """Start moving the rows this rank has just solved into every replica; returns the
        pending work items (waited for by the caller)."""
        dist, local = self.dist, self.local
        b, view = self.bounds[side], local.factor_view(side)
        how = self.exchange[side]
        if how == "inplace":  # equal blocks of the padded buffer: input = own block of output
            S = view.shape[0] // self.world
            return [dist.all_gather_into_tensor(view, view[self.rank * S:(self.rank + 1) * S],
                                                group=self.group, async_op=True)]
        if how == "padded":  # uneven, cost-balanced shards: one collective through staging rows
            stage = self._stage[side]
            S = stage.shape[0] // self.world
            mine = stage[self.rank * S:(self.rank + 1) * S]
            n = b[self.rank + 1] - b[self.rank]
            mine[:n].copy_(view[b[self.rank]:b[self.rank + 1]])
            return [dist.all_gather_into_tensor(stage, mine, group=self.group, async_op=True)]
        works = []  # "broadcast": one per rank (backends without a device all-gather)
        for r in range(self.world):
            if b[r + 1] > b[r]:
                works.append(dist.broadcast(view[b[r]:b[r + 1]], src=self._global_rank(r),
                                            group=self.group, async_op=True))
        return works

    def _finish_exchange(self, side: int, works) -> None:
        for w in works:
            w.wait()
        if self.exchange[side] == "padded":
            b, view, stage = self.bounds[side], self.local.factor_view(side), self._stage[side]
            S = stage.shape[0] // self.world
            for r in range(self.world):
                if r != self.rank and b[r + 1] > b[r]:
                    view[b[r]:b[r + 1]].copy_(stage[r * S:r * S + (b[r + 1] - b[r])])

    def _reduce_gramian(self, side: int, async_op: bool):
        self.local.partial_gramian(side)
        if not async_op:
            self._mark("gramian_ms")
        if self.world > 1:
            return self.dist.all_reduce(self.local.gramian_view(side), op=self.dist.ReduceOp.SUM,
                                        group=self.gram_group, async_op=async_op)
        return None

    def half_epoch(self, side: int, solver_config) -> None:
        local = self.local
        if self.timing and not self._marks:
            self._mark("start")
        # (1) Gramian of the other side: own rows, summed over ranks (K x K, latency bound) -
        #     unless the previous half-epoch already prefetched it
        if self.world == 1 and hasattr(local, "gramian"):
            # nothing to sum over ranks: reduction and scaling in one launch (irs_ials_gramian_async)
            local.gramian(side)
            self._mark("gramian_ms")
            self._mark("allreduce_ms")
        else:
            if not self._gram_ready[side]:
                self._reduce_gramian(side, async_op=False)
                self._mark("allreduce_ms")
            self._gram_ready[side] = False
            local.finish_gramian(side)
            self._mark("gramian_ms")
        # (2) solve this rank's rows of `side`
        local.half_step(side, solver_config)
        self._mark("solve_ms")
        if self.world == 1:
            return
        # (3) all-gather of the freshly solved rows; meanwhile (4) the next half-epoch's
        #     Gramian, which needs only those rows: partial sum + all-reduce on the second
        #     communicator
        works = self._exchange_rows(side)
        if self.overlap:
            red = self._reduce_gramian(1 - side, async_op=True)
            self._finish_exchange(side, works)
            red.wait()
            self._gram_ready[1 - side] = True
        else:
            self._finish_exchange(side, works)
        self._mark("allgather_ms")

    def invalidate(self) -> None:
        """Call after changing the factors from outside (the prefetched Gramian is stale)."""
        self._gram_ready = [False, False]

    def _global_rank(self, group_rank: int) -> int:
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def step(self, solver_config) -> None:
        """One epoch: user half then item half (hpp:784-788)."""
        if self.native:
            self.local.sharded_step(solver_config, self.bounds[0], self.bounds[1])
            return
        self.half_epoch(0, solver_config)
        self.half_epoch(1, solver_config)

    def synchronize(self) -> None:
        self.local.synchronize()


# ---------------------------------------------------------------------------------------
# kNN and evaluator: independent units, no data-path collective

def similarity_row_bounds(X_target: sps.spmatrix, world: int) -> List[int]:
    """Contiguous target-row ranges of ~equal multiply-add count for ``sharded_similarity``:
    row i of ``X_target @ X_target.T`` (knn.hpp:43-83: the rows one thread / rank computes)
    costs sum over its stored features f of nnz(column f) - a popular item's row is orders of
    magnitude dearer than a tail item's, so equal ROW counts are balanced only when the row
    order is random."""
    Xr = sps.csr_matrix(X_target)
    col_nnz = np.bincount(Xr.indices, minlength=Xr.shape[1]).astype(np.float64)
    per_entry = col_nnz[Xr.indices]
    cost = np.add.reduceat(np.concatenate([per_entry, [0.0]]), Xr.indptr[:-1].astype(np.int64))
    cost[np.diff(Xr.indptr) == 0] = 0.0  # (reduceat repeats the next row's first entry on empty rows)
    return balanced_bounds(cost + 1.0, world)


def _all_gather_bytes(buf: np.ndarray, group=None) -> List[np.ndarray]:
    """Every rank's byte buffer on every rank: one all_gather of the lengths, one of the buffers padded to
    the longest (tensor collectives - on the device with the nccl backend, on the host with gloo; no
    pickling: all_gather_object of the ML-20M item-kNN result's blocks cost 54 ms at 4 ranks, 7 x the
    compute it followed)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    n = torch.tensor([buf.size], dtype=torch.int64, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = [int(v) for v in sizes.cpu().tolist()]
    width = max(max(sizes), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=dev)
    if buf.size:
        mine[:buf.size] = torch.from_numpy(buf).to(dev)
    everyone = torch.empty(world * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    host = everyone.cpu().numpy()
    return [host[r * width:r * width + sizes[r]] for r in range(world)]


def sharded_similarity(compute_rows, n_rows: int, group=None,
                       bounds: Sequence[int] = None) -> sps.csr_matrix:
    """Item- / user-kNN over ``world`` ranks: rank r computes the target rows
    ``bounds[r : r + 2]`` (default ``even_bounds(n_rows, world)``; ``similarity_row_bounds``
    balances the multiply-adds instead of the row count) with ``compute_rows(begin, end)`` (a CSR
    block, e.g. ``computer.compute_similarity(X, top_k, rows=(begin, end))``); the blocks travel as
    one byte buffer per rank (row lengths, column ids, values) through two tensor all-gathers and are
    laid end to end in rank order on every rank - the arrays of ``scipy.sparse.vstack(blocks)``."""
    rank, world = _group_info(group)
    b = even_bounds(n_rows, world) if bounds is None else [int(v) for v in bounds]
    if len(b) != world + 1 or b[0] != 0 or b[-1] != n_rows or any(x > y for x, y in zip(b, b[1:])):
        raise ValueError("bounds must be world + 1 non-decreasing row offsets from 0 to n_rows.")
    mine = sps.csr_matrix(compute_rows(b[rank], b[rank + 1]))
    if world == 1:
        return mine
    if mine.shape[0] != b[rank + 1] - b[rank]:
        raise ValueError("compute_rows(begin, end) must return end - begin rows.")
    lens = np.diff(mine.indptr).astype(np.int64)
    head = np.array([mine.shape[1], int(mine.has_sorted_indices)], dtype=np.int64)
    buf = np.concatenate([head.view(np.uint8), lens.view(np.uint8),
                          np.ascontiguousarray(mine.indices, dtype=np.int32).view(np.uint8),
                          np.ascontiguousarray(mine.data, dtype=np.float64).view(np.uint8)])
    parts = _all_gather_bytes(buf, group)
    all_lens, all_idx, all_val, n_cols, is_sorted = [], [], [], mine.shape[1], True
    for r, part in enumerate(parts):
        rows_r = b[r + 1] - b[r]
        hd = part[:16].view(np.int64)
        if int(hd[0]) != n_cols:
            raise ValueError("the ranks' blocks have different column counts.")
        is_sorted = is_sorted and bool(hd[1])
        ln = part[16:16 + 8 * rows_r].view(np.int64)
        nnz_r = int(ln.sum())
        o = 16 + 8 * rows_r
        all_lens.append(ln)
        all_idx.append(part[o:o + 4 * nnz_r].view(np.int32))
        all_val.append(part[o + 4 * nnz_r:o + 12 * nnz_r].view(np.float64))
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.concatenate(all_lens), out=indptr[1:])
    indices = np.concatenate(all_idx)
    if indptr[-1] < 2 ** 31:
        indptr = indptr.astype(np.int32)
    out = sps.csr_matrix((np.concatenate(all_val), indices, indptr), shape=(n_rows, n_cols))
    out.has_sorted_indices = is_sorted
    return out


def sharded_metrics(evaluate_users, n_users: int, merged, group=None):
    """Evaluator over ``world`` ranks: rank r evaluates the users
    ``even_bounds(n_users, world)[r : r + 2]`` with ``evaluate_users(begin, end)`` (a
    ``Metrics``); ``merged`` (an empty ``Metrics``) receives the sum of all ranks' results in
    rank order (``Metrics::merge``, evaluator.cpp:76-85) on every rank."""
    import torch.distributed as dist

    rank, world = _group_info(group)
    b = even_bounds(n_users, world)
    mine = evaluate_users(b[rank], b[rank + 1])
    if world == 1:
        merged.merge(mine)
        return merged
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    for part in parts:
        merged.merge(part)
    return merged
