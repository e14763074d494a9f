"""Multi-GPU host loop of the iALS step: one process per GPU, rows sharded.

The reference has no distributed layer; this is the natural
sharding of ``IALSTrainer::step`` (IALSTrainer.hpp:758-789): the per-row solves
are independent, so users (items on the alternate half-epoch) are split into
contiguous, cost-balanced ranges; every rank keeps full replicas of both
factor matrices and, per half-epoch, takes part in

  1. an all-reduce of the K x K partial Gramian of its own rows
     (``Solver::prepare_p``, hpp:78-115, summed over ranks), and
  2. an all-gather of the freshly solved factor rows: ONE in-place
     ``all_gather_into_tensor`` when the shards are equal row blocks of the (row padded)
     factor buffer (``equal_bounds``; what ``bench.py`` uses - with RCCL over xGMI every
     shard leaves on its own links), else one broadcast per rank (cost-balanced uneven
     shards, ``shard_bounds``).

Collectives go through ``torch.distributed`` (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests) and run IN PLACE on torch views of the solver's own
buffers (zero-copy: the HIP library's device memory is wrapped through
``__cuda_array_interface__``), so no staging copies sit between the kernels and
the wire.  The arithmetic is behind a small ``LocalSolver`` interface whose
product implementation (``HipLocalSolver``) drives the HIP library.  Results equal
the single-GPU ones up to the summation order of the Gramian.
"""

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


class HipLocalSolver(LocalSolver):
    """Product implementation: the HIP trainer of this rank's GPU, on torch's current stream."""

    def __init__(self, model_config, X, shard: Tuple[int, int, int, int], device: int):
        import torch

        from .recommenders._ials_core import IALSTrainer

        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.trainer = IALSTrainer(model_config, X, device=device, shard=shard)
        self.trainer.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self._views = {}
        for which in range(4):
            ptr, rows, ld = self.trainer.device_buffer(which)
            if rows * ld == 0:
                self._views[which] = torch.empty((rows, ld), dtype=torch.float32, device=self.device)
            else:
                self._views[which] = torch.as_tensor(_DeviceArray(ptr, rows, ld), device=self.device)

    def factor_view(self, which: int):
        return self._views[which]

    def gramian_view(self, side: int):
        return self._views[2 + side]

    def partial_gramian(self, side: int) -> None:
        self.trainer.partial_gramian_async(side)

    def finish_gramian(self, side: int) -> None:
        self.trainer.finish_gramian_async(side)

    def half_step(self, side: int, solver_config) -> None:
        self.trainer.half_step_async(side, solver_config)

    def synchronize(self) -> None:
        self.trainer.synchronize()


class ShardedIALSTrainer:
    """``IALSTrainer.step`` over ``world_size`` ranks (see the module docstring)."""

    def __init__(self, local: LocalSolver, user_bounds: Sequence[int], item_bounds: Sequence[int],
                 group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.local = local
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:  # single process: no collectives are issued
            self.rank, self.world = 0, 1
        assert len(user_bounds) == self.world + 1 and len(item_bounds) == self.world + 1
        self.bounds = (list(user_bounds), list(item_bounds))
        self._gather_ok = True

    def half_epoch(self, side: int, solver_config) -> None:
        dist, local = self.dist, self.local
        # (1) Gramian of the other side: own rows, then sum over ranks (K x K, latency bound)
        local.partial_gramian(side)
        if self.world > 1:
            dist.all_reduce(local.gramian_view(side), op=dist.ReduceOp.SUM, group=self.group)
        local.finish_gramian(side)
        # (2) solve this rank's rows of `side`
        local.half_step(side, solver_config)
        # (3) all-gather of the freshly solved rows, in place on the view of the factor matrix
        if self.world > 1:
            b = self.bounds[side]
            view = local.factor_view(side)
            S = view.shape[0] // self.world
            equal = (view.shape[0] % self.world == 0 and
                     all(b[r] == min(r * S, b[-1]) for r in range(self.world)))
            if equal and self._gather_ok:
                # equal blocks of the padded buffer: one collective, input = own block of output
                try:
                    dist.all_gather_into_tensor(view, view[self.rank * S:(self.rank + 1) * S],
                                                group=self.group)
                    return
                except (RuntimeError, NotImplementedError):
                    self._gather_ok = False  # backend without it (gloo on device tensors)
            # uneven shards: every rank broadcasts its rows straight into the replicas
            works = []
            for r in range(self.world):
                if b[r + 1] > b[r]:
                    works.append(dist.broadcast(view[b[r]:b[r + 1]], src=self._global_rank(r),
                                                group=self.group, async_op=True))
            for w in works:
                w.wait()

    def _global_rank(self, group_rank: int) -> int:
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def step(self, solver_config) -> None:
        """One epoch: user half then item half (hpp:784-788)."""
        self.half_epoch(0, solver_config)
        self.half_epoch(1, solver_config)

    def synchronize(self) -> None:
        self.local.synchronize()
