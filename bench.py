"""bench.py — iALS user+item updates/sec at k=64 on the MovieLens-20M-shaped synthetic CSR.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON
line from rank 0.  A "step" is one ``IALSTrainer.step`` epoch (both Gramians, the
user half-epoch and the item half-epoch; IALSTrainer.hpp:784-788) over the
HBM-resident matrix.  For N > 1 the driver launches one process per GPU with
``torch.distributed.run``; rows are sharded (``irspack_amd.sharding``) and the
same matrix is solved by all ranks together (strong scaling), with an RCCL
all-reduce of the K x K Gramian and an in-place all-gather of the solved factor
shards every half-epoch.

Extra objects on the line: ``roofline`` for the dominant kernel (HIP-event timed
inside the library on the launch stream) and ``cpu_baseline`` (the CPU oracle —
a restatement of the reference's Eigen path, which cannot be built offline —
timed on this box's host cores on a bounded row sample, N = 1 only).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 MFMA / vector peak (spec)
PEAK_HBM_GBS = 8000.0    # HBM3E spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default="ml20m")
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--solver", default="CHOLESKY", choices=["CHOLESKY", "CG"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the kNN / evaluator figures reported next to the headline metric")
    return ap.parse_args()


def algorithmic_half(nnz, rows, n_other, K, solver, cg_steps=3):
    """Algorithmic flops / bytes of one half-epoch (SURVEY.md §8d with
    the rank update counted on the upper triangle like Eigen's selfadjoint rankUpdate)."""
    if solver == "CHOLESKY":
        flops = nnz * (K * (K + 1) + 2 * K) + rows * (K ** 3 / 3.0 + 2 * K * K)
    else:
        flops = nnz * (cg_steps + 2) * 4 * K + rows * (cg_steps + 1) * 2 * K * K
    byts = nnz * (4 + 4 + 4 * K) + rows * (4 * K + 4) + n_other * 4 * K + 4 * K * K
    if solver == "CG":
        byts += rows * 4 * K
    return float(flops), float(byts)


def cpu_baseline(X, K, solver, budget_s):
    """Oracle (CPU restatement of the Eigen path) on a bounded row sample, all host cores."""
    import oracle as O

    cores = os.cpu_count() or 1
    mc = O.model_config(K, alpha0=0.1, reg=1e-3, nu=1.0, init_stdev=0.1, random_seed=42)
    sc = O.solver_config(cores, solver, 3)
    U, I = X.shape
    rng = np.random.default_rng(0)
    user = (rng.standard_normal((U, K)) * 0.1).astype(np.float32)
    item = (rng.standard_normal((I, K)) * 0.1).astype(np.float32)
    Xt = X.T.tocsr()
    Xt.sort_indices()
    P_u = O.ials_gramian(item, 0.1, cores)
    P_i = O.ials_gramian(user, 0.1, cores)
    # calibrate on 1/64 of the rows, then size the sample for ~budget_s
    frac = 1.0 / 64
    t0 = time.perf_counter()
    O.ials_solver_step(user, X, item, P_u, mc, sc, 0, max(1, int(U * frac)))
    O.ials_solver_step(item, Xt, user, P_i, mc, sc, 0, max(1, int(I * frac)))
    cal = time.perf_counter() - t0
    frac = float(min(1.0, max(frac, frac * budget_s / max(cal, 1e-3))))
    nu_, ni_ = max(1, int(U * frac)), max(1, int(I * frac))
    t0 = time.perf_counter()
    O.ials_solver_step(user, X, item, P_u, mc, sc, 0, nu_)
    O.ials_solver_step(item, Xt, user, P_i, mc, sc, 0, ni_)
    dt = time.perf_counter() - t0
    return {
        "value": (nu_ + ni_) / dt,
        "unit": "updates/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"first {nu_} of {U} user rows + first {ni_} of {I} item rows of the same "
                   f"matrix (row order is random), one {solver} half-step each, {dt:.1f} s, "
                   f"{cores} threads; Gramians excluded"),
    }


def secondary_metrics(X, trainer, K):
    """The other two pieces of the hot path at the same ML-20M shape (reported, not the
    headline): item-kNN (cosine, top_k = 100, BASELINE.json configs[2]) and the fused
    score + nDCG@20 evaluator.  Kernel times are HIP-event timed inside the library."""
    import scipy.sparse as sps

    from irspack_amd.evaluation._core_evaluator import EvaluatorCore
    from irspack_amd.recommenders._knn import CosineSimilarityComputer, JaccardSimilarityComputer

    U, I = X.shape
    out = {}
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    Xt.data[:] = 1.0
    comp = CosineSimilarityComputer(Xt, 0.0, True)
    comp.compute_similarity(Xt, 100, rows=(0, 64))  # warm-up
    t0 = time.perf_counter()
    S = comp.compute_similarity(Xt, 100)
    wall = time.perf_counter() - t0
    ms = comp.last_kernel_ms
    bytes_per_mac = 4.0  # int32 column id; the all-ones value stream is not read
    out["knn"] = {
        "workload": f"cosine item-kNN top_k=100, {I} items x {U} users, binary interactions, fp64",
        "kernel_ms": ms, "wall_s_incl_pcie": wall, "macs": comp.last_macs,
        "item_pairs_per_s": I * float(I) / (ms * 1e-3),
        "gmacs_per_s": comp.last_macs / ms / 1e6,
        "roofline": {"bound": "hbm", "achieved": comp.last_macs * bytes_per_mac / (ms * 1e-3) / 1e9,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": comp.last_macs * bytes_per_mac / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     "traffic": None},
        "out_nnz": int(S.nnz),
    }
    # the other two SURVEY 8(d) variants, kernel time only
    variants = {}
    for name, other in (("cosine_normalize_false", CosineSimilarityComputer(Xt, 0.0, False)),
                        ("jaccard", JaccardSimilarityComputer(Xt, 0.0))):
        other.compute_similarity(Xt, 100)
        variants[name] = {"kernel_ms": other.last_kernel_ms,
                          "item_pairs_per_s": I * float(I) / (other.last_kernel_ms * 1e-3)}
        del other
    out["knn"]["variants"] = variants
    # evaluator: hold out one interaction per user as ground truth, mask the rest
    rng = np.random.default_rng(5)
    pick = X.indptr[:-1] + (rng.random(U) * np.diff(X.indptr)).astype(np.int64)
    gt = sps.csr_matrix((np.ones(U), (np.arange(U), X.indices[pick])), shape=X.shape)
    ev = EvaluatorCore(gt, [])
    keep = np.ones(X.nnz, dtype=bool)
    keep[pick] = False
    rows = np.repeat(np.arange(U), np.diff(X.indptr))
    mask = sps.csr_matrix((np.ones(int(keep.sum()), dtype=np.float32),
                           (rows[keep], X.indices[keep])), shape=X.shape)
    ev.get_metrics_ials(trainer, 0, 2048, mask[:2048], 20, 0, False)  # warm-up
    t0 = time.perf_counter()
    m = ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
    wall_first = time.perf_counter() - t0  # converts and uploads the mask (80 MB)
    t0 = time.perf_counter()
    m = ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
    wall = time.perf_counter() - t0        # the mask is resident, as in a tuning loop
    out["evaluator"] = {
        "workload": f"fused iALS k={K} scoring + nDCG@20 over {U} users x {I} items, fp32 scores",
        "wall_s_first_call_incl_mask_upload": wall_first,
        "wall_s_incl_pcie": wall, "users_per_s": U / wall,
        "scores_per_s": U * float(I) / wall, "ndcg@20": m.as_dict()["ndcg"],
    }
    return out


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (one-GPU boxes): IRSPACK_AMD_BENCH_BACKEND=gloo and
    # IRSPACK_AMD_BENCH_ONE_DEVICE=1 run the N > 1 control flow with every rank on cuda:0
    backend = os.environ.get("IRSPACK_AMD_BENCH_BACKEND", "nccl")
    if os.environ.get("IRSPACK_AMD_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback).")
    torch.cuda.set_device(local_rank)

    from irspack_amd.recommenders._ials_core import (IALSModelConfigBuilder,
                                                      IALSSolverConfigBuilder, SolverType)
    from irspack_amd.sharding import HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds
    from irspack_amd.synthetic import describe, make_interactions

    X = make_interactions(args.shape)  # identical on every rank (seeded)
    info = describe(X)
    U, I = X.shape
    K = args.K
    mc = (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-3).set_nu(1.0)
          .set_init_stdev(0.1).set_random_seed(42).build())
    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[args.solver])
          .set_max_cg_steps(3).build())
    # equal row blocks (random row order: cost-balanced to a few per cent) so that the solved
    # rows travel in ONE in-place all-gather per half-epoch
    ub, ib = equal_shard_bounds(X, world)
    shard = (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1])
    local = HipLocalSolver(mc, X, shard, local_rank)
    trainer = ShardedIALSTrainer(local, ub, ib)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(sc)
    trainer.synchronize()
    local.trainer.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(sc)
    trainer.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = local.trainer.profile_read()
    local.trainer.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        value = (U + I) * args.steps / elapsed
        # dominant kernel of this rank and its roofline
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else (None, None)
        roofline = None
        if dom[0] is not None:
            name, st = dom
            side = 0 if name.endswith("_user") else 1
            rows = shard[1] - shard[0] if side == 0 else shard[3] - shard[2]
            Xs = X if side == 0 else X.T.tocsr()
            b, e = (shard[0], shard[1]) if side == 0 else (shard[2], shard[3])
            nnz_side = int(Xs.indptr[e] - Xs.indptr[b])
            n_other = I if side == 0 else U
            flops, byts = algorithmic_half(nnz_side, rows, n_other, K, args.solver)
            t_launch = st["ms"] / st["launches"] * 1e-3
            if args.solver == "CHOLESKY":
                ach = flops / t_launch / 1e12
                roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_TFLOPS,
                            "unit": "TFLOP/s", "frac": ach / PEAK_F32_TFLOPS, "traffic": None}
            else:
                ach = byts / t_launch / 1e9
                roofline = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                            "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None}
            roofline.update({
                "kernel": name, "avg_launch_ms": st["ms"] / st["launches"],
                "launches": st["launches"], "algorithmic_gflop_per_launch": flops / 1e9,
                "algorithmic_gbyte_per_launch": byts / 1e9,
                "hbm_side_gbs": byts / t_launch / 1e9,
                "f32_tflops": flops / t_launch / 1e12,
            })
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    roofline["traffic"] = json.load(open(pmc)).get(name)
                except Exception:
                    pass
        result = {
            "metric": "iALS user+item updates/sec at k=64, ML-20M-shape CSR",
            "value": value,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{args.shape}-shape synthetic CSR {U}x{I} nnz={info['nnz']}, "
                             f"iALS k={K} fp32, solver={args.solver}"
                             + (", max_cg_steps=3" if args.solver == "CG" else "")),
                "alpha0": 0.1, "reg": 1e-3, "nu": 1.0, "loss": "IALSPP",
                "sharding": f"rows over {world} rank(s), replicated factors",
            },
            "kernels_ms_per_launch": {k: round(v["ms"] / v["launches"], 4) for k, v in prof.items()},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(X, K, args.solver, args.cpu_seconds)
        if world == 1 and not args.no_secondary and K <= 64:
            result["secondary"] = secondary_metrics(X, local.trainer, K)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
