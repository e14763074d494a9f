"""bench.py — iALS user+item updates/sec at k=64 on the MovieLens-20M-shaped synthetic CSR.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON
line from rank 0.  A "step" is one ``IALSTrainer.step`` epoch (both Gramians, the
user half-epoch and the item half-epoch; IALSTrainer.hpp:784-788) over the
HBM-resident matrix.  For N > 1 the driver launches one process per GPU with
``torch.distributed.run``; rows are sharded (``irspack_amd.sharding``) and the
same matrix is solved by all ranks together (strong scaling), with an RCCL
all-reduce of the K x K Gramian and an in-place all-gather of the solved factor
shards every half-epoch.

Objects on the line:
  roofline      dominant kernel of the headline (Cholesky) run, HIP-event timed inside the
                library on the launch stream; both the flop side and the HBM side
  ceilings      copy / triad GB/s, pure fp32-MFMA TF and LDS-atomic rate MEASURED in this run,
                next to the spec peaks (frac_of_spec / frac_of_measured in every roofline)
  cpu_baseline  the CPU oracle (a restatement of the reference's Eigen path, which cannot be
                built offline) on this box's host cores, bounded row sample, N = 1 only
  secondary     N = 1 only, each leg time-boxed and independent:
    ials_cg       the reference's DEFAULT solver (CG, 3 steps) on the same matrix
    ials_f32_mfma the Cholesky epoch again with the rank update on the fp32-input matrix instruction
                  (the headline path until round 5; IRSPACK_AMD_IALS_BF16X3=0)
    knn           cosine / jaccard item-kNN top-100 (configs[2]); headline = wall-inclusive call
    evaluator     fused score + nDCG@20 over all users (K = 64)
    fit           IALSTrainer(...) + 16 steps + factors to the host: the reference's learn() through the boundary
    k256          configs[4] on one GPU: K = 256 Cholesky + CG epochs and the fused nDCG@20
    c4            configs[3] shape on one GPU (10 M x 1 M, 95 M stored entries), K = 128, CG + Cholesky
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
                    help="skip every leg reported next to the headline metric")
    ap.add_argument("--legs", default="ials_cg,ials_f32_mfma,fit,knn,evaluator,k256,c4",
                    help="comma-separated secondary legs to run (N = 1)")
    ap.add_argument("--balance", default="auto", choices=["auto", "cost", "equal"],
                    help="N > 1 row shards: equal row blocks (one in-place all-gather) or "
                         "cost-balanced ranges (nnz (K^2 + 2K) + K^3 / 6 per row; padded exchange); "
                         "auto = cost when the longest row holds more than 1 %% of the entries")
    ap.add_argument("--c4-small", action="store_true",
                    help="run the c4 leg on the 1/5-scale matrix of the same generator instead of "
                         "the full 10 M x 1 M one (~20 s of host generation + construction)")
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


def algorithmic_epoch(X, K, solver):
    U, I = X.shape
    fu, bu = algorithmic_half(X.nnz, U, I, K, solver)
    fi, bi = algorithmic_half(X.nnz, I, U, K, solver)
    return fu + fi, bu + bi


def executed_cg_flops(X, K, cg_steps=3, short_max=32):
    """What the CG kernels EXECUTE per epoch: at K <= 128 rows above `short_max`
    stored entries build the explicit K x K system on the matrix cores (the rank update of the
    Cholesky path) and iterate on it with dense mat-vecs; rows up to `short_max` entries and
    every row at K > 128 (round 4: ials_mf_kernels.hpp; K > 256: gk_cg_kernel) run the
    matrix-free form the algorithmic count prices.  IRSPACK_AMD_IALS_MF=0 puts 128 < K <= 256
    back on the explicit build."""
    total = 0.0
    matrix_free_above = 128 if os.environ.get("IRSPACK_AMD_IALS_MF", "1") != "0" else 256
    for Xs in (X, X.T.tocsr()):
        nnz_r = np.diff(Xs.indptr).astype(np.float64)
        if K > matrix_free_above:
            total += float((nnz_r * (cg_steps + 2) * 4 * K + (cg_steps + 1) * 2.0 * K * K).sum())
            continue
        dense = nnz_r > short_max
        total += float((nnz_r[dense] * (K * (K + 1) + 2 * K) + (cg_steps + 1) * 2.0 * K * K).sum())
        total += float((nnz_r[~dense] * (cg_steps + 2) * 4 * K + (cg_steps + 1) * 2.0 * K * K).sum())
    return total


def executed_cholesky_flops(X, K, eig_sides, short_max=32):
    """What the Cholesky kernels EXECUTE per epoch.  Rows above `short_max` stored entries (and
    every row of a side that did not take the eigenbasis path) build and factorise the K x K
    system: the algorithmic count.  On a side in `eig_sides` (K <= 128) the rows of
    at most `short_max` entries are solved in the low-rank (Woodbury) form in the eigenbasis of the
    Gramian - per row of n entries: S = V~ D V~^T (n^2 K), its n x n factorisation (n^3 / 3), two
    n x K products (4 n K) - plus, once per half-step, the table product V~ = V Q (2 n_other K^2)
    and the rotation of the solved short rows back (2 R_short K^2); the float64 Jacobi
    decomposition (one workgroup) is not counted."""
    total = 0.0
    for side, Xs in enumerate((X, X.T.tocsr())):
        n = np.diff(Xs.indptr).astype(np.float64)
        dense = np.ones(n.shape, dtype=bool) if side not in eig_sides else n > short_max
        total += float((n[dense] * (K * (K + 1) + 2 * K) + K ** 3 / 3.0 + 2.0 * K * K).sum())
        s = n[~dense]
        total += float((s * s * K + s ** 3 / 3.0 + 4.0 * s * K).sum())
        if side in eig_sides:
            total += 2.0 * Xs.shape[1] * K * K + 2.0 * float((~dense).sum()) * K * K
    return total


def eigenbasis_sides(trainer, sc):
    """sides (0 user, 1 item) whose half-step takes the eigenbasis short-row path (a diagnostic of
    the library; one extra half-step per side, untimed)"""
    sides = []
    for side in (0, 1):
        trainer.partial_gramian_async(side)
        trainer.finish_gramian_async(side)
        trainer.half_step_async(side, sc)
        trainer.synchronize()
        if trainer.last_half_step_used_eigenbasis():
            sides.append(side)
    return sides


def roof(bound, achieved, ceilings, **more):
    """roofline object with the spec peak and the ceiling measured in this run"""
    if bound == "mfma":
        peak, unit, measured = PEAK_F32_TFLOPS, "TFLOP/s", (ceilings or {}).get("mfma_f32_tflops")
    else:
        peak, unit, measured = PEAK_HBM_GBS, "GB/s", (ceilings or {}).get("copy_gbs")
    r = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
         "frac_of_spec": achieved / peak,
         "frac_of_measured": (achieved / measured) if measured else None, "traffic": None}
    r.update(more)
    return r


def both_terms(flops, byts, seconds, ceilings, bound=None, **more):
    """BOTH terms shown.  `bound` names the roof that binds by the analysis of
    DESIGN.md 3.1 (Cholesky: fp32 matrix / vector issue - the gather is cache-served, PMC
    traffic is a fifth of the algorithmic bytes; CG: the gather side); None picks the larger
    fraction."""
    tf, gbs = flops / seconds / 1e12, byts / seconds / 1e9
    if bound is None:
        bound = "mfma" if tf / PEAK_F32_TFLOPS >= gbs / PEAK_HBM_GBS else "hbm"
    return roof(bound, tf if bound == "mfma" else gbs, ceilings, f32_tflops=tf,
                frac_f32=tf / PEAK_F32_TFLOPS, hbm_side_gbs=gbs, frac_hbm=gbs / PEAK_HBM_GBS,
                **more)


def cpu_baseline(X, K, solver, budget_s):
    """The CPU restatement of the Eigen path on a bounded row sample, all host cores, in two
    builds of the same sources: `value` is the TUNED build (oracle/_fast: -O3 -march=native
    -ffp-contract=fast, register-blocked rank update, compiled on this box by `make -C oracle
    fast`) - the fair baseline; `parity_oracle_value` is the parity oracle (x86-64-v3,
    -ffp-contract=off, plain loops), which is a checker, not a fast CPU program."""
    import oracle as O

    cores = os.cpu_count() or 1
    mc = O.model_config(K, alpha0=0.1, reg=1e-3, nu=1.0, init_stdev=0.1, random_seed=42)
    sc = O.solver_config(cores, solver, 3)
    scaling = None
    U, I = X.shape
    rng = np.random.default_rng(0)
    user = (rng.standard_normal((U, K)) * 0.1).astype(np.float32)
    item = (rng.standard_normal((I, K)) * 0.1).astype(np.float32)
    Xt = X.T.tocsr()
    Xt.sort_indices()
    P_u = O.ials_gramian(item, 0.1, cores)
    P_i = O.ials_gramian(user, 0.1, cores)

    def measure(budget):
        # calibrate on 1/64 of the rows, then size the sample for ~budget
        frac = 1.0 / 64
        t0 = time.perf_counter()
        O.ials_solver_step(user, X, item, P_u, mc, sc, 0, max(1, int(U * frac)))
        O.ials_solver_step(item, Xt, user, P_i, mc, sc, 0, max(1, int(I * frac)))
        cal = time.perf_counter() - t0
        frac = float(min(1.0, max(frac, frac * budget / max(cal, 1e-3))))
        nu_, ni_ = max(1, int(U * frac)), max(1, int(I * frac))
        reps, dt = 0, 0.0
        while reps < 1 or (dt < budget / 3 and reps < 8):  # whole passes; several when one is short
            t0 = time.perf_counter()
            O.ials_solver_step(user, X, item, P_u, mc, sc, 0, nu_)
            O.ials_solver_step(item, Xt, user, P_i, mc, sc, 0, ni_)
            dt += time.perf_counter() - t0
            reps += 1
        return (nu_ + ni_) * reps / dt, nu_, ni_, reps, dt

    parity, *_ = measure(budget_s / 3)
    fast_err = None
    try:
        O.use_fast_build()
        # The thread count that is fastest on THIS box, not "all of them": on the pool's 256-thread hosts the
        # row loop peaks at 32 threads (451 GFLOP/s, 14 per thread) and falls to 282 with 256 (round 6,
        # scripts/cpu_baseline_scaling.py) - memory system and SMT siblings, not the kernel (22 GFLOP/s on one).
        scaling = {}
        n_cal = max(1, min(U, 12000))
        for thr in sorted({min(cores, t) for t in (8, 16, 32, 64, 128, cores)}):
            sct = O.solver_config(thr, solver, 3)
            O.ials_solver_step(user, X, item, P_u, mc, sct, 0, min(n_cal, 2000))
            t0 = time.perf_counter()
            O.ials_solver_step(user, X, item, P_u, mc, sct, 0, n_cal)
            scaling[thr] = n_cal / (time.perf_counter() - t0)
        cores = max(scaling, key=scaling.get)
        sc = O.solver_config(cores, solver, 3)
        value, nu_, ni_, reps, dt = measure(budget_s)
        build = ("oracle/_fast/liboracle_fast.so: g++ -O3 -march=native -ffp-contract=fast -DORACLE_FAST "
                 "(register-blocked 4 x 16 rank update), built on this box")
    except Exception as exc:  # no compiler on the box: fall back to the parity build's figure
        fast_err = repr(exc)
        value, nu_, ni_, reps, dt = measure(budget_s)
        build = "oracle/liboracle.so (the parity build; the tuned build failed: " + fast_err + ")"
    finally:
        O.use_parity_build()
    flops, _ = algorithmic_epoch(X, K, solver)
    return {
        "value": value,
        "unit": "updates/s",
        "cores": cores,
        "kind": "port",
        "build": build,
        "cpu_gflops": value / (U + I) * flops / 1e9,
        "cpu_gflops_per_thread": value / (U + I) * flops / 1e9 / cores,
        "note": ("a restatement with a register-blocked 4 x 16 rank update (16 GFLOP/s on one core) and a "
                 "row-oriented LLT (4 GFLOP/s), one row per thread like the reference's loop, at the thread "
                 "count that is fastest on this box (`cores`; more threads are slower here: "
                 "`user_rows_per_s_by_thread_count`).  A reported baseline, not a target: the GPU / CPU ratio "
                 "says nothing about kernel quality - roofline.frac does."),
        "parity_oracle_value": parity,
        "host_threads_available": os.cpu_count(),
        "user_rows_per_s_by_thread_count": ({str(k): round(v, 1) for k, v in scaling.items()} if scaling else None),
        "sample": (f"first {nu_} of {U} user rows + first {ni_} of {I} item rows of the same "
                   f"matrix (row order is random), one {solver} half-step each, {reps} pass(es), "
                   f"{dt:.1f} s, {cores} threads; Gramians excluded"),
    }


PMC_WORKLOAD = "ml20m K=64"  # what the committed FETCH_SIZE / WRITE_SIZE passes ran (scripts/prof_bench.sh)
# the sources that define the kernels whose traffic profiles/pmc_traffic.json records
PMC_SOURCES = {"ials": ["ials.hip", "ials_kernels.hpp", "ials_chol16.hpp", "common.hpp"],
               "knn": ["knn.hip", "common.hpp"]}


def kernel_source_sha16():
    """sha256[:16] of the concatenated kernel sources per family: the key that ties a committed PMC figure
    to the code it was measured on (scripts/summarize_prof.py writes it, pmc_traffic() compares it)."""
    import hashlib

    out = {}
    for fam, files in PMC_SOURCES.items():
        h = hashlib.sha256()
        for f in files:
            with open(os.path.join(ROOT, "irspack_amd", "csrc", f), "rb") as fh:
                h.update(fh.read())
        out[fam] = h.hexdigest()[:16]
    return out


def pmc_traffic(workload=PMC_WORKLOAD, family="ials"):
    """HBM bytes from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    scripts/summarize_prof.py with the guide's gfx950 corrections): per LAUNCH for the ials_* kernels, per
    compute_similarity CALL for knn_tile_kernel.  PMC counters cannot be collected inside this process, so the
    figure is a committed one - and it is returned ONLY for the workload those passes ran AND while the
    kernels' sources hash to what the passes ran on (`_provenance.sources_sha16`); otherwise `traffic` is
    null rather than a number that no longer describes the code.  The line carries the provenance."""
    if workload != PMC_WORKLOAD:
        return {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}
    prov = d.get("_provenance") or {}
    if (prov.get("sources_sha16") or {}).get(family) != kernel_source_sha16()[family]:
        return {"_stale": {"reason": f"the {family} kernel sources changed since the PMC pass",
                           "pass": prov.get("collected_at_git_head"), "sources_now": kernel_source_sha16()[family],
                           "sources_at_pass": (prov.get("sources_sha16") or {}).get(family)}}
    return d


def time_epochs(trainer, sc, steps, warmup):
    """(seconds per epoch, per-kernel profile) of `steps` epochs of an unsharded IALSTrainer"""
    for _ in range(warmup):
        trainer.step(sc)
    trainer.synchronize()
    trainer.profile(2)  # (events on the dominant kernel only during the timed epochs: see main())
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(sc)
    trainer.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = trainer.profile_read()
    trainer.profile(True)
    trainer.step(sc)
    trainer.synchronize()
    for name, rec in trainer.profile_read().items():
        prof.setdefault(name, rec)
    trainer.profile(False)
    return dt, {k: round(v["ms"] / v["launches"], 4) for k, v in prof.items()}


def solver_config(kind):
    from irspack_amd.recommenders._ials_core import IALSSolverConfigBuilder, SolverType

    return (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType[kind])
            .set_max_cg_steps(3).build())


def model_config(K):
    from irspack_amd.recommenders._ials_core import IALSModelConfigBuilder

    return (IALSModelConfigBuilder().set_K(K).set_alpha0(0.1).set_reg(1e-3).set_nu(1.0)
            .set_init_stdev(0.1).set_random_seed(42).build())


def ials_leg(trainer, X, K, kind, steps, warmup, ceilings, workload=None):
    """One solver on one matrix: updates/s + roofline with both terms.  `workload`: "<shape> K=<K>" when
    a PMC pass of exactly this workload is committed (traffic_by_kernel is null otherwise)."""
    dt, kernels = time_epochs(trainer, solver_config(kind), steps, warmup)
    flops, byts = algorithmic_epoch(X, K, kind)
    U, I = X.shape
    more = {}
    priced = flops
    if kind == "CG":
        # The CG kernels of rows above 32 entries build the K x K system on the matrix cores and
        # iterate on it: the roof that binds them is the fp32 matrix / vector issue rate on the
        # flops they EXECUTE, not HBM on the matrix-free algorithmic bytes (which are
        # cache-served: round 2 priced them against HBM and got 1.11 of the measured ceiling).
        # Both counts are shown; `bound` is whichever fraction is larger.
        priced = executed_cg_flops(X, K)
        more = {"executed_gflop_per_epoch": priced / 1e9,
                "matrix_free_algorithmic_gflop_per_epoch": flops / 1e9,
                "executed_over_algorithmic_flops": priced / flops}
    else:
        # Cholesky: sides whose short rows are solved in the eigenbasis of the Gramian (configs[3])
        # execute FEWER flops than K^3 / 3 per row: price what runs, so that no fraction exceeds 1
        sides = eigenbasis_sides(trainer, solver_config(kind))
        if sides:
            priced = executed_cholesky_flops(X, K, sides)
            more = {"executed_gflop_per_epoch": priced / 1e9, "eigenbasis_sides": sides,
                    "executed_over_algorithmic_flops": priced / flops}
    return {
        "solver": kind + (" max_cg_steps=3" if kind == "CG" else ""),
        "ms_per_epoch": dt * 1e3, "updates_per_s": (U + I) / dt, "steps": steps, "warmup": warmup,
        "kernels_ms_per_launch": kernels,
        "roofline": both_terms(priced, byts, dt, ceilings,
                               # CG: when most of the executed flops are the explicit K x K build
                               # (rows above 32 entries) the kernel is issue-bound like Cholesky
                               bound="mfma" if kind == "CHOLESKY" or priced > 1.5 * flops else None,
                               scope="whole epoch (all kernels)",
                               algorithmic_gflop_per_epoch=flops / 1e9,
                               algorithmic_gbyte_per_epoch=byts / 1e9,
                               traffic_by_kernel=({k: v for k, v in pmc_traffic(workload).items()
                                                   if k.startswith("ials_") and ("cg" in k) == (kind == "CG")}
                                                  or None),
                               traffic_provenance=(pmc_traffic(workload).get("_provenance")
                                                   or pmc_traffic(workload).get("_stale")),
                               **more),
    }


def fit_leg(X, K, kind="CHOLESKY", epochs=16):
    """`fit` THROUGH THE BOUNDARY: what the reference's `IALSRecommender.learn()` does with the core
    (ials.py:92-131, base_earlystop.py:106-149): construct the trainer from the host CSR (copy,
    transpose, task lists, the libstdc++ random stream of the initial factors, uploads), 16 `step`s,
    both factor matrices back on the host.  Host clock around each part; the second of two fits (the
    first pays one-off allocations)."""
    from irspack_amd.recommenders._ials_core import IALSTrainer

    sc = solver_config(kind)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        tr = IALSTrainer(model_config(K), X)
        t1 = time.perf_counter()
        for _ in range(epochs):
            tr.step(sc)
        t2 = time.perf_counter()
        user, item = tr.user, tr.item
        t3 = time.perf_counter()
        best = {"solver": kind, "epochs": epochs, "fit_wall_s": t3 - t0, "create_s": t1 - t0,
                "epochs_s": t2 - t1, "factors_to_host_s": t3 - t2,
                "updates_per_s_through_the_boundary": (X.shape[0] + X.shape[1]) * epochs / (t3 - t0),
                "factor_bytes": int(user.nbytes + item.nbytes)}
        del tr, user, item
    return best


def ialspp_leg(trainer, X, K, steps, warmup):
    """iALS++ (the reference's subspace solver, hpp:387-630) with its default 64-dim blocks,
    one sweep per half-step.  Algorithmic flops of a half-step: per block the 64 x 64 system
    (nnz * 64 * 65 + R * (64^3 / 3 + 2 * 64^2)) plus the prediction work (nnz * 2 K once, nnz * 2 * 64
    per later block)."""
    from irspack_amd.recommenders._ials_core import IALSSolverConfigBuilder, SolverType

    sc = (IALSSolverConfigBuilder().set_n_threads(1).set_solver_type(SolverType.IALSPP)
          .set_ialspp_subspace_dimension(64).set_ialspp_iteration(1).build())
    dt, kernels = time_epochs(trainer, sc, steps, warmup)
    U, I = X.shape
    nb = -(-K // 64)
    flops = 0.0
    for R in (U, I):
        flops += nb * (X.nnz * 64.0 * 65.0 + R * (64.0 ** 3 / 3 + 2 * 64.0 ** 2))
        flops += X.nnz * 2.0 * K + (nb - 1) * X.nnz * 2.0 * 64
    return {"solver": "IALSPP subspace=64 iteration=1", "ms_per_epoch": dt * 1e3,
            "updates_per_s": (U + I) / dt, "steps": steps, "warmup": warmup,
            "kernels_ms_per_launch": kernels,
            "f32_tflops": flops / dt / 1e12, "frac_f32": flops / dt / 1e12 / PEAK_F32_TFLOPS,
            "algorithmic_gflop_per_epoch": flops / 1e9}


def f32_mfma_leg(X, K, steps, ceilings):
    """The same Cholesky epoch with the rank update on the fp32-INPUT matrix instruction
    (v_mfma_f32_16x16x4_f32; IRSPACK_AMD_IALS_BF16X3=0) - the headline path of rounds 1-5, beside the
    headline's default (round 6: the bf16 matrix cores on exact three-way splits of the fp32 values, six
    fp32-exact partial products per product, fp32 accumulate), so that the two rank updates can be compared
    on the driver's own box.  Both are priced against the fp32 peak on the same algorithmic flops."""
    from irspack_amd.recommenders._ials_core import IALSTrainer

    os.environ["IRSPACK_AMD_IALS_BF16X3"] = "0"
    try:
        tr = IALSTrainer(model_config(K), X)  # the switch is read when a trainer is created
    finally:
        del os.environ["IRSPACK_AMD_IALS_BF16X3"]
    out = ials_leg(tr, X, K, "CHOLESKY", steps, 2, ceilings)  # (the committed PMC pass is of the default path)
    out["rank_update"] = "fp32: v_mfma_f32_16x16x4_f32, one triangle of 16 x 16 tiles (IRSPACK_AMD_IALS_BF16X3=0)"
    return out


def holdout(X, seed=7):
    """evaluator inputs as SURVEY.md 8(d) names them: ground truth = a 20 % per-row hold-out of
    the matrix (own splitter, seed 7), mask = the remaining 80 % (the training entries)"""
    import scipy.sparse as sps

    from irspack_amd.synthetic import holdout_split

    train, test = holdout_split(X, 0.2, seed)
    gt = sps.csr_matrix(test, dtype=np.float64)
    mask = sps.csr_matrix(train, dtype=np.float32)
    return gt, mask


def block_path_leg(ev, trainer, mask, n_users=8192, mb_size=128):
    """The DROP-IN block path, timed on the first `n_users` users: what runs when a maintainer puts
    this library's EvaluatorCore / IALSTrainer under the reference's own Python Evaluator
    (INTEGRATION.md option A; evaluation/evaluator.py:417-438): per 128-user block
    ``get_score_block`` (scores to the host), ``scores[mask.nonzero()] = -inf`` in numpy,
    ``get_metrics_f32`` (block back to the device, ranked).  Beside it the same loop with this
    package's own block call (``get_metrics_masked``: the mask applied on the device copy)."""
    from irspack_amd.evaluation._core_evaluator import MaskRows, Metrics
    from irspack_amd.recommenders._ials_core import IALSSolverConfigBuilder

    sc = IALSSolverConfigBuilder().set_n_threads(1).build()
    n = min(n_users, ev.n_users)
    out = {}
    for name in ("reference_loop", "masked_call"):
        rows = MaskRows(mask, ev.n_items) if name == "masked_call" else None
        acc = Metrics(ev.n_items)
        t0 = time.perf_counter()
        for b in range(0, n, mb_size):
            e = min(b + mb_size, n)
            scores = trainer.user_scores(b, e, sc)
            if rows is None:
                scores[mask[b:e].nonzero()] = -np.inf
                acc.merge(ev.get_metrics_f32(scores, 20, b, 1, False))
            else:
                acc.merge(ev.get_metrics_masked(scores, rows, b, [20], b, 1, False)[0])
        dt = time.perf_counter() - t0
        out[name] = {"users": n, "mb_size": mb_size, "wall_s": dt, "users_per_s": n / dt,
                     "ndcg@20": acc.as_dict()["ndcg"]}
    return out


def evaluator_leg(X, trainer, K, ceilings):
    """Fused scoring + nDCG@20 over all users on a 20 % hold-out.  The scored model is fitted on
    the TRAINING entries only (three CG epochs; `trainer`, fitted on all of X, is not used: its
    scores would have seen the hold-out), so ndcg@20 is a held-out figure - of a synthetic matrix."""
    from irspack_amd.evaluation._core_evaluator import EvaluatorCore
    from irspack_amd.recommenders._ials_core import IALSTrainer

    U, I = X.shape
    gt, mask = holdout(X)
    trainer = IALSTrainer(model_config(K), mask)
    for _ in range(3):
        trainer.step(solver_config("CG"))
    ev = EvaluatorCore(gt, [])
    ev.get_metrics_ials(trainer, 0, 2048, mask[:2048], 20, 0, False)  # warm-up
    t0 = time.perf_counter()
    m = ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
    wall_first = time.perf_counter() - t0  # converts and uploads the mask (80 MB)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        m = ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
        walls.append(time.perf_counter() - t0)  # the mask is resident, as in a tuning loop
    wall = min(walls)
    # (the default re-hashes every byte of the resident mask on every call - an in-place edit must
    # not go unnoticed; a tuning loop that treats its mask as immutable opts out)
    type(ev).strict_mask_fingerprint = False
    try:
        ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
        sampled = []
        for _ in range(3):
            t0 = time.perf_counter()
            ev.get_metrics_ials(trainer, 0, U, mask, 20, 0, False)
            sampled.append(time.perf_counter() - t0)
    finally:
        type(ev).strict_mask_fingerprint = True
    wall_sampled = min(sampled)
    # The call no longer computes the dense contraction: the norm bound leaves a few percent of
    # the 64 x 64 score tiles (same lists, tests/test_gpu_fullsize.py).  `scores_per_s` stays
    # the dense-equivalent rate (what a caller gets); the roofline is priced on the flops the
    # kernels EXECUTED (sample pass + scored tiles).
    st = ev.last_call_stats()
    dense_flops = 2.0 * U * I * K
    flops = 2.0 * K * (64.0 * 64.0 * st["tiles_scored"] + float(U) * st["sample_items"])
    # HBM side of what runs: the sample pass and the candidate lists.  One-launch sample pass (K <= 128, 512
    # sample items): the user factors and the mask's item ids, read once; otherwise the [users, sample] score
    # block (written, masked, read)
    fused_sample = K <= 128 and st["sample_items"] == 512 and os.environ.get("IRSPACK_AMD_EVAL_SAMPLE_FUSED", "1") != "0"
    byts = (U * K * 4.0 + mask.nnz * 4.0 if fused_sample else 2.0 * U * st["sample_items"] * 4) + 2.0 * U * 20 * 8
    # Three clocks of the same call: the Python wall (fingerprint of the 80 MB mask + ctypes + the C
    # call), the C call alone (host clock inside the library), and the DEVICE span (HIP events on the
    # launch stream, first kernel -> last kernel done).  The roofline is priced on the device span: the
    # host's share says nothing about the kernels.
    span_s = max(st["device_span_ms"], 1e-6) * 1e-3
    return {
        "workload": (f"fused iALS k={K} scoring + nDCG@20 over {U} users x {I} items, fp32 scores; "
                     f"ground truth = 20 % per-row hold-out ({gt.nnz} entries), mask = the other 80 %"),
        "wall_s_first_call_incl_mask_upload": wall_first,
        "wall_s_incl_pcie": wall, "users_per_s": U / wall,
        "wall_s_sampled_mask_fingerprint": wall_sampled, "users_per_s_sampled_mask_fingerprint": U / wall_sampled,
        "library_call_ms": st["call_ms"], "device_span_ms": st["device_span_ms"],
        "host_ms_outside_the_library": wall_sampled * 1e3 - st["call_ms"],
        "mask_check": ("default: every byte of the resident mask hashed per call (irs_fingerprint, host threads); "
                       "EvaluatorCore.strict_mask_fingerprint = False: 1024-sample fingerprint"),
        "scores_per_s": U * float(I) / wall, "ndcg@20": m.as_dict()["ndcg"],
        "device_path": st,
        "model": "iALS fitted on the training entries (the mask) only: 3 CG epochs",
        "drop_in_block_path": block_path_leg(ev, trainer, mask),
        "tiles_scored_frac": st["tiles_scored"] / max(1, st["tiles_total"]),
        "roofline": both_terms(flops, byts, span_s, ceilings, bound=None,
                               scope="device span of the call (HIP events, first kernel -> last kernel done; "
                                     "includes the gap in which the host reads the flags / hard-row count back); "
                                     "after pruning the call is launch / latency bound",
                               executed_gflop=flops / 1e9, dense_gflop=dense_flops / 1e9,
                               dense_equivalent_tflops=dense_flops / span_s / 1e12,
                               dense_equivalent_tflops_python_wall=dense_flops / wall / 1e12,
                               sample_pass_and_lists_gbyte=byts / 1e9, sample_pass="one launch" if fused_sample else "score block in HBM"),
    }


def knn_leg(X, ceilings):
    """configs[2].  Headline = item pairs / s of the whole compute_similarity call (host CSR
    in, top-k, host CSR out: SURVEY 8(d)); the kernel-only figure beside it.  Roofline: the
    accumulation is one LDS atomic per multiply-add, so the ceiling is the LDS atomic rate
    measured in this run (irs_measure_ceilings), not HBM."""
    import scipy.sparse as sps

    from irspack_amd.recommenders._knn import CosineSimilarityComputer, JaccardSimilarityComputer

    U, I = X.shape
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    Xt.data[:] = 1.0
    t0 = time.perf_counter()
    comp = CosineSimilarityComputer(Xt, 0.0, True)
    create_s = time.perf_counter() - t0
    # the first call of a fresh computer (what a one-off `learn()` pays: it also sizes the scratch, the result
    # buffers and the page-locked staging buffer), then the steady state
    t0 = time.perf_counter()
    S = comp.compute_similarity(Xt, 100)
    first_call_s = time.perf_counter() - t0
    walls = []
    S = None
    for _ in range(3):
        S = None  # the previous result (32 MB) is released outside the timed call
        t0 = time.perf_counter()
        S = comp.compute_similarity(Xt, 100)
        walls.append(time.perf_counter() - t0)
    wall, ms, macs = min(walls), comp.last_kernel_ms, comp.last_macs
    atomic_peak = (ceilings or {}).get("lds_atomic_u32_gops")
    gmacs = macs / ms / 1e6
    knn_pmc = pmc_traffic(family="knn")
    traffic = knn_pmc.get("knn_tile_kernel")  # bytes per CALL (all launches of one compute_similarity)
    out = {
        "workload": f"cosine item-kNN top_k=100, {I} items x {U} users, binary interactions, fp64",
        "item_pairs_per_s": I * float(I) / wall,
        "wall_s_incl_pcie": wall, "kernel_ms": ms,
        "item_pairs_per_s_kernel_only": I * float(I) / (ms * 1e-3),
        "create_s": create_s, "first_call_wall_s": first_call_s, "fit_wall_s": create_s + first_call_s,
        "macs": macs, "gmacs_per_s_kernel": gmacs,
        "roofline": {"bound": "lds_atomic", "achieved": gmacs, "peak": atomic_peak,
                     "unit": "G lane-atomics/s (measured ds_add_u32, random banks)",
                     "frac": (gmacs / atomic_peak) if atomic_peak else None,
                     "frac_of_measured": (gmacs / atomic_peak) if atomic_peak else None,
                     "frac_of_spec": None, "traffic": traffic,
                     "hbm_side_gbs": macs * 4.0 / (ms * 1e-3) / 1e9,
                     "frac_hbm": macs * 4.0 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     "traffic_gbs": (traffic / (ms * 1e-3) / 1e9) if traffic else None,
                     "traffic_over_algorithmic": (traffic / (macs * 2.0)) if traffic else None,
                     "traffic_provenance": knn_pmc.get("_provenance") or knn_pmc.get("_stale"),
                     "note": "kernel_ms = device span from the first launch to the last row merge of the call's "
                             "three row chunks (accumulate + epilogue + select + merge; the host pass, the "
                             "index upload and the result copy of one chunk run beside the kernels of another); "
                             "fit_wall_s = the computer's construction (on the device since round 5) + the FIRST call "
                             "of that computer (buffers sized there), what one `learn()` of a kNN recommender costs; "
                             "HBM side prices the reference's 4 B column id per multiply-add; "
                             "traffic = rocprofv3 FETCH_SIZE + WRITE_SIZE of the tile kernel summed over the launches "
                             "of ONE call (profiles/pmc_traffic.json from the latest profiles/rNN_knn_pmc_hbm.json, "
                             "same matrix; null when knn.hip changed since that pass): the 2-byte column stream "
                             "(algorithmic: 2 B per multiply-add) is read once per column tile; "
                             "short slices waste part of their 128-byte lines"},
        "out_nnz": int(S.nnz),
    }
    variants = {}
    for name, other in (("cosine_normalize_false", CosineSimilarityComputer(Xt, 0.0, False)),
                        ("jaccard", JaccardSimilarityComputer(Xt, 0.0))):
        other.compute_similarity(Xt, 100, rows=(0, 64))
        ws = []
        for _ in range(3):  # like the headline variant: the minimum of three calls
            res = None
            t0 = time.perf_counter()
            res = other.compute_similarity(Xt, 100)
            ws.append(time.perf_counter() - t0)
        w = min(ws)
        del res
        variants[name] = {"item_pairs_per_s": I * float(I) / w, "wall_s_incl_pcie": w,
                          "wall_s_three_calls": ws,
                          "kernel_ms": other.last_kernel_ms,
                          "item_pairs_per_s_kernel_only": I * float(I) / (other.last_kernel_ms * 1e-3)}
        del other
    # weighted matrices (the kNN recommenders' `feature_weighting`): 64-bit fixed-point sums, no approximate
    # selection.  TF-IDF of binary interactions has ONE value per feature row of X_arg^T (the kernels then skip
    # the float64 value stream, round 5); BM25 does not (8 more bytes per multiply-add beside the 2-byte column)
    from irspack_amd.utils import okapi_BM_25_weight, tf_idf_weight

    for wname, weigh in (("cosine_tfidf_weighted", tf_idf_weight), ("cosine_bm25_weighted", okapi_BM_25_weight)):
        Xw = weigh(Xt)
        t0 = time.perf_counter()
        wcomp = CosineSimilarityComputer(Xw, 0.0, True)
        w_create = time.perf_counter() - t0
        ws = []
        for _ in range(3):
            res = None
            t0 = time.perf_counter()
            res = wcomp.compute_similarity(Xw, 100)
            ws.append(time.perf_counter() - t0)
        del res
        variants[wname] = {"item_pairs_per_s": I * float(I) / min(ws), "wall_s_incl_pcie": min(ws),
                           "wall_s_three_calls": ws, "create_s": w_create, "kernel_ms": wcomp.last_kernel_ms,
                           "item_pairs_per_s_kernel_only": I * float(I) / (wcomp.last_kernel_ms * 1e-3)}
        del wcomp, Xw
    out["variants"] = variants
    # `learn()` END TO END from the scipy CSR a user holds (knn.py:67-80: weighting -> computer on
    # X_weighted.T -> compute_similarity(X.T, top_k) -> remove_diagonal -> CSC): X.T goes to the library as the
    # CSC view it is (no host transpose), the weighting is applied on the device on the way in (the weighted
    # matrix never exists on the host), the target's columns are regrouped on host threads inside the call.
    # The recommender's constructor (base.py:94-101: a float64 CSR copy, the reference's own Python) is timed
    # beside it, not in it.  Minimum of three; the first (buffers, streams, page-locking) beside it.
    from irspack_amd.recommenders.knn import CosineKNNRecommender

    learn = {}
    for scheme in ("NONE", "TF_IDF", "BM_25"):
        t0 = time.perf_counter()
        rec = CosineKNNRecommender(X, shrinkage=0.0, normalize=True, top_k=100, feature_weighting=scheme)
        ctor = time.perf_counter() - t0
        ws = []
        for _ in range(3):
            rec._W = None
            t0 = time.perf_counter()
            rec.learn()
            ws.append(time.perf_counter() - t0)
        learn[scheme] = {"learn_wall_s": min(ws), "learn_wall_s_three_calls": ws, "recommender_ctor_s": ctor,
                         "item_pairs_per_s": I * float(I) / min(ws), "W_nnz": int(rec.W.nnz)}
        del rec
    out["learn"] = learn
    out["learn_wall_s"] = learn["NONE"]["learn_wall_s"]
    # ... and the model EVALUATED: Evaluator(test).get_score(model) for the item-kNN model fitted on the 80 %
    # training entries, nDCG@20 over all users on the 20 % hold-out.  The model scores as X_train[u] @ W: the
    # fused path forms that block on the device (scipy's per-entry order and rounding: the host product bit for
    # bit), masks and ranks it there; `block_loop` is the reference-shaped loop (the model's own
    # get_score_block through scipy, 128 users at a time, uploaded and ranked) on the first 4096 users.
    from irspack_amd.evaluation import Evaluator

    gt, train = holdout(X)
    rec = CosineKNNRecommender(train, shrinkage=0.0, normalize=True, top_k=100).learn()
    ev = Evaluator(gt, cutoff=20, target_metric="ndcg")
    ws = []
    for _ in range(3):
        t0 = time.perf_counter()
        ndcg = ev.get_score(rec)["ndcg"]
        ws.append(time.perf_counter() - t0)
    n_loop = min(4096, U)
    ev_loop = Evaluator(gt[:n_loop], cutoff=20, target_metric="ndcg", fused=False)
    t0 = time.perf_counter()
    ndcg_loop = ev_loop.get_score(rec)["ndcg"]
    loop_s = time.perf_counter() - t0
    out["evaluate"] = {"workload": f"nDCG@20 of the cosine item-kNN model (top_k = 100) over {U} users, 20 % hold-out",
                       "wall_s": min(ws), "wall_s_three_calls": ws, "users_per_s": U / min(ws), "ndcg@20": ndcg,
                       "block_loop": {"users": n_loop, "wall_s": loop_s, "users_per_s": n_loop / loop_s,
                                      "ndcg@20_first_users": ndcg_loop}}
    return out


def k256_leg(X, ceilings):
    """BASELINE configs[4] on one GPU: iALS K = 256 (Cholesky and CG) + fused nDCG@20."""
    from irspack_amd.recommenders._ials_core import IALSTrainer

    K = 256
    t0 = time.perf_counter()
    tr = IALSTrainer(model_config(K), X)
    out = {"workload": f"ml20m-shape {X.shape[0]}x{X.shape[1]} nnz={X.nnz}, iALS k={K} fp32",
           "create_s": time.perf_counter() - t0}
    out["cholesky"] = ials_leg(tr, X, K, "CHOLESKY", 3, 1, ceilings)
    out["cg"] = ials_leg(tr, X, K, "CG", 5, 1, ceilings)
    out["ialspp"] = ialspp_leg(tr, X, K, 3, 1)
    out["evaluator"] = evaluator_leg(X, tr, K, ceilings)
    return out


def c4_leg(full, ceilings):
    """BASELINE configs[3] shape on one GPU (K = 128): short rows, CG (reference default) and
    Cholesky.  Default = the full 10 M x 1 M matrix; --c4-small = the 1/5-scale one."""
    from irspack_amd.recommenders._ials_core import IALSTrainer
    from irspack_amd.synthetic import describe, make_interactions

    name, K = ("c4" if full else "c4_fifth"), 128
    t0 = time.perf_counter()
    X = make_interactions(name)
    gen_s = time.perf_counter() - t0
    info = describe(X)
    t0 = time.perf_counter()
    tr = IALSTrainer(model_config(K), X)
    out = {"workload": (f"{name} synthetic CSR {X.shape[0]}x{X.shape[1]} nnz={X.nnz} "
                        f"(geometric degrees, mean {info['mean_user_degree']:.1f}; Zipf items, "
                        f"max item degree {info['max_item_degree']}), iALS k={K} fp32"),
           "generate_s": gen_s, "create_s": time.perf_counter() - t0}
    out["cg"] = ials_leg(tr, X, K, "CG", 5, 1, ceilings)
    out["cholesky"] = ials_leg(tr, X, K, "CHOLESKY", 3, 1, ceilings)
    # fit through the boundary: the construction above + 16 CG epochs (the reference's default solver)
    # + both factor matrices to the host
    t0 = time.perf_counter()
    user, item = tr.user, tr.item
    d2h = time.perf_counter() - t0
    out["fit"] = {"solver": "CG max_cg_steps=3", "epochs": 16, "create_s": out["create_s"],
                  "epochs_s": 16 * out["cg"]["ms_per_epoch"] * 1e-3, "factors_to_host_s": d2h,
                  "fit_wall_s": out["create_s"] + 16 * out["cg"]["ms_per_epoch"] * 1e-3 + d2h,
                  "note": "create_s and factors_to_host_s measured, epochs_s = 16 x the measured epoch"}
    del user, item
    return out


def _max_over_ranks(seconds):
    import torch
    import torch.distributed as dist

    t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def knn_sharded_leg(X, device, barrier):
    """configs[2] over N ranks (EVERY rank calls this): the target rows of the item-item product are
    split into contiguous ranges of equal multiply-add count (sharding.similarity_row_bounds - the row
    split of knn.hpp:54-71 with ranks in the place of threads), every rank holds the whole computer and
    runs ``compute_similarity(rows=...)`` on its range; the CSR blocks are exchanged by two tensor
    all-gathers and laid end to end on every rank.  No collective inside the compute.  ``item_pairs_per_s`` = I^2 / (max over ranks
    of the wall of compute + exchange, between barriers); the compute-only maximum beside it."""
    import scipy.sparse as sps
    import torch.distributed as dist

    from irspack_amd.recommenders._knn import CosineSimilarityComputer
    from irspack_amd.sharding import sharded_similarity, similarity_row_bounds

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    U, I = X.shape
    Xt = sps.csr_matrix(X.T, dtype=np.float64)
    Xt.data[:] = 1.0
    bounds = similarity_row_bounds(Xt, world)
    comp = CosineSimilarityComputer(Xt, 0.0, True, device=device)
    mine = (bounds[rank], bounds[rank + 1])
    comp.compute_similarity(Xt, 100, rows=mine)  # the first call sizes scratch and staging buffers
    walls, computes, S = [], [], None
    for _ in range(3):
        S = None
        own = {}

        def rows(b, e):
            t0 = time.perf_counter()
            blk = comp.compute_similarity(Xt, 100, rows=(b, e))
            own["s"] = time.perf_counter() - t0
            return blk

        barrier()
        t0 = time.perf_counter()
        S = sharded_similarity(rows, I, bounds=bounds)
        barrier()
        walls.append(_max_over_ranks(time.perf_counter() - t0))
        computes.append(_max_over_ranks(own["s"]))
    wall, compute = min(walls), min(computes)
    return {
        "workload": f"cosine item-kNN top_k=100, {I} items x {U} users, binary interactions, fp64, "
                    f"target rows over {world} rank(s)",
        "item_pairs_per_s": I * float(I) / wall, "wall_s_max_over_ranks": wall,
        "item_pairs_per_s_compute_only": I * float(I) / compute, "compute_s_max_over_ranks": compute,
        "row_bounds": [int(b) for b in bounds], "balance": "equal multiply-add count per rank",
        "kernel_ms_rank0": comp.last_kernel_ms, "macs_rank0": comp.last_macs,
        "out_nnz": int(S.nnz), "scaling": "strong",
        "exchange": "two tensor all-gathers (lengths, then one padded byte buffer per rank: row lengths, column ids, "
                    "values); every rank ends with the whole result",
    }


def evaluator_sharded_leg(X, K, device, barrier):
    """The fused evaluator over N ranks (EVERY rank calls this): users are split into equal contiguous
    ranges (sharding.sharded_metrics), every rank scores and ranks its own users against the replicated
    item factors and the ``Metrics`` are summed in rank order on the host (Metrics::merge,
    evaluator.cpp:76-85).  The scored model is fitted by every rank on the training entries (3 CG epochs,
    identical replicas) - outside the timed region."""
    import torch.distributed as dist

    from irspack_amd.evaluation._core_evaluator import EvaluatorCore, Metrics
    from irspack_amd.recommenders._ials_core import IALSTrainer
    from irspack_amd.sharding import even_bounds, sharded_metrics

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    U, I = X.shape
    gt, mask = holdout(X)
    trainer = IALSTrainer(model_config(K), mask, device=device)
    for _ in range(3):
        trainer.step(solver_config("CG"))
    ev = EvaluatorCore(gt, [], device=device)
    b = even_bounds(U, world)
    my_mask = mask[b[rank]:b[rank + 1]]
    type(ev).strict_mask_fingerprint = False  # (the mask is immutable here; see evaluator_leg)
    try:
        ev.get_metrics_ials(trainer, b[rank], b[rank + 1], my_mask, 20, b[rank], False)  # uploads the mask rows
        walls, total = [], None
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            total = sharded_metrics(lambda lo, hi: ev.get_metrics_ials(trainer, lo, hi, my_mask, 20, lo, False),
                                    U, Metrics(I))
            barrier()
            walls.append(_max_over_ranks(time.perf_counter() - t0))
    finally:
        type(ev).strict_mask_fingerprint = True
    wall = min(walls)
    d = total.as_dict()
    return {"workload": f"fused iALS k={K} scoring + nDCG@20, {U} users over {world} rank(s) x {I} items",
            "users_per_s": U / wall, "wall_s_max_over_ranks": wall, "ndcg@20": d["ndcg"],
            "valid_user": int(d["valid_user"]), "total_user": int(d["total_user"]), "scaling": "strong",
            "exchange": "all_gather_object of one Metrics per rank, merged in rank order"}


def _try_attach_peers(local) -> bool:
    """Maps the peers' factor buffers / mailboxes on an RCCL communicator that was created without
    them (collective: every rank calls it); False when any rank cannot."""
    import ctypes as C

    import torch.distributed as dist

    from irspack_amd._lib import COMM_HANDLE_BYTES, check, lib

    world = dist.get_world_size()
    blob = None
    try:
        mine = (C.c_char * COMM_HANDLE_BYTES)()
        check(lib().irs_comm_export(local._comm, local.trainer._h, mine))
        blob = bytes(mine)
    except (RuntimeError, ValueError):
        pass
    blobs = [None] * world
    dist.all_gather_object(blobs, blob)
    ok = all(b is not None for b in blobs)
    if ok:
        try:
            allb = (C.c_char * (COMM_HANDLE_BYTES * world)).from_buffer_copy(b"".join(blobs))
            check(lib().irs_comm_attach(local._comm, local.trainer._h, allb))
        except (RuntimeError, ValueError) as exc:
            local.peers_error = repr(exc)
            ok = False
    votes = [None] * world
    dist.all_gather_object(votes, ok)
    local.peers_attached = all(votes)
    return local.peers_attached


def _factor_digest(tr) -> float:
    """A number that changes with any bit of the two factor matrices (float64 sums of the float32
    values weighted by position) - compared for EQUALITY between runs of the same arithmetic."""
    import numpy as np

    out = 0.0
    for F in (tr.user, tr.item):
        w = (np.arange(F.size, dtype=np.float64) % 8191.0 + 1.0).reshape(F.shape)
        out += float((F.astype(np.float64) * w).sum())
    return out


def _exchange_ab(trainer, local, sc, barrier) -> dict:
    """The row exchanges of the native RCCL path, A/B (every rank calls it; N > 1): a few untimed epochs
    of `auto`, `mesh` and - if the peers' memory can be mapped - `peer` from the same factors (max over
    ranks of the host clock around barrier + synchronize) and a digest of the factors each ends on.  The
    exchange moves rows, it does not compute: a mode is eligible only if every replica ends on the
    digest of `auto`; the fastest eligible mode is selected for the timed region
    (IRSPACK_AMD_BENCH_EXCHANGE pins one)."""
    import torch
    import torch.distributed as dist

    pinned = os.environ.get("IRSPACK_AMD_BENCH_EXCHANGE")
    modes = ["auto", "mesh"] + (["peer"] if local.peers_attached or _try_attach_peers(local) else [])
    ab_epochs = int(os.environ.get("IRSPACK_AMD_BENCH_AB_EPOCHS", "4"))
    user0, item0 = local.trainer.user, local.trainer.item
    out = {"epochs_each": ab_epochs, "modes": {}, "peers_mapped": bool(local.peers_attached),
           "peers_error": getattr(local, "peers_error", None)}
    for mode in modes:
        rec = {}
        try:
            trainer.set_exchange(mode)
            local.trainer.user, local.trainer.item = user0, item0
            trainer.step(sc)  # (first use of a mode: connection set-up inside RCCL)
            local.trainer.user, local.trainer.item = user0, item0
            barrier()
            t0 = time.perf_counter()
            for _ in range(ab_epochs):
                trainer.step(sc)
            trainer.synchronize()
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            rec["ms_per_epoch"] = float(dt.item()) / ab_epochs * 1e3
            digest = _factor_digest(local.trainer)
            lohi = torch.tensor([digest, -digest], dtype=torch.float64, device="cuda")
            dist.all_reduce(lohi, op=dist.ReduceOp.MIN)  # min(d), -max(d): equal on every rank?
            rec["digest"] = digest
            rec["replicas_identical"] = bool(lohi[0].item() == -lohi[1].item())
        except (RuntimeError, ValueError) as exc:  # (solver / argument errors: raised on every rank together)
            rec["error"] = repr(exc)
        out["modes"][mode] = rec
    base = out["modes"]["auto"]
    ok = [m for m, r in out["modes"].items()
          if "error" not in r and r["replicas_identical"] and r["digest"] == base.get("digest")]
    best = min(ok, key=lambda m: out["modes"][m]["ms_per_epoch"]) if ok else "auto"
    choice = pinned if pinned in out["modes"] and pinned in ok else best
    out["timed"] = choice
    out["agree_with_auto"] = ok
    trainer.set_exchange(choice)
    local.trainer.user, local.trainer.item = user0, item0
    return out


def launch_ranks(args, argv):
    """``python bench.py --gpus N`` (N > 1) WITHOUT an outer torch.distributed.run: start the N ranks as a
    CHILD process - the launcher the driver itself uses - before this process has touched the GPU (no HIP
    call, no ``torch.cuda.is_available()``; ``device_count()`` reads sysfs only), relay rank 0's JSON line
    and return the child's exit code.  A box with fewer than N devices is refused (exit 2) rather than
    benchmarked under the wrong ``n_gpus``; IRSPACK_AMD_BENCH_ONE_DEVICE=1 (tests: every rank on cuda:0
    over gloo) lifts that check."""
    import socket
    import subprocess

    if not os.environ.get("IRSPACK_AMD_BENCH_ONE_DEVICE"):
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} asked for but this node shows {have} HIP device(s); "
                  "refusing to print a line labelled with a GPU count that did not run", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # (stderr passes through)
    lines = []
    for ln in proc.stdout.splitlines():
        if ln.startswith("{"):
            lines.append(ln)
        else:  # (launcher / library chatter on the children's stdout)
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    for ln in lines:
        print(ln, flush=True)
    return proc.returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); "
                         "the line would carry a GPU count that did not run")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (one-GPU boxes): IRSPACK_AMD_BENCH_BACKEND=gloo and
    # IRSPACK_AMD_BENCH_ONE_DEVICE=1 run the N > 1 control flow with every rank on cuda:0
    backend = os.environ.get("IRSPACK_AMD_BENCH_BACKEND", "nccl")
    if os.environ.get("IRSPACK_AMD_BENCH_ONE_DEVICE"):
        local_rank = 0
    # IRSPACK_AMD_BENCH_FORCE_DIST=1 (tests): the N > 1 code path - process group, native transport,
    # preflight, exchange A/B, phase split - with ONE rank (RCCL accepts a world of one on one GPU)
    multi = world > 1 or bool(os.environ.get("IRSPACK_AMD_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback).")
    torch.cuda.set_device(local_rank)

    from irspack_amd import _lib
    from irspack_amd.sharding import (HipLocalSolver, ShardedIALSTrainer, equal_shard_bounds,
                                      shard_bounds)
    from irspack_amd.synthetic import describe, make_interactions

    X = make_interactions(args.shape)  # identical on every rank (seeded)
    info = describe(X)
    U, I = X.shape
    K = args.K
    mc = model_config(K)
    sc = solver_config(args.solver)
    # Row shards.  "equal": equal row blocks - on a matrix whose row order is random (ML-20M
    # shape) they are cost-balanced to a few per cent and the solved rows travel in ONE in-place
    # all-gather per half-epoch.  "cost": contiguous ranges of equal solve cost (one collective
    # through padded staging rows) - needed when single rows carry whole per cents of the work
    # (configs[3]: one item row holds 4.6 % of all entries = +37 % on its owner at N = 8).
    longest = max(info["max_user_degree"], info["max_item_degree"])
    balance = args.balance
    if balance == "auto":
        balance = "cost" if longest > 0.01 * max(info["nnz"], 1) else "equal"
    ub, ib = shard_bounds(X, K, args.solver, world) if balance == "cost" else equal_shard_bounds(X, world)
    shard = (ub[rank], ub[rank + 1], ib[rank], ib[rank + 1])
    local = HipLocalSolver(mc, X, shard, local_rank)
    # (overlap of the next Gramian's all-reduce with the all-gather: IRSPACK_AMD_BENCH_OVERLAP=1;
    # off by default until it has run on RCCL at world >= 2, see ShardedIALSTrainer)
    # N > 1: the epoch behind ONE C-ABI call (irs_ials_sharded_step: the transport driven from inside
    # the library, rows in place, the next Gramian overlapping the row exchange);
    # IRSPACK_AMD_BENCH_COMM=torch times the torch.distributed host loop instead, =local the RCCL-free
    # peer-store transport.  The native path is set up and tried COLLECTIVELY before anything is timed
    # (ShardedIALSTrainer: every rank votes; one failure = every rank on the host loop; a hang = exit
    # code 3 from the watchdog) and the line says which path was timed.  The per-phase split of the
    # line always comes from the host loop (a few extra, untimed epochs when the native path is timed).
    comm_choice = os.environ.get("IRSPACK_AMD_BENCH_COMM", "native")
    want_native = multi and comm_choice != "torch" and (backend == "nccl" or comm_choice == "local")
    overlap = bool(int(os.environ.get("IRSPACK_AMD_BENCH_OVERLAP", "0")))
    watchdog_s = float(os.environ.get("IRSPACK_AMD_BENCH_WATCHDOG_S", "600"))
    trainer = ShardedIALSTrainer(local, ub, ib, timing=False, overlap=overlap,
                                 native=("local" if comm_choice == "local" else "rccl") if want_native else False,
                                 exchange="peer" if comm_choice == "local" else "auto", watchdog_s=watchdog_s)
    preflight = None
    if want_native:
        preflight = {"requested": "local (peer stores)" if comm_choice == "local" else "rccl",
                     "created": bool(trainer.native), "first_epoch": False}
        if trainer.native:
            preflight["first_epoch"] = bool(trainer.preflight_step(sc))
        if trainer.native_error:
            preflight["error"] = trainer.native_error
        preflight["peers_mapped"] = bool(getattr(local, "peers_attached", False))
    native = bool(trainer.native)
    if multi and not native:  # the host loop is the timed path: it carries the phase marks itself
        trainer.timing = True

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The row exchanges of the native path, A/B: a few untimed epochs of each from the same factors
    # (max over ranks of the host clock around barrier + synchronize), a checksum of the factors they
    # end on (the exchange moves rows, it does not compute: every mode must end on the same bits), and
    # the fastest mode whose bits agree becomes the timed one (IRSPACK_AMD_BENCH_EXCHANGE pins it).
    exchange_ab = None
    if native and trainer.native_transport == "rccl":
        try:
            exchange_ab = _exchange_ab(trainer, local, sc, barrier)
        except Exception as exc:  # (a defect of the A/B itself - the same on every rank - must not cost the headline)
            exchange_ab = {"error": repr(exc), "timed": trainer.native_exchange}
    for _ in range(args.warmup):
        trainer.step(sc)
    trainer.synchronize()
    trainer.last_timing()  # drop the warm-up marks
    # The timed steps carry HIP events on the dominant kernel only (the solve of the side with more
    # rows) - the roofline below is priced on its duration, measured live in this region; event pairs
    # on all ten launches of an epoch cost 0.05 ms of 2.1.  The other kernels' durations come from two
    # extra, untimed epochs afterwards.
    local.trainer.profile(2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(sc)
    trainer.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = local.trainer.profile_read()
    local.trainer.profile(True)
    for _ in range(2):
        trainer.step(sc)
    trainer.synchronize()
    for name, rec in local.trainer.profile_read().items():
        prof.setdefault(name, rec)  # (the dominant kernel keeps its timed-region figure)
    local.trainer.profile(False)
    comm = None
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # per-phase device time of the timed epochs (events on the stream the kernels and the
        # collectives' waits run on), per step; max and min over ranks
        phase_steps = args.steps
        if native:  # the host loop, untimed, for the split only
            host_loop = ShardedIALSTrainer(local, ub, ib, timing=True, overlap=overlap)
            host_loop.step(sc)
            host_loop.synchronize()
            host_loop.last_timing()
            phase_steps = min(args.steps, 5)
            for _ in range(phase_steps):
                host_loop.step(sc)
            host_loop.synchronize()
            tm = host_loop.last_timing()
            exchange_plan = list(host_loop.exchange)
        else:
            tm = trainer.last_timing()
            exchange_plan = list(trainer.exchange)
        keys = ["gramian_ms", "allreduce_ms", "solve_ms", "allgather_ms", "exposed_comm_ms", "total_ms"]
        mine = torch.tensor([tm[k] / max(phase_steps, 1) for k in keys], dtype=torch.float64, device="cuda")
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        comm = {"per_step_ms_max_over_ranks": {k: float(v) for k, v in zip(keys, hi.tolist())},
                "per_step_ms_min_over_ranks": {k: float(v) for k, v in zip(keys, lo.tolist())},
                "compute_ms": float(hi[0] + hi[2]), "allreduce_ms": float(hi[1]),
                "allgather_ms": float(hi[3]), "exposed_comm_ms": float(hi[4]),
                "exchange": exchange_plan, "overlap": bool(trainer.overlap),
                "timed_path": ((f"native: irs_ials_sharded_step (transport {trainer.native_transport} inside the "
                                f"library, row exchange '{trainer.native_exchange}', in-place rows, next Gramian "
                                "overlapped)") if native else "torch.distributed host loop"),
                "native_preflight": preflight, "exchange_ab": exchange_ab,
                "phase_split_from": "torch.distributed host loop" + (" (extra untimed epochs)" if native else ""),
                "note": "solve_ms spread (max - min over ranks) = load imbalance; the collectives' "
                        "time includes waiting for the slowest rank"}

    # N > 1: the kNN and evaluator legs sharded over the same ranks (north_star: item pairs/s at 1/2/4/8).
    # Every rank takes part; a failing leg (the same exception on every rank: arguments, memory) is
    # recorded instead of costing the headline line.
    sharded_sec = {}
    if multi and not args.no_secondary and K <= 64:
        legs_n = [s for s in args.legs.split(",") if s]
        for name, fn in (("knn", lambda: knn_sharded_leg(X, local_rank, barrier)),
                         ("evaluator", lambda: evaluator_sharded_leg(X, K, local_rank, barrier))):
            if name not in legs_n:
                continue
            t0 = time.perf_counter()
            try:
                sharded_sec[name] = fn()
            except (RuntimeError, ValueError) as exc:
                sharded_sec[name] = {"error": repr(exc)}
            sharded_sec[name]["leg_wall_s"] = time.perf_counter() - t0

    result = None
    if rank == 0:
        ceilings = None
        try:
            ceilings = _lib.measure_ceilings(local_rank)
            ceilings.update({
                "spec_hbm_gbs": PEAK_HBM_GBS, "spec_f32_tflops": PEAK_F32_TFLOPS,
                "copy_frac_of_spec": ceilings["copy_gbs"] / PEAK_HBM_GBS,
                "mfma_f32_frac_of_spec": ceilings["mfma_f32_tflops"] / PEAK_F32_TFLOPS,
                "copy_note": ("copy_gbs is what THIS library's plain copy kernel reaches (5.1-5.2 TB/s in rounds "
                              "1-4); /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a tuned "
                              "copy, so every HBM-side frac_of_measured is ~20 % kinder than one priced "
                              "on the guide's figure"),
                "guide_copy_gbs": 6290.0,
                "how": "irs_measure_ceilings: 1 GiB copy / triad, v_mfma_f32_16x16x4_f32 loop on "
                       "every SIMD, random-bank ds_add_u32 loop on every CU, random 256 / 512-byte row gather "
                       "out of a 1 GiB table; best of 3-5, HIP events"})
        except Exception as exc:  # the headline line must still be printed
            ceilings = {"error": repr(exc)}
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
            roofline = both_terms(flops, byts, t_launch, ceilings,
                                  bound="mfma" if args.solver == "CHOLESKY" else "hbm", kernel=name,
                                  avg_launch_ms=st["ms"] / st["launches"], launches=st["launches"],
                                  algorithmic_gflop_per_launch=flops / 1e9,
                                  algorithmic_gbyte_per_launch=byts / 1e9)
            # which matrix instruction the rank update runs on: binary interactions, Cholesky,
            # 48 < K <= 64 -> the bf16 matrix cores on exact three-way splits (fp32-exact partial products,
            # fp32 accumulate) unless IRSPACK_AMD_IALS_BF16X3=0; `frac` stays priced on the ALGORITHMIC fp32
            # flops against the fp32 peak, the executed bf16 flops are shown against the bf16 peak beside it
            bf16x3 = (args.solver == "CHOLESKY" and 48 < K <= 64 and bool(np.all(X.data == 1.0))
                      and os.environ.get("IRSPACK_AMD_IALS_BF16X3", "1") != "0"
                      and os.environ.get("IRSPACK_AMD_IALS_UNIT", "1") != "0")
            if bf16x3:
                groups = float(np.ceil(np.diff(Xs.indptr[b:e + 1]) / 32.0).sum())  # 32-entry groups walked
                executed = groups * 60 * 16384.0  # 10 tiles x 6 partial products x v_mfma_f32_16x16x32_bf16
                roofline["rank_update"] = {
                    "form": "bf16x3: v_mfma_f32_16x16x32_bf16 on exact 3-way bf16 splits of the fp32 factors, "
                            "6 fp32-exact partial products per product, fp32 accumulate (IRSPACK_AMD_IALS_BF16X3=0: "
                            "v_mfma_f32_16x16x4_f32, leg secondary.ials_f32_mfma)",
                    "executed_bf16_tflops": executed / t_launch / 1e12, "peak_bf16_tflops": 2500.0,
                    "frac_bf16_peak": executed / t_launch / 1e12 / 2500.0,
                    "accuracy": "every row of both benchmark half-steps vs the float64 solve: worst 1.1e-5 (user) / "
                                "1.9e-6 (item), the float32 oracle 3.8e-5 / 5.7e-5 (tests/test_gpu_fullsize.py, "
                                "profiles/parity_r06.json)"}
            else:
                roofline["rank_update"] = {"form": "fp32: v_mfma_f32_16x16x4_f32, one triangle of 16 x 16 tiles"}
            pmc = pmc_traffic(f"{args.shape} K={K}") if world == 1 else {}
            roofline["traffic"] = pmc.get(name)
            roofline["traffic_provenance"] = pmc.get("_provenance") or pmc.get("_stale") or (
                "no PMC pass of this workload is committed" if world == 1 else "N > 1: per-rank shards, no pass")
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
                "balance": balance,
            },
            "comm": comm,
            "kernels_ms_per_launch": {k: round(v["ms"] / v["launches"], 4) for k, v in prof.items()},
            "roofline": roofline,
            "ceilings": ceilings,
        }
        if world == 1 and not args.no_secondary:
            legs = [s for s in args.legs.split(",") if s]
            sec = {}

            def run(name, fn):
                if name not in legs:
                    return
                t0 = time.perf_counter()
                try:
                    sec[name] = fn()
                except Exception as exc:  # a failing leg must not lose the headline line
                    sec[name] = {"error": repr(exc)}
                sec[name]["leg_wall_s"] = time.perf_counter() - t0

            # the solver the headline did not run (the reference's default is CG, 3 steps)
            other = "CG" if args.solver == "CHOLESKY" else "CHOLESKY"
            other_leg = "ials_" + other.lower()
            if "ials_cg" in legs and other_leg not in legs:
                legs.append(other_leg)
            run(other_leg, lambda: ials_leg(local.trainer, X, K, other, args.steps, 2, ceilings,
                                            workload=f"{args.shape} K={K}"))
            if K <= 64 and args.solver == "CHOLESKY":
                run("ials_f32_mfma", lambda: f32_mfma_leg(X, K, args.steps, ceilings))
            run("fit", lambda: fit_leg(X, K, args.solver))
            if K <= 64:
                run("knn", lambda: knn_leg(X, ceilings))
                run("evaluator", lambda: evaluator_leg(X, local.trainer, K, ceilings))
            trainer = local = None  # free the K = 64 trainer before the larger legs
            if args.shape == "ml20m":
                run("k256", lambda: k256_leg(X, ceilings))
                run("c4", lambda: c4_leg(not args.c4_small, ceilings))
            result["secondary"] = sec
        if sharded_sec:
            result["secondary"] = sharded_sec
        # last: 256 busy host threads just before a GPU leg disturb its (host-clocked) timing
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(X, K, args.solver, args.cpu_seconds)
    if multi:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
