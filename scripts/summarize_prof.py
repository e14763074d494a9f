"""Turn the rocprofv3 output of scripts/prof_bench.sh (gpurun_out/prof) into the tracked
summaries under profiles/: the --kernel-trace --stats table, and the per-kernel HBM
traffic from the separate FETCH_SIZE / WRITE_SIZE PMC passes.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts 64 B per 128 B request of a wide coalesced stream, so it is doubled;
WRITE_SIZE is exact for 16 B-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def per_kernel(path):
    """kernel name -> grid size -> [n, sum]"""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"]][int(r["Grid_Size"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def newest(pattern):
    """gpurun merges every run into gpurun_out/: keep the most recent match only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:] if files else []


def main():
    os.makedirs(DST, exist_ok=True)
    stats = newest(os.path.join(SRC, "kt", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))
    # the JSON line bench.py printed in that same (traced) process: its HIP-event durations are
    # the ones that have to agree with the trace
    ktlog = os.path.join(SRC, "kt.log")
    if os.path.exists(ktlog):
        for line in open(ktlog):
            if line.startswith('{"metric"'):
                json.dump(json.loads(line),
                          open(os.path.join(DST, f"{TAG}_bench_line_under_rocprofv3_kernel_trace.json"), "w"),
                          indent=1)
    # the user and the item half-step launch the SAME kernel symbol with different grids:
    # split the trace by grid so that each average can be set against bench.py's per-side
    # HIP-event numbers (the larger MODE-0 grid is the user side)
    trace = newest(os.path.join(SRC, "kt", "*", "*_kernel_trace.csv"))
    if trace:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            if "irs::" in r["Kernel_Name"]:
                by[(r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]),
                    int(r["VGPR_Count"]), int(r["LDS_Block_Size"]))].append(
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(DST, f"{TAG}_bench_kernel_by_grid.csv"), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["Name", "Grid_Size", "Workgroup_Size", "VGPRs", "LDS_Bytes", "Calls",
                         "AverageNs", "MinNs", "MaxNs"])
            for (name, grid, wg, vgpr, lds), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                wr.writerow([name, grid, wg, vgpr, lds, len(d), round(sum(d) / len(d), 1),
                             min(d), max(d)])
    # secondary paths (scripts/prof_secondary.sh): kNN + evaluator trace, iALS++ trace, SQ counters
    for src, dst in (("sec_kt", "knn_eval"), ("pp_kt", "ialspp"), ("ef_kt_64", "eval_fused_k64"),
                     ("ef_kt_256", "eval_fused_k256"), ("c4_kt", "c4"), ("pp64_kt", "ialspp_k64")):
        st = newest(os.path.join(SRC, src, "*", "*_kernel_stats.csv"))
        if st:
            shutil.copy(st[0], os.path.join(DST, f"{TAG}_{dst}_kernel_stats.csv"))
    pmc = newest(os.path.join(SRC, "sec_pmc", "*", "*_counter_collection.csv"))
    if pmc:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for r in csv.DictReader(open(pmc[0])):
            if "irs::" not in r["Kernel_Name"]:
                continue
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]].add(r["Dispatch_Id"])
        sq = {}
        for name, c in agg.items():
            n = max(len(launches[name]), 1)
            d = {k: v / n for k, v in sorted(c.items())}
            d["launches"] = n
            if d.get("GRBM_GUI_ACTIVE"):
                # SQ counters are summed over 32 shader engines (8 XCDs x 4)
                d["valu_busy_frac"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / (32 * d["GRBM_GUI_ACTIVE"])
                d["lds_inst_busy_frac"] = d.get("SQ_ACTIVE_INST_LDS", 0.0) / (32 * d["GRBM_GUI_ACTIVE"])
            sq[name] = d
        json.dump(sq, open(os.path.join(DST, f"{TAG}_knn_eval_pmc_sq.json"), "w"), indent=1)
    # SQ counters of the bench kernels (scripts/prof_pmc_sq.sh: two passes per solver)
    sq_all = {}
    for solver in ("CHOLESKY", "CG"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        for d in (f"sq_{solver}", f"sq2_{solver}"):
            for f in newest(os.path.join(SRC, d, "*", "*_counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    name = r["Kernel_Name"]
                    if "irs::ials::ials_" not in name:
                        continue
                    short = name.split("irs::ials::")[1].split("(")[0]
                    key = f"{solver}:{short}:grid{r['Grid_Size']}"
                    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                    launches[(key, d)].add(r["Dispatch_Id"])
        for key, c in agg.items():
            n = max(max(len(v) for (k, _), v in launches.items() if k == key), 1)
            d = {k: v / n for k, v in sorted(c.items())}
            if d.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE sums the 8 XCDs; 1024 SIMDs; SQ_ACTIVE_INST_* count quad-cycles
                simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024
                d["mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
                d["valu_active_frac"] = 4 * d.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
            sq_all[key] = d
    if sq_all:
        json.dump(sq_all, open(os.path.join(DST, f"{TAG}_bench_pmc_sq.json"), "w"), indent=1)
    out = {"units": "bytes per launch", "fetch_correction": "FETCH_SIZE KiB x 1024 x 2",
           "write_correction": "WRITE_SIZE KiB x 1024", "kernels": {}}
    fetch = newest(os.path.join(SRC, "fetch", "*", "*_counter_collection.csv"))
    write = newest(os.path.join(SRC, "write", "*", "*_counter_collection.csv"))
    if not (fetch and write):
        print("no PMC passes found")
        return
    f, w = per_kernel(fetch[0]), per_kernel(write[0])
    traffic = {}
    for name in f:
        if "irs::" not in name:
            continue
        for grid, (n, s) in sorted(f[name].items()):
            fb = s / n * 1024 * 2
            wn, ws = w.get(name, {}).get(grid, [1, 0.0])
            wb = ws / max(wn, 1) * 1024
            out["kernels"].setdefault(name, []).append(
                {"grid_size": grid, "launches": n, "fetch_bytes": fb, "write_bytes": wb,
                 "hbm_bytes": fb + wb})
    # map the two solve-kernel grids onto the library's per-side names (users = larger grid)
    for name, rows in out["kernels"].items():
        if "ials_solve_kernel<" in name and name.rstrip().endswith("0>(irs::ials::SolveParams)") is False:
            pass
    for name, rows in out["kernels"].items():
        if "ials_solve_kernel" not in name:
            continue
        targs = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
        solver = "cg" if targs[1] == "1" else "cholesky"   # <T, SOLVER, MODE, UNIT>
        kind = "split" if targs[2] == "1" else "solve"
        rows = sorted(rows, key=lambda r: r["grid_size"])
        if len(rows) >= 2:
            # users have more rows than items => larger grid for `solve`; for `split` the item side
            # has more split rows
            small, large = rows[0], rows[-1]
            user, item = (large, small) if kind == "solve" else (small, large)
            traffic[f"ials_{kind}_{solver}_user"] = user["hbm_bytes"]
            traffic[f"ials_{kind}_{solver}_item"] = item["hbm_bytes"]
    json.dump(out, open(os.path.join(DST, f"{TAG}_bench_pmc_hbm.json"), "w"), indent=1)
    # kNN tile kernel: FETCH_SIZE / WRITE_SIZE passes over scripts/quick_knn_eval.py --skip-eval
    # (full-matrix calls only: the warm-up call of 64 rows has a smaller grid of slots)
    kf = newest(os.path.join(SRC, "sec_fetch", "*", "*_counter_collection.csv"))
    kw = newest(os.path.join(SRC, "sec_write", "*", "*_counter_collection.csv"))
    if kf and kw:
        # PER CALL: a compute_similarity call launches the tile kernel once per row chunk (three on the ML-20M
        # shape); the profiled script makes KNN_CALLS full-matrix calls and nothing else that launches it
        KNN_CALLS = 2  # scripts/quick_knn_eval.py: `for rep in range(2)`

        def knn_sum(path, counter):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(path)):
                if "knn_tile_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            return sum(per.values()) / KNN_CALLS, len(per) / KNN_CALLS
        (fk, launches), (wk, _) = knn_sum(kf[0], "FETCH_SIZE"), knn_sum(kw[0], "WRITE_SIZE")
        fb, wb = fk * 1024 * 2, wk * 1024
        if fb > 0:
            traffic["knn_tile_kernel"] = fb + wb
            json.dump({"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                       "tile_kernel_launches_per_call": launches,
                       "units": "FETCH_SIZE x 2 KiB (gfx950 correction), WRITE_SIZE x 1 KiB; summed over the "
                                "launches of ONE compute_similarity call (all launches of the pass / calls made)"},
                      open(os.path.join(DST, f"{TAG}_knn_pmc_hbm.json"), "w"), indent=1)
    # provenance: bench.py prints these figures only while the kernels' sources are the ones the passes ran
    sys.path.insert(0, ROOT)
    from bench import kernel_source_sha16  # noqa: E402

    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    traffic["_provenance"] = {
        "collected_at_git_head": head, "tag": TAG, "sources_sha16": kernel_source_sha16(),
        "units": "HBM bytes (FETCH_SIZE x 2 KiB + WRITE_SIZE x 1 KiB): per LAUNCH for ials_*, per "
                 "compute_similarity CALL for knn_tile_kernel"}
    json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
