"""Summaries of scripts/prof_largek.sh (gpurun_out/prof/lk_*) -> profiles/<tag>_largek_*.{csv,json}."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def main():
    rows = []
    pmc = {}
    for K in (128, 256):
        for S in ("CHOLESKY", "CG"):
            tr = newest(os.path.join(SRC, f"lk_kt_{K}_{S}", "*", "*_kernel_trace.csv"))
            if tr:
                by = collections.defaultdict(list)
                for r in csv.DictReader(open(tr)):
                    if "irs::" in r["Kernel_Name"]:
                        by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]), int(r["VGPR_Count"]),
                            int(r["LDS_Block_Size"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                for (name, grid, vgpr, lds), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                    rows.append([K, S, name, grid, vgpr, lds, len(d), round(sum(d) / len(d) / 1e3, 1)])
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            launches = collections.defaultdict(set)
            for d in (f"lk_sq_{K}_{S}", f"lk_sq2_{K}_{S}"):
                f = newest(os.path.join(SRC, d, "*", "*_counter_collection.csv"))
                if not f:
                    continue
                for r in csv.DictReader(open(f)):
                    if "ials_" not in r["Kernel_Name"]:
                        continue
                    key = r["Kernel_Name"].split("irs::ials::")[1].split("(")[0] + f":grid{r['Grid_Size']}"
                    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
                    launches[(key, d)].add(r["Dispatch_Id"])
            for key, c in agg.items():
                n = max(max(len(v) for (k, _), v in launches.items() if k == key), 1)
                d = {k: v / n for k, v in sorted(c.items())}
                if d.get("GRBM_GUI_ACTIVE"):
                    simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 1024  # GRBM sums the 8 XCDs; 1024 SIMDs
                    d["mfma_busy_frac"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles
                    d["valu_active_frac"] = 4 * d.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles
                    if d.get("SQ_WAVE_CYCLES"):
                        d["wait_any_frac_of_wave_cycles"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
                        d["wait_inst_frac_of_wave_cycles"] = d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
                pmc[f"K{K}:{S}:{key}"] = d
    os.makedirs(DST, exist_ok=True)
    with open(os.path.join(DST, f"{TAG}_largek_kernel_by_grid.csv"), "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["K", "solver", "Name", "Grid_Size", "VGPRs", "LDS_Bytes", "Calls", "AverageUs"])
        wr.writerows(rows)
    json.dump(pmc, open(os.path.join(DST, f"{TAG}_largek_pmc_sq.json"), "w"), indent=1)
    for r in rows:
        print(r)
    for k, d in pmc.items():
        print(k, {a: round(d[a], 3) for a in ("mfma_busy_frac", "valu_active_frac", "wait_any_frac_of_wave_cycles",
                                             "wait_inst_frac_of_wave_cycles") if a in d},
              {a: int(d[a]) for a in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU") if a in d})


if __name__ == "__main__":
    main()
