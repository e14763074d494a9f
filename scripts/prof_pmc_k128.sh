# SQ counters of the K = 128 Cholesky / CG epochs on the ML-20M shape (development)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/k128_sq gpurun_out/prof/k128_sq2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/k128_sq -- python3 scripts/quick_ials.py --K 128 --epochs 2 --solvers CHOLESKY,CG > gpurun_out/prof/k128_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof/k128_sq2 -- python3 scripts/quick_ials.py --K 128 --epochs 2 --solvers CHOLESKY,CG > gpurun_out/prof/k128_sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in ("k128_sq", "k128_sq2"):
    for f in glob.glob(f"gpurun_out/prof/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ials_solve_kernel" not in r["Kernel_Name"]: continue
            key = (r["Kernel_Name"].split("(")[0][-44:], int(r["Grid_Size"]))
            a = agg[key][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for key, c in sorted(agg.items(), key=lambda kv: -kv[0][1]):
    v = {k: x[1] / x[0] for k, x in c.items()}
    simd = v["GRBM_GUI_ACTIVE"] * 128
    print(key, "ms", round(v["GRBM_GUI_ACTIVE"] / 8 / 2.4e6, 3), "mfma_busy", round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 3),
          "valu", round((v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) * 4 / simd, 3), "mfma", int(v["SQ_INSTS_MFMA"]), "valu_n", int(v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]), "waves", int(v["SQ_WAVES"]))
PY
