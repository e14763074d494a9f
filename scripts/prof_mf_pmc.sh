# SQ counters of the matrix-free CG kernels at K = 256 (two PMC passes; no tracing beside them)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
export IRSPACK_AMD_MF_FORK=0
rm -rf gpurun_out/prof/mf_sq gpurun_out/prof/mf_sq2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/mf_sq -- python3 scripts/quick_ials.py --K 256 --solvers CG --epochs 1 > gpurun_out/prof/mf_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/prof/mf_sq2 -- python3 scripts/quick_ials.py --K 256 --solvers CG --epochs 1 > gpurun_out/prof/mf_sq2.log 2>&1
python3 - <<'P'
import csv, glob, collections
for d in ("mf_sq", "mf_sq2"):
    fs = sorted(glob.glob(f"gpurun_out/prof/{d}/*/*counter_collection.csv"))
    if not fs:
        print(d, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[-1])):
        name = r["Kernel_Name"].split("(")[0][-40:]
        if "mf_" not in name: continue
        k = (name, int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        print(k, {a: f"{b:.3g}" for a, b in v.items()})
P
