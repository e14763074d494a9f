# PMC passes over the C4-shape CG epoch (bench.py --legs c4 --c4-small uses the 1/5 shape unless FULL=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
ARGS="--legs c4 --no-cpu-baseline"
[ "$FULL" = "1" ] || ARGS="$ARGS --c4-small"
rm -rf gpurun_out/prof/c4_sq gpurun_out/prof/c4_sq2 gpurun_out/prof/c4_hbm
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/c4_sq -- python3 bench.py $ARGS > gpurun_out/prof/c4_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/prof/c4_sq2 -- python3 bench.py $ARGS > gpurun_out/prof/c4_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/prof/c4_hbm -- python3 bench.py $ARGS > gpurun_out/prof/c4_hbm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("c4_sq", "c4_sq2", "c4_hbm"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/prof/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "short" not in k and "solve_kernel" not in k: continue
            k = k[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_VALU", "FETCH_SIZE"): n[k] += 1
    for k, v in acc.items():
        print(d, k, "launches", n[k], {c: round(x / max(n[k], 1), 1) for c, x in v.items()})
PY
