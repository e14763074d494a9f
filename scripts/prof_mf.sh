# rocprofv3 kernel trace of the matrix-free CG epoch at K = 256 (per-kernel, per-grid durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/mf_kt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof/mf_kt -- python3 scripts/quick_ials.py --K ${1:-256} --solvers CG --epochs 2 > gpurun_out/prof/mf_kt.log 2>&1
python3 - <<'P'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/prof/mf_kt/*/*_kernel_trace.csv"))[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
rows = list(csv.DictReader(open(f)))
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    k = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
    a = agg[k]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{us / n:10.1f} us x {n:4d}  grid {grid:9d}  {name}")
P
