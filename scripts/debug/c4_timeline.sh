#!/bin/bash
# kernel timeline of the configs[3] legs: where an epoch's wall time goes between the kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4tl
IRSPACK_AMD_IALS_EIG_RATIO=${RATIO:-64} IRSPACK_AMD_IALS_MAX_CHUNKS=${CAP:-256} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --legs c4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c4_timeline_bench.log 2>&1
f=$(find /tmp/c4tl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/c4_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# the last 400 kernels: the final Cholesky epochs
for r in rows[-1200:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    e = (int(r['End_Timestamp']) - t0) / 1e6
    print(f"{s:12.3f} {e:12.3f} {e - s:9.3f} q={r.get('Queue_Id','?')} grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))} {r['Kernel_Name'][:90]}")
PY
tail -5 $GRAFT_REPO_ROOT/gpurun_out/c4_timeline.txt
