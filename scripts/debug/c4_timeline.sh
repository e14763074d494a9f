#!/bin/bash
# kernel timeline of the configs[3] legs (rocprofv3 kernel trace of `bench.py --legs c4`): where an
# epoch's wall time goes between the kernels and the streams.  Writes gpurun_out/c4_timeline.txt
# (start ms, end ms, duration ms, queue, grid, kernel) for the last 1200 kernels of the run.
# Environment switches of the library (IRSPACK_AMD_IALS_EIG_RATIO, ...) pass through.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --legs c4 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c4_timeline_bench.log 2>&1
f=$(find /tmp/c4tl -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/c4_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows[-1200:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6
    e = (int(r['End_Timestamp']) - t0) / 1e6
    print(f"{s:12.3f} {e:12.3f} {e - s:9.3f} q={r.get('Queue_Id','?')} grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))} {r['Kernel_Name'][:90]}")
PY
tail -3 $GRAFT_REPO_ROOT/gpurun_out/c4_timeline.txt
