cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/prof
for K in 128 256; do
rm -rf gpurun_out/prof/pp_kt_$K
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pp_kt_$K -- python3 scripts/quick_ials.py --shape ml20m --K $K --solvers IALSPP --epochs 3 > gpurun_out/prof/pp_kt_$K.log 2>&1
tail -2 gpurun_out/prof/pp_kt_$K.log
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/prof/pp_kt_$K/*/*_kernel_trace.csv')[0]
by=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'ialspp' in r['Kernel_Name']:
        by[(r['Kernel_Name'][:50], r['Grid_Size_X'], r['VGPR_Count'], r['LDS_Block_Size'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in by.items(): print(k, len(v), sum(v)/len(v)/1e6, 'ms')
PY
done
