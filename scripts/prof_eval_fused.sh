cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
for K in 64 256; do
rm -rf gpurun_out/prof/ef_kt_$K
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ef_kt_$K -- python3 scripts/quick_eval_fused.py $K > gpurun_out/prof/ef_kt_$K.log 2>&1
grep "^K" gpurun_out/prof/ef_kt_$K.log
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof/ef_kt_$K/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
done
