cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/eval_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/eval_kt -- python3 scripts/quick_knn_eval.py --skip-knn > gpurun_out/prof/eval_kt.log 2>&1
cat gpurun_out/prof/eval_kt/*/*_kernel_stats.csv | cut -c1-160 | head -8
