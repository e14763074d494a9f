cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/knn_kt -- python3 scripts/quick_knn_eval.py > gpurun_out/prof/knn_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/knn_pmc -- python3 scripts/quick_knn_eval.py --skip-eval > gpurun_out/prof/knn_pmc.log 2>&1
cat gpurun_out/prof/knn_kt/*/*_kernel_stats.csv | cut -c1-200
