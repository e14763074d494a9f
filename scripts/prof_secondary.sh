# rocprofv3 passes over the kNN / evaluator / iALS++ paths (kernel trace + a PMC pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/sec_kt gpurun_out/prof/sec_pmc gpurun_out/prof/pp_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/sec_kt -- python3 scripts/quick_knn_eval.py > gpurun_out/prof/sec_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/sec_pmc -- python3 scripts/quick_knn_eval.py > gpurun_out/prof/sec_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pp_kt -- python3 scripts/quick_ials.py --shape ml20m --K 64 --solvers IALSPP --epochs 3 > gpurun_out/prof/pp_kt.log 2>&1
cat gpurun_out/prof/sec_kt/*/*_kernel_stats.csv | cut -c1-150 | head -8
cat gpurun_out/prof/pp_kt/*/*_kernel_stats.csv | cut -c1-150 | head -4
