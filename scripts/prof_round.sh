# Every rocprofv3 pass whose summary goes under profiles/ (one gpurun call).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/prof/kt gpurun_out/prof/fetch gpurun_out/prof/write gpurun_out/prof/sec_kt gpurun_out/prof/sec_pmc
bash scripts/prof_bench.sh > gpurun_out/prof_bench.log 2>&1
bash scripts/prof_pmc_sq.sh > gpurun_out/prof_pmc_sq.log 2>&1
bash scripts/prof_largek.sh > gpurun_out/prof_largek.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/sec_kt -- python3 scripts/quick_knn_eval.py > gpurun_out/prof/sec_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/sec_pmc -- python3 scripts/quick_knn_eval.py > gpurun_out/prof/sec_pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/sec_fetch -- python3 scripts/quick_knn_eval.py --skip-eval > gpurun_out/prof/sec_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/sec_write -- python3 scripts/quick_knn_eval.py --skip-eval > gpurun_out/prof/sec_write.log 2>&1
# short-row kernels (C4 shape at 1/5 scale, K = 128, CG and Cholesky)
rm -rf gpurun_out/prof/c4_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/c4_kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs c4 --c4-small > gpurun_out/prof/c4_kt.log 2>&1
ls gpurun_out/prof | head -40
# fused iALS evaluator (quick_eval_fused.py: 4 calls over all users at K = 64 and K = 256)
bash scripts/prof_eval_fused.sh > gpurun_out/prof/ef.log 2>&1
ls gpurun_out/prof | head -40
# iALS++ with 64-dim blocks at K = 128 (chained passes)
rm -rf gpurun_out/prof/pp_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pp_kt -- python3 scripts/quick_ials.py --shape ml20m --K 128 --solvers IALSPP --epochs 3 > gpurun_out/prof/pp_kt.log 2>&1
# iALS++ with one block at K = 64: the gradient-form (RESID) variant of the solve kernel (round 5)
rm -rf gpurun_out/prof/pp64_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/pp64_kt -- python3 scripts/quick_ials.py --shape ml20m --K 64 --solvers IALSPP --epochs 5 > gpurun_out/prof/pp64_kt.log 2>&1
