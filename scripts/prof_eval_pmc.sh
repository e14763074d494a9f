# counters of the fused iALS evaluator's kernels (K = 64, four calls over all ML-20M users); one pass per group
# (a pass with the TA_* counters hung on this pool in round 6 and is left out)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
rm -rf gpurun_out/prof/ef_sq gpurun_out/prof/ef_sq2 gpurun_out/prof/ef_tc1 gpurun_out/prof/ef_tc2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/ef_sq -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof/ef_sq2 -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_sq2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d gpurun_out/prof/ef_tc1 -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_tc1.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/prof/ef_tc2 -- python3 scripts/quick_eval_fused.py 64 > gpurun_out/prof/ef_tc2.log 2>&1
