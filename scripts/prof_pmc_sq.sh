cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd $R
for S in CHOLESKY CG; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/sq_$S -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --solver $S > gpurun_out/prof/sq_$S.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof/sq2_$S -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --solver $S > gpurun_out/prof/sq2_$S.log 2>&1
done
tail -2 gpurun_out/prof/sq_CG.log gpurun_out/prof/sq2_CG.log | cut -c1-300
